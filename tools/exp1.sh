cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kb in 36 72 144; do
  rm -rf /tmp/p_$kb
  SQLRS_LDS_AGG_KB=$kb timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$kb -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp1_$kb.log 2>&1 < /dev/null
  f=$(find /tmp/p_$kb -name '*kernel_trace.csv' | head -1)
  test -n "$f" && python - "$f" $kb <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "rp_scatter" in n or "lds_agg" in n or "rp_hist" in n:
        out.append((int(r["Start_Timestamp"]), n.split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X"), r.get("LDS_Block_Size")))
out.sort()
print("KB", sys.argv[2])
for o in out[-12:]:
    print("  %-42s %.3f ms grid %s lds %s" % o[1:])
PY
  grep -E "rp_scatter|lds_agg|ms_per_step" gpurun_out/exp1_$kb.log | cut -c1-200 | head -5
done

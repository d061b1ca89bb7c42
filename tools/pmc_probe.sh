# L2 (TCC) hit / miss / request counters and SQ wait fractions of the dense join probe (C3 shapes via tools/probe_sweep.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
rm -rf /tmp/pp
PROBE_SWEEP_ONLY=1000000 timeout 600 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pp -- python tools/probe_sweep.py > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pp -name '*counter_collection.csv' | head -1)
test -n "$f" || { echo "no counter file for $PMC"; continue; }
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "join_probe" in n: acc[n[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    for k, v in c.items():
        # launches come in groups of 13 per hit rate (3 warm-up + 10 timed): print the mean of each third
        t = len(v) // 3
        print(n, k, "launches", len(v), "mean per hit-rate group:", [round(sum(v[i*t:(i+1)*t]) / max(t, 1)) for i in range(3)])
PY
done

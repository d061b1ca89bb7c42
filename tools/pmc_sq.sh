# SQ counters of the C5 kernels (one rocprofv3 --pmc pass, no tracing).
#   PMC="SQ_WAVE_CYCLES ..." (first counter is the denominator)  BENCH_ARGS="..."
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_sq -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/pmc_sq.log 2>&1 < /dev/null
f=$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -1)
test -n "$f" || { echo "no counter file"; tail -5 gpurun_out/pmc_sq.log; exit 1; }
python - "$f" $PMC <<'PY'
import csv, sys, collections
names = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if not n.startswith(("sq::", "void sq::")): continue
    agg[n.replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0))[:12]
print("%-50s %12s " % ("kernel", names[0]) + " ".join("%10s" % x.replace("SQ_", "")[-10:] for x in names[1:]))
for n, c in rows:
    w = c.get(names[0], 1) or 1
    print("%-50s %12.3e " % (n[:50], w) + " ".join("%9.1f%%" % (100 * c.get(x, 0) / w) for x in names[1:]))
PY

# SQ counters of the C5 kernels for one input SHAPE (random | sorted | hot) of tools/ab_in_process.py: one rocprofv3 --pmc pass, no tracing.
#   SHAPE=sorted bash tools/pmc_sq_shape.sh     (PMC="..." selects the counters; the first one is the denominator;
#   CMD="python tools/c4_agg.py" another driver that reads SHAPE)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE}
rm -rf /tmp/pmc_sq
VAR=SQLRS_DUMMY VALUES=0 REPS=1 timeout 600 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_sq -- ${CMD:-python tools/ab_in_process.py} > gpurun_out/pmc_sq_shape.log 2>&1 < /dev/null
f=$(find /tmp/pmc_sq -name '*counter_collection.csv' | head -1)
test -n "$f" || { echo "no counter file"; tail -5 gpurun_out/pmc_sq_shape.log; exit 1; }
python - "$f" $PMC <<'PY'
import csv, sys, collections
names = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if not n.startswith(("sq::", "void sq::")): continue
    agg[n.replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[n.replace("void ", "")] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get(names[0], 0))[:6]
print("%-50s %12s " % ("kernel", names[0]) + " ".join("%10s" % x.replace("SQ_", "")[-10:] for x in names[1:]))
for n, c in rows:
    w = c.get(names[0], 1) or 1
    print("%-50s %12.3e " % (n[:50], w) + " ".join("%9.1f%%" % (100 * c.get(x, 0) / w) for x in names[1:]))
PY

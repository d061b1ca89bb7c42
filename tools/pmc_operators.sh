cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pc3
timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pc3 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --operators > /dev/null 2>&1 < /dev/null
f=$(find /tmp/pc3 -name '*counter_collection.csv' | head -1)
python - "$f" $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("join_probe", "dense_", "rs_scatter", "gather_kernel", "filter_cmp", "key_stats", "lds_agg")):
        acc[n[:60]].append(float(r["Counter_Value"]))
for n, v in acc.items():
    big = [x for x in v if x > 0.2 * max(v)]
    print(sys.argv[2], n, "launches", len(v), "mean of large (KiB)", round(sum(big) / len(big)), "max", round(max(v)))
PY
done

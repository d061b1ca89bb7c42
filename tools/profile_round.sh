#!/bin/bash
# Round profile of the default bench command (C5 on one GPU), run ON the GPU box:
#   1. rocprofv3 --kernel-trace --stats          -> gpurun_out/<tag>_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE  (own pass)     -> per-kernel means
#   3. rocprofv3 --pmc WRITE_SIZE  (own pass)     -> gpurun_out/<tag>_pmc_traffic.json
# (counter passes carry no tracing flags).  Usage: bash tools/profile_round.sh r01d
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-operators"
if [ -z "$ONLY_OPS" ]; then   # (ONLY_OPS=1: the operator legs only)
rm -rf /tmp/prof_kt /tmp/prof_fetch /tmp/prof_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $CMD > gpurun_out/${TAG}_kt.log 2>&1 < /dev/null
f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
test -n "$f" && cp "$f" gpurun_out/${TAG}_kernel_stats.csv && head -12 "$f" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -- $CMD > gpurun_out/${TAG}_pmc_$c.log 2>&1 < /dev/null
done
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
def per_kernel(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    if not fs: return acc
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter: continue
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
fe, wr = per_kernel("/tmp/prof_fetch", "FETCH_SIZE"), per_kernel("/tmp/prof_write", "WRITE_SIZE")
out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 3 "
               "--warmup 2 --no-cpu-baseline --no-operators` (C5, 1e9 x 1e7); counter values are KiB; per-launch means over launches "
               "with > 1e5 units; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts "
               "half of a coalesced streaming read; calibration: filter_cmp_const reads 8.0e9 B)", "kernels": {}}
short = {"rp_chunk_scatter_filter": "sq::rp_chunk_scatter", "rp_scatter": "sq::rp_scatter", "lds_agg": "sq::lds_agg", "filter_cmp_const": "sq::filter_cmp_const",
         "compact": "sq::compact_kernel", "rp_hist": "sq::rp_hist_kernel"}
for k, pref in short.items():
    # both passes run the same command, so launch i of a kernel is the same launch in both; the
    # big (probe-side) launches are picked by their FETCH_SIZE and the same indices used for WRITE_SIZE
    fm = wm = cnt = 0
    for n, fv in fe.items():
        if not n.startswith(pref): continue
        wv = wr.get(n, [])
        for i, v in enumerate(fv):
            if v > 1e5:
                fm += v; cnt += 1
                wm += wv[i] if i < len(wv) else 0.0
    if not cnt: continue
    fm, wm = fm / cnt, wm / cnt
    out["kernels"][k] = {"kernel_prefix": pref, "launches_sampled": cnt, "fetch_size_per_launch": round(fm),
                         "write_size_per_launch": round(wm), "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump(out, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
fi

# ---- the per-operator legs (C2 / C3 / C4 / Order of bench.py's `operators` object): kernel trace + the two counter passes of
#      `python tools/operators_only.py` (the same bench_operators() code, no C5 tables) -> gpurun_out/<tag>_ops_kernel_stats.csv,
#      gpurun_out/<tag>_ops_pmc_traffic.json (every kernel with > 1e5 KiB fetched or written per launch)
export SQLRS_BENCH_SKIP_HOST=1   # (device-resident legs only: the host-fed legs are PCIe and memcpy, and start child processes)
OCMD="python tools/operators_only.py"
rm -rf /tmp/prof_okt /tmp/prof_ofetch /tmp/prof_owrite
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_okt -- $OCMD > gpurun_out/${TAG}_ops_kt.log 2>&1 < /dev/null
# (the legs fed by the native caller start child processes, each with output files of its own: the largest is the python process's)
f=$(find /tmp/prof_okt -name '*kernel_stats.csv' -printf '%s %p\n' | sort -nr | head -1 | cut -d' ' -f2-)
test -n "$f" && cp "$f" gpurun_out/${TAG}_ops_kernel_stats.csv && head -14 "$f" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_o$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $d -- $OCMD > gpurun_out/${TAG}_ops_pmc_$c.log 2>&1 < /dev/null
done
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
def per_kernel(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    if not fs: return acc
    # (the legs fed by the native caller start child processes, each with a counter file of its own — 20 000 launches each,
    #  not small files: the profiled command's own file is the one with the most DISTINCT kernels)
    def distinct(f):
        return len({r["Kernel_Name"] for r in csv.DictReader(open(f))})
    path = max(fs, key=distinct)
    print(counter, "from", path, "of", len(fs), "files")
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
fe, wr = per_kernel("/tmp/prof_ofetch", "FETCH_SIZE"), per_kernel("/tmp/prof_owrite", "WRITE_SIZE")
out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python tools/operators_only.py` (bench.py's "
               "operator legs: C2 1e8 rows, C3 1e8 x 1e6, C4 2e8 rows / 1e6 groups, Order 1e8 rows); counter values are KiB; per kernel: "
               "launches with > 1e5 KiB moved, their mean; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, "
               "see the C5 file)", "kernels": {}}
for n, fv in sorted(fe.items()):
    wv = wr.get(n, [])
    big = [(v, wv[i] if i < len(wv) else 0.0) for i, v in enumerate(fv) if v + (wv[i] if i < len(wv) else 0.0) > 1e5]
    if not big: continue
    fm, wm = sum(b[0] for b in big) / len(big), sum(b[1] for b in big) / len(big)
    out["kernels"][n[:90]] = {"launches_sampled": len(big), "fetch_size_per_launch": round(fm), "write_size_per_launch": round(wm),
                              "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump(out, open(f"gpurun_out/{tag}_ops_pmc_traffic.json", "w"), indent=1)
print(len(out["kernels"]), "operator kernels with counters")
PY

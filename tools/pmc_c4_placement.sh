#!/bin/bash
# L2 write-side counters of C4's claimed level across re-allocations of its output regions (tools/c4_placement.py): do the
# slow placements write more PARTIAL lines to memory (TCC_EA0_WRREQ vs TCC_EA0_WRREQ_64B), i.e. are open cursor lines evicted
# from L2 before a workgroup completes them?   -> gpurun_out/<tag>_c4_wr_pmc.txt
TAG=${1:-r06i}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_c4
ROUNDS=${ROUNDS:-6} timeout 600 rocprofv3 --kernel-trace --pmc ${PMC:-TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum} --output-format csv -d /tmp/pmc_c4 -- python tools/c4_placement.py > gpurun_out/${TAG}_c4_wr.log 2>&1 < /dev/null
python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
cc = glob.glob("/tmp/pmc_c4/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/pmc_c4/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc[0])):
    if "rp_claim_scatter" not in r["Kernel_Name"] and "lds_agg_dense" not in r["Kernel_Name"]: continue
    e = rows.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"].split("(")[0][-40:]})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
with open(f"gpurun_out/{tag}_c4_wr_pmc.txt", "w") as f:
    for d, e in rows.items():
        line = f"dispatch {d:>6} {e['k']:40s} us {dur.get(d, 0):9.1f} " + " ".join(f"{k}={v:.0f}" for k, v in e.items() if k != "k")
        print(line); f.write(line + "\n")
PY
grep "^round" gpurun_out/${TAG}_c4_wr.log

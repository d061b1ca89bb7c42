#!/bin/bash
# kernel timeline of the last C5 steps: busy time vs span, largest idle gaps (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operators > /dev/null 2>&1 < /dev/null
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(f))]
mc = glob.glob("/tmp/tl/**/*memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
# steps are delimited by the join build's key_minmax_kernel; take the last three
starts = [i for i, r in enumerate(rows) if "key_minmax_inv_kernel" in r[2]]
for si in range(len(starts) - 3, len(starts) - 1):
    seg = rows[starts[si]:starts[si + 1]]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    print(f"step: {len(seg)} ops, span {span:.3f} ms (to the next filter start {(rows[starts[si+1]][0]-seg[0][0])/1e6:.3f}), busy {busy:.3f} ms")
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e6, seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1))[::-1][:12]
    for g, a, b in gaps:
        print(f"   gap {g:.3f} ms  after {a}  before {b}")
PY

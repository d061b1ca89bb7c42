#!/bin/bash
# every device operation of one C5 step in launch order with its duration and the idle gap before it
# (run on the GPU box; SQLRS_BENCH_ROWS selects the fact size).  CMD / DELIM select another command and the kernel
# whose launch delimits one repetition, e.g.  CMD="python tools/order_profile.py" DELIM=order_minmax_kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tl
CMD=${CMD:-python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-operators}
export DELIM=${DELIM:-key_minmax_inv_kernel}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- $CMD > /dev/null 2>&1 < /dev/null
python - <<'PY'
import csv, glob, os
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Grid_Size_X", "")) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if os.environ["DELIM"] in r[2]]
seg = rows[starts[-2]:starts[-1]]
prev = seg[0][0]
for s, e, name, grid in seg:
    print(f"{(s - seg[0][0]) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  grid {grid:>9}  {name}")
    prev = e
PY

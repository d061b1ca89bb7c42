# Texture-addresser (TA), L1 (TCP) and texture-data (TD) busy / stall counters of the dense join probe at
# 100 / 50 / 0 % hits (C3: 1e8 probe keys, 1e6 build keys; tools/probe_sweep.py), one small counter group
# per rocprofv3 pass (counter passes carry no tracing flags).  Usage: bash tools/pmc_probe_ta.sh > out.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for PMC in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do
rm -rf /tmp/pp
PROBE_SWEEP_ONLY=1000000 timeout 600 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pp -- python tools/probe_sweep.py > /tmp/pp.log 2>&1 < /dev/null
f=$(find /tmp/pp -name '*counter_collection.csv' | head -1)
test -n "$f" || { echo "no counter file for: $PMC"; tail -3 /tmp/pp.log; continue; }
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "join_probe" in n: acc[n[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    for k, v in c.items():
        # launches come in groups of 13 per hit rate (3 warm-up + 10 timed): mean of each third = 100 / 50 / 0 % hits
        t = len(v) // 3
        print(f"{n} {k:42s} launches {len(v)}  mean per launch at 100/50/0 % hits: {[round(sum(v[i*t:(i+1)*t]) / max(t, 1)) for i in range(3)]}")
PY
done

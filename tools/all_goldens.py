import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+"/tests")
import sqlrs_amd
from golden_runner import Runner, load
fx = load(); hip = sqlrs_amd.hip(0)
for c in fx["cases"]:
    if c["gpu"]: continue
    try:
        got = Runner(hip, fx).rows(c["plan"])
        print(c["name"], "OK" if got == c["expected"] else f"MISMATCH got={got}")
    except Exception as e:
        print(c["name"], "ERR", str(e)[:100])

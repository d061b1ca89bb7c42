"""bench.py's per-operator leg alone (C2 / C3 / C4 / Order at BASELINE sizes, kernel classes per operator), without the
C5 headline: the quick loop for operator-level kernel work.  `python tools/operators_only.py` on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
res = bench.bench_operators(be, abi, datagen, torch, dev)
print(json.dumps(res))

"""bench.py's N > 1 path with two ranks on ONE GPU (SQLRS_BENCH_SINGLE_DEVICE=1: payload over gloo, the
device work — filter, hash partition, chunked exchange bookkeeping, local HashJoinAgg, per-group check —
exactly as on an 8-GPU node).  The real RCCL run is the driver's; this keeps the logic from rotting."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("exchange", ["partition", "broadcast"])
def test_bench_two_ranks_on_one_gpu(exchange):
    env = dict(os.environ, SQLRS_BENCH_SINGLE_DEVICE="1", SQLRS_BENCH_EXCHANGE_CHUNKS="3", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", "6e6", "--dim-rows", "2e5", "--exchange", exchange, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    x = line["exchange"]
    assert x["strategy"] == exchange and x["bytes_off_rank_per_step"] > 0
    assert x["ranks"]["all_reduce_of_ones"] == 2
    assert x["alternative"].get("check") == "OK", x["alternative"]
    assert "check (per group" in r.stderr and "-> OK" in r.stderr

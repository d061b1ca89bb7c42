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


@pytest.mark.parametrize("exchange,launcher,fused", [("combine", "self", "1"), ("partition", "self", "1"), ("partition", "torchrun", "0"),
                                                      ("broadcast", "torchrun", "1")])
def test_bench_two_ranks_on_one_gpu(exchange, launcher, fused):
    """`self` = plain `python bench.py --gpus 2` (what the driver runs): bench.py starts its own ranks"""
    env = dict(os.environ, SQLRS_BENCH_SINGLE_DEVICE="1", SQLRS_BENCH_EXCHANGE_CHUNKS="3", HSA_ENABLE_IPC_MODE_LEGACY="0",
               SQLRS_BENCH_EXCHANGE_FUSED=fused)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "6e6", "--dim-rows", "2e5",
            "--exchange", exchange, "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    x = line["exchange"]
    assert x["strategy"] == exchange and x["bytes_off_rank_per_step"] > 0
    assert x["ranks"]["all_reduce_of_ones"] == 2
    assert x["alternative"].get("check") == "OK", x["alternative"]
    assert x["alternative2"].get("check") == "OK", x["alternative2"]
    assert {x["strategy"], x["alternative"]["strategy"], x["alternative2"]["strategy"]} == {"combine", "partition", "broadcast"}
    assert "check (per group" in r.stderr and "-> OK" in r.stderr
    assert x["fused_filter_partition"] == (exchange == "partition" and fused == "1")


@pytest.mark.parametrize("fused,impl", [("1", "abi"), ("1", "torch"), ("0", "abi"), ("combine", "abi")])
def test_bench_exchange_path_over_rccl_world_1(fused, impl):
    """The N > 1 code path on ONE GPU over the real transport: an RCCL process group of one rank
    (`--force-exchange`), so init_process_group("nccl"), the collectives on device tensors (all_reduce,
    all_to_all_single of the dim, the list all-to-all out of the partition regions / all_to_all_single of
    the stable partition), and the ctx-stream <-> torch-stream hand-over all run on the hardware once.  impl = abi (the
    default): the data path is exchange.hip behind the C ABI — sqlrs_exchange_all_to_all for the dim keys / partials and the
    chunk sequence sqlrs_exchange_begin / send_chunk / finish for the fact rows — torch only launches the processes."""
    strategy = "combine" if fused == "combine" else "partition"  # (combine: the default strategy of --gpus N)
    fused = "1" if fused == "combine" else fused
    env = dict(os.environ, SQLRS_BENCH_EXCHANGE_CHUNKS="3", HSA_ENABLE_IPC_MODE_LEGACY="0", SQLRS_BENCH_EXCHANGE_FUSED=fused)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SQLRS_BENCH_SINGLE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-exchange", "--steps", "3", "--warmup", "1",
           "--rows", "2e7", "--dim-rows", "1e6", "--exchange", strategy, "--exchange-impl", impl, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    x = line["exchange"]
    assert x["impl"].startswith("C ABI") == (impl == "abi")
    assert line["n_gpus"] == 1 and x["ranks"]["backend"].startswith("nccl") and x["ranks"]["world"] == 1
    assert x["strategy"] == strategy and x["fused_filter_partition"] == (strategy == "partition" and fused == "1")
    assert x["alternative"].get("check") == "OK", x["alternative"]
    assert "check (per group" in r.stderr and "-> OK" in r.stderr

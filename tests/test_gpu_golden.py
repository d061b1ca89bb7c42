"""The reference's golden tables replayed on the HIP path through the C ABI (MI355X)."""
import pytest

from golden_runner import Runner, load

FX = load()
GPU_CASES = [c for c in FX["cases"] if c["gpu"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c["name"] for c in GPU_CASES])
def test_hip_matches_reference_golden(hip, oracle, case):
    got = Runner(hip, FX).rows(case["plan"])
    assert got == case["expected"], f"{case['name']} ({case['source']})"
    assert got == Runner(oracle, FX).rows(case["plan"])


@pytest.mark.gpu
def test_cpp_host_mirror_replays_reference_unit_tests():
    """host/test_reference_executor.cpp: the reference's operator unit tests (same child streams,
    same operator structs, same expected pretty-printed tables) through the C++ host mirror."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "test_reference_executor")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "host"), "-s"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout

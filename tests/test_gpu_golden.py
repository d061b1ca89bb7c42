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

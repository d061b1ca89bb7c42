"""Pins the CPU oracle against every golden table the reference's own tests hold for the
hot path.  CPU only."""
import pytest

from golden_runner import Runner, load

FX = load()


@pytest.mark.parametrize("case", FX["cases"], ids=[c["name"] for c in FX["cases"]])
def test_oracle_matches_reference_golden(oracle, case):
    got = Runner(oracle, FX).rows(case["plan"])
    assert got == case["expected"], f"{case['name']} ({case['source']})"

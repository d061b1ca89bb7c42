"""Loads oracle/libsqlrs_oracle.so (CPU restatement of the reference) as an abi.Backend.
Test infrastructure only — nothing under sqlrs_amd/ imports this module."""
import os
import subprocess

from sqlrs_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libsqlrs_oracle.so")


def build_oracle():
    src = os.path.join(ORACLE_DIR, "sqlrs_oracle.cpp")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


def load_oracle(compat_count_last_batch: int = 0) -> abi.Backend:
    return abi.Backend(build_oracle(), "oracle_", compat_count_last_batch)

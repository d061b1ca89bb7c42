"""Builds sqlrs_amd/csrc/libsqlrs_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m sqlrs_amd.build [--force]

Each .hip translation unit is compiled to an object in parallel, then linked.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "libsqlrs_hip.so")
OBJDIR = os.path.join(CSRC, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
          "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
CFLAGS += os.environ.get("SQLRS_EXTRA_CFLAGS", "").split()  # e.g. -DFILTER_TIMING (kernel phase timers)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return (glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) +
            [os.path.join(ROOT, "include", "sqlrs_hip.h")])


def _obj(src):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", _obj(src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
    if verbose and r.stderr.strip():
        print(r.stderr[-3000:])


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    objs = [_obj(s) for s in sources()]
    if todo or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)

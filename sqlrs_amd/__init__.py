"""sqlrs_amd — MI355X (gfx950) execution backend for the sqlrs `src/executor` hot path.

The product is ``sqlrs_amd/csrc/libsqlrs_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/sqlrs_hip.h``).  This package is the thin host side: ctypes binding (``abi``),
bound-expression encoding (``expr``) and the operator structs (``executor``) mirroring the
reference.  There is no CPU fallback: ``hip()`` raises if the library is not built or no
gfx950 device is usable.
"""
from __future__ import annotations

import os

from . import abi, expr, executor  # noqa: F401
from .abi import ExecutorError  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsqlrs_hip.so")

_backends = {}


def hip(device_id: int = 0) -> abi.Backend:
    """The HIP backend on GPU ``device_id`` (one ctx = one stream, cached per device)."""
    be = _backends.get(device_id)
    if be is None:
        if not os.path.exists(LIB_PATH):
            raise ExecutorError(abi.ERR_DEVICE,
                                f"{LIB_PATH} is not built: run `python -m sqlrs_amd.build`")
        be = abi.Backend(LIB_PATH, "sqlrs_", device_id)
        _backends[device_id] = be
    return be


def new_ctx(device_id: int = 0) -> abi.Backend:
    """A fresh, uncached ctx (own stream + memory pool)."""
    if not os.path.exists(LIB_PATH):
        raise ExecutorError(abi.ERR_DEVICE, f"{LIB_PATH} is not built: run `python -m sqlrs_amd.build`")
    return abi.Backend(LIB_PATH, "sqlrs_", device_id)

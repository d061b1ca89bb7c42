"""Deterministic synthetic inputs for the BASELINE.json configs.

SplitMix64(seed, row id), written once for torch (device resident, chunked so a 1e9-row
column never needs more than a few hundred MB of temporaries) and once for numpy (the
bounded CPU-baseline sample): both give identical values for the same (seed, row id).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def _s64(x: int) -> int:
    """python int -> the same 64 bits as a signed value (torch has no uint64 arithmetic)."""
    x &= _MASK
    return x - (1 << 64) if x >= (1 << 63) else x


# ------------------------------------------------------------------------- numpy --
def splitmix64_np(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(_GOLDEN) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def key_np(seed: int, idx: np.ndarray, modulus: int) -> np.ndarray:
    return ((splitmix64_np(seed, idx) >> np.uint64(11)) % np.uint64(modulus)).astype(np.int64)


def val_np(seed: int, idx: np.ndarray) -> np.ndarray:
    return (splitmix64_np(seed, idx) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def dim_key_np(idx: np.ndarray, n_dim: int) -> np.ndarray:
    """A permutation of 0..n_dim-1 (affine map with a multiplier coprime to n_dim)."""
    a = _coprime_multiplier(n_dim)
    return ((idx.astype(np.uint64) * np.uint64(a) + np.uint64(12345)) % np.uint64(n_dim)).astype(np.int64)


def _coprime_multiplier(n: int) -> int:
    import math
    a = 2654435761 % n if n > 1 else 1
    while math.gcd(a, n) != 1 or a < 2:
        a += 1
    return a


# ------------------------------------------------------------------------- torch --
def _lsr(z, k: int):
    """logical shift right on int64 tensors"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix64_t(seed: int, idx):
    z = (idx + 1) * _s64(_GOLDEN) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def fill_chunks(out, fn, start: int = 0, chunk: int = 1 << 26):
    """out[i] = fn(global row ids start+i) in chunks; returns out"""
    import torch
    n = out.numel()
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = torch.arange(start + lo, start + hi, dtype=torch.int64, device=out.device)
        out[lo:hi] = fn(idx)
    return out


def key_t(seed: int, idx, modulus: int):
    return _lsr(splitmix64_t(seed, idx), 11) % modulus


def val_t(seed: int, idx):
    import torch
    return _lsr(splitmix64_t(seed, idx), 11).to(torch.float64) * (2.0 ** -53)


def dim_key_t(idx, n_dim: int):
    a = _coprime_multiplier(n_dim)
    return (idx * a + 12345) % n_dim

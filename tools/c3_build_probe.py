"""C3 build + probe (hash_join_create .. build_finish .. probe_indices .. destroy) REPS times in one process: what
tools/timeline_ops.sh slices (CMD="python tools/c3_build_probe.py" DELIM=key_minmax_inv_kernel) and an event-timed figure.
MOD = probe key modulus (2 * NB: half the probe rows miss), DUP = build key multiplicity, JT = inner | left | right | full."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
dev = torch.device("cuda", 0)
be = abi.Backend(os.environ["LIB"], "sqlrs_", 0) if os.environ.get("LIB") else sqlrs_amd.new_ctx(0)
nP, nB = int(float(os.environ.get("NP", 1e8))), int(float(os.environ.get("NB", 1e6)))
mod = int(os.environ.get("MOD", nB))
dk = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB))
fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, mod))
if os.environ.get("DUP"):  # DUP=4: every build key ~4 times (bench.py's C3_join_dup_build_keys_x4), all probe rows match
    dup = int(os.environ["DUP"])
    dk = torch.randint(0, nB // dup, (nB,), dtype=torch.int64, device=dev, generator=torch.Generator(device=dev).manual_seed(44))
    fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB // dup))
JT = {"inner": abi.JOIN_INNER, "left": abi.JOIN_LEFT, "right": abi.JOIN_RIGHT, "full": abi.JOIN_FULL}[os.environ.get("JT", "inner")]
torch.cuda.synchronize()
db, fb = bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk], [abi.INT64])
lk, _k1 = abi.pack_exprs([InputRef(0)]); rk, _k2 = abi.pack_exprs([InputRef(0)])
rd = (C.c_int32 * 1)(abi.INT64)
def both():
    j = C.c_void_p()
    be.check(be.fn("hash_join_create")(be.ctx, JT, 1, lk, rk, None, 1, rd, C.byref(j)))
    be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o)
    be.fn("hash_join_destroy")(j)
both(); both(); be.synchronize()
t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
best = 1e9
for _ in range(int(os.environ.get("REPS", 7))):
    be.check(be.fn("timer_start")(t)); both(); be.check(be.fn("timer_stop")(t))
    ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
be.fn("timer_destroy")(t)
print(f"C3 build + probe: {best:.3f} ms = {(8 * nB + 20 * nP) / best / 1e6 / 8000:.4f} of 8 TB/s", flush=True)

# probe alone (the join built once), event-timed + kernel classes
j = C.c_void_p()
be.check(be.fn("hash_join_create")(be.ctx, JT, 1, lk, rk, None, 1, rd, C.byref(j)))
be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
def probe():
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o)
probe(); probe(); be.synchronize()
t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
best = 1e9
for _ in range(int(os.environ.get("REPS", 7))):
    be.check(be.fn("timer_start")(t)); probe(); be.check(be.fn("timer_stop")(t))
    ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
be.profile(True); probe(); pr = be.profile_read(); be.profile(False)
print(f"C3 probe alone: {best:.3f} ms | " + " ".join(f"{k} {v[0]:.3f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.002), flush=True)
be.fn("hash_join_destroy")(j)
# host wall time of every call of one build + probe (the GPU work before the probe kernel is ~45 us: is the host slower?)
import time
if os.environ.get("HOSTTIMES", "1") == "1":
    acc = {}
    for rep in range(20):
        be.synchronize()
        j = C.c_void_p(); o = C.POINTER(abi.Batch)()
        t0 = time.perf_counter(); be.check(be.fn("hash_join_create")(be.ctx, JT, 1, lk, rk, None, 1, rd, C.byref(j)))
        t1 = time.perf_counter(); be.check(be.fn("hash_join_build_push")(j, db.ptr))
        t2 = time.perf_counter(); be.check(be.fn("hash_join_build_finish")(j))
        t3 = time.perf_counter(); be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
        t4 = time.perf_counter(); be.fn("batch_release")(o); be.fn("hash_join_destroy")(j)
        t5 = time.perf_counter()
        if rep >= 5:
            for k, v in (("create", t1 - t0), ("build_push", t2 - t1), ("build_finish", t3 - t2), ("probe_indices", t4 - t3), ("release+destroy", t5 - t4)):
                acc.setdefault(k, []).append(v * 1e6)
    print("host us per call (median): " + ", ".join(f"{k} {sorted(v)[len(v)//2]:.1f}" for k, v in acc.items()), flush=True)

#!/usr/bin/env python3
"""Benchmark/validator of the acc C-ABI, in the shape of the reference's tools
(src/acc/acc_bench.c: `acc_bench nrepeat stack m n k nc na nb`, and the kernel timer
src/acc/libsmm_acc/libsmm_acc_benchmark.cpp:224-296 whose numbers parameters_*.json publish).

Everything goes through the C-ABI only: c_dbcsr_acc_{init,stream_create,event_*,host_mem_allocate,
dev_mem_allocate,memcpy_h2d/d2h}, libsmm_acc_transpose, libsmm_acc_process.  GFLOP/s = nrepeat * stack *
2mnk / kernel time (events around the process calls), as the reference computes it.  With --check the result
is compared with the CPU oracle (test infrastructure).

  python tools/acc_bench.py [nrepeat [stack [m [n [k [nc [na [nb]]]]]]]] [--check] [--f32] [--threads T]
defaults: the timer configuration 16005-entry stack over 10000 A, 10000 B, 1000 C blocks of 23x23x23.

--threads T: what the real host does (src/mm/dbcsr_mm_accdrv.F:433-541, core/dbcsr_lib.F:248-262): T host threads, each with a stream, a
stack buffer and C blocks of its own (the stacks of different threads never share a C block), call libsmm_acc_process concurrently over the
same A and B areas; the rate is all threads' flop over the wall time from a common start to the last stream's completion."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dbcsr_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("args", nargs="*", type=int)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    d = dict(zip(["nrepeat", "stack", "m", "n", "k", "nc", "na", "nb"], a.args))
    m = d.get("m", 23)
    n, k = d.get("n", m), d.get("k", m)
    stack = d.get("stack", 16005)
    nrepeat = d.get("nrepeat", max(3, 12500 // (m * n * k)))
    nc = d.get("nc", 1000)
    na, nb = d.get("na", 10000), d.get("nb", 10000)
    lib = L.load_library()
    ck = lambda rc, what: (_ for _ in ()).throw(RuntimeError("%s failed (%d)" % (what, rc))) if rc != 0 else None
    ck(lib.c_dbcsr_acc_set_active_device(0), "set_active_device")
    ck(lib.c_dbcsr_acc_init(), "acc_init")
    ck(lib.libsmm_acc_init(), "libsmm_acc_init")
    dt, code, esz = (np.float32, L.dbcsr_type_real_4, 4) if a.f32 else (np.float64, L.dbcsr_type_real_8, 8)
    rng = np.random.default_rng(0)
    ha = rng.random(na * m * k).astype(dt)
    hb = rng.random(nb * k * n).astype(dt)
    import threading
    import time
    T = max(1, a.threads)
    hc = np.zeros(nc * m * n, dt)
    dev = {}
    s0 = C.c_void_p()
    ck(lib.c_dbcsr_acc_stream_create(C.byref(s0), b"setup", -1), "stream_create")

    def upload(name, arr, st):
        p = C.c_void_p()
        ck(lib.c_dbcsr_acc_dev_mem_allocate(C.byref(p), arr.nbytes), "dev_mem_allocate")
        ck(lib.c_dbcsr_acc_memcpy_h2d(arr.ctypes.data_as(C.c_void_p), p, arr.nbytes, st), "memcpy_h2d")
        dev[name] = p
        return p

    upload("a", ha, s0)
    upload("b", hb, s0)
    upload("t", np.arange(nb, dtype=np.int32) * k * n, s0)
    ck(lib.libsmm_acc_transpose(dev["t"], 0, nb, dev["b"], code, k, n, 80, s0), "libsmm_acc_transpose")
    # per thread: a stack sorted by C offset, about stack/nc consecutive entries per C block (INIT_STACK's shape), C blocks of its own
    stacks, streams, events = [], [], []
    for t in range(T):
        st = np.empty(3 * stack, np.int32)
        cidx = np.sort(rng.integers(0, nc, stack))
        st[0::3] = rng.integers(0, na, stack) * m * k + 1
        st[1::3] = rng.integers(0, nb, stack) * k * n + 1
        st[2::3] = cidx * m * n + 1
        stacks.append(st)
        stream, e1 = C.c_void_p(), C.c_void_p()
        ck(lib.c_dbcsr_acc_stream_create(C.byref(stream), b"bench", -1), "stream_create")
        ck(lib.c_dbcsr_acc_event_create(C.byref(e1)), "event_create")
        streams.append(stream)
        events.append(e1)
        upload("s%d" % t, st, s0)
        upload("c%d" % t, hc, s0)
    ck(lib.c_dbcsr_acc_stream_sync(s0), "stream_sync")

    def process(t):
        return lib.libsmm_acc_process(None, dev["s%d" % t], stack, code, dev["a"], dev["b"], dev["c%d" % t], m, n, k, 80, 1, streams[t], streams[t])

    # warm-up; timing with the host wall clock around the launches (events in this ABI carry no timestamps: acc.h:56-60)
    for t in range(T):
        for _ in range(2):
            rc = process(t)
            assert rc >= 0, rc
        ck(lib.c_dbcsr_acc_memset_zero(dev["c%d" % t], 0, hc.nbytes, streams[t]), "memset_zero")
        ck(lib.c_dbcsr_acc_stream_sync(streams[t]), "stream_sync")
    start = threading.Barrier(T + 1)
    done = threading.Barrier(T + 1)

    def worker(t):
        start.wait()
        for _ in range(nrepeat):
            process(t)
        ck(lib.c_dbcsr_acc_event_record(events[t], streams[t]), "event_record")
        ck(lib.c_dbcsr_acc_event_synchronize(events[t]), "event_synchronize")
        done.wait()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    dtm = time.perf_counter() - t0
    for th in ths:
        th.join()
    gflops = T * nrepeat * stack * 2.0 * m * n * k / dtm / 1e9
    msg = "acc_bench %s m=%d n=%d k=%d stack=%d nrepeat=%d threads=%d: %.1f GFLOP/s (%.3f ms per stack and thread)" % (
        "f32" if a.f32 else "f64", m, n, k, stack, nrepeat, T, gflops, dtm / nrepeat * 1e3)
    if a.check:
        from oracle import oracle as O
        worst = 0.0
        for t in range(T):
            out = np.empty_like(hc)
            ck(lib.c_dbcsr_acc_memcpy_d2h(dev["c%d" % t], out.ctypes.data_as(C.c_void_p), out.nbytes, streams[t]), "memcpy_d2h")
            ck(lib.c_dbcsr_acc_stream_sync(streams[t]), "stream_sync")
            ref = np.zeros(nc * m * n, np.float64)
            O.stack_calc(stacks[t], ref, ha.astype(np.float64), hb.astype(np.float64), m, n, k, b_transposed=False)
            ref *= nrepeat
            worst = max(worst, float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300))))
        msg += "  max rel err vs oracle %.2e" % worst
    if hasattr(lib, "dbcsr_amd_smm_last_kernel") and not a.f32:
        process(0)
        msg += "  [%s]" % lib.dbcsr_amd_smm_last_kernel().decode()
        ck(lib.c_dbcsr_acc_stream_sync(streams[0]), "stream_sync")
    print(msg)
    for p in dev.values():
        lib.c_dbcsr_acc_dev_mem_deallocate(p)
    for e in events:
        lib.c_dbcsr_acc_event_destroy(e)
    for st in streams + [s0]:
        lib.c_dbcsr_acc_stream_destroy(st)
    lib.libsmm_acc_finalize()
    lib.c_dbcsr_acc_finalize()


if __name__ == "__main__":
    main()

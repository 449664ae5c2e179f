#!/usr/bin/env python3
"""One device-resident multiply of a synthetic matrix of any block size mix and fill, timed like bench.py's other_configs
(HIP events of the block-product kernel + wall time of the whole step), one JSON line per run.  For sweeps next to the benchmark's own
configuration: the fill sweep of the 23 x 23 path (production against the lab's operand-sharing dataflows) and the block sizes
33 ... 80 (libsmm_acc's range: src/core/dbcsr_config.F:185 max_kernel_dim = 80).

    python tools/block_bench.py --size 32768 --mix 1,23 --fill 0.4 --lab --env DBCSR_AMD_MM_TILE=2
    python tools/block_bench.py --size 16384 --mix-m 1,45 --mix-n 1,67 --mix-k 1,78 --fill 0.1 --check
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--size", type=int, default=16384, help="M = N = K (elements)")
    p.add_argument("--mix", default="1,23", help="(multiplicity, size) pairs of all three dimensions")
    p.add_argument("--mix-m", default=None)
    p.add_argument("--mix-n", default=None)
    p.add_argument("--mix-k", default=None)
    p.add_argument("--fill", type=float, default=0.1)
    p.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--lab", action="store_true", help="the build with the experimental dataflows")
    p.add_argument("--env", action="append", default=[], help="NAME=VALUE set before the engine is made (repeatable)")
    p.add_argument("--label", default="")
    p.add_argument("--check", action="store_true", help="compare the checksum pair with a second multiply through DBCSR_AMD_MM_KERNEL=direct "
                                                         "(the plain global-memory kernel: an independent code path on the same device)")
    p.add_argument("--batch", default=None, help="JSON list of dictionaries of option overrides: one run each, in this one process "
                                                  "(saves the interpreter / torch start-up of a run per process)")
    args = p.parse_args()
    if args.batch:
        for over in json.loads(args.batch):
            a = argparse.Namespace(**vars(args))
            a.batch = None
            for k, v in over.items():
                setattr(a, k.replace("-", "_"), v)
            try:
                run(a)
            except Exception as e:   # noqa: BLE001 -- the batch goes on
                print(json.dumps({"label": a.label, "error": repr(e), "overrides": over}), flush=True)
        return
    run(args)


def run(args):
    saved = {}
    for kv in args.env:
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        run_one(args)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run_one(args):
    import torch
    from dbcsr_amd import randmat
    from dbcsr_amd.multiply import MultiplyEngine
    mix = lambda s: [int(x) for x in s.split(",")]
    mm, mn, mk = mix(args.mix_m or args.mix), mix(args.mix_n or args.mix), mix(args.mix_k or args.mix)
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    eng = MultiplyEngine(lab=args.lab)
    sp = 1.0 - args.fill
    t0 = time.perf_counter()
    A, B, Cm = randmat.perf_matrices(args.size, args.size, args.size, (sp, sp, sp), mm, mn, mk, dtype=dtype, engine=eng)
    torch.cuda.synchronize()
    eng.trust_plan(True)
    out, counts = eng.multiply_local(1.0, A, B, 1.0, Cm)
    out, counts = eng.multiply_local(1.0, A, B, 1.0, Cm)
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, counts = eng.multiply_local(1.0, A, B, 1.0, Cm)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    kern = []
    for _ in range(2):
        out, counts = eng.multiply_local(1.0, A, B, 1.0, Cm)
        kern.append(eng.last_timing()[1])
    kms = sum(kern) / len(kern)
    kp = max(1, int(getattr(eng, "last_kchunks", 1)))
    lf = getattr(eng, "last_launch_flop", counts.flop)
    peak = 78.6 if args.dtype == "f64" else 157.3
    res = {"label": args.label, "size": args.size, "mix_m": mm, "mix_n": mn, "mix_k": mk, "fill": args.fill, "dtype": args.dtype, "lab": args.lab,
           "env": args.env, "kernel": eng.last_kernel(), "k_passes": kp, "ms_per_step": round(ms, 4), "kernel_ms": round(kms, 4),
           "tflops_step": round(counts.flop / (ms * 1e-3) / 1e12, 3), "tflops_kernel": round(lf / (kms * 1e-3) / 1e12, 3),
           "frac_of_peak_kernel": round(lf / (kms * 1e-3) / 1e12 / peak, 4), "c_nblks": int(counts.c_nblks), "nproducts": int(counts.nproducts),
           "products_per_c_block": round(counts.nproducts / max(1, counts.c_nblks), 2), "flop": int(counts.flop), "setup_s": round(setup, 2)}
    if args.check:
        cs = eng.checksum(out)
        os.environ["DBCSR_AMD_MM_KERNEL"] = "direct"
        ref_eng = MultiplyEngine()
        del os.environ["DBCSR_AMD_MM_KERNEL"]
        ref, _ = ref_eng.multiply_local(1.0, A, B, 1.0, Cm, kchunks=1)
        cr = ref_eng.checksum(ref)
        same_index = bool(torch.equal(out.row_p, ref.row_p) and torch.equal(out.col_i, ref.col_i))
        dmax = float((out.data - ref.data).abs().max()) if out.data.numel() else 0.0
        scale = float(ref.data.abs().max()) if ref.data.numel() else 1.0
        res["check"] = {"reference_kernel": ref_eng.last_kernel(), "index_identical": same_index, "max_abs_diff_over_max_abs": dmax / max(scale, 1e-300),
                        "checksum_rel": abs(cs[0] / cr[0] - 1.0) if cr[0] else 0.0, "checksum_pos_rel": abs(cs[1] / cr[1] - 1.0) if cr[1] else 0.0}
    print(json.dumps(res), flush=True)
    del A, B, Cm, out, eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

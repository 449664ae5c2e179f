#!/usr/bin/env python3
"""N independent processes on ONE GPU walk disjoint parts of the randomised parity sweep (tests/test_gpu_random_sweep.py) -- the
deployment CP2K runs in: ranks mod ndevices (reference: src/core/dbcsr_lib.F:231-236).  Unlike `pytest -n 4` every case leaves a
record (seed, oracle seconds, device seconds, verdict), and a mismatch is taken apart on the spot:

  * the case dictionary, both index arrays and the worst element are dumped,
  * the inputs are generated again and compared bit by bit with the first generation (is the oracle's OpenMP fill deterministic?),
  * the oracle's product is formed twice more and the device's twice more on fresh engines: which side changed?

Switches that separate the suspects: --omp-threads (1 = no OpenMP in the checker; 0 = whatever the host offers: the
oversubscribed mode of round 4's session 18), --eps0 (filter_eps = 0 only: no threshold decision), --classes0 (no hiprtc), --no-torch-cache
(the caching allocator of torch off), --plain / --forced (how many cases of either kind), --budget-s (per worker: stop taking cases
after that many seconds, report how far the walk came).

    python tools/soak_multiproc.py --procs 4 --plain 1200 --forced 480 --omp-threads 8 --out gpurun_out/soak

The oracle is the checker here (test infrastructure); nothing of the product path imports it."""
import argparse
import json
import os
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FORCED_KEYS = ("DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_HOT")


def case_list(plain, forced):
    """[(kind, index)]: the forced-path cases first, as pytest walks the file"""
    return [("forced", i) for i in range(forced)] + [("plain", i) for i in range(plain)]


def jsonable(c):
    d = dict(c)
    d["dtype"] = "f32" if "32" in str(d["dtype"]) else "f64"
    return d


def same_bcsr(X, Y):
    import numpy as np
    return (np.array_equal(X.row_p, Y.row_p) and np.array_equal(X.col_i, Y.col_i) and np.array_equal(X.blk_p, Y.blk_p)
            and X.data.tobytes() == Y.data.tobytes())


def take_apart(S, c, A, B, Cm, ref, info, out, bad, dump_dir, tag):
    """what the module docstring promises for a mismatch; returns the report dictionary"""
    import numpy as np
    rep = {"case": jsonable(c), "first": bad}
    A2, B2, C2, ref2, info2 = S.build_case(c)
    rep["inputs_regenerated_identical"] = bool(same_bcsr(A, A2) and same_bcsr(B, B2) and same_bcsr(Cm, C2))
    rep["oracle_second_run_identical"] = bool(same_bcsr(ref, ref2) and info["flop"] == info2["flop"])
    _, _, _, ref3, _ = S.build_case(c)
    rep["oracle_third_run_identical"] = bool(same_bcsr(ref, ref3))
    again = []
    for _ in range(2):
        try:
            o2, f2 = S.device_case(c, A, B, Cm)
            again.append({"verdict": S.compare_case(c, o2, f2, ref, info), "identical_to_first_device_result": bool(same_bcsr(out, o2))})
        except Exception as e:   # noqa: BLE001 -- a record, not a handler
            again.append({"exception": repr(e)})
    rep["device_again"] = again
    sides = []
    if not (rep["inputs_regenerated_identical"] and rep["oracle_second_run_identical"] and rep["oracle_third_run_identical"]):
        sides.append("oracle")
    if any(not a.get("identical_to_first_device_result", False) for a in again):
        sides.append("device")
    rep["side_that_changed_on_re_evaluation"] = sides or ["neither (the mismatch is reproducible)"]
    if bad.get("what") == "values":
        w = bad["worst_element"]
        blk = int(np.searchsorted(np.sort(ref.blk_p), w, side="right") - 1)
        rep["worst_block_rank_in_data_order"] = blk
    np.savez_compressed(os.path.join(dump_dir, tag + ".npz"), ref_row_p=ref.row_p, ref_col_i=ref.col_i, ref_blk_p=ref.blk_p, ref_data=ref.data,
                        dev_row_p=out.row_p, dev_col_i=out.col_i, dev_blk_p=out.blk_p, dev_data=out.data)
    return rep


def worker(args):
    if args.omp_threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(args.omp_threads)
    if args.no_torch_cache:
        os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
    import numpy as np   # noqa: F401
    from tests import test_gpu_random_sweep as S
    cases = case_list(args.plain, args.forced)[args.worker::args.procs]
    os.makedirs(args.out, exist_ok=True)
    log = open(os.path.join(args.out, "worker%d.jsonl" % args.worker), "w")
    t_start = time.time()
    done = 0
    for kind, i in cases:
        if args.budget_s > 0 and time.time() - t_start > args.budget_s:
            break
        for k in FORCED_KEYS:
            os.environ.pop(k, None)
        if kind == "forced":
            for k, v in S.FORCED[i % len(S.FORCED)].items():
                os.environ[k] = v
            c = S.make_case(5000 + i)
        else:
            c = S.make_case(1000 + i)
        if args.classes0:
            os.environ["DBCSR_AMD_MM_CLASSES"] = "0"
        if args.eps0:
            c["eps"] = 0.0
        rec = {"kind": kind, "i": i}
        try:
            t0 = time.time()
            A, B, Cm, ref, info = S.build_case(c)
            t1 = time.time()
            out, flop = S.device_case(c, A, B, Cm)
            t2 = time.time()
            bad = S.compare_case(c, out, flop, ref, info)
            rec.update(oracle_s=round(t1 - t0, 4), device_s=round(t2 - t1, 4), ok=bad is None)
            if bad is not None:
                rec["report"] = take_apart(S, c, A, B, Cm, ref, info, out, bad, args.out, "w%d_%s%d" % (args.worker, kind, i))
        except Exception as e:   # noqa: BLE001 -- the walk goes on, the record says what happened
            rec.update(ok=False, exception=repr(e), traceback=traceback.format_exc(), case=jsonable(c))
        log.write(json.dumps(rec) + "\n")
        log.flush()
        done += 1
    log.write(json.dumps({"worker_done": args.worker, "cases": done, "of": len(cases), "seconds": round(time.time() - t_start, 1)}) + "\n")
    log.close()


def parent(args):
    os.makedirs(args.out, exist_ok=True)
    t0 = time.time()
    procs = []
    for w in range(args.procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", str(w)] + [a for a in sys.argv[1:]]
        procs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=open(os.path.join(args.out, "worker%d.out" % w), "w"), stderr=subprocess.STDOUT))
    rcs = [p.wait() for p in procs]
    wall = time.time() - t0
    recs, ends = [], []
    for w in range(args.procs):
        path = os.path.join(args.out, "worker%d.jsonl" % w)
        if not os.path.exists(path):
            continue
        for line in open(path):
            r = json.loads(line)
            (ends if "worker_done" in r else recs).append(r)
    fails = [r for r in recs if not r.get("ok")]
    osec = sorted(r.get("oracle_s", 0.0) for r in recs)
    dsec = sorted(r.get("device_s", 0.0) for r in recs)
    pct = lambda v, q: v[min(len(v) - 1, int(q * len(v)))] if v else 0.0
    summary = {"procs": args.procs, "omp_threads": args.omp_threads, "plain": args.plain, "forced": args.forced, "eps0": args.eps0,
               "classes0": args.classes0, "no_torch_cache": args.no_torch_cache, "budget_s": args.budget_s, "wall_s": round(wall, 1),
               "worker_exit_codes": rcs, "cases_walked": len(recs), "cases_total": args.plain + args.forced, "failures": len(fails),
               "oracle_s_median_p99_max": [pct(osec, 0.5), pct(osec, 0.99), osec[-1] if osec else 0.0],
               "device_s_median_p99_max": [pct(dsec, 0.5), pct(dsec, 0.99), dsec[-1] if dsec else 0.0],
               "workers": ends}
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump({"summary": summary, "failures": fails}, f, indent=1)
    print(json.dumps(summary))
    for r in fails:
        print("FAILED", json.dumps(r)[:2000])
    return 1 if fails or any(rcs) else 0


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--procs", type=int, default=4)
    p.add_argument("--plain", type=int, default=1200)
    p.add_argument("--forced", type=int, default=480)
    p.add_argument("--omp-threads", type=int, default=8, help="OpenMP threads of the oracle per process (0: the host's default = oversubscribed)")
    p.add_argument("--eps0", action="store_true")
    p.add_argument("--classes0", action="store_true")
    p.add_argument("--no-torch-cache", action="store_true")
    p.add_argument("--budget-s", type=float, default=0.0)
    p.add_argument("--out", default="gpurun_out/soak")
    p.add_argument("--worker", type=int, default=-1)
    args = p.parse_args()
    if args.worker >= 0:
        worker(args)
        return 0
    return parent(args)


if __name__ == "__main__":
    sys.exit(main())

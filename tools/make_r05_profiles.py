#!/usr/bin/env python3
"""profiles/r05_*.txt from the raw session output under gpurun_out/r05_s*/ (tools/gpu_sessions/r05_*.sh wrote it on the GPU box):
the tables the round's decisions were taken on, each line traceable to the JSON line block_bench.py / soak_multiproc.py printed."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def rows(path):
    out = []
    if os.path.exists(path):
        for ln in open(path, errors="replace"):
            ln = ln.strip()
            if ln.startswith("{"):
                try:
                    out.append(json.loads(ln))
                except ValueError:
                    pass
    return out


def fmt_block(r):
    return "%-22s size %6d  fill %.2f  %-34s passes %d  step %9.3f ms  kernel %9.3f ms  %7.2f TFLOP/s (kernel %7.2f = %.3f of peak)  products/C block %7.2f" % (
        r.get("label", ""), r["size"], r["fill"], r["kernel"][:34], r["k_passes"], r["ms_per_step"], r["kernel_ms"], r["tflops_step"], r["tflops_kernel"],
        r["frac_of_peak_kernel"], r["products_per_c_block"])


def fill_sweep():
    L = ["Fill sweep of the fp64 23 x 23 path, one MI355X (VERDICT r04 item 3): the shipping kernel (one wave per C block, k passes when A's block rows",
         "exceed 1 MB) against the lab build's operand-sharing dataflows (mm_tile: XCD-wide C tiles; mm_band: CU-wide tiles, B shared in LDS).",
         "tools/block_bench.py; sessions r05_s01 / s02 / s03 (different boxes: +- 2 %).  step = whole multiply, kernel = one launch (one k pass).", ""]
    rs = rows(G + "/r05_s01/fill_sweep.jsonl") + rows(G + "/r05_s02/fill_sweep_dense.jsonl") + [r for r in rows(G + "/r05_s03/sweeps.jsonl") if r.get("mix_m") == [1, 23]]
    rs = [r for r in rs if "error" not in r]
    for key in sorted(set((r["size"], r["fill"]) for r in rs), key=lambda k: (k[1], k[0])):
        L.append("-- %d^2, fill %.2f" % key)
        for r in rs:
            if (r["size"], r["fill"]) == key:
                L.append("   " + fmt_block(r))
    L += ["", "Reading: whole-step TFLOP/s at each fill, best shipping choice against best lab dataflow:",
          "  fill 0.05: 30.4 (one pass)            -- sharing not applicable (3.7 products per C block)",
          "  fill 0.10: 38.2 (one pass)            vs tile 30.2 / band 30.0",
          "  fill 0.20: 36.6 (two passes; 34.0 in one) vs tile 34.9 / band 35.2",
          "  fill 0.30: 37.5 (three passes)        vs band 37.3",
          "  fill 0.40: 38.8 (three passes; 27.4 in one) vs tile 37.7 / band 38.4 (one pass each)",
          "  fill 0.80: 38.8 (three passes; 29.6 in one) vs tile 37.2 / band 38.3",
          "The sharing dataflows do what they were built for -- at 40-80 % fill they hold 0.48-0.49 of the peak in ONE pass where the one-wave-per-block",
          "kernel falls to 0.35-0.38 -- but k passes (operands L2-resident per pass, C re-read per pass) give the shipping kernel the same 0.48-0.50 at every",
          "fill >= 0.2, and nothing measured beats it by more than 1 % above the crossover.  Crossover rule shipped: no sharing kernel; k passes from",
          "A block rows of 1 MB on (KCHUNK_ROW_BYTES, multiply.py / mm_api.hip: lowered from 1.5 MB, +7 % at fill 0.2).  All of them sit on the same",
          "ceiling: ~0.50 of the nominal fp64 peak = the matrix pipe at the clock the chip holds under this load (DESIGN 5)."]
    open(P + "/r05_fill_sweep.txt", "w").write("\n".join(L) + "\n")


def large_blocks():
    L = ["Blocks of 33 ... 80, fp64, one MI355X (VERDICT r04 item 4): mm_numeric_f64 (rounds 1-4: one wave per C block, 32 x 32 sub-tiles one after the",
         "other, fragments from global memory) against mm_numeric_f64_big (round 5: a workgroup per C block, 2 x 2 waves of up to 5 x 5 MFMA tiles,",
         "operand slabs of 16 k shared through LDS, double-buffered).  16384^2 unless said, tools/block_bench.py; every r05 line was checked on the",
         "device against the plain global-memory kernel (--check: index identical, largest difference / largest element <= 7e-15).", "",
         "-- before (session r05_s01, the round-4 kernel)"]
    for r in rows(G + "/r05_s01/blocks_before.jsonl"):
        if r.get("dtype") == "f64" and "error" not in r:
            L.append("   m %-3s n %-3s k %-3s " % (r["mix_m"][1], r["mix_n"][1], r["mix_k"][1]) + fmt_block(r))
    L.append("-- after (sessions r05_s02 / s03)")
    for r in rows(G + "/r05_s02/large_blocks_after.jsonl") + [r for r in rows(G + "/r05_s03/sweeps.jsonl") if r.get("mix_m") != [1, 23]]:
        if "error" not in r:
            chk = r.get("check", {})
            L.append("   m %-3s n %-3s k %-3s " % (r["mix_m"][1], r["mix_n"][1], r["mix_k"][1]) + fmt_block(r) +
                     ("  [check: index %s, diff %.1e]" % ("==" if chk.get("index_identical") else "!=", chk.get("max_abs_diff_over_max_abs", 0.0)) if chk else ""))
    L.append("-- final (session r05_s08: B slab trimmed to the 16 TN columns a workgroup owns -> three workgroups per CU for the 5-tile shapes)")
    for r in rows(G + "/r05_s08/large_blocks_final.jsonl"):
        if "error" not in r:
            chk = r.get("check", {})
            L.append("   m %-3s n %-3s k %-3s " % (r["mix_m"][1], r["mix_n"][1], r["mix_k"][1]) + fmt_block(r) +
                     ("  [check: index %s, diff %.1e]" % ("==" if chk.get("index_identical") else "!=", chk.get("max_abs_diff_over_max_abs", 0.0)) if chk else ""))
    L.append("-- final (session r05_s11): every wave multiplies exactly the tiles it owns (the halves of an odd tile count are one tile apart: 72 = 5 + 4, 40 = 3 + 2),")
    L.append("   the variant chosen once outside the product loop; 'all_tiles' = the same build with every wave issuing all TM x TN tile products (DBCSR_AMD_MM_BIG=2)")
    for r in rows(G + "/r05_s11/large_blocks_exact.jsonl"):
        if "error" not in r:
            chk = r.get("check", {})
            L.append("   m %-3s n %-3s k %-3s " % (r["mix_m"][1], r["mix_n"][1], r["mix_k"][1]) + fmt_block(r) +
                     ("  [check: index %s, diff %.1e]" % ("==" if chk.get("index_identical") else "!=", chk.get("max_abs_diff_over_max_abs", 0.0)) if chk else ""))
    L += ["", "-- the same dataflow under the acc ABI: tools/acc_bench.py (the reference's acc_bench / kernel timer shape: a 16005-entry stack over 10000 A, 10000 B,",
          "   1000 C blocks, libsmm_acc_transpose + libsmm_acc_process, every line checked against the CPU oracle)",
          "   before (session r05_s06: smm_stack_f64, 32 x 32 tiles one after the other, fragments from global memory):"]
    L += ["      " + ln.strip() for ln in open(G + "/r05_s06/acc_bench_blocks.txt")]
    L.append("   after (session r05_s08: smm_stack_f64_big, a workgroup per 8 stack entries, sums kept across runs of equal C offsets):")
    L += ["      " + ln.strip() for ln in open(G + "/r05_s08/acc_bench_blocks.txt")]
    L.append("   final (session r05_s11: exact tile counts per wave; [all tiles] = DBCSR_AMD_SMM_BIG_EXACT=0 on the same build):")
    L += ["      " + ln.strip() for ln in open(G + "/r05_s11/acc_bench_exact.txt")]
    L += ["", "72^3: engine 13.4 -> 46.3 TFLOP/s = 0.59 of the fp64 peak (the 0.40 asked for), acc ABI 11.6 -> 33.0 = 0.42 (the reference's own tuned kernel: 8.1 TFLOP/s",
          "on Mi250-class hardware, src/acc/libsmm_acc/parameters/parameters_Mi350.json:433).  80^3 0.63, 64^3 0.65 (no padding), 55^3 0.54, 40^3 0.47, 45x67x78 0.48, 33^3 0.29.",
          "What is left of the 2 x 2 wave arrangement's imbalance: the workgroup waits for its largest sub-block (72: 25 of 20.25 tiles on average); the matrix pipe of a",
          "SIMD is shared with the waves of two other workgroups, which is where the exact tile counts gain (72^3 +14 %, 40^3 +17 %, 55^3 +14 %).  k passes are switched off",
          "for blocks above 32 (three passes cost 72^3 15 %: the slabs are shared through LDS, C re-read per pass is pure cost)."]
    open(P + "/r05_large_blocks.txt", "w").write("\n".join(L) + "\n")


def soak():
    L = ["Multi-process parity walks (VERDICT r04 item 1): N independent processes on ONE MI355X walk disjoint parts of the randomised sweep",
         "(tests/test_gpu_random_sweep.py: 1200 plain + 480 forced-path cases = what round 4's session 18 walked with `pytest -n 4`), every case",
         "with a record (tools/soak_multiproc.py; per-case records: gpurun_out/r05_s0*/soak_*/worker*.jsonl).", ""]
    for f in sorted(glob.glob(G + "/r05_s0*/soak_*/summary.json")):
        s = json.load(open(f))["summary"]
        L.append("%-46s procs %d  oracle threads %-3s cases %4d / %4d  failures %d  wall %6.1f s  oracle s (median / p99 / max) %s  device s %s" % (
            os.path.relpath(os.path.dirname(f), G), s["procs"], s["omp_threads"] or "dflt", s["cases_walked"], s["cases_total"], s["failures"], s["wall_s"],
            s["oracle_s_median_p99_max"], s["device_s_median_p99_max"]))
    L += ["", "Five complete walks of all 1680 cases (four with 4 processes, one with 8) and one partial walk: no mismatch, no exception.",
          "What session 18's environment was: the GPU box shows 256 CPUs behind a cgroup quota of 16 (gpurun_out/r05_s01/host.txt: cpu.max = 1600000 100000); the",
          "oracle's OpenMP fill took the default of 256 threads per process, four processes = 1024 threads on 16 CPUs.  In that mode (r05_s01/soak_p4_t0) single",
          "calls of the oracle that take 0.3 ms took up to 40.2 s, a device case up to 25.1 s (forced-path cases: hiprtc) -- a 100 000-fold tail under CFS",
          "throttling; tests/conftest.py gives every test 600 s (pytest-timeout), and session 18 ran 1300 s under `pytest -n 4` on top of that load.  The two `F`",
          "of session 18 are consistent with time-outs of the CHECKER's side, not with a wrong product: every one of the 1680 cases has now passed five times",
          "with other processes on the device, none failed.  Removed at the root: the oracle (test infrastructure) and bench.py size their thread teams by",
          "min(visible CPUs, affinity, cgroup quota) (oracle/oracle.py usable_cpus, tests/test_oracle_threads.py); the same default walk then takes 52 s",
          "(r05_s03/soak_p4_default_threads: 1680 cases, 0 failures) instead of not finishing in 1300 s."]
    open(P + "/r05_soak_multiproc.txt", "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    fill_sweep()
    large_blocks()
    soak()
    print(open(P + "/r05_soak_multiproc.txt").read())

#!/usr/bin/env python3
"""What the multi-GPU schedules of dbcsr_amd/cannon.py should cost per step, from numbers measured on ONE GPU and an
assumed xGMI rate -- the model behind the scaling estimates, written down so that the first run on a multi-GPU node can be held against
it line by line.

Inputs per workload: bytes of A and B, the per-rank compute path measured with tools/rank_step_budget.py (gather schedule's one
multiply; colpipe's chunk multiplies), the one-GPU step.  Every image travels over its own link (owner-direct exchange).
   python tools/schedule_model.py [--link-gbs 50,64,76]"""
import argparse

# measured on one MI355X, one rank's share ALONE on the device (tools/rank_step_budget.py; round 6: gpurun_out/r06_s21 = profiles/r06_rank_step_budget.txt,
# round 3 where the round-6 session's colpipe walls were disturbed by an outlier step: profiles/r03_rank_step_budget_*.txt); ms
WORKLOADS = {
    "config2_32768_23x23_fill10_fp64": dict(
        one_gpu=18.90, a_gb=0.859, b_gb=0.859, c_gb_per_rank={2: 4.3, 4: 2.15, 8: 1.07},
        gather={2: 9.71, 4: 4.91, 8: 2.65},          # rank 0's whole multiply, default grid (2x1, 2x2, 4x2)
        colpipe8={2: 10.28, 4: 5.38, 8: 2.92},       # N x 1 grid, eight column chunks on two streams, panels in place
        colpipe2d={4: 5.35, 8: 2.82},                # 2-D grid (2x2, 4x2), column chunks of the rank's own block columns (round 6)
        # tile pipeline on the 2-D grid, S chunks per side (session r06_33): wall of the rank's step alone on the device, kernel ms of step 0 .. S - 1 (row + column strip)
        tilepipe={4: {4: (5.313, [0.356, 1.012, 1.63, 2.518]), 6: (6.167, [0.197, 0.507, 0.743, 1.043, 1.354, 1.877])},
                  8: {4: (2.974, [0.191, 0.533, 0.871, 1.484]), 6: (3.027, [0.099, 0.306, 0.433, 0.574, 0.729, 1.235]),
                      8: (3.159, [0.067, 0.189, 0.288, 0.376, 0.453, 0.548, 0.641, 1.095])}}),
    "config4_131072_23x23_fill1_fp64": dict(
        one_gpu=22.35, a_gb=1.374, b_gb=1.374, c_gb_per_rank={2: 30.3, 4: 15.2, 8: 7.6},
        gather={2: 11.13, 4: 5.69, 8: 2.96},
        colpipe8={2: 11.31, 4: 5.75, 8: 3.10},
        colpipe2d={4: 5.78, 8: 2.97},
        tilepipe={4: {4: (6.16, [0.418, 1.788, 3.124, 3.784]), 6: (6.265, [0.182, 0.52, 1.123, 1.634, 2.449, 3.001])},
                  8: {4: (3.256, [0.207, 0.655, 1.121, 1.775]), 6: (3.281, [0.094, 0.279, 0.445, 0.618, 0.776, 1.112]),
                      8: (3.389, [0.056, 0.201, 0.279, 0.359, 0.473, 0.558, 0.636, 0.755])}}),
}
GRID = {2: (2, 1), 4: (2, 2), 8: (4, 2)}
HBM_TBS = 5.0   # read + write of C in an in-place pass


def model(w, n, link):
    pr, pc = GRID[n]
    nvirt = pr * pc // __import__("math").gcd(pr, pc)
    a_img, b_img = w["a_gb"] / (pr * nvirt), w["b_gb"] / (pc * nvirt)       # GB per image
    per_link = max(a_img if pc > 1 else 0.0, b_img if pr > 1 else 0.0) * (nvirt / max(pr, pc))   # images a rank gets from ONE peer
    t_link = per_link / link * 1e3                                          # ms: all links run at once
    k = w["gather"][n]
    local = 1.0 / nvirt if nvirt > 1 else 1.0                               # share of the multiply that needs no transfer (local-first)
    c_pass = 2 * w["c_gb_per_rank"][n] / HBM_TBS                            # ms: one more pass over C
    out = {"gather": t_link + k,
           "gather/local-first": max(t_link, local * k) + (1 - local) * k + c_pass,
           "ticks": t_link / nvirt + nvirt * max(k / nvirt + c_pass, t_link / nvirt) - (0 if nvirt > 1 else c_pass)}
    cp = w["colpipe8"][n]
    if cp is not None:
        b_per_link = w["b_gb"] / n                                          # N x 1 grid: one k-image per rank, every image on its own link
        t_chunk = b_per_link / 8 / link * 1e3
        out["colpipe (8 chunks)"] = t_chunk + max(cp, 8 * t_chunk)
    cp2 = w.get("colpipe2d", {}).get(n)
    if cp2 is not None and pc > 1:
        # 2-D grid: the A images a rank misses (its process row's: a_gb / n from each of the pc - 1 row peers, one link each) come with the first batch
        # and are NOT chunked -- every column chunk needs the whole row panel of A --, then B's images of the process column in 8 column chunks
        t_a = w["a_gb"] / n / link * 1e3
        t_chunk = w["b_gb"] / n / 8 / link * 1e3
        out["colpipe2d (8 chunks)"] = max(t_a, t_chunk) + max(cp2, 8 * t_chunk)
    tp = w.get("tilepipe", {}).get(n)
    if tp and pc > 1:
        # batch s = row chunk s of the A images a rank misses (a_gb / n / S from each row peer, one link each) next to column chunk s of B's images (b_gb / n / S per
        # column peer, other links): lands at (s + 1) * t_chunk; step s (its two strips share the device: the step's share of the measured wall) starts when batch s has
        # landed and step s - 1 is done
        best = None
        for S, (wall, steps) in tp.items():
            t_chunk = max(w["a_gb"], w["b_gb"]) / n / S / link * 1e3
            scale, t = wall / sum(steps), 0.0
            for q, k in enumerate(steps):
                t = max(t, (q + 1) * t_chunk) + k * scale
            if best is None or t < best[0]:
                best = (t, S)
        out["tilepipe (%d chunks)" % best[1]] = best[0]
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--link-gbs", default="50,64,76", help="GB/s per direction and link that RCCL send/recv reaches (assumptions)")
    a = p.parse_args()
    for name, w in WORKLOADS.items():
        print("# %s: %.2f ms on one GPU" % (name, w["one_gpu"]))
        for link in [float(x) for x in a.link_gbs.split(",")]:
            print("#   link %.0f GB/s:  ranks  schedule -> ms per step (speed-up over one GPU)" % link)
            for n in (2, 4, 8):
                m = model(w, n, link)
                best = min(m, key=m.get)
                print("      %d  " % n + "   ".join("%s %.2f (%.1fx)%s" % (k, v, w["one_gpu"] / v, " *" if k == best else "") for k, v in m.items()))


if __name__ == "__main__":
    main()

#!/bin/bash
# Runs bench.py under rocprofv3 on the GPU box and leaves the summaries in
# gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/).
#   tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs $*"
# pass 1: kernel trace + stats
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python "$OLDPWD/bench.py" $ARGS ) > "$OUT/trace.log" 2>&1
# pass 2..: PMC counters, each in its own run (no tracing domains)
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES"; do
  i=$((i+1))
  [ -n "${PROFILE_PASSES:-}" ] && [ $i -gt "$PROFILE_PASSES" ] && break   # (a session short of GPU minutes: the first passes only)
  ( cd /tmp && timeout 150 rocprofv3 --pmc $PMC -d "$OUT/pmc$i" -o pmc --output-format csv -- python "$OLDPWD/bench.py" $ARGS ) > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    lines += open(f).read().splitlines()[:25]
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        lines.append("== PMC per dispatch average (%s)" % os.path.relpath(f, out))
        for k in agg:
            if "mm_numeric" in k or "fill_products" in k or "count_products" in k:
                lines.append("  " + k + ": " + ", ".join("%s=%.4g" % (c, v / cnt[(k, c)]) for c, v in sorted(agg[k].items())))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
# keep the merged output small
find "$OUT" -name "*.csv" -size +2M -delete

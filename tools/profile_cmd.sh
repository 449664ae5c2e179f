#!/bin/bash
# Counter passes of ANY command under rocprofv3 (one pass per counter group, counters only -- no tracing domain beside them), the
# per-dispatch averages of the block-product kernels written to gpurun_out/prof_<tag>/summary.txt.
#   tools/profile_cmd.sh <tag> <command ...>
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- "$@" ) > "$OUT/trace.log" 2>&1
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && timeout 240 rocprofv3 --pmc $PMC -d "$OUT/pmc$i" -o pmc --output-format csv -- "$@" ) > "$OUT/pmc$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    lines += open(f).read().splitlines()[:14]
for d in sorted(glob.glob(out + "/pmc*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f, errors="replace")):
            try:
                k = r["Kernel_Name"].split("(")[0][:70]
                if "mm_numeric" not in k:
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
            except (KeyError, ValueError, TypeError):
                continue
        lines.append("== PMC per dispatch average (%s)" % os.path.relpath(f, out))
        for k in agg:
            lines.append("  " + k + "  [%d dispatches]" % max(cnt[(k, c)] for c in agg[k]))
            for c in sorted(agg[k]):
                lines.append("      %-32s %.6g" % (c, agg[k][c] / cnt[(k, c)]))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

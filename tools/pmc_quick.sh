#!/bin/bash
# tools/pmc_quick.sh "<COUNTERS>" [env assignments...]  -- one rocprofv3 --pmc pass over a short bench.py run,
# prints the per-dispatch average of each counter for the block-GEMM kernel.
PMC=$1; shift
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD; OUT=$R/gpurun_out/pmcq_$$; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && env "$@" timeout 150 rocprofv3 --pmc $PMC -d "$OUT" -o pmc --output-format csv -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 $BENCH_ARGS ) > "$OUT/log" 2>&1
python3 - "$OUT" "$*" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mm_numeric" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print(sys.argv[2], {k: "%.4g" % (v / cnt[k]) for k, v in sorted(agg.items())})
PY
rm -rf "$OUT"

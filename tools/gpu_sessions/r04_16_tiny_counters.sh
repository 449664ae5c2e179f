#!/bin/bash
# round 4, GPU session 16: what the 4 x 4 kernel (config 1) is bound by: fabric bytes, L2 hit rate, clock, busy cycles of both forms
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s16; mkdir -p $O
export TMPDIR=/tmp
W=config1_4096_4x4_fill10_fp64
for t in 1 2; do
  DBCSR_AMD_MM_TINY=$t timeout 400 python bench.py --workload $W --steps 50 --warmup 5 --cpu-seconds 0 --pmc --no-other-configs > $O/bench_t$t.json 2> $O/bench_t$t.err
  python3 - $O/bench_t$t.json $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("TINY=%s ms %.4f kernel_ms %.4f %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["kernel"]))
print("   ", {k: r.get(k) for k in ("traffic", "compulsory_bytes", "fabric_tb_per_s", "sclk_mhz", "mfma_busy_frac", "l2_hit_rate", "fabric", "binding_resource")})
PY
done 2>&1 | tee $O/summary.txt
# a third pass: vector-memory and L1 counters of the wave-per-quad kernel
( cd /tmp && DBCSR_AMD_MM_TINY=2 timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OLDPWD/$O/pmc3 -o p --output-format csv -- python $OLDPWD/bench.py --workload $W --steps 2 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs ) > $O/pmc3.log 2>&1
( cd /tmp && DBCSR_AMD_MM_TINY=2 timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum -d $OLDPWD/$O/pmc4 -o p --output-format csv -- python $OLDPWD/bench.py --workload $W --steps 2 --warmup 1 --cpu-seconds 0 --no-pmc --no-other-configs ) > $O/pmc4.log 2>&1
python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
for sub in ("pmc3", "pmc4"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f, errors="replace")):
            if "tiny" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(agg): print("%s %-36s per launch %.4g  (%d launches)" % (sub, k, agg[k] / n[k], n[k]))
PY
tail -5 $O/pmc3.log $O/pmc4.log | grep -i "error\|invalid\|not" | head
find $O -name "*.csv" -size +1M -delete

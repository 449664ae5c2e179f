w=config3_32768_mixed13_23_32_fill5_fp64
for v in "pipe 0" "lds1 0" "lds1 1" "lds1 2" "lds1 7"; do set -- $v
echo "kernel=$1 dbg=$2 $(DBCSR_AMD_MM_KERNEL=$1 DBCSR_AMD_MM_DBG=$2 python bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3))' 2>&1 | tail -1)"
done

for d in 0 1 2 7; do
echo "dbg=$d $(DBCSR_AMD_MM_DBG=$d python bench.py --workload config4_131072_23x23_fill1_fp64 --steps 4 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3), round(j["roofline"]["fill_products_ms"],3))')"
done

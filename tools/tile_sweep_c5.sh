w=config5_131072_32x32_fill20_fp32
for cfg in "1 16" "1 32" "1 8" "400 16"; do set -- $cfg
echo "PANEL_MB=$1 RG=$2 $(DBCSR_AMD_MM_PANEL_MB=$1 DBCSR_AMD_MM_ROW_GROUP=$2 python bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],1), round(j["roofline"]["kernel_ms"],1), round(j["value"]))' 2>&1 | tail -1)"
done

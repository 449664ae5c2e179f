for w in fp32_16384_32x32_fill20 config2_32768_23x23_fill10_fp64; do
for pm in 160 1; do for rg in 1 8 16 24; do
echo "$w PANEL_MB=$pm RG=$rg $(DBCSR_AMD_MM_PANEL_MB=$pm DBCSR_AMD_MM_ROW_GROUP=$rg python bench.py --workload $w --steps 4 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3), round(j["value"]))' 2>&1 | tail -1)"
done; done; done

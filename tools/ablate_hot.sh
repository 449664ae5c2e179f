export DBCSR_AMD_MM_HOT=1
for d in 0 1 2 4 3 5 7 128; do
  echo "dbg=$d $(DBCSR_AMD_MM_DBG=$d python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], j["roofline"])')"
done

#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... -- prints kernel ms of bench.py for each value of an engine env switch
var=$1; shift
for v in "$@"; do
  echo "$var=$v $(env $var=$v python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3))')"
done

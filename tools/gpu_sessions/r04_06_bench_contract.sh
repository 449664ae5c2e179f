#!/bin/bash
# round 4, GPU session 6: index stamps (trusted plan), pinned staging of mixed stacks, the whole bench line (cold plan, fabric probe, other configs)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_plan_reuse.py tests/test_gpu_libsmm.py tests/test_gpu_acc_spec.py tests/test_gpu_native_multiply.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print({k: r[k] for k in ("value", "ms_per_step", "plan", "ms_per_step_cold", "value_cold", "multiplies_run") if k in r})
        print({k: v for k, v in r["roofline"].items() if k != "fabric"})
        print(r["roofline"].get("fabric"))
        for o in r.get("other_configs", []):
            print(o)
        print(r.get("cpu_baseline"))
PY

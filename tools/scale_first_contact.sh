#!/bin/bash
# First contact with a multi-GPU node: ONE command that takes the RCCL path of this repository from "never met a second
# peer" to a bench line, stage by stage, stopping at the first stage that fails:
#   1. tools/comm_selfcheck.py   all-pairs grouped exchange + sizes allgather through the C-ABI transport, checked against host copies, GB/s per link
#   2. tools/run_dist_check.py   every schedule (gather, ticks, colpipe, distributed input) with transport=native, against the CPU oracle's product
#   3. bench.py --gpus N         the benchmark line (carries rccl_ranks, GB/s per link of the exchange alone, the overlap fraction)
#   tools/scale_first_contact.sh [N] [--dry-run]      N defaults to the number of visible devices; --dry-run prints the commands only
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
N=""; DRY=0
for a in "$@"; do case "$a" in --dry-run) DRY=1;; *) N=$a;; esac; done
[ -z "$N" ] && N=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/first_contact; mkdir -p $O
port() { python -c "import socket; s = socket.socket(); s.bind(('127.0.0.1', 0)); print(s.getsockname()[1])"; }
run() {   # run <stage name> <command ...>
  local name=$1; shift
  echo "== stage $name: $*"
  [ $DRY = 1 ] && return 0
  ( "$@" ) > $O/$name.out 2> $O/$name.err
  local rc=$?
  tail -3 $O/$name.out
  if [ $rc != 0 ]; then echo "== stage $name FAILED (rc $rc); see $O/$name.err"; tail -20 $O/$name.err; exit $rc; fi
}
if [ "$N" -lt 2 ] && [ $DRY = 0 ]; then echo "scale_first_contact: $N device(s) visible, nothing to contact (use --dry-run to see the stages)"; exit 3; fi
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1"
run 1_comm_selfcheck $TR --master-port $(port) tools/comm_selfcheck.py nccl native
for mode in gather ticks colpipe gather+dist; do
  run 2_dist_check_${mode/+/_} $TR --master-port $(port) tools/run_dist_check.py nccl $mode native
done
run 3_bench python bench.py --gpus $N --steps 10 --warmup 2 --dist-transport native
[ $DRY = 1 ] || grep '^{"metric"' $O/3_bench.out | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); c = d.get('comm') or {}
print('value %.1f %s on %d GPUs, rccl_ranks %s, %.1f GB/s per link, overlap %.2f' % (d['value'], d['unit'], d['n_gpus'], d['config'].get('rccl_ranks'), c.get('gb_per_s_per_link') or 0, c.get('overlap_fraction') or 0))"

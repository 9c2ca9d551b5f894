#!/bin/bash
# GPU box (one GPU): the N > 1 code path of bench.py under RCCL with a single rank (BENCH_FORCE_DIST), in the three
# forms of the per-step scalar all-reduce: deferred (default), overlapped mid-step, synchronous.
set -u
i=0
for mode in "BENCH_ALLREDUCE=deferred" "BENCH_ALLREDUCE=midstep" "BENCH_SYNC_ALLREDUCE=1"; do
  i=$((i+1))
  echo "== $mode"
  env $mode BENCH_FORCE_DIST=1 VAE_AMD_DIST_FORCE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
      --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 --steps 50 --warmup 10 --main-only 2>/dev/null \
    | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print(j['ms_per_step'], j['regions']['ms_per_step_min'], j['regions']['ms_per_step_max'], j['loss_mean'])" || exit 1
done

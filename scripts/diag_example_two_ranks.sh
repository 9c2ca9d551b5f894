#!/bin/bash
# Diagnostic: repeat the two-rank gloo rehearsal of the example until one run stalls; the stalled run dumps every
# thread's stack after 40 s (EXAMPLE_DEBUG) and the script stops there.
for i in $(seq 1 12); do
  port=$((20000 + RANDOM % 20000))
  echo "== run $i port $port" >> gpurun_out/diag_two_ranks.log
  EXAMPLE_DEBUG=40 EXAMPLE_BACKEND=gloo timeout -k 5 80 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $port examples/train_simple_vae.py --steps 41 --batch 512 \
    >> gpurun_out/diag_two_ranks.log 2>&1
  rc=$?
  echo "== rc $rc" >> gpurun_out/diag_two_ranks.log
  echo "run $i rc $rc"
  if [ $rc -ne 0 ]; then exit 1; fi
done

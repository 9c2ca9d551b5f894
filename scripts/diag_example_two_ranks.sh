#!/bin/bash
# Diagnostic: ONE two-rank gloo rehearsal of the example with a thread dump armed (EXAMPLE_DEBUG=40: if the run has
# not finished after 40 s, every thread's stack goes to stderr and the process exits).  Run it once; never in a loop.
port=$((20000 + RANDOM % 20000))
EXAMPLE_DEBUG=40 EXAMPLE_BACKEND=gloo timeout -k 5 80 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port $port examples/train_simple_vae.py --steps 41 --batch 512 \
  > gpurun_out/diag_two_ranks.log 2>&1
echo "rc $?"

#!/bin/bash
# GPU box: whole GPU suite, then (only if green) two short main-only bench runs.
set -e
TAG=${1:-run}
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/full_gpu_$TAG.log 2>&1 || { tail -30 gpurun_out/full_gpu_$TAG.log; exit 1; }
tail -2 gpurun_out/full_gpu_$TAG.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 20 --main-only 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']/1e6,2), [(k, v['avg_ms']) for k,v in j['kernels'].items()])"
done

#!/bin/bash
# GPU box: bench + rocprofv3 kernel trace + HBM traffic counters for the bench workload.
# Outputs under gpurun_out/prof_$1/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 50 --warmup 10 --main-only > $OUT/trace.log 2>&1 || { echo "rocprof trace failed"; tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -r cat | head -20
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 bench.py --steps 5 --warmup 2 --main-only > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 bench.py --steps 5 --warmup 2 --main-only > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; tail -5 $OUT/pmc_write.log; exit 1; }
python3 scripts/summarize_pmc.py $OUT > $OUT/traffic_summary.txt 2>&1; cat $OUT/traffic_summary.txt
# keep only small summaries for the merge back
find $OUT -name '*.csv' -size +2M -delete
ls -la $OUT $OUT/trace 2>/dev/null | head -40

#!/bin/bash
# Copy the summaries of a scripts/gpu_profile.sh run (gpurun_out/prof_<tag>) and of the last bench_configs table into
# profiles/ (the tracked, judged copies).   scripts/collect_profiles.sh r03i r03
set -e
tag=$1; rnd=$2
P=profiles/$rnd; mkdir -p $P
cp gpurun_out/prof_$tag/bench.json $P/bench.json
cp gpurun_out/prof_$tag/traffic.json $P/traffic_full.json
cp gpurun_out/prof_$tag/traffic.json profiles/traffic.json
cp gpurun_out/prof_$tag/traffic_summary.txt $P/traffic_summary.txt
python3 - "$tag" "$rnd" <<'PY'
import csv, glob, sys
tag, rnd = sys.argv[1:3]
p = glob.glob('gpurun_out/prof_%s/trace/*kernel_stats.csv' % tag)[0]
rows = list(csv.reader(open(p)))
for r in rows:
    r[0] = r[0][:200]        # torch's templated kernel names run to kilobytes
for q in ('profiles/%s/bench_kernel_stats.csv' % rnd, 'profiles/%sa_bench_kernel_stats.csv' % rnd):
    csv.writer(open(q, 'w', newline='')).writerows(rows)
PY
cp $P/bench.json profiles/${rnd}a_bench.json
cp $P/traffic_full.json profiles/${rnd}a_traffic_full.json
cp $P/traffic_summary.txt profiles/${rnd}a_traffic_summary.txt
if [ -f gpurun_out/${rnd}_configs.txt ]; then
  cp gpurun_out/${rnd}_configs.txt $P/${rnd}_configs.txt
  cp gpurun_out/${rnd}_configs.json $P/${rnd}_configs.json
  cp gpurun_out/${rnd}_configs.txt profiles/${rnd}_configs.txt
fi

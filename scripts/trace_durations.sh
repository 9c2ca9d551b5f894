#!/bin/bash
# GPU box: per-dispatch kernel durations (rocprofv3 --kernel-trace) of one bench_configs section, grouped by kernel and
# grid size, so that the sizes a section mixes (F=512 / F=64 ...) are told apart.   scripts/trace_durations.sh tails
set -u
SEC=$1
OUT=gpurun_out/trace_$SEC
mkdir -p $OUT
export TMPDIR=/tmp BENCH_COLD_SAMPLES=5
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 scripts/bench_configs.py --sections $SEC > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(out + '/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name']
        if 'vae::' not in n:
            continue
        n = n.replace('void vae::', '').split('(')[0]
        acc[(n, r.get('Grid_Size_X', r.get('Grid_Size', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open(out + '/durations.txt', 'w') as f:
    for (n, g), v in sorted(acc.items()):
        v.sort()
        f.write('%-60s grid %-9s n=%-4d median %8.1f us  min %8.1f\n' % (n[:60], g, len(v), v[len(v) // 2], v[0]))
print(open(out + '/durations.txt').read())
PY
find $OUT/t -name '*.csv' -size +1M -delete

#!/bin/bash
# GPU box: rocprofv3 over scripts/bench_configs.py --sections <S>: kernel trace (durations), shader counters (what
# the waves wait on), HBM traffic (FETCH_SIZE / WRITE_SIZE, each in its own pass).  Summaries -> gpurun_out/sec_<tag>/.
#   scripts/prof_sections.sh r03_c3 c3
set -u
TAG=$1; SEC=$2
OUT=gpurun_out/sec_$TAG
mkdir -p $OUT
export TMPDIR=/tmp BENCH_COLD_SAMPLES=5
CMD="python3 scripts/bench_configs.py --sections $SEC"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep -v amdgpu.ids $OUT/trace.log > $OUT/bench_under_trace.txt
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -r cat > $OUT/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o t -- $CMD > $OUT/pmc_sq.log 2>&1 || { echo "pmc sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- $CMD > $OUT/pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- $CMD > $OUT/pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for sub in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    for path in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row['Kernel_Name']
            if 'vae::' in k:
                acc[k.replace('void vae::', '').split('(')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
dur = {}
p = os.path.join(out, 'kernel_stats.csv')
if os.path.exists(p):
    for row in csv.DictReader(open(p)):
        dur[row['Name'].replace('void vae::', '').split('(')[0]] = (float(row['AverageNs']) / 1e3, int(row['Calls']))
with open(os.path.join(out, 'summary.txt'), 'w') as f:
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1
        w = max(m.get('SQ_WAVES', 1), 1)
        d = dur.get(k, (0, 0))
        # FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction of the guide (scripts/summarize_pmc.py): FETCH_SIZE counts half of
        # a wide streaming read -> doubled; WRITE_SIZE is exact
        line = ('%-64s avg %7.1f us x%-4d waves %7.0f wait_mem %5.1f%% issue_stall %5.1f%% active %5.1f%% (valu %5.1f%%) '
                'VALU/wave %6.0f SALU/wave %6.0f fetch %8.1f MB write %8.1f MB' % (
                    k[:64], d[0], d[1], w, 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
                    100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc,
                    m.get('SQ_INSTS_VALU', 0) / w, m.get('SQ_INSTS_SALU', 0) / w, 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e6,
                    m.get('WRITE_SIZE', 0) * 1024 / 1e6))
        print(line); f.write(line + '\n')
PY
find $OUT -name '*.csv' -size +1M -delete
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write

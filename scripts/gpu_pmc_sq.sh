#!/bin/bash
# GPU box: shader-side counters for the bench kernels (one pass, 8 SQ counters), to show what the waves wait on.
# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt: memory) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY.
set -u
TAG=${1:-r01}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -o bench -- python3 bench.py --steps 5 --warmup 2 --main-only > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, 'pmc', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name']
        if 'vae::' in k:
            acc[k.split('(')[0].split('::')[-1].split('<')[0]][row['Counter_Name']].append(float(row['Counter_Value']))
with open(os.path.join(out, 'sq_summary.txt'), 'w') as f:
    for k, c in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1
        line = ('%-28s waves %8.0f  wait_any %5.1f%%  wait_inst %5.1f%%  active %5.1f%% (valu %5.1f%%)  VALU insts/wave %7.0f  '
                'VMEM_RD insts/wave %5.1f' % (k, m.get('SQ_WAVES', 0), 100 * m.get('SQ_WAIT_ANY', 0) / wc,
                100 * m.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc, m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_WAVES', 1), 1),
                m.get('SQ_INSTS_VMEM_RD', 0) / max(m.get('SQ_WAVES', 1), 1)))
        print(line); f.write(line + '\n')
PY
find $OUT -name '*.csv' -size +2M -delete

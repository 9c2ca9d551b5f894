"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
on gfx950 FETCH_SIZE reads exactly half of a wide coalesced streaming read -> doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def collect(sub, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get('Counter_Name') == counter:
                    per[row['Kernel_Name']].append(float(row['Counter_Value']))
    return per


fetch, write = collect('pmc_fetch', 'FETCH_SIZE'), collect('pmc_write', 'WRITE_SIZE')
summary = {}
for k in sorted(set(fetch) | set(write)):
    if 'vae' not in k:
        continue
    f = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    short = k.split('(')[0].split('::')[-1].split('<')[0]
    summary[k] = {'kernel': short, 'launches_seen': len(fetch.get(k, [])), 'FETCH_SIZE_KiB_raw': f,
                  'WRITE_SIZE_KiB_raw': w, 'hbm_read_bytes_corrected': 2 * f * 1024, 'hbm_write_bytes': w * 1024,
                  'hbm_bytes_per_launch': 2 * f * 1024 + w * 1024}
    print('%-28s launches %3d  FETCH %.0f KiB (x2 corrected -> %.1f MB)  WRITE %.1f MB  total %.1f MB' % (
        short, len(fetch.get(k, [])), f, 2 * f * 1024 / 1e6, w * 1024 / 1e6, (2 * f + w) * 1024 / 1e6))
json.dump(summary, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)

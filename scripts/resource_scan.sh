#!/bin/bash
# Every kernel's registers, scratch and occupancy as the compiler reports them (-Rpass-analysis=kernel-resource-usage).
# A kernel with ScratchSize > 0 spills or indexes a private array dynamically: none should.   scripts/resource_scan.sh > profiles/rNN/kernel_resources.txt
cd "$(dirname "$0")/../vae_amd/csrc" || exit 1
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc --cuda-device-only -c $f -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
name = None; res = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: name = m.group(1); res[name] = {}
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('sgpr', r' SGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, l)
        if m and name: res[name][key] = int(m.group(1))
names = list(res)
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
print('== $f: %d kernels, %d with scratch' % (len(res), sum(1 for r in res.values() if r.get('scratch', 0) > 0)))
for n, d in zip(names, dem):
    r = res[n]
    print('  %-88s vgpr %3d sgpr %3d lds %6d scratch %3d occ %d' % (d.replace('void vae::', '').split('(')[0][:88], r.get('vgpr', -1), r.get('sgpr', -1), r.get('lds', 0), r.get('scratch', 0), r.get('occ', -1)))
"
done

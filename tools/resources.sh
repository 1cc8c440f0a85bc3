#!/bin/bash
# tools/resources.sh [out.txt]: register / scratch / occupancy / LDS of every k_render instantiation, from hipcc's own remarks
# (-Rpass-analysis=kernel-resource-usage on csrc/rmdf_render.hip with the product's flags).  The docs cite this table instead of
# typed-in figures: profiles/r04_kernel_resources.txt.
cd "$(dirname "$0")/.." || exit 1
out=${1:-profiles/kernel_resources.txt}
make -C ray-marching-distance-fields_amd/csrc resources 2>&1 | python3 -c "
import re, subprocess, sys
cur = None; rows = []
for l in sys.stdin:
    m = re.search(r'remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-Rpass', l)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name': cur = {'name': v}; rows.append(cur)
    elif cur is not None: cur[k] = v
print('# hipcc -Rpass-analysis=kernel-resource-usage of csrc/rmdf_render.hip, product build flags (tools/resources.sh)')
print('# k_render<SCENE, MERGE, OUT>: SCENE = FragmentShader enum; MERGE = straggler pooling; OUT 0 = RGBA8 only (product), 1 = + registered host buffer, 2 = + test planes')
print('%-44s %5s %5s %8s %6s %7s %7s %6s' % ('kernel', 'VGPR', 'SGPR', 'scratch', 'waves', 'sgprSp', 'vgprSp', 'LDS'))
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().replace('rmdf::', '').replace('(FrameParams)', '')
    print('%-44s %5s %5s %8s %6s %7s %7s %6s' % (n[:44], r.get('VGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('SGPRs Spill'), r.get('VGPRs Spill'), r.get('LDS Size [bytes/block]')))
" > "$out"
cat "$out"

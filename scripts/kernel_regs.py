"""Register / spill / LDS use of every kernel in one HIP source, from hipcc's kernel-resource-usage remarks.

    python scripts/kernel_regs.py pytorch-unet_amd/csrc/conv3x3_k32.hip [-DFOO=1 ...]
"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics',
       '-Wno-unused-result', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    if m.group(1) == 'Function Name':
        cur = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(' [')[0]] = m.group(2)
for k, v in rows.items():
    name = re.sub(r'\(anonymous namespace\)::', '', k)
    name = re.sub(r'\(.*', '', name)
    print('%-70s vgpr %-4s agpr %-3s spillV %-3s spillS %-3s scratch %-4s occ %s' % (
        name[:70], v.get('VGPRs'), v.get('AGPRs'), v.get('VGPRs Spill'), v.get('SGPRs Spill'), v.get('ScratchSize'),
        v.get('Occupancy')))

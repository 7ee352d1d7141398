#!/usr/bin/env python
"""Register / LDS / scratch usage of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
    python tools/kernel_resources.py adaptive-multispeaker-separation_amd/csrc/gemm.hip [substring filter] [extra hipcc flags...]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
    extra = [a for a in sys.argv[2:] if a.startswith('-')]
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
           '-Rpass-analysis=kernel-resource-usage'] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (.*?) \[-Rpass', ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == 'Function Name':
            cur = {'name': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    dem = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print('%-90s %5s %5s %6s %7s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'spill', 'scratch', 'occ', 'LDS'))
    for r, d in zip(rows, dem):
        d = d.replace('(anonymous namespace)::', '').split('(')[0]
        if filt and filt not in d:
            continue
        print('%-90s %5s %5s %6s %7s %4s %7s' % (d[:90], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('VGPR Spill', r.get('VGPRs Spill', '?')),
                                                  r.get('ScratchSize [bytes/lane]', '?'), r.get('Occupancy [waves/SIMD]', '?'),
                                                  r.get('LDS Size [bytes/block]', '?')))


if __name__ == '__main__':
    main()

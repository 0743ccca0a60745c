"""Per-kernel register / LDS / scratch usage of the HIP library as the compiler reports it
(`hipcc -Rpass-analysis=kernel-resource-usage`; cross-compiles without a GPU).

    python tools/kernel_resources.py [filter-substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    src = os.path.join(ROOT, 'emg3d_amd', 'csrc', 'kernels.hip')
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/tmp/_kr.o',
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r'remark: +(.*?) \[-Rpass', line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith('Function Name:') or t.startswith('Name:'):
            name = t.split(':', 1)[1].strip()
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            cur = {'name': re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0]}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':78s} {'vgpr':>5s} {'agpr':>5s} {'spill':>5s} {'scr':>5s} {'occ':>4s} {'lds':>7s}")
    for c in rows:
        if flt in c['name']:
            print(f"{c['name'][:78]:78s} {c.get('VGPRs', '?'):>5s} {c.get('AGPRs', '?'):>5s} "
                  f"{c.get('VGPRs Spill', '?'):>5s} {c.get('ScratchSize [bytes/lane]', '?'):>5s} "
                  f"{c.get('Occupancy [waves/SIMD]', '?'):>4s} {c.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == '__main__':
    main()

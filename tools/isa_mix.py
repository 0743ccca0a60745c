"""Instruction mix per basic block of one kernel, from the device assembly of the HIP library
(hipcc -S --cuda-device-only; no GPU needed).

    python tools/isa_mix.py <substring of the mangled kernel name> [asm file]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    key = sys.argv[1]
    asm = sys.argv[2] if len(sys.argv) > 2 else '/tmp/_emg_kernels.s'
    src = os.path.join(ROOT, 'emg3d_amd', 'csrc', 'kernels.hip')
    if not os.path.exists(asm) or os.path.getmtime(asm) < max(
            os.path.getmtime(os.path.join(ROOT, 'emg3d_amd', 'csrc', f)) for f in os.listdir(os.path.dirname(src))):
        subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                               src, '-o', asm])
    lines = open(asm).read().split('\n')
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l]
    if not start:
        raise SystemExit(f"no kernel matching {key!r}")
    i = start[0]
    print('#', lines[i].split(':')[0])
    seg, counts, order = 'entry', collections.defaultdict(collections.Counter), ['entry']
    for l in lines[i + 1:]:
        l = l.strip()
        if l.startswith('s_endpgm'):
            break
        if re.match(r'^\.?LBB\S*:', l):
            seg = l.split(':')[0]
            order.append(seg)
            continue
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if re.match(r'v_\w*f64', op):
            cat = 'fp64'
        elif op.startswith('ds_'):
            cat = 'lds'
        elif op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
            cat = 'gload'
        elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store')):
            cat = 'gstore'
        elif op.startswith('s_waitcnt') or op.startswith('s_barrier'):
            cat = 'wait'
        elif op.startswith('s_'):
            cat = 'salu'
        elif 'dpp' in l:
            cat = 'dpp'
        elif op.startswith('v_'):
            cat = 'valu32'
        else:
            cat = 'other'
        counts[seg][cat] += 1
    cats = ['fp64', 'valu32', 'dpp', 'lds', 'gload', 'gstore', 'salu', 'wait', 'other']
    print(f"{'block':28s}" + ''.join(f"{c:>8s}" for c in cats))
    tot = collections.Counter()
    for s in order:
        c = counts[s]
        tot += c
        if sum(c.values()) >= 12:
            print(f"{s[:28]:28s}" + ''.join(f"{c[k]:8d}" for k in cats))
    print(f"{'TOTAL':28s}" + ''.join(f"{tot[k]:8d}" for k in cats))


if __name__ == '__main__':
    main()

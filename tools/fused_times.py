"""Per smoothing call (nu = 2) on the slab / rod levels of config 3: the colour passes one launch each (k_line_colour,
k_line_wide) against all passes in one launch (k_line_fused, patches of w node planes). Back-to-back calls on one
level (warm caches: what a call costs in a cycle is measured by tools/level_times.py). Through gpurun:
    python tools/fused_times.py [n ...]          (levels 256 x n x n and permutations; default 4 8 16)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from emg3d_amd import _lib                      # noqa: E402
from microbench import make_level               # noqa: E402


def time_call(lv, lr, reps=40):
    for _ in range(3):
        lv.smooth(lr, 2)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        lv.smooth(lr, 2)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def main():
    lib = _lib.lib()
    ns = [int(a) for a in sys.argv[1:]] or [4, 8, 16]
    for n in ns:
        for shape, lrs in (((256, n, n), (2, 3)), ((n, 256, n), (1, 3)), ((n, n, 256), (1, 2))):
            lv, grid = make_level(0, 'triaxial', shape=shape)
            for lr in lrs:
                row = []
                for name, opts in (('colour', {}), ('wide', {'line_wide': 64})):
                    for k, v in opts.items():
                        lib.emg3d_set_option(k.encode(), v)
                    row.append(f"{name} {time_call(lv, lr):7.1f}")
                    for k in opts:
                        lib.emg3d_set_option(k.encode(), 0)
                lib.emg3d_set_option(b'line_fused', 17)
                for w in (4, 8, 16, 32, 64):
                    lib.emg3d_set_option(b'line_fused_w', w)
                    row.append(f"fused w={w} {time_call(lv, lr):7.1f}")
                lib.emg3d_set_option(b'line_fused', 0)
                lib.emg3d_set_option(b'line_fused_w', 8)
                print(f"{str(shape):>14s} lr={lr}  us per call of 7 passes:  " + '   '.join(row), flush=True)


if __name__ == '__main__':
    main()

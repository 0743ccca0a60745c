"""Where the cycles of a k_line_fused launch go: s_memtime stamps of workgroup 1 (thread 0) after every workgroup
barrier, from a -DEMG_FUSED_STAMPS build of the library (experiment build, not the product's). Through gpurun:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEMG_FUSED_STAMPS emg3d_amd/csrc/kernels.hip \
        -o emg3d_amd/lib/libemg3d_amd_fstamps.so          (here; the .so travels)
    python tools/fused_stamps.py [w]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emg3d_amd import _lib                      # noqa: E402
_lib.LIBPATH = os.path.join(ROOT, 'emg3d_amd', 'lib', 'libemg3d_amd_fstamps.so')
sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import make_level               # noqa: E402


def main():
    lib = _lib.lib()
    dbg = ctypes.CDLL(_lib.LIBPATH).emg3d_debug_fused_stamps
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lib.emg3d_set_option(b'line_fused', 17)
    lib.emg3d_set_option(b'line_fused_w', w)
    for shape, lr in (((256, 4, 4), 2), ((4, 4, 256), 1)):
        lv, grid = make_level(0, 'triaxial', shape=shape)
        for rep in range(3):
            lv.smooth(lr, 2)
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 128)()
        assert dbg(out) == 0
        st = np.array(out[:], dtype=np.int64)
        n = 3 + 7 * 5 + 1
        d = np.diff(st[:n])
        print(f"{shape} lr={lr} w={w}: total {st[n - 1] - st[0]} ticks; patch geometry {d[0]}, copy-in {d[1]}")
        for p in range(7):
            print(f"   pass {p + 1}: A {d[2 + 5 * p]:6d}  F {d[3 + 5 * p]:6d}  C {d[4 + 5 * p]:6d}  B {d[5 + 5 * p]:6d}  E {d[6 + 5 * p]:6d}")
        print(f"   owned planes out: {d[2 + 35]}")
        a = st[64:71]
        print(f"   pass 2, phase A of thread 0: line identity {a[1] - a[0]}, records issued {a[2] - a[1]}, right-hand side {a[3] - a[2]}, "
              f"wait for the records {a[4] - a[3]}, T from LDS + g {a[5] - a[4]}, to the barrier {a[6] - a[5]}; barrier released {st[3 + 5] - a[6]}")


if __name__ == '__main__':
    main()

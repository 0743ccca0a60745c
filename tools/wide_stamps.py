"""Where the cycles of a k_line_wide launch go: s_memtime stamps of workgroup 0 at the phase boundaries (thread 0 =
a block thread / chain lane of wave 0; thread 192 = a middle-block thread of wave 3), from a -DEMG_WIDE_STAMPS
build of the library (experiment build, not the product's). Through gpurun:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEMG_WIDE_STAMPS emg3d_amd/csrc/kernels.hip \
        -o emg3d_amd/lib/libemg3d_amd_stamps.so          (here; the .so travels)
    python tools/wide_stamps.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emg3d_amd import _lib                      # noqa: E402
_lib.LIBPATH = os.path.join(ROOT, 'emg3d_amd', 'lib', 'libemg3d_amd_stamps.so')
sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import make_level               # noqa: E402

NAMES = ['A rhs+records+g', 'barrier', 'F forward chain', 'barrier', 'C per block / middle', 'barrier',
         'B backward chain', 'barrier', 'E x + scatter']


def main():
    lib = _lib.lib()
    dbg = ctypes.CDLL(_lib.LIBPATH).emg3d_debug_wide_stamps
    lib.emg3d_set_option(b'line_wide', 64)
    for shape in ((256, 4, 4), (256, 8, 8), (256, 16, 16), (256, 32, 32)):
        lv, grid = make_level(0, 'triaxial', shape=shape)
        for lr in (2,):
            for rep in range(3):                 # third call: factors built, data as warm as a repeated call leaves it
                lv.smooth(lr, 1)
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 32)()
            assert dbg(out) == 0
            st = np.array(out[:], dtype=np.int64)
            for who, o in (('thread 0 (block thread, chain lane)', 0), ('thread 192 (middle block)', 16)):
                d = np.diff(st[o:o + 10])
                print(f"{shape} lr={lr} {who}: total {st[o + 9] - st[o]}  " +
                      '  '.join(f"{n}: {v}" for n, v in zip(NAMES, d)))
            # launch time with HIP events
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(50):
                lv.smooth(lr, 2)
            ev[1].record()
            torch.cuda.synchronize()
            print(f"{shape} lr={lr}: {ev[0].elapsed_time(ev[1]) / 50 / 7 * 1e3:.2f} us per launch (7 launches per call, back to back)")
            for w in (0, 64):
                lib.emg3d_set_option(b'line_wide', w)
                lv._factors = {}
                lv.smooth(lr, 2)
                torch.cuda.synchronize()
                ev[0].record()
                for _ in range(50):
                    lv.smooth(lr, 2)
                ev[1].record()
                torch.cuda.synchronize()
                print(f"   line_wide={w}: {ev[0].elapsed_time(ev[1]) / 50 / 7 * 1e3:.2f} us per launch")


if __name__ == '__main__':
    main()

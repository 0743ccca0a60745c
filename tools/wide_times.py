"""k_line_colour against k_line_wide, per launch, on the levels 256 x n x n of config 3 (back-to-back calls of nu = 2). Through gpurun: python tools/wide_times.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from emg3d_amd import _lib
from microbench import make_level


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


lib = _lib.lib()
for n in (2, 4, 8, 16, 32, 64):
    shape = (256, n, n)
    lv, grid = make_level(0, 'triaxial', shape=shape)
    for lr in (2, 3):
        row = []
        for name, opts in (('colour', {'line_wide': 0}), ('wide', {'line_wide': 64, 'line_wide_bt': 192}), ('wide 256', {'line_wide': 64, 'line_wide_bt': 256})):
            for k, v in opts.items():
                lib.emg3d_set_option(k.encode(), v)
            row.append(f"{name} {time_call(lv, lr) / 7:7.2f}")
        lib.emg3d_set_option(b'line_wide', 33); lib.emg3d_set_option(b'line_wide_bt', 0)
        print(f"{str(shape):>14s} lr={lr}  us per launch:  " + '   '.join(row), flush=True)

"""k_line_colour against k_line_wide with the N records fetched / formed in the kernel, per launch, on the levels 256 x n x n of config 3 (back-to-back calls of nu = 2). Through gpurun: python tools/wide_times.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from emg3d_amd import _lib
from microbench import make_level
from fused_times import time_call
lib = _lib.lib()
for n in (2, 4, 8, 16, 32):
    shape = (256, n, n)
    lv, grid = make_level(0, 'triaxial', shape=shape)
    for lr in (2, 3):
        row = []
        for name, opts in (('colour', {'line_wide': 0, 'line_lanes': 0}), ('wide', {'line_wide': 65, 'line_lanes': 0}), ('lanes', {'line_lanes': 33})):
            for k, v in opts.items():
                lib.emg3d_set_option(k.encode(), v)
            row.append(f"{name} {time_call(lv, lr) / 7:7.2f}")
        lib.emg3d_set_option(b'line_wide', 17); lib.emg3d_set_option(b'line_lanes', 17)
        print(f"{str(shape):>14s} lr={lr}  us per launch:  " + '   '.join(row), flush=True)

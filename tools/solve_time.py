"""Wall time of whole solves (upload, setup, cycles, download) of a bench workload.

    python tools/solve_time.py [--workload marine128] [--repeat 3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                       # noqa: E402
import emg3d_amd as emg3d          # noqa: E402
from bench import workload         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='marine128')
    ap.add_argument('--repeat', type=int, default=3)
    args = ap.parse_args()
    wl = workload(args.workload)
    grid = emg3d.TensorMesh(wl['h'], wl['origin'])
    model = emg3d.Model(grid, **wl['res'])
    sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
    for name, kw in (('multigrid', dict(sslsolver=False)), ('bicgstab+mg (default)', dict(sslsolver=True))):
        opts = dict(wl['opts'])
        opts.update(kw)
        for r in range(args.repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e, info = emg3d.solve(model, sfield, return_info=True, tol=1e-6, verb=0, **opts)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{args.workload} {name:24s} run {r}: {dt * 1e3:8.1f} ms  exit={info['exit']} "
                  f"it_mg={info['it_mg']} it_ssl={info['it_ssl']} rel_error={info['rel_error']:.2e}", flush=True)


if __name__ == '__main__':
    main()

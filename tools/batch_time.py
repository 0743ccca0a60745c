"""Wall time of 8 sources of one frequency on the 128^3 marine model (BASELINE.json config 4) on
ONE GPU with solve_batch (right-hand sides as one more grid dimension of every launch), for
several batch sizes; compare tools/survey_time.py (separate solves).

    python tools/batch_time.py [--batch 1,2,4,8] [--opt name=value]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                        # noqa: E402
import emg3d_amd as emg3d           # noqa: E402
from emg3d_amd import _lib          # noqa: E402
from bench import workload          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', default='1,2,4,8')
    ap.add_argument('--tol', type=float, default=1e-6)
    ap.add_argument('--opt', action='append', default=[])
    ap.add_argument('--ssl', action='store_true', help='BiCGSTAB + multigrid (the default of solve) instead of multigrid')
    args = ap.parse_args()
    for o in args.opt:
        k, v = o.split('=')
        assert _lib.lib().emg3d_set_option(k.encode(), int(v)) == 0, o
    wls = [workload('marine128', source_index=i) for i in range(8)]
    grid = emg3d.TensorMesh(wls[0]['h'], wls[0]['origin'])
    model = emg3d.Model(grid, **wls[0]['res'])
    opts = {k: v for k, v in wls[0]['opts'].items() if k != 'sslsolver'}
    opts.update(tol=args.tol, verb=0, sslsolver=bool(args.ssl))
    emg3d.solve_batch(model, [emg3d.get_source_field(grid, wls[0]['source'], wls[0]['frequency'])] * 2, **opts)
    for nb in [int(x) for x in args.batch.split(',')]:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its, work = [], 0.0
        for i0 in range(0, 8, nb):
            sfs = [emg3d.get_source_field(grid, w['source'], w['frequency']) for w in wls[i0:i0 + nb]]
            for sf in sfs:
                sf._trust_sparse = True
            for e, info in emg3d.solve_batch(model, sfs, **opts):
                its.append(info['it_mg'])
                work += info['smoother_cell_sweeps']
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"8 sources in batches of {nb}: {dt * 1e3:8.1f} ms  ({dt / 8 * 1e3:6.1f} ms per source, "
              f"{work / dt / 1e6:7.1f} Mcell-sweeps/s, cycles {its})", flush=True)


if __name__ == '__main__':
    main()

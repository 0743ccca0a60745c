"""Wall time of a small survey on ONE GPU: BASELINE.json config 4 (128^3 marine model, 8
x-dipoles, 1 Hz), every pair a complete solve (upload, setup, cycles to tol, download).

    python tools/survey_time.py [--per-gpu 1,3] [--tol 1e-6]

Compares separate hierarchies per pair (what the reference's process pool does) with one
hierarchy per frequency shared by the pairs of a worker (parallel.compute(reuse=True)), with
only the receiver responses leaving the GPU, and with the pairs solved together (batch=4, 8).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                        # noqa: E402
import emg3d_amd as emg3d           # noqa: E402
from emg3d_amd import parallel      # noqa: E402
from bench import workload          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--per-gpu', default='1,3')
    ap.add_argument('--tol', type=float, default=1e-6)
    args = ap.parse_args()
    wls = [workload('marine128', source_index=i) for i in range(8)]
    grid = emg3d.TensorMesh(wls[0]['h'], wls[0]['origin'])
    model = emg3d.Model(grid, **wls[0]['res'])
    sources = {f'S{i}': w['source'] for i, w in enumerate(wls)}
    freqs = {'f1': wls[0]['frequency']}
    opts = dict(wls[0]['opts'], sslsolver=False, tol=args.tol, verb=0)
    parallel.compute(model, grid, {'S0': sources['S0']}, freqs, opts)          # warm the process
    import numpy as np
    rng = np.random.default_rng(1)
    rec = (rng.uniform(-2000, 2000, 50), rng.uniform(-500, 500, 50), np.full(50, -990.), 0., 0.)
    parallel.compute(model, grid, {'S0': sources['S0']}, freqs, opts, receivers=rec)   # imports SciPy's interpolate
    for k in [int(x) for x in args.per_gpu.split(',')]:
        for reuse in (False, True, 'responses only'):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            kw = dict(receivers=rec, keep_fields=False) if reuse == 'responses only' else {}
            out = parallel.compute(model, grid, sources, freqs, opts, per_gpu=k, reuse=bool(reuse), **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            report(f"per_gpu={k}, reuse={reuse}", out, dt)
    for nb in (4, 8):
        for kw, tag in ((dict(), 'fields'), (dict(receivers=rec, keep_fields=False), 'responses only')):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = parallel.compute(model, grid, sources, freqs, opts, batch=nb, **kw)
            torch.cuda.synchronize()
            report(f"batch={nb}, {tag}", out, time.perf_counter() - t0)


def report(tag, out, dt):
    its = sorted(v[1]['it_mg'] for kk, v in out.items() if kk != '_all_info')
    work = sum(v[1]['smoother_cell_sweeps'] for kk, v in out.items() if kk != '_all_info')
    print(f"8 sources, {tag}: {dt * 1e3:8.1f} ms  ({dt / 8 * 1e3:6.1f} ms per source, "
          f"{work / dt / 1e6:7.1f} Mcell-sweeps/s, cycles {its})", flush=True)


if __name__ == '__main__':
    main()

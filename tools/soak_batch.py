"""Soak test (through gpurun): solve_batch against separate solves on random mid-size models -- three sources at one
frequency, random cycle / smoother options, multigrid alone (fields, cycle counts and error histories must be
bit-identical) or BiCGSTAB + multigrid (exit states equal, fields to 10 tol: the batch shares the direction cycling, so sources that
finish at different iterations see other preconditioner variants than on their own).
    SEED_BASE=... NSEEDS=... SSL=1 python tools/soak_batch.py"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import torch
import emg3d_amd as emg3d
from helpers import widths, relerr

bad = 0
t0 = time.time()
base = int(os.environ.get('SEED_BASE', 113000))
ssl = bool(int(os.environ.get('SSL', '0')))
for seed in range(int(os.environ.get('NSEEDS', 24))):
    rng = np.random.default_rng(base + seed)
    shape = tuple(int(rng.choice([16, 24, 32, 40, 48, 64, 80])) for _ in range(3))
    h = [widths(n // 2, n // 4, 25., float(rng.choice([1.03, 1.08, 1.15]))) for n in shape]
    grid = emg3d.TensorMesh(h, [-w.sum() / 2 for w in h])
    shape = grid.shape_cells
    blocks = tuple(max(n // 8, 1) for n in shape)
    rho = np.kron(10 ** rng.uniform(-0.5, 1.5, blocks), np.ones([-(-n // b) for n, b in zip(shape, blocks)]))
    rho = np.asfortranarray(rho[:shape[0], :shape[1], :shape[2]])
    if rng.integers(0, 4) == 0:
        rho[:, :, -max(shape[2] // 5, 1):] = 1e6
    case = int(rng.integers(0, 3))
    props = [(rho,), (rho, None, 2.0 * rho), (rho, 1.5 * rho, 2.5 * rho)][case]
    model = emg3d.Model(grid, *props)
    freq = float(rng.choice([2.0, 0.5, 0.1, -1.0]))
    sfs = [emg3d.get_source_field(grid, (float(rng.uniform(-80, 80)), float(rng.uniform(-80, 80)), float(rng.uniform(-80, 80)),
                                         float(rng.uniform(0, 90)), float(rng.uniform(-30, 30))), freq) for _ in range(3)]
    kw = dict(cycle=str(rng.choice(['V', 'W', 'F'])), semicoarsening=[False, True, 1, 23, 312][int(rng.integers(0, 5))],
              linerelaxation=[False, True, 2, 45, 7][int(rng.integers(0, 5))], maxit=30, tol=float(rng.choice([1e-6, 1e-9])),
              nu_pre=int(rng.integers(1, 4)), nu_post=int(rng.integers(1, 4)), sslsolver='bicgstab' if ssl else False)
    try:
        sep = [emg3d.solve(model, sf, return_info=True, **kw) for sf in sfs]
        bat = emg3d.solve_batch(model, sfs, **kw)
        msgs = []
        for b, ((e1, i1), (e2, i2)) in enumerate(zip(sep, bat)):
            same_exit = i1['exit'] == i2['exit']
            if ssl:
                ok = same_exit and (i1['exit'] != 0 or relerr(e2.field, e1.field) < 10 * kw['tol'])
            else:
                # (a mid-solve switch to the residual equation is taken by the whole batch: not bit-identical then)
                sw = 'switched' in (i1['residual_form'], i2['residual_form'])
                ok = same_exit and (sw or (i1['it_mg'] == i2['it_mg'] and np.array_equal(i1['error_at_cycle'], i2['error_at_cycle'])
                                           and np.array_equal(e1.field, e2.field)))
                if sw:
                    ok = ok and (i1['exit'] != 0 or relerr(e2.field, e1.field) < 1e-7)
            msgs.append('ok' if ok else 'DIFFERENT(%d: exit %d %d, cycles %d %d, fields %.1e)' % (
                b, i1['exit'], i2['exit'], i1['it_mg'], i2['it_mg'], relerr(e2.field, e1.field) if np.any(e1.field) else np.nan))
        good = all(m == 'ok' for m in msgs)
        bad += 0 if good else 1
        print('SEED', seed, shape, 'case', case, 'f', freq, kw, '| exits', [i['exit'] for _, i in sep], 'cycles', [i['it_mg'] for _, i in sep],
              'ok' if good else msgs, flush=True)
    except Exception as exc:
        bad += 1
        print('SEED', seed, shape, kw, 'EXC', repr(exc)[:300], flush=True)
    del model, sfs
    torch.cuda.empty_cache()
    if time.time() - t0 > float(os.environ.get('TIME_LIMIT', 600)):
        print('time limit at', seed); break
print('done, failures:', bad, 'seconds %.0f' % (time.time() - t0))

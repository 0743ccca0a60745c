"""Soak test (through gpurun): the streamed line kernel (k_line_stream: lines too long for LDS records, 16-line
workgroups) on random shapes -- 130 .. 300 blocks along the line, enough lines per colour class for 16 per
workgroup, odd and even counts, complex and real -- against the oracle in the same ordering.
    python tools/soak_stream.py"""
import sys, os, time
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import test_gpu_parity as t
from emg3d_amd import core
from oracle import core as ocore, mg_ref
bad = 0
t0 = time.time()
for seed in range(14):
    rng = np.random.default_rng(7000 + seed)
    long_ = int(rng.integers(130, 301))
    others = [int(rng.integers(92, 108)) for _ in range(2)]
    pos = int(rng.integers(0, 3))
    shape = tuple(others[:pos] + [long_] + others[pos:])
    case = str(rng.choice(['isotropic', 'VTI', 'triaxial']))
    freq = float(rng.choice([1.0, 0.1, -1.0]))
    h = [rng.uniform(5., 15., n) * 1.02 ** np.abs(np.arange(n) - n // 2) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sx = 10 ** rng.uniform(-1, 1, shape)
    sy = 10 ** rng.uniform(-1, 1, shape) if case == 'triaxial' else None
    sz = 10 ** rng.uniform(-1, 1, shape) if case in ('VTI', 'triaxial') else None
    vm = mg_ref.volume_model(grid, freq, sx, sy, sz)
    dtype = complex if freq > 0 else float
    s = mg_ref.Field(grid, dtype=dtype); e0 = mg_ref.Field(grid, dtype=dtype)
    for f in (s, e0):
        f.field[:] = rng.standard_normal(f.field.size)
        if dtype is complex:
            f.field[:] += 1j * rng.standard_normal(f.field.size)
    for f in (e0.fx[:, 0, :], e0.fx[:, -1, :], e0.fx[:, :, 0], e0.fx[:, :, -1], e0.fy[0], e0.fy[-1],
              e0.fy[:, :, 0], e0.fy[:, :, -1], e0.fz[0], e0.fz[-1], e0.fz[:, 0], e0.fz[:, -1]):
        f[...] = 0
    nu = int(rng.integers(1, 4))
    fn = t.SMOOTHERS[pos + 1]
    a, b = e0.copy(), e0.copy()
    getattr(ocore, fn)(a.fx, a.fy, a.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta, *grid.h, nu, order=1)
    getattr(core, fn)(b.fx, b.fy, b.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z, vm.zeta, *grid.h, nu)
    err = t.relerr(b.field, a.field)
    print('seed', seed, shape, case, freq, nu, fn, 'err %.2e' % err, flush=True)
    if not err < 1e-10:
        bad += 1
    if time.time() - t0 > 900:
        print('time limit at', seed); break
print('done, failures:', bad, 'seconds %.0f' % (time.time() - t0))

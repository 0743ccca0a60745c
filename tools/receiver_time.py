"""Time of the post-solve steps on a 128^3 field: magnetic field, 200 receivers (cubic),
device path against the same SciPy calls on the host (what the reference executes)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import emg3d_amd as emg3d
from bench import workload

wl = workload('marine128')
grid = emg3d.TensorMesh(wl['h'], wl['origin'])
model = emg3d.Model(grid, **wl['res'])
rng = np.random.default_rng(0)
e = emg3d.Field(grid, rng.standard_normal(grid.n_edges) + 1j * rng.standard_normal(grid.n_edges), frequency=1.0)
n = 200
rec = (rng.uniform(-2000, 2000, n), rng.uniform(-2000, 2000, n), rng.uniform(-1500, -500, n),
       rng.uniform(-180, 180, n), rng.uniform(-90, 90, n))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = emg3d.get_magnetic_field(model, e)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    re = e.get_receiver(rec)
    rh = h.get_receiver(rec)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"device: magnetic field {1e3 * (t1 - t0):7.1f} ms, 2 x 200 receivers (E and H, cubic) {1e3 * (t2 - t1):7.1f} ms", flush=True)
import scipy.ndimage as ndi
t0 = time.perf_counter()
c = ndi.spline_filter(e.fx.real, order=3, mode='constant')
t1 = time.perf_counter()
print(f"host scipy: spline prefilter of the real part of ONE component: {1e3 * (t1 - t0):7.1f} ms "
      f"(a receiver call filters 3 components x (re, im), E and H: x 12)")

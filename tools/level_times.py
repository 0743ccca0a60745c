"""Where a cycle's time goes, per level of the hierarchy: HIP events around every smoothing call, residual,
restriction and prolongation of every level during the bench's timed cycles (captured graphs off, so that
the calls of the coarse levels are launched one by one and can be bracketed). Through gpurun:
    EMG3D_AMD_GRAPHS=0 python tools/level_times.py [workload] [cycles] [opt=value ...]
Prints, per (level shape, call kind): calls and launches per cycle, microseconds per launch, ms per cycle.
The events cost ~2-3 us per bracketed call (host-side, hidden while the queue is full); the whole-cycle wall time
with and without the brackets is printed so that the sum can be judged against it."""
import os
import sys
import time
from collections import defaultdict

os.environ.setdefault('EMG3D_AMD_GRAPHS', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402
import emg3d_amd as emg3d                       # noqa: E402
from emg3d_amd import _lib, _device             # noqa: E402
import bench                                    # noqa: E402


def main():
    wlname = sys.argv[1] if len(sys.argv) > 1 else 'triaxial256'
    ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for o in sys.argv[3:]:
        k, v = o.split('=')
        assert _lib.lib().emg3d_set_option(k.encode(), int(v)) == 0, o
    wl = bench.workload(wlname)
    model = emg3d.Model(emg3d.TensorMesh(wl['h'], wl['origin']), **wl['res'])
    b = bench.Bench(wl, model, torch.device('cuda', 0))
    b.cycles(6)                                  # every (sc_dir, lr_dir) variant twice: levels and factors exist
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.cycles(ncyc)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / ncyc * 1e3

    rec = []
    skip = bool(_lib.lib().emg3d_get_option(b'skip_repeat'))

    def wrap(name, launches):
        orig = getattr(_device.DeviceLevel, name)

        def timed(self, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(self, *a, **kw)
            e1.record()
            rec.append((tuple(self.grid.shape_cells), name, a, launches(*a, **kw), e0, e1))
            return r
        setattr(_device.DeviceLevel, name, timed)

    wrap('smooth', lambda lr, nu: (4 * nu - ((nu - 1) if skip else 0)) if lr else 0)
    wrap('residual', lambda *a, **k: 1)
    wrap('restrict_to', lambda *a, **k: 1)
    wrap('prolong_from', lambda *a, **k: 1)
    b.hier.top.smooth = _device.DeviceLevel.smooth.__get__(b.hier.top)   # (the bench's own bracket off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.cycles(ncyc)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / ncyc * 1e3
    acc = defaultdict(lambda: [0, 0, 0.0])
    for shape, name, a, nl, e0, e1 in rec:
        key = (shape, name + (f" lr={a[0]}" if name == 'smooth' else ''))
        st = acc[key]
        st[0] += 1
        st[1] += nl
        st[2] += e0.elapsed_time(e1)
    print(f"# {wlname}: {plain:.2f} ms per cycle plain (eager), {wall:.2f} ms with the brackets; options {sys.argv[3:]}")
    print(f"{'level':>16s} {'call':>14s} {'calls/cyc':>9s} {'launch/cyc':>10s} {'us/launch':>10s} {'ms/cycle':>9s}")
    tot = 0.0
    per_level = defaultdict(float)
    for (shape, name), (calls, nl, ms) in sorted(acc.items(), key=lambda kv: (-kv[0][0][0] * kv[0][0][1] * kv[0][0][2], kv[0][1])):
        per = ms / ncyc
        tot += per
        per_level[shape] += per
        print(f"{str(shape):>16s} {name:>14s} {calls / ncyc:9.1f} {nl / ncyc:10.1f} {1e3 * ms / max(nl, 1) if nl else 0:10.2f} {per:9.3f}")
    print(f"# sum of brackets {tot:.2f} ms per cycle")
    for shape, ms in sorted(per_level.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
        print(f"#   {str(shape):>16s} {ms:8.3f} ms per cycle ({100 * ms / tot:5.1f} %)")


if __name__ == '__main__':
    main()

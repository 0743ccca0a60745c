"""Which operation of a batched config-5 solve takes seconds now and then (BENCH_r04: 2.9 / 2.5 s for two of the four
pairs solved together; reproduced by `bench.py --only-config5`: ONE cycle of a solve takes 2.5 - 3 s)? Every device
operation of the solve is bracketed by host clocks with a device synchronisation behind it; operations above 0.2 s
are printed with the level they ran on. Through gpurun:
    python tools/config5_stall.py [repeats] [batch]
STALL_SYNC=0: no synchronisation behind the operations; STALL_KEEP_CACHE=1: the caching allocator keeps its blocks
between the solves (no torch.cuda.empty_cache()); `malloc`: the allocation pattern alone (see malloc_probe)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import emg3d_amd as emg3d                       # noqa: E402
from emg3d_amd import _device, solver, models  # noqa: E402
import bench                                    # noqa: E402

SLOW = []
SYNC = os.environ.get('STALL_SYNC', '1') != '0'


def wrap(cls, name):
    orig = getattr(cls, name)

    def timed(self, *a, **kw):
        t0 = time.perf_counter()
        r = orig(self, *a, **kw)
        t1 = time.perf_counter()
        if SYNC:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        if t2 - t0 > 0.2:
            shape = tuple(self.grid.shape_cells) if hasattr(self, 'grid') else ''
            SLOW.append((name, shape, a[:2], round(t1 - t0, 3), round(t2 - t1, 3)))
        return r
    setattr(cls, name, timed)


def malloc_probe():
    """Freed device memory handed back to the driver, then allocated again: how long does hipMalloc take?"""
    dev = torch.device('cuda', 0)
    for rep in range(4):
        bufs = []
        slow = []
        for i in range(14):
            t0 = time.perf_counter()
            b = torch.empty(int(3.9 * 2 ** 30), dtype=torch.uint8, device=dev)
            t1 = time.perf_counter()
            b.fill_(1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            slow.append((round(t1 - t0, 3), round(t2 - t1, 3)))
            bufs.append(b)
        print(f"round {rep}: 14 x 3.9 GB (malloc s, fill s): {slow}", flush=True)
        del bufs, b
        t0 = time.perf_counter()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        print(f"   empty_cache: {time.perf_counter() - t0:.3f} s", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'malloc':
        return malloc_probe()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for n in ('smooth', 'residual', 'restrict_to', 'prolong_from', 'child', 'line_factors', 'zero_field', 'point_factors'):
        wrap(_device.DeviceLevel, n)
    for n in ('need_gs', 'need_ws', 'upload'):
        wrap(_device.Workspace, n)
    wls = [bench.workload('salt384', source_index=i) for i in range(8)]
    grid = emg3d.TensorMesh(wls[0]['h'], wls[0]['origin'])
    model = emg3d.Model(grid, **wls[0]['res'])
    opts = {k: v for k, v in wls[0]['opts'].items() if k != 'sslsolver'}
    opts.update(tol=1e-6, verb=0)
    dev = torch.device('cuda', 0)
    for rep in range(reps):
        for fi in (2, 3):
            pair = wls[2 * fi:2 * fi + nb]
            torch.cuda.synchronize()
            if os.environ.get('STALL_KEEP_CACHE', '0') == '0':
                torch.cuda.empty_cache()          # (what bench.py's block did in round 4: every call from freed memory)
            SLOW.clear()
            t0 = time.perf_counter()
            sfs = [emg3d.get_source_field(grid, w['source'], w['frequency']) for w in pair]
            if nb > 1:
                res = emg3d.solve_batch(model, sfs, keep_fields=False, **opts)
            else:
                res = [emg3d.solve(model, sfs[0], sslsolver=False, return_info=True, **opts)]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"rep {rep} f={pair[0]['frequency']}: {dt:.2f} s, cycles {[int(i['it_mg']) for _, i in res]}, "
                  f"reserved {torch.cuda.memory_reserved(dev) / 2 ** 30:.1f} GB; operations above 0.2 s "
                  f"(name, level, arguments, host seconds, seconds until the device was idle): {SLOW}", flush=True)


if __name__ == '__main__':
    main()

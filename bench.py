#!/usr/bin/env python
"""Benchmark of the multigrid inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload marine128|triaxial256|...]

One "step" = one multigrid cycle (pre-smoothing, residual, restriction, recursion,
prolongation, post-smoothing, residual norm) of BASELINE.json's configuration on one
synthetic model, with everything resident in HBM when the timed region starts. The
headline metric is BASELINE.json's: Mcells*smoother-iterations per second, i.e.

    work = sum over all smoother calls of the cycle of  nu * n_cells(level) [* directions]

divided by wall time, aggregated over all GPUs of the job (one independent source per
GPU: weak scaling). Rank 0 prints ONE JSON line, which also carries

* "roofline": algorithmic HBM bytes of the dominant kernel (a level-0 smoother; 184 B per
  cell-sweep for VTI) over its measured duration (HIP events on the
  launch stream, inside the timed region), against the 8 TB/s HBM peak;
* "smoothers_256" (1 GPU): the four smoothers on a 256^3 tri-axial level -- the kernel figure
  BASELINE.json's target (>= 40 % of the HBM roofline on gauss_seidel at 256^3) is stated for;
* "cpu_baseline": the oracle (C restatement of the reference's sequential numba kernels +
  its multigrid driver, oracle/) timed on this host on one cycle of the same workload.

Multi-GPU: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`; the model is broadcast from rank 0 over RCCL, no collective inside a solve.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver of these boxes supports dmabuf IPC only: RCCL / device-tensor sharing across
# processes needs it (already exported there; kept for any environment built from here)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_CELL_SWEEP = {       # complex fp64, SURVEY.md Appendix C
    'isotropic': 168, 'VTI': 184, 'HTI': 184, 'triaxial': 200}


def widths(ncore, npad, width, factor):
    pad = width * factor ** (np.arange(npad) + 1.0)
    return np.r_[pad[::-1], np.full(ncore, float(width)), pad]


# ------------------------------------------------------------------------- workloads ---
def workload(name, source_index=0):
    """Grid widths, origin, resistivities (host arrays), source and solver settings of the
    BASELINE.json configurations."""
    if name in ('marine128', 'marine64', 'marine32'):
        # config 2 (and reduced copies of it for quick checks): stretched marine halfspace,
        # VTI sediments, deep water; F-cycle + semicoarsening + line relaxation
        n = int(name[6:])
        hx = widths(n // 2, n // 4, 50., 1.05 if n == 128 else 1.1)
        hz = widths(n // 2, n // 4, 25., 1.06 if n == 128 else 1.12)
        core_z = (n // 2) * 25.
        ztop = 0.1875 * core_z                      # 300 m above sea level at n = 128
        origin = (-hx.sum() / 2, -hx.sum() / 2, ztop - core_z - hz[:n // 4].sum())
        zc = origin[2] + np.cumsum(hz) - hz / 2
        rh = np.where(zc > -1000 * n / 128, 0.3, 1.0)
        rv = np.where(zc > -1000 * n / 128, 0.3, 2.0)
        shape = (n, n, n)
        res = {'property_x': np.broadcast_to(rh[None, None, :], shape),
               'property_z': np.broadcast_to(rv[None, None, :], shape)}
        # config 4: 8 x-dipoles at x = -1400 ... +1400 step 400
        sx = (-1400. + 400. * (source_index % 8)) * n / 128 if source_index else 0.
        src = (sx, 0., -950. * n / 128, 0., 0.)
        opts = dict(cycle='F', semicoarsening=True, linerelaxation=True)
        return dict(h=[hx, hx, hz], origin=origin, res=res, source=src, frequency=1.0,
                    opts=opts, case='VTI', label=f"{n}^3 stretched marine halfspace, VTI, "
                    "x-dipole, 1 Hz, F-cycle + semicoarsening + line relaxation")
    if name in ('triaxial256', 'triaxial128', 'triaxial64'):
        # config 3: stretched grid, blocky tri-axial model, W-cycle + sc + lr
        n = int(name[8:])
        h = widths(n // 2, n // 4, 25., 1.03 if n == 256 else 1.06)
        origin = (-h.sum() / 2,) * 3
        rng = np.random.default_rng(20260928)
        lat = 10 ** rng.uniform(-0.5, 1.5, (16, 16, 16))
        px = np.kron(lat, np.ones((n // 16,) * 3))
        res = {'property_x': px, 'property_y': 1.5 * px, 'property_z': 2.5 * px}
        opts = dict(cycle='W', semicoarsening=True, linerelaxation=True)
        return dict(h=[h, h, h], origin=origin, res=res, source=(0., 0., 0., 0., 0.),
                    frequency=1.0, opts=opts, case='triaxial',
                    label=f"{n}^3 stretched grid, tri-axial blocky model, W-cycle + "
                    "semicoarsening + line relaxation")
    if name in ('uniform256', 'uniform128', 'uniform32'):
        # config 1 family / the reference's own benchmark (docs/dev/tests.rst:193-219):
        # uniform fullspace, plain F-cycle: exercises the POINT smoother
        n = int(name[7:])
        h = np.full(n, 50.)
        res = {'property_x': np.ones((n, n, n))}
        opts = dict(cycle='F', semicoarsening=False, linerelaxation=False)
        return dict(h=[h, h, h], origin=(-25. * n,) * 3, res=res, source=(0., 0., 0., 0., 0.),
                    frequency=1.0, opts=opts, case='isotropic',
                    label=f"{n}^3 uniform fullspace, plain F-cycle (point smoother)")
    if name in ('salt384', 'salt96'):
        # config 5: 384 x 256 x 256 (or a quarter-size copy), isotropic: sea water above
        # z = -1000 m, sediments 1 -> 3 Ohm m linear with depth, an analytic "salt" body =
        # union of three ellipsoids at 100 Ohm m (deterministic stand-in for the SEG/EAGE salt
        # model, which is not part of the reference repository); 4 frequencies x 2 sources =
        # 8 independent (source, frequency) pairs, pair index = source_index
        q = 1 if name == 'salt384' else 4
        hx = widths(256 // q, 64 // q, 50. * q, 1.04 ** q)
        hy = widths(128 // q, 64 // q, 50. * q, 1.04 ** q)
        origin = (-hx.sum() / 2, -hy.sum() / 2, -hy[:64 // q].sum() - 5400.)
        xc, yc, zc = (o + np.cumsum(h) - h / 2 for o, h in zip(origin, (hx, hy, hy)))
        X, Y, Z = np.meshgrid(xc, yc, zc, indexing='ij')
        rho = np.where(Z > -1000., 0.3, 1.0 + 2.0 * np.clip((-1000. - Z) / 5000., 0., 1.))
        for cx, cy, cz, ax, ay, az in ((-800., 0., -3000., 2200., 1500., 900.),
                                       (1500., 600., -3600., 1400., 1800., 700.),
                                       (200., -900., -2400., 900., 700., 500.)):
            inside = ((X - cx) / ax) ** 2 + ((Y - cy) / ay) ** 2 + ((Z - cz) / az) ** 2 < 1.
            rho = np.where(inside & (Z <= -1000.), 100., rho)
        freq = (0.25, 0.5, 1.0, 2.0)[(source_index // 2) % 4]
        src = (-2000. if source_index % 2 == 0 else 2000., 0., -950., 0., 0.)
        opts = dict(cycle='F', semicoarsening=True, linerelaxation=True)
        return dict(h=[hx, hy, hy], origin=origin, res={'property_x': rho}, source=src,
                    frequency=freq, opts=opts, case='isotropic',
                    label=f"{hx.size} x {hy.size} x {hy.size} salt-like isotropic model, x-dipole at "
                    f"x = {src[0]:+.0f} m, {freq} Hz, F-cycle + semicoarsening + line relaxation")
    raise ValueError(f"unknown workload {name!r}")


# -------------------------------------------------------------------------- GPU side ---
class Bench:
    def __init__(self, wl, device):
        import torch
        import emg3d_amd as emg3d
        from emg3d_amd import solver
        self.torch, self.solver = torch, solver
        grid = emg3d.TensorMesh(wl['h'], wl['origin'])
        model = emg3d.Model(grid, **wl['res'])
        self.sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        vmodel = emg3d.models.VolumeModel(model, self.sfield)
        self.grid, self.case = grid, model.case
        self.hier = solver.Hierarchy(vmodel, device)
        self.efield = emg3d.Field(grid, frequency=wl['frequency'])
        self.hier.upload(self.sfield, self.efield)
        self.var = solver.MGParameters(verb=0, sslsolver=False, shape_cells=grid.shape_cells,
                                       tol=0.0, maxit=10 ** 9, **wl['opts'])
        self.var.l2_refe = float(np.linalg.norm(self.sfield.field))
        self.events = []          # (lr, nu, start_event, end_event) of level-0 smoother calls
        from emg3d_amd import _lib
        self.skip_repeat = bool(_lib.lib().emg3d_get_option(b'skip_repeat'))
        self._instrument()

    def _instrument(self):
        """Record HIP events (on the launch stream = torch's current stream) around every
        level-0 smoother call so the dominant kernel's duration is measured live."""
        top, torch = self.hier.top, self.torch
        orig = top.smooth

        def timed_smooth(lr, nu):
            if not self.recording:
                return orig(lr, nu)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            orig(lr, nu)
            b.record()
            self.events.append((lr, nu, a, b))
        self.recording = False
        top.smooth = timed_smooth

    def cycles(self, n):
        self.var.fixed_cycles = n
        self.solver._multigrid(self.hier.top, self.var, 0, 0)

    def kernel_stats(self):
        """Per level-0 smoother kernel: launches, total ms, mean ms per launch (one launch
        = one colour pass = a quarter sweep)."""
        stats = {}
        for lr, nu, a, b in self.events:
            ms = a.elapsed_time(b)
            st = stats.setdefault(lr, {'launches': 0, 'ms': 0.0})
            # colour passes actually launched: consecutive sweeps meet at one colour class, and the
            # library does not repeat that pass (it would reproduce the same values; option
            # skip_repeat) -- nu sweeps = 4 nu - (nu - 1) launches
            st['launches'] += 4 * nu - ((nu - 1) if self.skip_repeat else 0)
            st['ms'] += ms
        return stats


def smoothers_256(device, n=256, nu=2, reps=5):
    """BASELINE.json's north-star kernel figure: each smoother on a 256^3 tri-axial level
    (random model and fields, complex fp64), nu sweeps per call, HIP events on the launch
    stream; algorithmic bytes = 200 B per cell and sweep."""
    import torch
    from emg3d_amd._device import DeviceLevel
    import emg3d_amd as emg3d
    shape = (n, n, n)
    rng = np.random.default_rng(1)
    h = [widths(n // 2, n // 4, 25., 1.03)] * 3
    grid = emg3d.TensorMesh(h, (0, 0, 0))
    vol = grid.cell_volumes.reshape(shape, order='F')
    smu0 = 2j * np.pi * 1.25663706127e-06

    class VM:
        pass
    vm = VM()
    vm.grid, vm.case = grid, 'triaxial'
    sig = 10 ** rng.uniform(-1.5, 0.5, shape)
    vm.eta_x = np.asfortranarray(-smu0 * vol * sig)
    vm.eta_y = np.asfortranarray(vm.eta_x / 1.5)
    vm.eta_z = np.asfortranarray(vm.eta_x / 2.5)
    vm.zeta = np.asfortranarray(vol)
    lv = DeviceLevel.from_host(vm, device)
    gen = torch.Generator(device=device).manual_seed(1)
    for t in (lv.e, lv.s):
        t.copy_(torch.complex(torch.randn(grid.n_edges, generator=gen, device=device, dtype=torch.float64),
                              torch.randn(grid.n_edges, generator=gen, device=device, dtype=torch.float64)))
    lv.pec_zero()
    out = {}
    names = {0: 'gauss_seidel (k_gs_point_tile)', 1: 'gauss_seidel_x', 2: 'gauss_seidel_y',
             3: 'gauss_seidel_z'}
    for lr in (0, 1, 2, 3):
        lv.smooth(lr, nu)                       # builds factors, warms up
        lv.smooth(lr, nu)
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            lv.smooth(lr, nu)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = float(np.median(ts)) / nu
        gbs = BYTES_PER_CELL_SWEEP['triaxial'] * grid.n_cells / (ms * 1e-3) / 1e9
        out[names[lr]] = {'ms_per_sweep': ms, 'achieved': gbs, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                          'gcell_sweeps_per_s': grid.n_cells / (ms * 1e-3) / 1e9}
        lv._factors.pop(lr, None)               # 15 GB of line factors per direction: free them
        torch.cuda.empty_cache()
    return {'level': f'{n}^3 tri-axial, complex fp64, {nu} sweeps per call',
            'note': ('line smoothers / plain point smoother: a call of nu sweeps launches 4 nu - (nu - 1) colour '
                     'passes -- the pass that would repeat the previous sweep\'s last colour class reproduces the '
                     'same values bit by bit and is not launched (option skip_repeat); the tiled point smoother '
                     'launches all of its passes'),
            'bytes_per_cell_sweep': BYTES_PER_CELL_SWEEP['triaxial'], 'peak': HBM_PEAK_GBS, 'smoothers': out}


def survey_8(device):
    """BASELINE.json config 4 on ONE GPU: the 8 sources of the 128^3 marine model (1 Hz) as whole
    solves to tol 1e-6 -- one after the other, and solved together (solver.solve_batch: the
    right-hand sides as one more grid dimension of every launch; identical fields). Reported for
    information; the judged `value` above is the single-source cycle."""
    import torch
    import emg3d_amd as emg3d
    wls = [workload('marine128', source_index=i) for i in range(8)]
    grid = emg3d.TensorMesh(wls[0]['h'], wls[0]['origin'])
    model = emg3d.Model(grid, **wls[0]['res'])
    opts = {k: v for k, v in wls[0]['opts'].items() if k != 'sslsolver'}
    opts.update(tol=1e-6, verb=0)
    out = {'workload': 'config 4: 8 x-dipoles, 128^3 marine model, 1 Hz, whole solves to tol 1e-6'}
    for tag, nb in (('one_by_one', 1), ('solved_together', 8)):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        its, work = [], 0.0
        for i0 in range(0, 8, nb):
            sfs = [emg3d.get_source_field(grid, w['source'], w['frequency']) for w in wls[i0:i0 + nb]]
            if nb == 1:
                res = [emg3d.solve(model, sfs[0], sslsolver=False, return_info=True, **opts)]
            else:
                res = emg3d.solve_batch(model, sfs, **opts)
            for _, info in res:
                its.append(int(info['it_mg']))
                work += info['smoother_cell_sweeps']
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out[tag] = {'seconds': dt, 'ms_per_source': dt / 8 * 1e3, 'Mcell_sweeps_per_s': work / dt / 1e6,
                    'cycles': its}
    return out


def pmc_traffic(workload_name, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
    of this very command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950);
    None if there is no entry for this workload and kernel."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f)[workload_name][kernel]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def run_gpu(args, rank, world):
    import torch
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', 0))
    # EMG3D_BENCH_BACKEND=gloo: dry run of the multi-rank code path on a box with fewer GPUs
    # than ranks (all ranks share the visible devices, collectives go through host tensors)
    backend = os.environ.get('EMG3D_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    cdev = device if backend == 'nccl' else torch.device('cpu')     # where collectives run
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    wl = workload(args.workload, source_index=rank if world > 1 else 0)
    if world > 1:
        # the model is built on rank 0 and broadcast over RCCL/xGMI; every rank builds its
        # own eta(f), zeta and source locally
        for k in sorted(wl['res']):
            t = torch.from_numpy(np.ascontiguousarray(wl['res'][k], dtype=np.float64)).to(cdev)
            if rank != 0:
                t.zero_()
            dist.broadcast(t, 0)
            wl['res'][k] = t.cpu().numpy()
    b = Bench(wl, device)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Untimed preparation, independent of --warmup: every (sc_dir, lr_dir) variant of the
    # cycle builds its coarse levels and line factors on first use and captures its
    # coarse-grid HIP graph on its third use (solver._GRAPH_AFTER). Three passes over the
    # variants put the solver in its steady state, like a solve that is a few cycles old.
    b.cycles((b.solver._GRAPH_AFTER + 1) * b.var.maxcycle)
    if args.warmup > 0:
        b.cycles(args.warmup)
    w0 = b.var.smoother_cell_sweeps
    b.recording = True
    sync()
    t0 = time.perf_counter()
    b.cycles(args.steps)
    sync()
    dt = time.perf_counter() - t0
    b.recording = False
    work = b.var.smoother_cell_sweeps - w0
    l2 = b.var.l2 / b.var.l2_refe

    tt = torch.tensor([dt, float(work)], dtype=torch.float64, device=cdev)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, work_all = tmax[0].item(), tsum[1].item()
    else:
        dt_max, work_all = dt, float(work)

    out = None
    if rank == 0:
        stats = b.kernel_stats()
        n0 = b.grid.n_cells
        names = {0: 'k_gs_point', 1: 'k_gs_line<x>', 2: 'k_gs_line<y>', 3: 'k_gs_line<z>'}
        dom = max(stats, key=lambda k: stats[k]['ms'])
        bytes_per_launch = BYTES_PER_CELL_SWEEP[b.case] * n0 / 4.0
        ms_launch = stats[dom]['ms'] / stats[dom]['launches']
        achieved = bytes_per_launch / (ms_launch * 1e-3) / 1e9
        out = {
            'metric': 'Mcells*smoother-iters/s (fp64) per multigrid cycle',
            'value': work_all / dt_max / 1e6,
            'unit': 'Mcell-sweeps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt_max / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'c128 (complex fp64)' if b.hier.top.is_complex else 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload, 'description': wl['label'],
                       'cells': n0, 'cycle': wl['opts']['cycle'],
                       'cell_sweeps_per_step': work / args.steps,
                       'rel_error_after_run': l2,
                       'parallelism': f'{world} independent sources, 1 per GPU'},
            'roofline': {
                'bound': 'hbm', 'kernel': names[dom],
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic(args.workload, names[dom]),
                'bytes_per_launch': bytes_per_launch, 'ms_per_launch': ms_launch,
                'launches_timed': stats[dom]['launches'],
                'gcell_sweeps_per_s': n0 / 4.0 / (ms_launch * 1e-3) / 1e9,
                'all_level0_smoothers': {names[k]: {
                    'ms_per_launch': v['ms'] / v['launches'], 'launches': v['launches'],
                    'GB/s': bytes_per_launch / (v['ms'] / v['launches'] * 1e-3) / 1e9}
                    for k, v in stats.items()}},
        }
    if rank == 0 and world == 1 and not args.no_256:
        del b
        torch.cuda.empty_cache()
        out['smoothers_256'] = smoothers_256(device)
    if rank == 0 and world == 1 and not args.no_survey:
        torch.cuda.empty_cache()
        try:
            out['survey_8_sources'] = survey_8(device)
        except Exception as exc:        # informational block: never takes the bench line down
            out['survey_8_sources'] = {'error': repr(exc)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out, wl


# -------------------------------------------------------------------------- CPU side ---
def run_cpu_baseline(wl):
    """One multigrid cycle of the oracle (sequential C restatement of the reference's
    kernels, lexicographic order, one thread) on the same workload."""
    from oracle import mg_ref
    grid = mg_ref.Grid(wl['h'], wl['origin'])
    cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in wl['res'].items()}
    vm = mg_ref.volume_model(grid, wl['frequency'], cond['property_x'], cond.get('property_y'),
                             cond.get('property_z'))
    import emg3d_amd as emg3d
    sf = emg3d.get_source_field(emg3d.TensorMesh(wl['h'], wl['origin']), wl['source'],
                                wl['frequency'])
    s = mg_ref.Field(grid, sf.field.copy())
    t0 = time.perf_counter()
    _, info = mg_ref.solve(vm, s, maxit=1, **wl['opts'])
    dt = time.perf_counter() - t0
    return {'value': info['smooth_work'] / dt / 1e6, 'unit': 'Mcell-sweeps/s', 'cores': 1,
            'kind': 'port', 'seconds': dt,
            'sample': f"1 multigrid cycle of the same workload ({info['smooth_work']:.3g} "
                      "cell-sweeps), oracle/ C kernels in the reference's sequential order, "
                      f"1 thread of {os.cpu_count()} host cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='marine128')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--opt', action='append', default=[],
                    help='library tuning option name=value (emg3d_set_option), for experiments')
    ap.add_argument('--no-survey', action='store_true', help='skip the 8-source survey block')
    ap.add_argument('--no-256', action='store_true',
                    help="skip the separate 256^3 smoother measurement ('smoothers_256')")
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node "
                         f"{args.gpus} bench.py --gpus {args.gpus}")
    if args.opt:
        from emg3d_amd import _lib
        for o in args.opt:
            k, v = o.split('=')
            if _lib.lib().emg3d_set_option(k.encode(), int(v)) != 0:
                raise SystemExit(f"unknown option {o}")
    out, wl = run_gpu(args, rank, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = run_cpu_baseline(workload(args.workload))
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))


if __name__ == '__main__':
    main()

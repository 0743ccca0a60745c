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

* "roofline": algorithmic HBM bytes of the dominant kernel PER LAUNCH (a level-0 line smoother
  colour pass; 200 B per cell-sweep tri-axial) over its measured duration
  (HIP events on the launch stream, inside the timed region), against the 8 TB/s HBM peak;
  "roofline.north_star_kernel": the same for core.gauss_seidel (k_gs_point_tile) at 256^3;
* "smoothers_256" (1 GPU): the four smoothers on a 256^3 tri-axial level -- the kernel figure
  BASELINE.json's target (>= 40 % of the HBM roofline on gauss_seidel at 256^3) is stated for;
* "cpu_baseline": the oracle (C restatement of the reference's sequential numba kernels,
  oracle/) timed on this host: one thread on a bounded sample of the same workload, and all
  cores in throughput mode (one job per core), CPU model and core count stated.

The default workload is BASELINE.json config 3 (256^3 tri-axial, W-cycle + semicoarsening + line
relaxation), the largest single-GPU configuration.

Multi-GPU: `python bench.py --gpus N` spawns its N ranks itself (torch.distributed.run on
127.0.0.1); a launcher that already set RANK / WORLD_SIZE is honoured. One independent source
per rank, the model broadcast from rank 0 over RCCL, no collective inside a solve.
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
def workload(name, source_index=0, with_model=True):
    """Grid widths, origin, resistivities (host arrays), source and solver settings of the
    BASELINE.json configurations. with_model=False: everything but the
    resistivity arrays (`res` is None) -- ranks > 0 of a multi-GPU run receive the model by
    broadcast and never build it."""
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
               'property_z': np.broadcast_to(rv[None, None, :], shape)} if with_model else None
        # config 4: 8 x-dipoles at x = -1400 ... +1400 step 400
        sx = (-1400. + 400. * (source_index % 8)) * n / 128 if source_index else 0.
        src = (sx, 0., -950. * n / 128, 0., 0.)
        opts = dict(cycle='F', semicoarsening=True, linerelaxation=True)
        return dict(h=[hx, hx, hz], origin=origin, res=res, source=src, frequency=1.0,
                    opts=opts, case='VTI', label=f"{n}^3 stretched marine halfspace, VTI, "
                    "x-dipole, 1 Hz, F-cycle + semicoarsening + line relaxation")
    if name in ('triaxial512', 'triaxial256', 'triaxial128', 'triaxial64'):
        # config 3: stretched grid, blocky tri-axial model, W-cycle + sc + lr
        n = int(name[8:])
        h = widths(n // 2, n // 4, 25., {512: 1.015, 256: 1.03}.get(n, 1.06))
        origin = (-h.sum() / 2,) * 3
        res = None
        if with_model:
            rng = np.random.default_rng(20260928)
            lat = 10 ** rng.uniform(-0.5, 1.5, (16, 16, 16))
            px = np.kron(lat, np.ones((n // 16,) * 3))
            res = {'property_x': px, 'property_y': 1.5 * px, 'property_z': 2.5 * px}
        opts = dict(cycle='W', semicoarsening=True, linerelaxation=True)
        # rank r of a multi-GPU run solves its own source: x-dipoles 100 m apart
        return dict(h=[h, h, h], origin=origin, res=res, source=(100. * source_index, 0., 0., 0., 0.),
                    frequency=1.0, opts=opts, case='triaxial',
                    label=f"{n}^3 stretched grid, tri-axial blocky model, W-cycle + "
                    "semicoarsening + line relaxation")
    if name in ('uniform256', 'uniform128', 'uniform32'):
        # config 1 family / the reference's own benchmark (docs/dev/tests.rst:193-219):
        # uniform fullspace, plain F-cycle: exercises the POINT smoother
        n = int(name[7:])
        h = np.full(n, 50.)
        res = {'property_x': np.ones((n, n, n))} if with_model else None
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
        rho = None
        if with_model:
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
        return dict(h=[hx, hy, hy], origin=origin, res={'property_x': rho} if with_model else None, source=src,
                    frequency=freq, opts=opts, case='isotropic',
                    label=f"{hx.size} x {hy.size} x {hy.size} salt-like isotropic model, x-dipole at "
                    f"x = {src[0]:+.0f} m, {freq} Hz, F-cycle + semicoarsening + line relaxation")
    raise ValueError(f"unknown workload {name!r}")


# -------------------------------------------------------------------------- GPU side ---
class Bench:
    def __init__(self, wl, model, device, line_compact=None):
        import torch
        import emg3d_amd as emg3d
        from emg3d_amd import solver
        self.torch, self.solver = torch, solver
        grid = model.grid
        self.model = model
        self.sfield = emg3d.get_source_field(grid, wl['source'], wl['frequency'])
        vmodel = emg3d.models.VolumeModel(model, self.sfield)
        self.grid, self.case = grid, model.case
        self.hier = solver.Hierarchy(vmodel, device, line_compact=line_compact)
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


def cycle_fp64_storage(wl, model, device, steps):
    """The same timed cycles with every coefficient record in fp64 (Hierarchy(line_compact=False): round 5's storage, the
    finest level in direct form) next to the headline, which runs the solver's defaults -- single-precision STORAGE of the
    streamed line passes' T / w records and of the coarse levels' eta sums where the model's block condition allows it,
    all arithmetic fp64, converged fields and cycle counts those of the fp64 oracle (tests, profiles/r06_full_size_converged.txt)."""
    import torch
    b = Bench(wl, model, device, line_compact=False)
    b.cycles((b.solver._GRAPH_AFTER + 1) * b.var.maxcycle)
    w0 = b.var.smoother_cell_sweeps
    b.recording = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.cycles(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.recording = False
    stats = b.kernel_stats()
    dom = max(stats, key=lambda k: stats[k]['ms'])
    ms_launch = stats[dom]['ms'] / stats[dom]['launches']
    bpl = BYTES_PER_CELL_SWEEP[b.case] * b.grid.n_cells / 4.0
    return {'line_factor_storage': 'fp64', 'residual_form': bool(getattr(b.var, 'residual_form', False)), 'steps': steps,
            'ms_per_step': dt / steps * 1e3, 'value': (b.var.smoother_cell_sweeps - w0) / dt / 1e6,
            'dominant_level0_kernel': {1: 'k_gs_line<x>', 2: 'k_gs_line<y>', 3: 'k_gs_line<z>', 0: 'k_gs_point'}[dom],
            'ms_per_launch': ms_launch, 'frac': bpl / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS}


def line_kernel_name(lr, shape, batch=1, is_complex=True):
    """The HIP kernel that runs a colour pass of line direction lr (1/2/3) on a level of `shape` cells
    (what rocprofv3 lists), as the library's own launcher decides it under the current options
    (emg3d_line_kernel_name: csrc/kernels.hip, line_plan)."""
    from emg3d_amd import _lib
    nx, ny, nz = (int(n) for n in shape)
    return _lib.lib().emg3d_line_kernel_name(int(lr), nx, ny, nz, int(bool(is_complex)), int(batch)).decode()


def smoothers_256(device, n=256, nu=2, reps=5):
    """BASELINE.json's north-star kernel figure: each smoother on a 256^3 tri-axial level
    (random model and fields, complex fp64), nu sweeps per call, HIP events on the launch
    stream; algorithmic bytes = 200 B per cell and sweep.

    Per smoother two fractions of the 8 TB/s roofline are reported:
    * ``frac`` -- PER LAUNCH: algorithmic bytes of the work the launches of a call execute,
      over the call's duration. A call of nu sweeps does not execute 4 nu colour passes: the
      pass that would repeat the previous sweep's last colour class reproduces the same values
      bit by bit and is not launched (library option skip_repeat; 4 nu - (nu - 1) passes), and
      the tiled point smoother runs the tiles where two sweeps meet once for both sweeps, minus
      the one repeated node colour (31 of 32 quarter-tile colour steps for nu = 2);
    * ``frac_delivered`` -- nu whole sweeps credited (what a caller gets per second)."""
    import torch
    from emg3d_amd._device import DeviceLevel
    from emg3d_amd import _lib
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
    lib = _lib.lib()
    skip = bool(lib.emg3d_get_option(b'skip_repeat'))
    fuse = bool(lib.emg3d_get_option(b'tile_fuse'))
    # (the tiles where two sweeps meet drop a repeated node colour only under the mirrored node-colour rule)
    skip_node = skip and lib.emg3d_get_option(b'point_order') == 0
    out = {}
    lk = {lr: line_kernel_name(lr, shape) for lr in (1, 2, 3)}
    names = {0: 'gauss_seidel (k_gs_point_tile)', 1: f'gauss_seidel_x ({lk[1]}<0>)',
             2: f'gauss_seidel_y ({lk[2]}<1>)', 3: f'gauss_seidel_z ({lk[3]}<2>)'}
    # the line smoothers twice: fp64 line records, and COMPACT ones (single-precision T and w records, all arithmetic
    # fp64: library option line_compact = 1 here, the level flag in a solve) -- what solver.Hierarchy uses by default
    # where the model's block condition allows it (the bench workloads: yes)
    for lr, compact in ((0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (1, 1), (2, 1), (3, 1)):
        lib.emg3d_set_option(b'line_compact' if lr else b'point_compact', compact)
        for key in [k for k in lv._factors if isinstance(k, tuple) and k[0] == 'point']:
            del lv._factors[key]                # (the eta sums are laid out for the storage they were built with)
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
        ms_call = float(np.median(ts))
        if lr == 0:
            launches = 4 * nu - ((nu - 1) if fuse else 0)
            # executed share of the 4 nu tile-pair passes x 4 node colours
            executed = (16 * nu - ((nu - 1) if (fuse and skip_node) else 0)) / (16.0 * nu)
        else:
            launches = 4 * nu - ((nu - 1) if skip else 0)
            executed = launches / (4.0 * nu)
        full = BYTES_PER_CELL_SWEEP['triaxial'] * grid.n_cells * nu          # nu whole sweeps
        gbs_exec = full * executed / (ms_call * 1e-3) / 1e9
        gbs_deliv = full / (ms_call * 1e-3) / 1e9
        # secondary check: fp64 rate of the work executed. Operations per cell and sweep
        # counted from the kernels' instruction mix (tools/isa_mix.py): point 546 fp64 instructions per
        # node, lines 4 lanes x (51 forward + 51 backward) + ~130 right-hand side per block; ~85 % of
        # them fused multiply-adds -> ~1.0 kflop per cell-sweep either way (the reference: ~1.1 kflop)
        kflop = 1.01 if lr == 0 else 1.0
        tflops = kflop * 1e3 * grid.n_cells * nu * executed / (ms_call * 1e-3) / 1e12
        out[names[lr] + (', compact records' if compact else '')] = {
                          'ms_per_call': ms_call, 'launches_per_call': launches,
                          'approx_fp64_tflops': tflops, 'approx_fp64_frac_of_78.6_vector_peak': tflops / 78.6,
                          'ms_per_launch': ms_call / launches, 'ms_per_delivered_sweep': ms_call / nu,
                          'achieved': gbs_exec, 'unit': 'GB/s', 'frac': gbs_exec / HBM_PEAK_GBS,
                          'achieved_delivered': gbs_deliv, 'frac_delivered': gbs_deliv / HBM_PEAK_GBS,
                          'gcell_sweeps_per_s_delivered': grid.n_cells * nu / (ms_call * 1e-3) / 1e9}
        lv._factors.pop(lr, None)               # 5 GB of line factors per direction: back to the caching allocator
        lib.emg3d_set_option(b'line_compact' if lr else b'point_compact', 0)
    return {'level': f'{n}^3 tri-axial, complex fp64, {nu} sweeps per call',
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


def survey_config5(device, name='salt384', repeats=2):
    """BASELINE.json config 5 on ONE GPU: 4 frequencies x 2 sources on the 384 x 256 x 256 salt-like model as whole
    solves to tol 1e-6. Sources of one frequency share the model, hence every level's line factorisation: the two
    sources of a frequency are solved TOGETHER (solver.solve_batch; on the levels with long lines one workgroup
    serves its lines for both right-hand sides per factor fetch, k_line_stream) -- against one after the other
    on a hierarchy they share (what `parallel.compute(reuse=True)` does). Fields are bit-identical either way.

    Every (mode, frequency) call is made `repeats` times and is taken apart: source vectors on the host, the device
    hierarchy (allocations, model upload, finest-level buffers), the solve (coarse levels, factorisations and graph
    captures happen inside its first cycles: the per-cycle wall times are reported), and what the caching
    allocator held before and after. `ms_per_source` is computed from the FASTEST repeat of every frequency;
    `first_repeat` gives the same sum over the first calls (what a process that solves every pair once pays).

    The buffers of a finished call stay with torch's caching allocator, as they do in `parallel.compute` (round 4
    returned them to the driver with torch.cuda.empty_cache() before every call: after ~50 GB have been handed
    back, one of the next hipMalloc calls takes 1.7 - 2 s on this stack -- reproduced with torch.empty alone,
    tools/config5_stall.py malloc, profiles/r05_config5_stall.txt; that was the 2.9 / 2.5 s of two of the four
    batched pairs in BENCH_r04.json, not the kernels)."""
    import torch
    import emg3d_amd as emg3d
    from emg3d_amd import solver, models
    wls = [workload(name, source_index=i) for i in range(8)]         # pair index = 2 x frequency index + source
    grid = emg3d.TensorMesh(wls[0]['h'], wls[0]['origin'])
    model = emg3d.Model(grid, **wls[0]['res'])
    opts = {k: v for k, v in wls[0]['opts'].items() if k != 'sslsolver'}
    opts.update(tol=1e-6, verb=0)
    out = {'workload': f'config 5: 4 frequencies x 2 sources, {grid.shape_cells} salt-like model, whole solves to '
           'tol 1e-6 (setup of the hierarchy of each frequency included)', 'repeats': repeats}
    gb = 1.0 / 2 ** 30

    def one_call(pair, together):
        torch.cuda.synchronize(device)
        rec = {'reserved_gb_before': torch.cuda.memory_reserved(device) * gb}
        t0 = time.perf_counter()
        sfs = [emg3d.get_source_field(grid, w['source'], w['frequency']) for w in pair]
        t1 = time.perf_counter()
        hier = solver.Hierarchy(models.VolumeModel(model, sfs[0]), batch=2 if together else 1)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        if together:
            res = emg3d.solve_batch(model, sfs, keep_fields=False, hierarchy=hier, **opts)
        else:
            res = [emg3d.solve(model, sf, sslsolver=False, return_info=True, hierarchy=hier, _download=False, **opts)
                   for sf in sfs]
        torch.cuda.synchronize(device)
        t3 = time.perf_counter()
        rec['reserved_gb_after'] = torch.cuda.memory_reserved(device) * gb
        rec['allocated_gb_after'] = torch.cuda.memory_allocated(device) * gb
        del hier
        infos = [info for _, info in res]
        # wall time of the cycles of every solve call (cycle 1 .. 3 hold level set-up, factorisations, graph captures)
        cyc = [np.diff(np.asarray(info['runtime_at_cycle'], dtype=float)) for info in infos[:1 if together else 2]]
        rec.update(seconds=t3 - t0, source_vectors_s=t1 - t0, hierarchy_s=t2 - t1, solve_s=t3 - t2,
                   cycles=[int(info['it_mg']) for info in infos],
                   cycle_wall_ms=[[round(1e3 * float(x), 1) for x in c] for c in cyc],
                   work=float(sum(info['smoother_cell_sweeps'] for info in infos)))
        return rec

    for tag, together in (('one_by_one', False), ('two_sources_of_a_frequency_together', True)):
        per_freq, best, first, its, work = [], 0.0, 0.0, [], 0.0
        for fi in range(4):
            pair = wls[2 * fi:2 * fi + 2]
            calls = [one_call(pair, together) for _ in range(repeats)]
            secs = [c['seconds'] for c in calls]
            best += min(secs)
            first += secs[0]
            its += calls[0]['cycles']
            work += calls[0].pop('work')
            for c in calls[1:]:
                c.pop('work')
            per_freq.append({'frequency': pair[0]['frequency'], 'seconds': min(secs), 'calls': calls})
        out[tag] = {'seconds': best, 'ms_per_source': best / 8 * 1e3, 'Mcell_sweeps_per_s': work / best / 1e6,
                    'cycles': its, 'first_repeat': {'seconds': first, 'ms_per_source': first / 8 * 1e3},
                    'per_frequency': per_freq}
    return out


# Cycles to tol 1e-6 of config 5's pairs by frequency (0.25 / 0.5 / 1 / 2 Hz: 9 / 7 / 5 / 4, measured by
# `survey_config5`, profiles/r04_bench_triaxial256.json): the cost estimates a multi-rank run hands to parallel.shard (longest-processing-time first)
PAIR_COSTS = {'salt384': [c for c in (9, 7, 5, 4) for _ in range(2)]}

REDUCED_COPY = {'triaxial256': 'triaxial64', 'marine128': 'marine64', 'salt384': 'salt96'}


def time_to_tol(name, wl, b, tol=1e-6):
    """Time to solution next to the cell-sweep rate. The GPU sweeps lines in four colours, the
    reference sequentially (emg3d/core.py:602-624): same converged field, but on models like config
    3's the coloured order needs more cycles -- invisible in a metric that counts cell-sweeps. This
    block reports the whole solve of the bench workload to `tol` on the GPU (cycles, seconds on the
    warm hierarchy) and, on the reduced copy of the workload the oracle can solve in seconds, the
    cycle counts of both orders; ``cycle_ratio`` = lexicographic / four-colour (<= 1) converts the
    headline value into lexicographic-equivalent cell-sweeps (filled in by the CPU leg)."""
    import torch
    import emg3d_amd as emg3d
    out = {'tol': tol, 'workload': name}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, info = emg3d.solve(b.model, b.sfield, sslsolver=False, tol=tol, return_info=True, hierarchy=b.hier,
                          _download=False, **wl['opts'])
    torch.cuda.synchronize()
    out['gpu'] = {'cycles': int(info['it_mg']), 'seconds': time.perf_counter() - t0, 'exit': int(info['exit']),
                  'rel_error': float(info['rel_error']),
                  'ordering': ('four-colour lines, cyclic passes 1,2,3,0,1,...' if wl['opts'].get('linerelaxation')
                               else 'four-colour nodes 0,2,3,1 in every sweep (tiled on large levels)')}
    small = REDUCED_COPY.get(name)
    if small:
        ws = workload(small)
        grid = emg3d.TensorMesh(ws['h'], ws['origin'])
        sf = emg3d.get_source_field(grid, ws['source'], ws['frequency'])
        _, i2 = emg3d.solve(emg3d.Model(grid, **ws['res']), sf, sslsolver=False, tol=tol, return_info=True, **ws['opts'])
        out['reduced_copy'] = {'workload': small, 'gpu_cycles': int(i2['it_mg'])}
    return out


def time_to_tol_cpu(ttt, value, seconds=40.0):
    """The oracle's cycle count in the reference's lexicographic order on the reduced copy (CPU)."""
    from oracle import mg_ref
    rc = ttt.get('reduced_copy')
    if not rc:
        return
    ws = workload(rc['workload'])
    import emg3d_amd as emg3d
    grid = emg3d.TensorMesh(ws['h'], ws['origin'])
    sf = emg3d.get_source_field(grid, ws['source'], ws['frequency'])
    og = mg_ref.Grid(grid.h, grid.origin)
    cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in ws['res'].items()}
    vm = mg_ref.volume_model(og, ws['frequency'], cond['property_x'], cond.get('property_y'), cond.get('property_z'))
    t0 = time.perf_counter()
    _, io = mg_ref.solve(vm, mg_ref.Field(og, sf.field.copy()), tol=ttt['tol'], **ws['opts'])
    rc['oracle_lexicographic_cycles'] = int(io['it_mg'])
    rc['oracle_seconds'] = time.perf_counter() - t0
    ratio = min(1.0, io['it_mg'] / max(rc['gpu_cycles'], 1))
    ttt['cycle_ratio_lexicographic_over_four_colour'] = ratio
    ttt['value_lexicographic_equivalent'] = value * ratio
    ttt['note'] = ('value x cycle_ratio: cell-sweeps per second weighted by what a sweep in the GPU ordering is '
                   'worth in cycles of the reference ordering, measured on the reduced copy')


def csrc_sha16():
    """First 16 hex digits of the SHA-256 over the library's source files (name order): names the build a committed
    measurement belongs to on a box without git."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'emg3d_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload_name, kernel):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from): read from the
    committed PMC summary of the newest round (profiles/rNN_pmc_traffic.json: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes of this very command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950) -- NOT measured in this run, counters cannot be read
    from inside the process; (None, None) if there is no entry for this workload and kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        name = os.path.basename(path)
        try:
            with open(path) as f:
                doc = json.load(f)
            entry = doc[workload_name][kernel]
            src = {'file': 'profiles/' + name, 'measured_in_this_run': False,
                   'how': 'separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command '
                          '(tools/profile_bench.sh), FETCH_SIZE x 2'}
            for k in ('library_commit', 'date', 'passes', 'csrc_sha16'):
                if k in doc.get('_meta', {}):
                    src[k] = doc['_meta'][k]
            # stale: the counters were collected with other library sources than the ones this run uses
            src['stale'] = src.get('csrc_sha16') != csrc_sha16()
            return entry['bytes_per_launch'], src
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def run_gpu(args):
    import torch
    import torch.distributed as dist
    import emg3d_amd as emg3d
    from emg3d_amd import parallel
    gpu_t0 = time.perf_counter()
    # EMG3D_BENCH_BACKEND=gloo: dry run of the multi-rank code path on a box with fewer GPUs
    # than ranks (all ranks share the visible devices, collectives go through host tensors)
    backend = os.environ.get('EMG3D_BENCH_BACKEND', 'nccl')
    rank, world, cdev = parallel.init(backend)               # the product's process-group setup
    if cdev.type == 'cuda':
        device = cdev
    else:
        local = int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)

    # rank 0 builds the model; the others receive it through the product's broadcast (one RCCL
    # broadcast per property array over xGMI, the received arrays stay in HBM and eta / zeta are
    # formed from them on the device: parallel.broadcast_model)
    # which (source, frequency) pair this rank cycles on: its own source; for a workload whose pairs differ in
    # cost (config 5: cycles to tolerance by frequency, PAIR_COSTS) the first pair of the rank's share under the
    # product's longest-processing-time sharding -- with as many ranks as pairs every rank has exactly one
    pair = rank if world > 1 else 0
    if world > 1 and args.workload in PAIR_COSTS:
        costs = PAIR_COSTS[args.workload]
        pair = (parallel.shard(len(costs), rank, world, costs) or [rank % len(costs)])[0]
    wl = workload(args.workload, source_index=pair, with_model=(rank == 0))
    model = None
    if rank == 0:
        model = emg3d.Model(emg3d.TensorMesh(wl['h'], wl['origin']), **wl['res'])
    broadcast_ms = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        model = parallel.broadcast_model(model, 0, cdev)
        torch.cuda.synchronize()
        dist.barrier()
        broadcast_ms = (time.perf_counter() - tb) * 1e3
    b = Bench(wl, model, device)

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
        tmin_ = tt.clone()
        dist.all_reduce(tmin_, op=dist.ReduceOp.MIN)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, dt_min, work_all = tmax[0].item(), tmin_[0].item(), tsum[1].item()
    else:
        dt_max, dt_min, work_all = dt, dt, float(work)

    out = None
    if rank == 0:
        stats = b.kernel_stats()
        n0 = b.grid.n_cells
        names = {0: 'k_gs_point', 1: 'k_gs_line<x>', 2: 'k_gs_line<y>', 3: 'k_gs_line<z>'}
        # the HIP kernel behind each label (csrc/kernels.hip; what rocprofv3 lists)
        from emg3d_amd import _lib
        nx_, ny_, nz_ = b.grid.shape_cells
        tmin = _lib.lib().emg3d_get_option(b'point_tile_min')
        tiled = tmin > 0 and (nx_ - 1) * (ny_ - 1) * (nz_ - 1) >= tmin and (nx_ + 1) * (ny_ + 1) * (nz_ + 1) < 2 ** 31
        hip_names = {0: 'k_gs_point_tile' if tiled else 'k_gs_point'}
        for d_ in (1, 2, 3):
            hip_names[d_] = f'{line_kernel_name(d_, (nx_, ny_, nz_))}<T, DIR={d_ - 1}, ...>'
        dom = max(stats, key=lambda k: stats[k]['ms'])
        bytes_per_launch = BYTES_PER_CELL_SWEEP[b.case] * n0 / 4.0
        ms_launch = stats[dom]['ms'] / stats[dom]['launches']
        achieved = bytes_per_launch / (ms_launch * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(args.workload, names[dom])
        out = {
            'metric': 'Mcells*smoother-iters/s (fp64) per multigrid cycle',
            'value': work_all / dt_max / 1e6,
            'unit': 'Mcell-sweeps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt_max / args.steps * 1e3,
            # the ranks' own clocks around the same timed region (scalars, for whoever computes the scaling curve):
            # slowest (= ms_per_step) and fastest rank, and the model broadcast that precedes the timed region
            'ms_per_step_rank_max': dt_max / args.steps * 1e3, 'ms_per_step_rank_min': dt_min / args.steps * 1e3,
            'broadcast_ms': broadcast_ms, 'backend': backend if world > 1 else None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'c128 (complex fp64)' if b.hier.top.is_complex else 'f64',
            'data': 'synthetic',
            'config': {'workload': args.workload, 'description': wl['label'],
                       'cells': n0, 'cycle': wl['opts']['cycle'],
                       'cell_sweeps_per_step': work / args.steps,
                       'rel_error_after_run': l2,
                       'line_factors': b.hier.line_factors,
                       # single-precision STORAGE of the streamed line passes' T and w records (all arithmetic fp64;
                       # solver.Hierarchy(line_compact=), 'auto' by the model's block condition); where it is on the
                       # finest level cycles in residual form, as the solver itself does
                       'line_factor_storage': 'compact (fp32 T and w records on the streamed levels)'
                       if b.hier.line_compact else 'fp64',
                       'residual_form': bool(getattr(b.var, 'residual_form', False)),
                       'hbm_allocated_gb': torch.cuda.max_memory_allocated() / 1e9,
                       'parallelism': f'{world} independent sources, 1 per GPU', 'pair_of_rank_0': pair,
                       'model_distribution': None if world == 1 else
                       'parallel.broadcast_model from rank 0 (one broadcast per property array, received '
                       f'arrays stay in HBM), backend {backend}', 'broadcast_ms': broadcast_ms},
            'roofline': {
                'bound': 'hbm', 'kernel': names[dom], 'hip_kernel': hip_names[dom],
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                # (scalars next to the nested entries, for parsers that keep scalars only)
                'traffic_ratio': (traffic / bytes_per_launch) if traffic else None,
                'traffic_stale': bool(traffic_source['stale']) if traffic_source else None,
                'bytes_per_launch': bytes_per_launch, 'ms_per_launch': ms_launch,
                'launches_timed': stats[dom]['launches'],
                'gcell_sweeps_per_s': n0 / 4.0 / (ms_launch * 1e-3) / 1e9,
                'all_level0_smoothers': {names[k]: {
                    'ms_per_launch': v['ms'] / v['launches'], 'launches': v['launches'],
                    'GB/s': bytes_per_launch / (v['ms'] / v['launches'] * 1e-3) / 1e9}
                    for k, v in stats.items()}},
        }
    if rank == 0 and world == 1 and not args.no_ttt:
        try:
            out['time_to_tol'] = time_to_tol(args.workload, wl, b)
        except Exception as exc:        # informational block: never takes the bench line down
            out['time_to_tol'] = {'error': repr(exc)}
    if rank == 0 and world == 1 and not args.no_ttt and b.hier.line_compact:
        try:       # both storages in one line (the review of round 5): the cycle with fp64 records, same process, same box
            out['fp64_storage'] = fs = cycle_fp64_storage(wl, model, device, max(2, min(args.steps, 10)))
            out['roofline']['frac_fp64_storage'] = fs['frac']
            out['ms_per_step_fp64_storage'] = fs['ms_per_step']
        except Exception as exc:        # informational block: never takes the bench line down
            out['fp64_storage'] = {'error': repr(exc)}
    if rank == 0 and world == 1 and not args.no_256:
        del b            # (its buffers stay with the caching allocator: handing tens of GB back to the driver makes a later
                         #  hipMalloc take 1.7 - 2 s on this stack, profiles/r05_config5_stall.txt)
        out['smoothers_256'] = sm = smoothers_256(device)
        # the north-star kernel (gauss_seidel at 256^3) as a second roofline entry, per launch
        # (with the coefficient storage the solver uses by default on this kind of model: single-precision eta sums,
        #  all arithmetic fp64 -- solver.Hierarchy(line_compact='auto'); the fp64-stored figure next to it)
        pt = sm['smoothers']['gauss_seidel (k_gs_point_tile), compact records']
        pt64 = sm['smoothers']['gauss_seidel (k_gs_point_tile)']
        out['roofline']['north_star_kernel'] = {
            'kernel': 'k_gs_point_tile (core.gauss_seidel, 256^3 tri-axial)', 'bound': 'hbm',
            'achieved': pt['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': pt['frac'],
            'frac_delivered': pt['frac_delivered'], 'ms_per_launch': pt['ms_per_launch'],
            'traffic': pmc_traffic('smoothers_256', 'k_gs_point_tile')[0],
            'traffic_source': pmc_traffic('smoothers_256', 'k_gs_point_tile')[1]}
        # ... and as scalars of the roofline object itself: per launch, per delivered sweep
        out['roofline']['north_star_frac'] = pt['frac']
        out['roofline']['north_star_frac_fp64_sums'] = pt64['frac']
        out['roofline']['north_star_storage'] = 'eta sums in single precision (fp64 arithmetic)'
        out['roofline']['north_star_ms_per_launch'] = pt['ms_per_launch']
        out['roofline']['north_star_ms_per_sweep'] = pt['ms_per_delivered_sweep']
        out['roofline']['north_star_traffic_ratio'] = (
            out['roofline']['north_star_kernel']['traffic'] / (pt['achieved'] * 1e9 * pt['ms_per_launch'] * 1e-3)
            if out['roofline']['north_star_kernel']['traffic'] else None)
        for lab, v in sm['smoothers'].items():
            if lab.startswith('gauss_seidel_'):
                out['roofline'][f"line_{lab[13]}_256_frac{'_compact' if lab.endswith('compact records') else '_fp64'}"] = v['frac']
    if rank == 0 and world == 1 and not args.no_survey:
        try:
            out['survey_8_sources'] = survey_8(device)
        except Exception as exc:        # informational block: never takes the bench line down
            out['survey_8_sources'] = {'error': repr(exc)}
    if rank == 0 and world == 1 and not args.no_survey:
        try:
            out['survey_config5'] = survey_config5(device)
        except Exception as exc:        # informational block: never takes the bench line down
            out['survey_config5'] = {'error': repr(exc)}
    if world > 1:
        parallel.finalize()
    if out is not None:
        # wall time of everything this process did on the GPU (setup, warm-up, timed region, the
        # informational blocks), for reconciliation with an outside sampler of GPU activity: the
        # timed region is `steps` x `ms_per_step` of it
        torch.cuda.synchronize()
        out['gpu_seconds_total'] = time.perf_counter() - gpu_t0
    return out, wl, rank, world


# -------------------------------------------------------------------------- CPU side ---
def _host_cores():
    """(threads to use, description): the cores this process may run on, capped by the
    container's CPU quota (cgroup v2 cpu.max) -- more threads than the quota are throttled."""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, per = f.read().split()
            if q != 'max':
                quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    model = 'unknown CPU'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    use = min(ncpu, quota) if quota else ncpu
    return use, f"{model}; {ncpu} hardware threads visible" + (f", cgroup CPU quota {quota}" if quota else "")


def _oracle_level(n, case, frequency=1.0, stretch=1.03):
    """A random n^3 level (model, source, field) in the oracle's host types."""
    from oracle import mg_ref
    rng = np.random.default_rng(1)
    h = [widths(n // 2, n // 4, 25., stretch)] * 3
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sx = 10 ** rng.uniform(-1.5, 0.5, (n, n, n))
    vm = mg_ref.volume_model(grid, frequency, sx, sx / 1.5 if case == 'triaxial' else None,
                             sx / 2.5 if case in ('VTI', 'triaxial') else None)
    return grid, vm


def run_cpu_baseline(wl, seconds=25.0):
    """The reference's execution model on this host's cores, with the oracle (oracle/: C
    restatement of the reference's sequential numba kernels, lexicographic order; kind "port"):

    * ``value``: ONE thread -- the reference's kernels are serial (numba nogil, no prange,
      emg3d/core.py:43) -- on a bounded sample of the same workload: the level-0 pre-smoothing
      call of its first cycle (nu_pre = 2 sweeps of each line direction of lr_dir 4, or of the
      point smoother) on the full-size grid, ~10-30 s;
    * ``all_cores``: throughput mode, one independent smoothing job per core at the same time
      (what the reference's process pool does with one solve per worker,
      emg3d/_multiprocessing.py:49-56), each on its own 128^3 (or smaller, memory permitting)
      level of the same anisotropy; aggregate cell-sweeps per second.

    Compiler flags: the faster of the strict parity build (-O2 -fno-fast-math) and a timing
    build (-O3 -march=native -ffast-math ~ numba's fastmath=True at LLVM O3), compiled here on
    the bench host, is used and named."""
    import ctypes
    import threading
    from oracle import core as ocore
    from oracle import mg_ref
    opts = wl['opts']
    lr = 4 if opts.get('linerelaxation') else 0
    fns = {0: ('gauss_seidel',), 4: ('gauss_seidel_y', 'gauss_seidel_z')}[lr]
    nu = 2

    def sweeps(lib, grid, vm, e, s, which=fns):
        ocore._lib = lib
        for fn in which:
            getattr(ocore, fn)(e.fx, e.fy, e.fz, s.fx, s.fy, s.fz, vm.eta_x, vm.eta_y, vm.eta_z,
                               vm.zeta, *grid.h, nu)

    # the two builds, calibrated on a 48^3 level
    libs = {'-O2 -fno-fast-math (strict parity build)': ocore.lib()}
    try:
        import subprocess
        bdir = os.path.join(ROOT, 'oracle', '_build')
        os.makedirs(bdir, exist_ok=True)
        fast = os.path.join(bdir, 'liboracle_fast.so')
        subprocess.check_call(['gcc', '-O3', '-march=native', '-ffast-math', '-std=c99', '-fPIC', '-shared',
                               '-o', fast, os.path.join(ROOT, 'oracle', 'core_oracle.c'), '-lm'],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        libs['-O3 -march=native -ffast-math (timing build)'] = ctypes.CDLL(fast)
    except Exception:       # no gcc on the host: the strict build alone
        pass
    g0, vm0 = _oracle_level(48, wl['case'])
    e0, s0 = mg_ref.Field(g0), mg_ref.Field(g0)
    s0.field[:] = 1.0
    rate = {}
    for tag, lib in libs.items():
        sweeps(lib, g0, vm0, e0, s0)
        t0 = time.perf_counter()
        sweeps(lib, g0, vm0, e0, s0)
        rate[tag] = len(fns) * nu * g0.n_cells / (time.perf_counter() - t0)
    build = max(rate, key=rate.get)
    lib = libs[build]

    # (i) one thread, the workload's own level 0
    grid = mg_ref.Grid(wl['h'], wl['origin'])
    cond = {k: 1.0 / np.asarray(v, dtype=float) for k, v in wl['res'].items()}
    vm = mg_ref.volume_model(grid, wl['frequency'], cond['property_x'], cond.get('property_y'),
                             cond.get('property_z'))
    import emg3d_amd as emg3d
    sf = emg3d.get_source_field(emg3d.TensorMesh(wl['h'], wl['origin']), wl['source'], wl['frequency'])
    s, e = mg_ref.Field(grid, sf.field.copy()), mg_ref.Field(grid)
    est = len(fns) * nu * grid.n_cells / rate[build]
    which = fns if est <= 1.6 * seconds else fns[:1]          # bound the sample
    t0 = time.perf_counter()
    sweeps(lib, grid, vm, e, s, which)
    dt = time.perf_counter() - t0
    work = len(which) * nu * grid.n_cells
    out = {'value': work / dt / 1e6, 'unit': 'Mcell-sweeps/s', 'cores': 1, 'kind': 'port', 'seconds': dt,
           'build': build, 'builds_calibrated_Mcs': {k: v / 1e6 for k, v in rate.items()},
           'sample': f"level-0 pre-smoothing call of the first cycle of the same workload: nu = {nu} sweeps of "
                     f"{' + '.join(which)} on the {grid.shape_cells[0]}x{grid.shape_cells[1]}x{grid.shape_cells[2]} "
                     f"grid ({work:.3g} cell-sweeps), oracle/ C kernels in the reference's sequential order, 1 thread"}
    del vm, s, e, cond

    # (ii) all cores: one job per core, each on its own level
    ncores, desc = _host_cores()
    try:
        with open('/proc/meminfo') as f:
            avail = [int(x.split()[1]) * 1024 for x in f if x.startswith('MemAvailable')][0]
    except (OSError, IndexError):
        avail = 16 << 30
    n = 128
    while n > 32 and ncores * 2 * 3 * 16 * (n + 1) ** 3 > 0.4 * avail:      # e and s per job
        n //= 2
    gj, vmj = _oracle_level(n, wl['case'])
    jobs = [(mg_ref.Field(gj), mg_ref.Field(gj)) for _ in range(ncores)]
    for ej, sj in jobs:
        sj.field[:] = 1.0
    reps = max(1, int(round(4.0 / (len(fns) * nu * gj.n_cells / rate[build]))))

    def job(ej, sj):
        for _ in range(reps):
            sweeps(lib, gj, vmj, ej, sj)      # ctypes releases the GIL inside the C kernels
    threads = [threading.Thread(target=job, args=j) for j in jobs]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dta = time.perf_counter() - t0
    worka = ncores * reps * len(fns) * nu * gj.n_cells
    out['all_cores'] = {'value': worka / dta / 1e6, 'unit': 'Mcell-sweeps/s', 'cores': ncores, 'host': desc,
                        'seconds': dta, 'kind': 'port',
                        'sample': f"{ncores} concurrent jobs (one per core), each {reps} x nu = {nu} sweeps of "
                                  f"{' + '.join(fns)} on its own {n}^3 {wl['case']} level"}
    ocore._lib = libs['-O2 -fno-fast-math (strict parity build)']
    return out


def _spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-run this command under
    torch.distributed.run with N ranks on this node (rendezvous on 127.0.0.1) and pass its
    exit status on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='triaxial256',
                    help="default: BASELINE.json config 3 (256^3 tri-axial, W-cycle + sc + lr), the largest "
                         "single-GPU configuration, the size the metric's 40 %% target is stated at")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--opt', action='append', default=[],
                    help='library tuning option name=value (emg3d_set_option), for experiments')
    ap.add_argument('--no-survey', action='store_true', help='skip the 8-source survey block')
    ap.add_argument('--no-ttt', action='store_true', help="skip the time-to-tolerance block")
    ap.add_argument('--no-256', action='store_true',
                    help="skip the separate 256^3 smoother measurement ('smoothers_256')")
    ap.add_argument('--only-config5', action='store_true',
                    help="only the informational block 'survey_config5' (config 5 on one GPU, taken apart per call)")
    ap.add_argument('--line-factors', default=None, choices=['resident', 'rebuild', 'single'],
                    help="factor-memory policy of the hierarchy (solver.Hierarchy; default: resident)")
    args = ap.parse_args()
    if args.line_factors:
        os.environ['EMG3D_AMD_LINE_FACTORS'] = args.line_factors
    if 'RANK' not in os.environ and args.gpus > 1:
        raise SystemExit(_spawn_ranks(args))
    if args.opt:
        from emg3d_amd import _lib
        for o in args.opt:
            k, v = o.split('=')
            if _lib.lib().emg3d_set_option(k.encode(), int(v)) != 0:
                raise SystemExit(f"unknown option {o}")
    if args.only_config5:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps({'survey_config5': survey_config5(torch.device('cuda', 0))}))
        return
    out, wl, rank, world = run_gpu(args)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = run_cpu_baseline(wl)
            if 'reduced_copy' in out.get('time_to_tol', {}):
                time_to_tol_cpu(out['time_to_tol'], out['value'])
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))


if __name__ == '__main__':
    main()

"""Multigrid solver driver: ``solve`` / ``multigrid`` / ``krylov`` with the reference's
signatures, keyword arguments, defaults, return conventions, ``info_dict`` keys and exit
messages (reference emg3d/solver.py:52-449, 471-649, 1074-1381, 1482-1664), running the
cycle on a device-resident level hierarchy (emg3d_amd/_device.py) with the HIP kernels.

Host side (this file): cycle control flow -- V/W/F recursion, semicoarsening and
line-relaxation cycling, termination rules, logging. It is O(levels) Python per cycle.
Device side: everything that touches a field entry.

Differences from the reference that are intended:
* smoothers use a four-colour ordering (include/emg3d_amd.h); converged fields agree
  with the reference to the solver tolerance, per-cycle error histories differ slightly;
* the residual norm on coarse levels (emg3d/solver.py:530) is only evaluated when its
  value is used (level 0, or ``verb > 4``);
* coarse grids / models / weights are built once per (level, sc_dir) and reused.
"""
import os

import numpy as np
import torch

from emg3d_amd import _cycle, _lib, fields, models
from emg3d_amd._cycle import ConvergenceError as _ConvergenceError
from emg3d_amd._cycle import current_lr_dir as _current_lr_dir
from emg3d_amd._cycle import current_sc_dir as _current_sc_dir
from emg3d_amd._cycle import one_liner as _print_one_liner
from emg3d_amd._cycle import record_cycle as _print_cycle_info
from emg3d_amd._cycle import smooth_level as _smooth
from emg3d_amd._cycle import terminate as _terminate
from emg3d_amd._device import DeviceLevel
from emg3d_amd._params import MGParameters, Timer      # noqa: F401  (Timer: reference name utils.Timer)

__all__ = ['solve', 'solve_batch', 'solve_source', 'multigrid', 'krylov', 'smoothing', 'restriction',
           'prolongation', 'residual', 'MGParameters', 'RegularGridProlongator']


def __dir__():
    return __all__


# ------------------------------------------------------------------------------ solve ---
# keywords of `solve` that are not solver settings: name -> default
_SOLVE_EXTRAS = {'always_return': False, 'plain': False, 'efield': None, 'hierarchy': None,
                 '_download': True,           # False: the result stays in hierarchy.top.e only
                 '_sparse_source': False,     # the source goes up as its few non-zeros
                 'smoother_omega': 1.0,       # != 1: extrapolated smoothing calls (_cycle.smooth_level)
                 'residual_form': 'auto',     # finest level in residual form (_cycle.run_cycles)
                 'line_compact': None}        # Hierarchy(line_compact=): compact records of the streamed line passes


def _residual_form(choice, var, model, sfield):
    """Does multigrid as a SOLVER run its finest level in residual form (``_cycle.run_cycles``)?
    True / False, or 'auto': yes where the accuracy the line smoothers' stored block inverses can
    reach on the model -- eps / (|s| mu0 sigma_min h_min^2) x the conductivity contrast sigma_max /
    sigma_min: the blocks' condition in the most resistive cells, and how much larger the fields (whose
    scale the block solves' errors have) are there than the source's norm suggests; measured floors of the
    direct form, relative to the source: 1e-9 on blocky tri-axial models with a contrast of 100-250 where
    the first factor alone says 1e-11 (tools/soak_same_order.py) -- is not well below the tolerance asked
    for. (As a Krylov preconditioner multigrid is in residual form anyway; a stall that the rule does not
    foresee makes _cycle.run_cycles switch by itself.)"""
    if choice in (True, False):
        return bool(choice)
    if choice == 'on-stall':          # direct form until (unless) it stalls above the tolerance
        return False
    if choice != 'auto':
        raise ValueError(f"`residual_form` must be True, False, 'auto' or 'on-stall'. Provided: {choice!r}.")
    if var.sslsolver or not var.cycle or sfield.sval is None:
        return False
    # Tight tolerances: always. Relative to the norm of a dipole source the direct form's floor lies between
    # 1e-10 and 1e-8 on most models (soaks of round 3: at tol 1e-9 one solve in ten hovers around its floor for
    # several cycles, or stalls, where the oracle in the same ordering -- or the residual form -- converges
    # straight through); three copies and an update per cycle (2 %) buy the reference's cycle counts.
    if var.tol < 1e-7:
        return True
    # (cheap on purpose: two reductions per property array -- the smallest conductivity of the model
    # with the smallest cell width, whether or not they meet in one cell)
    hmin = min(float(np.min(h)) for h in model.grid.h)
    # (not cached: models are edited in place -- `model.property_x[:, :, -1] = 1e8` adds the air
    # layer this rule exists for -- and two reductions per array cost 5 ms at 128^3)
    sig, sig_max = np.inf, 0.0
    with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
        for name in ('property_x', 'property_y', 'property_z'):
            prop = getattr(model, name)
            if prop is not None:
                ends = models._MAPS[model.mapping](np.array([np.min(prop), np.max(prop)], dtype=float))
                sig, sig_max = min(sig, float(np.min(ends))), max(sig_max, float(np.max(ends)))
        cond = np.float64(1.0) / np.float64(abs(complex(sfield.sval)) * fields.MU_0 * sig * hmin ** 2)
        cond = cond * np.float64(sig_max) / np.float64(sig)
    return bool(np.isfinite(cond) and np.finfo(float).eps * cond > 0.01 * var.tol)


def _check_omega(omega):
    omega = float(omega)
    if not 0.0 < omega < 2.0:
        raise ValueError(f"`smoother_omega` must lie in (0, 2). Provided: {omega}.")
    return omega


def solve(model, sfield, sslsolver=True, semicoarsening=True, linerelaxation=True, verb=0,
          **kwargs):
    """Solver for three-dimensional electromagnetic diffusion on one MI355X.

    Same call as the reference's ``emg3d.solve`` (emg3d/solver.py:52-449): multigrid
    (cycle 'F', 'V', 'W') as solver or as preconditioner of a Krylov method, with
    semicoarsening and line relaxation.

    Parameters: ``model`` (Model), ``sfield`` (Field), ``sslsolver`` {True, False,
    'bicgstab', 'cgs', 'gcrotmk'}, ``semicoarsening`` / ``linerelaxation`` {bool, int},
    ``verb`` int, and by keyword ``cycle='F'``, ``efield=None``, ``tol=1e-6``,
    ``maxit=50``, ``nu_init=0``, ``nu_pre=2``, ``nu_coarse=1``, ``nu_post=2``,
    ``clevel=-1``, ``return_info=False``, ``log=1``, ``plain=False``.

    Not in the reference: ``line_compact=None`` (True / False / 'auto', see ``Hierarchy``: the streamed line
    passes keep their factor records in single precision where the model allows it -- same converged field and
    cycle counts, per-cycle values differ by eps32 x cond of the blocks); ``hierarchy=`` (a ``Hierarchy`` built for
    the same model, grid and frequency) reuses the device-resident levels, line factorisations and captured graphs of
    an earlier solve -- what several sources at one frequency share; ``smoother_omega=1.0``: a
    value != 1 extrapolates every smoothing call, e <- e_before + omega (e_after - e_before) --
    same solution, and on models where the four-colour ordering costs cycles against the
    reference's sequential sweeps, fewer of them (1.2-1.3: 0-17 % in DESIGN.md 4.1; too large a
    value diverges); ``residual_form='auto'`` (True / False): multigrid as a solver runs every cycle
    on the residual equation A d = s - A e from d = 0 and adds d to the field -- the same iteration
    in exact arithmetic, but the rounding errors of the smoothers then scale with the residual
    instead of the field, so that the iteration converges to round-off where the stored block
    inverses of the line smoothers would otherwise stall it (air layers, very low frequencies:
    DESIGN.md 4.3); 'auto' switches it on where model and tolerance call for it, and in the middle of
    a solve whose direct-form cycles stagnate above the tolerance (the cycling then continues on the
    residual equation instead of returning STAGNATED); 'on-stall' does only the latter.

    Returns ``efield`` (if none was provided) and/or ``info_dict`` (if ``return_info``; the reference's
    keys plus ``smoother_cell_sweeps`` and ``residual_form``: False, True or 'switched').
    """
    extra = {name: kwargs.pop(name, default) for name, default in _SOLVE_EXTRAS.items()}
    if extra['plain']:       # plain multigrid: whatever was left at its default is switched off
        sslsolver, semicoarsening, linerelaxation = (
            False if flag is True else flag for flag in (sslsolver, semicoarsening, linerelaxation))
    var = MGParameters(verb, sslsolver, semicoarsening, linerelaxation, model.shape, **kwargs)
    var.cprint(f"\n:: emg3d START :: {var.time.now} :: emg3d_amd (MI355X)\n", 2)
    var.cprint(var, 2)

    if sfield.frequency is None:
        raise ValueError(
            "Source field is missing frequency information; Create "
            "it with `emg3d.fields.get_source_field`, or initiate it "
            "with `emg3d.fields.Field`, providing frequency information.")
    var.smoother_omega = _check_omega(extra['smoother_omega'])
    var.residual_form = _residual_form(extra['residual_form'], var, model, sfield)
    var.residual_form_auto = isinstance(extra['residual_form'], str)      # 'auto': may still switch on a stall
    # the source as its few non-zero entries: when the caller vouches for an unmodified field (parallel.solve), or when
    # the field's dense buffer has never been handed out (fields.Field._untouched: a source straight from
    # get_source_field) -- its norm and its upload are then a dozen numbers instead of two passes over 1.2 GB
    var.sparse_source = ((bool(extra['_sparse_source']) or getattr(sfield, '_untouched', False))
                         and getattr(sfield, '_sparse', None) is not None)
    if var.sparse_source and not extra['_sparse_source']:
        sfield._assemble_on_device = False
    var.download = bool(extra['_download'])
    var.l2_refe = _host_norm(sfield._sparse[1] if var.sparse_source else sfield.field)
    var.error_at_cycle[0] = var.l2_refe

    vmodel = models.VolumeModel(model, sfield)
    efield, note = _start_field(model, vmodel, sfield, extra['efield'], extra['always_return'], var)
    if var.l2_refe < 100 * np.finfo(float).tiny:         # zero source: zero field
        var.l2_refe = np.nan
        note = _nothing_to_do(var, "   > RETURN ZERO E-FIELD (provided sfield is zero)\n")
        efield = fields.Field(model.grid, dtype=sfield.dtype, frequency=sfield._frequency)

    _log_table_head(var)
    if extra['hierarchy'] is not None:
        extra['hierarchy'].check(vmodel)
        if not var.sslsolver and not var.cycle:
            # nothing to iterate (zero source / start field already good enough): the hierarchy's
            # field must still be THIS solve's result -- receivers and the gradient read it from
            # HBM, where the previous pair's field would otherwise linger
            extra['hierarchy'].upload_field(efield)
    var.line_compact = extra['line_compact']
    if var.sslsolver:
        krylov(vmodel, sfield, efield, var, hierarchy=extra['hierarchy'])
    elif var.cycle:
        multigrid(vmodel, sfield, efield, var, hierarchy=extra['hierarchy'])
    _log_summary(var, note)

    out = []
    if var.do_return:
        out.append(efield)
    if var.return_info:
        out.append(_info_dict(var))
    return out[0] if len(out) == 1 else (tuple(out) if out else None)


def _nothing_to_do(var, note):
    var.sslsolver = var.cycle = None
    var.exit_message = "CONVERGED"
    return note


def _start_field(model, vmodel, sfield, efield, always_return, var):
    """The field the iteration starts from (emg3d/solver.py:325-370): a new zero field, returned
    at the end; or the caller's, updated in place after its PEC faces were zeroed -- and left
    alone if it already satisfies the tolerance. Returns (efield, note for the log)."""
    if efield is None:
        efield = fields.Field(model.grid, dtype=sfield.dtype, frequency=sfield._frequency)
        efield._is_zero = True
        var.do_return = True
        return efield, ""
    if sfield.dtype != efield.dtype:
        raise ValueError(
            "Source field and electric field must have the same "
            "dtype; complex (f-domain) or real (s-domain). Provided:"
            f"sfield: {sfield.dtype}; efield: {efield.dtype}.")
    if efield.frequency is None:
        efield._frequency = sfield._frequency
    # tangential components on the six boundary faces (PEC)
    for comp, tangential_to in ((efield.fx, (1, 2)), (efield.fy, (0, 2)), (efield.fz, (0, 1))):
        for axis in tangential_to:
            face = [slice(None)] * 3
            face[axis] = [0, -1]
            comp[tuple(face)] = 0.
    var.do_return = always_return
    var.l2 = residual(vmodel, sfield, efield, True)
    if var.l2 < var.tol * var.l2_refe:
        return efield, _nothing_to_do(var, "   > NOTHING DONE (provided efield already good enough)\n")
    return efield, ""


def _log_table_head(var):
    head = f"   [hh:mm:ss]  {'rel. error':<22}"
    if var.sslsolver:
        head += f"{'solver':<20}" + (f"{'MG':<11} l s" if var.cycle else "")
    elif var.cycle:
        head += f"{'[abs. error, last/prev]':>29}   l s"
    else:
        return
    var.cprint(head + "\n", 3)


def _log_summary(var, note):
    failed = var.exit_message != 'CONVERGED'
    if var.verb in (1, 2):
        _print_one_liner(var, var.l2, True)
    elif var.verb > 2:
        if var.sslsolver:
            note = f"   > Solver steps     : {var.ssl_it}\n"
            if var.cycle:
                note += f"   > MG prec. steps   : {var.it}\n"
        elif var.cycle:
            note = f"   > MG cycles        : {var.it}\n"
        note += (f"   > Final rel. error : {var.l2/var.l2_refe:.3e}\n\n"
                 f":: emg3d END   :: {var.time.now} :: runtime = {var.time.runtime}\n")
        var.cprint(note, 2)
    elif var.verb == 0 and failed:
        var.cprint(f"* WARNING :: {var.exit_message}", -1)


# keys of the info dict of `solve` (emg3d/solver.py:416-432) -> where the value comes from
_INFO = (('exit', lambda v: int(v.exit_message != 'CONVERGED')), ('exit_message', lambda v: v.exit_message),
         ('abs_error', lambda v: v.l2), ('rel_error', lambda v: v.l2 / v.l2_refe),
         ('ref_error', lambda v: v.l2_refe), ('tol', lambda v: v.tol), ('it_mg', lambda v: v.it),
         ('it_ssl', lambda v: v.ssl_it), ('time', lambda v: v.runtime_at_cycle[-1]),
         ('runtime_at_cycle', lambda v: v.runtime_at_cycle), ('error_at_cycle', lambda v: v.error_at_cycle),
         ('log', lambda v: v.log_message),
         # addition of this package (not in the reference):
         ('smoother_cell_sweeps', lambda v: v.smoother_cell_sweeps),
         # the finest level ran (True), or ended up running ('switched'), on the residual equation
         ('residual_form', lambda v: 'switched' if getattr(v, 'residual_form_switched', False)
          else bool(getattr(v, 'residual_form', False))),
         # the hierarchy kept compact (single-precision) coefficient records on its large levels (Hierarchy.line_compact)
         ('line_compact', lambda v: bool(getattr(v, 'hierarchy_compact', False))))


def _info_dict(var):
    return {key: get(var) for key, get in _INFO}


def solve_batch(model, sfields, semicoarsening=True, linerelaxation=True, verb=0, **kwargs):
    """Several sources of ONE frequency on one model, solved together: by multigrid
    (``sslsolver=False``, the default here) or by BiCGSTAB with multigrid as preconditioner
    (``sslsolver=True`` / ``'bicgstab'``: every source runs its own Krylov iteration, the
    preconditioner and operator applications are shared, see ``_bicgstab_batch``).

    Not in the reference (which solves every source-frequency pair separately,
    emg3d/simulations.py:1453-1464): the sources share the model, hence the coarse models, the
    line factorisations and every kernel launch -- the smoothers, residuals and transfers take
    the right-hand sides as one more grid dimension (``emg3d_level::batch``). The launches of
    the coarse levels cost what they cost for one source, and the factors are read once.
    Every source gets exactly the iterates, cycle count and field of its own
    ``solve(model, sfield, sslsolver=False, ...)``: a source that meets the tolerance is copied
    out at that cycle while the others carry on.

    Returns a list of ``(efield, info_dict)``. Keyword arguments as ``solve`` (multigrid only),
    plus ``receivers`` (one ``(x, y, z, azimuth, elevation)`` for all sources or a list with
    one per source: ``info['responses']``, interpolated on the device), ``receiver_method`` and
    ``keep_fields`` (False: no field download, ``efield`` is None), ``hierarchy`` (a
    ``Hierarchy(vmodel, batch=len(sfields))`` of an earlier batch of the same frequency).
    """
    sslsolver = kwargs.pop('sslsolver', False)
    if kwargs.pop('plain', False):       # as in solve(): only what was left at True is switched off
        sslsolver, semicoarsening, linerelaxation = (
            False if flag is True else flag for flag in (sslsolver, semicoarsening, linerelaxation))
    for k in ('efield', 'return_info', 'always_return'):
        kwargs.pop(k, None)
    receivers = kwargs.pop('receivers', None)            # one tuple for all, or one per source
    receiver_method = kwargs.pop('receiver_method', 'cubic')
    keep_fields = kwargs.pop('keep_fields', True)
    hierarchy = kwargs.pop('hierarchy', None)            # a Hierarchy(vmodel, batch=len(sfields)) to reuse
    omega = _check_omega(kwargs.pop('smoother_omega', 1.0))     # extrapolated smoothing calls, as in solve()
    resform = kwargs.pop('residual_form', 'auto')               # finest level in residual form, as in solve()
    line_compact = kwargs.pop('line_compact', None)             # Hierarchy(line_compact=), as in solve()
    sfields = list(sfields)
    nb = len(sfields)
    if nb == 0:
        return []
    first = sfields[0]
    for sf in sfields:
        if sf.frequency is None:
            raise ValueError("Source field is missing frequency information.")
        if sf.grid != first.grid or sf._frequency != first._frequency or sf.dtype != first.dtype:
            raise ValueError("solve_batch: all sources must share grid and frequency.")
    vmodel = models.VolumeModel(model, first)
    def new_var():
        v = MGParameters(sslsolver=sslsolver, semicoarsening=semicoarsening, linerelaxation=linerelaxation,
                         shape_cells=model.shape, verb=verb, **kwargs)
        v.smoother_omega = omega
        return v
    vars_ = [new_var() for _ in sfields]
    var = svar = new_var()             # carries the structure of the cycle, shared by all sources
    svar.residual_form = _residual_form(resform, svar, model, first)
    svar.residual_form_auto = isinstance(resform, str)
    for v in vars_:
        v.residual_form = svar.residual_form
    if var.sslsolver not in (None, False, 'bicgstab') or (var.sslsolver and not var.cycle):
        raise ValueError("solve_batch: multigrid, or BiCGSTAB with multigrid as preconditioner.")
    def rec_of(b):
        if receivers is None:
            return None
        return receivers[b] if isinstance(receivers, list) else receivers

    if nb == 1 or var.clevel[var.sc_dir] == 0:
        out = []
        for b, sf in enumerate(sfields):
            ef, info = solve(model, sf, sslsolver=sslsolver, semicoarsening=semicoarsening,
                             linerelaxation=linerelaxation, verb=verb, return_info=True, always_return=True,
                             smoother_omega=omega, residual_form=resform, line_compact=line_compact, **kwargs)
            if rec_of(b) is not None:
                info['responses'] = fields.get_receiver(ef, rec_of(b), receiver_method)
            out.append((ef if keep_fields else None, info))
        return out
    if hierarchy is not None and hierarchy.top.batch == nb:
        hierarchy.check(vmodel)
        hier = hierarchy
    else:
        hier = Hierarchy(vmodel, batch=nb, line_compact=line_compact,
                         point_compact_top=bool(var.sslsolver) or bool(getattr(svar, 'residual_form', False)))
    top = hier.top
    for v in vars_:
        v.hierarchy_compact = hier.line_compact
    n = top.grid.n_edges
    efields = []
    for b, (sf, v) in enumerate(zip(sfields, vars_)):
        sparse = getattr(sf, '_sparse', None) is not None and (bool(getattr(sf, '_trust_sparse', False)) or
                                                               getattr(sf, '_untouched', False))
        if sparse and not getattr(sf, '_trust_sparse', False):
            sf._assemble_on_device = False
        v.l2_refe = _host_norm(sf._sparse[1] if sparse else sf.field)
        v.error_at_cycle[0] = v.l2_refe
        hier.put_source(sf, top.s[b * n:(b + 1) * n], sparse)
        efields.append(fields.Field(model.grid, dtype=sf.dtype, frequency=sf._frequency))
    top.zero_field()
    nonzero = [v.l2_refe >= 100 * np.finfo(float).tiny for v in vars_]
    if var.sslsolver:
        done = _bicgstab_batch(hier, svar, vars_, nonzero)
    else:
        done, _ = _multigrid_batch(top, svar, vars_, nonzero)
    out = []
    for b, (ef, v) in enumerate(zip(efields, vars_)):
        zero = v.l2_refe < 100 * np.finfo(float).tiny     # zero source: zero field (solver.py:372-379)
        if zero:
            v.exit_message = "CONVERGED"
            v.l2, v.l2_refe = 0.0, np.nan            # as solve() reports a zero source
        elif keep_fields and done[b] is not None:
            torch.from_numpy(ef.field).copy_(done[b])
        info = _info_dict(v)
        if rec_of(b) is not None:     # from the solution while it is in HBM
            info['responses'] = fields.get_receiver(ef, rec_of(b), receiver_method,
                                                    device_field=None if zero or done[b] is None else done[b])
        out.append((ef if keep_fields else None, info))
    return out


def _multigrid_batch(lv, svar, vars_, active=None):
    """Level 0 of ``_multigrid`` for ``lv.batch`` right-hand sides in lock step: the structure
    of the cycle (sc/lr cycling, cycmax; carried by ``svar``) does not depend on the data, so one
    recursion serves all; norms, stagnation buffers, counters and termination are per source
    (``vars_``). Returns (done, failed): the solution of every source as it was when that source
    terminated (device tensors; None for sources that were not active), and which sources
    ended with a ``_ConvergenceError`` (multigrid as preconditioner: diverged / stagnated)."""
    nb = lv.batch
    n = lv.grid.n_edges
    cycmax = svar.cycmax
    it = 0
    # (finest level in residual form, as _cycle.run_cycles does it for one source)
    resform = bool(getattr(svar, 'residual_form', False)) and not svar.sslsolver
    may_switch = bool(getattr(svar, 'residual_form_auto', False)) and not svar.sslsolver
    if not resform and not svar.sslsolver and lv.uses_line_compact(lines=bool(svar.lr_cycle) or svar.lr_dir != 0):
        # (compact line records on this level: it must see residuals, as in _cycle.run_cycles)
        resform = svar.residual_form = True
        for v in vars_:
            v.residual_form = True
    if resform:
        lv._b_valid = False
    if getattr(lv, '_resmode', None) is not None:      # (left behind by an interrupted solve)
        lv._resmode = None
    l2_last = lv.residual(store=resform, norm=True)
    l2_stag = np.ones((nb, svar.maxcycle)) * l2_last[:, None]
    active = [True] * nb if active is None else list(active)
    done = [None] * nb
    failed = [False] * nb
    base = [v.smoother_cell_sweeps for v in vars_]
    s0 = svar.smoother_cell_sweeps
    if svar.nu_init > 0:
        _smooth(lv, svar.nu_init, svar.lr_dir, svar)
        if resform:
            lv.residual(store=True, norm=False)
    while any(active):
        l2_prev = l2_last.copy()
        l2_stag[:, (it - 1) % svar.maxcycle] = l2_last
        if resform:
            lv.to_residual_equation()
        if svar.nu_pre > 0:
            _smooth(lv, svar.nu_pre, svar.lr_dir, svar)
        sc_dir = _current_sc_dir(svar.sc_dir, lv.grid)
        lv.residual(store=True, norm=False)
        clv = lv.restrict_to(sc_dir)
        _cycle.coarse_correction(clv, svar, cycmax, top=lv)       # eager, or the captured graph of this variant
        lv.prolong_from(sc_dir)
        if svar.nu_post > 0:
            _smooth(lv, svar.nu_post, svar.lr_dir, svar)
        it += 1
        if resform:
            lv.from_residual_equation()
        l2_last = lv.residual(store=resform, norm=True)
        sc_now, lr_now = svar.sc_dir, svar.lr_dir
        if svar.sc_cycle:
            svar.sc_dir = next(svar.sc_cycle)
        if svar.lr_cycle:
            svar.lr_dir = next(svar.lr_cycle)
        switch, stalled = False, []
        for b, v in enumerate(vars_):
            if not active[b]:
                continue
            v.it += 1
            v.smoother_cell_sweeps = base[b] + (svar.smoother_cell_sweeps - s0)
            v.sc_dir, v.lr_dir = sc_now, lr_now           # what the log line of this cycle shows
            _print_cycle_info(v, float(l2_last[b]), float(l2_prev[b]))
            v.sc_dir, v.lr_dir = svar.sc_dir, svar.lr_dir
            stag = float(l2_stag[b, (it - 1) % svar.maxcycle])
            if may_switch and not resform and it < v.maxit:
                # (as _cycle.run_cycles: a direct form that stalls above the tolerance goes on in residual form --
                # here the whole batch does, from the next cycle on)
                reason = _cycle.stop_reason(v, float(l2_last[b]), stag, it)
                if reason is not None and reason[0] == "STAGNATED" and l2_last[b] < 1e-3 * v.l2_refe:
                    switch = True
                    stalled.append(b)
                    continue
            try:
                finished = _terminate(v, float(l2_last[b]), stag, it)
            except _ConvergenceError:
                finished = failed[b] = True
            if finished:
                active[b] = False
                v.l2 = float(l2_last[b])
                done[b] = lv.solution()[b * n:(b + 1) * n].clone()      # (residual form: the accumulated field)
        if switch and any(active):
            # From the next cycle on the WHOLE batch cycles on the residual equation (one set of launches serves
            # all right-hand sides): after a switch a source's arithmetic depends on its batch-mates -- every
            # source still active records it (info['residual_form'] == 'switched'); until then fields, counts
            # and histories are those of separate solves. Stagnation is judged afresh only for the sources that
            # stalled; the others keep their histories (a source that truly stagnates is still caught on time).
            resform = svar.residual_form = True
            lv._b_valid = False
            lv.residual(store=True, norm=False)
            for b in stalled:
                l2_stag[b, :] = np.inf
            for b, v in enumerate(vars_):
                if active[b]:
                    v.residual_form_switched = True
    lv.leave_residual_form()
    return done, failed


def _bicgstab_batch(hier, svar, vars_, nonzero):
    """BiCGSTAB with multigrid as preconditioner for the right-hand sides of a batch
    (``_krylov.bicgstab_batch``: every source iterates with its own scalars; the preconditioner
    and operator applications are shared launches). Identical to separate solves as long as the
    sources run the same number of cycles inside every preconditioner call (the rule: ``maxit``
    cycles; a source may stop earlier on convergence) -- otherwise the sc/lr cycling of the shared
    structure and of a separate solve drift apart, and the fields agree to the tolerance only.
    Returns the solutions (device tensors; None for zero sources / failed solves)."""
    from emg3d_amd import _krylov
    top = hier.top
    nb, n = top.batch, top.grid.n_edges
    live = list(nonzero)
    failed = [False] * nb
    mover = _krylov.Vectors(top)

    def precondition(SRC, OUT, live_now):
        mover.copy(top.s, SRC)
        top.zero_field()
        done, bad = _multigrid_batch(top, svar, vars_, live_now)
        for b in range(nb):
            if live_now[b]:
                mover.copy(OUT[b * n:(b + 1) * n], done[b])
                if bad[b]:
                    failed[b] = True
                    live_now[b] = False

    X, code = _krylov.bicgstab_batch(hier, svar, vars_, live, precondition,
                                     lambda b, l2: _krylov_callback(vars_[b], l2))
    done = [None] * nb
    for b, v in enumerate(vars_):
        if not nonzero[b]:
            continue
        if failed[b]:
            v.exit_message += " (returned field is zero)"
            continue
        if code[b] < 0:
            v.exit_message = v.exit_message or f"Error in {v.sslsolver} ({code[b]})"
        else:
            v.exit_message = "CONVERGED" if code[b] == 0 else "MAX. ITERATION REACHED, NOT CONVERGED"
        done[b] = X[b * n:(b + 1) * n]
    return done


def _host_norm(x):
    """2-norm of a host array (scipy.linalg.norm in the reference, emg3d/solver.py:312) with
    the BLAS pool capped: OpenBLAS / OpenMP start one spinning thread per visible core (256
    on an MI355X host) for one nrm2 call, and under a container CPU quota the process is
    then throttled for most of a scheduler period -- measured as a ~80 ms stall in a 200 ms
    solve."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                  # pragma: no cover
        return float(np.linalg.norm(x))
    with threadpool_limits(limits=4):
        return float(np.linalg.norm(x))


def solve_source(model, source, frequency, **kwargs):
    """``get_source_field`` + ``solve`` (emg3d/solver.py:452-467)."""
    sfield = fields.get_source_field(model.grid, source, frequency)
    return solve(model, sfield, **kwargs)


# -------------------------------------------------------------------------- multigrid ---
def _device():
    _lib.require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


# largest block condition estimate under which 'auto' stores the streamed line records in single precision:
# eps32 x 3e4 = 2e-3 relative perturbation of a line solve. (tools/compact_cycles.py: cycle counts unchanged up to 2e6
# on a marine model; the soak of round 6, profiles/r06_soak_large.txt: 11 of 12 compact solves with the oracle's cycle
# count, one Laplace-domain case at 8e4 with 7 cycles instead of 6 -- hence the bound below it; the bench workloads are
# at 4e2 / 1.6e4 / 2e4; an air layer of 1e8 Ohm m is 2e10 and keeps fp64 records)
COMPACT_COND_MAX = 3e4


def block_condition(vmodel):
    """Estimate of the condition of the line smoothers' 5 x 5 blocks, 1 / (|s| mu0 sigma_min h_min^2): the ratio of
    the curl-curl part to the conduction part of a block in the most resistive cell if it had the smallest width
    (cheap on purpose, like ``_residual_form``: two reductions per property array)."""
    hmin = min(float(np.min(h)) for h in vmodel.grid.h)
    model = getattr(vmodel, '_model', None)
    with np.errstate(divide='ignore', over='ignore', invalid='ignore'):
        if model is not None:
            sig = np.inf
            for name in ('property_x', 'property_y', 'property_z'):
                prop = getattr(model, name)
                if prop is not None:
                    ends = models._MAPS[model.mapping](np.array([np.min(prop), np.max(prop)], dtype=float))
                    sig = min(sig, float(np.min(ends)))
            smu_sig = abs(complex(vmodel._sval)) * fields.MU_0 * sig
        else:
            # eta / zeta holders (tests, tools): |eta| / V = |s| mu0 sigma
            vol = np.asarray(vmodel.grid.cell_volumes).reshape(vmodel.grid.shape_cells, order='F')
            smu_sig = min(float(np.min(np.abs(np.asarray(e)) / vol)) for e in (vmodel.eta_x, vmodel.eta_y, vmodel.eta_z))
        cond = np.float64(1.0) / (np.float64(smu_sig) * np.float64(hmin) ** 2)
    return float(cond) if np.isfinite(cond) else np.inf


class Hierarchy:
    """Level 0 on the device for one (model, frequency): upload once, cycle many times."""

    def __init__(self, vmodel, device=None, batch=1, line_factors=None, line_compact=None, point_compact_top=False):
        """line_compact: True / False / 'auto' (default; or the environment's EMG3D_AMD_LINE_COMPACT) -- COMPACT line
        records: on the levels whose line passes stream their records through HBM (lines of ~128 blocks and more) the
        inverse blocks of the stored line factorisation and the forward pass's w records are kept in single precision
        (all arithmetic, right-hand sides and solutions stay fp64): 890 instead of 1 210 B per block and colour pass,
        184 instead of 304 B of factor memory per cell and direction. The line solve is then a PERTURBED smoother
        (relative eps32 x cond of the 5 x 5 blocks), so every level must solve a correction equation: the coarse levels
        always do, the finest runs in residual form (``_cycle.run_cycles`` does that by itself on such a hierarchy; as
        a Krylov preconditioner it is in that form anyway). 'auto': where the block condition estimate
        1 / (|s| mu0 sigma_min h_min^2) is at most ``COMPACT_COND_MAX``. Same converged field, same cycle counts
        (tools/compact_cycles.py, tests); batches stay bit-identical to separate solves on such a hierarchy.

        line_factors: 'resident' (default; or the environment's EMG3D_AMD_LINE_FACTORS) keeps the line
        factorisation of every direction a level has used in HBM -- 304 B per cell and direction, ~2.0 kB per
        finest-level cell for a whole semicoarsened hierarchy with three directions; 'rebuild' keeps two directions
        per level and re-factorises on change (``DeviceLevel._line_factor_slots``): ~1.7 kB per cell, one more
        factorisation per level and cycle when the line-relaxation code cycles (+8 % per cycle); 'single' keeps one
        direction per level and re-factorises at every change of direction: ~1.1 kB per cell, +47 %."""
        self.device = device or _device()
        line_factors = line_factors or os.environ.get('EMG3D_AMD_LINE_FACTORS', 'resident')
        self.line_factors = line_factors
        self.top = DeviceLevel.from_host(vmodel, self.device, batch=batch, line_factors=line_factors)
        self.shape = tuple(vmodel.grid.shape_cells)
        self.sval = complex(vmodel._sval)
        if line_compact is None:
            line_compact = os.environ.get('EMG3D_AMD_LINE_COMPACT', 'auto')
            line_compact = {'0': False, 'false': False, '1': True, 'true': True}.get(str(line_compact).lower(), 'auto')
        if line_compact not in (True, False, 'auto'):
            raise ValueError(f"`line_compact` must be True, False or 'auto'. Provided: {line_compact!r}.")
        if line_compact == 'auto':
            line_compact = block_condition(vmodel) <= COMPACT_COND_MAX
        self.line_compact = bool(line_compact)
        # point_compact_top: also the FINEST level's tiled point smoother keeps its eta sums in single precision. That
        # level must then cycle in residual form; `solve` asks for it where it does anyway (tolerance below 1e-7, models
        # with air, multigrid as a Krylov preconditioner), not for a plain multigrid solve at a loose tolerance, where the
        # residual form would cost more (~0.9 ms per 256^3 cycle) than the narrower sums save (~0.35 ms).
        self.point_compact_top = bool(point_compact_top) and self.line_compact
        if self.line_compact:
            self.top.set_line_compact(True, point_here=self.point_compact_top)

    def check(self, vmodel):
        """A hierarchy handed to ``solve`` must belong to the same grid shape and frequency
        (that the model is the same is the caller's responsibility)."""
        if tuple(vmodel.grid.shape_cells) != self.shape or complex(vmodel._sval) != self.sval:
            raise ValueError("hierarchy: built for another grid shape or frequency")

    def put_source(self, sfield, out, sparse=False):
        """Source field -> device tensor `out`. sparse: trust the (index, value) list that
        ``get_source_field`` left on the field (valid while nobody modified the field: only
        ``parallel.solve``, which makes the field itself, asks for it) -- a dipole touches a
        handful of edges, the dense field is 100 MB at 128^3."""
        sp = getattr(sfield, '_sparse', None) if sparse else None
        # (the device assembly rounds differently from the host routine, 1e-13: only for callers that ask for it by
        #  `_sparse_source`; an untouched field goes up as the host routine's own values)
        seg = getattr(sfield, '_segments', None) if (sparse and getattr(sfield, '_assemble_on_device', True)) else None
        if seg is not None and out.device.type == 'cuda':
            # a dipole / wire made by get_source_field and not modified since: assembled on the
            # device from its few points (csrc/adjoint.h), nothing field-sized crosses PCIe
            fields.source_field_device(sfield.grid, seg[0], sfield._frequency, seg[1], out=out)
        elif sp is None:
            out.copy_(torch.from_numpy(np.ascontiguousarray(sfield.field)), non_blocking=False)
        else:
            out.zero_()
            if sp[0].size:
                out[self.top.work.upload(sp[0])] = self.top.work.upload(sp[1])

    def upload(self, sfield, efield, sparse=False):
        self.put_source(sfield, self.top.s, sparse)
        self.upload_field(efield)

    def upload_field(self, efield):
        if getattr(efield, '_is_zero', False):
            self.top.zero_field()        # the start field solve() made itself: nothing to send
        else:
            self.top.e.copy_(torch.from_numpy(np.ascontiguousarray(efield.field)), non_blocking=False)

    def download(self, efield):
        efield._is_zero = False
        out = efield.field
        if out.flags.c_contiguous and out.flags.writeable:
            torch.from_numpy(out).copy_(self.top.e)     # one pass, straight into the field's buffer
        else:
            out[:] = self.top.e.cpu().numpy()


def multigrid(model, sfield, efield, var, **kwargs):
    """Multigrid solver (reference emg3d/solver.py:471-649).

    ``model`` is a ``VolumeModel``; the result is stored in place in ``efield``; cycle
    count in ``var.it``, final error in ``var.l2``. The level hierarchy lives in HBM for the
    duration of the call; pass ``hierarchy=`` (a ``Hierarchy``) to reuse one.
    """
    hier = kwargs.get('hierarchy') or Hierarchy(model, line_compact=getattr(var, 'line_compact', None),
                                                point_compact_top=bool(getattr(var, 'residual_form', False)))
    var.hierarchy_compact = hier.line_compact
    hier.upload(sfield, efield, getattr(var, 'sparse_source', False))
    try:
        _multigrid(hier.top, var, 0, 0)
    finally:
        if getattr(var, 'download', True):
            hier.download(efield)


_GRAPH_AFTER = _cycle.GRAPH_AFTER          # (read by bench.py / tools)


def _multigrid(lv, var, level, new_cycmax):
    """Cycles on device levels: level 0 = the whole solve loop (``_cycle.run_cycles``), level >= 1
    = one coarse-grid correction with the given visit budget (``_cycle.coarse_correction``)."""
    if level == 0:
        _cycle.run_cycles(lv, var)
    else:
        _cycle.coarse_correction(lv, var, new_cycmax, first_level=level, graphed=False)


def _coarse_correction_graphed(clv, var, new_cycmax):
    _cycle.coarse_correction(clv, var, new_cycmax)


# ----------------------------------------------------------------------------- krylov ---
def krylov(model, sfield, efield, var, hierarchy=None):
    """Krylov subspace solver with multigrid preconditioner (emg3d/solver.py:652-784).

    ``bicgstab`` (the default of ``solve``), ``cgs`` and ``gcrotmk`` run entirely on the device
    (emg3d_amd/_krylov.py): vectors stay in HBM, their updates and inner products are the fused
    kernels of csrc/krylov.h with the recurrence scalars in device memory, the operator is
    ``emg3d_dev_apply_operator``, the preconditioner the multigrid cycle on the same hierarchy.
    Iterations, stopping rules and status codes are SciPy's (the reference's solvers); GCROT's
    small dense problems (Hessenberg QR, least squares) are solved on the host with SciPy's own
    routines.
    """
    from emg3d_amd import _krylov
    # (as a preconditioner every level, the finest included, solves a correction equation)
    hier = hierarchy or Hierarchy(model, line_compact=getattr(var, 'line_compact', None), point_compact_top=True)
    var.hierarchy_compact = hier.line_compact
    device_solver = {'bicgstab': _krylov.bicgstab, 'cgs': _krylov.cgs, 'gcrotmk': _krylov.gcrotmk}[var.sslsolver]
    try:
        status = _krylov_on_device(device_solver, hier, sfield, efield, var)
    except _ConvergenceError:            # the preconditioner diverged or stagnated
        status = -1
        var.exit_message += " (returned field is zero)"      # (the reference's message, emg3d/solver.py:767-770)
        if getattr(efield, '_is_zero', False):
            # the zero field solve() made itself is what comes back
            efield.field[:] = 0
            hier.top.zero_field()        # responses taken from the device field are those of the zero field too
        else:
            # a start field the caller provided stays what it was (SciPy iterates on a copy in the
            # reference, the assignment never happens); the device field follows it
            hier.upload_field(efield)

    outcome = {True: "CONVERGED", False: "MAX. ITERATION REACHED, NOT CONVERGED"}
    if status >= 0:
        var.exit_message = outcome[status == 0]
        lead = (50 * " " + "\r" if var.verb == 3 else "\n") + "   > "
    else:
        var.exit_message = var.exit_message or f"Error in {var.sslsolver} ({status})"
        lead = "\n* ERROR   :: "
    var.cprint(lead + var.exit_message, 2)


def _krylov_callback(var, l2):
    """After every Krylov iteration: counters, histories, log line (emg3d/solver.py:731-757)."""
    var.ssl_it += 1
    var.l2 = l2
    var.runtime_at_cycle = np.append(var.runtime_at_cycle, var.time.elapsed)
    var.error_at_cycle = np.append(var.error_at_cycle, l2)
    if var.verb in (2, 3):
        _print_one_liner(var, l2)
    elif var.verb > 3:
        first = var.ssl_it == 1 and var.it == 0 and var.cycle is not None
        var.cprint(f"   [{var.time.now}]   {l2 / var.l2_refe:.3e}  after {var.ssl_it:3} {var.sslsolver}-cycles"
                   + ("\n" if first else ""), 3)


def _krylov_on_device(method, hier, sfield, efield, var):
    """Source and start vector to the device, ``method`` (``_krylov.bicgstab`` / ``cgs``), solution
    back into ``efield`` -- and into ``hier.top.e``, where receivers are read from."""
    top = hier.top
    b = torch.empty(top.e.numel(), dtype=top.e.dtype, device=hier.device)
    hier.put_source(sfield, b, getattr(var, 'sparse_source', False))
    x = torch.empty_like(b)
    if getattr(efield, '_is_zero', False):
        _lib.check(_lib.lib().emg3d_dev_zero(x.data_ptr(), x.numel() * x.element_size(),
                                             torch.cuda.current_stream().cuda_stream), 'emg3d_dev_zero')
    else:
        x.copy_(torch.from_numpy(np.ascontiguousarray(efield.field)))
    status = method(hier, b, x, var, _cycle.run_cycles, lambda l2: _krylov_callback(var, l2))
    efield._is_zero = False
    top.e.copy_(x)          # the hierarchy's field is the solution (not the last preconditioner output)
    if getattr(var, 'download', True):
        out = efield.field
        if out.flags.c_contiguous and out.flags.writeable:
            torch.from_numpy(out).copy_(x)
        else:
            out[:] = x.cpu().numpy()
    return status


# ----------------------------------------- host-object wrappers (reference signatures) ---
def _level_for(model, sfield, efield):
    lv = DeviceLevel.from_host(model, _device())
    lv.s.copy_(torch.from_numpy(np.ascontiguousarray(sfield.field)))
    if efield is not None:
        lv.e.copy_(torch.from_numpy(np.ascontiguousarray(efield.field)))
    return lv


def smoothing(model, sfield, efield, nu, lr_dir):
    """Smooth ``efield`` in place (emg3d/solver.py:788-846)."""
    lv = _level_for(model, sfield, efield)

    class _V:
        smoother_cell_sweeps = 0
    _smooth(lv, nu, lr_dir, _V)
    efield.field[:] = lv.e.cpu().numpy()


def residual(model, sfield, efield, norm=False):
    """Residual field, or its l2-norm if ``norm`` (emg3d/solver.py:1022-1070)."""
    lv = _level_for(model, sfield, efield)
    if norm:
        return lv.residual(store=False, norm=True)
    lv.residual(store=True, norm=False)
    return fields.Field(sfield.grid, lv.r.cpu().numpy(), frequency=sfield._frequency)


class _CoarseModel:
    """Coarse-grid model as returned by ``restriction`` (emg3d/solver.py:909-926)."""

    def __init__(self, lv):
        shp = lv.grid.shape_cells
        self.case, self.grid = lv.case, lv.grid
        get = {}

        def host(t):
            if id(t) not in get:
                get[id(t)] = np.asfortranarray(t.cpu().numpy().reshape(shp, order='F'))
            return get[id(t)]
        self.eta_x, self.eta_y, self.eta_z = host(lv.eta_x), host(lv.eta_y), host(lv.eta_z)
        self.zeta = host(lv.zeta)


def restriction(model, sfield, residual, sc_dir):
    """Coarse model, coarse source (restricted residual) and zero coarse field
    (emg3d/solver.py:849-944)."""
    lv = _level_for(model, sfield, None)
    lv.r.copy_(torch.from_numpy(np.ascontiguousarray(residual.field)))
    c = lv.restrict_to(sc_dir)
    cs = fields.Field(c.grid, c.s.cpu().numpy(), frequency=sfield._frequency)
    ce = fields.Field(c.grid, dtype=sfield.dtype, frequency=sfield._frequency)
    return _CoarseModel(c), cs, ce


def prolongation(efield, cefield, sc_dir):
    """``efield += P cefield`` in place, PEC enforced (emg3d/solver.py:947-1019)."""

    class _M:   # prolongation needs the grids only
        pass
    m = _M()
    m.grid, m.case = efield.grid, 'isotropic'
    one = np.ones(efield.grid.shape_cells, order='F')
    m.eta_x = m.eta_y = m.eta_z = one.astype(efield.dtype)
    m.zeta = one
    lv = _level_for(m, efield, efield)
    c = lv.child(sc_dir)['level']
    if c.grid.shape_cells != cefield.grid.shape_cells:
        raise ValueError("prolongation: coarse field does not match sc_dir.")
    c.e.copy_(torch.from_numpy(np.ascontiguousarray(cefield.field)))
    lv.prolong_from(sc_dir)
    efield.field[:] = lv.e.cpu().numpy()


class RegularGridProlongator:
    """Bilinear interpolation from a coarse to a fine 2-D tensor grid with precomputed
    weights (emg3d/solver.py:1385-1478). Host utility with the reference's call
    signature; the device prolongation kernel uses the same 1-D tables
    (emg3d_amd/_device.py:interp_table)."""

    def __init__(self, cx, cy, x, y):
        from emg3d_amd._device import interp_table
        self._ix, self._wx = interp_table(np.asarray(cx, float), np.asarray(x, float))
        self._iy, self._wy = interp_table(np.asarray(cy, float), np.asarray(y, float))
        self.size = self._ix.size * self._iy.size

    def __call__(self, values):
        ix, iy = self._ix[:, None], self._iy[None, :]
        wx, wy = self._wx[:, None], self._wy[None, :]
        out = (values[ix, iy] * ((1 - wx) * (1 - wy)) + values[ix, iy + 1] * ((1 - wx) * wy) +
               values[ix + 1, iy] * (wx * (1 - wy)) + values[ix + 1, iy + 1] * (wx * wy))
        return out.ravel('F')


# --------------------------------------------------------------------------- helpers ---
def _restrict_model_parameters(param, sc_dir):
    """Sum of the 2/4/8 fine cells (emg3d/solver.py:1667-1718), on the device."""
    from emg3d_amd._device import _ptr, _stream, coarsen_flags
    _lib.require_gpu()
    dev = _device()
    p = np.asfortranarray(param)
    cplx = int(np.iscomplexobj(p))
    nx, ny, nz = p.shape
    cx, cy, cz = coarsen_flags(sc_dir)
    cshape = (nx // (2 if cx else 1), ny // (2 if cy else 1), nz // (2 if cz else 1))
    tin = torch.from_numpy(p.ravel('F').copy()).to(dev)
    tout = torch.empty(int(np.prod(cshape)), dtype=tin.dtype, device=dev)
    _lib.check(_lib.lib().emg3d_dev_restrict_param(_ptr(tout), _ptr(tin), nx, ny, nz, sc_dir,
                                                   cplx, _stream()), 'emg3d_dev_restrict_param')
    return np.asfortranarray(tout.cpu().numpy().reshape(cshape, order='F'))


def _get_restriction_weights(grid, cgrid, sc_dir):
    """(wx, wy, wz), each (wl, w0, wr); dummies for non-coarsened directions
    (emg3d/solver.py:1721-1780)."""
    from emg3d_amd import core
    out = []
    for d, skip in enumerate(([1, 5, 6], [2, 4, 6], [3, 4, 5])):
        if sc_dir not in skip:
            nodes = (grid.nodes_x, grid.nodes_y, grid.nodes_z)[d]
            cc = (grid.cell_centers_x, grid.cell_centers_y, grid.cell_centers_z)[d]
            cnodes = (cgrid.nodes_x, cgrid.nodes_y, cgrid.nodes_z)[d]
            ccc = (cgrid.cell_centers_x, cgrid.cell_centers_y, cgrid.cell_centers_z)[d]
            out.append(core.restrict_weights(nodes, cc, grid.h[d], cnodes, ccc, cgrid.h[d]))
        else:
            z = np.zeros(grid.shape_nodes[d], dtype=np.float64)
            out.append((z, np.ones(grid.shape_nodes[d], dtype=np.float64), z))
    return tuple(out)

"""Multigrid cycle control on device levels.

What the reference expresses as one recursive function (``solver.multigrid``,
emg3d/solver.py:471-649) is split here into

* ``coarse_schedule``: the visiting order of a V-, W- or F-cycle below the finest level,
  compiled ONCE per (cycle type, depth, budget) into a flat list of steps by an explicit stack
  -- it does not depend on the data, which is also what lets the same steps be captured into a
  HIP graph and replayed;
* ``CoarseCorrection``: runs such a list on the chain of ``DeviceLevel`` objects (eagerly, or
  through a captured graph);
* ``run_cycles``: the finest-level loop -- smoothing, coarse-grid correction, residual norm,
  direction cycling, log lines and the termination rules (``StopRules``).

The numbers it produces -- iteration counts, error histories, exit messages, log lines -- are
the reference's; tests compare them.
"""
import os

import numpy as np
import torch

from emg3d_amd import _lib

__all__ = ['run_cycles', 'coarse_correction', 'coarse_schedule', 'smooth_level', 'stop_reason',
           'ConvergenceError', 'current_sc_dir', 'current_lr_dir']


class ConvergenceError(Exception):
    """Multigrid as preconditioner diverged or stagnated: aborts the Krylov solver."""


# ------------------------------------------------------------------ direction rules -----
def current_sc_dir(sc_dir, grid):
    """Semicoarsening code 0..6 for this grid (emg3d/solver.py:1482-1531): a direction is
    halved only if its cell count is even, larger than two, and it is not the direction the
    cycle's ``sc_dir`` keeps. Code = which directions are KEPT: 0 none, 1/2/3 x/y/z, 4 yz, 5 xz,
    6 xy (or all)."""
    kept = tuple(n % 2 != 0 or n < 3 or sc_dir == axis + 1 for axis, n in enumerate(grid.shape_cells))
    codes = {(0, 0, 0): 0, (1, 0, 0): 1, (0, 1, 0): 2, (0, 0, 1): 3, (0, 1, 1): 4, (1, 0, 1): 5, (1, 1, 0): 6,
             (1, 1, 1): 6}
    return codes[tuple(int(k) for k in kept)]


# lr_dir -> set of line directions (1 x, 2 y, 3 z); the smoothers run in this order
_LR_AXES = {0: (), 1: (1,), 2: (2,), 3: (3,), 4: (2, 3), 5: (1, 3), 6: (1, 2), 7: (1, 2, 3)}
_LR_CODE = {axes: code for code, axes in _LR_AXES.items()}


def current_lr_dir(lr_dir, grid):
    """Line-relaxation code for this grid: directions with only two cells are dropped
    (emg3d/solver.py:1534-1588)."""
    axes = tuple(a for a in _LR_AXES[int(lr_dir)] if grid.shape_cells[a - 1] != 2)
    return _LR_CODE[axes]


def smooth_level(lv, nu, lr_dir, var):
    """``solver.smoothing`` on a device level (emg3d/solver.py:788-846): the point smoother,
    or the line smoothers of the code's directions in the order x, y, z.

    Not in the reference: ``var.smoother_omega`` != 1 extrapolates the ``nu`` sweeps of each
    direction, e <- e_before + omega (e_after - e_before) (``solve(..., smoother_omega=)``; same fixed
    point, fewer cycles on models where the multi-colour ordering costs cycles)."""
    axes = _LR_AXES[current_lr_dir(lr_dir, lv.grid)]
    omega = getattr(var, 'smoother_omega', 1.0)
    lv._factor_keep = axes             # (policy 'rebuild': the directions that should not make room)
    for lr in axes or (0,):
        if omega != 1.0:
            lv.keep_field()
        lv.smooth(lr, nu)
        if omega != 1.0:             # (per direction: measured better than once per call)
            lv.extrapolate_field(omega)
    var.smoother_cell_sweeps += nu * max(len(axes), 1) * lv.n_cells


# ------------------------------------------------------------------ visiting order ------
SMOOTH, DOWN, UP, ENTER = 'smooth', 'down', 'up', 'enter'


def coarse_schedule(cycle, cycmax, depth, first_level, budget, nu_pre, nu_coarse, nu_post):
    """Steps of the coarse-grid correction that starts at ``first_level`` (>= 1) and reaches down
    to level ``depth``: a list of tuples (step, level, nu, ...). DOWN = residual + restriction to
    level + 1, UP = prolongation from level + 1, ENTER marks the arrival on a level (log only);
    SMOOTH steps also carry what the verbose log prints (kind, visit number, visits).

    A level above the coarsest one is visited ``cycmax`` times in a row (V: 1, W: 2); in an
    F-cycle the n-th visit of a level hands a budget reduced by n to the next coarser level, so
    the second descent is a V-cycle (emg3d/solver.py:521-527, 605). The coarsest level is
    smoothed once."""
    steps = []
    # frame: [level, visits, done, awaiting_return]
    stack = [[first_level, _visits(cycle, cycmax, depth, first_level, budget), 0, False]]
    steps.append((ENTER, first_level, 0))
    while stack:
        frame = stack[-1]
        level, visits, done, back = frame
        if back:                                   # returned from the coarser level
            steps.append((UP, level, 0))
            if nu_post > 0:
                steps.append((SMOOTH, level, nu_post, "post-smoothing", done, visits))
            frame[2], frame[3] = done + 1, False
            continue
        if done == visits:
            stack.pop()
            continue
        if level == depth:
            steps.append((SMOOTH, level, nu_coarse, "coarsest level", done, visits))
            frame[2] = done + 1
            continue
        if nu_pre > 0:
            steps.append((SMOOTH, level, nu_pre, "pre-smoothing", done, visits))
        steps.append((DOWN, level, 0))
        frame[3] = True
        steps.append((ENTER, level + 1, 0))
        stack.append([level + 1, _visits(cycle, cycmax, depth, level + 1, visits - done), 0, False])
    return steps


def _visits(cycle, cycmax, depth, level, budget):
    if level == depth:
        return 1
    if budget == 0 or cycle != 'F':
        return cycmax
    return budget


class CoarseCorrection:
    """Executes a coarse schedule on the levels below ``first`` (a DeviceLevel at level >= 1 whose
    source has been filled by the restriction from above)."""

    def __init__(self, first, first_level, var):
        self.var = var
        self.levels = {first_level: first}

    def run(self, steps):
        var = self.var
        loud = var.verb > 4
        trace = var.first_cycle and var.verb > 3
        sc_at = {}
        for step in steps:
            kind, level = step[0], step[1]
            lv = self.levels[level]
            if kind == SMOOTH:
                smooth_level(lv, step[2], var.lr_dir, var)
                if loud:
                    _log_smoothing(var, level, lv, step[3], step[4], step[5])
            elif kind == DOWN:
                sc_at[level] = current_sc_dir(var.sc_dir, lv.grid)
                lv.residual(store=True, norm=False)
                self.levels[level + 1] = lv.restrict_to(sc_at[level])
            elif kind == UP:
                lv.prolong_from(sc_at[level])
                if trace:
                    var.level_all.append(level)
            elif trace:       # ENTER
                var.level_all.append(level)


def _log_smoothing(var, level, lv, what, it=0, cycmax=None):
    """Log line after a smoothing step (verb > 4; emg3d/solver.py:1865-1892)."""
    n = lv.grid.shape_cells
    norm = lv.residual(store=False, norm=True)
    visits = var.cycmax if cycmax is None else cycmax
    var.cprint(f"     {it:2} {level} {visits} [{n[0]:3}, {n[1]:3}, {n[2]:3}]: {norm:.3e} {what}", 4)


# The coarse-grid correction (everything below level 0) is hundreds of short, launch-bound
# kernels whose sequence depends only on the cycle's directions: captured once per variant into
# a HIP graph and replayed (MI355X_MICROARCH.md: a dependent kernel boundary costs ~1.5 us
# inside a graph against ~5-10 us of host time per eager launch).
USE_GRAPHS = os.environ.get('EMG3D_AMD_GRAPHS', '1') != '0'
GRAPH_AFTER = int(os.environ.get('EMG3D_AMD_GRAPH_AFTER', '2'))   # eager occurrences before capture
# > 0 while several host threads solve on one GPU (parallel.compute(per_gpu > 1)): stream capture
# is then off -- a synchronous copy in one thread is illegal while another captures
CONCURRENT = 0


def coarse_correction(clv, var, budget, first_level=1, graphed=None, top=None):
    """Everything below ``first_level - 1``: eager, or through a HIP graph captured at the
    variant's third occurrence (the first, eager one builds all levels, factors and scratch;
    capturing costs about as much host time as an eager pass and pays off only for variants
    that recur -- long solves, multigrid as a Krylov preconditioner, small grids)."""
    depth = int(var.clevel[var.sc_dir])
    steps = coarse_schedule(var.cycle, var.cycmax, depth, first_level, budget, var.nu_pre, var.nu_coarse,
                            var.nu_post)
    clv.work.factor_epoch += 1         # (policy 'rebuild': coarse levels start a correction without factors)
    runner = CoarseCorrection(clv, first_level, var)
    if graphed is None:
        graphed = first_level == 1 and var.verb < 5 and USE_GRAPHS and CONCURRENT == 0
    if not graphed:
        runner.run(steps)
        return
    cache = clv.__dict__.setdefault('_graphs', {})
    # (the library's options select kernels and factor buffers: a graph captured under other
    # options would replay launches into buffers that may have been freed since)
    key = (int(var.sc_dir), int(var.lr_dir), budget, var.cycle, var.nu_pre, var.nu_post, var.nu_coarse, depth,
           float(getattr(var, 'smoother_omega', 1.0)), _lib.options_fingerprint())
    entry = cache.get(key)
    if entry is None or entry['graph'] is None:
        if entry is None:
            # graphs captured under other option values will not be replayed again (the options select kernels
            # and buffers): drop them, and with them the option-dependent factor buffers they kept alive
            stale = [k for k in cache if k[-1] != key[-1]]
            if stale:
                for k in stale:
                    del cache[k]
                # (from the finest level on, where the caller names it: its eta-sum buffer is the largest of them, and
                #  the subtrees of the other semicoarsening directions hang off it. Their graph caches go first: a
                #  sibling's graph captured under the old options holds the addresses of the buffers freed here, and its
                #  key would match again as soon as the options return to their old values)
                root = top if top is not None else clv
                _purge_stale_graphs(root, key[-1])
                _drop_stale_point_factors(root)
            entry = cache[key] = {'work': None, 'graph': None, 'seen': 0}
        if entry['seen'] < GRAPH_AFTER:
            before = var.smoother_cell_sweeps
            runner.run(steps)
            entry['work'] = var.smoother_cell_sweeps - before
            entry['seen'] += 1
            return
        before = var.smoother_cell_sweeps
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: other host threads may launch their own solves meanwhile
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            runner.run(steps)
        entry['work'] = var.smoother_cell_sweeps - before
        var.smoother_cell_sweeps = before           # capturing does not execute
        entry['graph'] = graph
    entry['graph'].replay()
    var.smoother_cell_sweeps += entry['work']


def _purge_stale_graphs(lv, fingerprint, seen=None):
    """Delete every captured graph below `lv` that was captured under another option set than `fingerprint`."""
    seen = set() if seen is None else seen
    if id(lv) in seen:
        return
    seen.add(id(lv))
    cache = lv.__dict__.get('_graphs')
    if cache:
        for k in [k for k in cache if k[-1] != fingerprint]:
            del cache[k]
    for link in lv.children.values():
        _purge_stale_graphs(link['level'], fingerprint, seen)


def _drop_stale_point_factors(lv, seen=None):
    """The eta-sum buffers of the point smoother are kept per value of the option point_tile_min
    (DeviceLevel.point_factors); once no graph of another option set is left, only the current value's are."""
    seen = set() if seen is None else seen
    if id(lv) in seen:
        return
    seen.add(id(lv))
    now = ('point', _lib.lib().emg3d_get_option(b'point_tile_min'))
    for k in [k for k in lv._factors if isinstance(k, tuple) and k[0] == 'point' and k != now]:
        del lv._factors[k]
    for link in lv.children.values():
        _drop_stale_point_factors(link['level'], seen)


# ------------------------------------------------------------------ termination ---------
def stop_reason(var, l2_last, l2_stag, it):
    """Why the cycling ends after this cycle, as (message, is_failure), or None to carry on
    (emg3d/solver.py:1622-1645). Checked in this order: converged; diverged (error above ten
    times the reference, or not finite); stagnated (after more than two cycles the error is not
    below what it was one round of the direction schedule ago); iteration limit."""
    if l2_last < var.tol * var.l2_refe:
        return "CONVERGED", False
    if l2_last > 10 * var.l2_refe or not np.isfinite(l2_last):
        return "DIVERGED", True
    if it > 2 and l2_last >= l2_stag:
        return "STAGNATED", True
    if it == var.maxit:
        return "MAX. ITERATION REACHED, NOT CONVERGED", False
    return None


def terminate(var, l2_last, l2_stag, it):
    """True when the cycling is over; sets ``var.exit_message``. As preconditioner of a Krylov
    solver a failure raises ``ConvergenceError`` and reaching ``maxit`` is silent
    (emg3d/solver.py:1591-1664)."""
    reason = stop_reason(var, l2_last, l2_stag, it)
    if reason is None:
        return False
    message, failure = reason
    precond = bool(var.sslsolver)
    if not (precond and message.startswith("MAX.")):
        var.exit_message = message
    if precond and failure:
        raise ConvergenceError
    if not precond:
        lead = {3: 50 * " " + "\r"}.get(var.verb, "\n" if var.verb < 5 else "")
        var.cprint(f"{lead}   > {var.exit_message}", 2)
    return True


# ------------------------------------------------------------------ log lines -----------
def one_liner(var, l2_last, last=False):
    """The continuously updated status line of verb 1-3 (emg3d/solver.py:1895-1919)."""
    count = f"{var.ssl_it}({var.it}); " if var.sslsolver else f"{var.it}; "
    text = f":: emg3d :: {l2_last / var.l2_refe:.1e}; {count}{var.time.runtime}"
    if last:
        var.cprint(f"{text}; {var.exit_message}", -100)
    else:
        var.cprint(text, -100, end='\r')


def record_cycle(var, l2_last, l2_prev):
    """Book-keeping and log line after a finest-level cycle (emg3d/solver.py:1788-1862; the
    ASCII picture of the first cycle is not drawn)."""
    var.runtime_at_cycle = np.append(var.runtime_at_cycle, var.time.elapsed)
    var.error_at_cycle = np.append(var.error_at_cycle, l2_last)
    if var.verb in (2, 3):
        one_liner(var, l2_last)
    if var.verb < 4:
        return
    var.first_cycle = False
    stamp = f"   [{var.time.now}]   {l2_last / var.l2_refe:.3e}  "
    if var.sslsolver:
        body = f"after {19 * ' '} {var.it:3} {var.cycle}-cycles "
    else:
        body = f"after {var.it:3} {var.cycle}-cycles   [{l2_last:.3e}, {l2_last / l2_prev:.3f}]"
    pad = "\n" if var.verb > 4 else ""
    var.cprint(f"{pad}{stamp}{body}   {var.lr_dir} {var.sc_dir}{pad}", 3)


# ------------------------------------------------------------------ finest level --------
def run_cycles(top, var):
    """Multigrid cycles on the finest level until ``terminate`` says stop (or, for benchmarking,
    exactly ``var.fixed_cycles`` cycles). In place on ``top.e``; sets ``var.it``, ``var.l2`` and
    the histories. (The finest-level part of emg3d/solver.py:512-649.)"""
    loud = var.verb > 4
    if var.first_cycle and var.verb > 3:
        var.level_all.append(0)
    # Residual form (solve(..., residual_form=); not for preconditioner calls, which are in that form
    # already, nor for batches): every cycle solves A d = r = s - A e from d = 0 and adds d to e.
    # For this linear, stationary iteration that is the same cycle in exact arithmetic; in floating
    # point the errors of the smoothers -- the line solves multiply by stored block inverses: eps x
    # cond of a block, relative to the right-hand side they are given -- then scale with the
    # residual, not with the field, and the iteration converges to round-off.
    resform = bool(getattr(var, 'residual_form', False)) and not var.sslsolver and top.batch == 1
    if (not resform and not var.sslsolver and top.batch == 1 and
            getattr(top, 'uses_line_compact', lambda lines=True: False)(lines=bool(var.lr_cycle) or var.lr_dir != 0)):
        # compact line records (solver.Hierarchy(line_compact=...)): the streamed line solves are perturbed by
        # eps32 x cond, the finest level must see residuals, not the field
        resform = var.residual_form = True
    if resform:
        top._b_valid = False
    if getattr(top, '_resmode', None) is not None:      # (left behind by a solve that was interrupted: this solve's
        top._resmode = None                             #  field and source are in the level's own buffers)
    l2_last = top.residual(store=resform, norm=True)
    ring = np.full(var.maxcycle, l2_last)           # errors one round of the schedule ago
    var.cprint("     it cycmax               error", 4)
    var.cprint("      level [  dimension  ]            info\n", 4)
    if loud:
        n = top.grid.shape_cells
        var.cprint(f"     {0:2} 0 {_level0_visits(var)} [{n[0]:3}, {n[1]:3}, {n[2]:3}]: {l2_last:.3e} initial error", 4)
    if var.nu_init > 0:
        smooth_level(top, var.nu_init, var.lr_dir, var)
        if loud:
            _log_smoothing(var, 0, top, "initial smoothing", 0, _level0_visits(var))
        if resform:
            top.residual(store=True, norm=False)
    fixed = getattr(var, 'fixed_cycles', None)
    try:
        _finest_level_loop(top, var, resform, l2_last, ring, fixed, loud)
    finally:
        _leave(top)             # (residual form: the field and the source back into the level's own buffers)


def _finest_level_loop(top, var, resform, l2_last, ring, fixed, loud):
    it = 0
    while True:
        l2_prev = l2_last
        ring[(it - 1) % var.maxcycle] = l2_last
        entered = False
        try:
            if resform:
                top.to_residual_equation()
                entered = True
            _one_cycle(top, var, it, loud)
            if resform:
                top.from_residual_equation()
                entered = False
        except BaseException:
            if entered:        # (e, s) hold (d, r): give the caller's field and source back before unwinding
                try:
                    top.abandon_residual_equation()
                except Exception:      # (a HIP error took the cycle down: the copies fail too -- the first error counts)
                    pass
            raise
        it += 1
        var.it += 1
        l2_last = top.residual(store=resform, norm=True)
        record_cycle(var, l2_last, l2_prev)
        if var.sc_cycle:
            var.sc_dir = next(var.sc_cycle)
        if var.lr_cycle:
            var.lr_dir = next(var.lr_cycle)
        if fixed:
            if it >= fixed:
                break
            continue
        l2_stag = ring[(it - 1) % var.maxcycle]
        reason = stop_reason(var, l2_last, l2_stag, it)
        if (reason is not None and reason[0] == "STAGNATED" and it < var.maxit and l2_last < 1e-3 * var.l2_refe and
                not resform and top.batch == 1 and
                not var.sslsolver and getattr(var, 'residual_form_auto', False) and _can_switch(top)):
            # The direct form has stalled above the tolerance: the floor of the line smoothers' stored
            # block inverses (or of the point smoother's pivots) on this model lies higher than the 'auto'
            # rule of solver._residual_form predicted. The reference's banded LDL^T would go on converging;
            # so do the cycles from here on, on the residual equation (their round-off scales with the
            # residual). Stagnation is judged afresh after another round of the direction schedule. (An
            # iteration that stagnates before it has gained three orders is not at a round-off floor.)
            resform = var.residual_form = var.residual_form_switched = True
            top._b_valid = False
            top.residual(store=True, norm=False)
            ring[:] = np.inf
            var.cprint(f"   > stalled at {l2_last / var.l2_refe:.1e} in direct form: continuing on the residual equation", 3)
            continue
        if terminate(var, l2_last, l2_stag, it):
            break
    var.l2 = l2_last


def _leave(top):
    leave = getattr(top, 'leave_residual_form', None)
    if leave is not None:
        leave()


def _can_switch(top):
    """The residual form needs two more field-sized buffers on the finest level: reserved here, before the
    switch; without the memory for them the solve ends as it would have without the switch (STAGNATED)."""
    try:
        top.reserve_residual_equation()
    except torch.cuda.OutOfMemoryError:
        return False
    return True


def _one_cycle(top, var, it, loud):
    """Smoothing, coarse-grid correction, smoothing on the finest level (emg3d/solver.py:512-637)."""
    if var.clevel[var.sc_dir] == 0:            # a single level: nothing to recurse into
        smooth_level(top, var.nu_coarse, var.lr_dir, var)
        if loud:
            _log_smoothing(var, 0, top, "coarsest level", it, 1)
        return
    if var.nu_pre > 0:
        smooth_level(top, var.nu_pre, var.lr_dir, var)
        if loud:
            _log_smoothing(var, 0, top, "pre-smoothing", it)
    sc = current_sc_dir(var.sc_dir, top.grid)
    top.residual(store=True, norm=False)
    coarse_correction(top.restrict_to(sc), var, var.cycmax, top=top)
    top.prolong_from(sc)
    if var.first_cycle and var.verb > 3:
        var.level_all.append(0)
    if var.nu_post > 0:
        smooth_level(top, var.nu_post, var.lr_dir, var)
        if loud:
            _log_smoothing(var, 0, top, "post-smoothing", it)


def _level0_visits(var):
    return 1 if var.clevel[var.sc_dir] == 0 else var.cycmax

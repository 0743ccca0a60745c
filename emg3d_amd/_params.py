"""Solver settings and per-solve state.

``MGParameters`` is the object ``solve`` hands through ``multigrid`` / ``krylov`` -- the
boundary B2 of DESIGN.md. Its constructor arguments, defaults, attribute names, error
messages and printed summary are the reference's (emg3d/solver.py:1074-1381), because
callers and tests read them; how the values are derived is organised here as

* a table of option specifications (name -> default, validator),
* ``_DirectionSchedule``: the semicoarsening / line-relaxation direction cycling as a small
  iterator class (``next(var.sc_cycle)``; falsy when there is a single, fixed direction),
* ``coarsening_depths``: how often each direction can be halved.
"""
import time
from datetime import datetime, timedelta

import numpy as np

__all__ = ['MGParameters', 'Timer']


class Timer:
    """Wall-clock timer with the reference's attributes (emg3d/utils.py:169-197)."""

    def __init__(self):
        self._t0 = time.perf_counter()

    @property
    def t0(self):
        return self._t0

    @property
    def now(self):
        return datetime.now().strftime("%H:%M:%S")

    @property
    def elapsed(self):
        return time.perf_counter() - self._t0

    @property
    def runtime(self):
        return str(timedelta(seconds=np.round(self.elapsed)))


# --------------------------------------------------------------------- directions -------
class _DirectionSchedule:
    """Directions a solve cycles through, one per multigrid cycle.

    ``spec`` is what the user passes as ``semicoarsening`` / ``linerelaxation``: ``True`` (the
    default rotation), ``False`` / a single digit (fixed direction), or a multi-digit integer
    whose digits are the rotation (e.g. 1213). ``next(schedule)`` yields the direction of the
    following cycle; a schedule with a single direction is falsy and is never advanced
    (reference behaviour: ``sc_cycle`` / ``lr_cycle`` are ``False`` then, emg3d/solver.py:1272-1339).
    """

    def __init__(self, spec, rotation, ndirs, complaint):
        if spec is True:
            digits = list(rotation)
        elif spec is False or (isinstance(spec, (int, np.integer)) and 0 <= spec < ndirs):
            digits = [int(spec)]
        else:
            try:
                digits = [int(ch) for ch in str(abs(int(spec)))]
            except (TypeError, ValueError):
                raise ValueError(complaint.format(spec)) from None
            if any(d >= ndirs for d in digits):
                raise ValueError(complaint.format(spec))
        self.directions = np.array(digits)
        self.rotates = spec is True or len(digits) > 1
        self._pos = 0

    def __bool__(self):
        return self.rotates

    def __len__(self):
        return len(self.directions)

    def __iter__(self):
        return self

    def __next__(self):
        d = self.directions[self._pos % len(self.directions)]
        self._pos += 1
        return d

    def first(self):
        """Direction of the first cycle (consumes it when the schedule rotates)."""
        return next(self) if self.rotates else self.directions[0]


_SC_COMPLAINT = ("`semicoarsening` must be one of {{False;True;0;1;2;3}}. "
                 "Or a combination of {{0;1;2;3}} to cycle, e.g. 1213. "
                 "Provided: {}.")
_LR_COMPLAINT = ("`linerelaxation` must be one of "
                 "{{False;True;0;1;2;3;4;5;6;7}}. Or a combination of "
                 "{{1;2;3;4;5;6;7}} to cycle, e.g. 1213. "
                 "Provided: {}.")


# ----------------------------------------------------------------- coarsening depth -----
def coarsening_depths(shape_cells, limit):
    """Per direction: how many times the cell count can be halved (even and > 2 each time,
    emg3d/solver.py:1214-1224), capped by ``limit`` when that is >= 0. Returns
    (depths[3], coarsest shape)."""
    depths = []
    for n in shape_cells:
        d = 0
        while n % 2 == 0 and n > 2 and (limit < 0 or d < limit):
            n //= 2
            d += 1
        depths.append(d)
    coarsest = tuple(int(n) // 2 ** d for n, d in zip(shape_cells, depths))
    return np.array(depths, dtype=np.int64), coarsest


# ------------------------------------------------------------------------ options -------
def _check_cycle(value):
    if value not in ('F', 'V', 'W', None):
        raise ValueError("`cycle` must be one of {'F';'V';'W';None}. "
                         f"Provided: {value}.")
    return value


_SOLVERS = ['bicgstab', 'cgs', 'gcrotmk']


def _check_sslsolver(value):
    if value is True:
        return 'bicgstab'
    if value is False or value in _SOLVERS:
        return value
    raise ValueError(f"`sslsolver` must be True, False, or one of {_SOLVERS}. "
                     f"Provided: {value!r}.")


def _same(value):
    return value


# keyword -> (default, normaliser); the order is the reference's signature order
_OPTIONS = {
    'cycle': ('F', _check_cycle),
    'tol': (1e-6, _same),
    'maxit': (50, _same),
    'nu_init': (0, _same),
    'nu_pre': (2, _same),
    'nu_coarse': (1, _same),
    'nu_post': (2, _same),
    'clevel': (-1, _same),
    'return_info': (False, _same),
    'log': (0, _same),
}


class MGParameters:
    """Multigrid solver settings and state (reference emg3d/solver.py:1074-1381).

    ``MGParameters(verb, sslsolver, semicoarsening, linerelaxation, shape_cells, cycle='F',
    tol=1e-6, maxit=50, nu_init=0, nu_pre=2, nu_coarse=1, nu_post=2, clevel=-1,
    return_info=False, log=0)``.
    """

    def __init__(self, verb, sslsolver, semicoarsening, linerelaxation, shape_cells, *args, **kwargs):
        names = list(_OPTIONS)
        if len(args) > len(names):
            raise TypeError(f"MGParameters takes at most {5 + len(names)} positional arguments")
        given = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in _OPTIONS:
                raise TypeError(f"MGParameters got an unexpected keyword argument {k!r}")
            if k in given:
                raise TypeError(f"MGParameters got multiple values for argument {k!r}")
            given[k] = v
        self.verb = verb
        self.shape_cells = shape_cells
        for k, (default, norm) in _OPTIONS.items():
            setattr(self, k, norm(given.get(k, default)))

        # per-solve state
        self.it, self.ssl_it = 0, 0              # multigrid cycles / Krylov iterations so far
        self.l2, self.l2_refe = 1.0, 1.0         # current and reference error
        self.exit_message, self.log_message = '', ''
        self.time = Timer()
        self.runtime_at_cycle = np.array([0.])
        self.error_at_cycle = np.array([0.])
        self.do_return = True
        self.level_all, self.first_cycle = [], True      # levels visited in the first cycle (verb > 3)
        self.smoother_cell_sweeps = 0            # sum over smoother calls of nu * n_cells (not in the reference)

        self._set_depths()
        self._set_directions(semicoarsening, linerelaxation)
        self._set_solver(sslsolver)

    # ---------------------------------------------------------------- derived settings --
    def _set_depths(self):
        shape = tuple(self.shape_cells)
        if min(shape) < 2:
            raise ValueError(
                "Nr. of cells must be at least two in each direction "
                f"Provided shape: ({shape[0]}, {shape[1]}, "
                f"{shape[2]}).")
        requested = self.clevel
        depths, coarsest = coarsening_depths(shape, requested)
        dx, dy, dz = depths
        # index = semicoarsening direction 0..3: the direction that is kept does not count
        self.clevel = np.array([max(dx, dy, dz), max(dy, dz), max(dx, dz), max(dx, dy)])
        cap = np.inf if requested < 0 else requested
        improvable = any(d < cap and n > 7 for d, n in zip(depths, coarsest)) or any(depths < min(cap, 3))
        self._repr_clevel = {
            'n_cells': int(np.prod(coarsest)), 'shape_cells': coarsest, 'clevel': depths,
            'message': "  :: Grid not optimal for MG solver ::" if improvable else ""}

    def _set_directions(self, semicoarsening, linerelaxation):
        sc = _DirectionSchedule(semicoarsening, (1, 2, 3), 4, _SC_COMPLAINT)
        lr = _DirectionSchedule(linerelaxation, (4, 5, 6), 8, _LR_COMPLAINT)
        self.raw_sc_cycle, self.raw_lr_cycle = sc.directions, lr.directions
        self.sc_dir, self.lr_dir = sc.first(), lr.first()
        self.sc_cycle = sc if sc else False
        self.lr_cycle = lr if lr else False
        self.semicoarsening = bool(self.sc_dir != 0)
        self.linerelaxation = bool(self.lr_dir != 0)
        self._repr_sc_dir = f"{self.semicoarsening} {sc.directions}"
        self._repr_lr_dir = f"{self.linerelaxation} {lr.directions}"
        self.maxcycle = max(len(sc), len(lr))

    def _set_solver(self, sslsolver):
        self.sslsolver = _check_sslsolver(sslsolver)
        if not self.sslsolver and not self.cycle:
            raise ValueError(
                "At least `cycle` or `sslsolver` is required. Provided"
                f"input: cycle={self.cycle}; sslsolver={self.sslsolver}.")
        self.cycmax = {'F': 2, 'W': 2}.get(self.cycle, 1)      # visits of the next coarser level per level
        self._repr_maxit = f"{self.maxit}"
        self.ssl_maxit = 0
        if self.sslsolver:
            # maxit bounds the Krylov iterations; multigrid as preconditioner runs one round of
            # the direction schedule per application
            self.ssl_maxit = self.maxit
            if self.cycle is not None:
                self.maxit = self.maxcycle
                self._repr_maxit += f" ({self.maxit})"

    # ----------------------------------------------------------------------- output -----
    def __repr__(self):
        n = self.shape_cells
        rc = self._repr_clevel
        c, d = rc['shape_cells'], rc['clevel']
        rows = [
            (f"MG-cycle       : {self.cycle!r:17}", f"sslsolver : {self.sslsolver!r}"),
            (f"semicoarsening : {self._repr_sc_dir:17}", f"tol       : {self.tol}"),
            (f"linerelaxation : {self._repr_lr_dir:17}", f"maxit     : {self._repr_maxit}"),
            (f"nu_{{i,1,c,2}}   : {self.nu_init}, {self.nu_pre}, {self.nu_coarse}, {self.nu_post}       ",
             f"verb      : {self.verb}"),
            (f"Original grid  : {n[0]:3} x {n[1]:3} x {n[2]:3}  ", f"=> {n[0] * n[1] * n[2]:,} cells"),
            (f"Coarsest grid  : {c[0]:3} x {c[1]:3} x {c[2]:3}  ", f"=> {rc['n_cells']:,} cells"),
            (f"Coarsest level : {d[0]:3} ; {d[1]:3} ;{d[2]:4}", f"{rc['message']}"),
        ]
        return "".join(f"   {left}   {right}\n" for left, right in rows)

    def cprint(self, info, verbosity, **kwargs):
        """Print and/or log ``info`` if ``self.verb > verbosity`` (``log``: -1 log only, 0 print
        only, 1 both; emg3d/solver.py:1181-1200)."""
        if self.verb <= verbosity:
            return
        if self.log:
            self.log_message += f"{info}\n"
        if self.log >= 0:
            print(info, **kwargs)

"""Device BiCGSTAB, CGS and GCROT(m,k) around the multigrid preconditioner.

The reference hands ``scipy.sparse.linalg.bicgstab / cgs / gcrotmk`` a host ``LinearOperator``
and a host preconditioner (emg3d/solver.py:652-784). Here the vectors never leave HBM: every
update and inner product is one ``emg3d_dev_krylov_step`` (csrc/krylov.h), the recurrence
scalars live in a small device table, and the host reads that table only where the algorithm
branches -- twice per BiCGSTAB iteration, together with the true residual norm the reference's
callback reports. Iteration, breakdown tests, stopping rule ``|r| <= max(atol, rtol |b|)`` and
return codes are those of SciPy >= 1.12 (``rtol=tol, atol=1e-30, maxiter=maxit``): 0 converged,
> 0 iteration limit, -10 / -11 breakdown (rho / omega or rt.v vanish).
"""
import ctypes

import numpy as np
import torch

from emg3d_amd import _lib
from emg3d_amd._device import _ptr, _stream

_vp = ctypes.c_void_p
DIV, MUL, NEG, COPY = 0, 1, 2, 3           # scalar instructions of emg3d_dev_krylov_step


class Vectors:
    """Work vectors of one Krylov solve on the device of ``top`` + the scalar table."""

    NSLOTS = 32

    def __init__(self, top, n=None, nslots=None):
        self.top = top
        self.n = top.e.numel() if n is None else int(n)       # (n: one right-hand side of a batch)
        self.is_complex = top.is_complex
        if nslots is not None:
            self.NSLOTS = int(nslots)
        self.table = torch.zeros(2 * self.NSLOTS, dtype=torch.float64, device=top.device)
        self.ws = torch.empty(_lib.lib().emg3d_krylov_ws_len(), dtype=torch.float64, device=top.device)
        self._names = {}
        self._host = torch.empty(2 * self.NSLOTS, dtype=torch.float64).pin_memory()
        self._stage = torch.empty(2 * self.NSLOTS, dtype=torch.float64).pin_memory()

    def put(self, names, values):
        """Host scalars into the table slots ``names`` (coefficients the host computed from a small
        dense problem): consecutive slots, one small copy."""
        s0 = self.slot(names[0])
        for i, nm in enumerate(names):
            assert self.slot(nm) == s0 + i, 'put: the slots must be consecutive'
        torch.cuda.current_stream().synchronize()        # (the staging buffer may still be in flight)
        h = self._stage.numpy()
        for i, v in enumerate(values):
            h[2 * i], h[2 * i + 1] = np.real(v), np.imag(v)
        self.table[2 * s0:2 * (s0 + len(names))].copy_(self._stage[:2 * len(names)], non_blocking=True)

    def new(self):
        return torch.empty(self.n, dtype=self.top.dtype, device=self.top.device)

    def slot(self, name):
        if name not in self._names:
            self._names[name] = len(self._names)
            assert len(self._names) <= self.NSLOTS
        return self._names[name]

    def step(self, y=None, terms=(), dots=(), prog=()):
        """y = sum coef * x over ``terms`` = [(x, coef)], coef a slot name, a float, or
        (slot name, float); ``dots`` = [(slot name, a, b)] -> table[slot] = conj(a) . b (with the new
        y); ``prog`` = [(op, dst, a, b)] with slot names (b may be None)."""
        nt, nd, npg = len(terms), len(dots), len(prog)
        xs = (_vp * max(nt, 1))()
        slots = (ctypes.c_int * max(nt, 1))()
        scales = (ctypes.c_double * max(nt, 1))()
        for i, (x, coef) in enumerate(terms):
            xs[i] = x.data_ptr()
            if isinstance(coef, tuple):
                slots[i], scales[i] = self.slot(coef[0]), float(coef[1])
            elif isinstance(coef, str):
                slots[i], scales[i] = self.slot(coef), 1.0
            else:
                slots[i], scales[i] = -1, float(coef)
        das, dbs = (_vp * max(nd, 1))(), (_vp * max(nd, 1))()
        dslots = (ctypes.c_int * max(nd, 1))()
        for k, (name, a, b) in enumerate(dots):
            das[k], dbs[k], dslots[k] = a.data_ptr(), b.data_ptr(), self.slot(name)
        pr = (ctypes.c_int * max(4 * npg, 1))()
        for i, (op, dst, a, b) in enumerate(prog):
            pr[4 * i:4 * i + 4] = [op, self.slot(dst), self.slot(a), -1 if b is None else self.slot(b)]
        _lib.check(_lib.lib().emg3d_dev_krylov_step(
            self.n, self.is_complex, _ptr(y) if y is not None else None, nt, xs, slots, scales, nd, das, dbs, dslots,
            npg, pr, _ptr(self.table), _ptr(self.ws), self.ws.numel(), _stream()), 'emg3d_dev_krylov_step')

    def read(self, *names, extra=None):
        """Table entries (complex, or float for real fields) on the host: ONE synchronising copy,
        which also brings ``extra`` (a small device tensor, e.g. a residual sum of squares)."""
        self._host.copy_(self.table, non_blocking=True)
        ex = extra.to('cpu', non_blocking=True) if extra is not None else None
        torch.cuda.current_stream().synchronize()
        h = self._host.numpy()
        out = []
        for nm in names:
            s = self.slot(nm)
            out.append(complex(h[2 * s], h[2 * s + 1]) if self.is_complex else float(h[2 * s]))
        return (out, ex.numpy().copy()) if extra is not None else out

    def copy(self, dst, src):
        _lib.check(_lib.lib().emg3d_dev_copy(_ptr(dst), _ptr(src), src.numel() * src.element_size(), _stream()),
                   'emg3d_dev_copy')


def _preconditioner(top, var, run_cycles, vec):
    """out <- M vec: multigrid cycles on (source = vec, field = 0); identity without a cycle."""
    def apply(src, out):
        if not var.cycle:
            vec.copy(out, src)
            return
        vec.copy(top.s, src)
        top.zero_field()
        run_cycles(top, var)            # maxit = one round of the direction schedule (solver.py:1376-1381)
        vec.copy(out, top.e)
    return apply


def bicgstab(hier, b, x, var, run_cycles, callback):
    """Preconditioned BiCGSTAB (van der Vorst 1992). ``b``, ``x``: device vectors (x is updated
    in place). ``callback(l2)`` after every iteration with the true residual norm. Returns the
    SciPy status code."""
    top = hier.top
    V = Vectors(top)
    psolve = _preconditioner(top, var, run_cycles, V)
    r, v, t, p, phat, shat, rt = (V.new() for _ in range(7))
    eps = np.finfo(np.float64).eps
    rhotol = omegatol = eps ** 2

    top.apply_A(x, r)
    V.step(r, [(b, 1.0), (r, -1.0)], dots=[('bb', b, b)])
    V.copy(rt, r)
    V.step(None, dots=[('rr', r, r), ('rho', rt, r)])
    bb, rr, rho = V.read('bb', 'rr', 'rho')
    atol = max(1e-30, var.tol * np.sqrt(abs(bb)))
    omega = alpha = 1.0
    for iteration in range(var.ssl_maxit):
        if np.sqrt(abs(rr)) < atol:
            return 0
        if abs(rho) < rhotol:
            return -10
        if iteration > 0:
            if abs(omega) < omegatol:
                return -11
            V.step(p, [(r, 1.0), (p, 'beta'), (v, 'nbo')])      # p = r + beta (p - omega v)
        else:
            V.copy(p, r)
        psolve(p, phat)
        top.apply_A(phat, v)
        # alpha = rho / (rt . v);  s = r - alpha v  (kept in r)
        V.step(None, dots=[('rv', rt, v)], prog=[(DIV, 'alpha', 'rho', 'rv'), (NEG, 'nalpha', 'alpha', None)])
        V.step(r, [(r, 1.0), (v, 'nalpha')], dots=[('ss', r, r)])
        rv, ss, alpha = V.read('rv', 'ss', 'alpha')
        if rv == 0:
            return -11
        if np.sqrt(abs(ss)) < atol:
            V.step(x, [(x, 1.0), (phat, 'alpha')])
            return 0
        psolve(r, shat)
        top.apply_A(shat, t)
        # omega = (t . s) / (t . t);  x += alpha phat + omega shat;  r = s - omega t
        V.step(None, dots=[('ts', t, r), ('tt', t, t)],
               prog=[(DIV, 'omega', 'ts', 'tt'), (NEG, 'nomega', 'omega', None)])
        V.step(x, [(x, 1.0), (phat, 'alpha'), (shat, 'omega')])
        V.step(r, [(r, 1.0), (t, 'nomega')], dots=[('rr', r, r), ('rho_next', rt, r)],
               prog=[(DIV, 'q1', 'rho_next', 'rho'), (DIV, 'q2', 'alpha', 'omega'), (MUL, 'beta', 'q1', 'q2'),
                     (MUL, 'bo', 'beta', 'omega'), (NEG, 'nbo', 'bo', None), (COPY, 'rho', 'rho_next', None)])
        sumsq = top.residual_sumsq(x, b)                 # the reference's callback: |b - A x|
        (rr, rho, omega), l2 = V.read('rr', 'rho', 'omega', extra=sumsq)
        callback(float(np.sqrt(l2[0])))
    return var.ssl_maxit


def cgs(hier, b, x, var, run_cycles, callback):
    """Preconditioned CGS (Sonneveld 1989), as scipy.sparse.linalg.cgs iterates."""
    top = hier.top
    V = Vectors(top)
    psolve = _preconditioner(top, var, run_cycles, V)
    r, rt, u, p, q, phat, vhat, uhat, tmp = (V.new() for _ in range(9))
    eps = np.finfo(np.float64).eps
    rhotol = eps ** 2

    top.apply_A(x, r)
    V.step(r, [(b, 1.0), (r, -1.0)], dots=[('bb', b, b)])
    V.copy(rt, r)
    V.step(None, dots=[('rr', r, r), ('rho', rt, r)])
    bb, rr, rho = V.read('bb', 'rr', 'rho')
    atol = max(1e-30, var.tol * np.sqrt(abs(bb)))
    if np.sqrt(abs(bb)) == 0:
        return 0
    for iteration in range(var.ssl_maxit):
        if np.sqrt(abs(rr)) < atol:
            return 0
        if abs(rho) < rhotol:
            return -10
        if iteration > 0:
            V.step(u, [(r, 1.0), (q, 'beta')])                   # u = r + beta q
            V.step(p, [(u, 1.0), (q, 'beta'), (p, 'beta2')])     # p = u + beta (q + beta p)
        else:
            V.copy(p, r)
            V.copy(u, r)
        psolve(p, phat)
        top.apply_A(phat, vhat)
        V.step(None, dots=[('rv', rt, vhat)], prog=[(DIV, 'alpha', 'rho', 'rv'), (NEG, 'nalpha', 'alpha', None)])
        V.step(q, [(u, 1.0), (vhat, 'nalpha')])                  # q = u - alpha vhat
        V.step(tmp, [(u, 1.0), (q, 1.0)])
        rv, = V.read('rv')
        if rv == 0:
            return -11
        psolve(tmp, uhat)
        V.step(x, [(x, 1.0), (uhat, 'alpha')])
        # SciPy recomputes the residual from x (error build-up of the recurrence r -= alpha A uhat)
        top.apply_A(x, tmp)
        V.step(r, [(b, 1.0), (tmp, -1.0)], dots=[('rr', r, r), ('rho_next', rt, r)],
               prog=[(DIV, 'beta', 'rho_next', 'rho'), (MUL, 'beta2', 'beta', 'beta'), (COPY, 'rho', 'rho_next', None)])
        rr, rho = V.read('rr', 'rho')
        callback(float(np.sqrt(abs(rr))))          # = |b - A x|, what the reference's callback evaluates
    return var.ssl_maxit


def _combine(V, out, vecs, names):
    """out = sum_i table[names[i]] * vecs[i] (any number of terms, four per launch)."""
    terms = [(v, nm) for v, nm in zip(vecs, names)]
    V.step(out, terms[:4])
    for i in range(4, len(terms), 3):
        V.step(out, [(out, 1.0)] + terms[i:i + 3])


def gcrotmk(hier, b, x, var, run_cycles, callback, m=20, k=None):
    """Flexible GCROT(m,k) (de Sturler 1999; Hicken & Zingg 2010) as scipy.sparse.linalg.gcrotmk
    iterates it with its defaults (m = 20 inner FGMRES steps, k = m recycled pairs, the oldest pair
    dropped): the vectors -- up to m + k Arnoldi vectors and as many preconditioned ones, k pairs
    (c, u) -- stay in HBM; the modified Gram-Schmidt sweep of an inner step is a chain of fused
    launches (subtract the previous projection, inner product with the next vector: csrc/krylov.h)
    whose coefficients the host reads ONCE per inner step; the small dense problems (QR update of
    the Hessenberg matrix, least squares) run on the host with SciPy's own routines, exactly as in
    scipy/sparse/linalg/_isolve/_gcrotmk.py. ``callback(l2)``: at the start of every outer
    iteration with the true residual norm, as the reference's callback evaluates it. Returns the
    SciPy status: 0 converged, else the number of outer iterations."""
    from numpy.linalg import LinAlgError
    from scipy.linalg import qr_insert, lstsq
    top = hier.top
    k = m if k is None else k
    nproj = m + 2 * k + 2                                 # Gram-Schmidt coefficients of one inner step
    V = Vectors(top, nslots=2 * nproj + 16)
    psolve = _preconditioner(top, var, run_cycles, V)
    dt = np.complex128 if V.is_complex else np.float64
    a_names = [f'a{i}' for i in range(nproj)]             # projections of one inner step
    c_names = [f'c{i}' for i in range(nproj)]             # host coefficients of a linear combination
    for nm in a_names + c_names:
        V.slot(nm)
    eps = np.finfo(np.float64).eps
    CU = []
    r = V.new()

    top.apply_A(x, r)
    V.step(r, [(b, 1.0), (r, -1.0)], dots=[('bb', b, b), ('rr', r, r)])
    bb, rr = V.read('bb', 'rr')
    b_norm = float(np.sqrt(abs(bb)))
    if b_norm == 0:
        V.copy(x, b)
        return 0
    beta_tol = max(1e-30, var.tol * b_norm)

    def fgmres(v0, ml, atol, cs):
        vs, zs = [v0], []
        B = np.zeros((len(cs), ml), dtype=dt)
        Q, R = np.ones((1, 1), dtype=dt), np.zeros((1, 0), dtype=dt)
        breakdown = False
        for j in range(ml):
            z, w = V.new(), V.new()
            psolve(vs[-1], z)
            top.apply_A(z, w)
            basis = cs + vs
            # w_norm, then w -= (q_i . w) q_i for the c's, then the v's, one after the other
            V.step(None, dots=[('wn', w, w), (a_names[0], basis[0], w)])
            for i in range(1, len(basis)):
                V.step(w, [(w, 1.0), (basis[i - 1], (a_names[i - 1], -1.0))], dots=[(a_names[i], basis[i], w)])
            V.step(w, [(w, 1.0), (basis[-1], (a_names[len(basis) - 1], -1.0))], dots=[('hh', w, w)])
            got = V.read('wn', 'hh', *a_names[:len(basis)])
            w_norm, hlast, alphas = float(np.sqrt(abs(got[0]))), float(np.sqrt(abs(got[1]))), got[2:]
            B[:, j] = alphas[:len(cs)]
            hcur = np.zeros(j + 2, dtype=dt)
            hcur[:j + 1] = alphas[len(cs):]
            hcur[j + 1] = hlast
            with np.errstate(over='ignore', divide='ignore'):
                alpha = 1 / hlast
            if np.isfinite(alpha):
                V.step(w, [(w, float(alpha))])
            if not (hlast > eps * w_norm):
                breakdown = True
            vs.append(w)
            zs.append(z)
            Q2 = np.zeros((j + 2, j + 2), dtype=dt, order='F')
            Q2[:j + 1, :j + 1] = Q
            Q2[j + 1, j + 1] = 1
            R2 = np.zeros((j + 2, j), dtype=dt, order='F')
            R2[:j + 1, :] = R
            Q, R = qr_insert(Q2, R2, hcur, j, which='col', overwrite_qru=True, check_finite=False)
            res = abs(Q[0, -1])
            if res < atol or breakdown:
                break
        if not np.isfinite(R[j, j]):
            raise LinAlgError()
        y, _, _, _ = lstsq(R[:j + 1, :j + 1], Q[0, :j + 1].conj())
        return Q, R, B[:, :j + 1], vs, zs, y

    j_outer = -1
    for j_outer in range(var.ssl_maxit):
        sumsq = top.residual_sumsq(x, b)                 # the reference's callback: |b - A x|
        (rr,), l2 = V.read('rr', extra=sumsq)
        callback(float(np.sqrt(l2[0])))
        beta = float(np.sqrt(abs(rr)))
        if beta <= beta_tol and (j_outer > 0 or CU):
            top.apply_A(x, r)
            V.step(r, [(b, 1.0), (r, -1.0)], dots=[('rr', r, r)])
            beta = float(np.sqrt(abs(V.read('rr')[0])))
        if beta <= beta_tol:
            j_outer = -1
            break
        ml = m + max(k - len(CU), 0)
        cs = [c for c, _ in CU]
        v0 = V.new()
        V.step(v0, [(r, 1.0 / beta)])
        try:
            Q, R, B, vs, zs, y = fgmres(v0, ml, beta_tol / beta, cs)
        except LinAlgError:
            break
        y = y * beta
        # ux = Z y - U (B y);  cx = V (Q R y)
        by = B.dot(y)
        ux, cx = V.new(), V.new()
        coef = list(y) + [-v for v in by]
        V.put(c_names[:len(coef)], coef)
        _combine(V, ux, zs[:len(y)] + [u for _, u in CU], c_names[:len(coef)])
        with np.errstate(invalid='ignore'):
            hy = Q.dot(R.dot(y))
        V.put(c_names[:len(hy)], list(hy))
        _combine(V, cx, vs[:len(hy)], c_names[:len(hy)])
        V.step(None, dots=[('cc', cx, cx)])
        cc = abs(V.read('cc')[0])
        with np.errstate(divide='ignore', invalid='ignore'):
            alpha = 1.0 / np.sqrt(cc)
        if not np.isfinite(alpha):
            continue
        V.step(cx, [(cx, float(alpha))])
        V.step(ux, [(ux, float(alpha))], dots=[('gamma', cx, r)])
        V.step(r, [(r, 1.0), (cx, ('gamma', -1.0))], dots=[('rr', r, r)])     # r -= gamma cx
        V.step(x, [(x, 1.0), (ux, 'gamma')])                                  # x += gamma ux
        while len(CU) >= k and CU:                       # truncate='oldest'
            del CU[0]
        CU.append((cx, ux))
        del vs, zs
    return j_outer + 1


def bicgstab_batch(hier, var_cycle, vars_, live, precondition, callback):
    """BiCGSTAB for the ``top.batch`` right-hand sides of a batched hierarchy. Every source runs
    the iteration of ``bicgstab`` above on its own rows with its own scalar table (its own
    breakdown tests and stopping rule); what the sources share are the two preconditioner
    applications (``precondition(SRC, OUT, live)``: multigrid cycles on all right-hand sides at
    once; it clears ``live[b]`` for sources whose preconditioner failed) and the two operator
    applications per iteration. ``live``: which sources iterate (updated in place).
    Returns (X, codes): the solutions stacked like the fields, and the SciPy status per source."""
    top = hier.top
    nb, n = top.batch, top.grid.n_edges
    rows = lambda t, b: t[b * n:(b + 1) * n]           # noqa: E731
    W = [Vectors(top, n) for _ in range(nb)]
    full = Vectors(top)
    B, X = full.new(), full.new()
    full.copy(B, top.s)
    _lib.check(_lib.lib().emg3d_dev_zero(_ptr(X), X.numel() * X.element_size(), _stream()), 'emg3d_dev_zero')
    R, V, T, P, PHAT, SHAT, RT = (full.new() for _ in range(7))
    eps = np.finfo(np.float64).eps
    rhotol = omegatol = eps ** 2
    maxit = var_cycle.ssl_maxit
    code = [maxit] * nb

    top.apply_A(X, R)
    atol, rr, rho, omega = [0.0] * nb, [0.0] * nb, [0.0] * nb, [1.0] * nb
    for b in range(nb):
        if not live[b]:
            continue
        W[b].step(rows(R, b), [(rows(B, b), 1.0), (rows(R, b), -1.0)], dots=[('bb', rows(B, b), rows(B, b))])
        W[b].copy(rows(RT, b), rows(R, b))
        W[b].step(None, dots=[('rr', rows(R, b), rows(R, b)), ('rho', rows(RT, b), rows(R, b))])
    for b in range(nb):
        if live[b]:
            bb, rr[b], rho[b] = W[b].read('bb', 'rr', 'rho')
            atol[b] = max(1e-30, vars_[b].tol * np.sqrt(abs(bb)))
    for iteration in range(maxit):
        for b in range(nb):                              # ---- up to the first preconditioner call
            if not live[b]:
                continue
            r, p, v = rows(R, b), rows(P, b), rows(V, b)
            if np.sqrt(abs(rr[b])) < atol[b]:
                code[b], live[b] = 0, False
            elif abs(rho[b]) < rhotol:
                code[b], live[b] = -10, False
            elif iteration > 0 and abs(omega[b]) < omegatol:
                code[b], live[b] = -11, False
            elif iteration > 0:
                W[b].step(p, [(r, 1.0), (p, 'beta'), (v, 'nbo')])
            else:
                W[b].copy(p, r)
        if not any(live):
            break
        precondition(P, PHAT, live)
        top.apply_A(PHAT, V)
        for b in range(nb):                              # ---- between the two calls
            if live[b]:
                r, v = rows(R, b), rows(V, b)
                W[b].step(None, dots=[('rv', rows(RT, b), v)],
                          prog=[(DIV, 'alpha', 'rho', 'rv'), (NEG, 'nalpha', 'alpha', None)])
                W[b].step(r, [(r, 1.0), (v, 'nalpha')], dots=[('ss', r, r)])
        for b in range(nb):
            if not live[b]:
                continue
            rv, ss = W[b].read('rv', 'ss')
            if rv == 0:
                code[b], live[b] = -11, False
            elif np.sqrt(abs(ss)) < atol[b]:
                W[b].step(rows(X, b), [(rows(X, b), 1.0), (rows(PHAT, b), 'alpha')])
                code[b], live[b] = 0, False
        if not any(live):
            break
        precondition(R, SHAT, live)
        top.apply_A(SHAT, T)
        for b in range(nb):                              # ---- after the second call
            if live[b]:
                r, t, x = rows(R, b), rows(T, b), rows(X, b)
                W[b].step(None, dots=[('ts', t, r), ('tt', t, t)],
                          prog=[(DIV, 'omega', 'ts', 'tt'), (NEG, 'nomega', 'omega', None)])
                W[b].step(x, [(x, 1.0), (rows(PHAT, b), 'alpha'), (rows(SHAT, b), 'omega')])
                W[b].step(r, [(r, 1.0), (t, 'nomega')], dots=[('rr', r, r), ('rho_next', rows(RT, b), r)],
                          prog=[(DIV, 'q1', 'rho_next', 'rho'), (DIV, 'q2', 'alpha', 'omega'), (MUL, 'beta', 'q1', 'q2'),
                                (MUL, 'bo', 'beta', 'omega'), (NEG, 'nbo', 'bo', None), (COPY, 'rho', 'rho_next', None)])
        sumsq = top.residual_sumsq(X, B)                 # all right-hand sides in one launch
        l2 = np.sqrt(sumsq[:nb].cpu().numpy())
        for b in range(nb):
            if live[b]:
                rr[b], rho[b], omega[b] = W[b].read('rr', 'rho', 'omega')
                callback(b, float(l2[b]))
    return X, code

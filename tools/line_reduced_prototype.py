"""Numerical experiment (CPU, numpy): the line system with the along-line edges eliminated analytically.

The line system of core.gauss_seidel_x/_y/_z (emg3d/core.py:632-772) couples, per node j of the line,
the four transverse edges t_j and, per cell k, the along-line edge E_k. E_k has no neighbour of its
own kind: E_k = beta_k (r0_k - b_k . (t_{k+1} - t_k)), beta_k = 1 / M_k(0,0). Substituting it leaves a
block-tridiagonal system in the t_j alone -- 4 x 4 blocks, couplings C_j = diag(c_j) + beta_j b_j b_j^T
-- whose two-sided explicit-inverse chain would need 240 instead of 304 B per block and pass and ~35 %
fewer instructions per chain step. Before any kernel is written: is it as accurate as
the present scheme (explicit inverses of the 5 x 5 Schur complements of the reference's blocks)?

Blocks come from the kernels' own assembly (tests/emu: stencil.h line_matrix / line_rhs compiled for the
host); reference solution: dense LU in extended precision (numpy longdouble).

    python tools/line_reduced_prototype.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from emu import emu                      # noqa: E402
from oracle import mg_ref                # noqa: E402


def assemble(dg, mid, l0, ld):
    """Dense matrix of the line system in the reference's unknown order (5 n0 - 4)."""
    n0 = dg.shape[0]
    n = 5 * n0 - 4
    A = np.zeros((n, n), complex)
    for k in range(n0):
        rows = 5 if k < n0 - 1 else 1
        o = 5 * k
        for r in range(rows):
            A[o + r, o + r] = dg[k, r]
            for m in range(r):
                A[o + r, o + m] = A[o + m, o + r] = mid[k, r, m]
        if k > 0:
            p = 5 * (k - 1)
            for m in range(1, 5):
                A[o, p + m] = A[p + m, o] = l0[k, m]
                if rows == 5:
                    A[o + m, p + m] = A[p + m, o + m] = ld[k, m]
    return A


def solve_extended(A, b):
    """Gaussian elimination with partial pivoting in numpy longdouble (the reference solution)."""
    A = A.astype(np.clongdouble)
    x = b.astype(np.clongdouble)
    n = A.shape[0]
    for i in range(n):
        hi = min(n, i + 12)                       # half-bandwidth 5 (+ pivoting fill)
        p = i + int(np.argmax(np.abs(A[i:hi, i])))
        if p != i:
            A[[i, p]] = A[[p, i]]
            x[[i, p]] = x[[p, i]]
        f = A[i + 1:hi, i] / A[i, i]
        A[i + 1:hi, i:min(n, i + 24)] -= np.outer(f, A[i, i:min(n, i + 24)])
        x[i + 1:hi] -= f * x[i]
    for i in range(n - 1, -1, -1):
        x[i] = (x[i] - A[i, i + 1:min(n, i + 24)] @ x[i + 1:min(n, i + 24)]) / A[i, i]
    return x


def blocks5(dg, mid, l0, ld, rhs):
    n0 = dg.shape[0]
    M, B = [], []
    for k in range(n0):
        m = np.diag(dg[k]).astype(complex)
        for r in range(5):
            for c in range(r):
                m[r, c] = m[c, r] = mid[k, r, c]
        b = np.zeros((5, 5))
        b[0, 1:] = l0[k, 1:]
        b[np.arange(1, 5), np.arange(1, 5)] = ld[k, 1:]
        if k == n0 - 1:        # last block: the along-line edge only -> identity on the missing unknowns
            m[1:, :] = 0; m[:, 1:] = 0
            m[np.arange(1, 5), np.arange(1, 5)] = 1.0
            b[1:, :] = 0
        M.append(m); B.append(b)
    r = rhs.copy()
    r[-1, 1:] = 0
    return M, B, r


def present_one_sided(dg, mid, l0, ld, rhs, inverse=True):
    """Top-down block elimination of the reference's blocks; explicit inverses T_k (the kernels'
    arithmetic) or LU substitution."""
    M, B, r = blocks5(dg, mid, l0, ld, rhs)
    n0 = len(M)
    T, w = [], []
    for k in range(n0):
        S = M[k] - (B[k] @ T[k - 1] @ B[k].T if k else 0)
        c = r[k] - (B[k] @ w[k - 1] if k else 0)
        if inverse:
            T.append(np.linalg.inv(S)); w.append(T[k] @ c)
        else:
            T.append(np.linalg.inv(S)); w.append(np.linalg.solve(S, c))     # (T only feeds the Schur update)
    T_solve = (lambda k, v: T[k] @ v) if inverse else None
    x = [None] * n0
    x[-1] = w[-1]
    Ssave = None
    for k in range(n0 - 2, -1, -1):
        q = B[k + 1].T @ x[k + 1]
        if inverse:
            x[k] = w[k] - T_solve(k, q)
        else:
            S = M[k] - (B[k] @ T[k - 1] @ B[k].T if k else 0)
            x[k] = w[k] - np.linalg.solve(S, q)
    return np.concatenate([x[k][:5 if k < n0 - 1 else 1] for k in range(n0)])


def reduced(dg, mid, l0, ld, rhs, two_sided=True, inverse=True):
    """The t-only system: C_j = diag(c_j) + beta_j b_j b_j^T couples t_j and t_{j+1} (j = 1 .. n0-2)."""
    n0 = dg.shape[0]
    beta = 1.0 / dg[:, 0]
    b = np.zeros((n0, 4))
    b[:n0 - 1] = mid[:n0 - 1, 1:, 0]
    b[n0 - 1] = -l0[n0 - 1, 1:]
    assert n0 < 3 or np.allclose(-l0[1:n0 - 1, 1:], mid[1:n0 - 1, 1:, 0], rtol=1e-15, atol=0)
    r0 = rhs[:, 0]
    nt = n0 - 1                                      # nodes j = 1 .. n0-1 -> index j-1
    N, g, C = [], [], []
    for j in range(1, n0):
        k = j - 1                                    # block that holds t_j
        m = np.diag(dg[k, 1:]).astype(complex)
        for r in range(1, 5):
            for c in range(1, r):
                m[r - 1, c - 1] = m[c - 1, r - 1] = mid[k, r, c]
        m -= beta[j - 1] * np.outer(b[j - 1], b[j - 1]) + beta[j] * np.outer(b[j], b[j])
        N.append(m)
        g.append(rhs[k, 1:] - beta[j - 1] * r0[j - 1] * b[j - 1] + beta[j] * r0[j] * b[j])
    for j in range(1, n0 - 1):                       # C_j couples t_j, t_{j+1}: c_j = ld of block j
        C.append(np.diag(ld[j, 1:]).astype(complex) + beta[j] * np.outer(b[j], b[j]))

    def apply(S, v):
        return np.linalg.inv(S) @ v if inverse else np.linalg.solve(S, v)
    t = [None] * nt
    if not two_sided:
        S, w = [], []
        for i in range(nt):
            S.append(N[i] - (C[i - 1] @ np.linalg.inv(S[i - 1]) @ C[i - 1] if i else 0))
            w.append(apply(S[i], g[i] - (C[i - 1] @ w[i - 1] if i else 0)))
        t[-1] = w[-1]
        for i in range(nt - 2, -1, -1):
            t[i] = w[i] - apply(S[i], C[i] @ t[i + 1])
    else:
        mid_i = nt // 2
        St, wt = {}, {}
        for i in range(mid_i):                       # top chain
            St[i] = N[i] - (C[i - 1] @ np.linalg.inv(St[i - 1]) @ C[i - 1] if i else 0)
            wt[i] = apply(St[i], g[i] - (C[i - 1] @ wt[i - 1] if i else 0))
        Sb, wb = {}, {}
        for i in range(nt - 1, mid_i, -1):           # bottom chain
            Sb[i] = N[i] - (C[i] @ np.linalg.inv(Sb[i + 1]) @ C[i] if i < nt - 1 else 0)
            wb[i] = apply(Sb[i], g[i] - (C[i] @ wb[i + 1] if i < nt - 1 else 0))
        Sm = N[mid_i].copy()
        gm = g[mid_i].copy()
        if mid_i > 0:
            Sm -= C[mid_i - 1] @ np.linalg.inv(St[mid_i - 1]) @ C[mid_i - 1]
            gm -= C[mid_i - 1] @ wt[mid_i - 1]
        if mid_i < nt - 1:
            Sm -= C[mid_i] @ np.linalg.inv(Sb[mid_i + 1]) @ C[mid_i]
            gm -= C[mid_i] @ wb[mid_i + 1]
        t[mid_i] = apply(Sm, gm)
        for i in range(mid_i - 1, -1, -1):
            t[i] = wt[i] - apply(St[i], C[i] @ t[i + 1])
        for i in range(mid_i + 1, nt):
            t[i] = wb[i] - apply(Sb[i], C[i - 1] @ t[i - 1])
    tz = [np.zeros(4, complex)] + t + [np.zeros(4, complex)]       # t_0 = t_{n0} = 0
    E = [beta[k] * (r0[k] - b[k] @ (tz[k + 1] - tz[k])) for k in range(n0)]
    out = []
    for k in range(n0):
        out.append([E[k]])
        if k < n0 - 1:
            out.append(tz[k + 1])
    return np.concatenate(out)


def case(n0, freq, air, stretch, seed=0):
    rng = np.random.default_rng(seed)
    shape = (n0, 6, 6)
    h = [25. * stretch ** np.abs(np.arange(n) - n // 2) for n in shape]
    grid = mg_ref.Grid(h, (0., 0., 0.))
    sig = 10 ** rng.uniform(-1, 0.5, shape)
    if air:
        sig[:, :, 3:] = 1e-8
    vm = mg_ref.volume_model(grid, freq, sig, sig / 1.5, sig / 2.5)
    s, e = mg_ref.Field(grid), mg_ref.Field(grid)
    for f in (s, e):
        f.field[:] = rng.standard_normal(f.field.size) + 1j * rng.standard_normal(f.field.size)
    return emu.line_blocks(e, s, vm, 0, 3, 3)


def main():
    print(f"{'case':44s} {'scheme':46s} {'rel. error':>11s} {'rel. residual':>14s}")
    for n0, freq, air, stretch in ((64, 1.0, False, 1.0), (64, 0.01, False, 1.03), (128, 1.0, True, 1.03),
                                   (128, 0.05, True, 1.05), (256, 0.01, False, 1.02), (256, 1.0, True, 1.0)):
        dg, mid, l0, ld, rhs = case(n0, freq, air, stretch)
        A = assemble(dg, mid, l0, ld)
        bvec = np.concatenate([rhs[k][:5 if k < n0 - 1 else 1] for k in range(n0)])
        xr = solve_extended(A, bvec)
        tag = f"n0={n0} f={freq} Hz air={air} stretch={stretch}"
        for name, fn in (("present: 5x5 blocks, one-sided, explicit inverses", lambda: present_one_sided(dg, mid, l0, ld, rhs, True)),
                         ("present blocks, LU substitution", lambda: present_one_sided(dg, mid, l0, ld, rhs, False)),
                         ("reduced 4x4, one-sided, explicit inverses", lambda: reduced(dg, mid, l0, ld, rhs, False, True)),
                         ("reduced 4x4, two-sided, explicit inverses", lambda: reduced(dg, mid, l0, ld, rhs, True, True)),
                         ("reduced 4x4, two-sided, LU substitution", lambda: reduced(dg, mid, l0, ld, rhs, True, False))):
            x = fn()
            err = float(np.linalg.norm((x - xr).astype(complex)) / np.linalg.norm(xr.astype(complex)))
            res = float(np.linalg.norm((bvec - A.astype(np.clongdouble) @ x.astype(np.clongdouble)).astype(complex)) /
                        np.linalg.norm(bvec))
            print(f"{tag:44s} {name:46s} {err:11.2e} {res:14.2e}")
        print(f"{'':44s} {'numpy fp64 dense solve':46s} "
              f"{float(np.linalg.norm(np.linalg.solve(A, bvec) - xr.astype(complex)) / np.linalg.norm(xr.astype(complex))):11.2e}")


if __name__ == '__main__':
    main()

/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the emg3d_amd product path).
 *
 * Type-generic CPU restatement of the nine numba kernels of the reference
 * (emg3d/core.py), in the reference's exact sequential order. Included twice by
 * core_oracle.c: once with T = double (Laplace domain) and once with
 * T = double complex (frequency domain).
 *
 *   FN(name)   -> name##_d / name##_z
 *   T          -> field / eta / amat / bvec scalar type
 *
 * Layout (reference emg3d/fields.py:201-259, emg3d/meshes.py:101-103): Fortran
 * order, x fastest;  ex (nx,ny+1,nz+1), ey (nx+1,ny,nz+1), ez (nx+1,ny+1,nz),
 * eta_x,eta_y,eta_z and zeta (nx,ny,nz).
 *
 * `order` argument of the smoothers (an addition of the oracle; the reference
 * only has order 0):
 *   0  lexicographic, exactly the reference's loop nests
 *   1  four-colour ordering used by the HIP kernels:
 *      point:  colour = ((ix+iz)&1) | (((iy+iz)&1)<<1)
 *      lines:  colour = (a&1) | ((b&1)<<1), (a,b) the two transverse node indices
 *              x-line: (iy,iz); y-line: (ix,iz); z-line: (ix,iy)
 *      point smoother: every sweep visits the node colours 0,2,3,1 (oracle_colour_order;
 *      oracle_point_repeat = 0: a backward sweep visits them in reverse, the rule of rounds
 *      1-2); line smoothers: the passes of a call cycle through
 *      1,2,3,0,1,... (oracle_line_cycle; sweep `it` takes positions 3 it .. 3 it + 3), or follow
 *      the point smoother's mirrored rule (oracle_set_line_order(0, ...)). Inside a colour the
 *      nodes/lines are independent, so any order gives the same result.
 */

#define EX(a, i, j, k) (a)[(size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)(ny + 1) * (size_t)(k))]
#define EY(a, i, j, k) (a)[(size_t)(i) + (size_t)(nx + 1) * ((size_t)(j) + (size_t)ny * (size_t)(k))]
#define EZ(a, i, j, k) (a)[(size_t)(i) + (size_t)(nx + 1) * ((size_t)(j) + (size_t)(ny + 1) * (size_t)(k))]
/* in-pass over-relaxation (experiments only; oracle_omega = 1: plain assignment, the reference) */
#ifndef RELAX
#define RELAX(dst, val) do { if (oracle_omega == 1.0) (dst) = (val); else (dst) = (dst) + oracle_omega * ((val) - (dst)); } while (0)
#endif
#ifndef ORACLE_COLOUR
/* colour class visited at position cc of a forward (iback = 0) / backward sweep: backward = the forward
 * sequence reversed, unless an experiment set its own backward sequence (oracle_set_colour_order_backward) */
#define ORACLE_MIRRORED(iback, cc) ((iback) ? oracle_colour_order[3 - (cc)] : oracle_colour_order[cc])
#define ORACLE_COLOUR(iback, cc) ((iback) ? (oracle_backward_custom ? oracle_colour_order_b[cc] : (oracle_point_repeat ? oracle_colour_order[cc] : oracle_colour_order[3 - (cc)])) : oracle_colour_order[cc])
#endif
#ifndef ORACLE_LINE_COLOUR
/* LINE smoothers, order 1: by default the passes of a call cycle through oracle_line_cycle (1,2,3,0,1,...),
 * sweep `it` taking positions 3 it .. 3 it + 3 -- the order of the HIP kernels since round 3 (emg3d_amd/
 * csrc/launch.h: line_sweep_colour); oracle_line_cyclic = 0: the mirrored rule of ORACLE_COLOUR. */
#define ORACLE_LINE_COLOUR(it, iback, cc) (oracle_line_cyclic == 2 ? oracle_line_cycle[(cc) & 3] : oracle_line_cyclic ? oracle_line_cycle[(3 * (it) + (cc)) & 3] : (oracle_backward_custom ? ORACLE_COLOUR(iback, cc) : ORACLE_MIRRORED(iback, cc)))
#endif
#define CC(a, i, j, k) (a)[(size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)ny * (size_t)(k))]

/* ------------------------------------------------------------------------- */
/* core.solve -- reference emg3d/core.py:1481-1616                            */
/* Non-pivoting LDL^T of a complex-symmetric band matrix (5 sub-diagonals),   */
/* A(i,j) -> amat[i+5j]; solution overwrites bvec; diagonal replaced by 1/D.  */
/* ------------------------------------------------------------------------- */
void FN(solve)(T *amat, T *bvec, int n)
{
    T h, d;
    int i, j, k;

    /* core.py:1561-1565 */
    d = 1.0 / amat[0];
    for (i = 1; i < (n < 6 ? n : 6); i++) amat[i] *= d;

    /* core.py:1568-1587 */
    for (j = 1; j < n; j++) {
        h = 0.0;
        for (k = (j - 5 > 0 ? j - 5 : 0); k < j; k++)
            h += amat[j + 5 * k] * amat[j + 5 * k] * amat[6 * k];
        amat[6 * j] -= h;
        d = 1.0 / amat[6 * j];
        for (i = j + 1; i < (n < j + 6 ? n : j + 6); i++) {
            h = 0.0;
            for (k = (i - 5 > 0 ? i - 5 : 0); k < j; k++)
                h += amat[i + 5 * k] * amat[j + 5 * k] * amat[6 * k];
            amat[i + 5 * j] -= h;
            amat[i + 5 * j] *= d;
        }
    }

    /* core.py:1589-1592 */
    amat[6 * (n - 1)] = d;
    for (j = n - 2; j >= 0; j--) amat[6 * j] = 1.0 / amat[6 * j];

    /* core.py:1597-1603 forward substitution */
    for (j = 1; j < n; j++) {
        h = 0.0;
        for (k = (j - 5 > 0 ? j - 5 : 0); k < j; k++) h += amat[j + 5 * k] * bvec[k];
        bvec[j] -= h;
    }
    /* core.py:1606-1607 */
    for (j = 0; j < n; j++) bvec[j] *= amat[6 * j];
    /* core.py:1610-1616 backward substitution */
    for (j = n - 2; j >= 0; j--) {
        h = 0.0;
        for (k = j + 1; k < (n < j + 6 ? n : j + 6); k++) h += amat[k + 5 * j] * bvec[k];
        bvec[j] -= h;
    }
}

/* ------------------------------------------------------------------------- */
/* core.blocks_to_amat -- reference emg3d/core.py:1351-1477                   */
/* `left` is float64 in the reference (core.py:587); `middle`,`rhs` are T.    */
/* ------------------------------------------------------------------------- */
void FN(blocks_to_amat)(T *amat, T *bvec, const T *middle, const double *left, const T *rhs,
                        int im, int nc)
{
    int fam = 5 * im, mam = fam - 5, k, m;

    if (im == 0) { /* core.py:1440-1449 */
        for (k = 0; k < 5; k++) bvec[k] = rhs[k];
        for (k = 0; k < 5; k++)
            for (m = 0; m <= k; m++) amat[k + 5 * m] = middle[k + 5 * m];
    } else if (im <= nc - 2 && nc > 2) { /* core.py:1451-1465 */
        for (k = 0; k < 5; k++) bvec[k + fam] = rhs[k];
        for (m = 1; m < 5; m++)
            for (k = 0; k <= m; k++) amat[k + fam + 5 * (m + mam)] = left[k + 5 * m];
        for (k = 0; k < 5; k++)
            for (m = 0; m <= k; m++) amat[k + fam + 5 * (m + fam)] = middle[k + 5 * m];
    } else if (im == nc - 1) { /* core.py:1467-1477 */
        bvec[fam] = rhs[0];
        for (m = 1; m < 5; m++) amat[fam + 5 * (m + mam)] = left[5 * m];
        amat[6 * fam] = middle[0];
    }
}

/* ------------------------------------------------------------------------- */
/* core.amat_x -- reference emg3d/core.py:57-206:  r -= A e                   */
/* ------------------------------------------------------------------------- */
void FN(amat_x)(T *rx, T *ry, T *rz, const T *ex, const T *ey, const T *ez, const T *eta_x,
                const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                const double *hy, const double *hz, int nx, int ny, int nz)
{
    int ix, iy, iz;
    for (iz = 0; iz < nz; iz++) {
        int izm = iz > 0 ? iz - 1 : 0, izp = iz + 1;
        for (iy = 0; iy < ny; iy++) {
            int iym = iy > 0 ? iy - 1 : 0, iyp = iy + 1;
            for (ix = 0; ix < nx; ix++) {
                int ixm = ix > 0 ? ix - 1 : 0, ixp = ix + 1;
                T v1pp, v1mp, v1pm, v2pp, v2mp, v2pm, v3pp, v3mp, v3pm, rrx, rry, rrz, stx, sty, stz;

                /* 1. curl (core.py:136-155) */
                v1pp = ((EZ(ez, ix, iyp, iz) - EZ(ez, ix, iy, iz)) / hy[iy] -
                        (EY(ey, ix, iy, izp) - EY(ey, ix, iy, iz)) / hz[iz]);
                v1mp = ((EZ(ez, ix, iy, iz) - EZ(ez, ix, iym, iz)) / hy[iym] -
                        (EY(ey, ix, iym, izp) - EY(ey, ix, iym, iz)) / hz[iz]);
                v1pm = ((EZ(ez, ix, iyp, izm) - EZ(ez, ix, iy, izm)) / hy[iy] -
                        (EY(ey, ix, iy, iz) - EY(ey, ix, iy, izm)) / hz[izm]);

                v2pp = ((EX(ex, ix, iy, izp) - EX(ex, ix, iy, iz)) / hz[iz] -
                        (EZ(ez, ixp, iy, iz) - EZ(ez, ix, iy, iz)) / hx[ix]);
                v2mp = ((EX(ex, ixm, iy, izp) - EX(ex, ixm, iy, iz)) / hz[iz] -
                        (EZ(ez, ix, iy, iz) - EZ(ez, ixm, iy, iz)) / hx[ixm]);
                v2pm = ((EX(ex, ix, iy, iz) - EX(ex, ix, iy, izm)) / hz[izm] -
                        (EZ(ez, ixp, iy, izm) - EZ(ez, ix, iy, izm)) / hx[ix]);

                v3pp = ((EY(ey, ixp, iy, iz) - EY(ey, ix, iy, iz)) / hx[ix] -
                        (EX(ex, ix, iyp, iz) - EX(ex, ix, iy, iz)) / hy[iy]);
                v3mp = ((EY(ey, ix, iy, iz) - EY(ey, ixm, iy, iz)) / hx[ixm] -
                        (EX(ex, ixm, iyp, iz) - EX(ex, ixm, iy, iz)) / hy[iy]);
                v3pm = ((EY(ey, ixp, iym, iz) - EY(ey, ix, iym, iz)) / hx[ix] -
                        (EX(ex, ix, iy, iz) - EX(ex, ix, iym, iz)) / hy[iym]);

                /* 2. times face-average of zeta (core.py:160-170) */
                v1pp *= CC(zeta, ixm, iy, iz) + CC(zeta, ix, iy, iz);
                v1mp *= CC(zeta, ixm, iym, iz) + CC(zeta, ix, iym, iz);
                v1pm *= CC(zeta, ixm, iy, izm) + CC(zeta, ix, iy, izm);

                v2pp *= CC(zeta, ix, iym, iz) + CC(zeta, ix, iy, iz);
                v2mp *= CC(zeta, ixm, iym, iz) + CC(zeta, ixm, iy, iz);
                v2pm *= CC(zeta, ix, iym, izm) + CC(zeta, ix, iy, izm);

                v3pp *= CC(zeta, ix, iy, izm) + CC(zeta, ix, iy, iz);
                v3mp *= CC(zeta, ixm, iy, izm) + CC(zeta, ixm, iy, iz);
                v3pm *= CC(zeta, ix, iym, izm) + CC(zeta, ix, iym, iz);

                /* 3. second curl (core.py:174-176) */
                rrx = v3pp / hy[iy] - v3pm / hy[iym] - v2pp / hz[iz] + v2pm / hz[izm];
                rry = v1pp / hz[iz] - v1pm / hz[izm] - v3pp / hx[ix] + v3mp / hx[ixm];
                rrz = v2pp / hx[ix] - v2mp / hx[ixm] - v1pp / hy[iy] + v1mp / hy[iym];

                /* 4. eta edge-averages (core.py:181-186) */
                stx = (CC(eta_x, ix, iym, izm) + CC(eta_x, ix, iym, iz) + CC(eta_x, ix, iy, izm) +
                       CC(eta_x, ix, iy, iz));
                sty = (CC(eta_y, ixm, iy, izm) + CC(eta_y, ix, iy, izm) + CC(eta_y, ixm, iy, iz) +
                       CC(eta_y, ix, iy, iz));
                stz = (CC(eta_z, ixm, iym, iz) + CC(eta_z, ix, iym, iz) + CC(eta_z, ixm, iy, iz) +
                       CC(eta_z, ix, iy, iz));

                /* PEC rows (core.py:193-198) */
                if (iy == 0 || iz == 0) rrx = 0;
                if (ix == 0 || iz == 0) rry = 0;
                if (ix == 0 || iy == 0) rrz = 0;

                /* 5. update (core.py:204-206) */
                EX(rx, ix, iy, iz) -= 0.5 * rrx - 0.25 * stx * EX(ex, ix, iy, iz);
                EY(ry, ix, iy, iz) -= 0.5 * rry - 0.25 * sty * EY(ey, ix, iy, iz);
                EZ(rz, ix, iy, iz) -= 0.5 * rrz - 0.25 * stz * EZ(ez, ix, iy, iz);
            }
        }
    }
}

/* The 24 zeta face-averages around node (ix,iy,iz) (core.py:351-374). Naming as
 * the reference: m{a}{b}{L|R}{c}{m|p}. For the line kernels the reference leaves
 * four of them commented out; computing all of them does not change any result
 * (indices stay in range because the callers clamp ix/iy/iz exactly as the
 * reference does). */
#define ORACLE_M_COEFFS                                                                  \
    double mzyLxm = ky[iym] * (CC(zeta, ixm, iym, iz) + CC(zeta, ixm, iym, izm));         \
    double mzyRxm = ky[iy] * (CC(zeta, ixm, iy, iz) + CC(zeta, ixm, iy, izm));            \
    double myzLxm = kz[izm] * (CC(zeta, ixm, iy, izm) + CC(zeta, ixm, iym, izm));         \
    double myzRxm = kz[iz] * (CC(zeta, ixm, iy, iz) + CC(zeta, ixm, iym, iz));            \
    double mzyLxp = ky[iym] * (CC(zeta, ix, iym, iz) + CC(zeta, ix, iym, izm));           \
    double mzyRxp = ky[iy] * (CC(zeta, ix, iy, iz) + CC(zeta, ix, iy, izm));              \
    double myzLxp = kz[izm] * (CC(zeta, ix, iy, izm) + CC(zeta, ix, iym, izm));           \
    double myzRxp = kz[iz] * (CC(zeta, ix, iy, iz) + CC(zeta, ix, iym, iz));              \
    double mzxLym = kx[ixm] * (CC(zeta, ixm, iym, iz) + CC(zeta, ixm, iym, izm));         \
    double mzxRym = kx[ix] * (CC(zeta, ix, iym, iz) + CC(zeta, ix, iym, izm));            \
    double mxzLym = kz[izm] * (CC(zeta, ix, iym, izm) + CC(zeta, ixm, iym, izm));         \
    double mxzRym = kz[iz] * (CC(zeta, ix, iym, iz) + CC(zeta, ixm, iym, iz));            \
    double mzxLyp = kx[ixm] * (CC(zeta, ixm, iy, iz) + CC(zeta, ixm, iy, izm));           \
    double mzxRyp = kx[ix] * (CC(zeta, ix, iy, iz) + CC(zeta, ix, iy, izm));              \
    double mxzLyp = kz[izm] * (CC(zeta, ix, iy, izm) + CC(zeta, ixm, iy, izm));           \
    double mxzRyp = kz[iz] * (CC(zeta, ix, iy, iz) + CC(zeta, ixm, iy, iz));              \
    double myxLzm = kx[ixm] * (CC(zeta, ixm, iy, izm) + CC(zeta, ixm, iym, izm));         \
    double myxRzm = kx[ix] * (CC(zeta, ix, iy, izm) + CC(zeta, ix, iym, izm));            \
    double mxyLzm = ky[iym] * (CC(zeta, ix, iym, izm) + CC(zeta, ixm, iym, izm));         \
    double mxyRzm = ky[iy] * (CC(zeta, ix, iy, izm) + CC(zeta, ixm, iy, izm));            \
    double myxLzp = kx[ixm] * (CC(zeta, ixm, iy, iz) + CC(zeta, ixm, iym, iz));           \
    double myxRzp = kx[ix] * (CC(zeta, ix, iy, iz) + CC(zeta, ix, iym, iz));              \
    double mxyLzp = ky[iym] * (CC(zeta, ix, iym, iz) + CC(zeta, ixm, iym, iz));           \
    double mxyRzp = ky[iy] * (CC(zeta, ix, iy, iz) + CC(zeta, ixm, iy, iz));

/* The six eta edge-sums around a node (core.py:377-388). */
#define ORACLE_ST_SUMS                                                                    \
    T st0 = (CC(eta_x, ixm, iy, iz) + CC(eta_x, ixm, iy, izm) + CC(eta_x, ixm, iym, iz) +  \
             CC(eta_x, ixm, iym, izm));                                                   \
    T st1 = (CC(eta_x, ix, iy, iz) + CC(eta_x, ix, iy, izm) + CC(eta_x, ix, iym, iz) +     \
             CC(eta_x, ix, iym, izm));                                                    \
    T st2 = (CC(eta_y, ix, iym, iz) + CC(eta_y, ix, iym, izm) + CC(eta_y, ixm, iym, iz) +  \
             CC(eta_y, ixm, iym, izm));                                                   \
    T st3 = (CC(eta_y, ix, iy, iz) + CC(eta_y, ix, iy, izm) + CC(eta_y, ixm, iy, iz) +     \
             CC(eta_y, ixm, iy, izm));                                                    \
    T st4 = (CC(eta_z, ix, iy, izm) + CC(eta_z, ix, iym, izm) + CC(eta_z, ixm, iy, izm) +  \
             CC(eta_z, ixm, iym, izm));                                                   \
    T st5 = (CC(eta_z, ix, iy, iz) + CC(eta_z, ix, iym, iz) + CC(eta_z, ixm, iy, iz) +     \
             CC(eta_z, ixm, iym, iz));

/* One node update of the point smoother -- core.py:346-503. */
static void FN(gs_node)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz, const T *eta_x,
                        const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                        const double *hy, const double *hz, const double *kx, const double *ky,
                        const double *kz, int nx, int ny, int nz, int ix, int iy, int iz)
{
    int ixm = ix - 1, ixp = ix + 1, iym = iy - 1, iyp = iy + 1, izm = iz - 1, izp = iz + 1, k;
    T amat[36], rhs[6], st[6];
    (void)nz;
    ORACLE_M_COEFFS
    ORACLE_ST_SUMS

    st[0] = st0 / 4.; st[1] = st1 / 4.; st[2] = st2 / 4.;
    st[3] = st3 / 4.; st[4] = st4 / 4.; st[5] = st5 / 4.;

    for (k = 0; k < 36; k++) amat[k] = 0.;
    for (k = 0; k < 6; k++) amat[6 * k] = -st[k];

    /* diagonals (core.py:401-412) */
    amat[0] += mzyRxm / hy[iy] + mzyLxm / hy[iym];
    amat[0] += myzRxm / hz[iz] + myzLxm / hz[izm];
    amat[6] += mzyRxp / hy[iy] + mzyLxp / hy[iym];
    amat[6] += myzRxp / hz[iz] + myzLxp / hz[izm];
    amat[12] += mzxRym / hx[ix] + mzxLym / hx[ixm];
    amat[12] += mxzRym / hz[iz] + mxzLym / hz[izm];
    amat[18] += mzxRyp / hx[ix] + mzxLyp / hx[ixm];
    amat[18] += mxzRyp / hz[iz] + mxzLyp / hz[izm];
    amat[24] += myxRzm / hx[ix] + myxLzm / hx[ixm];
    amat[24] += mxyRzm / hy[iy] + mxyLzm / hy[iym];
    amat[30] += myxRzp / hx[ix] + myxLzp / hx[ixm];
    amat[30] += mxyRzp / hy[iy] + mxyLzp / hy[iym];

    /* off-diagonals (core.py:419-430) */
    amat[2] = -mzyLxm / hx[ixm];
    amat[3] = mzyRxm / hx[ixm];
    amat[4] = -myzLxm / hx[ixm];
    amat[5] = myzRxm / hx[ixm];
    amat[7] = mzyLxp / hx[ix];
    amat[8] = -mzyRxp / hx[ix];
    amat[9] = myzLxp / hx[ix];
    amat[10] = -myzRxp / hx[ix];
    amat[14] = -mxzLym / hy[iym];
    amat[15] = mxzRym / hy[iym];
    amat[19] = mxzLyp / hy[iy];
    amat[20] = -mxzRyp / hy[iy];

    /* rhs (core.py:436-492) */
    rhs[0] = EX(sx, ixm, iy, iz); rhs[1] = EX(sx, ix, iy, iz);
    rhs[2] = EY(sy, ix, iym, iz); rhs[3] = EY(sy, ix, iy, iz);
    rhs[4] = EZ(sz, ix, iy, izm); rhs[5] = EZ(sz, ix, iy, iz);

    rhs[0] += mzyRxm * (EY(ey, ixm, iy, iz) / hx[ixm] + EX(ex, ixm, iyp, iz) / hy[iy]);
    rhs[0] += mzyLxm * (-EY(ey, ixm, iym, iz) / hx[ixm] + EX(ex, ixm, iym, iz) / hy[iym]);
    rhs[0] += myzRxm * (EZ(ez, ixm, iy, iz) / hx[ixm] + EX(ex, ixm, iy, izp) / hz[iz]);
    rhs[0] += myzLxm * (-EZ(ez, ixm, iy, izm) / hx[ixm] + EX(ex, ixm, iy, izm) / hz[izm]);

    rhs[1] += mzyRxp * (-EY(ey, ixp, iy, iz) / hx[ix] + EX(ex, ix, iyp, iz) / hy[iy]);
    rhs[1] += mzyLxp * (EY(ey, ixp, iym, iz) / hx[ix] + EX(ex, ix, iym, iz) / hy[iym]);
    rhs[1] += myzRxp * (-EZ(ez, ixp, iy, iz) / hx[ix] + EX(ex, ix, iy, izp) / hz[iz]);
    rhs[1] += myzLxp * (EZ(ez, ixp, iy, izm) / hx[ix] + EX(ex, ix, iy, izm) / hz[izm]);

    rhs[2] += mzxRym * (EY(ey, ixp, iym, iz) / hx[ix] + EX(ex, ix, iym, iz) / hy[iym]);
    rhs[2] += mzxLym * (EY(ey, ixm, iym, iz) / hx[ixm] - EX(ex, ixm, iym, iz) / hy[iym]);
    rhs[2] += mxzRym * (EZ(ez, ix, iym, iz) / hy[iym] + EY(ey, ix, iym, izp) / hz[iz]);
    rhs[2] += mxzLym * (-EZ(ez, ix, iym, izm) / hy[iym] + EY(ey, ix, iym, izm) / hz[izm]);

    rhs[3] += mzxRyp * (EY(ey, ixp, iy, iz) / hx[ix] - EX(ex, ix, iyp, iz) / hy[iy]);
    rhs[3] += mzxLyp * (EY(ey, ixm, iy, iz) / hx[ixm] + EX(ex, ixm, iyp, iz) / hy[iy]);
    rhs[3] += mxzRyp * (-EZ(ez, ix, iyp, iz) / hy[iy] + EY(ey, ix, iy, izp) / hz[iz]);
    rhs[3] += mxzLyp * (EZ(ez, ix, iyp, izm) / hy[iy] + EY(ey, ix, iy, izm) / hz[izm]);

    rhs[4] += myxRzm * (EZ(ez, ixp, iy, izm) / hx[ix] + EX(ex, ix, iy, izm) / hz[izm]);
    rhs[4] += myxLzm * (EZ(ez, ixm, iy, izm) / hx[ixm] - EX(ex, ixm, iy, izm) / hz[izm]);
    rhs[4] += mxyRzm * (EZ(ez, ix, iyp, izm) / hy[iy] + EY(ey, ix, iy, izm) / hz[izm]);
    rhs[4] += mxyLzm * (EZ(ez, ix, iym, izm) / hy[iym] - EY(ey, ix, iym, izm) / hz[izm]);

    rhs[5] += myxRzp * (EZ(ez, ixp, iy, iz) / hx[ix] - EX(ex, ix, iy, izp) / hz[iz]);
    rhs[5] += myxLzp * (EZ(ez, ixm, iy, iz) / hx[ixm] + EX(ex, ixm, iy, izp) / hz[iz]);
    rhs[5] += mxyRzp * (EZ(ez, ix, iyp, iz) / hy[iy] - EY(ey, ix, iy, izp) / hz[iz]);
    rhs[5] += mxyLzp * (EZ(ez, ix, iym, iz) / hy[iym] + EY(ey, ix, iym, izp) / hz[iz]);

    FN(solve)(amat, rhs, 6);

    /* core.py:498-503 */
    RELAX(EX(ex, ixm, iy, iz), rhs[0]);
    RELAX(EX(ex, ix, iy, iz), rhs[1]);
    RELAX(EY(ey, ix, iym, iz), rhs[2]);
    RELAX(EY(ey, ix, iy, iz), rhs[3]);
    RELAX(EZ(ez, ix, iy, izm), rhs[4]);
    RELAX(EZ(ez, ix, iy, izm + 1), rhs[5]);
}

/* core.gauss_seidel -- reference emg3d/core.py:210-503 */
void FN(gauss_seidel)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz, const T *eta_x,
                      const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                      const double *hy, const double *hz, int nx, int ny, int nz, int nu, int order)
{
    double *kx = (double *)malloc(sizeof(double) * (size_t)(nx + ny + nz));
    double *ky = kx + nx, *kz = ky + ny;
    int i, it, iback = 0, ixh, iyh, izh, c, cc;
    for (i = 0; i < nx; i++) kx[i] = 0.5 / hx[i];
    for (i = 0; i < ny; i++) ky[i] = 0.5 / hy[i];
    for (i = 0; i < nz; i++) kz[i] = 0.5 / hz[i];

    for (it = 0; it < nu; it++) {
        iback = 1 - iback; /* first sweep is BACKWARD (core.py:301,311) */
        if (order == 0) {
            for (izh = 1; izh < nz; izh++) {
                int iz = iback ? nz - izh : izh;
                for (iyh = 1; iyh < ny; iyh++) {
                    int iy = iback ? ny - iyh : iyh;
                    for (ixh = 1; ixh < nx; ixh++) {
                        int ix = iback ? nx - ixh : ixh;
                        FN(gs_node)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz,
                                    kx, ky, kz, nx, ny, nz, ix, iy, iz);
                    }
                }
            }
        } else if (order == 1) {
            for (cc = 0; cc < 4; cc++) {
                c = ORACLE_COLOUR(iback, cc);
                /* (nodes of one colour are independent: threads change nothing, oracle_set_threads) */
#pragma omp parallel num_threads(oracle_threads) if (oracle_threads > 1)
                {
                    int q1, q2, q3;
#pragma omp for collapse(2) schedule(static)
                    for (q3 = 1; q3 < nz; q3++)
                        for (q2 = 1; q2 < ny; q2++)
                            for (q1 = 1; q1 < nx; q1++)
                                if ((((q1 + q3) & 1) | (((q2 + q3) & 1) << 1)) == c)
                                    FN(gs_node)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx,
                                                hy, hz, kx, ky, kz, nx, ny, nz, q1, q2, q3);
                }
            }
        } else {
            /* order 2: tiles of oracle_tile[0..2] nodes, eight tile colours
             * (tx&1)|((ty&1)<<1)|((tz&1)<<2) visited in oracle_tile_order (backward:
             * reversed); inside a tile the four node colours of order 1. */
            const int bx = oracle_tile[0], by = oracle_tile[1], bz = oracle_tile[2];
            const int ntx = (nx - 2) / bx + 1, nty = (ny - 2) / by + 1, ntz = (nz - 2) / bz + 1;
            int t8;
            for (t8 = 0; t8 < 8; t8++) {
                const int tc = oracle_tile_order[(iback && !oracle_tile_repeat) ? 7 - t8 : t8];
                const int tz0 = (tc >> 2) & 1, ty0 = (tc >> 1) & 1, tx0 = tc & 1;
                /* (tiles of one tile colour share no edge that either reads or writes: independent) */
#pragma omp parallel num_threads(oracle_threads) if (oracle_threads > 1)
                {
                    int tx, ty, tz, q1, q2, q3, c2, cq;
#pragma omp for collapse(3) schedule(static)
                    for (tz = tz0; tz < ntz; tz += 2)
                        for (ty = ty0; ty < nty; ty += 2)
                            for (tx = tx0; tx < ntx; tx += 2)
                                for (cq = 0; cq < 4; cq++) {
                                    c2 = ORACLE_COLOUR(iback, cq);
                                    for (q3 = 1 + tz * bz; q3 < nz && q3 < 1 + (tz + 1) * bz; q3++)
                                        for (q2 = 1 + ty * by; q2 < ny && q2 < 1 + (ty + 1) * by; q2++)
                                            for (q1 = 1 + tx * bx; q1 < nx && q1 < 1 + (tx + 1) * bx; q1++)
                                                if ((((q1 + q3) & 1) | (((q2 + q3) & 1) << 1)) == c2)
                                                    FN(gs_node)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z,
                                                                zeta, hx, hy, hz, kx, ky, kz, nx, ny, nz,
                                                                q1, q2, q3);
                                }
                }
            }
        }
    }
    free(kx);
}

/* One x-line: assemble (core.py:632-769), solve (:772), scatter (:775-783). */
static void FN(gs_line_x)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz,
                          const T *eta_x, const T *eta_y, const T *eta_z, const double *zeta,
                          const double *hx, const double *hy, const double *hz, const double *kx,
                          const double *ky, const double *kz, int nx, int ny, int nz, int iy, int iz,
                          T *amat, T *bvec)
{
    int nr = 5 * nx - 4, ixh, k;
    int iym = iy - 1, iyp = iy + 1, izm = iz - 1, izp = iz + 1;
    T middle[25], rhs[5], st[5];
    double left[25];
    (void)nz;
    for (k = 0; k < 25; k++) { middle[k] = 0.; left[k] = 0.; }
    for (k = 0; k < nr; k++) bvec[k] = 0.;
    for (k = 0; k < 6 * nr; k++) amat[k] = 0.;

    for (ixh = 1; ixh < nx + 1; ixh++) {
        int ix = ixh < nx - 1 ? ixh : nx - 1; /* core.py:635 */
        int ixm = ixh - 1;
        ORACLE_M_COEFFS
        ORACLE_ST_SUMS
        (void)mzyLxp; (void)mzyRxp; (void)myzLxp; (void)myzRxp; (void)st1;

        st[0] = st0 / 4.; st[1] = st2 / 4.; st[2] = st3 / 4.; st[3] = st4 / 4.; st[4] = st5 / 4.;
        for (k = 0; k < 5; k++) middle[6 * k] = -st[k];

        middle[0] += mzyRxm / hy[iy] + mzyLxm / hy[iym];
        middle[0] += myzRxm / hz[iz] + myzLxm / hz[izm];
        middle[6] += mzxRym / hx[ix] + mzxLym / hx[ixm];
        middle[6] += mxzRym / hz[iz] + mxzLym / hz[izm];
        middle[12] += mzxRyp / hx[ix] + mzxLyp / hx[ixm];
        middle[12] += mxzRyp / hz[iz] + mxzLyp / hz[izm];
        middle[18] += myxRzm / hx[ix] + myxLzm / hx[ixm];
        middle[18] += mxyRzm / hy[iy] + mxyLzm / hy[iym];
        middle[24] += myxRzp / hx[ix] + myxLzp / hx[ixm];
        middle[24] += mxyRzp / hy[iy] + mxyLzp / hy[iym];

        middle[1] = -mzyLxm / hx[ixm];
        middle[2] = mzyRxm / hx[ixm];
        middle[3] = -myzLxm / hx[ixm];
        middle[4] = myzRxm / hx[ixm];
        middle[8] = -mxzLym / hy[iym];
        middle[9] = mxzRym / hy[iym];
        middle[13] = mxzLyp / hy[iy];
        middle[14] = -mxzRyp / hy[iy];

        left[5] = mzyLxm / hx[ixm];
        left[10] = -mzyRxm / hx[ixm];
        left[15] = myzLxm / hx[ixm];
        left[20] = -myzRxm / hx[ixm];
        left[6] = -mzxLym / hx[ixm];
        left[12] = -mzxLyp / hx[ixm];
        left[18] = -myxLzm / hx[ixm];
        left[24] = -myxLzp / hx[ixm];

        rhs[0] = EX(sx, ixm, iy, iz);
        rhs[1] = EY(sy, ix, iym, iz);
        rhs[2] = EY(sy, ix, iy, iz);
        rhs[3] = EZ(sz, ix, iy, izm);
        rhs[4] = EZ(sz, ix, iy, iz);

        rhs[0] += mzyRxm * EX(ex, ixm, iyp, iz) / hy[iy];
        rhs[0] += mzyLxm * EX(ex, ixm, iym, iz) / hy[iym];
        rhs[0] += myzRxm * EX(ex, ixm, iy, izp) / hz[iz];
        rhs[0] += myzLxm * EX(ex, ixm, iy, izm) / hz[izm];

        rhs[1] += (mzxRym * EX(ex, ix, iym, iz) - mzxLym * EX(ex, ixm, iym, iz) +
                   mxzRym * EZ(ez, ix, iym, iz) - mxzLym * EZ(ez, ix, iym, izm)) / hy[iym];
        rhs[1] += mxzRym * EY(ey, ix, iym, izp) / hz[iz];
        rhs[1] += mxzLym * EY(ey, ix, iym, izm) / hz[izm];

        rhs[2] += (mzxLyp * EX(ex, ixm, iyp, iz) - mzxRyp * EX(ex, ix, iyp, iz) +
                   mxzLyp * EZ(ez, ix, iyp, izm) - mxzRyp * EZ(ez, ix, iyp, iz)) / hy[iy];
        rhs[2] += mxzRyp * EY(ey, ix, iy, izp) / hz[iz];
        rhs[2] += mxzLyp * EY(ey, ix, iy, izm) / hz[izm];

        rhs[3] += (myxRzm * EX(ex, ix, iy, izm) - myxLzm * EX(ex, ixm, iy, izm) +
                   mxyRzm * EY(ey, ix, iy, izm) - mxyLzm * EY(ey, ix, iym, izm)) / hz[izm];
        rhs[3] += mxyRzm * EZ(ez, ix, iyp, izm) / hy[iy];
        rhs[3] += mxyLzm * EZ(ez, ix, iym, izm) / hy[iym];

        rhs[4] += (myxLzp * EX(ex, ixm, iy, izp) - myxRzp * EX(ex, ix, iy, izp) +
                   mxyLzp * EY(ey, ix, iym, izp) - mxyRzp * EY(ey, ix, iy, izp)) / hz[iz];
        rhs[4] += mxyRzp * EZ(ez, ix, iyp, iz) / hy[iy];
        rhs[4] += mxyLzp * EZ(ez, ix, iym, iz) / hy[iym];

        FN(blocks_to_amat)(amat, bvec, middle, left, rhs, ixm, nx);
    }

    FN(solve)(amat, bvec, nr);

    for (ixh = 1; ixh < nx + 1; ixh++) {
        int ixm = ixh - 1;
        RELAX(EX(ex, ixm, iy, iz), bvec[5 * ixm]);
        if (ixm < nx - 1) {
            RELAX(EY(ey, ixh, iym, iz), bvec[1 + 5 * ixm]);
            RELAX(EY(ey, ixh, iy, iz), bvec[2 + 5 * ixm]);
            RELAX(EZ(ez, ixh, iy, izm), bvec[3 + 5 * ixm]);
            RELAX(EZ(ez, ixh, iy, iz), bvec[4 + 5 * ixm]);
        }
    }
}

/* core.gauss_seidel_x -- reference emg3d/core.py:506-783 */
void FN(gauss_seidel_x)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz, const T *eta_x,
                        const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                        const double *hy, const double *hz, int nx, int ny, int nz, int nu, int order)
{
    int nr = 5 * nx - 4;
    double *kx = (double *)malloc(sizeof(double) * (size_t)(nx + ny + nz));
    double *ky = kx + nx, *kz = ky + ny;
    T *bvec = (T *)malloc(sizeof(T) * (size_t)nr * 7);
    T *amat = bvec + nr;
    int i, it, iback = 0, iyh, izh, c, cc;
    for (i = 0; i < nx; i++) kx[i] = 0.5 / hx[i];
    for (i = 0; i < ny; i++) ky[i] = 0.5 / hy[i];
    for (i = 0; i < nz; i++) kz[i] = 0.5 / hz[i];

    for (it = 0; it < nu; it++) {
        iback = 1 - iback;
        if (order == 0) {
            for (izh = 1; izh < nz; izh++) {
                int iz = iback ? nz - izh : izh;
                for (iyh = 1; iyh < ny; iyh++) {
                    int iy = iback ? ny - iyh : iyh;
                    FN(gs_line_x)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, kx,
                                  ky, kz, nx, ny, nz, iy, iz, amat, bvec);
                }
            }
        } else {
            /* the lines of a colour class are independent: any order, any number of threads gives the
             * same values bit by bit (oracle_set_threads; full-size parity tests) */
            for (cc = 0; cc < 4; cc++) {
                c = ORACLE_LINE_COLOUR(it, iback, cc);
#pragma omp parallel num_threads(oracle_threads) if (oracle_threads > 1)
                {
                    T *bv = oracle_threads > 1 ? (T *)malloc(sizeof(T) * (size_t)nr * 7) : bvec;
                    T *am = bv + nr;
                    int q1, q2;
#pragma omp for collapse(2) schedule(static)
                    for (q2 = 1; q2 < nz; q2++)
                        for (q1 = 1; q1 < ny; q1++)
                            if (((q1 & 1) | ((q2 & 1) << 1)) == c)
                                FN(gs_line_x)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                                              hz, kx, ky, kz, nx, ny, nz, q1, q2, am, bv);
                    if (oracle_threads > 1) free(bv);
                }
            }
        }
    }
    free(bvec);
    free(kx);
}

/* One y-line: core.py:917-1068. Unknown order per block
 * [ey(iym); ex(ixm,iy), ex(ix,iy); ez(izm), ez(iz)]. */
static void FN(gs_line_y)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz,
                          const T *eta_x, const T *eta_y, const T *eta_z, const double *zeta,
                          const double *hx, const double *hy, const double *hz, const double *kx,
                          const double *ky, const double *kz, int nx, int ny, int nz, int ix, int iz,
                          T *amat, T *bvec)
{
    int nr = 5 * ny - 4, iyh, k;
    int ixm = ix - 1, ixp = ix + 1, izm = iz - 1, izp = iz + 1;
    T middle[25], rhs[5], st[5];
    double left[25];
    (void)nz;
    for (k = 0; k < 25; k++) { middle[k] = 0.; left[k] = 0.; }
    for (k = 0; k < nr; k++) bvec[k] = 0.;
    for (k = 0; k < 6 * nr; k++) amat[k] = 0.;

    for (iyh = 1; iyh < ny + 1; iyh++) {
        int iy = iyh < ny - 1 ? iyh : ny - 1; /* core.py:920 */
        int iym = iyh - 1;
        ORACLE_M_COEFFS
        ORACLE_ST_SUMS
        (void)mzxLyp; (void)mzxRyp; (void)mxzLyp; (void)mxzRyp; (void)st3;

        st[0] = st2 / 4.; st[1] = st0 / 4.; st[2] = st1 / 4.; st[3] = st4 / 4.; st[4] = st5 / 4.;
        for (k = 0; k < 5; k++) middle[6 * k] = -st[k];

        middle[0] += mzxRym / hx[ix] + mzxLym / hx[ixm];
        middle[0] += mxzRym / hz[iz] + mxzLym / hz[izm];
        middle[6] += mzyRxm / hy[iy] + mzyLxm / hy[iym];
        middle[6] += myzRxm / hz[iz] + myzLxm / hz[izm];
        middle[12] += mzyRxp / hy[iy] + mzyLxp / hy[iym];
        middle[12] += myzRxp / hz[iz] + myzLxp / hz[izm];
        middle[18] += myxRzm / hx[ix] + myxLzm / hx[ixm];
        middle[18] += mxyRzm / hy[iy] + mxyLzm / hy[iym];
        middle[24] += myxRzp / hx[ix] + myxLzp / hx[ixm];
        middle[24] += mxyRzp / hy[iy] + mxyLzp / hy[iym];

        middle[1] = -mzyLxm / hx[ixm];
        middle[2] = mzyLxp / hx[ix];
        middle[3] = -mxzLym / hy[iym];
        middle[4] = mxzRym / hy[iym];
        middle[8] = -myzLxm / hx[ixm];
        middle[9] = myzRxm / hx[ixm];
        middle[13] = myzLxp / hx[ix];
        middle[14] = -myzRxp / hx[ix];

        left[5] = mzxLym / hy[iym];
        left[10] = -mzxRym / hy[iym];
        left[15] = mxzLym / hy[iym];
        left[20] = -mxzRym / hy[iym];
        left[6] = -mzyLxm / hy[iym];
        left[12] = -mzyLxp / hy[iym];
        left[18] = -mxyLzm / hy[iym];
        left[24] = -mxyLzp / hy[iym];

        rhs[0] = EY(sy, ix, iym, iz);
        rhs[1] = EX(sx, ixm, iy, iz);
        rhs[2] = EX(sx, ix, iy, iz);
        rhs[3] = EZ(sz, ix, iy, izm);
        rhs[4] = EZ(sz, ix, iy, iz);

        rhs[0] += mzxRym * EY(ey, ixp, iym, iz) / hx[ix];
        rhs[0] += mzxLym * EY(ey, ixm, iym, iz) / hx[ixm];
        rhs[0] += mxzRym * EY(ey, ix, iym, izp) / hz[iz];
        rhs[0] += mxzLym * EY(ey, ix, iym, izm) / hz[izm];

        rhs[1] += (mzyRxm * EY(ey, ixm, iy, iz) - mzyLxm * EY(ey, ixm, iym, iz) +
                   myzRxm * EZ(ez, ixm, iy, iz) - myzLxm * EZ(ez, ixm, iy, izm)) / hx[ixm];
        rhs[1] += myzRxm * EX(ex, ixm, iy, izp) / hz[iz];
        rhs[1] += myzLxm * EX(ex, ixm, iy, izm) / hz[izm];

        rhs[2] += (mzyLxp * EY(ey, ixp, iym, iz) - mzyRxp * EY(ey, ixp, iy, iz) +
                   myzLxp * EZ(ez, ixp, iy, izm) - myzRxp * EZ(ez, ixp, iy, iz)) / hx[ix];
        rhs[2] += myzRxp * EX(ex, ix, iy, izp) / hz[iz];
        rhs[2] += myzLxp * EX(ex, ix, iy, izm) / hz[izm];

        rhs[3] += (myxRzm * EX(ex, ix, iy, izm) - myxLzm * EX(ex, ixm, iy, izm) +
                   mxyRzm * EY(ey, ix, iy, izm) - mxyLzm * EY(ey, ix, iym, izm)) / hz[izm];
        rhs[3] += myxRzm * EZ(ez, ixp, iy, izm) / hx[ix];
        rhs[3] += myxLzm * EZ(ez, ixm, iy, izm) / hx[ixm];

        rhs[4] += (myxLzp * EX(ex, ixm, iy, izp) - myxRzp * EX(ex, ix, iy, izp) +
                   mxyLzp * EY(ey, ix, iym, izp) - mxyRzp * EY(ey, ix, iy, izp)) / hz[iz];
        rhs[4] += myxRzp * EZ(ez, ixp, iy, iz) / hx[ix];
        rhs[4] += myxLzp * EZ(ez, ixm, iy, iz) / hx[ixm];

        FN(blocks_to_amat)(amat, bvec, middle, left, rhs, iym, ny);
    }

    FN(solve)(amat, bvec, nr);

    for (iyh = 1; iyh < ny + 1; iyh++) {
        int iym = iyh - 1;
        RELAX(EY(ey, ix, iym, iz), bvec[5 * iym]);
        if (iym < ny - 1) {
            RELAX(EX(ex, ixm, iyh, iz), bvec[1 + 5 * iym]);
            RELAX(EX(ex, ix, iyh, iz), bvec[2 + 5 * iym]);
            RELAX(EZ(ez, ix, iyh, izm), bvec[3 + 5 * iym]);
            RELAX(EZ(ez, ix, iyh, iz), bvec[4 + 5 * iym]);
        }
    }
}

/* core.gauss_seidel_y -- reference emg3d/core.py:786-1068 (loops iz outer, ix inner) */
void FN(gauss_seidel_y)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz, const T *eta_x,
                        const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                        const double *hy, const double *hz, int nx, int ny, int nz, int nu, int order)
{
    int nr = 5 * ny - 4;
    double *kx = (double *)malloc(sizeof(double) * (size_t)(nx + ny + nz));
    double *ky = kx + nx, *kz = ky + ny;
    T *bvec = (T *)malloc(sizeof(T) * (size_t)nr * 7);
    T *amat = bvec + nr;
    int i, it, iback = 0, ixh, izh, c, cc;
    for (i = 0; i < nx; i++) kx[i] = 0.5 / hx[i];
    for (i = 0; i < ny; i++) ky[i] = 0.5 / hy[i];
    for (i = 0; i < nz; i++) kz[i] = 0.5 / hz[i];

    for (it = 0; it < nu; it++) {
        iback = 1 - iback;
        if (order == 0) {
            for (izh = 1; izh < nz; izh++) {
                int iz = iback ? nz - izh : izh;
                for (ixh = 1; ixh < nx; ixh++) {
                    int ix = iback ? nx - ixh : ixh;
                    FN(gs_line_y)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, kx,
                                  ky, kz, nx, ny, nz, ix, iz, amat, bvec);
                }
            }
        } else {
            /* the lines of a colour class are independent: any order, any number of threads gives the
             * same values bit by bit (oracle_set_threads; full-size parity tests) */
            for (cc = 0; cc < 4; cc++) {
                c = ORACLE_LINE_COLOUR(it, iback, cc);
#pragma omp parallel num_threads(oracle_threads) if (oracle_threads > 1)
                {
                    T *bv = oracle_threads > 1 ? (T *)malloc(sizeof(T) * (size_t)nr * 7) : bvec;
                    T *am = bv + nr;
                    int q1, q2;
#pragma omp for collapse(2) schedule(static)
                    for (q2 = 1; q2 < nz; q2++)
                        for (q1 = 1; q1 < nx; q1++)
                            if (((q1 & 1) | ((q2 & 1) << 1)) == c)
                                FN(gs_line_y)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                                              hz, kx, ky, kz, nx, ny, nz, q1, q2, am, bv);
                    if (oracle_threads > 1) free(bv);
                }
            }
        }
    }
    free(bvec);
    free(kx);
}

/* One z-line: core.py:1197-1348. Unknown order per block
 * [ez(izm); ex(ixm), ex(ix); ey(iym), ey(iy)]. */
static void FN(gs_line_z)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz,
                          const T *eta_x, const T *eta_y, const T *eta_z, const double *zeta,
                          const double *hx, const double *hy, const double *hz, const double *kx,
                          const double *ky, const double *kz, int nx, int ny, int nz, int ix, int iy,
                          T *amat, T *bvec)
{
    int nr = 5 * nz - 4, izh, k;
    int ixm = ix - 1, ixp = ix + 1, iym = iy - 1, iyp = iy + 1;
    T middle[25], rhs[5], st[5];
    double left[25];
    for (k = 0; k < 25; k++) { middle[k] = 0.; left[k] = 0.; }
    for (k = 0; k < nr; k++) bvec[k] = 0.;
    for (k = 0; k < 6 * nr; k++) amat[k] = 0.;

    for (izh = 1; izh < nz + 1; izh++) {
        int iz = izh < nz - 1 ? izh : nz - 1; /* core.py:1200 */
        int izm = izh - 1;
        ORACLE_M_COEFFS
        ORACLE_ST_SUMS
        (void)myxLzp; (void)myxRzp; (void)mxyLzp; (void)mxyRzp; (void)st5;

        st[0] = st4 / 4.; st[1] = st0 / 4.; st[2] = st1 / 4.; st[3] = st2 / 4.; st[4] = st3 / 4.;
        for (k = 0; k < 5; k++) middle[6 * k] = -st[k];

        middle[0] += myxRzm / hx[ix] + myxLzm / hx[ixm];
        middle[0] += mxyRzm / hy[iy] + mxyLzm / hy[iym];
        middle[6] += mzyRxm / hy[iy] + mzyLxm / hy[iym];
        middle[6] += myzRxm / hz[iz] + myzLxm / hz[izm];
        middle[12] += mzyRxp / hy[iy] + mzyLxp / hy[iym];
        middle[12] += myzRxp / hz[iz] + myzLxp / hz[izm];
        middle[18] += mzxRym / hx[ix] + mzxLym / hx[ixm];
        middle[18] += mxzRym / hz[iz] + mxzLym / hz[izm];
        middle[24] += mzxRyp / hx[ix] + mzxLyp / hx[ixm];
        middle[24] += mxzRyp / hz[iz] + mxzLyp / hz[izm];

        middle[1] = -myzLxm / hx[ixm];
        middle[2] = myzLxp / hx[ix];
        middle[3] = -mxzLym / hy[iym];
        middle[4] = mxzLyp / hy[iy];
        middle[8] = -mzyLxm / hx[ixm];
        middle[9] = mzyRxm / hx[ixm];
        middle[13] = mzyLxp / hx[ix];
        middle[14] = -mzyRxp / hx[ix];

        left[5] = myxLzm / hz[izm];
        left[10] = -myxRzm / hz[izm];
        left[15] = mxyLzm / hz[izm];
        left[20] = -mxyRzm / hz[izm];
        left[6] = -myzLxm / hz[izm];
        left[12] = -myzLxp / hz[izm];
        left[18] = -mxzLym / hz[izm];
        left[24] = -mxzLyp / hz[izm];

        rhs[0] = EZ(sz, ix, iy, izm);
        rhs[1] = EX(sx, ixm, iy, iz);
        rhs[2] = EX(sx, ix, iy, iz);
        rhs[3] = EY(sy, ix, iym, iz);
        rhs[4] = EY(sy, ix, iy, iz);

        rhs[0] += myxRzm * (EZ(ez, ixp, iy, izm) / hx[ix]);
        rhs[0] += myxLzm * (EZ(ez, ixm, iy, izm) / hx[ixm]);
        rhs[0] += mxyRzm * (EZ(ez, ix, iyp, izm) / hy[iy]);
        rhs[0] += mxyLzm * (EZ(ez, ix, iym, izm) / hy[iym]);

        rhs[1] += (mzyRxm * EY(ey, ixm, iy, iz) - mzyLxm * EY(ey, ixm, iym, iz) +
                   myzRxm * EZ(ez, ixm, iy, iz) - myzLxm * EZ(ez, ixm, iy, izm)) / hx[ixm];
        rhs[1] += mzyRxm * EX(ex, ixm, iyp, iz) / hy[iy];
        rhs[1] += mzyLxm * EX(ex, ixm, iym, iz) / hy[iym];

        rhs[2] += (mzyLxp * EY(ey, ixp, iym, iz) - mzyRxp * EY(ey, ixp, iy, iz) +
                   myzLxp * EZ(ez, ixp, iy, izm) - myzRxp * EZ(ez, ixp, iy, iz)) / hx[ix];
        rhs[2] += mzyRxp * EX(ex, ix, iyp, iz) / hy[iy];
        rhs[2] += mzyLxp * EX(ex, ix, iym, iz) / hy[iym];

        rhs[3] += (mzxRym * EX(ex, ix, iym, iz) - mzxLym * EX(ex, ixm, iym, iz) +
                   mxzRym * EZ(ez, ix, iym, iz) - mxzLym * EZ(ez, ix, iym, izm)) / hy[iym];
        rhs[3] += mzxRym * EY(ey, ixp, iym, iz) / hx[ix];
        rhs[3] += mzxLym * EY(ey, ixm, iym, iz) / hx[ixm];

        rhs[4] += (mzxLyp * EX(ex, ixm, iyp, iz) - mzxRyp * EX(ex, ix, iyp, iz) +
                   mxzLyp * EZ(ez, ix, iyp, izm) - mxzRyp * EZ(ez, ix, iyp, iz)) / hy[iy];
        rhs[4] += mzxRyp * EY(ey, ixp, iy, iz) / hx[ix];
        rhs[4] += mzxLyp * EY(ey, ixm, iy, iz) / hx[ixm];

        FN(blocks_to_amat)(amat, bvec, middle, left, rhs, izm, nz);
    }

    FN(solve)(amat, bvec, nr);

    for (izh = 1; izh < nz + 1; izh++) {
        int izm = izh - 1;
        RELAX(EZ(ez, ix, iy, izm), bvec[5 * izm]);
        if (izm < nz - 1) {
            RELAX(EX(ex, ixm, iy, izh), bvec[1 + 5 * izm]);
            RELAX(EX(ex, ix, iy, izh), bvec[2 + 5 * izm]);
            RELAX(EY(ey, ix, iym, izh), bvec[3 + 5 * izm]);
            RELAX(EY(ey, ix, iy, izh), bvec[4 + 5 * izm]);
        }
    }
}

/* core.gauss_seidel_z -- reference emg3d/core.py:1071-1348 (loops iy outer, ix inner) */
void FN(gauss_seidel_z)(T *ex, T *ey, T *ez, const T *sx, const T *sy, const T *sz, const T *eta_x,
                        const T *eta_y, const T *eta_z, const double *zeta, const double *hx,
                        const double *hy, const double *hz, int nx, int ny, int nz, int nu, int order)
{
    int nr = 5 * nz - 4;
    double *kx = (double *)malloc(sizeof(double) * (size_t)(nx + ny + nz));
    double *ky = kx + nx, *kz = ky + ny;
    T *bvec = (T *)malloc(sizeof(T) * (size_t)nr * 7);
    T *amat = bvec + nr;
    int i, it, iback = 0, ixh, iyh, c, cc;
    for (i = 0; i < nx; i++) kx[i] = 0.5 / hx[i];
    for (i = 0; i < ny; i++) ky[i] = 0.5 / hy[i];
    for (i = 0; i < nz; i++) kz[i] = 0.5 / hz[i];

    for (it = 0; it < nu; it++) {
        iback = 1 - iback;
        if (order == 0) {
            for (iyh = 1; iyh < ny; iyh++) {
                int iy = iback ? ny - iyh : iyh;
                for (ixh = 1; ixh < nx; ixh++) {
                    int ix = iback ? nx - ixh : ixh;
                    FN(gs_line_z)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy, hz, kx,
                                  ky, kz, nx, ny, nz, ix, iy, amat, bvec);
                }
            }
        } else {
            /* the lines of a colour class are independent: any order, any number of threads gives the
             * same values bit by bit (oracle_set_threads; full-size parity tests) */
            for (cc = 0; cc < 4; cc++) {
                c = ORACLE_LINE_COLOUR(it, iback, cc);
#pragma omp parallel num_threads(oracle_threads) if (oracle_threads > 1)
                {
                    T *bv = oracle_threads > 1 ? (T *)malloc(sizeof(T) * (size_t)nr * 7) : bvec;
                    T *am = bv + nr;
                    int q1, q2;
#pragma omp for collapse(2) schedule(static)
                    for (q2 = 1; q2 < ny; q2++)
                        for (q1 = 1; q1 < nx; q1++)
                            if (((q1 & 1) | ((q2 & 1) << 1)) == c)
                                FN(gs_line_z)(ex, ey, ez, sx, sy, sz, eta_x, eta_y, eta_z, zeta, hx, hy,
                                              hz, kx, ky, kz, nx, ny, nz, q1, q2, am, bv);
                    if (oracle_threads > 1) free(bv);
                }
            }
        }
    }
    free(bvec);
    free(kx);
}

#undef ORACLE_M_COEFFS
#undef ORACLE_ST_SUMS

/* ------------------------------------------------------------------------- */
/* core.restrict -- reference emg3d/core.py:1620-2001.                        */
/* cnx,cny,cnz = number of COARSE nodes; nx,ny,nz = number of FINE nodes      */
/* (core.py:1661-1664). Weight arrays are (wl, w0, wr) per direction.         */
/* The seven sc_dir branches of the reference differ only in which directions */
/* are coarsened; they are restated here with one loop nest and per-direction */
/* switches, evaluating the same sums in the same order as each branch.       */
/* ------------------------------------------------------------------------- */
void FN(restrict)(T *crx, T *cry, T *crz, const T *rx, const T *ry, const T *rz, const double *wxl,
                  const double *wx0, const double *wxr, const double *wyl, const double *wy0,
                  const double *wyr, const double *wzl, const double *wz0, const double *wzr,
                  int cnx, int cny, int cnz, int nx, int ny, int nz, int sc_dir)
{
    /* which directions are coarsened (solver.py:891-897) */
    int cx = !(sc_dir == 1 || sc_dir == 5 || sc_dir == 6);
    int cy = !(sc_dir == 2 || sc_dir == 4 || sc_dir == 6);
    int cz = !(sc_dir == 3 || sc_dir == 4 || sc_dir == 5);
    int cix, ciy, ciz, a, b;

#define FRX(i, j, k) rx[(size_t)(i) + (size_t)(nx - 1) * ((size_t)(j) + (size_t)ny * (size_t)(k))]
#define FRY(i, j, k) ry[(size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)(ny - 1) * (size_t)(k))]
#define FRZ(i, j, k) rz[(size_t)(i) + (size_t)nx * ((size_t)(j) + (size_t)ny * (size_t)(k))]
#define CRX(i, j, k) crx[(size_t)(i) + (size_t)(cnx - 1) * ((size_t)(j) + (size_t)cny * (size_t)(k))]
#define CRY(i, j, k) cry[(size_t)(i) + (size_t)cnx * ((size_t)(j) + (size_t)(cny - 1) * (size_t)(k))]
#define CRZ(i, j, k) crz[(size_t)(i) + (size_t)cnx * ((size_t)(j) + (size_t)cny * (size_t)(k))]

    for (ciz = 0; ciz < cnz; ciz++) {
        int iz = cz ? 2 * ciz : ciz;
        int izs[3]; double wzs[3]; int nzt = cz ? 3 : 1;
        /* order of the reference's terms: centre, left(minus), right(plus) */
        izs[0] = iz; izs[1] = (iz - 1 > 0 ? iz - 1 : 0); izs[2] = (iz + 1 < nz - 1 ? iz + 1 : nz - 1);
        wzs[0] = cz ? wz0[ciz] : 1.0; wzs[1] = cz ? wzl[ciz] : 0.0; wzs[2] = cz ? wzr[ciz] : 0.0;
        for (ciy = 0; ciy < cny; ciy++) {
            int iy = cy ? 2 * ciy : ciy;
            int iys[3]; double wys[3]; int nyt = cy ? 3 : 1;
            iys[0] = iy; iys[1] = (iy - 1 > 0 ? iy - 1 : 0); iys[2] = (iy + 1 < ny - 1 ? iy + 1 : ny - 1);
            wys[0] = cy ? wy0[ciy] : 1.0; wys[1] = cy ? wyl[ciy] : 0.0; wys[2] = cy ? wyr[ciy] : 0.0;
            for (cix = 0; cix < cnx; cix++) {
                int ix = cx ? 2 * cix : cix;
                int ixs[3]; double wxs[3]; int nxt = cx ? 3 : 1;
                ixs[0] = ix; ixs[1] = (ix - 1 > 0 ? ix - 1 : 0); ixs[2] = (ix + 1 < nx - 1 ? ix + 1 : nx - 1);
                wxs[0] = cx ? wx0[cix] : 1.0; wxs[1] = cx ? wxl[cix] : 0.0; wxs[2] = cx ? wxr[cix] : 0.0;

                /* x-field: pair sum along x (if coarsened), weights in y and z */
                if (cix < cnx - 1) {
                    T acc = 0.;
                    for (a = 0; a < nyt; a++) {
                        T inner = 0.;
                        for (b = 0; b < nzt; b++) {
                            T v = FRX(ix, iys[a], izs[b]);
                            if (cx) v += FRX(ixs[2], iys[a], izs[b]);
                            if (cz) inner += wzs[b] * v; else inner = v;
                        }
                        if (cy) acc += wys[a] * inner; else acc = inner;
                    }
                    CRX(cix, ciy, ciz) = acc;
                }
                /* y-field */
                if (ciy < cny - 1) {
                    T acc = 0.;
                    for (a = 0; a < nxt; a++) {
                        T inner = 0.;
                        for (b = 0; b < nzt; b++) {
                            T v = FRY(ixs[a], iy, izs[b]);
                            if (cy) v += FRY(ixs[a], iys[2], izs[b]);
                            if (cz) inner += wzs[b] * v; else inner = v;
                        }
                        if (cx) acc += wxs[a] * inner; else acc = inner;
                    }
                    CRY(cix, ciy, ciz) = acc;
                }
                /* z-field */
                if (ciz < cnz - 1) {
                    T acc = 0.;
                    for (a = 0; a < nxt; a++) {
                        T inner = 0.;
                        for (b = 0; b < nyt; b++) {
                            T v = FRZ(ixs[a], iys[b], iz);
                            if (cz) v += FRZ(ixs[a], iys[b], izs[2]);
                            if (cy) inner += wys[b] * v; else inner = v;
                        }
                        if (cx) acc += wxs[a] * inner; else acc = inner;
                    }
                    CRZ(cix, ciy, ciz) = acc;
                }
            }
        }
    }
#undef FRX
#undef FRY
#undef FRZ
#undef CRX
#undef CRY
#undef CRZ
}

#undef EX
#undef EY
#undef EZ
#undef CC

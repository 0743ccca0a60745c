// Per-node / per-line / per-edge bodies of the multigrid inner loop, written once as
// EMG_HD functions: the HIP kernels in kernels.hip call them with (blockIdx, threadIdx)
// derived indices; the CPU emulation harness in tests/emu/ (unit tests without a GPU)
// calls the very same bodies from plain loops.
//
// What each body computes follows the reference's numba kernels (emg3d/core.py, cited per
// function); how it is organised does not: nodes / lines are visited in a four-colour
// order so that all work items of one launch are independent, the
// banded line systems are factorised in a streaming row-by-row LDL^T that never
// materialises the reference's `amat`, and x-, y- and z-line relaxation share ONE body
// through the cyclic symmetry (x,y,z) -> (y,z,x) of the curl-curl operator.
//
// Layout (reference emg3d/fields.py:201-259): Fortran order, x fastest;
//   ex (nx, ny+1, nz+1), ey (nx+1, ny, nz+1), ez (nx+1, ny+1, nz); eta_*, zeta (nx,ny,nz).
#pragma once
#include "cplx.h"

namespace emg {

// One grid level as the kernels see it. All pointers are device pointers (or host
// pointers in the emulation harness). eta_x/eta_y/eta_z may alias (isotropic / VTI / HTI,
// reference emg3d/models.py:693-712) and are never written.
template <class T> struct Level {
    int nx, ny, nz;                    // cells
    T *ex, *ey, *ez;                   // electric field (updated in place by the smoothers)
    const T *sx, *sy, *sz;             // source / right-hand side
    const T *eta_x, *eta_y, *eta_z;    // -s mu0 sigma V   (field dtype)
    const double *zeta;                // V / mu_r
    const double *ihx, *ihy, *ihz;     // inverse cell widths 1/h
    // several right-hand sides that share the model (sources of one frequency): source b's
    // field and source buffers start b * bstride elements behind source 0's. One launch then
    // serves all of them (one more grid dimension): the coarse levels, whose launches are
    // latency-bound whatever they carry, cost the same for `batch` sources as for one.
    int batch = 1;
    size_t bstride = 0;
    int flags = 0;                     // emg3d_level::flags (LEVEL_ETA_IMAG)
};
// all eta values have a real part of exactly zero (diffusive approximation at a real
// frequency: eta = -i omega mu0 sigma V): the eta edge sums are then stored as 8-byte doubles
constexpr int LEVEL_ETA_IMAG = 1;
// the level solves a CORRECTION equation (its right-hand side is a residual, its result is added to a field that is
// kept in full precision elsewhere): the streamed line passes may keep their T and w records in single precision
// (kernels.hip: line_compact_used, k_line_stream<.., COMPACT>)
constexpr int LEVEL_LINE_COMPACT = 2;
// ... and the tiled point smoother may keep its eta edge sums (coefficients of its 6 x 6 systems) in single precision
constexpr int LEVEL_POINT_COMPACT = 4;
template <class T> EMG_HD Level<T> source_level(Level<T> L, int b)
{
    size_t o = (size_t)b * L.bstride;
#if defined(__HIP_DEVICE_COMPILE__)
    // the offset is uniform over the workgroup: keep it (and the shifted pointers) in scalar registers
    o = ((size_t)__builtin_amdgcn_readfirstlane((unsigned)(o >> 32)) << 32) | (size_t)__builtin_amdgcn_readfirstlane((unsigned)o);
#endif
    L.ex += o; L.ey += o; L.ez += o;
    L.sx += o; L.sy += o; L.sz += o;
    return L;
}

// ---------------------------------------------------------------------------------------
// Axis-permuted accessors. DIR = 0,1,2 selects the "line" axis a0 = x,y,z; (a1,a2) follow
// cyclically: DIR 0: (x,y,z)  DIR 1: (y,z,x)  DIR 2: (z,x,y). Abstract indices (i0,i1,i2)
// are indices along (a0,a1,a2). E0/E1/E2 are the field components along a0/a1/a2.
// ---------------------------------------------------------------------------------------
template <class T, int DIR> struct Axes {
    const Level<T> &L;
    // element offset of the right-hand side this thread works on (Level::batch): added to the
    // field / source pointers AFTER the component is selected. (Shifting the pointers inside a
    // copy of the Level makes the compiler index that copy in scratch memory.)
    size_t boff;
    EMG_HD Axes(const Level<T> &l, size_t off = 0) : L(l), boff(off) {}

    EMG_HD int n0() const { return DIR == 0 ? L.nx : DIR == 1 ? L.ny : L.nz; }
    EMG_HD int n1() const { return DIR == 0 ? L.ny : DIR == 1 ? L.nz : L.nx; }
    EMG_HD int n2() const { return DIR == 0 ? L.nz : DIR == 1 ? L.nx : L.ny; }
    EMG_HD const double *ih0() const { return DIR == 0 ? L.ihx : DIR == 1 ? L.ihy : L.ihz; }
    EMG_HD const double *ih1() const { return DIR == 0 ? L.ihy : DIR == 1 ? L.ihz : L.ihx; }
    EMG_HD const double *ih2() const { return DIR == 0 ? L.ihz : DIR == 1 ? L.ihx : L.ihy; }

    // physical (ix,iy,iz) from abstract (i0,i1,i2)
    EMG_HD int px(int i0, int i1, int i2) const { return DIR == 0 ? i0 : DIR == 1 ? i2 : i1; }
    EMG_HD int py(int i0, int i1, int i2) const { return DIR == 0 ? i1 : DIR == 1 ? i0 : i2; }
    EMG_HD int pz(int i0, int i1, int i2) const { return DIR == 0 ? i2 : DIR == 1 ? i1 : i0; }

    EMG_HD int iex(int x, int y, int z) const { return x + L.nx * (y + (L.ny + 1) * z); }
    EMG_HD int iey(int x, int y, int z) const { return x + (L.nx + 1) * (y + L.ny * z); }
    EMG_HD int iez(int x, int y, int z) const { return x + (L.nx + 1) * (y + (L.ny + 1) * z); }
    EMG_HD int icc(int x, int y, int z) const { return x + L.nx * (y + L.ny * z); }

    // linear index of component c (0: along a0, 1: a1, 2: a2) at abstract position
    EMG_HD int idx(int c, int i0, int i1, int i2) const
    {
        const int x = px(i0, i1, i2), y = py(i0, i1, i2), z = pz(i0, i1, i2);
        const int phys = (c + DIR) % 3;  // physical component 0:x 1:y 2:z
        return phys == 0 ? iex(x, y, z) : phys == 1 ? iey(x, y, z) : iez(x, y, z);
    }
    EMG_HD T *E(int c) const
    {
        const int phys = (c + DIR) % 3;
        return (phys == 0 ? L.ex : phys == 1 ? L.ey : L.ez) + boff;
    }
    EMG_HD const T *S(int c) const
    {
        const int phys = (c + DIR) % 3;
        return (phys == 0 ? L.sx : phys == 1 ? L.sy : L.sz) + boff;
    }
    EMG_HD const T *ETA(int c) const
    {
        const int phys = (c + DIR) % 3;
        return phys == 0 ? L.eta_x : phys == 1 ? L.eta_y : L.eta_z;
    }
    EMG_HD T e(int c, int i0, int i1, int i2) const { return E(c)[idx(c, i0, i1, i2)]; }
    EMG_HD T s(int c, int i0, int i1, int i2) const { return S(c)[idx(c, i0, i1, i2)]; }
    EMG_HD T eta(int c, int i0, int i1, int i2) const
    {
        return ETA(c)[icc(px(i0, i1, i2), py(i0, i1, i2), pz(i0, i1, i2))];
    }
    EMG_HD double zeta(int i0, int i1, int i2) const
    {
        return L.zeta[icc(px(i0, i1, i2), py(i0, i1, i2), pz(i0, i1, i2))];
    }
};

// The operands of the residual of extended cell (ix,iy,iz): 24 field values, 8 zeta, 12 eta. Split from the
// arithmetic (residual_compute) so that a thread that walks a column of cells upwards can CARRY the values it
// will need again one plane higher (residual_load_roll) instead of loading them a second time: 28 instead of 53
// loads per cell, the same operands, the same arithmetic, the same bits.
template <class T> struct ResIn {
    T ex_c, ey_c, ez_c;
    T ex_zp, exm_zp, exm_c, ex_zm, ex_yp, exm_yp, ex_ym;        // EX(ix,iy,izp) (ixm,iy,izp) (ixm,iy,iz) (ix,iy,izm) (ix,iyp,iz) (ixm,iyp,iz) (ix,iym,iz)
    T ey_zp, eym_zp, eym_c, ey_zm, ey_xp, ey_xm, eym_xp;        // EY(ix,iy,izp) (ix,iym,izp) (ix,iym,iz) (ix,iy,izm) (ixp,iy,iz) (ixm,iy,iz) (ixp,iym,iz)
    T ez_yp, ez_ym, ez_xp, ez_xm, ez_yp_zm, ez_zm, ez_xp_zm;    // EZ(ix,iyp,iz) (ix,iym,iz) (ixp,iy,iz) (ixm,iy,iz) (ix,iyp,izm) (ix,iy,izm) (ixp,iy,izm)
    double z000, z100, z010, z110, z001, z101, z011, z111;
    T etx[4], ety[4], etz[4];                                   // in the order of the sums of core.py:181-186
    double hx1, hx0, hy1, hy0, hz1, hz0;
};

// all operands from memory
template <class T> EMG_HD void residual_load(const Level<T> &L, int ix, int iy, int iz, ResIn<T> &in)
{
    const Axes<T, 0> A(L);
    const int ixm = ix > 0 ? ix - 1 : 0, iym = iy > 0 ? iy - 1 : 0, izm = iz > 0 ? iz - 1 : 0;
    const int ixp = ix + 1, iyp = iy + 1, izp = iz + 1;
    in.hx1 = L.ihx[ix]; in.hx0 = L.ihx[ixm];
    in.hy1 = L.ihy[iy]; in.hy0 = L.ihy[iym];
    in.hz1 = L.ihz[iz]; in.hz0 = L.ihz[izm];
#define EXv(i, j, k) L.ex[A.iex(i, j, k)]
#define EYv(i, j, k) L.ey[A.iey(i, j, k)]
#define EZv(i, j, k) L.ez[A.iez(i, j, k)]
#define ZT(i, j, k) L.zeta[A.icc(i, j, k)]
#define ETv(p, i, j, k) (p)[A.icc(i, j, k)]
    in.ex_c = EXv(ix, iy, iz); in.ey_c = EYv(ix, iy, iz); in.ez_c = EZv(ix, iy, iz);
    in.ex_zp = EXv(ix, iy, izp); in.exm_zp = EXv(ixm, iy, izp); in.exm_c = EXv(ixm, iy, iz); in.ex_zm = EXv(ix, iy, izm);
    in.ex_yp = EXv(ix, iyp, iz); in.exm_yp = EXv(ixm, iyp, iz); in.ex_ym = EXv(ix, iym, iz);
    in.ey_zp = EYv(ix, iy, izp); in.eym_zp = EYv(ix, iym, izp); in.eym_c = EYv(ix, iym, iz); in.ey_zm = EYv(ix, iy, izm);
    in.ey_xp = EYv(ixp, iy, iz); in.ey_xm = EYv(ixm, iy, iz); in.eym_xp = EYv(ixp, iym, iz);
    in.ez_yp = EZv(ix, iyp, iz); in.ez_ym = EZv(ix, iym, iz); in.ez_xp = EZv(ixp, iy, iz); in.ez_xm = EZv(ixm, iy, iz);
    in.ez_yp_zm = EZv(ix, iyp, izm); in.ez_zm = EZv(ix, iy, izm); in.ez_xp_zm = EZv(ixp, iy, izm);
    in.z000 = ZT(ixm, iym, izm); in.z100 = ZT(ix, iym, izm); in.z010 = ZT(ixm, iy, izm); in.z110 = ZT(ix, iy, izm);
    in.z001 = ZT(ixm, iym, iz); in.z101 = ZT(ix, iym, iz); in.z011 = ZT(ixm, iy, iz); in.z111 = ZT(ix, iy, iz);
    in.etx[0] = ETv(L.eta_x, ix, iym, izm); in.etx[1] = ETv(L.eta_x, ix, iym, iz);
    in.etx[2] = ETv(L.eta_x, ix, iy, izm); in.etx[3] = ETv(L.eta_x, ix, iy, iz);
    in.ety[0] = ETv(L.eta_y, ixm, iy, izm); in.ety[1] = ETv(L.eta_y, ix, iy, izm);
    in.ety[2] = ETv(L.eta_y, ixm, iy, iz); in.ety[3] = ETv(L.eta_y, ix, iy, iz);
    in.etz[0] = ETv(L.eta_z, ixm, iym, iz); in.etz[1] = ETv(L.eta_z, ix, iym, iz);
    in.etz[2] = ETv(L.eta_z, ixm, iy, iz); in.etz[3] = ETv(L.eta_z, ix, iy, iz);
}

// the operands of cell (ix,iy,iz), iz >= 1, when `in` still holds those of cell (ix,iy,iz-1): what lay in plane
// iz (izp of the cell below) moves to this cell's own plane, what lay in the cell's plane moves to izm; only the
// rest is loaded
template <class T> EMG_HD void residual_load_roll(const Level<T> &L, int ix, int iy, int iz, ResIn<T> &in)
{
    const Axes<T, 0> A(L);
    const int ixm = ix > 0 ? ix - 1 : 0, iym = iy > 0 ? iy - 1 : 0;
    const int ixp = ix + 1, iyp = iy + 1, izp = iz + 1;
    in.hz0 = in.hz1;
    in.hz1 = L.ihz[iz];
    // plane iz -> izm
    in.ex_zm = in.ex_c; in.ey_zm = in.ey_c;
    in.ez_zm = in.ez_c; in.ez_xp_zm = in.ez_xp; in.ez_yp_zm = in.ez_yp;
    in.z000 = in.z001; in.z100 = in.z101; in.z010 = in.z011; in.z110 = in.z111;
    in.etx[0] = in.etx[1]; in.etx[2] = in.etx[3];
    in.ety[0] = in.ety[2]; in.ety[1] = in.ety[3];
    // plane izp -> iz
    in.ex_c = in.ex_zp; in.exm_c = in.exm_zp; in.ey_c = in.ey_zp; in.eym_c = in.eym_zp;
    // loaded
    in.ez_c = EZv(ix, iy, iz);
    in.ex_zp = EXv(ix, iy, izp); in.exm_zp = EXv(ixm, iy, izp);
    in.ex_yp = EXv(ix, iyp, iz); in.exm_yp = EXv(ixm, iyp, iz); in.ex_ym = EXv(ix, iym, iz);
    in.ey_zp = EYv(ix, iy, izp); in.eym_zp = EYv(ix, iym, izp);
    in.ey_xp = EYv(ixp, iy, iz); in.ey_xm = EYv(ixm, iy, iz); in.eym_xp = EYv(ixp, iym, iz);
    in.ez_yp = EZv(ix, iyp, iz); in.ez_ym = EZv(ix, iym, iz); in.ez_xp = EZv(ixp, iy, iz); in.ez_xm = EZv(ixm, iy, iz);
    in.z001 = ZT(ixm, iym, iz); in.z101 = ZT(ix, iym, iz); in.z011 = ZT(ixm, iy, iz); in.z111 = ZT(ix, iy, iz);
    in.etx[1] = ETv(L.eta_x, ix, iym, iz); in.etx[3] = ETv(L.eta_x, ix, iy, iz);
    in.ety[2] = ETv(L.eta_y, ixm, iy, iz); in.ety[3] = ETv(L.eta_y, ix, iy, iz);
    in.etz[0] = ETv(L.eta_z, ixm, iym, iz); in.etz[1] = ETv(L.eta_z, ix, iym, iz);
    in.etz[2] = ETv(L.eta_z, ixm, iy, iz); in.etz[3] = ETv(L.eta_z, ix, iy, iz);
#undef EXv
#undef EYv
#undef EZv
#undef ZT
#undef ETv
}

// The three values of extended cell (ix,iy,iz), ix < nx etc. (the entries core.amat_x touches), from its operands:
//   SRC:  r = s - A e   (solver.residual)            !SRC:  r = -A e   (the Krylov operator)
template <class T, bool SRC>
EMG_HD void residual_compute(const Level<T> &L, const ResIn<T> &in, int ix, int iy, int iz, T &ox, T &oy, T &oz)
{
    const Axes<T, 0> A(L);
    const double hx1 = in.hx1, hx0 = in.hx0, hy1 = in.hy1, hy0 = in.hy0, hz1 = in.hz1, hz0 = in.hz0;
    const T ex_c = in.ex_c, ey_c = in.ey_c, ez_c = in.ez_c;
    // 1. curl on the faces around the three edges (core.py:136-155)
    T v1pp = (in.ez_yp - ez_c) * hy1 - (in.ey_zp - ey_c) * hz1;
    T v1mp = (ez_c - in.ez_ym) * hy0 - (in.eym_zp - in.eym_c) * hz1;
    T v1pm = (in.ez_yp_zm - in.ez_zm) * hy1 - (ey_c - in.ey_zm) * hz0;

    T v2pp = (in.ex_zp - ex_c) * hz1 - (in.ez_xp - ez_c) * hx1;
    T v2mp = (in.exm_zp - in.exm_c) * hz1 - (ez_c - in.ez_xm) * hx0;
    T v2pm = (ex_c - in.ex_zm) * hz0 - (in.ez_xp_zm - in.ez_zm) * hx1;

    T v3pp = (in.ey_xp - ey_c) * hx1 - (in.ex_yp - ex_c) * hy1;
    T v3mp = (ey_c - in.ey_xm) * hx0 - (in.exm_yp - in.exm_c) * hy1;
    T v3pm = (in.eym_xp - in.eym_c) * hx1 - (ex_c - in.ex_ym) * hy0;

    // 2. face averages of zeta (core.py:160-170)
    v1pp *= in.z011 + in.z111;
    v1mp *= in.z001 + in.z101;
    v1pm *= in.z010 + in.z110;
    v2pp *= in.z101 + in.z111;
    v2mp *= in.z001 + in.z011;
    v2pm *= in.z100 + in.z110;
    v3pp *= in.z110 + in.z111;
    v3mp *= in.z010 + in.z011;
    v3pm *= in.z100 + in.z101;

    // 3. second curl (core.py:174-176)
    T rrx = v3pp * hy1 - v3pm * hy0 - v2pp * hz1 + v2pm * hz0;
    T rry = v1pp * hz1 - v1pm * hz0 - v3pp * hx1 + v3mp * hx0;
    T rrz = v2pp * hx1 - v2mp * hx0 - v1pp * hy1 + v1mp * hy0;

    // 4. eta edge sums (core.py:181-186)
    const T stx = in.etx[0] + in.etx[1] + in.etx[2] + in.etx[3];
    const T sty = in.ety[0] + in.ety[1] + in.ety[2] + in.ety[3];
    const T stz = in.etz[0] + in.etz[1] + in.etz[2] + in.etz[3];
    // PEC rows (core.py:193-198)
    if (iy == 0 || iz == 0) rrx = zero<T>();
    if (ix == 0 || iz == 0) rry = zero<T>();
    if (ix == 0 || iy == 0) rrz = zero<T>();

    // 5. r = s - (0.5 rr - 0.25 st e)   (core.py:204-206)
    const T ax = 0.5 * rrx - 0.25 * (stx * ex_c), ay = 0.5 * rry - 0.25 * (sty * ey_c), az = 0.5 * rrz - 0.25 * (stz * ez_c);
    if (SRC) {
        ox = L.sx[A.iex(ix, iy, iz)] - ax;
        oy = L.sy[A.iey(ix, iy, iz)] - ay;
        oz = L.sz[A.iez(ix, iy, iz)] - az;
    } else {
        ox = -ax; oy = -ay; oz = -az;
    }
}

// (one cell on its own: the compiler interleaves loads and arithmetic -- 92 registers against 170 with the
// operands gathered first; the rolled column above trades registers for half of the loads)
// The three values of extended cell (ix,iy,iz), ix < nx etc. (the entries core.amat_x touches):
//   SRC:  r = s - A e   (solver.residual)            !SRC:  r = -A e   (the Krylov operator)
template <class T, bool SRC>
EMG_HD void residual_values(const Level<T> &L, int ix, int iy, int iz, T &ox, T &oy, T &oz)
{
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    (void)nx; (void)ny; (void)nz;
    const Axes<T, 0> A(L);
    const int ixm = ix > 0 ? ix - 1 : 0, iym = iy > 0 ? iy - 1 : 0, izm = iz > 0 ? iz - 1 : 0;
    const int ixp = ix + 1, iyp = iy + 1, izp = iz + 1;
    const double hx1 = L.ihx[ix], hx0 = L.ihx[ixm];
    const double hy1 = L.ihy[iy], hy0 = L.ihy[iym];
    const double hz1 = L.ihz[iz], hz0 = L.ihz[izm];
#define EXv(i, j, k) L.ex[A.iex(i, j, k)]
#define EYv(i, j, k) L.ey[A.iey(i, j, k)]
#define EZv(i, j, k) L.ez[A.iez(i, j, k)]
#define ZT(i, j, k) L.zeta[A.icc(i, j, k)]
    const T ex_c = EXv(ix, iy, iz), ey_c = EYv(ix, iy, iz), ez_c = EZv(ix, iy, iz);
    // 1. curl on the faces around the three edges (core.py:136-155)
    T v1pp = (EZv(ix, iyp, iz) - ez_c) * hy1 - (EYv(ix, iy, izp) - ey_c) * hz1;
    T v1mp = (ez_c - EZv(ix, iym, iz)) * hy0 - (EYv(ix, iym, izp) - EYv(ix, iym, iz)) * hz1;
    T v1pm = (EZv(ix, iyp, izm) - EZv(ix, iy, izm)) * hy1 - (ey_c - EYv(ix, iy, izm)) * hz0;

    T v2pp = (EXv(ix, iy, izp) - ex_c) * hz1 - (EZv(ixp, iy, iz) - ez_c) * hx1;
    T v2mp = (EXv(ixm, iy, izp) - EXv(ixm, iy, iz)) * hz1 - (ez_c - EZv(ixm, iy, iz)) * hx0;
    T v2pm = (ex_c - EXv(ix, iy, izm)) * hz0 - (EZv(ixp, iy, izm) - EZv(ix, iy, izm)) * hx1;

    T v3pp = (EYv(ixp, iy, iz) - ey_c) * hx1 - (EXv(ix, iyp, iz) - ex_c) * hy1;
    T v3mp = (ey_c - EYv(ixm, iy, iz)) * hx0 - (EXv(ixm, iyp, iz) - EXv(ixm, iy, iz)) * hy1;
    T v3pm = (EYv(ixp, iym, iz) - EYv(ix, iym, iz)) * hx1 - (ex_c - EXv(ix, iym, iz)) * hy0;

    // 2. face averages of zeta (core.py:160-170)
    const double z000 = ZT(ixm, iym, izm), z100 = ZT(ix, iym, izm);
    const double z010 = ZT(ixm, iy, izm), z110 = ZT(ix, iy, izm);
    const double z001 = ZT(ixm, iym, iz), z101 = ZT(ix, iym, iz);
    const double z011 = ZT(ixm, iy, iz), z111 = ZT(ix, iy, iz);
    v1pp *= z011 + z111;
    v1mp *= z001 + z101;
    v1pm *= z010 + z110;
    v2pp *= z101 + z111;
    v2mp *= z001 + z011;
    v2pm *= z100 + z110;
    v3pp *= z110 + z111;
    v3mp *= z010 + z011;
    v3pm *= z100 + z101;

    // 3. second curl (core.py:174-176)
    T rrx = v3pp * hy1 - v3pm * hy0 - v2pp * hz1 + v2pm * hz0;
    T rry = v1pp * hz1 - v1pm * hz0 - v3pp * hx1 + v3mp * hx0;
    T rrz = v2pp * hx1 - v2mp * hx0 - v1pp * hy1 + v1mp * hy0;

    // 4. eta edge sums (core.py:181-186)
#define ETv(p, i, j, k) (p)[A.icc(i, j, k)]
    const T stx = ETv(L.eta_x, ix, iym, izm) + ETv(L.eta_x, ix, iym, iz) +
                  ETv(L.eta_x, ix, iy, izm) + ETv(L.eta_x, ix, iy, iz);
    const T sty = ETv(L.eta_y, ixm, iy, izm) + ETv(L.eta_y, ix, iy, izm) +
                  ETv(L.eta_y, ixm, iy, iz) + ETv(L.eta_y, ix, iy, iz);
    const T stz = ETv(L.eta_z, ixm, iym, iz) + ETv(L.eta_z, ix, iym, iz) +
                  ETv(L.eta_z, ixm, iy, iz) + ETv(L.eta_z, ix, iy, iz);
#undef ETv
    // PEC rows (core.py:193-198)
    if (iy == 0 || iz == 0) rrx = zero<T>();
    if (ix == 0 || iz == 0) rry = zero<T>();
    if (ix == 0 || iy == 0) rrz = zero<T>();

    // 5. r = s - (0.5 rr - 0.25 st e)   (core.py:204-206)
    const T ax = 0.5 * rrx - 0.25 * (stx * ex_c), ay = 0.5 * rry - 0.25 * (sty * ey_c), az = 0.5 * rrz - 0.25 * (stz * ez_c);
    if (SRC) {
        ox = L.sx[A.iex(ix, iy, iz)] - ax;
        oy = L.sy[A.iey(ix, iy, iz)] - ay;
        oz = L.sz[A.iez(ix, iy, iz)] - az;
    } else {
        ox = -ax; oy = -ay; oz = -az;
    }
#undef EXv
#undef EYv
#undef EZv
#undef ZT
}

// ---------------------------------------------------------------------------------------
// Residual  r = s - A e  at the three "lower" edges of extended cell (ix,iy,iz),
// 0 <= ix <= nx etc. Follows core.amat_x (reference emg3d/core.py:57-206) for the cells
// 0..n-1 and leaves the source value on upper-boundary entries (which core.amat_x never
// touches), so one launch produces the complete residual buffer of
// solver.residual (reference emg3d/solver.py:1022-1070). Returns |rx|^2+|ry|^2+|rz|^2 of
// the entries this cell owns (for the fused l2-norm).
// `r*` may alias `s*` (in-place form of core.amat_x: r -= A e).
// ---------------------------------------------------------------------------------------
template <class T>
EMG_HD double residual_cell(const Level<T> &L, T *rx, T *ry, T *rz, int ix, int iy, int iz)
{
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    const Axes<T, 0> A(L);
    const bool inx = ix < nx, iny = iy < ny, inz = iz < nz;
    double acc = 0.0;

    if (inx && iny && inz) {
        T ox, oy, oz;
        residual_values<T, true>(L, ix, iy, iz, ox, oy, oz);
        acc = abs2(ox) + abs2(oy) + abs2(oz);
        if (rx) {
            rx[A.iex(ix, iy, iz)] = ox;
            ry[A.iey(ix, iy, iz)] = oy;
            rz[A.iez(ix, iy, iz)] = oz;
        }
    } else {
        // Upper-boundary entries: r = s (untouched by core.amat_x).
        if (inx) {  // ex exists for ix < nx, any iy <= ny, iz <= nz
            const T v = L.sx[A.iex(ix, iy, iz)];
            acc += abs2(v);
            if (rx) rx[A.iex(ix, iy, iz)] = v;
        }
        if (iny) {
            const T v = L.sy[A.iey(ix, iy, iz)];
            acc += abs2(v);
            if (ry) ry[A.iey(ix, iy, iz)] = v;
        }
        if (inz) {
            const T v = L.sz[A.iez(ix, iy, iz)];
            acc += abs2(v);
            if (rz) rz[A.iez(ix, iy, iz)] = v;
        }
    }
    return acc;
}

// The cells (ix, iy, z0 .. z1-1) of one column by one thread, bottom to top: the operands a cell shares with the
// cell below it (half of the field values, zeta and eta of the plane between them) are carried in registers
// (residual_load_roll). Same operands, same arithmetic (ONE call site of residual_compute), same order of the
// sum of squares as a loop over residual_cell.
template <class T>
EMG_HD double residual_column(const Level<T> &L, T *rx, T *ry, T *rz, int ix, int iy, int z0, int z1)
{
    const Axes<T, 0> A(L);
    double acc = 0.0;
    if (ix < L.nx && iy < L.ny) {
        ResIn<T> in;
        bool have = false;
        for (int iz = z0; iz < z1; ++iz) {
            if (iz < L.nz) {
                if (have) residual_load_roll<T>(L, ix, iy, iz, in);
                else residual_load<T>(L, ix, iy, iz, in);
                have = true;
                T ox, oy, oz;
                residual_compute<T, true>(L, in, ix, iy, iz, ox, oy, oz);
                acc += abs2(ox) + abs2(oy) + abs2(oz);
                if (rx) {
                    rx[A.iex(ix, iy, iz)] = ox;
                    ry[A.iey(ix, iy, iz)] = oy;
                    rz[A.iez(ix, iy, iz)] = oz;
                }
            } else {
                acc += residual_cell<T>(L, rx, ry, rz, ix, iy, iz);      // the top plane: r = s on ex / ey
            }
        }
    } else {
        for (int iz = z0; iz < z1; ++iz) acc += residual_cell<T>(L, rx, ry, rz, ix, iy, iz);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------
// Point smoother: one node update of core.gauss_seidel (reference emg3d/core.py:346-503).
// The 6x6 complex-symmetric system of the six edges attached to node (ix,iy,iz) is
// assembled in registers and solved by an unrolled LDL^T without pivoting (core.solve,
// core.py:1481-1616, n = 6); entry (1,0) is structurally zero and stays zero.
// ---------------------------------------------------------------------------------------
EMG_HD constexpr bool pt_nz(int i, int j) { return !(i == 1 && j == 0); }

template <class T> EMG_HD void solve6(const T (&dg)[6], const double (&od)[6][6], T (&b)[6])
{
    T Lm[6][6];
    T dinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T u[6];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            if (!pt_nz(i, j)) continue;
            T t = T(od[i][j]);
#pragma unroll
            for (int k = 0; k < j; ++k)
                if (pt_nz(i, k) && pt_nz(j, k)) t = nmad(u[k], Lm[j][k], t);
            u[j] = t;
            Lm[i][j] = t * dinv[j];
        }
        T d = dg[i];
#pragma unroll
        for (int k = 0; k < i; ++k)
            if (pt_nz(i, k)) d = nmad(u[k], Lm[i][k], d);
        dinv[i] = recip_fast(d);
    }
    // forward substitution, diagonal scaling, backward substitution (core.py:1597-1616)
#pragma unroll
    for (int i = 1; i < 6; ++i) {
#pragma unroll
        for (int k = 0; k < i; ++k)
            if (pt_nz(i, k)) b[i] = nmad(Lm[i][k], b[k], b[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] *= dinv[i];
#pragma unroll
    for (int j = 4; j >= 0; --j) {
#pragma unroll
        for (int k = j + 1; k < 6; ++k)
            if (pt_nz(k, j)) b[j] = nmad(Lm[k][j], b[k], b[j]);
    }
}

// Where the point smoother finds the field: directly in the global arrays ...
template <class T> struct EdgesGlobal {
    T *ex, *ey, *ez;
    int nx, ny;
    EMG_HD explicit EdgesGlobal(const Level<T> &L) : ex(L.ex), ey(L.ey), ez(L.ez), nx(L.nx), ny(L.ny) {}
    EMG_HD T &x(int i, int j, int k) const { return ex[i + nx * (j + (ny + 1) * k)]; }
    EMG_HD T &y(int i, int j, int k) const { return ey[i + (nx + 1) * (j + ny * k)]; }
    EMG_HD T &z(int i, int j, int k) const { return ez[i + (nx + 1) * (j + (ny + 1) * k)]; }
};
// ... or in an LDS copy of one tile of BX x BY x BZ nodes starting at node (ox,oy,oz), with
// the one-edge halo the 6x6 node systems read (launch.h: tiled schedule). Three dense
// boxes, x fastest:  ex [BX+1][BY+2][BZ+2],  ey [BX+2][BY+1][BZ+2],  ez [BX+2][BY+2][BZ+1].
template <class T, int BX, int BY, int BZ> struct EdgesTile {
    static constexpr int NXE = (BX + 1) * (BY + 2) * (BZ + 2);
    static constexpr int NYE = (BX + 2) * (BY + 1) * (BZ + 2);
    static constexpr int NZE = (BX + 2) * (BY + 2) * (BZ + 1);
    static constexpr int ELEMS = NXE + NYE + NZE;
    static constexpr int LDS_ELEMS = ELEMS + 1;   // + one slot that absorbs out-of-range copies
    // zeta of the (BX+1)(BY+1)(BZ+1) cells around the tile's nodes, as doubles behind the edges
    static constexpr int NZC = (BX + 1) * (BY + 1) * (BZ + 1);
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * sizeof(T) + (size_t)(NZC + 1) * sizeof(double);
    EMG_HD double *zbox() const { return reinterpret_cast<double *>(lds + LDS_ELEMS); }
    // cell (i,j,k), i in [x0-1, x0+BX-1] etc.  (ox = x0-1 is also the first cell)
    EMG_HD double &zc(int i, int j, int k) const
    {
        return zbox()[(i - ox) + (BX + 1) * ((j - oy) + (BY + 1) * (k - oz))];
    }
    T *lds;
    int ox, oy, oz;
    EMG_HD EdgesTile(T *l, int x0, int y0, int z0) : lds(l), ox(x0 - 1), oy(y0 - 1), oz(z0 - 1) {}
    EMG_HD T &x(int i, int j, int k) const { return lds[(i - ox) + (BX + 1) * ((j - oy) + (BY + 2) * (k - oz))]; }
    EMG_HD T &y(int i, int j, int k) const
    {
        return lds[NXE + (i - ox) + (BX + 2) * ((j - oy) + (BY + 1) * (k - oz))];
    }
    EMG_HD T &z(int i, int j, int k) const
    {
        return lds[NXE + NYE + (i - ox) + (BX + 2) * ((j - oy) + (BY + 2) * (k - oz))];
    }
};

// The model- and source-dependent inputs of one node update: eta edge sums (core.py:377-390),
// zeta of the 8 surrounding cells, source at the 6 edges. Loading them (point_load) is
// separate from the update (point_update) so that a kernel can fetch the inputs of its next
// node while it computes the current one.
// `pst` (optional): precomputed eta edge sums [stx|sty|stz], shaped like ex/ey/ez
// (point_setup_cell) -- 6 loads instead of 24; the sums are formed in the same order, so
// both paths give identical bits.
template <class T> struct PointIn {
    T st[6];
    double z[8];   // z000 z100 z010 z110 z001 z101 z011 z111  (x fastest)
    T s[6];
};

// zeta straight from the level's array ...
template <class T> struct ZetaGlobal {
    const Level<T> &L;
    EMG_HD double operator()(int i, int j, int k) const { return L.zeta[i + L.nx * (j + L.ny * k)]; }
};
// ... or from the tile's LDS copy (EdgesTile::zc)
template <class E> struct ZetaTile {
    const E &ed;
    EMG_HD double operator()(int i, int j, int k) const { return ed.zc(i, j, k); }
};

template <class T, class Z> EMG_HD void point_load_zeta(const Z &zeta, int ix, int iy, int iz, PointIn<T> &in)
{
    const int ixm = ix - 1, iym = iy - 1, izm = iz - 1;
    in.z[0] = zeta(ixm, iym, izm); in.z[1] = zeta(ix, iym, izm);
    in.z[2] = zeta(ixm, iy, izm);  in.z[3] = zeta(ix, iy, izm);
    in.z[4] = zeta(ixm, iym, iz);  in.z[5] = zeta(ix, iym, iz);
    in.z[6] = zeta(ixm, iy, iz);   in.z[7] = zeta(ix, iy, iz);
}
// source at the node's six edges
template <class T> EMG_HD void point_load_source(const Level<T> &L, int ix, int iy, int iz, PointIn<T> &in)
{
    const Axes<T, 0> A(L);
    const int ixm = ix - 1, iym = iy - 1, izm = iz - 1;
    in.s[0] = L.sx[A.iex(ixm, iy, iz)]; in.s[1] = L.sx[A.iex(ix, iy, iz)];
    in.s[2] = L.sy[A.iey(ix, iym, iz)]; in.s[3] = L.sy[A.iey(ix, iy, iz)];
    in.s[4] = L.sz[A.iez(ix, iy, izm)]; in.s[5] = L.sz[A.iez(ix, iy, iz)];
}
// eta sums of the node's six edges: from the edge-shaped arrays of point_setup_cell (ST), or
// formed on the fly from eta (core.py:377-390) -- same order of additions, identical bits
template <class T, bool ST>
EMG_HD void point_load_eta(const Level<T> &L, const T *pst, int ix, int iy, int iz, PointIn<T> &in)
{
    const Axes<T, 0> A(L);
    const int ixm = ix - 1, iym = iy - 1, izm = iz - 1;
    if (ST) {
        const int e0 = A.iex(ixm, iy, iz), e1 = A.iex(ix, iy, iz), e2 = A.iey(ix, iym, iz), e3 = A.iey(ix, iy, iz);
        const int e4 = A.iez(ix, iy, izm), e5 = A.iez(ix, iy, iz);
        const T *sty = pst + (size_t)L.nx * (L.ny + 1) * (L.nz + 1);
        const T *stz = sty + (size_t)(L.nx + 1) * L.ny * (L.nz + 1);
        in.st[0] = pst[e0]; in.st[1] = pst[e1]; in.st[2] = sty[e2]; in.st[3] = sty[e3];
        in.st[4] = stz[e4]; in.st[5] = stz[e5];
    } else {
#define ETv(p, i, j, k) (p)[A.icc(i, j, k)]
        in.st[0] = ETv(L.eta_x, ixm, iy, iz) + ETv(L.eta_x, ixm, iy, izm) +
                   ETv(L.eta_x, ixm, iym, iz) + ETv(L.eta_x, ixm, iym, izm);
        in.st[1] = ETv(L.eta_x, ix, iy, iz) + ETv(L.eta_x, ix, iy, izm) +
                   ETv(L.eta_x, ix, iym, iz) + ETv(L.eta_x, ix, iym, izm);
        in.st[2] = ETv(L.eta_y, ix, iym, iz) + ETv(L.eta_y, ix, iym, izm) +
                   ETv(L.eta_y, ixm, iym, iz) + ETv(L.eta_y, ixm, iym, izm);
        in.st[3] = ETv(L.eta_y, ix, iy, iz) + ETv(L.eta_y, ix, iy, izm) +
                   ETv(L.eta_y, ixm, iy, iz) + ETv(L.eta_y, ixm, iy, izm);
        in.st[4] = ETv(L.eta_z, ix, iy, izm) + ETv(L.eta_z, ix, iym, izm) +
                   ETv(L.eta_z, ixm, iy, izm) + ETv(L.eta_z, ixm, iym, izm);
        in.st[5] = ETv(L.eta_z, ix, iy, iz) + ETv(L.eta_z, ix, iym, iz) +
                   ETv(L.eta_z, ixm, iy, iz) + ETv(L.eta_z, ixm, iym, iz);
#undef ETv
    }
}
template <class T, bool ST>
EMG_HD void point_load_model(const Level<T> &L, const T *pst, int ix, int iy, int iz, PointIn<T> &in)
{
    point_load_source<T>(L, ix, iy, iz, in);
    point_load_eta<T, ST>(L, pst, ix, iy, iz, in);
}
template <class T, bool ST, class Z>
EMG_HD void point_load(const Level<T> &L, const T *pst, const Z &zeta, int ix, int iy, int iz, PointIn<T> &in)
{
    point_load_zeta<T>(zeta, ix, iy, iz, in);
    point_load_model<T, ST>(L, pst, ix, iy, iz, in);
}

// Eta edge sums of the three "lower" edges of extended cell (ix,iy,iz) (the ones attached
// to interior nodes; all others stay 0), for the `pst` form of point_load.
template <class T> EMG_HD void point_setup_cell(const Level<T> &L, T *pst, int ix, int iy, int iz)
{
    const Axes<T, 0> A(L);
    T *sty = pst + (size_t)L.nx * (L.ny + 1) * (L.nz + 1);
    T *stz = sty + (size_t)(L.nx + 1) * L.ny * (L.nz + 1);
#define ETv(p, i, j, k) (p)[A.icc(i, j, k)]
    if (ix < L.nx && iy >= 1 && iy < L.ny && iz >= 1 && iz < L.nz)
        pst[A.iex(ix, iy, iz)] = ETv(L.eta_x, ix, iy, iz) + ETv(L.eta_x, ix, iy, iz - 1) +
                                 ETv(L.eta_x, ix, iy - 1, iz) + ETv(L.eta_x, ix, iy - 1, iz - 1);
    if (iy < L.ny && ix >= 1 && ix < L.nx && iz >= 1 && iz < L.nz)
        sty[A.iey(ix, iy, iz)] = ETv(L.eta_y, ix, iy, iz) + ETv(L.eta_y, ix, iy, iz - 1) +
                                 ETv(L.eta_y, ix - 1, iy, iz) + ETv(L.eta_y, ix - 1, iy, iz - 1);
    if (iz < L.nz && ix >= 1 && ix < L.nx && iy >= 1 && iy < L.ny)
        stz[A.iez(ix, iy, iz)] = ETv(L.eta_z, ix, iy, iz) + ETv(L.eta_z, ix, iy - 1, iz) +
                                 ETv(L.eta_z, ix - 1, iy, iz) + ETv(L.eta_z, ix - 1, iy - 1, iz);
#undef ETv
}

template <class T, class E>
EMG_HD void point_update(const Level<T> &L, const PointIn<T> &in, const E &ed, int ix, int iy, int iz)
{
    const int ixm = ix - 1, ixp = ix + 1, iym = iy - 1, iyp = iy + 1, izm = iz - 1, izp = iz + 1;
    const double hx0 = L.ihx[ixm], hx1 = L.ihx[ix];
    const double hy0 = L.ihy[iym], hy1 = L.ihy[iy];
    const double hz0 = L.ihz[izm], hz1 = L.ihz[iz];
    const double kx0 = 0.5 * hx0, kx1 = 0.5 * hx1, ky0 = 0.5 * hy0, ky1 = 0.5 * hy1;
    const double kz0 = 0.5 * hz0, kz1 = 0.5 * hz1;

    // zeta of the 8 cells around the node: z[a][b][c], a/b/c = 0 (minus) or 1 (this)
    const double z000 = in.z[0], z100 = in.z[1], z010 = in.z[2], z110 = in.z[3];
    const double z001 = in.z[4], z101 = in.z[5], z011 = in.z[6], z111 = in.z[7];

    // the 24 face averages (core.py:351-374), names as in the reference
    const double mzyLxm = ky0 * (z001 + z000), mzyRxm = ky1 * (z011 + z010);
    const double myzLxm = kz0 * (z010 + z000), myzRxm = kz1 * (z011 + z001);
    const double mzyLxp = ky0 * (z101 + z100), mzyRxp = ky1 * (z111 + z110);
    const double myzLxp = kz0 * (z110 + z100), myzRxp = kz1 * (z111 + z101);
    const double mzxLym = kx0 * (z001 + z000), mzxRym = kx1 * (z101 + z100);
    const double mxzLym = kz0 * (z100 + z000), mxzRym = kz1 * (z101 + z001);
    const double mzxLyp = kx0 * (z011 + z010), mzxRyp = kx1 * (z111 + z110);
    const double mxzLyp = kz0 * (z110 + z010), mxzRyp = kz1 * (z111 + z011);
    const double myxLzm = kx0 * (z010 + z000), myxRzm = kx1 * (z110 + z100);
    const double mxyLzm = ky0 * (z100 + z000), mxyRzm = ky1 * (z110 + z010);
    const double myxLzp = kx0 * (z011 + z001), myxRzp = kx1 * (z111 + z101);
    const double mxyLzp = ky0 * (z101 + z001), mxyRzp = ky1 * (z111 + z011);

    const T st0 = in.st[0], st1 = in.st[1], st2 = in.st[2], st3 = in.st[3], st4 = in.st[4], st5 = in.st[5];

    // diagonal (core.py:396-412): -st/4 + real curl-curl part
    T dg[6];
    dg[0] = (mzyRxm * hy1 + mzyLxm * hy0 + myzRxm * hz1 + myzLxm * hz0) - 0.25 * st0;
    dg[1] = (mzyRxp * hy1 + mzyLxp * hy0 + myzRxp * hz1 + myzLxp * hz0) - 0.25 * st1;
    dg[2] = (mzxRym * hx1 + mzxLym * hx0 + mxzRym * hz1 + mxzLym * hz0) - 0.25 * st2;
    dg[3] = (mzxRyp * hx1 + mzxLyp * hx0 + mxzRyp * hz1 + mxzLyp * hz0) - 0.25 * st3;
    dg[4] = (myxRzm * hx1 + myxLzm * hx0 + mxyRzm * hy1 + mxyLzm * hy0) - 0.25 * st4;
    dg[5] = (myxRzp * hx1 + myxLzp * hx0 + mxyRzp * hy1 + mxyLzp * hy0) - 0.25 * st5;

    // strictly lower off-diagonals, all real (core.py:419-430); (1,0),(3,2),(5,4) are zero
    double od[6][6];
    od[1][0] = 0.0;
    od[2][0] = -mzyLxm * hx0; od[3][0] = mzyRxm * hx0; od[4][0] = -myzLxm * hx0; od[5][0] = myzRxm * hx0;
    od[2][1] = mzyLxp * hx1; od[3][1] = -mzyRxp * hx1; od[4][1] = myzLxp * hx1; od[5][1] = -myzRxp * hx1;
    od[3][2] = 0.0;
    od[4][2] = -mxzLym * hy0; od[5][2] = mxzRym * hy0;
    od[4][3] = mxzLyp * hy1; od[5][3] = -mxzRyp * hy1;
    od[5][4] = 0.0;

    // right-hand side: source + terms of the 24 neighbouring edges (core.py:436-492)
#define EXv(i, j, k) ed.x(i, j, k)
#define EYv(i, j, k) ed.y(i, j, k)
#define EZv(i, j, k) ed.z(i, j, k)
    T rhs[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) rhs[r] = in.s[r];

    // the 24 neighbour edges, each used twice
    const T ex_mpc = EXv(ixm, iyp, iz), ex_mmc = EXv(ixm, iym, iz);   // ex[ixm, iy+-1, iz]
    const T ex_mcp = EXv(ixm, iy, izp), ex_mcm = EXv(ixm, iy, izm);   // ex[ixm, iy, iz+-1]
    const T ex_cpc = EXv(ix, iyp, iz), ex_cmc = EXv(ix, iym, iz);     // ex[ix, iy+-1, iz]
    const T ex_ccp = EXv(ix, iy, izp), ex_ccm = EXv(ix, iy, izm);     // ex[ix, iy, iz+-1]
    const T ey_mcc = EYv(ixm, iy, iz), ey_mmc = EYv(ixm, iym, iz);    // ey[ixm, iy|iym, iz]
    const T ey_pcc = EYv(ixp, iy, iz), ey_pmc = EYv(ixp, iym, iz);    // ey[ixp, iy|iym, iz]
    const T ey_cmp = EYv(ix, iym, izp), ey_cmm = EYv(ix, iym, izm);   // ey[ix, iym, iz+-1]
    const T ey_ccp = EYv(ix, iy, izp), ey_ccm = EYv(ix, iy, izm);     // ey[ix, iy, iz+-1]
    const T ez_mcc = EZv(ixm, iy, iz), ez_mcm = EZv(ixm, iy, izm);    // ez[ixm, iy, iz|izm]
    const T ez_pcc = EZv(ixp, iy, iz), ez_pcm = EZv(ixp, iy, izm);    // ez[ixp, iy, iz|izm]
    const T ez_cmc = EZv(ix, iym, iz), ez_cmm = EZv(ix, iym, izm);    // ez[ix, iym, iz|izm]
    const T ez_cpc = EZv(ix, iyp, iz), ez_cpm = EZv(ix, iyp, izm);    // ez[ix, iyp, iz|izm]
#undef EXv
#undef EYv
#undef EZv

    rhs[0] += mzyRxm * (ey_mcc * hx0 + ex_mpc * hy1);
    rhs[0] += mzyLxm * (ex_mmc * hy0 - ey_mmc * hx0);
    rhs[0] += myzRxm * (ez_mcc * hx0 + ex_mcp * hz1);
    rhs[0] += myzLxm * (ex_mcm * hz0 - ez_mcm * hx0);

    rhs[1] += mzyRxp * (ex_cpc * hy1 - ey_pcc * hx1);
    rhs[1] += mzyLxp * (ey_pmc * hx1 + ex_cmc * hy0);
    rhs[1] += myzRxp * (ex_ccp * hz1 - ez_pcc * hx1);
    rhs[1] += myzLxp * (ez_pcm * hx1 + ex_ccm * hz0);

    rhs[2] += mzxRym * (ey_pmc * hx1 + ex_cmc * hy0);
    rhs[2] += mzxLym * (ey_mmc * hx0 - ex_mmc * hy0);
    rhs[2] += mxzRym * (ez_cmc * hy0 + ey_cmp * hz1);
    rhs[2] += mxzLym * (ey_cmm * hz0 - ez_cmm * hy0);

    rhs[3] += mzxRyp * (ey_pcc * hx1 - ex_cpc * hy1);
    rhs[3] += mzxLyp * (ey_mcc * hx0 + ex_mpc * hy1);
    rhs[3] += mxzRyp * (ey_ccp * hz1 - ez_cpc * hy1);
    rhs[3] += mxzLyp * (ez_cpm * hy1 + ey_ccm * hz0);

    rhs[4] += myxRzm * (ez_pcm * hx1 + ex_ccm * hz0);
    rhs[4] += myxLzm * (ez_mcm * hx0 - ex_mcm * hz0);
    rhs[4] += mxyRzm * (ez_cpm * hy1 + ey_ccm * hz0);
    rhs[4] += mxyLzm * (ez_cmm * hy0 - ey_cmm * hz0);

    rhs[5] += myxRzp * (ez_pcc * hx1 - ex_ccp * hz1);
    rhs[5] += myxLzp * (ez_mcc * hx0 + ex_mcp * hz1);
    rhs[5] += mxyRzp * (ez_cpc * hy1 - ey_ccp * hz1);
    rhs[5] += mxyLzp * (ez_cmc * hy0 + ey_cmp * hz1);

    solve6<T>(dg, od, rhs);

    // write the six edges (core.py:498-503)
    ed.x(ixm, iy, iz) = rhs[0];
    ed.x(ix, iy, iz) = rhs[1];
    ed.y(ix, iym, iz) = rhs[2];
    ed.y(ix, iy, iz) = rhs[3];
    ed.z(ix, iy, izm) = rhs[4];
    ed.z(ix, iy, iz) = rhs[5];
}
// load + update in one go; pst = nullptr: eta edge sums formed on the fly
template <class T, class E>
EMG_HD void gs_point_node(const Level<T> &L, const T *pst, const E &ed, int ix, int iy, int iz)
{
    PointIn<T> in;
    const ZetaGlobal<T> zg{L};
    if (pst) point_load<T, true>(L, pst, zg, ix, iy, iz, in);
    else point_load<T, false>(L, pst, zg, ix, iy, iz, in);
    point_update<T, E>(L, in, ed, ix, iy, iz);
}
template <class T> EMG_HD void gs_point_node(const Level<T> &L, const T *pst, int ix, int iy, int iz)
{
    gs_point_node<T, EdgesGlobal<T>>(L, pst, EdgesGlobal<T>(L), ix, iy, iz);
}

// ---------------------------------------------------------------------------------------
// Line smoother (core.gauss_seidel_x/_y/_z, reference emg3d/core.py:506-1348).
//
// One line along axis a0 at transverse node (i1,i2), 1 <= i1 <= n1-1, 1 <= i2 <= n2-1.
// Block k = 0..n0-1 holds the unknowns
//     [ E0(k,i1,i2) ;  E1(k+1,i1-1,i2), E1(k+1,i1,i2) ;  E2(k+1,i1,i2-1), E2(k+1,i1,i2) ]
// (the last block only E0). For DIR=0 this is exactly the reference's x-line ordering
// (core.py:678,733,775-783); for DIR=1,2 it is the cyclic image of it, which is the same
// linear system as the reference's y-/z-line system with unknowns 1,2 <-> 3,4 exchanged.
//
// The line matrix is block tridiagonal: diagonal blocks M_k (5x5 complex symmetric,
// `middle`), sub-diagonal blocks B_k (real, only first row and diagonal non-zero, `left`).
// It depends on eta, zeta and h only -- NOT on the field -- so its block factorisation
//     S_0 = M_0,   S_k = M_k - B_k S_{k-1}^{-1} B_k^T,   S_k = C_k D_k C_k^T  (LDL^T, no pivoting)
// is computed ONCE per level and direction (line_setup) and kept in HBM (MI355X has the
// capacity: 304 B per cell and direction) in the form of the explicit inverses
// T_k = S_k^{-1} (obtained from the LDL^T factors by five triangular solves); it is the
// factorisation core.solve (core.py:1481-1616) performs on every call, in block form.
// Storing T_k instead of (C_k, D_k) turns the two dependent triangular solves per block
// into five independent dot products -- the sequential recurrence along the line is
// latency-bound, so the depth of the dependency chain per block is what matters.
// A smoothing sweep then is
//     (1) rhs_k  : source + terms of edges NOT on the line          (parallel, line_rhs)
//     (2) forward: c_k = rhs_k - B_k w_{k-1},  w_k = T_k c_k         (per line)
//     (3) backward: x_k = w_k - T_k B_{k+1}^T x_{k+1}                (per line)
//     (4) scatter x into the field                                   (parallel, line_scatter)
// The reference's `amat`/`bvec`/`middle`/`left` arrays (core.py:586-593) never exist.
//
// Storage: one record per (block k, line lid) of a colour class with `nlines` lines,
// block-major so that a line's consecutive blocks are `nlines` records apart and the
// lines of a wave sit next to each other:
//   fac [(k*nlines + lid)*15 + j]   T_k = S_k^{-1}, symmetric, lower triangle packed
//                                   row-major: T(r,m), m <= r, at j = r(r+1)/2 + m
//   lfac[(k*nlines + lid)*8  + j]   j = 0..3: B_k(0, m), m = 1..4; j = 4..7: B_k(m, m)
//   vec [(k*nlines + lid)*5  + r]   rhs -> w -> x in place
// The forward/backward kernels stream these records with FOUR lanes per line (each lane
// loads a quarter of a record, the quad exchanges it through DPP), because a wave that
// serves 64 lines cannot keep enough bytes in flight to hide HBM latency along the
// sequential block recurrence.
// ---------------------------------------------------------------------------------------
EMG_HD constexpr int tri(int r, int m) { return r * (r - 1) / 2 + m; }   // index of C(r,m), m<r

// Matrix part of block k (core.py:638-721 in the abstract axes): diagonal dg[5], strictly
// lower real part mid[r][m] (m<r), coupling to the previous block left0[m] = B(0,m),
// leftd[m] = B(m,m), m = 1..4.
template <class T, int DIR>
EMG_HD void line_matrix(const Axes<T, DIR> &A, int k, int i1, int i2, T (&dg)[5], double (&mid)[5][5],
                        double (&left0)[5], double (&leftd)[5])
{
    const int n0 = A.n0();
    const int i0m = k;                                  // "ixm": minus index along the line
    const int i0 = (k + 1 < n0 - 1) ? k + 1 : n0 - 1;   // "ix" clamped (core.py:635)
    const int i1m = i1 - 1, i2m = i2 - 1;
    const double h00 = A.ih0()[i0m], h01 = A.ih0()[i0];
    const double h10 = A.ih1()[i1m], h11 = A.ih1()[i1];
    const double h20 = A.ih2()[i2m], h21 = A.ih2()[i2];
    const double k00 = 0.5 * h00, k01 = 0.5 * h01, k10 = 0.5 * h10, k11 = 0.5 * h11;
    const double k20 = 0.5 * h20, k21 = 0.5 * h21;

    const double z000 = A.zeta(i0m, i1m, i2m), z100 = A.zeta(i0, i1m, i2m);
    const double z010 = A.zeta(i0m, i1, i2m), z110 = A.zeta(i0, i1, i2m);
    const double z001 = A.zeta(i0m, i1m, i2), z101 = A.zeta(i0, i1m, i2);
    const double z011 = A.zeta(i0m, i1, i2), z111 = A.zeta(i0, i1, i2);

    // face averages with (x,y,z) read as (a0,a1,a2); the four "xp" ones are not needed
    const double mzyLxm = k10 * (z001 + z000), mzyRxm = k11 * (z011 + z010);
    const double myzLxm = k20 * (z010 + z000), myzRxm = k21 * (z011 + z001);
    const double mzxLym = k00 * (z001 + z000), mzxRym = k01 * (z101 + z100);
    const double mxzLym = k20 * (z100 + z000), mxzRym = k21 * (z101 + z001);
    const double mzxLyp = k00 * (z011 + z010), mzxRyp = k01 * (z111 + z110);
    const double mxzLyp = k20 * (z110 + z010), mxzRyp = k21 * (z111 + z011);
    const double myxLzm = k00 * (z010 + z000), myxRzm = k01 * (z110 + z100);
    const double mxyLzm = k10 * (z100 + z000), mxyRzm = k11 * (z110 + z010);
    const double myxLzp = k00 * (z011 + z001), myxRzp = k01 * (z111 + z101);
    const double mxyLzp = k10 * (z101 + z001), mxyRzp = k11 * (z111 + z011);

    // eta sums (core.py:665-678)
    const T st0 = A.eta(0, i0m, i1, i2) + A.eta(0, i0m, i1, i2m) + A.eta(0, i0m, i1m, i2) +
                  A.eta(0, i0m, i1m, i2m);
    const T st2 = A.eta(1, i0, i1m, i2) + A.eta(1, i0, i1m, i2m) + A.eta(1, i0m, i1m, i2) +
                  A.eta(1, i0m, i1m, i2m);
    const T st3 = A.eta(1, i0, i1, i2) + A.eta(1, i0, i1, i2m) + A.eta(1, i0m, i1, i2) +
                  A.eta(1, i0m, i1, i2m);
    const T st4 = A.eta(2, i0, i1, i2m) + A.eta(2, i0, i1m, i2m) + A.eta(2, i0m, i1, i2m) +
                  A.eta(2, i0m, i1m, i2m);
    const T st5 = A.eta(2, i0, i1, i2) + A.eta(2, i0, i1m, i2) + A.eta(2, i0m, i1, i2) +
                  A.eta(2, i0m, i1m, i2);

    // diagonal of `middle` (core.py:683-697)
    dg[0] = (mzyRxm * h11 + mzyLxm * h10 + myzRxm * h21 + myzLxm * h20) - 0.25 * st0;
    dg[1] = (mzxRym * h01 + mzxLym * h00 + mxzRym * h21 + mxzLym * h20) - 0.25 * st2;
    dg[2] = (mzxRyp * h01 + mzxLyp * h00 + mxzRyp * h21 + mxzLyp * h20) - 0.25 * st3;
    dg[3] = (myxRzm * h01 + myxLzm * h00 + mxyRzm * h11 + mxyLzm * h10) - 0.25 * st4;
    dg[4] = (myxRzp * h01 + myxLzp * h00 + mxyRzp * h11 + mxyLzp * h10) - 0.25 * st5;

    // strictly lower part of `middle` (core.py:704-711); (2,1) and (4,3) are zero
    mid[1][0] = -mzyLxm * h00; mid[2][0] = mzyRxm * h00; mid[3][0] = -myzLxm * h00; mid[4][0] = myzRxm * h00;
    mid[2][1] = 0.0;
    mid[3][1] = -mxzLym * h10; mid[4][1] = mxzRym * h10;
    mid[3][2] = mxzLyp * h11; mid[4][2] = -mxzRyp * h11;
    mid[4][3] = 0.0;

    // `left` (core.py:714-721): first row and diagonal
    left0[0] = 0.0;
    left0[1] = mzyLxm * h00; left0[2] = -mzyRxm * h00; left0[3] = myzLxm * h00; left0[4] = -myzRxm * h00;
    leftd[0] = 0.0;
    leftd[1] = -mzxLym * h00; leftd[2] = -mzxLyp * h00; leftd[3] = -myxLzm * h00; leftd[4] = -myxLzp * h00;
}

// Right-hand side of block k (core.py:727-766): source + terms of the edges that are NOT
// on the line, with the current field values. Two parts, so that callers can take them from
// different record rows (the mirrored blocks of the two-sided solve, below: E0(k) with t(k)):
//   line_rhs_e0: entry 0, the along-line edge E0(k)                              (core.py:727, 735-738)
//   line_rhs_t : entries 1..4, the transverse edges t(k+1) at node k+1           (core.py:729-732, 740-766);
//                zero for the last block k = n0 - 1, which has only the along-line edge
// Every entry is evaluated by the same expression, in the same order, wherever it is called from.
template <class T, int DIR>
EMG_HD T line_rhs_e0(const Axes<T, DIR> &A, int k, int i1, int i2)
{
    const int i0m = k;
    const int i1m = i1 - 1, i1p = i1 + 1, i2m = i2 - 1, i2p = i2 + 1;
    const double h10 = A.ih1()[i1m], h11 = A.ih1()[i1];
    const double h20 = A.ih2()[i2m], h21 = A.ih2()[i2];
    const double k10 = 0.5 * h10, k11 = 0.5 * h11, k20 = 0.5 * h20, k21 = 0.5 * h21;
    const double z000 = A.zeta(i0m, i1m, i2m), z010 = A.zeta(i0m, i1, i2m);
    const double z001 = A.zeta(i0m, i1m, i2), z011 = A.zeta(i0m, i1, i2);
    const double mzyLxm = k10 * (z001 + z000), mzyRxm = k11 * (z011 + z010);
    const double myzLxm = k20 * (z010 + z000), myzRxm = k21 * (z011 + z001);
    T r0 = A.s(0, i0m, i1, i2);
    r0 += (mzyRxm * h11) * A.e(0, i0m, i1p, i2);
    r0 += (mzyLxm * h10) * A.e(0, i0m, i1m, i2);
    r0 += (myzRxm * h21) * A.e(0, i0m, i1, i2p);
    r0 += (myzLxm * h20) * A.e(0, i0m, i1, i2m);
    return r0;
}
template <class T, int DIR>
EMG_HD void line_rhs_t(const Axes<T, DIR> &A, int k, int i1, int i2, T (&rhs)[5])
{
    const int n0 = A.n0();
    const int i0m = k;
    const int i0 = (k + 1 < n0 - 1) ? k + 1 : n0 - 1;
    const int i1m = i1 - 1, i1p = i1 + 1, i2m = i2 - 1, i2p = i2 + 1;
    const double h00 = A.ih0()[i0m], h01 = A.ih0()[i0];
    const double h10 = A.ih1()[i1m], h11 = A.ih1()[i1];
    const double h20 = A.ih2()[i2m], h21 = A.ih2()[i2];
    const double k00 = 0.5 * h00, k01 = 0.5 * h01, k10 = 0.5 * h10, k11 = 0.5 * h11;
    const double k20 = 0.5 * h20, k21 = 0.5 * h21;

    const double z000 = A.zeta(i0m, i1m, i2m), z010 = A.zeta(i0m, i1, i2m);
    const double z001 = A.zeta(i0m, i1m, i2), z011 = A.zeta(i0m, i1, i2);
    // (the last block has only the along-line edge: its entries 1..4 are zeroed at the end --
    // no early return, so that callers can batch several blocks with all loads in flight)
    const double z100 = A.zeta(i0, i1m, i2m), z110 = A.zeta(i0, i1, i2m);
    const double z101 = A.zeta(i0, i1m, i2), z111 = A.zeta(i0, i1, i2);
    const double mzxLym = k00 * (z001 + z000), mzxRym = k01 * (z101 + z100);
    const double mxzLym = k20 * (z100 + z000), mxzRym = k21 * (z101 + z001);
    const double mzxLyp = k00 * (z011 + z010), mzxRyp = k01 * (z111 + z110);
    const double mxzLyp = k20 * (z110 + z010), mxzRyp = k21 * (z111 + z011);
    const double myxLzm = k00 * (z010 + z000), myxRzm = k01 * (z110 + z100);
    const double mxyLzm = k10 * (z100 + z000), mxyRzm = k11 * (z110 + z010);
    const double myxLzp = k00 * (z011 + z001), myxRzp = k01 * (z111 + z101);
    const double mxyLzp = k10 * (z101 + z001), mxyRzp = k11 * (z111 + z011);

    rhs[1] = A.s(1, i0, i1m, i2);
    rhs[2] = A.s(1, i0, i1, i2);
    rhs[3] = A.s(2, i0, i1, i2m);
    rhs[4] = A.s(2, i0, i1, i2);

    rhs[1] += h10 * (mzxRym * A.e(0, i0, i1m, i2) - mzxLym * A.e(0, i0m, i1m, i2) +
                     mxzRym * A.e(2, i0, i1m, i2) - mxzLym * A.e(2, i0, i1m, i2m));
    rhs[1] += (mxzRym * h21) * A.e(1, i0, i1m, i2p);
    rhs[1] += (mxzLym * h20) * A.e(1, i0, i1m, i2m);

    rhs[2] += h11 * (mzxLyp * A.e(0, i0m, i1p, i2) - mzxRyp * A.e(0, i0, i1p, i2) +
                     mxzLyp * A.e(2, i0, i1p, i2m) - mxzRyp * A.e(2, i0, i1p, i2));
    rhs[2] += (mxzRyp * h21) * A.e(1, i0, i1, i2p);
    rhs[2] += (mxzLyp * h20) * A.e(1, i0, i1, i2m);

    rhs[3] += h20 * (myxRzm * A.e(0, i0, i1, i2m) - myxLzm * A.e(0, i0m, i1, i2m) +
                     mxyRzm * A.e(1, i0, i1, i2m) - mxyLzm * A.e(1, i0, i1m, i2m));
    rhs[3] += (mxyRzm * h11) * A.e(2, i0, i1p, i2m);
    rhs[3] += (mxyLzm * h10) * A.e(2, i0, i1m, i2m);

    rhs[4] += h21 * (myxLzp * A.e(0, i0m, i1, i2p) - myxRzp * A.e(0, i0, i1, i2p) +
                     mxyLzp * A.e(1, i0, i1m, i2p) - mxyRzp * A.e(1, i0, i1, i2p));
    rhs[4] += (mxyRzp * h11) * A.e(2, i0, i1p, i2);
    rhs[4] += (mxyLzp * h10) * A.e(2, i0, i1m, i2);
    if (k == n0 - 1) rhs[1] = rhs[2] = rhs[3] = rhs[4] = zero<T>();
}
template <class T, int DIR>
EMG_HD void line_rhs(const Axes<T, DIR> &A, int k, int i1, int i2, T (&rhs)[5])
{
    line_rhs_t<T, DIR>(A, k, i1, i2, rhs);
    rhs[0] = line_rhs_e0<T, DIR>(A, k, i1, i2);
}


// The eight coupling entries of record k of a line's `lfac` (line_setup: put_B), recomputed from zeta and the
// widths: c[m-1] = first row, c[3+m] = diagonal, m = 1..4. Records 1 .. m of the top half hold B_k (left0 /
// leftd of line_matrix(k)), records >= m + 1 of the bottom half U_k of the mirrored block (mid[.][0] / leftd of
// line_matrix(k); zero for the last block n0 - 1), record 0 and the identity padding blocks behind n0 - 1
// zeros. Products only, evaluated as line_matrix evaluates them: the same bits as the stored records.
template <class T, int DIR>
EMG_HD void line_coupling(const Axes<T, DIR> &A, int k, int i1, int i2, bool mirrored, double (&c)[8])
{
    const int n0 = A.n0();
    const int kk = k < 0 ? 0 : (k > n0 - 1 ? n0 - 1 : k);
    const int i1m = i1 - 1, i2m = i2 - 1;
    const double h00 = A.ih0()[kk];
    const double h10 = A.ih1()[i1m], h11 = A.ih1()[i1];
    const double h20 = A.ih2()[i2m], h21 = A.ih2()[i2];
    const double k00 = 0.5 * h00, k10 = 0.5 * h10, k11 = 0.5 * h11, k20 = 0.5 * h20, k21 = 0.5 * h21;
    const double z000 = A.zeta(kk, i1m, i2m), z010 = A.zeta(kk, i1, i2m);
    const double z001 = A.zeta(kk, i1m, i2), z011 = A.zeta(kk, i1, i2);
    const double mzyLxm = k10 * (z001 + z000), mzyRxm = k11 * (z011 + z010);
    const double myzLxm = k20 * (z010 + z000), myzRxm = k21 * (z011 + z001);
    const double mzxLym = k00 * (z001 + z000), mzxLyp = k00 * (z011 + z010);
    const double myxLzm = k00 * (z010 + z000), myxLzp = k00 * (z011 + z001);
    const bool any = mirrored ? (k >= 1 && k < n0 - 1) : (k >= 1 && k <= n0 - 1);
    const double a0 = mzyLxm * h00, a1 = mzyRxm * h00, a2 = myzLxm * h00, a3 = myzRxm * h00;
    // top: B_k(0, m) = (+, -, +, -); mirrored: U_k(0, m) = M_k(m, 0) = the negatives
    c[0] = any ? (mirrored ? -a0 : a0) : 0.0;
    c[1] = any ? (mirrored ? a1 : -a1) : 0.0;
    c[2] = any ? (mirrored ? -a2 : a2) : 0.0;
    c[3] = any ? (mirrored ? a3 : -a3) : 0.0;
    c[4] = any ? -(mzxLym * h00) : 0.0;
    c[5] = any ? -(mzxLyp * h00) : 0.0;
    c[6] = any ? -(myxLzm * h00) : 0.0;
    c[7] = any ? -(myxLzp * h00) : 0.0;
}

// In-register LDL^T of a symmetric 5x5 block given by its lower triangle S (S[r][m], m<=r):
// C[tri(r,m)] (m<r) and dinv[r]. nrows < 5 factorises the leading nrows x nrows part.
template <class T> EMG_HD void ldlt5(const T (&S)[5][5], int nrows, T (&C)[10], T (&dinv)[5])
{
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        if (r < nrows) {
            T u[5];
#pragma unroll
            for (int m = 0; m < r; ++m) {
                T t = S[r][m];
#pragma unroll
                for (int kk = 0; kk < m; ++kk) t -= u[kk] * C[tri(m, kk)];
                u[m] = t;
                C[tri(r, m)] = t * dinv[m];
            }
            T d = S[r][r];
#pragma unroll
            for (int kk = 0; kk < r; ++kk) d -= u[kk] * C[tri(r, kk)];
            dinv[r] = recip(d);
        } else {
            dinv[r] = T(1.0);
#pragma unroll
            for (int m = 0; m < r; ++m) C[tri(r, m)] = zero<T>();
        }
    }
}

// v <- S^{-1} v with the LDL^T factors (C, dinv) of S.
template <class T> EMG_HD void ldlt5_solve(const T (&C)[10], const T (&dinv)[5], T (&v)[5])
{
#pragma unroll
    for (int r = 1; r < 5; ++r) {
#pragma unroll
        for (int m = 0; m < r; ++m) v[r] -= C[tri(r, m)] * v[m];
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) v[r] *= dinv[r];
#pragma unroll
    for (int m = 3; m >= 0; --m) {
#pragma unroll
        for (int r = m + 1; r < 5; ++r) v[m] -= C[tri(r, m)] * v[r];
    }
}

// ---- two-sided ("twisted") block factorisation -----------------------------------------
// The recurrence along a line is sequential, and on the GPU its length is what a sweep
// costs: one wave serves 16 lines, a colour class has far fewer waves than the chip has
// SIMDs, and every wave is bound by the issue rate / latency of its own chain. The line is
// therefore eliminated from BOTH ends by two waves working at the same time.
//
// Unknowns along a line: E0(k), k = 0..n0-1 (the edges on the line) and t(k), k = 1..n0-1
// (the four transverse edges at node k). The reference's block k is {E0(k), t(k+1)}
// ("standard" grouping; the last block has E0 only). Eliminating standard blocks from the
// far end is numerically poor -- the near-null (gradient) direction of such a trailing
// Schur complement is spread over all five unknowns and products with its inverse cancel --
// so the far half uses the MIRRORED grouping {E0(k), t(k)}, for which the elimination from
// the far end is the exact mirror image of the reference's from the near end:
//
//   top     k = 0 .. m-1      standard blocks {E0(k), t(k+1)},  S_k = M_k - B_k T_{k-1} B_k^T
//   bottom  k = n0-1 .. m+2   mirrored blocks {E0(k), t(k)},    S_k = M'_k - U_k T_{k+1} U_k^T
//   middle  Q = {E0(m), t(m+1), E0(m+1)}  (standard block m merged with mirrored block m+1,
//           which share t(m+1)), 6 x 6:  S_Q = M_Q - [B_m T_{m-1} B_m^T] - [U_{m+1} T_{m+2} U_{m+1}^T]
// with T = S^{-1}; M'_k(0,0) = M_k(0,0), M'_k(0,j) = B_k(0,j), M'_k(a,b) = M_{k-1}(a,b) and
// the coupling of mirrored block k to mirrored block k+1  U_k = e0 u^T + diag(0, d),
// u_j = M_k(0,j), d_j = B_k(j,j) -- the same shape as B_k. A solve is
//   forward:  top     w_k = T_k (r_k - B_k w_{k-1}),        k = 0 .. m-1
//             bottom  w_k = T_k (r'_k - U_k w_{k+1}),       k = n0-1 .. m+2
//   middle:   x_Q = T_Q (r_Q - [B_m w_{m-1}] - [U_{m+1} w_{m+2}])
//   backward: top     x_k = w_k - T_k B_{k+1}^T x_{k+1},    k = m-1 .. 0
//             bottom  x_k = w_k - T_k U_{k-1}^T x_{k-1},    k = m+2 .. n0-1
// i.e. two chains of half the length per pass, both of the reference's top-down form. It
// is the same direct solve of the line system as core.solve's LDL^T
// (emg3d/core.py:1481-1616), in a different elimination order.
//
// Records (block-major, record k of line lid at k*nlines + lid):
//   fac : k < m: T_k; k >= m+2: T_k of the mirrored block; records m and m+1 together hold
//         the 21 packed entries of T_Q (15 + 6)
//   lfac: k <= m: B_k; k >= m+1: U_k        (j = 0..3: first row, j = 4..7: diagonal)
//   vec : slot (k, 0) belongs to E0(k), slot (k, j) to t(k+1)_j -- whatever the grouping
// m = line_mid(n0) is a multiple of P = line_pad(n0); behind block n0-1 the bottom half is padded
// with identity blocks (T = 1, U = 0, rhs = 0) to a multiple of P, so that both half-walks run a
// P-times unrolled, branch-free software pipeline. P = LINE_PAD = 4 blocks in flight per line;
// lines of at most LINE_SHORT blocks use P = 2: with four, a 4-block line (coarse levels of a
// semicoarsened hierarchy have thousands of launches of those) would walk 2 real + 2 identity
// blocks in one half and none in the other.
constexpr int LINE_PAD = 4, LINE_PAD_SHORT = 2, LINE_SHORT = 6;
EMG_HD int line_pad(int n0) { return n0 <= LINE_SHORT ? LINE_PAD_SHORT : LINE_PAD; }
EMG_HD int line_mid(int n0)
{
    const int p = line_pad(n0);
    return (n0 / 2) / p * p;
}
EMG_HD int line_padded(int n0)
{
    const int m = line_mid(n0), p = line_pad(n0);
    return m + 2 + (n0 - 2 - m + p - 1) / p * p;
}

// packed index of T(r,m) = T(m,r)
EMG_HD constexpr int sym(int r, int m) { return r >= m ? r * (r + 1) / 2 + m : m * (m + 1) / 2 + r; }

// S -= B T B^T, T = (C D C^T)^{-1}, B = e0 l0^T + diag(0, d1..d4)  (lower triangle of S)
template <class T>
EMG_HD void sub_lower_coupling(T (&S)[5][5], const T (&C)[10], const T (&dinv)[5], const double (&left0)[5],
                               const double (&leftd)[5])
{
    // columns of T that are needed: T l0 and T e_m (m=1..4) -> 5 solves
    T tl[5];                              // T l0
    tl[0] = zero<T>();
#pragma unroll
    for (int m = 1; m < 5; ++m) tl[m] = T(left0[m]);
    ldlt5_solve<T>(C, dinv, tl);
    T s00 = zero<T>();
#pragma unroll
    for (int m = 1; m < 5; ++m) s00 += left0[m] * tl[m];
    S[0][0] -= s00;
#pragma unroll
    for (int r = 1; r < 5; ++r) S[r][0] -= leftd[r] * tl[r];       // (D T l0)_r
    // D T D, lower triangle: columns T e_m
#pragma unroll
    for (int m = 1; m < 5; ++m) {
        T col[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) col[r] = (r == m) ? T(1.0) : zero<T>();
        ldlt5_solve<T>(C, dinv, col);
#pragma unroll
        for (int r = m; r < 5; ++r) S[r][m] -= (leftd[r] * leftd[m]) * col[r];
    }
}
// explicit inverse (packed symmetric) from the LDL^T factors: five solves
template <class T> EMG_HD void invert5(const T (&C)[10], const T (&dinv)[5], T (&Tp)[15])
{
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        T col[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) col[r] = (r == m) ? T(1.0) : zero<T>();
        ldlt5_solve<T>(C, dinv, col);
#pragma unroll
        for (int r = m; r < 5; ++r) Tp[tri(r + 1, m)] = col[r];
    }
}
// inverse of a dense complex-symmetric 6 x 6 (lower triangle of S given) by LDL^T without
// pivoting, packed symmetric result Tq[sym(r,m)]
template <class T> EMG_HD void invert6(const T (&S)[6][6], T (&Tq)[21])
{
    // fully unrolled: every array index is a compile-time constant, so that the arrays live
    // in registers (a kernel that needs scratch memory stalls the whole queue while the
    // runtime resizes it -- measured 74 ms)
    T Lm[6][6], dinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T u[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j < i) {
                T t = S[i][j];
#pragma unroll
                for (int kk = 0; kk < 6; ++kk)
                    if (kk < j) t -= u[kk] * Lm[j][kk];
                u[j] = t;
                Lm[i][j] = t * dinv[j];
            }
        }
        T d = S[i][i];
#pragma unroll
        for (int kk = 0; kk < 6; ++kk)
            if (kk < i) d -= u[kk] * Lm[i][kk];
        dinv[i] = recip(d);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        T b[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) b[r] = (r == m) ? T(1.0) : zero<T>();
#pragma unroll
        for (int i = 1; i < 6; ++i)
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
                if (kk < i) b[i] -= Lm[i][kk] * b[kk];
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] *= dinv[i];
#pragma unroll
        for (int j = 4; j >= 0; --j)
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
                if (kk > j) b[j] -= Lm[kk][j] * b[kk];
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if (r >= m) Tq[sym(r, m)] = b[r];
    }
}

// Matrix of the MIRRORED block k = {E0(k), t(k)} and its coupling U_k to mirrored block k+1,
// from the standard quantities of blocks k and k-1 (see above). 1 <= k <= n0-1.
template <class T, int DIR>
EMG_HD void line_matrix_mirrored(const Axes<T, DIR> &A, int k, int i1, int i2, T (&S)[5][5], double (&u0)[5],
                                 double (&ud)[5])
{
    T dg[5], dgp[5];
    double mid[5][5], midp[5][5], left0[5], leftd[5], l0p[5], ldp[5];
    line_matrix<T, DIR>(A, k, i1, i2, dg, mid, left0, leftd);
    line_matrix<T, DIR>(A, k - 1, i1, i2, dgp, midp, l0p, ldp);
    const bool last = k == A.n0() - 1;
    S[0][0] = dg[0];
#pragma unroll
    for (int a = 1; a < 5; ++a) {
        S[a][0] = T(left0[a]);
        S[a][a] = dgp[a];
#pragma unroll
        for (int b = 1; b < a; ++b) S[a][b] = T(midp[a][b]);
        u0[a] = last ? 0.0 : mid[a][0];
        ud[a] = last ? 0.0 : leftd[a];
    }
    u0[0] = 0.0;
    ud[0] = 0.0;
}

// Setup of one line: two-sided block factorisation, stored in (fac, lfac), once per level
// and direction. The two chains are independent (line_setup_top / line_setup_bottom: the HIP
// kernel runs them in two waves) and meet in line_setup_middle, which needs the LDL^T
// factors (C, dinv) of the last block of either chain.
// FT: the storage type of the T records -- T itself, or compact_of<T> (cplx.h: rounded to single precision when
// stored, "compact line factors"); the factorisation itself is carried in T whatever the storage
template <class T, class FT = T> struct LineStore {
    FT *fac;
    double *lfac;
    int nlines, lid;
    EMG_HD void put_T(int k, const T (&Tp)[15]) const
    {
        FT *f = fac + ((size_t)k * nlines + lid) * 15;
#pragma unroll
        for (int j = 0; j < 15; ++j) f[j] = narrow<FT>(Tp[j]);
    }
    EMG_HD void put_B(int k, const double (&b0)[5], const double (&bd)[5], bool any) const
    {
        double *lf = lfac + ((size_t)k * nlines + lid) * 8;
#pragma unroll
        for (int m = 1; m < 5; ++m) {
            lf[m - 1] = any ? b0[m] : 0.0;
            lf[3 + m] = any ? bd[m] : 0.0;
        }
    }
};
template <class T> EMG_HD void std_block(const T (&dg)[5], const double (&mid)[5][5], T (&S)[5][5])
{
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        S[r][r] = dg[r];
#pragma unroll
        for (int m = 0; m < r; ++m) S[r][m] = T(mid[r][m]);
    }
}

// top chain: standard blocks k = 0 .. m-1; (C, dinv) = factors of S_{m-1} (untouched if m = 0)
template <class T, int DIR, class FT = T>
EMG_HD void line_setup_top(const Level<T> &L, int i1, int i2, const LineStore<T, FT> &st, T (&C)[10], T (&dinv)[5])
{
    const Axes<T, DIR> A(L);
    const int mk = line_mid(A.n0());
    T dg[5];
    double mid[5][5], left0[5], leftd[5];
    for (int k = 0; k < mk; ++k) {
        line_matrix<T, DIR>(A, k, i1, i2, dg, mid, left0, leftd);
        T S[5][5];
        std_block<T>(dg, mid, S);
        if (k > 0) sub_lower_coupling<T>(S, C, dinv, left0, leftd);
        ldlt5<T>(S, 5, C, dinv);
        T Tp[15];
        invert5<T>(C, dinv, Tp);
        st.put_T(k, Tp);
        st.put_B(k, left0, leftd, k > 0);
    }
}
// bottom chain: mirrored blocks k = n0-1 .. m+2; (Cb, db) = factors of S_{m+2} (untouched if
// the chain is empty); also writes the identity padding blocks behind block n0-1
template <class T, int DIR, class FT = T>
EMG_HD void line_setup_bottom(const Level<T> &L, int i1, int i2, const LineStore<T, FT> &st, int n0p, T (&Cb)[10],
                              T (&db)[5])
{
    const Axes<T, DIR> A(L);
    const int n0 = A.n0();
    const int mk = line_mid(n0);
    double u0[5], ud[5];
    for (int k = n0 - 1; k >= mk + 2; --k) {
        T S[5][5];
        line_matrix_mirrored<T, DIR>(A, k, i1, i2, S, u0, ud);
        if (k < n0 - 1) sub_lower_coupling<T>(S, Cb, db, u0, ud);
        ldlt5<T>(S, 5, Cb, db);
        T Tp[15];
        invert5<T>(Cb, db, Tp);
        st.put_T(k, Tp);
        st.put_B(k, u0, ud, true);
    }
    for (int k = n0; k < n0p; ++k) {
        FT *f = st.fac + ((size_t)k * st.nlines + st.lid) * 15;
        double *lf = st.lfac + ((size_t)k * st.nlines + st.lid) * 8;
        for (int r = 0; r < 5; ++r)
            for (int m = 0; m <= r; ++m) f[tri(r + 1, m)] = narrow<FT>((r == m) ? T(1.0) : zero<T>());
        for (int j = 0; j < 8; ++j) lf[j] = 0.0;
    }
}
// middle block Q = {E0(m), t(m+1), E0(m+1)}
template <class T, int DIR, class FT = T>
EMG_HD void line_setup_middle(const Level<T> &L, int i1, int i2, const LineStore<T, FT> &st, const T (&C)[10],
                              const T (&dinv)[5], const T (&Cb)[10], const T (&db)[5])
{
    const Axes<T, DIR> A(L);
    const int n0 = A.n0();
    const int mk = line_mid(n0);
    T SQ[6][6];
    {   // standard part: S_m = M_m - B_m T_{m-1} B_m^T  (rows/cols 0..4 of S_Q)
        T dg[5];
        double mid[5][5], left0[5], leftd[5];
        line_matrix<T, DIR>(A, mk, i1, i2, dg, mid, left0, leftd);
        T S[5][5];
        std_block<T>(dg, mid, S);
        if (mk > 0) sub_lower_coupling<T>(S, C, dinv, left0, leftd);
        st.put_B(mk, left0, leftd, mk > 0);
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int m = 0; m <= r; ++m) SQ[r][m] = S[r][m];
    }
    {   // mirrored part: block m+1 = {E0(m+1), t(m+1)} -> Q indices {5, 1..4}
        T S[5][5];
        double u0[5], ud[5];
        line_matrix_mirrored<T, DIR>(A, mk + 1, i1, i2, S, u0, ud);
        st.put_B(mk + 1, u0, ud, true);
        // only what block m has not contributed: E0(m+1) row/column (t(m+1) x t(m+1) is M_m's)
        T Sc[5][5];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int m = 0; m <= r; ++m) Sc[r][m] = zero<T>();
        Sc[0][0] = S[0][0];
#pragma unroll
        for (int a = 1; a < 5; ++a) Sc[a][0] = S[a][0];
        if (mk + 1 < n0 - 1) sub_lower_coupling<T>(Sc, Cb, db, u0, ud);
        SQ[5][5] = Sc[0][0];
        SQ[5][0] = zero<T>();                    // E0(m+1) and E0(m) are not coupled
#pragma unroll
        for (int a = 1; a < 5; ++a) {
            SQ[5][a] = Sc[a][0];
#pragma unroll
            for (int b = 1; b <= a; ++b) SQ[a][b] += Sc[a][b];
        }
    }
    T Tq[21];
    invert6<T>(SQ, Tq);
    FT *f = st.fac + ((size_t)mk * st.nlines + st.lid) * 15;
    FT *g = st.fac + ((size_t)(mk + 1) * st.nlines + st.lid) * 15;
#pragma unroll
    for (int j = 0; j < 15; ++j) f[j] = narrow<FT>(Tq[j]);
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = narrow<FT>(Tq[15 + j]);
#pragma unroll
    for (int j = 6; j < 15; ++j) g[j] = narrow<FT>(zero<T>());
}
// all of it by one thread (CPU emulation of the unit tests)
template <class T, int DIR, class FT = T>
EMG_HD void line_setup(const Level<T> &L, int i1, int i2, FT *fac, double *lfac, int nlines, int lid, int n0p)
{
    const LineStore<T, FT> st{fac, lfac, nlines, lid};
    T C[10], dinv[5], Cb[10], db[5];
#pragma unroll
    for (int j = 0; j < 10; ++j) C[j] = Cb[j] = zero<T>();
#pragma unroll
    for (int j = 0; j < 5; ++j) dinv[j] = db[j] = T(1.0);
    line_setup_top<T, DIR, FT>(L, i1, i2, st, C, dinv);
    line_setup_bottom<T, DIR, FT>(L, i1, i2, st, n0p, Cb, db);
    line_setup_middle<T, DIR, FT>(L, i1, i2, st, C, dinv, Cb, db);
}

// q = B y  /  q = B^T y  for B = e0 l0^T + diag(0, ld)
template <class T> EMG_HD void couple_lower(const double (&l0)[4], const double (&ld)[4], const T (&y)[5], T (&q)[5])
{
    T v0 = zero<T>();
#pragma unroll
    for (int m = 1; m < 5; ++m) v0 += l0[m - 1] * y[m];
    q[0] = v0;
#pragma unroll
    for (int m = 1; m < 5; ++m) q[m] = ld[m - 1] * y[m];
}
template <class T> EMG_HD void couple_upper(const double (&l0)[4], const double (&ld)[4], const T (&y)[5], T (&q)[5])
{
    q[0] = zero<T>();
#pragma unroll
    for (int m = 1; m < 5; ++m) q[m] = l0[m - 1] * y[0] + ld[m - 1] * y[m];
}
template <class T> EMG_HD void sym_matvec(const T (&Tk)[15], const T (&z)[5], T (&o)[5])
{
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        T acc = zero<T>();
#pragma unroll
        for (int m = 0; m < 5; ++m) acc += Tk[sym(r, m)] * z[m];
        o[r] = acc;
    }
}

// Reference walks of one line (one thread per line): used by the CPU emulation of the unit
// tests and as the specification of what the quad kernels in kernels.hip compute.
// vec slots of a block: standard block k -> (k,0..4); mirrored block k -> (k,0), (k-1,1..4).
// FT / WT: storage types of the T records / of the w records (T, or compact_of<T>: the compact form, in which the
// forward pass reads its right-hand sides from `rhs` (T) and stores w rounded to WT, and the backward pass writes
// the solution, which never exists in WT, to `xout` (T). Plain form: rhs = xout = vec.)
template <class T, class FT = T, class WT = T> struct LineRef {
    int nlines, lid;
    const FT *fac;
    const double *lfac;
    WT *vec;
    const T *rhs;
    T *xout;
    EMG_HD size_t rec(int k) const { return (size_t)k * nlines + lid; }
    EMG_HD size_t at(int k, int r, bool mirrored) const { return rec(mirrored && r > 0 ? k - 1 : k) * 5 + r; }
    EMG_HD T slot(int k, int r, bool mirrored) const { return widen(vec[at(k, r, mirrored)]); }
    EMG_HD void get(int k, bool mirrored, T (&v)[5]) const
    {
        for (int r = 0; r < 5; ++r) v[r] = slot(k, r, mirrored);
    }
    EMG_HD void get_rhs(int k, bool mirrored, T (&v)[5]) const
    {
        for (int r = 0; r < 5; ++r) v[r] = rhs[at(k, r, mirrored)];
    }
    EMG_HD void put(int k, bool mirrored, const T (&v)[5]) const
    {
        for (int r = 0; r < 5; ++r) vec[at(k, r, mirrored)] = narrow<WT>(v[r]);
    }
    EMG_HD void put_x(int k, bool mirrored, const T (&v)[5]) const
    {
        for (int r = 0; r < 5; ++r) xout[at(k, r, mirrored)] = v[r];
    }
    EMG_HD void T15(int k, T (&Tk)[15]) const
    {
        for (int j = 0; j < 15; ++j) Tk[j] = widen(fac[rec(k) * 15 + j]);
    }
    EMG_HD void B(int k, double (&l0)[4], double (&ld)[4]) const
    {
        for (int j = 0; j < 4; ++j) { l0[j] = lfac[rec(k) * 8 + j]; ld[j] = lfac[rec(k) * 8 + 4 + j]; }
    }
};
// forward: both half-chains; n0 = real blocks, n0p = padded records
template <class T, class FT = T, class WT = T>
EMG_HD void line_forward_ref(int n0, int n0p, int nlines, int lid, const FT *fac, const double *lfac, WT *vec,
                             const T *rhs = nullptr)
{
    // (rhs == nullptr: the plain form, right-hand sides and w records in one buffer -- FT = WT = T)
    const LineRef<T, FT, WT> R{nlines, lid, fac, lfac, vec, rhs ? rhs : reinterpret_cast<const T *>(vec), nullptr};
    const int mk = line_mid(n0);
    T w[5], q[5], z[5], v[5], Tk[15];
    double l0[4], ld[4];
    for (int half = 0; half < 2; ++half) {
        const bool mir = half == 1;
        for (int r = 0; r < 5; ++r) w[r] = zero<T>();
        // top: k = 0 .. m-1 (w_k = T_k (r_k - B_k w_{k-1})); bottom: k = n0p-1 .. m+2 with U_k
        for (int i = 0; i < (mir ? n0p - 2 - mk : mk); ++i) {
            const int k = mir ? n0p - 1 - i : i;
            R.T15(k, Tk); R.B(k, l0, ld); R.get_rhs(k, mir, v);
            couple_lower<T>(l0, ld, w, q);
            for (int r = 0; r < 5; ++r) z[r] = v[r] - q[r];
            sym_matvec<T>(Tk, z, w);
            R.put(k, mir, w);
        }
    }
}
// middle: x_Q = T_Q (r_Q - [B_m w_{m-1}] - [U_{m+1} w_{m+2}]), Q = {E0(m), t(m+1), E0(m+1)}
template <class T, class FT = T, class WT = T>
EMG_HD void line_middle(int n0, int n0p, int nlines, int lid, const FT *fac, const double *lfac, WT *vec,
                        T (&xq)[6])
{
    const LineRef<T, FT, WT> R{nlines, lid, fac, lfac, vec, nullptr, nullptr};
    const int mk = line_mid(n0);
    T z[6], q[5], y[5];
    double l0[4], ld[4];
    for (int r = 0; r < 5; ++r) z[r] = R.slot(mk, r, false);
    z[5] = R.slot(mk + 1, 0, false);
    if (mk > 0) {
        R.B(mk, l0, ld); R.get(mk - 1, false, y);
        couple_lower<T>(l0, ld, y, q);
        for (int r = 0; r < 5; ++r) z[r] -= q[r];
    }
    if (mk + 2 < n0p) {
        R.B(mk + 1, l0, ld); R.get(mk + 2, true, y);
        couple_lower<T>(l0, ld, y, q);           // mirrored block m+1: q[0] -> E0(m+1), q[j] -> t(m+1)_j
        z[5] -= q[0];
        for (int r = 1; r < 5; ++r) z[r] -= q[r];
    }
    T Tq[21];
    for (int j = 0; j < 15; ++j) Tq[j] = widen(fac[R.rec(mk) * 15 + j]);
    for (int j = 0; j < 6; ++j) Tq[15 + j] = widen(fac[R.rec(mk + 1) * 15 + j]);
    for (int r = 0; r < 6; ++r) {
        T acc = zero<T>();
        for (int m = 0; m < 6; ++m) acc += Tq[sym(r, m)] * z[m];
        xq[r] = acc;
    }
}
template <class T, class FT = T, class WT = T>
EMG_HD void line_backward_ref(int n0, int n0p, int nlines, int lid, const FT *fac, const double *lfac, WT *vec,
                              T *xout = nullptr)
{
    // (xout == nullptr: the plain form, the solution overwrites the w records -- FT = WT = T)
    const LineRef<T, FT, WT> R{nlines, lid, fac, lfac, vec, nullptr, xout ? xout : reinterpret_cast<T *>(vec)};
    const int mk = line_mid(n0);
    T xq[6], x[5], q[5], tq[5], v[5], Tk[15];
    double l0[4], ld[4];
    line_middle<T, FT, WT>(n0, n0p, nlines, lid, fac, lfac, vec, xq);
    for (int half = 0; half < 2; ++half) {
        const bool mir = half == 1;
        // the part of x_Q the half couples to: standard block m / mirrored block m+1
        x[0] = mir ? xq[5] : xq[0];
        for (int r = 1; r < 5; ++r) x[r] = xq[r];
        int kprev = mir ? mk + 1 : mk;            // record whose B / U couples to the next block
        for (int i = 0; i < (mir ? n0p - 2 - mk : mk); ++i) {
            const int k = mir ? mk + 2 + i : mk - 1 - i;
            R.T15(k, Tk); R.B(kprev, l0, ld); R.get(k, mir, v);
            couple_upper<T>(l0, ld, x, q);
            sym_matvec<T>(Tk, q, tq);
            for (int r = 0; r < 5; ++r) x[r] = v[r] - tq[r];
            R.put_x(k, mir, x);
            kprev = k;
        }
    }
    for (int r = 0; r < 5; ++r) R.xout[R.at(mk, r, false)] = xq[r];
    R.xout[R.at(mk + 1, 0, false)] = xq[5];
}

// Scatter block k of the solution into the field (core.py:775-783).
template <class T, int DIR>
EMG_HD void line_scatter(const Axes<T, DIR> &A, int k, int i1, int i2, const T *rec)
{
    const int n0 = A.n0();
    A.E(0)[A.idx(0, k, i1, i2)] = rec[0];
    if (k < n0 - 1) {
        A.E(1)[A.idx(1, k + 1, i1 - 1, i2)] = rec[1];
        A.E(1)[A.idx(1, k + 1, i1, i2)] = rec[2];
        A.E(2)[A.idx(2, k + 1, i1, i2 - 1)] = rec[3];
        A.E(2)[A.idx(2, k + 1, i1, i2)] = rec[4];
    }
}

// ---- the same line solve with SHORT dependent chains ("wide" form; kernels.hip: k_line_wide) ----------
// On the small levels of a hierarchy a colour class has fewer lines than the chip has SIMDs, and a launch
// costs what ONE wave needs to issue the instructions of its chain (a lone wave issues one instruction per
// ~5 cycles whatever its kind; the step above is ~130 of them). The coupling blocks C_k (B_k / U_k) have the
// shape e0 l0^T + diag(0, d): of the previous block's w only entries 1..4 enter the next one. So the
// recurrences can be restated in FOUR unknowns with a model-only 4 x 4 matrix, and everything else of the
// solve becomes independent per block:
//
//   forward   y_k = g_k - N_k y_kn            y_k = w_k[1..4],  g_k = (T_k r_k)[1..4],  N_k = (T_k C_k)[1..4, 1..4]
//                                             (kn = the block before k in forward order; C of the first block is 0)
//   per block c_k = r_k - C_k w_kn  (needs y_kn only),   w_k[0] = (T_k c_k)[0],   g'_kn = C_k^T w_k
//   middle    x_Q as in line_middle (needs y of the two blocks next to it), h of the two blocks next to it
//   backward  h_k = g'_k - N_kp^T h_kp        h_k = C_kp^T x_kp (entry 0 is zero),  kp = the block before k in
//                                             backward order;  (C^T T)[1..4, 1..4] = N^T because T is symmetric
//   per block x_k = T_k (c_k - h_k)
//
// The 16 entries of N_k sit in 16 lanes: one complex multiply-add per lane and a two-stage sum per step --
// within the quads (DPP quad_perm) in even steps, across the quads (DPP row_ror) in odd steps, with the
// matrix fetched transposed in odd steps, so that the result of one step already lies where the next step needs
// it. ~25 instructions per step instead of ~130. It is the same direct solve of the line system as
// core.solve (emg3d/core.py:1481-1616), with the same factors T_k; N_k is one more rounding of T_k C_k.
// The functions below are the per-block arithmetic shared by the kernel and by line_wide_ref (the CPU walk of
// the unit tests); the sums of a chain step are associated (p0 + p2) + (p1 + p3) everywhere. Every product that
// feeds a sum is an explicit fused multiply-add (cplx.h: mad / nmad), so that the compiler's contraction cannot
// round an expression differently in the single-source and the batched instantiation of the kernel.
constexpr int WIDE_N0_MAX = 64;                 // longest lines that can take the wide form
constexpr size_t WIDE_RECORDS_MAX = (size_t)1 << 19;   // ... on levels with at most this many block records per direction
EMG_HD bool line_wide_capable(int n0, size_t records) { return n0 >= 2 && n0 <= WIDE_N0_MAX && records <= WIDE_RECORDS_MAX; }

// N = (T C)[1..4, 1..4]: N[4 (a-1) + (b-1)] = T(a,0) l0[b] + T(a,b) d[b]
template <class T> EMG_HD void wide_n_record(const T (&Tk)[15], const double (&lf)[8], T (&N)[16])
{
#pragma unroll
    for (int a = 1; a < 5; ++a)
#pragma unroll
        for (int b = 1; b < 5; ++b) {
            const T first = lf[b - 1] * Tk[sym(a, 0)];
            N[4 * (a - 1) + (b - 1)] = mad(lf[3 + b], Tk[sym(a, b)], first);
        }
}
// row `row` of T z for a packed symmetric 5 x 5 (two accumulators, like the chain steps of kernels.hip)
template <class T, int NT> EMG_HD T wide_row5(const T (&Tk)[NT], int row, const T (&z)[5])
{
    const T lo = mad(Tk[sym(row, 4)], z[4], mad(Tk[sym(row, 1)], z[1], mad(Tk[sym(row, 0)], z[0], zero<T>())));
    const T hi = mad(Tk[sym(row, 3)], z[3], mad(Tk[sym(row, 2)], z[2], zero<T>()));
    return lo + hi;
}
// g = (T r)[1..4]
template <class T, int NT> EMG_HD void wide_g(const T (&Tk)[NT], const T (&r)[5], T (&g)[4])
{
#pragma unroll
    for (int a = 1; a < 5; ++a) g[a - 1] = wide_row5<T, NT>(Tk, a, r);
}
// c = r - C yp  (yp = entries 1..4 of the previous block's w)
template <class T> EMG_HD void wide_c(const double (&lf)[8], const T (&r)[5], const T (&yp)[4], T (&c)[5])
{
    T q0 = lf[0] * yp[0];
#pragma unroll
    for (int b = 1; b < 4; ++b) q0 = mad(lf[b], yp[b], q0);
    c[0] = r[0] - q0;
#pragma unroll
    for (int b = 0; b < 4; ++b) c[b + 1] = nmad(lf[4 + b], yp[b], r[b + 1]);
}
// g' = C^T w  (entries 1..4; w = (w0, y))
template <class T> EMG_HD void wide_gp(const double (&lf)[8], const T w0, const T (&y)[4], T (&gp)[4])
{
#pragma unroll
    for (int b = 0; b < 4; ++b) gp[b] = mad(lf[b], w0, lf[4 + b] * y[b]);
}
// x = T (c - (0, h))
template <class T, int NT> EMG_HD void wide_x(const T (&Tk)[NT], const T (&c)[5], const T (&h)[4], T (&x)[5])
{
    T z[5];
    z[0] = c[0];
#pragma unroll
    for (int b = 0; b < 4; ++b) z[b + 1] = c[b + 1] - h[b];
#pragma unroll
    for (int r = 0; r < 5; ++r) x[r] = wide_row5<T, NT>(Tk, r, z);
}
// middle block: x_Q = T_Q (r_Q - [B_m w_{m-1}] - [U_{m+1} w_{m+2}]) and the h of the two neighbouring blocks,
// hT = B_m^T x_Q(standard part), hB = U_{m+1}^T x_Q(mirrored part). yT / yB: entries 1..4 of w_{m-1} / w_{m+2}
// (zeros where that half has no blocks).
template <class T>
EMG_HD void wide_middle(const T (&Tq)[21], const double (&lfB)[8], const double (&lfU)[8], const T (&rq)[6],
                        const T (&yT)[4], const T (&yB)[4], T (&xq)[6], T (&hT)[4], T (&hB)[4])
{
    T z[6];
    T q0 = lfB[0] * yT[0], q5 = lfU[0] * yB[0];
#pragma unroll
    for (int b = 1; b < 4; ++b) { q0 = mad(lfB[b], yT[b], q0); q5 = mad(lfU[b], yB[b], q5); }
    z[0] = rq[0] - q0;
    z[5] = rq[5] - q5;
#pragma unroll
    for (int b = 0; b < 4; ++b) z[b + 1] = nmad(lfU[4 + b], yB[b], nmad(lfB[4 + b], yT[b], rq[b + 1]));
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const T lo = mad(Tq[sym(r, 4)], z[4], mad(Tq[sym(r, 2)], z[2], mad(Tq[sym(r, 0)], z[0], zero<T>())));
        const T hi = mad(Tq[sym(r, 5)], z[5], mad(Tq[sym(r, 3)], z[3], mad(Tq[sym(r, 1)], z[1], zero<T>())));
        xq[r] = lo + hi;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        hT[b] = mad(lfB[b], xq[0], lfB[4 + b] * xq[b + 1]);
        hB[b] = mad(lfU[b], xq[5], lfU[4 + b] * xq[b + 1]);
    }
}
// one chain step on a 4-vector: v <- acc - M v with M = N (transposed = false) or N^T; every entry is summed
// as (p0 + p2) + (p1 + p3), p_j = (j == 0 ? acc : 0) - M(.,j) v_j -- what the 16 lanes of the kernel compute
template <class T> EMG_HD void wide_chain_step(const T (&N)[16], bool transposed, const T (&acc)[4], T (&v)[4])
{
    T o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = nmad(transposed ? N[4 * j + r] : N[4 * r + j], v[j], j == 0 ? acc[r] : zero<T>());
        o[r] = (p[0] + p[2]) + (p[1] + p[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = o[r];
}

// The blocks of a line in the wide form: item j = 0 .. n0-3 is top block j (j < m) or bottom block j + 2.
struct WideBlock { int k, mir; };
EMG_HD WideBlock wide_block(int j, int mk) { return j < mk ? WideBlock{j, 0} : WideBlock{j + 2, 1}; }

// Right-hand side of a block in its own grouping: standard block k = {E0(k), t(k+1)} (record row k), mirrored
// block k = {E0(k), t(k)} (entry 0 of row k, entries 1..4 of row k - 1).
template <class T, int DIR>
EMG_HD void wide_block_rhs(const Axes<T, DIR> &A, int k, int mir, int i1, int i2, T (&r)[5])
{
    line_rhs_t<T, DIR>(A, k - mir, i1, i2, r);
    r[0] = line_rhs_e0<T, DIR>(A, k, i1, i2);
}
// scatter of a block's solution (line_scatter, core.py:775-783, in the block's grouping)
template <class T, int DIR>
EMG_HD void wide_block_scatter(const Axes<T, DIR> &A, int k, int mir, int i1, int i2, const T (&x)[5])
{
    const int node = k - mir + 1;
    A.E(0)[A.idx(0, k, i1, i2)] = x[0];
    A.E(1)[A.idx(1, node, i1 - 1, i2)] = x[1];
    A.E(1)[A.idx(1, node, i1, i2)] = x[2];
    A.E(2)[A.idx(2, node, i1, i2 - 1)] = x[3];
    A.E(2)[A.idx(2, node, i1, i2)] = x[4];
}

// CPU walk of one line in the wide form (unit tests; the specification of k_line_wide): right-hand sides,
// solve and scatter, in place on the field. nrec: the line's N records (record k at nrec[k * nlines * 16]).
template <class T, int DIR>
EMG_HD void line_wide_ref(const Axes<T, DIR> &A, int i1, int i2, int nlines, int lid, const T *fac,
                          const double *lfac, const T *nfac)
{
    const int n0 = A.n0(), mk = line_mid(n0);
    const int nbt = mk, nbb = n0 - mk - 2;
    auto rec = [&](int k) { return (size_t)k * nlines + lid; };
    auto getT = [&](int k, T (&Tk)[15]) { for (int j = 0; j < 15; ++j) Tk[j] = fac[rec(k) * 15 + j]; };
    auto getC = [&](int k, double (&lf)[8]) { for (int j = 0; j < 8; ++j) lf[j] = lfac[rec(k) * 8 + j]; };
    auto getN = [&](int k, T (&N)[16]) { for (int j = 0; j < 16; ++j) N[j] = nfac[rec(k) * 16 + j]; };
    // per-block state (indexed by block k)
    T r[WIDE_N0_MAX][5], c[WIDE_N0_MAX][5], gy[WIDE_N0_MAX][4], gh[WIDE_N0_MAX][4];
    for (int j = 0; j < n0 - 2; ++j) {                     // phase A
        const WideBlock b = wide_block(j, mk);
        T Tk[15];
        wide_block_rhs<T, DIR>(A, b.k, b.mir, i1, i2, r[b.k]);
        getT(b.k, Tk);
        wide_g<T, 15>(Tk, r[b.k], gy[b.k]);
    }
    T rq[6];
    {
        T rm[5];
        line_rhs<T, DIR>(A, mk, i1, i2, rm);
        for (int j = 0; j < 5; ++j) rq[j] = rm[j];
        rq[5] = line_rhs_e0<T, DIR>(A, mk + 1, i1, i2);
    }
    for (int half = 0; half < 2; ++half) {                 // forward chains
        const int nb = half ? nbb : nbt, k0 = half ? n0 - 1 : 0, dk = half ? -1 : 1;
        T v[4] = {zero<T>(), zero<T>(), zero<T>(), zero<T>()};
        for (int i = 0; i < nb; ++i) {
            const int k = k0 + i * dk;
            T N[16];
            getN(k, N);
            wide_chain_step<T>(N, false, gy[k], v);
            for (int e = 0; e < 4; ++e) gy[k][e] = v[e];
        }
    }
    for (int j = 0; j < n0 - 2; ++j) {                     // per block: c, w0, g'
        const WideBlock b = wide_block(j, mk);
        const int k = b.k, kn = b.mir ? k + 1 : k - 1;
        const bool first = b.mir ? k == n0 - 1 : k == 0;
        T Tk[15];
        double lf[8];
        getT(k, Tk); getC(k, lf);
        wide_c<T>(lf, r[k], gy[first ? k : kn], c[k]);
        const T w0 = wide_row5<T, 15>(Tk, 0, c[k]);
        if (!first) wide_gp<T>(lf, w0, gy[k], gh[kn]);
    }
    T xq[6];
    {                                                      // middle block
        T Tq[21], yT[4], yB[4], hT[4], hB[4];
        double lfB[8], lfU[8];
        for (int j = 0; j < 15; ++j) Tq[j] = fac[rec(mk) * 15 + j];
        for (int j = 0; j < 6; ++j) Tq[15 + j] = fac[rec(mk + 1) * 15 + j];
        getC(mk, lfB); getC(mk + 1, lfU);
        for (int e = 0; e < 4; ++e) {
            yT[e] = nbt > 0 ? gy[mk - 1][e] : zero<T>();
            yB[e] = nbb > 0 ? gy[mk + 2][e] : zero<T>();
        }
        wide_middle<T>(Tq, lfB, lfU, rq, yT, yB, xq, hT, hB);
        for (int e = 0; e < 4; ++e) {
            if (nbt > 0) gh[mk - 1][e] = hT[e];
            if (nbb > 0) gh[mk + 2][e] = hB[e];
        }
    }
    for (int half = 0; half < 2; ++half) {                 // backward chains
        const int nb = half ? nbb : nbt, kb0 = half ? mk + 2 : mk - 1, dk = half ? 1 : -1;
        if (nb < 1) continue;
        T v[4];
        for (int e = 0; e < 4; ++e) v[e] = gh[kb0][e];
        for (int i = 0; i + 1 < nb; ++i) {
            const int k = kb0 + (i + 1) * dk;
            T N[16];
            getN(kb0 + i * dk, N);
            wide_chain_step<T>(N, true, gh[k], v);
            for (int e = 0; e < 4; ++e) gh[k][e] = v[e];
        }
    }
    for (int j = 0; j < n0 - 2; ++j) {                     // per block: x, scatter
        const WideBlock b = wide_block(j, mk);
        T Tk[15], x[5];
        getT(b.k, Tk);
        wide_x<T, 15>(Tk, c[b.k], gh[b.k], x);
        wide_block_scatter<T, DIR>(A, b.k, b.mir, i1, i2, x);
    }
    {
        const T xs[5] = {xq[0], xq[1], xq[2], xq[3], xq[4]};
        wide_block_scatter<T, DIR>(A, mk, 0, i1, i2, xs);
        A.E(0)[A.idx(0, mk + 1, i1, i2)] = xq[5];
    }
}

// ---------------------------------------------------------------------------------------
// Restriction of the residual, core.restrict (reference emg3d/core.py:1620-2001).
// One body for the seven sc_dir variants: c{x,y,z} say whether a direction is coarsened.
// w?[0..2] point to (wl, w0, wr) of that direction (unused when not coarsened).
// ---------------------------------------------------------------------------------------
template <class T> struct Restrict {
    int cx, cy, cz;              // direction coarsened? (0/1)
    int nxn, nyn, nzn;           // FINE node counts (nx+1, ...)
    int cnxn, cnyn, cnzn;        // COARSE node counts
    const T *rx, *ry, *rz;       // fine residual
    T *crx, *cry, *crz;          // coarse source
    T *cex = nullptr, *cey = nullptr, *cez = nullptr;   // optional: coarse FIELD, set to zero (solver.py:941)
    const double *wx[3], *wy[3], *wz[3];
    int batch = 1;               // right-hand sides (Level::batch); strides in elements
    size_t fstride = 0, cstride = 0;
};

template <class T> EMG_HD void restrict_node(const Restrict<T> &R, int cix, int ciy, int ciz)
{
    const int nx = R.nxn, ny = R.nyn, nz = R.nzn;
    const int cnx = R.cnxn, cny = R.cnyn;
    const int ix = R.cx ? 2 * cix : cix, iy = R.cy ? 2 * ciy : ciy, iz = R.cz ? 2 * ciz : ciz;
    int ixs[3], iys[3], izs[3];
    double wxs[3], wys[3], wzs[3];
    ixs[0] = ix; ixs[1] = ix > 0 ? ix - 1 : 0; ixs[2] = ix + 1 < nx - 1 ? ix + 1 : nx - 1;
    iys[0] = iy; iys[1] = iy > 0 ? iy - 1 : 0; iys[2] = iy + 1 < ny - 1 ? iy + 1 : ny - 1;
    izs[0] = iz; izs[1] = iz > 0 ? iz - 1 : 0; izs[2] = iz + 1 < nz - 1 ? iz + 1 : nz - 1;
    wxs[0] = R.cx ? R.wx[1][cix] : 1.0; wxs[1] = R.cx ? R.wx[0][cix] : 0.0; wxs[2] = R.cx ? R.wx[2][cix] : 0.0;
    wys[0] = R.cy ? R.wy[1][ciy] : 1.0; wys[1] = R.cy ? R.wy[0][ciy] : 0.0; wys[2] = R.cy ? R.wy[2][ciy] : 0.0;
    wzs[0] = R.cz ? R.wz[1][ciz] : 1.0; wzs[1] = R.cz ? R.wz[0][ciz] : 0.0; wzs[2] = R.cz ? R.wz[2][ciz] : 0.0;
    const int nxt = R.cx ? 3 : 1, nyt = R.cy ? 3 : 1, nzt = R.cz ? 3 : 1;

    if (cix < cnx - 1) {   // x-edges: fine shape (nx-1, ny, nz)
        T acc = zero<T>();
        for (int a = 0; a < nyt; ++a) {
            T inner = zero<T>();
            for (int b = 0; b < nzt; ++b) {
                T v = R.rx[ix + (nx - 1) * (iys[a] + ny * izs[b])];
                if (R.cx) v += R.rx[ixs[2] + (nx - 1) * (iys[a] + ny * izs[b])];
                inner += wzs[b] * v;
            }
            acc += wys[a] * inner;
        }
        R.crx[cix + (cnx - 1) * (ciy + cny * ciz)] = acc;
        if (R.cex) R.cex[cix + (cnx - 1) * (ciy + cny * ciz)] = zero<T>();
    }
    if (ciy < cny - 1) {   // y-edges: fine shape (nx, ny-1, nz)
        T acc = zero<T>();
        for (int a = 0; a < nxt; ++a) {
            T inner = zero<T>();
            for (int b = 0; b < nzt; ++b) {
                T v = R.ry[ixs[a] + nx * (iy + (ny - 1) * izs[b])];
                if (R.cy) v += R.ry[ixs[a] + nx * (iys[2] + (ny - 1) * izs[b])];
                inner += wzs[b] * v;
            }
            acc += wxs[a] * inner;
        }
        R.cry[cix + cnx * (ciy + (cny - 1) * ciz)] = acc;
        if (R.cey) R.cey[cix + cnx * (ciy + (cny - 1) * ciz)] = zero<T>();
    }
    if (ciz < R.cnzn - 1) {   // z-edges: fine shape (nx, ny, nz-1)
        T acc = zero<T>();
        for (int a = 0; a < nxt; ++a) {
            T inner = zero<T>();
            for (int b = 0; b < nyt; ++b) {
                T v = R.rz[ixs[a] + nx * (iys[b] + ny * iz)];
                if (R.cz) v += R.rz[ixs[a] + nx * (iys[b] + ny * izs[2])];
                inner += wys[b] * v;
            }
            acc += wxs[a] * inner;
        }
        R.crz[cix + cnx * (ciy + cny * ciz)] = acc;
        if (R.cez) R.cez[cix + cnx * (ciy + cny * ciz)] = zero<T>();
    }
}

// ---------------------------------------------------------------------------------------
// Prolongation  efield += P cefield  (reference emg3d/solver.py:947-1019, 1385-1478):
// bilinear in the two directions transverse to the edge, piecewise constant along it,
// interior nodes only. Per direction d the host supplies for every FINE node the lower
// coarse node index il[d][i] and the weight w[d][i] of the UPPER coarse node
// (solver.py:1457-1462); for a non-coarsened direction il = i (clamped), w = 0/1 exactly
// as np.searchsorted gives it.
// ---------------------------------------------------------------------------------------
template <class T> struct Prolong {
    int cx, cy, cz;              // direction coarsened?
    int nx, ny, nz;              // fine cells
    int cnx, cny, cnz;           // coarse cells
    T *ex, *ey, *ez;             // fine field (updated)
    const T *cex, *cey, *cez;    // coarse field
    const int *ilx, *ily, *ilz;  // lower coarse node per fine node
    const double *wx, *wy, *wz;  // weight of upper coarse node per fine node
    int batch = 1;               // right-hand sides (Level::batch); strides in elements
    size_t fstride = 0, cstride = 0;
};

// Fine "extended cell" (ix,iy,iz): updates ex[ix,iy,iz], ey[ix,iy,iz], ez[ix,iy,iz] when
// they exist and are interior in their transverse directions.
template <class T> EMG_HD void prolong_cell(const Prolong<T> &P, int ix, int iy, int iz)
{
    const int nx = P.nx, ny = P.ny, nz = P.nz, cnx = P.cnx, cny = P.cny;
    const bool ty = iy >= 1 && iy <= ny - 1, tz = iz >= 1 && iz <= nz - 1, tx = ix >= 1 && ix <= nx - 1;
    // the four weights are formed in the order of itertools.product((0,1),(0,1)):
    // (lo,lo),(lo,hi),(hi,lo),(hi,hi) with first index = first transverse direction
    if (ix < nx && ty && tz) {
        const int cix = P.cx ? ix / 2 : ix;
        const int a = P.ily[iy], b = P.ilz[iz];
        const double wa = P.wy[iy], wb = P.wz[iz];
        const T *c = P.cex + cix;
        const T v = c[cnx * (a + (cny + 1) * b)] * ((1 - wa) * (1 - wb)) +
                    c[cnx * (a + (cny + 1) * (b + 1))] * ((1 - wa) * wb) +
                    c[cnx * (a + 1 + (cny + 1) * b)] * (wa * (1 - wb)) +
                    c[cnx * (a + 1 + (cny + 1) * (b + 1))] * (wa * wb);
        P.ex[ix + nx * (iy + (ny + 1) * iz)] += v;
    }
    if (iy < ny && tx && tz) {
        const int ciy = P.cy ? iy / 2 : iy;
        const int a = P.ilx[ix], b = P.ilz[iz];
        const double wa = P.wx[ix], wb = P.wz[iz];
        const T *c = P.cey + (cnx + 1) * ciy;
        const T v = c[a + (cnx + 1) * cny * b] * ((1 - wa) * (1 - wb)) +
                    c[a + (cnx + 1) * cny * (b + 1)] * ((1 - wa) * wb) +
                    c[a + 1 + (cnx + 1) * cny * b] * (wa * (1 - wb)) +
                    c[a + 1 + (cnx + 1) * cny * (b + 1)] * (wa * wb);
        P.ey[ix + (nx + 1) * (iy + ny * iz)] += v;
    }
    if (iz < nz && tx && ty) {
        const int ciz = P.cz ? iz / 2 : iz;
        const int a = P.ilx[ix], b = P.ily[iy];
        const double wa = P.wx[ix], wb = P.wy[iy];
        const T *c = P.cez + (cnx + 1) * (cny + 1) * ciz;
        const T v = c[a + (cnx + 1) * b] * ((1 - wa) * (1 - wb)) +
                    c[a + (cnx + 1) * (b + 1)] * ((1 - wa) * wb) +
                    c[a + 1 + (cnx + 1) * b] * (wa * (1 - wb)) +
                    c[a + 1 + (cnx + 1) * (b + 1)] * (wa * wb);
        P.ez[ix + (nx + 1) * (iy + (ny + 1) * iz)] += v;
    }
}

// ---------------------------------------------------------------------------------------
// Coarse model parameter = SUM of the 2/4/8 fine cells (reference
// emg3d/solver.py:1667-1718). fx,fy,fz in {1,2}: coarsening factor per direction.
// ---------------------------------------------------------------------------------------
template <class T>
EMG_HD void restrict_param_cell(T *out, const T *in, int nx, int ny, int cnx, int cny, int fx, int fy,
                                int fz, int cix, int ciy, int ciz)
{
    T acc = zero<T>();
    for (int c = 0; c < fz; ++c)
        for (int b = 0; b < fy; ++b)
            for (int a = 0; a < fx; ++a)
                acc += in[(fx * cix + a) + nx * ((fy * ciy + b) + ny * (fz * ciz + c))];
    out[cix + cnx * (ciy + cny * ciz)] = acc;
}

}  // namespace emg

// ---------------------------------------------------------------------------------------
// The reference's banded storage, kept only for its known-answer tests
// (emg3d_core_solve / emg3d_core_blocks_to_amat): A(i,j) -> amat[i+5j].
// ---------------------------------------------------------------------------------------
namespace emg {

// core.solve (reference emg3d/core.py:1481-1616) for arbitrary n, sequential.
template <class T> EMG_HD void band_solve(T *amat, T *bvec, int n)
{
    // factorisation, row by row (same L and D as the reference's column sweep)
    for (int i = 0; i < n; ++i) {
        const int k0 = i - 5 > 0 ? i - 5 : 0;
        for (int j = k0; j < i; ++j) {
            T t = amat[i + 5 * j];
            for (int k = k0; k < j; ++k) t -= amat[i + 5 * k] * amat[j + 5 * k] * amat[6 * k];
            amat[i + 5 * j] = t * recip(amat[6 * j]);
        }
        T d = amat[6 * i];
        for (int k = k0; k < i; ++k) d -= amat[i + 5 * k] * amat[i + 5 * k] * amat[6 * k];
        amat[6 * i] = d;
    }
    for (int j = 0; j < n; ++j) amat[6 * j] = recip(amat[6 * j]);   // core.py:1589-1592
    for (int j = 1; j < n; ++j) {                                   // core.py:1597-1603
        T h = zero<T>();
        for (int k = (j - 5 > 0 ? j - 5 : 0); k < j; ++k) h += amat[j + 5 * k] * bvec[k];
        bvec[j] -= h;
    }
    for (int j = 0; j < n; ++j) bvec[j] *= amat[6 * j];             // core.py:1606-1607
    for (int j = n - 2; j >= 0; --j) {                              // core.py:1610-1616
        T h = zero<T>();
        for (int k = j + 1; k < (n < j + 6 ? n : j + 6); ++k) h += amat[k + 5 * j] * bvec[k];
        bvec[j] -= h;
    }
}

// core.blocks_to_amat (reference emg3d/core.py:1351-1477)
template <class T>
EMG_HD void blocks_to_amat(T *amat, T *bvec, const T *middle, const double *left, const T *rhs,
                           int im, int nc)
{
    const int fam = 5 * im, mam = fam - 5;
    if (im == 0) {
        for (int k = 0; k < 5; ++k) bvec[k] = rhs[k];
        for (int k = 0; k < 5; ++k)
            for (int m = 0; m <= k; ++m) amat[k + 5 * m] = middle[k + 5 * m];
    } else if (im <= nc - 2 && nc > 2) {
        for (int k = 0; k < 5; ++k) bvec[k + fam] = rhs[k];
        for (int m = 1; m < 5; ++m)
            for (int k = 0; k <= m; ++k) amat[k + fam + 5 * (m + mam)] = T(left[k + 5 * m]);
        for (int k = 0; k < 5; ++k)
            for (int m = 0; m <= k; ++m) amat[k + fam + 5 * (m + fam)] = middle[k + 5 * m];
    } else if (im == nc - 1) {
        bvec[fam] = rhs[0];
        for (int m = 1; m < 5; ++m) amat[fam + 5 * (m + mam)] = T(left[5 * m]);
        amat[6 * fam] = middle[0];
    }
}

}  // namespace emg

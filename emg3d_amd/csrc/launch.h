// Work decomposition shared by the HIP kernels (kernels.hip) and by the CPU emulation
// harness of the unit tests (tests/emu/emu.cpp): which node / line / cell a global thread
// index owns, and how large the grids are. Keeping it in one place means the emulation
// exercises exactly the index arithmetic the GPU runs.
#pragma once
#include "stencil.h"

namespace emg {

struct Dim3 { int x, y, z; };

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// number of integers p in [1, n-1] with p % 2 == par
inline int cnt_par(int n, int par) { return par ? n / 2 : (n - 1) / 2; }
// first integer >= 1 with parity par
EMG_HD int first_par(int par) { return par ? 1 : 2; }

// Colour class visited at position cc (0..3) of a sweep. A forward sweep visits the classes
// in the sequence 0,2,3,1, a backward sweep in the reverse; the first sweep of a smoother
// call is backward, like the reference's (emg3d/core.py:301,311). Of the distinct
// sequences this one gives the best multigrid convergence factor (measured with the
// oracle: 0.128 vs 0.156 for 0,1,2,3 on the point smoother; lexicographic 0.088).
inline int sweep_colour(int iback, int cc)
{
    const int seq[4] = {0, 2, 3, 1};
    return seq[iback ? 3 - cc : cc];
}

// ---- point smoother: colour = ((ix+iz)&1) | (((iy+iz)&1)<<1); global thread (gx,gy,gz);
//      one launch covers the node planes iz0 .. iz0+izn-1.
inline Dim3 gs_point_block() { return Dim3{64, 4, 1}; }
inline Dim3 gs_point_grid(int nx, int ny, int izn)
{
    return Dim3{cdiv(cdiv(nx - 1, 2), 64), cdiv(cdiv(ny - 1, 2), 4), izn};
}
template <class T>
EMG_HD void gs_point_thread(const Level<T> &L, int colour, int iz0, int gx, int gy, int gz)
{
    const int iz = iz0 + gz;
    const int parx = (colour & 1) ^ (iz & 1);
    const int pary = ((colour >> 1) & 1) ^ (iz & 1);
    const int ix = first_par(parx) + 2 * gx;
    const int iy = first_par(pary) + 2 * gy;
    if (ix > L.nx - 1 || iy > L.ny - 1 || iz > L.nz - 1) return;
    gs_point_node<T>(L, ix, iy, iz);
}

// Launch schedule of ONE four-colour sweep of the point smoother.
//
// slab <= 0: four launches, one per colour class, each over all planes (the plain
// schedule). slab = T > 0: the planes are processed in rounds of T; in round r the colour
// at sweep position cc = 0..3 runs on planes [1 + rT - cc, 1 + (r+1)T - cc). The skew by
// one plane per position keeps every dependency of the plain schedule (a node conflicts
// only with nodes in planes iz-1, iz, iz+1): position cc at plane k sees positions < cc
// already updated and positions > cc not yet updated on k-1..k+1 -- the result is
// bit-identical to the plain schedule, but the ~4 consecutive launches of a round touch
// the same T+3 planes, which then come from the 256 MiB Infinity Cache instead of HBM.
// launch(colour, iz0, izn) is called for every kernel launch, in order.
template <class F> inline void gs_point_schedule(int nz, int slab, int iback, F launch)
{
    const int nplanes = nz - 1;
    if (nplanes <= 0) return;
    if (slab <= 0 || slab >= nplanes) {
        for (int cc = 0; cc < 4; ++cc) launch(sweep_colour(iback, cc), 1, nplanes);
        return;
    }
    const int rounds = cdiv(nplanes + 3, slab);
    for (int r = 0; r < rounds; ++r)
        for (int cc = 0; cc < 4; ++cc) {
            int a = 1 + r * slab - cc, b = a + slab;
            if (a < 1) a = 1;
            if (b > nz) b = nz;
            if (b > a) launch(sweep_colour(iback, cc), a, b - a);
        }
}

// ---- line smoothers: (p,q) = transverse PHYSICAL node indices in memory order (p faster):
//      DIR 0 (x-lines): (iy,iz)   DIR 1 (y-lines): (ix,iz)   DIR 2 (z-lines): (ix,iy)
//      colour = (p&1) | ((q&1)<<1). Lines of one colour class are numbered
//      lid = tp + cntp*tq with p = first_par(colour&1) + 2 tp, q likewise.
inline int line_np(int dir, int nx, int ny, int nz) { (void)nz; return dir == 0 ? ny : nx; }
inline int line_nq(int dir, int nx, int ny, int nz) { (void)nx; return dir == 2 ? ny : nz; }
inline int line_n0(int dir, int nx, int ny, int nz) { return dir == 0 ? nx : dir == 1 ? ny : nz; }

// The factor / rhs records of a line are padded to a multiple of LINE_PAD blocks with
// "identity" blocks (C = 0, 1/D = 1, B = 0, rhs = 0), so that the forward / backward
// kernels can run a LINE_PAD-times unrolled, branch-free software pipeline.
constexpr int LINE_PAD = 4;
// elements at the tail of the rhs/solution scratch that absorb the stores of the surplus
// quads of the last wave of the forward / backward kernels (16 quads x 5 entries)
constexpr int LINE_DUMMY = 80;
EMG_HD int line_padded(int n0) { return (n0 + LINE_PAD - 1) / LINE_PAD * LINE_PAD; }

// Geometry of one colour class of one direction on one level.
struct LineClass {
    int n0, n0p, cntp, cntq, lines;   // blocks per line (real, padded), lines along p / q, total
    size_t fac_off, lfac_off;         // element offsets of the class in the factor buffers
};
inline LineClass line_class(int dir, int nx, int ny, int nz, int colour)
{
    LineClass c;
    c.n0 = line_n0(dir, nx, ny, nz);
    size_t before = 0;
    for (int cc = 0; cc <= colour; ++cc) {
        const int cp = cnt_par(line_np(dir, nx, ny, nz), cc & 1);
        const int cq = cnt_par(line_nq(dir, nx, ny, nz), (cc >> 1) & 1);
        if (cc == colour) { c.cntp = cp; c.cntq = cq; c.lines = cp * cq; }
        else before += (size_t)cp * cq;
    }
    c.n0p = line_padded(c.n0);
    c.fac_off = (size_t)15 * c.n0p * before;
    c.lfac_off = (size_t)8 * c.n0p * before;
    return c;
}
// all lines of a direction: (np-1)(nq-1)
inline size_t line_total(int dir, int nx, int ny, int nz)
{
    return (size_t)(line_np(dir, nx, ny, nz) - 1) * (line_nq(dir, nx, ny, nz) - 1);
}
// elements of the factor buffers and of the rhs/solution scratch of one direction
inline size_t line_fac_elems(int dir, int nx, int ny, int nz)
{
    return (size_t)15 * line_padded(line_n0(dir, nx, ny, nz)) * line_total(dir, nx, ny, nz);
}
inline size_t line_lfac_elems(int dir, int nx, int ny, int nz)
{
    return (size_t)8 * line_padded(line_n0(dir, nx, ny, nz)) * line_total(dir, nx, ny, nz);
}
inline size_t line_vec_elems(int dir, int nx, int ny, int nz)
{   // largest colour class (odd,odd)
    const size_t lines = (size_t)cnt_par(line_np(dir, nx, ny, nz), 1) * cnt_par(line_nq(dir, nx, ny, nz), 1);
    return (size_t)5 * line_padded(line_n0(dir, nx, ny, nz)) * lines + LINE_DUMMY;
}

// per-line kernels (setup, forward, backward): one thread per line
inline Dim3 line_block() { return Dim3{64, 1, 1}; }
inline Dim3 line_grid(const LineClass &c) { return Dim3{cdiv(c.cntp, 64), c.cntq, 1}; }
// per-(line, block) kernels (rhs, scatter): thread (tp, tq, k)
inline Dim3 lineblk_block() { return Dim3{64, 1, 1}; }
inline Dim3 lineblk_grid(const LineClass &c, bool padded) { return Dim3{cdiv(c.cntp, 64), c.cntq, padded ? c.n0p : c.n0}; }

// (i1, i2, lid) of thread (tp, tq) in colour class `colour`; false if out of range
template <int DIR>
EMG_HD bool line_of_thread(int colour, int cntp, int cntq, int tp, int tq, int &i1, int &i2, int &lid)
{
    if (tp >= cntp || tq >= cntq) return false;
    const int p = first_par(colour & 1) + 2 * tp;
    const int q = first_par((colour >> 1) & 1) + 2 * tq;
    lid = tp + cntp * tq;
    // abstract (i1,i2): DIR 0: (iy,iz)=(p,q); DIR 1: (iz,ix)=(q,p); DIR 2: (ix,iy)=(p,q)
    i1 = DIR == 1 ? q : p;
    i2 = DIR == 1 ? p : q;
    return true;
}

template <class T, int DIR>
EMG_HD void line_setup_thread(const Level<T> &L, int colour, int cntp, int cntq, int tp, int tq, T *fac,
                              double *lfac)
{
    int i1, i2, lid;
    if (!line_of_thread<DIR>(colour, cntp, cntq, tp, tq, i1, i2, lid)) return;
    line_setup<T, DIR>(L, i1, i2, fac, lfac, cntp * cntq, lid, line_padded(Axes<T, DIR>(L).n0()));
}

template <class T, int DIR>
EMG_HD void line_rhs_thread(const Level<T> &L, int colour, int cntp, int cntq, int tp, int tq, int k, T *vec)
{
    int i1, i2, lid;
    if (!line_of_thread<DIR>(colour, cntp, cntq, tp, tq, i1, i2, lid)) return;
    const Axes<T, DIR> A(L);
    T rhs[5];
    if (k < A.n0()) {
        line_rhs<T, DIR>(A, k, i1, i2, rhs);
    } else {   // padding block
#pragma unroll
        for (int r = 0; r < 5; ++r) rhs[r] = zero<T>();
    }
    T *o = vec + ((size_t)k * (cntp * cntq) + lid) * 5;
#pragma unroll
    for (int r = 0; r < 5; ++r) o[r] = rhs[r];
}

template <class T, int DIR>
EMG_HD void line_scatter_thread(const Level<T> &L, int colour, int cntp, int cntq, int tp, int tq, int k,
                                const T *vec)
{
    int i1, i2, lid;
    if (!line_of_thread<DIR>(colour, cntp, cntq, tp, tq, i1, i2, lid)) return;
    const Axes<T, DIR> A(L);
    line_scatter<T, DIR>(A, k, i1, i2, vec + ((size_t)k * (cntp * cntq) + lid) * 5);
}

// forward/backward kernels: FOUR lanes per line, 16 lines per wave
inline Dim3 linequad_block() { return Dim3{64, 1, 1}; }
inline Dim3 linequad_grid(const LineClass &c) { return Dim3{cdiv(c.lines * 4, 64), 1, 1}; }

// ---- "extended cell" kernels (residual, prolongation, PEC): one thread per node-indexed
//      cell (ix,iy,iz), 0 <= ix <= nx etc.
inline Dim3 cell_block() { return Dim3{64, 4, 1}; }
inline Dim3 cell_grid(int n1, int n2, int n3) { return Dim3{cdiv(n1, 64), cdiv(n2, 4), n3}; }

struct ScDirs { int cx, cy, cz; };
// which directions are coarsened for a given sc_dir (reference emg3d/solver.py:891-897)
inline ScDirs sc_flags(int sc_dir)
{
    ScDirs f;
    f.cx = !(sc_dir == 1 || sc_dir == 5 || sc_dir == 6);
    f.cy = !(sc_dir == 2 || sc_dir == 4 || sc_dir == 6);
    f.cz = !(sc_dir == 3 || sc_dir == 4 || sc_dir == 5);
    return f;
}

template <class T>
inline Restrict<T> make_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry,
                                 const void *rz, const double *const w[9], int nx, int ny, int nz, int sc_dir)
{
    const ScDirs f = sc_flags(sc_dir);
    Restrict<T> R;
    R.cx = f.cx; R.cy = f.cy; R.cz = f.cz;
    R.nxn = nx + 1; R.nyn = ny + 1; R.nzn = nz + 1;
    R.cnxn = (f.cx ? nx / 2 : nx) + 1; R.cnyn = (f.cy ? ny / 2 : ny) + 1; R.cnzn = (f.cz ? nz / 2 : nz) + 1;
    R.rx = (const T *)rx; R.ry = (const T *)ry; R.rz = (const T *)rz;
    R.crx = (T *)crx; R.cry = (T *)cry; R.crz = (T *)crz;
    for (int i = 0; i < 3; ++i) { R.wx[i] = w[i]; R.wy[i] = w[3 + i]; R.wz[i] = w[6 + i]; }
    return R;
}

template <class T>
inline Prolong<T> make_prolong(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                               const int *ilx, const int *ily, const int *ilz, const double *wx,
                               const double *wy, const double *wz, int nx, int ny, int nz, int sc_dir)
{
    const ScDirs f = sc_flags(sc_dir);
    Prolong<T> P;
    P.cx = f.cx; P.cy = f.cy; P.cz = f.cz;
    P.nx = nx; P.ny = ny; P.nz = nz;
    P.cnx = f.cx ? nx / 2 : nx; P.cny = f.cy ? ny / 2 : ny; P.cnz = f.cz ? nz / 2 : nz;
    P.ex = (T *)ex; P.ey = (T *)ey; P.ez = (T *)ez;
    P.cex = (const T *)cex; P.cey = (const T *)cey; P.cez = (const T *)cez;
    P.ilx = ilx; P.ily = ily; P.ilz = ilz; P.wx = wx; P.wy = wy; P.wz = wz;
    return P;
}

}  // namespace emg

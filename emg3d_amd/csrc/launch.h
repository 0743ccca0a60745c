// Work decomposition shared by the HIP kernels (kernels.hip) and by the CPU emulation
// harness of the unit tests (tests/emu/emu.cpp): which node / line / cell a global thread
// index owns, and how large the grids are. Keeping it in one place means the emulation
// exercises exactly the index arithmetic the GPU runs.
#pragma once
#include "stencil.h"

namespace emg {

struct Dim3 { int x, y, z; };

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// a / b for 0 <= a < 2^19, b > 0 without the integer-division sequence (~40 instructions on the GPU, and the small line
// kernels are bound by what one wave can issue): the quotient of (a + 1/2) / b in single precision is never within
// rounding of an integer
EMG_HD int fast_div(int a, int b) { return (int)(((float)a + 0.5f) / (float)b); }
// number of integers p in [1, n-1] with p % 2 == par
inline int cnt_par(int n, int par) { return par ? n / 2 : (n - 1) / 2; }
// first integer >= 1 with parity par
EMG_HD int first_par(int par) { return par ? 1 : 2; }

// Colour class visited at position cc (0..3) of a sweep.
// mirrored_colour: a forward sweep visits the classes in the sequence 0,2,3,1, a backward sweep in the
// reverse (the first sweep of a smoother call is backward, like the reference's, emg3d/core.py:301,311)
// -- the rule of rounds 1-2 for all smoothers. Of the six distinct forward sequences 0,2,3,1 (and its
// x<->y mirror image 0,1,3,2) gave the best convergence factor WITH mirrored backward sweeps (oracle:
// 0.128 vs 0.156 for 0,1,2,3 on the point smoother; lexicographic 0.088).
inline int mirrored_colour(int iback, int cc)
{
    const int seq[4] = {0, 2, 3, 1};
    return seq[iback ? 3 - cc : cc];
}
// POINT smoother (round 3; option point_order, default 1): every sweep visits the NODE colours in the
// same sequence 0,2,3,1, whatever its direction (the sweep direction still mirrors the TILE order of
// the tiled schedule below). Measured with the oracle (cycles to 1e-8, mirrored ->
// repeated, reference order): uniform 32^3 9 -> 7 (8), tri-axial 64^3 27 -> 19 (20), the tiled schedule
// 26 -> 19 and 17 -> 12 on the marine model (11); nowhere slower. In the tiled kernel it costs nothing (a
// tile runs four colour steps per visit either way); on small levels a call of two sweeps is eight
// launches instead of seven. point_order = 0: the mirrored rule.
inline int &point_order_ref() { static int order = 1; return order; }
inline int sweep_colour(int iback, int cc)
{
    return mirrored_colour(point_order_ref() == 0 ? iback : 0, cc);
}
// LINE smoothers (round 3): the colour passes of a call CYCLE through the classes 1,2,3,0,1,2,3,0,...
// -- sweep `it` (0, 1, ...) of the call takes positions 3 it .. 3 it + 3, so that it begins with the
// class the previous sweep ended with (that pass reproduces the same values and is not launched:
// 4 nu - (nu - 1) passes per call, as with the mirrored order). Measured with the oracle on reduced
// copies of BASELINE.json's configurations: every mirrored pair of sequences (a
// backward sweep followed by its reverse, the sweep_colour() rule above) is among the slowest of
// all 576 pairs -- 24 / 11 / 10 cycles to 1e-10 on configs 3 / 2 / 5 --, every cyclic one among the
// fastest with seven passes per two sweeps -- 21 / 9 / 9 (the reference's sequential order: 17 / 8 / 8).
// order 0: the mirrored rule (the definition of rounds 1-2, kept for comparison). order 2: the SAME
// sequence 1,2,3,0 in every sweep -- no pass is shared between sweeps (4 nu launches per call), the
// oracle needs as few cycles with it as with the reference's sequential sweeps.
inline int line_sweep_colour(int order, int it, int cc)
{
    if (order == 0) return mirrored_colour((it + 1) & 1, cc);     // first sweep backward
    const int seq[4] = {1, 2, 3, 0};
    if (order == 2) return seq[cc];
    return seq[(3 * it + cc) & 3];
}
// the first pass of sweep `it` repeats the last pass of the sweep before it (and is not launched)
inline bool line_pass_repeats(int order, int it)
{
    return it > 0 && line_sweep_colour(order, it, 0) == line_sweep_colour(order, it - 1, 3);
}

// ---- point smoother: colour = ((ix+iz)&1) | (((iy+iz)&1)<<1); global thread (gx,gy,gz);
//      one launch covers the node planes iz0 .. iz0+izn-1.
inline Dim3 gs_point_block() { return Dim3{64, 4, 1}; }
inline Dim3 gs_point_grid(int nx, int ny, int izn)
{
    return Dim3{cdiv(cdiv(nx - 1, 2), 64), cdiv(cdiv(ny - 1, 2), 4), izn};
}
template <class T>
EMG_HD void gs_point_thread(const Level<T> &L, const T *pst, int colour, int iz0, int gx, int gy, int gz)
{
    const int iz = iz0 + gz;
    const int parx = (colour & 1) ^ (iz & 1);
    const int pary = ((colour >> 1) & 1) ^ (iz & 1);
    const int ix = first_par(parx) + 2 * gx;
    const int iy = first_par(pary) + 2 * gy;
    if (ix > L.nx - 1 || iy > L.ny - 1 || iz > L.nz - 1) return;
    gs_point_node<T>(L, pst, ix, iy, iz);
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

// ---- point smoother, TILED schedule (large levels).
//
// The plain schedule streams the whole level through HBM once per colour class (4 passes
// per sweep, each reading every field value although it updates a quarter of the nodes).
// The tiled schedule cuts the interior nodes into tiles of TB::BX x BY x BZ nodes; one
// workgroup stages a tile's edges (+ the one-edge halo) in LDS, runs the four colour
// classes on it back to back, and writes the tile's edges out -- one pass over the field
// per sweep. Tiles that run concurrently must not touch: the tiles are coloured
// (tx&1)|((ty&1)<<1)|((tz&1)<<2), visited in the order 0,7,1,6,2,5,3,4 (backward: reversed); a
// sweep is four launches of two complementary colours each (tile_colour_at). The result is a Gauss-Seidel sweep in the order "tile colour, then
// node colour inside every tile" -- a different, equally valid ordering than the plain
// schedule's (the oracle restates it: oracle/core_generic.h, order 2).
template <int BX_, int BY_, int BZ_> struct TileBox {
    static constexpr int BX = BX_, BY = BY_, BZ = BZ_;
    static constexpr int THREADS = BX * BY * BZ / 4;   // one thread per node of a colour class
    static_assert(BX % 2 == 0 && BY % 2 == 0, "tile extents in x and y must be even");
};
using PointTile = TileBox<32, 4, 6>;

struct TileCount { int x, y, z; };
template <class TB> inline TileCount tile_count(int nx, int ny, int nz)
{
    return TileCount{cdiv(nx - 1, TB::BX), cdiv(ny - 1, TB::BY), cdiv(nz - 1, TB::BZ)};
}
// launch grid of one tile colour: tiles t = par + 2 b along every axis
template <class TB> inline Dim3 tile_grid(int nx, int ny, int nz, int tc)
{
    const TileCount n = tile_count<TB>(nx, ny, nz);
    return Dim3{(n.x - (tc & 1) + 1) / 2, (n.y - ((tc >> 1) & 1) + 1) / 2, (n.z - ((tc >> 2) & 1) + 1) / 2};
}
// the rule that selects the tiled schedule (and with it the sweep order) for a level
inline bool point_tiled(int nx, int ny, int nz, int tile_min)
{
    // (the row-wise tile copies number the elements of a field component with 32 bits)
    return tile_min > 0 && (long long)(nx - 1) * (ny - 1) * (nz - 1) >= tile_min &&
           (long long)(nx + 1) * (ny + 1) * (nz + 1) < (1LL << 31);
}
// Tile colours in visiting order: 0,7,1,6,2,5,3,4 (backward: reversed). Consecutive pairs are
// complementary colours (c, 7-c): their tiles differ in the parity of ALL three tile indices,
// so they are at least corner-diagonal -- they share no edge either of them writes or reads --
// and one launch runs both (four launches per sweep, each with twice the tiles: a smaller
// share of the launch is its last, partly filled round of workgroups).
inline int tile_colour_at(int iback, int t8)
{
    const int seq[8] = {0, 7, 1, 6, 2, 5, 3, 4};
    return seq[iback ? 7 - t8 : t8];
}
// the two colours of launch p (0..3) of a sweep and their grids
struct TilePair { int tc[2], gx[2], gy[2], gz[2]; };
template <class TB> inline TilePair tile_pair(int nx, int ny, int nz, int iback, int p)
{
    TilePair P;
    for (int h = 0; h < 2; ++h) {
        P.tc[h] = tile_colour_at(iback, 2 * p + h);
        const Dim3 g = tile_grid<TB>(nx, ny, nz, P.tc[h]);
        const bool any = g.x > 0 && g.y > 0 && g.z > 0;
        P.gx[h] = any ? g.x : 0; P.gy[h] = any ? g.y : 0; P.gz[h] = any ? g.z : 0;
    }
    return P;
}
// the four node colours of a sweep packed two bits each, first-visited in the low bits
inline int sweep_colours_packed(int iback)
{
    int p = 0;
    for (int cc = 0; cc < 4; ++cc) p |= sweep_colour(iback, cc) << (2 * cc);
    return p;
}

// Phase 1 (thread t of THREADS): copy the tile's edges and halo into LDS. All loads are
// issued before the first LDS store (fully unrolled, branch-free: clamped source address;
// slots past the end of a box go to the spare LDS slot), so that a thread has its ~20
// 16-byte loads in flight together. Box element e of each array is LDS element e.
template <class T, class TB>
EMG_HD void tile_load_generic(const Level<T> &L, T *lds, int x0, int y0, int z0, int t)
{
    using E = EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    const EdgesGlobal<T> g(L);
    const int ox = x0 - 1, oy = y0 - 1, oz = z0 - 1;
    constexpr int TX = (E::NXE + TB::THREADS - 1) / TB::THREADS, TY = (E::NYE + TB::THREADS - 1) / TB::THREADS;
    constexpr int TZ = (E::NZE + TB::THREADS - 1) / TB::THREADS;
    T vx[TX], vy[TY], vz[TZ];
#pragma unroll
    for (int it = 0; it < TX; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % (TB::BX + 1), r = e / (TB::BX + 1), lj = r % (TB::BY + 2), lk = r / (TB::BY + 2);
        const int i = ox + li, j = oy + lj, k = oz + lk;
        const bool ok = e < E::NXE && i < L.nx && j <= L.ny && k <= L.nz;
        vx[it] = g.x(ok ? i : 0, ok ? j : 0, ok ? k : 0);
    }
#pragma unroll
    for (int it = 0; it < TY; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % (TB::BX + 2), r = e / (TB::BX + 2), lj = r % (TB::BY + 1), lk = r / (TB::BY + 1);
        const int i = ox + li, j = oy + lj, k = oz + lk;
        const bool ok = e < E::NYE && i <= L.nx && j < L.ny && k <= L.nz;
        vy[it] = g.y(ok ? i : 0, ok ? j : 0, ok ? k : 0);
    }
#pragma unroll
    for (int it = 0; it < TZ; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % (TB::BX + 2), r = e / (TB::BX + 2), lj = r % (TB::BY + 2), lk = r / (TB::BY + 2);
        const int i = ox + li, j = oy + lj, k = oz + lk;
        const bool ok = e < E::NZE && i <= L.nx && j <= L.ny && k < L.nz;
        vz[it] = g.z(ok ? i : 0, ok ? j : 0, ok ? k : 0);
    }
#pragma unroll
    for (int it = 0; it < TX; ++it) {
        const int e = t + it * TB::THREADS;
        lds[e < E::NXE ? e : E::ELEMS] = vx[it];
    }
#pragma unroll
    for (int it = 0; it < TY; ++it) {
        const int e = t + it * TB::THREADS;
        lds[e < E::NYE ? E::NXE + e : E::ELEMS] = vy[it];
    }
#pragma unroll
    for (int it = 0; it < TZ; ++it) {
        const int e = t + it * TB::THREADS;
        lds[e < E::NZE ? E::NXE + E::NYE + e : E::ELEMS] = vz[it];
    }
    // zeta of the cells around the tile's nodes (each is used by up to eight node updates)
    constexpr int TC = (E::NZC + TB::THREADS - 1) / TB::THREADS;
    double vc[TC];
#pragma unroll
    for (int it = 0; it < TC; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % (TB::BX + 1), r = e / (TB::BX + 1), lj = r % (TB::BY + 1), lk = r / (TB::BY + 1);
        const int i = ox + li, j = oy + lj, k = oz + lk;
        const bool ok = e < E::NZC && i < L.nx && j < L.ny && k < L.nz;
        vc[it] = L.zeta[ok ? i + L.nx * (j + L.ny * k) : 0];
    }
    double *const zb = reinterpret_cast<double *>(lds + E::LDS_ELEMS);
#pragma unroll
    for (int it = 0; it < TC; ++it) {
        const int e = t + it * TB::THREADS;
        zb[e < E::NZC ? e : E::NZC] = vc[it];
    }
}
// Phase 2 (once per node colour): the node of colour class `colour` that thread t owns;
// false if it lies outside the level (partial tile) -- the coordinates are then clamped to
// a valid node so that its inputs can still be fetched unconditionally.
template <class TB>
EMG_HD bool tile_node(int nx, int ny, int nz, int x0, int y0, int z0, int colour, int t, int &ix, int &iy, int &iz)
{
    const int jx = t % (TB::BX / 2), r = t / (TB::BX / 2), jy = r % (TB::BY / 2), lz = r / (TB::BY / 2);
    iz = z0 + lz;
    ix = x0 + 2 * jx + (((colour & 1) ^ (x0 + iz)) & 1);
    iy = y0 + 2 * jy + ((((colour >> 1) & 1) ^ (y0 + iz)) & 1);
    const bool ok = ix <= nx - 1 && iy <= ny - 1 && iz <= nz - 1;
    if (!ok) { ix = 1; iy = 1; iz = 1; }
    return ok;
}
template <class T, class TB>
EMG_HD void tile_colour(const Level<T> &L, const T *pst, T *lds, int x0, int y0, int z0, int colour, int t)
{
    int ix, iy, iz;
    if (!tile_node<TB>(L.nx, L.ny, L.nz, x0, y0, z0, colour, t, ix, iy, iz)) return;
    using E = EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    const E ed(lds, x0, y0, z0);
    const ZetaTile<E> zt{ed};
    PointIn<T> in;
    if (pst) point_load<T, true>(L, pst, zt, ix, iy, iz, in);
    else point_load<T, false>(L, pst, zt, ix, iy, iz, in);
    point_update<T, E>(L, in, ed, ix, iy, iz);
}
// ---- eta edge sums in TILE-MAJOR order (tiled point smoother).
// Stored in the edge-shaped arrays of point_setup_cell, the six sums of a node are strided
// gathers: the nodes of one colour class sit two apart in x, so every colour step of a tile
// uses half of each cache line it touches, and the four steps of a tile re-fetch the same
// lines (the tiles in flight on an XCD exceed its L2). Here the sums are stored per (tile,
// node colour, entry r = 0..5, thread t): a wave reads 64 consecutive values per entry, every
// byte of the buffer is read exactly once per sweep. Each edge sum exists twice (once for
// either end node) -- 96 B per node instead of 48 -- unless the level's eta are purely
// imaginary (LEVEL_ETA_IMAG; real fields: always): then only that half is stored, 48 B per node.
template <class TB> EMG_HD size_t tile_pst_index(int ntx, int nty, int tx, int ty, int tz, int colour, int r, int t)
{
    const size_t tile = (size_t)tx + (size_t)ntx * ((size_t)ty + (size_t)nty * tz);
    return ((tile * 4 + colour) * 6 + r) * TB::THREADS + t;
}
inline size_t tile_pst_elems(int nx, int ny, int nz, int bx, int by, int bz)
{
    return (size_t)cdiv(nx - 1, bx) * cdiv(ny - 1, by) * cdiv(nz - 1, bz) * 6 * (size_t)(bx * by * bz);
}
// MODE (the ST of k_gs_point_tile): 2 full values (T); 3 stored halves (double); and, on levels that solve a correction
// equation (LEVEL_POINT_COMPACT: the sums are coefficients of the smoother's local systems -- rounded to single precision
// they perturb the smoother by 6e-8 relative, not the equation), 4 stored halves as float, 5 full values as compact_of<T>
constexpr int PST_FULL = 2, PST_HALF = 3, PST_HALF_F32 = 4, PST_FULL_F32 = 5;
EMG_HD constexpr size_t tile_pst_bytes(int mode)
{
    return mode == PST_FULL ? 16 : mode == PST_HALF_F32 ? 4 : 8;      // (per COMPLEX value; real fields: halves only)
}
// setup: thread t of the workgroup of tile (tx,ty,tz) writes the sums of its four nodes
template <class T, class TB, int MODE>
EMG_HD void tile_pst_setup(const Level<T> &L, void *pst, int ntx, int nty, int tx, int ty, int tz, int t)
{
    const int x0 = 1 + tx * TB::BX, y0 = 1 + ty * TB::BY, z0 = 1 + tz * TB::BZ;
    for (int c = 0; c < 4; ++c) {
        int ix, iy, iz;
        const bool ok = tile_node<TB>(L.nx, L.ny, L.nz, x0, y0, z0, c, t, ix, iy, iz);
        PointIn<T> in;
        point_load_eta<T, false>(L, nullptr, ix, iy, iz, in);
        for (int r = 0; r < 6; ++r) {
            const size_t o = tile_pst_index<TB>(ntx, nty, tx, ty, tz, c, r, t);
            const T v = ok ? in.st[r] : zero<T>();
            using CT = typename compact_of<T>::type;
            if (MODE == PST_HALF) reinterpret_cast<double *>(pst)[o] = imag_of(v);
            else if (MODE == PST_HALF_F32) reinterpret_cast<float *>(pst)[o] = (float)imag_of(v);
            else if (MODE == PST_FULL_F32) reinterpret_cast<CT *>(pst)[o] = narrow<CT>(v);
            else reinterpret_cast<T *>(pst)[o] = v;
        }
    }
}
// smoother: the six sums of thread t's node of colour class `colour` in tile (tx,ty,tz)
template <class T, class TB, int MODE>
EMG_HD void tile_pst_load(const void *pst, int ntx, int nty, int tx, int ty, int tz, int colour, int t, PointIn<T> &in)
{
    const size_t o = tile_pst_index<TB>(ntx, nty, tx, ty, tz, colour, 0, t);
#if !defined(__HIPCC__)
    for (int r = 0; r < 6; ++r)
#else
#pragma unroll
    for (int r = 0; r < 6; ++r)
#endif
    {
        using CT = typename compact_of<T>::type;
        if (MODE == PST_HALF) in.st[r] = from_stored<T>(reinterpret_cast<const double *>(pst)[o + (size_t)r * TB::THREADS]);
        else if (MODE == PST_HALF_F32) in.st[r] = from_stored<T>((double)reinterpret_cast<const float *>(pst)[o + (size_t)r * TB::THREADS]);
        else if (MODE == PST_FULL_F32) in.st[r] = widen(reinterpret_cast<const CT *>(pst)[o + (size_t)r * TB::THREADS]);
        else in.st[r] = reinterpret_cast<const T *>(pst)[o + (size_t)r * TB::THREADS];
    }
}

// Phase 3: write the edges attached to the tile's nodes back (the halo is read-only).
template <class T, class TB>
EMG_HD void tile_store_generic(const Level<T> &L, const T *lds, int x0, int y0, int z0, int t)
{
    using E = EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    const E ed(const_cast<T *>(lds), x0, y0, z0);
    const EdgesGlobal<T> g(L);
    const int x1 = x0 + TB::BX - 1 < L.nx - 1 ? x0 + TB::BX - 1 : L.nx - 1;   // last node of the tile
    const int y1 = y0 + TB::BY - 1 < L.ny - 1 ? y0 + TB::BY - 1 : L.ny - 1;
    const int z1 = z0 + TB::BZ - 1 < L.nz - 1 ? z0 + TB::BZ - 1 : L.nz - 1;
    constexpr int MX = (TB::BX + 1) * TB::BY * TB::BZ, MY = TB::BX * (TB::BY + 1) * TB::BZ;
    constexpr int MZ = TB::BX * TB::BY * (TB::BZ + 1);
#pragma unroll
    for (int it = 0; it < (MX + TB::THREADS - 1) / TB::THREADS; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % (TB::BX + 1), r = e / (TB::BX + 1);
        const int i = x0 - 1 + li, j = y0 + r % TB::BY, k = z0 + r / TB::BY;
        if (e < MX && i <= x1 && j <= y1 && k <= z1) g.x(i, j, k) = ed.x(i, j, k);
    }
#pragma unroll
    for (int it = 0; it < (MY + TB::THREADS - 1) / TB::THREADS; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % TB::BX, r = e / TB::BX;
        const int i = x0 + li, j = y0 - 1 + r % (TB::BY + 1), k = z0 + r / (TB::BY + 1);
        if (e < MY && i <= x1 && j <= y1 && k <= z1) g.y(i, j, k) = ed.y(i, j, k);
    }
#pragma unroll
    for (int it = 0; it < (MZ + TB::THREADS - 1) / TB::THREADS; ++it) {
        const int e = t + it * TB::THREADS;
        const int li = e % TB::BX, r = e / TB::BX;
        const int i = x0 + li, j = y0 + r % TB::BY, k = z0 - 1 + r / TB::BY;
        if (e < MZ && i <= x1 && j <= y1 && k <= z1) g.z(i, j, k) = ed.z(i, j, k);
    }
}


// ---- row-wise tile copies (tiles 32 nodes wide: THREADS = 32 * G).
// The generic copies above decode a flat element number into (i, j, k) with divisions by the
// box extents -- ~30 integer instructions per element, 1000 per thread and tile, a sixth of
// the kernel's instruction issue. Here the 32 lanes of group g = t / 32 walk row j = g of every
// plane of a box: the row index is the group, the plane index the loop counter, LDS offsets are
// compile-time constants, the global address advances by the plane stride. The one or two
// columns beyond the 32nd are copied by flat element number (2-3 elements per thread).
template <class TB> struct TileRows {
    static constexpr int G = TB::THREADS / 32;
    static constexpr bool ok = TB::BX == 32 && TB::BY == 4 && TB::BZ == 6;      // (the load fence lists this tile's values)
};
// One box of RL x NJ x NK elements (x fastest) whose element (i,j,k) is the global element
// (ox+i) + sj*(oy+j) + sk*(oz+k), valid while ox+i < lim_i, oy+j < lim_j, oz+k < lim_k. Element
// numbers are 32-bit (the launcher checks that a component has fewer than 2^31 entries).
// A thread owns NK "row" elements -- column t & 31 of row t >> 5 in every plane -- and PER
// "extra" elements of the columns beyond the 32nd, taken by flat number.
// (Free functions on plain arrays: with the walk wrapped in a struct, or with
// __builtin_amdgcn_sched_barrier anywhere near, the compiler keeps the value arrays in scratch
// memory.)
struct BoxGeo { int ox, oy, oz, sj, sk, lim_i, lim_j, lim_k; };
template <int RL, int NJ, int NK, class TB> struct BoxDims {
    static constexpr int XC = RL > 32 ? RL - 32 : 1;
    static constexpr int NX = (RL - 32) * NJ * NK, PER = (NX + TB::THREADS - 1) / TB::THREADS;
    static constexpr int PERA = PER > 0 ? PER : 1;         // array extent (no zero-length arrays)
};
#if defined(__HIPCC__)
#define EMG_UNROLL _Pragma("unroll")
#else
#define EMG_UNROLL
#endif
template <int RL, int NJ, int NK, class TB, class V>
EMG_HD void box_load(const V *g, int t, const BoxGeo q, V (&rows)[NK], V (&extra)[BoxDims<RL, NJ, NK, TB>::PERA])
{
    using D = BoxDims<RL, NJ, NK, TB>;
    const int li = t & 31, rg = t >> 5;
    const bool rowok = rg < NJ && q.ox + li < q.lim_i && q.oy + rg < q.lim_j;
    const int e0 = (q.ox + li) + q.sj * (q.oy + rg) + q.sk * q.oz;
    EMG_UNROLL
    for (int lk = 0; lk < NK; ++lk) {
        const bool ok = rowok && q.oz + lk < q.lim_k;
        rows[lk] = g[ok ? e0 + q.sk * lk : 0];             // elements that do not exist read element 0
    }
    EMG_UNROLL
    for (int it = 0; it < D::PER; ++it) {
        const int e = t + it * TB::THREADS;
        const int c = e % D::XC, r = e / D::XC, j = r % NJ, k = r / NJ, i = 32 + c;
        const bool ok = e < D::NX && q.ox + i < q.lim_i && q.oy + j < q.lim_j && q.oz + k < q.lim_k;
        extra[it] = g[ok ? (q.ox + i) + q.sj * (q.oy + j) + q.sk * (q.oz + k) : 0];
    }
}
// into the LDS box (same extents); `spare` absorbs what a thread does not have
template <int RL, int NJ, int NK, class TB, class V>
EMG_HD void box_put(V *box, int spare, int t, const V (&rows)[NK], const V (&extra)[BoxDims<RL, NJ, NK, TB>::PERA])
{
    using D = BoxDims<RL, NJ, NK, TB>;
    const int li = t & 31, rg = t >> 5;
    EMG_UNROLL
    for (int lk = 0; lk < NK; ++lk) box[rg < NJ ? li + RL * (rg + NJ * lk) : spare] = rows[lk];
    EMG_UNROLL
    for (int it = 0; it < D::PER; ++it) {
        const int e = t + it * TB::THREADS;
        const int c = e % D::XC, r = e / D::XC, j = r % NJ, k = r / NJ, i = 32 + c;
        box[e < D::NX ? i + RL * (j + NJ * k) : spare] = extra[it];
    }
}
// from an LDS box of row length LI with LJ rows per plane, in which this box sits at offset
// (DI, DJ, DK), to global memory
template <int RL, int NJ, int NK, class TB, int LI, int LJ, int DI, int DJ, int DK, class V>
EMG_HD void box_store(V *g, const V *box, int t, const BoxGeo q)
{
    using D = BoxDims<RL, NJ, NK, TB>;
    const int li = t & 31, rg = t >> 5;
    const bool rowok = rg < NJ && q.ox + li < q.lim_i && q.oy + rg < q.lim_j;
    const int e0 = (q.ox + li) + q.sj * (q.oy + rg) + q.sk * q.oz;
    EMG_UNROLL
    for (int lk = 0; lk < NK; ++lk)
        if (rowok && q.oz + lk < q.lim_k) g[e0 + q.sk * lk] = box[(li + DI) + LI * ((rg + DJ) + LJ * (lk + DK))];
    EMG_UNROLL
    for (int it = 0; it < D::PER; ++it) {
        const int e = t + it * TB::THREADS;
        const int c = e % D::XC, r = e / D::XC, j = r % NJ, k = r / NJ, i = 32 + c;
        if (e < D::NX && q.ox + i < q.lim_i && q.oy + j < q.lim_j && q.oz + k < q.lim_k)
            g[(q.ox + i) + q.sj * (q.oy + j) + q.sk * (q.oz + k)] = box[(i + DI) + LI * ((j + DJ) + LJ * (k + DK))];
    }
}

template <class T> EMG_HD double first_word(const T &v);
template <> EMG_HD double first_word<double>(const double &v) { return v; }
template <> EMG_HD double first_word<cplx>(const cplx &v) { return v.re; }

template <class T, class TB>
EMG_HD void tile_load_rows(const Level<T> &L, T *lds, int x0, int y0, int z0, int t)
{
    using E = EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    constexpr int BX = TB::BX, BY = TB::BY, BZ = TB::BZ;
    static_assert(BZ == 6 && BoxDims<BX + 2, BY + 2, BZ + 1, TB>::PER == 1 && BoxDims<BX + 2, BY + 1, BZ + 2, TB>::PER == 1,
                  "the load fence below lists the values of a 32 x 4 x 6 tile");
    const int ox = x0 - 1, oy = y0 - 1, oz = z0 - 1;
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    T rx[BZ + 2], qx[BoxDims<BX + 1, BY + 2, BZ + 2, TB>::PERA];
    T ry[BZ + 2], qy[BoxDims<BX + 2, BY + 1, BZ + 2, TB>::PERA];
    T rz[BZ + 1], qz[BoxDims<BX + 2, BY + 2, BZ + 1, TB>::PERA];
    double rc[BZ + 1], qc[BoxDims<BX + 1, BY + 1, BZ + 1, TB>::PERA];
    box_load<BX + 1, BY + 2, BZ + 2, TB>(L.ex, t, BoxGeo{ox, oy, oz, nx, nx * (ny + 1), nx, ny + 1, nz + 1}, rx, qx);
    box_load<BX + 2, BY + 1, BZ + 2, TB>(L.ey, t, BoxGeo{ox, oy, oz, nx + 1, (nx + 1) * ny, nx + 1, ny, nz + 1}, ry, qy);
    box_load<BX + 2, BY + 2, BZ + 1, TB>(L.ez, t, BoxGeo{ox, oy, oz, nx + 1, (nx + 1) * (ny + 1), nx + 1, ny + 1, nz}, rz, qz);
    box_load<BX + 1, BY + 1, BZ + 1, TB>(L.zeta, t, BoxGeo{ox, oy, oz, nx, nx * ny, nx, ny, nz}, rc, qc);
#if defined(__HIP_DEVICE_COMPILE__)
    // Load fence: every load is issued before the first LDS store. Left alone, the instruction
    // scheduler sinks the loads of a box below the stores of the previous one (fewer live
    // registers) and the copy becomes a chain of four memory latencies instead of one. The empty
    // asm statements consume one word of every loaded value and produce `dep` (always 0), which
    // the LDS addresses depend on: the loads cannot sink below them, the stores cannot rise above.
#define FW(x) "v"(first_word<T>(x))
    int dep = 0;
    asm volatile("" : "+v"(dep) : FW(rx[0]), FW(rx[1]), FW(rx[2]), FW(rx[3]), FW(rx[4]), FW(rx[5]), FW(rx[6]), FW(rx[7]), FW(qx[0]),
                 FW(ry[0]), FW(ry[1]), FW(ry[2]), FW(ry[3]), FW(ry[4]), FW(ry[5]), FW(ry[6]), FW(ry[7]), FW(qy[0]));
    asm volatile("" : "+v"(dep) : FW(rz[0]), FW(rz[1]), FW(rz[2]), FW(rz[3]), FW(rz[4]), FW(rz[5]), FW(rz[6]), FW(qz[0]),
                 "v"(rc[0]), "v"(rc[1]), "v"(rc[2]), "v"(rc[3]), "v"(rc[4]), "v"(rc[5]), "v"(rc[6]), "v"(qc[0]));
#undef FW
    lds += dep;
#endif
    box_put<BX + 1, BY + 2, BZ + 2, TB>(lds, E::ELEMS, t, rx, qx);
    box_put<BX + 2, BY + 1, BZ + 2, TB>(lds + E::NXE, E::ELEMS - E::NXE, t, ry, qy);
    box_put<BX + 2, BY + 2, BZ + 1, TB>(lds + E::NXE + E::NYE, E::ELEMS - E::NXE - E::NYE, t, rz, qz);
    box_put<BX + 1, BY + 1, BZ + 1, TB>(reinterpret_cast<double *>(lds + E::LDS_ELEMS), E::NZC, t, rc, qc);
}
// the edges attached to the tile's nodes (three boxes inside the LDS boxes) go back
template <class T, class TB>
EMG_HD void tile_store_rows(const Level<T> &L, const T *lds, int x0, int y0, int z0, int t)
{
    using E = EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    constexpr int BX = TB::BX, BY = TB::BY, BZ = TB::BZ;
    const int nx = L.nx, ny = L.ny;
    // one past the last node of the tile: an edge is written if its (i,j,k) lies below these
    const int xe = (x0 + BX < L.nx ? x0 + BX : L.nx), ye = (y0 + BY < L.ny ? y0 + BY : L.ny);
    const int ze = (z0 + BZ < L.nz ? z0 + BZ : L.nz);
    // ex: i = x0-1 .. x0+BX-1, j = y0 .., k = z0 ..: LDS box position (i, j+1, k+1)
    box_store<BX + 1, BY, BZ, TB, BX + 1, BY + 2, 0, 1, 1>(L.ex, lds, t, BoxGeo{x0 - 1, y0, z0, nx, nx * (ny + 1), xe, ye, ze});
    // ey: i = x0 .., j = y0-1 .. y0+BY-1, k = z0 ..: LDS box position (i+1, j, k+1)
    box_store<BX, BY + 1, BZ, TB, BX + 2, BY + 1, 1, 0, 1>(L.ey, lds + E::NXE, t,
                                                           BoxGeo{x0, y0 - 1, z0, nx + 1, (nx + 1) * ny, xe, ye, ze});
    // ez: i = x0 .., j = y0 .., k = z0-1 .. z0+BZ-1: LDS box position (i+1, j+1, k)
    box_store<BX, BY, BZ + 1, TB, BX + 2, BY + 2, 1, 1, 0>(L.ez, lds + E::NXE + E::NYE, t,
                                                           BoxGeo{x0, y0, z0 - 1, nx + 1, (nx + 1) * (ny + 1), xe, ye, ze});
}
template <class T, class TB> EMG_HD void tile_load(const Level<T> &L, T *lds, int x0, int y0, int z0, int t)
{
    if constexpr (TileRows<TB>::ok) tile_load_rows<T, TB>(L, lds, x0, y0, z0, t);
    else tile_load_generic<T, TB>(L, lds, x0, y0, z0, t);
}
template <class T, class TB> EMG_HD void tile_store(const Level<T> &L, const T *lds, int x0, int y0, int z0, int t)
{
    if constexpr (TileRows<TB>::ok) tile_store_rows<T, TB>(L, lds, x0, y0, z0, t);
    else tile_store_generic<T, TB>(L, lds, x0, y0, z0, t);
}

// ---- line smoothers: (p,q) = transverse PHYSICAL node indices in memory order (p faster):
//      DIR 0 (x-lines): (iy,iz)   DIR 1 (y-lines): (ix,iz)   DIR 2 (z-lines): (ix,iy)
//      colour = (p&1) | ((q&1)<<1). Lines of one colour class are numbered
//      lid = tp + cntp*tq with p = first_par(colour&1) + 2 tp, q likewise.
inline int line_np(int dir, int nx, int ny, int nz) { (void)nz; return dir == 0 ? ny : nx; }
inline int line_nq(int dir, int nx, int ny, int nz) { (void)nx; return dir == 2 ? ny : nz; }
inline int line_n0(int dir, int nx, int ny, int nz) { return dir == 0 ? nx : dir == 1 ? ny : nz; }

// (LINE_PAD, line_mid, line_padded: stencil.h -- record layout of the two-sided factorisation)
// elements at the tail of the rhs/solution scratch that absorb the stores of the surplus
// quads of the last wave of the forward / backward kernels (16 quads x 5 entries)
constexpr int LINE_DUMMY = 80;

// Split records of the fused line kernel (k_line_colour, VMODE 3) for lines whose records do not
// fit the LDS of a CU even with slot 4 left out: the `rows` record rows around the middle block
// keep their slots 0..3 in LDS, the outer rows stay in the global scratch. Host and device use
// this one rule. rows = 0: not applicable (the window must hold the rows the middle block reads).
struct LineSplit { int klo, khi; size_t lds_bytes; };
EMG_HD LineSplit line_split_rows(int n0, int n0p, int lpw, size_t elem_bytes)
{
    const size_t lds_cu = 160 * 1024;
    int rows = (int)((lds_cu / elem_bytes - LINE_DUMMY) / ((size_t)lpw * 4));
    rows &= ~1;
    if (rows > n0p) rows = n0p;
    const int mk = line_mid(n0);
    int klo = mk + 1 - rows / 2;
    if (klo > n0p - rows) klo = n0p - rows;
    if (klo < 0) klo = 0;
    LineSplit sp{klo, klo + rows, ((size_t)lpw * rows * 4 + LINE_DUMMY) * elem_bytes};
    if (rows < 8 || sp.klo > mk - 1 || sp.khi < mk + 3 || sp.khi > n0p) sp = LineSplit{0, 0, 0};
    return sp;
}

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
// block records of a direction (all classes), and the elements of the N records of the wide form (kernels.hip:
// k_line_wide) that follow the T records in the factor buffer of a level that can take it
inline size_t line_records(int dir, int nx, int ny, int nz)
{
    return (size_t)line_padded(line_n0(dir, nx, ny, nz)) * line_total(dir, nx, ny, nz);
}
inline size_t line_nfac_elems(int dir, int nx, int ny, int nz)
{
    const size_t rec = line_records(dir, nx, ny, nz);
    return line_wide_capable(line_n0(dir, nx, ny, nz), rec) ? 16 * rec : 0;
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

template <class T, int DIR, class FT = T>
EMG_HD void line_setup_thread(const Level<T> &L, int colour, int cntp, int cntq, int tp, int tq, FT *fac,
                              double *lfac)
{
    int i1, i2, lid;
    if (!line_of_thread<DIR>(colour, cntp, cntq, tp, tq, i1, i2, lid)) return;
    line_setup<T, DIR, FT>(L, i1, i2, fac, lfac, cntp * cntq, lid, line_padded(Axes<T, DIR>(L).n0()));
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

// HIP kernels (gfx950) + C ABI of the emg3d multigrid inner loop. See include/emg3d_amd.h
// for the contract of every entry point and the reference lines each one replaces.
//
// Launch geometry: thread x runs along the contiguous (x) axis of the Fortran-ordered
// arrays wherever the work decomposition allows it, workgroups are 256 threads (4 waves of
// 64). All kernels are HBM-bandwidth bound fp64 stencil / small-dense-solve work: there is
// no dense contraction and therefore no MFMA use.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "../../include/emg3d_amd.h"
#include "launch.h"

using emg::cplx;

namespace {

thread_local std::string g_err;

int fail(int code, const char *what)
{
    g_err = what;
    return code;
}

int hipfail(hipError_t e, const char *where)
{
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (int)e;
}

#define HIP_TRY(call)                                      \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hipfail(e_, #call);   \
    } while (0)

template <class T> emg::Level<T> to_level(const emg3d_level *lv)
{
    emg::Level<T> L;
    L.nx = lv->nx; L.ny = lv->ny; L.nz = lv->nz;
    L.ex = (T *)lv->ex; L.ey = (T *)lv->ey; L.ez = (T *)lv->ez;
    L.sx = (const T *)lv->sx; L.sy = (const T *)lv->sy; L.sz = (const T *)lv->sz;
    L.eta_x = (const T *)lv->eta_x; L.eta_y = (const T *)lv->eta_y; L.eta_z = (const T *)lv->eta_z;
    L.zeta = lv->zeta;
    L.ihx = lv->ihx; L.ihy = lv->ihy; L.ihz = lv->ihz;
    L.batch = lv->batch > 1 ? lv->batch : 1;
    L.bstride = lv->batch > 1 ? (size_t)lv->batch_stride : 0;
    L.flags = lv->flags;
    return L;
}

using emg::cdiv;
using emg::cnt_par;
using emg::sc_flags;
using emg::ScDirs;

inline dim3 d3(emg::Dim3 d) { return dim3(d.x, d.y, d.z); }

// Plane-slab thickness of the point smoother's launch schedule (launch.h:
// gs_point_schedule). 0 = plain four launches per sweep. Tunable at run time through
// emg3d_set_option("point_slab", T); the result does not depend on it.
int g_point_slab = 0;
// Point smoother: levels with at least this many interior nodes use the tiled schedule
// (launch.h) -- it changes the ORDER of the Gauss-Seidel sweep (documented in
// include/emg3d_amd.h), so it is part of the algorithm, not a free tuning knob: the
// oracle applies the same rule. <= 0 never, 1 always.
int g_point_tile_min = 1 << 20;
// Line smoothers: 0 = three launches per colour (rhs, forward, backward), 1 = one fused
// launch per colour, 2 = fused when the colour class has at most g_line_fuse_max lines.
// Fused is faster at every size measured (256^3: 8.4-9.0 against 9.5-10.0 ms per two sweeps).
int g_line_fuse = 2;
int g_line_fuse_max = 1 << 30;
// fused line kernel: keep the right-hand-side / solution records of a workgroup's lines in
// LDS when they fit into this many bytes (0 = never)
// skip the colour pass that repeats the previous sweep's last one (bit-identical results)
int g_skip_repeat = 1;
// tiled point smoother: the tiles where two consecutive sweeps meet run both on one LDS copy
int g_tile_fuse = 1;
int g_line_lds = 1;
// tiled point smoother: software prefetch of the next colour step's inputs (0 none, 1 source, 2 source + eta sums)
int g_point_prefetch = 0;
// residual kernel: planes a workgroup walks on large levels (1 = one plane per workgroup)
int g_residual_zb = 8;
// ... and carries the operands a cell shares with the cell below it in registers (1, default; 0: every cell loads all of its own)
int g_residual_roll = 1;
// fused line kernel with the records in the global scratch (the largest levels): the instantiation
// that is held to 256 registers, so that two workgroups share a CU and overlap their phases
int g_line_occ2 = 0;
// point smoother: levels with at most this many interior nodes run all passes of a call in one
// single-workgroup launch (k_gs_point_small); 0: off
int g_point_small = 512;
// fused line kernel: lines per workgroup (0 = automatic: 4, 8 or 16)
int g_line_lpw = 0;
// the levels whose records do not fit the LDS of a CU: k_line_stream -- right-hand sides, coupling entries and
// w records staged through LDS rings by producer waves while the chain waves substitute. 2 (default): wherever
// the full records of the workgroup's lines do not fit (lines of ~128 blocks and more with 16 lines per
// workgroup; since round 4, when the producers took over the coupling entries: 128-block lines -6 % per launch
// against k_line_colour with slots 0..3 in LDS); 1: only where not even slots 0..3 fit (~160 blocks and more);
// 0: k_line_colour everywhere
int g_line_stream = 2;
// sequence of the colour passes of the LINE smoothers (launch.h: line_sweep_colour): 1 (default) cyclic
// 1,2,3,0,1,...; 0 mirrored sweeps (0,2,3,1 forward / its reverse backward: rounds 1-2); 2 the classes
// 1,2,3,0 in every sweep (eight launches per two sweeps). Like
// point_tile_min this selects the ORDER of the Gauss-Seidel sweep, i.e. it is part of the algorithm
// definition (converged fields are the same; per-sweep values and cycle counts are not).
int g_line_order = 1;
int g_line_stream_r = 0;           // rows per chunk of the ring (0: 16)
// several right-hand sides (emg3d_level::batch > 1): levels whose colour passes would keep their records in
// the global scratch and whose lines have at least this many blocks run k_line_stream -- groups of up
// to four right-hand sides per workgroup, the factors fetched once per group (<= 0: never)
int g_line_stream_bmin = 64;
// one source: the coupling entries of a block (8 reals) recomputed by the producer waves from zeta / h and handed
// to the chain waves through a second LDS ring instead of being fetched from the lfac records (1, default; 0: fetched)
int g_line_stream_lf = 1;
// the wide form of the line pass (k_line_wide: four-unknown chains on sixteen lanes per half-line, one thread per
// block for everything else) on lines of at most this many blocks, where the level holds the N records (launch.h:
// line_wide_capable -- a function of the level's shape alone, so that buffers sized once stay valid whatever the
// option says); 0: never. Default 33 (round 5): on 4 ... 32-block lines a launch takes 5.7 / 5.7 / 8.4 / 17.3 us against 6.0 /
// 8.1 / 12.0 / 20.4 of k_line_colour, on 64-block lines it is no faster (profiles/r05_small_level_experiments.txt)
int g_line_wide = 33;
// block threads of a k_line_wide workgroup (+ 64 for the middle blocks): 0 (default) 192, or 256 where that gives a
// workgroup more lines (lines of 26 and more blocks: 8 instead of 6 lines of 32 blocks -- 2 048 lines per class are
// then one workgroup per CU: 21.1 -> 17.3 us per launch at 256 x 32 x 32, k_line_colour 20.4; on shorter lines the fifth
// wave only adds to the barriers: 5.8 -> 6.5 us); 192 / 256: forced
int g_line_wide_bt = 0;
// TIMING EXPERIMENTS ONLY (wrong results): bit 0: the records of all blocks of a line alias one row of
// the global scratch -- what the level-0 pass would cost if its right-hand-side / solution records
// never left the chip
int g_line_debug = 0;
// COMPACT line factors (k_line_stream): the T records and the w records of the streamed colour passes stored in
// single precision (120 + 2 x 40 instead of 240 + 2 x 80 B per block and pass; every operation in fp64). 0 (default):
// where the level asks for it (emg3d_level::flags & EMG3D_LEVEL_LINE_COMPACT -- the caller's promise that the level
// solves a correction equation, include/emg3d_amd.h); 1: on every level whose direction streams (timing / tests);
// -1: never
int g_line_compact = 0;
// ... depth of the chain waves' factor prefetch ring in the compact kernel (4 or 8 block steps ahead: a compact ring
// entry holds 12 instead of 24 registers). 0 (default): 8 for x-lines, 4 for y- and z-lines -- same-box A/B at 256^3,
// ms per launch x / y / z: fp64 records 0.885 / 0.955 / 0.969, compact with 4: 0.817 / 0.757 / 0.748, with 8: 0.740 /
// 0.794 / 0.789 (the producers of y / z lines, whose gathers use half of every cache line, are what the chains wait
// for: more chain loads in flight delay them; x-line producers read whole lines)
int g_line_compact_rd = 0;
// (Measured and not kept as options -- profiles/r06_compact_line_kernel_ab.txt: two workgroups of four waves per CU with 8
// rows per chunk, 0.915 / 0.761 / 0.773 ms per launch against 0.817 / 0.757 / 0.748; four producer waves instead of six,
// 0.732 / 0.780 / 0.770 against 0.697 / 0.766 / 0.764; the w records of the rows nearest the middle block in LDS, +4-10 %.)
// ... also on the levels that run k_line_colour with lines of more than LINE_SHORT blocks (1, default; 0: streamed levels only)
int g_line_compact_colour = 1;
constexpr int LS_PROD_COMPACT = 384;      // producer threads of the compact single-source kernel: six waves

// eta edge sums of the tiled point smoother: 8-byte storage (launch.h: tile_pst_*) for real
// fields and for complex fields whose eta are purely imaginary (emg3d_level::flags)
template <class T> bool pst_stored_half(int flags);
template <> bool pst_stored_half<double>(int) { return true; }
template <> bool pst_stored_half<cplx>(int flags) { return (flags & emg::LEVEL_ETA_IMAG) != 0; }
// ... and the storage mode of the buffer (launch.h: PST_*): single precision on levels flagged LEVEL_POINT_COMPACT
// (option point_compact: 0 by the flag, 1 always, -1 never)
int g_point_compact = 0;
template <class T> int pst_mode(int flags)
{
    const bool compact = g_point_compact > 0 || (g_point_compact == 0 && (flags & emg::LEVEL_POINT_COMPACT));
    if (pst_stored_half<T>(flags)) return compact ? emg::PST_HALF_F32 : emg::PST_HALF;
    return compact ? emg::PST_FULL_F32 : emg::PST_FULL;
}

// Opt a kernel in to more than 64 KB of dynamic LDS. The attribute belongs to the (kernel,
// device) pair: remembered per device ordinal, so a process that drives several GPUs sets it on
// each of them.
hipError_t allow_lds(const void *kernel, size_t bytes)
{
    constexpr int MAXDEV = 64, MAXK = 256;
    static const void *seen[MAXDEV][MAXK];
    static size_t seen_bytes[MAXDEV][MAXK];
    static int nseen[MAXDEV];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < MAXDEV) {
        for (int i = 0; i < nseen[dev]; ++i)
            if (seen[dev][i] == kernel && seen_bytes[dev][i] >= bytes) return hipSuccess;
    }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < MAXDEV) {
        for (int i = 0; i < nseen[dev]; ++i)
            if (seen[dev][i] == kernel) { seen_bytes[dev][i] = bytes; return e; }      // (a larger request of a known kernel)
    }
    if (e == hipSuccess && dev >= 0 && dev < MAXDEV && nseen[dev] < MAXK) {
        seen[dev][nseen[dev]] = kernel;
        seen_bytes[dev][nseen[dev]] = bytes;
        ++nseen[dev];
    }
    return e;
}

// compute units of the current device
int compute_units()
{
    constexpr int MAXDEV = 64;
    static int cus[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// ----------------------------------------------------------------------------- kernels --

// Point smoother, one colour. colour = ((ix+iz)&1) | (((iy+iz)&1)<<1).
// pst: eta edge sums from k_point_setup, or nullptr (formed on the fly).
template <class T>
__global__ __launch_bounds__(256) void k_gs_point(emg::Level<T> L, const T *pst, int colour, int iz0, int izn)
{
    // grid.z = planes x right-hand sides
    const int b = blockIdx.z / izn, z = blockIdx.z - b * izn;
    emg::gs_point_thread<T>(emg::source_level(L, b), pst, colour, iz0, blockIdx.x * blockDim.x + threadIdx.x,
                            blockIdx.y * blockDim.y + threadIdx.y, z);
}

// Point smoother on a level small enough for ONE workgroup: all colour passes of all `nu` sweeps in
// one launch, workgroup barriers between the passes (the waves of a workgroup share their CU's L1:
// what one wave stored, the others read after the barrier). `passes`: the colour classes in the
// order of the plain schedule, two bits each (the launcher leaves out the repeated class of
// skip_repeat). The coarsest levels of a W-cycle are visited 2^levels times per cycle; their four
// to seven launches of a microsecond of work each become one.
template <class T>
__global__ __launch_bounds__(256) void k_gs_point_small(emg::Level<T> L, const T *pst, unsigned long long passes, int npass)
{
    const int hx = (L.nx + 1) / 2, hy = (L.ny + 1) / 2, nzp = L.nz - 1;      // (gx, gy) cover both parities
    const int n = hx * hy * nzp;
    for (int p = 0; p < npass; ++p) {
        const int colour = (int)((passes >> (2 * p)) & 3ULL);
        for (int i = threadIdx.x; i < n; i += 256) {
            const int gx = i % hx, r = i / hx, gy = r % hy, gz = r / hy;
            emg::gs_point_thread<T>(L, pst, colour, 1, gx, gy, gz);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// eta edge sums of a level (stencil.h: point_setup_cell), one thread per extended cell
template <class T> __global__ __launch_bounds__(256) void k_point_setup(emg::Level<T> L, T *pst)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix <= L.nx && iy <= L.ny) emg::point_setup_cell<T>(L, pst, ix, iy, blockIdx.z);
}

// workgroup barrier that orders LDS traffic only: global loads issued before it (the
// prefetch of the next node's inputs) stay in flight across it
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Point smoother, tiled schedule (launch.h): one workgroup = one tile of one tile colour;
// the tile's edges live in LDS while the four node colours run on it. The model/source
// inputs of the next colour's node are fetched while the current node is solved.
// ST: where the eta edge sums come from: 0 formed on the fly from eta; 2 tile-major buffer of
// k_point_setup_tile, full values; 3 the same, stored halves (8 bytes: launch.h tile_pst_*).
// PF: software prefetch -- the global inputs (source; PF = 2: eta sums too) of the NEXT colour
// step are requested before the current node is solved, so their latency overlaps the ~600
// fp64 instructions of the 6x6 solve inside the wave, not only across waves.
template <class T, class TB, int ST, bool BATCH, int PFV>
__global__ __launch_bounds__(TB::THREADS, 2) void k_gs_point_tile(emg::Level<T> L, const void *pst, emg::TilePair P,
                                                                  int colours, int nsteps)
{
    // grid.z = (tiles of colour P.tc[0], then of P.tc[1]) [x right-hand sides if BATCH: a
    // separate instantiation, the single-source kernel keeps its register count]
    const int gz = P.gz[0] + P.gz[1];
    int bz = blockIdx.z;
    if (BATCH) {
        const int bsrc = blockIdx.z / gz;
        bz -= bsrc * gz;
        L = emg::source_level(L, bsrc);
    }
    const int which = bz >= P.gz[0] ? 1 : 0;
    bz -= which ? P.gz[0] : 0;
    if ((int)blockIdx.x >= P.gx[which] || (int)blockIdx.y >= P.gy[which]) return;
    const int tc = P.tc[which];
    extern __shared__ double2 tile_smem[];
    T *lds = reinterpret_cast<T *>(tile_smem);
    using E = emg::EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    const int tx = (tc & 1) + 2 * blockIdx.x, ty = ((tc >> 1) & 1) + 2 * blockIdx.y, tz = ((tc >> 2) & 1) + 2 * bz;
    const int x0 = 1 + tx * TB::BX, y0 = 1 + ty * TB::BY, z0 = 1 + tz * TB::BZ;
    const int ntx = (L.nx - 1 + TB::BX - 1) / TB::BX, nty = (L.ny - 1 + TB::BY - 1) / TB::BY;
    const int t = threadIdx.x;
    const E ed(lds, x0, y0, z0);
    auto node = [&](int cc, int &ix, int &iy, int &iz, int &colour) {
        colour = (colours >> (2 * cc)) & 3;
        return emg::tile_node<TB>(L.nx, L.ny, L.nz, x0, y0, z0, colour, t, ix, iy, iz);
    };
    auto load_eta = [&](int ix, int iy, int iz, int colour, emg::PointIn<T> &in) {
        if (ST == 0) emg::point_load_eta<T, false>(L, nullptr, ix, iy, iz, in);
        else emg::tile_pst_load<T, TB, ST>(pst, ntx, nty, tx, ty, tz, colour, t, in);
    };
    // PFV = 3: paired source loads. A thread's four nodes (one per node colour) are the 2 x 2
    // patch (x0 + 2 jx + {0,1}, y0 + 2 jy + {0,1}) of its plane; the two nodes of a row differ
    // in the low colour bit. Fetched per node, the source values of a row are stride-2 gathers
    // that touch every cache line of the row in BOTH steps. Here the first of the two steps
    // fetches the values of both nodes (same lines, same instruction stream) and holds the
    // partner's six values (24 registers per row) until its step comes.
    constexpr bool PAIR = PFV == 3;
    constexpr int PF = PAIR ? 0 : PFV;
    T held_lo[6], held_hi[6];                  // partner values of the row with colour bit 1 = 0 / 1
    int held_lo_c = -1, held_hi_c = -1;        // the colour they belong to (-1: none)
    auto pair_source = [&](int ix, int iy, int iz, int colour, bool ok, T (&held)[6], int &held_c, emg::PointIn<T> &in) {
        if (held_c == colour) {
#pragma unroll
            for (int r = 0; r < 6; ++r) in.s[r] = held[r];
            held_c = -1;
            return;
        }
        emg::point_load_source<T>(L, ix, iy, iz, in);
        int px, py, pz;
        const bool pok = emg::tile_node<TB>(L.nx, L.ny, L.nz, x0, y0, z0, colour ^ 1, t, px, py, pz);
        if (!pok) { px = ix; py = iy; pz = iz; }
        (void)ok;
        emg::PointIn<T> q;
        emg::point_load_source<T>(L, px, py, pz, q);
#pragma unroll
        for (int r = 0; r < 6; ++r) held[r] = q.s[r];
        held_c = colour ^ 1;
    };
    emg::PointIn<T> in;
    int ix, iy, iz, colour;
    bool ok = node(0, ix, iy, iz, colour);
    if (PF >= 1) emg::point_load_source<T>(L, ix, iy, iz, in);      // issued ahead of the tile copy
    if (PF >= 2) load_eta(ix, iy, iz, colour, in);
    emg::tile_load<T, TB>(L, lds, x0, y0, z0, t);
    lds_barrier();
    // two workgroups per CU (launch bounds: <= 256 registers, 2 x 79 KB of LDS): while one
    // waits for the inputs of its next node, the other one computes
#pragma unroll 1
    for (int cc = 0; cc < nsteps; ++cc) {     // node colours, two bits each (4, or 7-8 for two fused sweeps)
        if (PAIR) {
            if (colour & 2) pair_source(ix, iy, iz, colour, ok, held_hi, held_hi_c, in);
            else pair_source(ix, iy, iz, colour, ok, held_lo, held_lo_c, in);
        } else if (PF < 1) emg::point_load_source<T>(L, ix, iy, iz, in);
        if (PF < 2) load_eta(ix, iy, iz, colour, in);
        emg::point_load_zeta<T>(emg::ZetaTile<E>{ed}, ix, iy, iz, in);
        // the next step's node and its global inputs (the last step asks for its own again)
        emg::PointIn<T> nxt;
        int jx, jy, jz, ncol;
        const bool nok = node(min(cc + 1, nsteps - 1), jx, jy, jz, ncol);
        if (PF >= 1) emg::point_load_source<T>(L, jx, jy, jz, nxt);
        if (PF >= 2) load_eta(jx, jy, jz, ncol, nxt);
        if (ok) emg::point_update<T, E>(L, in, ed, ix, iy, iz);
        lds_barrier();
        if (PF >= 1) {
#pragma unroll
            for (int r = 0; r < 6; ++r) in.s[r] = nxt.s[r];
        }
        if (PF >= 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r) in.st[r] = nxt.st[r];
        }
        ix = jx; iy = jy; iz = jz; colour = ncol; ok = nok;
    }
    emg::tile_store<T, TB>(L, lds, x0, y0, z0, t);
}

// eta edge sums in tile-major order (launch.h: tile_pst_setup), one workgroup per tile
template <class T, class TB, int MODE>
__global__ __launch_bounds__(TB::THREADS) void k_point_setup_tile(emg::Level<T> L, void *pst)
{
    emg::tile_pst_setup<T, TB, MODE>(L, pst, gridDim.x, gridDim.y, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x);
}

// does any of n complex values have a non-zero real part? (sets *flag)
__global__ __launch_bounds__(256) void k_any_real_part(const cplx *a, size_t n, int *flag)
{
    bool any = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) any |= a[i].re != 0.0;
    if (__syncthreads_or(any) && threadIdx.x == 0) atomicOr(flag, 1);
}

// Line smoothers (stencil.h: line_setup / line_rhs / line_forward / line_backward /
// line_scatter); thread mappings in launch.h.
// Two waves per 64 lines: wave 0 factorises the top chains, wave 1 the bottom chains (they are
// independent and the setup is one long dependent recurrence per line); the LDL^T factors of
// the bottom chain's last block go through LDS to wave 0, which finishes with the middle block.
// All four colour classes in one launch (grid.z = colour): the recurrence of a line is one long
// dependent chain, and a class alone puts one wave pair on every CU -- four of them interleave.
struct SetupClasses { int cntp[4], cntq[4]; size_t fac_off[4], lfac_off[4]; };
// FT: storage type of the T records (T, or emg::compact_of<T>: rounded when stored)
template <class T, int DIR, class FT = T>
__global__ __launch_bounds__(128) void k_line_setup(emg::Level<T> L, SetupClasses S, FT *fac0, double *lfac0)
{
    __shared__ T xch[15][64];
    const int colour = blockIdx.z, cntp = S.cntp[colour], cntq = S.cntq[colour];
    FT *const fac = fac0 + S.fac_off[colour];
    double *const lfac = lfac0 + S.lfac_off[colour];
    const int role = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i1, i2, lid;
    const bool valid = emg::line_of_thread<DIR>(colour, cntp, cntq, blockIdx.x * 64 + lane, blockIdx.y, i1, i2, lid);
    const emg::LineStore<T, FT> st{fac, lfac, cntp * cntq, lid};
    const int n0p = emg::line_padded(emg::Axes<T, DIR>(L).n0());
    T C[10], dinv[5];
#pragma unroll
    for (int j = 0; j < 10; ++j) C[j] = emg::zero<T>();
#pragma unroll
    for (int j = 0; j < 5; ++j) dinv[j] = T(1.0);
    if (valid) {
        if (role == 0) {
            emg::line_setup_top<T, DIR, FT>(L, i1, i2, st, C, dinv);
        } else {
            emg::line_setup_bottom<T, DIR, FT>(L, i1, i2, st, n0p, C, dinv);
#pragma unroll
            for (int j = 0; j < 10; ++j) xch[j][lane] = C[j];
#pragma unroll
            for (int j = 0; j < 5; ++j) xch[10 + j][lane] = dinv[j];
        }
    }
    __syncthreads();
    if (valid && role == 0) {
        T Cb[10], db[5];
#pragma unroll
        for (int j = 0; j < 10; ++j) Cb[j] = xch[j][lane];
#pragma unroll
        for (int j = 0; j < 5; ++j) db[j] = xch[10 + j][lane];
        emg::line_setup_middle<T, DIR, FT>(L, i1, i2, st, C, dinv, Cb, db);
    }
}

template <class T, int DIR>
__global__ __launch_bounds__(64) void k_line_rhs(emg::Level<T> L, int colour, int cntp, int cntq, T *vec)
{
    emg::line_rhs_thread<T, DIR>(L, colour, cntp, cntq, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y,
                                 blockIdx.z, vec);
}

// x-lines: the field is contiguous ALONG the line, the records are contiguous ACROSS the
// lines. A 16 (blocks) x 16 (lines) tile is assembled with the threads running along the
// line (256-B segments of ex/ey/ez/s per row), transposed through LDS, and written with the
// threads running across the lines (16 records = 1280 contiguous bytes per block).
template <class T>
__global__ __launch_bounds__(256) void k_line_rhs_xt(emg::Level<T> L, int colour, int cntp, int cntq, int n0p,
                                                     T *vec)
{
    __shared__ T tile[16][16][5];
    const int a = threadIdx.x & 15, b = threadIdx.x >> 4;
    const int tq = blockIdx.y;
    {   // phase 1: a -> block, b -> line
        const int k = blockIdx.z * 16 + a, tp = blockIdx.x * 16 + b;
        int i1, i2, lid;
        T rhs[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) rhs[r] = emg::zero<T>();
        if (k < L.nx && emg::line_of_thread<0>(colour, cntp, cntq, tp, tq, i1, i2, lid)) {
            const emg::Axes<T, 0> A(L);
            emg::line_rhs<T, 0>(A, k, i1, i2, rhs);
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) tile[b][a][r] = rhs[r];
    }
    __syncthreads();
    {   // phase 2: a -> line, b -> block
        const int k = blockIdx.z * 16 + b, tp = blockIdx.x * 16 + a;
        if (k < n0p && tp < cntp) {
            T *o = vec + ((size_t)k * (cntp * cntq) + (tp + cntp * tq)) * 5;
#pragma unroll
            for (int r = 0; r < 5; ++r) o[r] = tile[a][b][r];
        }
    }
}

// ---- forward / backward substitution along the lines: four lanes per line -------------
// The recurrence along a line is sequential and latency-bound; its per-block critical
// path is what sets the kernel time. Lane j of a quad owns ROW j of the block: it loads
// row j of T_k (5 entries, straight from the packed symmetric record), computes entry j of
// the 5-vectors, and the quad exchanges the 4+1 entries with DPP quad_perm moves; entry 4
// is formed from the partial products T(j,4) c_j that the lanes already hold (quad sum).
// Records of the next QD blocks are kept in flight in a register ring, so that a wave has
// 16 lines x QD blocks outstanding instead of 64 lines x 1 block.
// ---- the arithmetic of the chain steps, written out --------------------------------------------------
// From here to the end of the line kernels floating-point contraction is OFF and every operation of the
// forward / middle / backward steps is spelled as a multiplication, an addition or a fused multiply-add
// (xop::mul / add / sub, emg::mad / nmad). With contraction left to the compiler, `a b - c d` may become
// fma(a, b, -(c d)) or fma(-c, d, a b) depending on how often the products are used in the surrounding code
// -- the same source expression then rounds differently in the single-source kernels, in the batched
// kernel, and even between the unrolled right-hand sides of one batched kernel (measured: 4e-13 after
// three sweeps). Spelled out, every line kernel performs the same operations in the same order: one
// source gives the same bits whichever kernel runs it, alone or in a batch.
#pragma clang fp contract(off)
namespace xop {
__device__ __forceinline__ double mul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx mul(double a, cplx b) { return cplx(a * b.re, a * b.im); }
__device__ __forceinline__ cplx mul(cplx a, cplx b)
{
    return cplx(__builtin_fma(-a.im, b.im, a.re * b.re), __builtin_fma(a.im, b.re, a.re * b.im));
}
__device__ __forceinline__ double add(double a, double b) { return a + b; }
__device__ __forceinline__ cplx add(cplx a, cplx b) { return cplx(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ double sub(double a, double b) { return a - b; }
__device__ __forceinline__ cplx sub(cplx a, cplx b) { return cplx(a.re - b.re, a.im - b.im); }
}  // namespace xop

template <int CTRL> __device__ __forceinline__ double dpp_move(double x)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ cplx dpp_move(cplx x)
{
    return cplx(dpp_move<CTRL>(x.re), dpp_move<CTRL>(x.im));
}
// broadcast lane LN of every quad
template <int LN, class T> __device__ __forceinline__ T quad_bcast(T x) { return dpp_move<LN * 0x55>(x); }
// sum over the four lanes of every quad (result in all lanes)
template <class T> __device__ __forceinline__ T quad_sum(T x)
{
    x = xop::add(x, dpp_move<0xB1>(x));   // quad_perm:[1,0,3,2]
    x = xop::add(x, dpp_move<0x4E>(x));   // quad_perm:[2,3,0,1]
    return x;
}
// cyclic shift inside every quad: lane j receives the value of lane (j + R) & 3
template <int R, class T> __device__ __forceinline__ T quad_rot(T x)
{
    static_assert(R >= 1 && R <= 3, "rotation 1..3");
    return dpp_move<R == 1 ? 0x39 : (R == 2 ? 0x4E : 0x93)>(x);   // quad_perm [1,2,3,0] / [2,3,0,1] / [3,0,1,2]
}
// Lane j's share of one block record (all loads unconditional, addresses per lane).
// The row of T_k is kept in ROTATED order, t[r] = T_k(j, (j + r) & 3): the quad all-gathers a
// 4-vector with three cyclic shifts (12 DPP moves) instead of four broadcasts (16), lane j
// then holds entry (j + r) & 3 in its r-th register.
// FT: the type the T entries are HELD in -- T, or their compact storage type (emg::compact_of<T>): the ring then keeps
// the values as they were loaded (half the registers) and the step functions widen them where they use them. (Widened
// right after the load, the compiler converts a whole ring pass at the end of the loop body behind s_waitcnt
// vmcnt(0): the prefetch distance is gone -- measured 8 % slower than fp64 records.)
template <class T, class FT = T> struct QuadRow {
    FT t[5];         // T_k(j, (j+r)&3), r = 0..3;  t[4] = T_k(j, 4)
    FT t44;          // T_k(4,4)
    T v, v4;         // vec[j], vec[4]
    double bA;       // lane j >= 1: B_k(0, j);  lane 0: B_k(0, 4)
    double bD;       // lane j >= 1: B_k(j, j)   (lane 0: B_k(1,1), masked out)
    double b04, d4;  // B_k(0, 4), B_k(4, 4)
    __device__ __forceinline__ void load_b(const double *lfac, size_t rec, int j)
    {
        const double *lf = lfac + rec * 8;
        bA = lf[j == 0 ? 3 : j - 1];
        bD = lf[4 + max(j, 1) - 1];
        b04 = lf[3];
        d4 = lf[7];
    }
    // RECS = false: factors and coupling only (the streamed kernel takes its right-hand sides from LDS)
    // LF = false: without the coupling entries (k_line_stream for one source takes them from its LDS ring)
    template <class A, bool RECS = true, bool LF = true> __device__ __forceinline__ void load(const A &a, int k)
    {
        const char *f = a.fac + (size_t)k * a.frow, *lf = a.lfac + (size_t)k * a.lrow;
        static_assert(std::is_same<FT, typename A::fac_t>::value, "QuadRow: held type = storage type of the T records");
#pragma unroll
        for (int r = 0; r < 5; ++r) t[r] = *reinterpret_cast<const FT *>(f + a.ft[r]);
        t44 = *reinterpret_cast<const FT *>(f + a.ft[5]);
        if constexpr (!RECS) {
        } else if constexpr (A::split) {
            // split records: the slot is in LDS or in the global scratch, depending on the row --
            // both are read (the one that does not apply at a fixed valid address), no branch
            const bool in = a.in_lds(k);
            const T *const pl = in ? a.pvj(k) : a.ldum;
            const T *const pg = in ? a.gdum : a.gpvj(k);
            const T vl = *pl, vg = *pg;
            v = in ? vl : vg;
        } else {
            v = *a.pvj(k);
        }
        if constexpr (RECS) v4 = *a.pv4(k);
        if constexpr (LF) {
            bA = *reinterpret_cast<const double *>(lf + a.la);
            bD = *reinterpret_cast<const double *>(lf + a.ld);
            b04 = *reinterpret_cast<const double *>(lf + a.l04);
            d4 = *reinterpret_cast<const double *>(lf + a.l4);
        }
    }
    // the coupling entries of this lane from the eight values of a record (lfac layout)
    __device__ __forceinline__ void take_b(const double *lf, int j)
    {
        bA = lf[j == 0 ? 3 : j - 1];
        bD = lf[4 + max(j, 1) - 1];
        b04 = lf[3];
        d4 = lf[7];
    }
};

// QD (template parameter of the walks below) = blocks in flight per line = padding granule of the
// line's records: emg::line_pad(n0), 4 or 2

// One half-chain of the two-sided line solve (stencil.h). HALF 0: the top half, standard
// blocks k = 0 .. m-1 walked upwards; HALF 1: the bottom half, mirrored blocks walked from
// the padded far end n0p-1 down to m+2. A half has `steps` blocks (a multiple of QD,
// possibly 0); step i works on block kof(i).
template <int HALF> struct HalfWalk {
    int mk, n0p, steps;
    __device__ __forceinline__ HalfWalk(int n0, int n0p_) : mk(emg::line_mid(n0)), n0p(n0p_)
    {
        steps = HALF ? n0p - 2 - mk : mk;
    }
    // forward pass: towards the middle
    __device__ __forceinline__ int fwd(int i) const { return HALF ? n0p - 1 - i : i; }
    // backward pass: away from the middle
    __device__ __forceinline__ int bwd(int i) const { return HALF ? mk + 2 + i : mk - 1 - i; }
    __device__ __forceinline__ int clampi(int i) const { return max(min(i, steps - 1), 0); }
};
// Where the right-hand-side / solution records of the lines live. Slots 0..3 of (block k,
// line): base + (k * stride + line - line0) * width + r; slot 4: base4 + (k * stride4 +
// line - line04) * 5 + 4.
//   global scratch:       both in `vec`, stride = lines of the colour class, width 5
//   LDS (fused kernel):   the workgroup's 16 lines, stride 16, width 5, both in LDS
//   LDS, partial:         slots 0..3 in LDS (width 4), slot 4 stays in the global scratch --
//                         for lines so long that 16 full records exceed the LDS of a CU
template <class T> struct VecRef {
    T *base;
    int stride, line0, width;
    T *base4;
    int stride4, line04;
    // split records (fused kernel, VMODE 3): rows klo <= k < khi keep their slots 0..3 in LDS
    // (`base`, shifted by -klo rows, width 4), all other rows in the global scratch (`gbase`, laid
    // out like VecRef::global); slot 4 of every row is in the global scratch
    T *gbase;
    int gstride, klo, khi;
    __device__ __forceinline__ T *pg(int k, int line, int r) const { return gbase + ((size_t)k * gstride + line) * 5 + r; }
    __device__ __forceinline__ T *p(int k, int line, int r) const
    {
        return base + ((size_t)k * stride + (line - line0)) * width + r;
    }
    __device__ __forceinline__ T *p4(int k, int line) const
    {
        return base4 + ((size_t)k * stride4 + (line - line04)) * 5 + 4;
    }
    static __device__ __forceinline__ VecRef global(T *vec, int nlines)
    {
        return VecRef{vec, nlines, 0, 5, vec, nlines, 0, vec, nlines, 0, 0};
    }
};
// Addresses of one lane inside the half-chain loops, split into a per-block part that is
// uniform over the wave (block index x row size: scalar registers, scalar multiplies) and a
// loop-invariant, non-negative 32-bit per-lane part. Without the split every load of every
// step pays 64-bit per-lane multiplies (v_mad_u64_u32: quarter rate) -- a quarter of the
// issue slots of a step.
// FT / WT: storage types of the T records / of the right-hand-side and w records (T, or emg::compact_of<T>)
template <class T, int HALF, bool SPLIT = false, class FT = T, class WT = T> struct LaneAddr {
    static constexpr bool split = SPLIT;
    using fac_t = FT;
    const char *fac, *lfac;      // uniform
    size_t frow, lrow;           // bytes of one block row of the factor arrays (all lines)
    unsigned ft[6];              // lane byte offsets in a fac row: T(j,(j+r)&3) r=0..3, T(j,4), T(4,4)
    unsigned la, ld, l04, l4;    // lane byte offsets in an lfac row: bA, bD, B(0,4), B(4,4)
    char *vb, *vb4;              // uniform bases of the vec slots 0..3 / slot 4
    size_t vrow, vrow4;          // bytes of one block row of the records
    unsigned vj, v4;             // lane byte offsets of slot j / slot 4
    // SPLIT: the global home of slots 0..3, the row range held in LDS, this lane's row offset, and
    // the fixed addresses read when a slot lives in the other space
    char *gvb;
    size_t gvrow;
    unsigned gvj;
    int klo, khi, radd;
    const WT *ldum, *gdum;
    __device__ __forceinline__ LaneAddr(const FT *f, const double *lf, int nlines, int line, int j, const VecRef<WT> &V)
    {
        fac = reinterpret_cast<const char *>(f);
        lfac = reinterpret_cast<const char *>(lf);
        frow = (size_t)nlines * 15 * sizeof(FT);
        lrow = (size_t)nlines * 8 * sizeof(double);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (j + r) & 3;
            const int idx = j >= m ? j * (j + 1) / 2 + m : m * (m + 1) / 2 + j;
            ft[r] = (unsigned)((line * 15 + idx) * sizeof(FT));
        }
        ft[4] = (unsigned)((line * 15 + 10 + j) * sizeof(FT));
        ft[5] = (unsigned)((line * 15 + 14) * sizeof(FT));
        la = (unsigned)((line * 8 + (j == 0 ? 3 : j - 1)) * sizeof(double));
        ld = (unsigned)((line * 8 + 4 + max(j, 1) - 1) * sizeof(double));
        l04 = (unsigned)((line * 8 + 3) * sizeof(double));
        l4 = (unsigned)((line * 8 + 7) * sizeof(double));
        // a mirrored block (HALF 1) keeps entries 1..4 in record k-1, entry 0 in record k: the
        // uniform part uses k-1, lane 0 adds one row
        vb = reinterpret_cast<char *>(V.base);
        vb4 = reinterpret_cast<char *>(V.base4);
        vrow = (size_t)V.stride * V.width * sizeof(WT);
        vrow4 = (size_t)V.stride4 * 5 * sizeof(WT);
        vj = (unsigned)(((line - V.line0) * V.width + j) * sizeof(WT)) + ((HALF && j == 0) ? (unsigned)vrow : 0u);
        v4 = (unsigned)(((line - V.line04) * 5 + 4) * sizeof(WT));
        if (SPLIT) {
            gvb = reinterpret_cast<char *>(V.gbase);
            gvrow = (size_t)V.gstride * 5 * sizeof(WT);
            gvj = (unsigned)((line * 5 + j) * sizeof(WT));
            klo = V.klo; khi = V.khi;
            radd = (HALF && j == 0) ? 1 : 0;
            ldum = reinterpret_cast<const WT *>(vb + (size_t)V.klo * vrow + (unsigned)(((line - V.line0) * V.width + j) * sizeof(WT)));
            gdum = reinterpret_cast<const WT *>(gvb + gvj);
        }
    }
    // SPLIT: is this lane's slot of block k (record row k, or k-1 (+1 for lane 0) in a mirrored half) in LDS?
    __device__ __forceinline__ bool in_lds(int k) const
    {
        const int row = (HALF ? k - 1 : k) + radd;
        return row >= klo && row < khi;
    }
    __device__ __forceinline__ WT *gpvj(int k) const
    {
        return reinterpret_cast<WT *>(gvb + (size_t)((HALF ? k - 1 : k) + radd) * gvrow + gvj);
    }
    __device__ __forceinline__ WT *pvj(int k) const
    {
        return reinterpret_cast<WT *>(vb + (size_t)(HALF ? k - 1 : k) * vrow + vj);
    }
    __device__ __forceinline__ WT *pv4(int k) const
    {
        return reinterpret_cast<WT *>(vb4 + (size_t)(HALF ? k - 1 : k) * vrow4 + v4);
    }
};

// The loops below are branch-free inside: loads and stores are unconditional (the halves
// are padded to a multiple of QD blocks with identity blocks, stencil.h), because with
// branches around memory operations the compiler's s_waitcnt insertion falls back to
// vmcnt(0) at the loop head and drains the prefetch ring every iteration. Quads beyond the
// last line walk the last line again but store into a dummy area behind the records.
// One block step of a forward half-chain: (v, v4) = this lane's right-hand-side entries j and 4 of
// the block, q its factor record; updates the carried (wsel, w4p), returns w_j and w_4.
template <class T, class Q>
__device__ __forceinline__ void quad_forward_step(const Q &q, const T v, const T v4, const double nz,
                                                  const double is0, T &wsel, T &w4p, T &wn, T &w4)
{
    const T t0 = emg::widen(q.t[0]), t1 = emg::widen(q.t[1]), t2 = emg::widen(q.t[2]), t3 = emg::widen(q.t[3]),
            t4 = emg::widen(q.t[4]), t44 = emg::widen(q.t44);
    // c_j = rhs_j - (B w_prev)_j ; row 0: the row sum ; row j: B(j,j) w_j
    const T rowsum = quad_sum(xop::mul(q.bA, wsel));
    const T cj = emg::nmad(xop::mul(q.bD, nz), wsel, emg::nmad(is0, rowsum, v));
    const T c4 = emg::nmad(q.d4, w4p, v4);
    const T c1 = quad_rot<1>(cj), c2 = quad_rot<2>(cj), c3 = quad_rot<3>(cj);
    // w_j = sum_m T(j,m) c_m ; w_4 from the partial products T(j,4) c_j
    // two accumulators, four fused multiply-adds per complex product (cplx.h: mad)
    wn = xop::add(emg::mad(t4, c4, emg::mad(t1, c1, xop::mul(t0, cj))), emg::mad(t3, c3, xop::mul(t2, c2)));
    w4 = emg::mad(t44, c4, quad_sum(xop::mul(t4, cj)));
    wsel = emg::mad(nz, wn, xop::mul(is0, w4));
    w4p = w4;
}

template <class T, int HALF, int QD, bool SPLIT = false, class FT = T>
__device__ __forceinline__ void quad_forward(int n0, int n0p, int nlines, int qline, int qend, int j, const FT *fac,
                                             const double *lfac, const VecRef<T> V, T *dummy, T *dummy4)
{
    // dummy / dummy4: store targets of surplus quads, in the address spaces of V.base / V.base4
    // quads with qline >= qend (the end of the caller's line range) are surplus: they walk
    // line qend-1 again and store into the dummy slots
    const HalfWalk<HALF> W(n0, n0p);
    const bool active = qline < qend;
    const int line = min(qline, qend - 1);
    T *const dslot = dummy + ((threadIdx.x & 63) >> 2) * 5;
    T *const dslot4 = dummy4 + ((threadIdx.x & 63) >> 2) * 5;
    QuadRow<T, FT> ring[QD];
    const LaneAddr<T, HALF, SPLIT, FT, T> LA(fac, lfac, nlines, line, j, V);
    auto fetch = [&](QuadRow<T, FT> &q, int i) { q.load(LA, W.fwd(W.clampi(i))); };
#pragma unroll
    for (int d = 0; d < QD; ++d) fetch(ring[d], d);
    // Of w_{k-1} the coupling needs: lane j >= 1 its own entry (B(j,j) w_j), everybody w_4,
    // and lane 0 the row sum  sum_m B(0,m) w_m, m = 1..4 -- formed as a quad sum of one
    // product per lane (lane j >= 1: B(0,j) w_j, lane 0: B(0,4) w_4) instead of broadcasting w.
    T wsel = emg::zero<T>();                     // lane 0: w_4, lane j >= 1: own entry w_j
    T w4p = emg::zero<T>();                      // w_4 of the previous block
    const double nz = j != 0 ? 1.0 : 0.0, is0 = 1.0 - nz;
    for (int i0 = 0; i0 < W.steps; i0 += QD) {
#pragma unroll
        for (int d = 0; d < QD; ++d) {
            const int k = W.fwd(i0 + d);
            const QuadRow<T, FT> &q = ring[d];
            T wn, w4;
            quad_forward_step(q, q.v, q.v4, nz, is0, wsel, w4p, wn, w4);
            T *const o4 = active ? LA.pv4(k) : dslot4 + 4;
            if constexpr (SPLIT) {
                // (dummy: LDS, dummy4: global -- one store into either space, the idle one to its dummy slot)
                const bool in = LA.in_lds(k);
                T *const ol = (active && in) ? LA.pvj(k) : dslot + j;
                T *const og = (active && !in) ? LA.gpvj(k) : dslot4 + j;
                *ol = wn;
                *og = wn;
            } else {
                T *const oj = active ? LA.pvj(k) : dslot + j;
                *oj = wn;
            }
            *o4 = w4;
            fetch(ring[d], i0 + d + QD);
        }
    }
}

template <class T, int QD>
__global__ __launch_bounds__(64) void k_line_forward(int n0, int n0p, int nlines, const T *fac, const double *lfac,
                                                     T *vec, T *dummy)
{
    const int gt = blockIdx.x * 64 + threadIdx.x;
    const VecRef<T> V = VecRef<T>::global(vec, nlines);
    if (blockIdx.y == 0) quad_forward<T, 0, QD>(n0, n0p, nlines, gt >> 2, nlines, gt & 3, fac, lfac, V, dummy, dummy);
    else quad_forward<T, 1, QD>(n0, n0p, nlines, gt >> 2, nlines, gt & 3, fac, lfac, V, dummy, dummy);
}

// Middle block of the two-sided solve (stencil.h: line_middle), by both half-waves:
// x_Q = T_Q (r_Q - [B_m w_{m-1}] - [U_{m+1} w_{m+2}]). Every lane forms the 6-vector z (cheap,
// real x complex), lane j the rows j and j2 = 4 + (j & 1) of T_Q z (rows 4 / 5 are computed
// twice); xa = x_Q[j], xb = x_Q[j2].
template <class T, class FT = T, class WT = T>
__device__ __forceinline__ void quad_middle(int n0, int n0p, int nlines, int line, int j, const FT *fac,
                                            const double *lfac, const VecRef<WT> V, T &xa, T &xb)
{
    const int mk = emg::line_mid(n0);
    const size_t rm = (size_t)mk * nlines + line, rp = rm + nlines;

    // rows of the packed symmetric T_Q (15 entries in record m, 6 in record m+1), loaded
    // first: they do not depend on the forward pass
    const int j2 = 4 + (j & 1);
    T ta[6], tb[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int ia = j >= m ? j * (j + 1) / 2 + m : m * (m + 1) / 2 + j;
        const int ib = j2 >= m ? j2 * (j2 + 1) / 2 + m : m * (m + 1) / 2 + j2;
        ta[m] = emg::widen(ia < 15 ? fac[rm * 15 + ia] : fac[rp * 15 + (ia - 15)]);
        tb[m] = emg::widen(ib < 15 ? fac[rm * 15 + ib] : fac[rp * 15 + (ib - 15)]);
    }
    T z[6];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = emg::widen(*V.p(mk, line, r));
    z[4] = emg::widen(*V.p4(mk, line));
    z[5] = emg::widen(*V.p(mk + 1, line, 0));
    {   // top coupling B_m w_{m-1} (zero if there is no top half: B_0 is stored as zeros)
        const int kt = mk > 0 ? mk - 1 : mk;
        const double *lf = lfac + rm * 8;
        T q0 = emg::zero<T>();
#pragma unroll
        for (int m = 1; m < 5; ++m) {
            const T y = emg::widen(m < 4 ? *V.p(kt, line, m) : *V.p4(kt, line));
            q0 = emg::mad(lf[m - 1], y, q0);
            z[m] = emg::nmad(lf[3 + m], y, z[m]);
        }
        z[0] = xop::sub(z[0], q0);
    }
    {   // bottom coupling U_{m+1} w_{m+2}; w_{m+2} = slots (m+2, 0), (m+1, 1..4); U of an
        // identity padding block is zero, so no guard is needed
        const double *lf = lfac + rp * 8;
        T q0 = emg::zero<T>();
#pragma unroll
        for (int m = 1; m < 5; ++m) {
            const T y = emg::widen(m < 4 ? *V.p(mk + 1, line, m) : *V.p4(mk + 1, line));
            q0 = emg::mad(lf[m - 1], y, q0);
            z[m] = emg::nmad(lf[3 + m], y, z[m]);
        }
        z[5] = xop::sub(z[5], q0);
    }
    xa = xop::add(emg::mad(ta[4], z[4], emg::mad(ta[2], z[2], xop::mul(ta[0], z[0]))),
                 emg::mad(ta[5], z[5], emg::mad(ta[3], z[3], xop::mul(ta[1], z[1]))));
    xb = xop::add(emg::mad(tb[4], z[4], emg::mad(tb[2], z[2], xop::mul(tb[0], z[0]))),
                 emg::mad(tb[5], z[5], emg::mad(tb[3], z[3], xop::mul(tb[1], z[1]))));
}

// Backward substitution of one half, outwards from the middle block, fused with the scatter
// into the field (core.py:775-783): lane j writes entry j of block k straight to its edge,
// every lane writes entry 4. Entries that do not exist (padding blocks, surplus quads) go to
// the dummy area through an address select -- no predicate, no branch. The top half also
// writes the middle block.
// MIDFIRST: the middle block is solved BEFORE the register ring is filled (its ~70 registers
// and the ring's 160 are then not live together: the batched kernel must stay under 256
// registers so that two workgroups share a CU); otherwise the ring fetch is in flight while
// the middle block is solved.
template <class T, int DIR, int HALF, int QD, bool MIDFIRST = false, bool SPLIT = false, class FT = T>
__device__ __forceinline__ void quad_backward(const emg::Level<T> &L, int colour, int cntp, int cntq, int n0p,
                                              int qline, int qend, int j, const FT *fac, const double *lfac,
                                              const VecRef<T> V, T *dummy, size_t boff = 0)
{
    const emg::Axes<T, DIR> A(L, boff);
    const int n0 = A.n0();
    const HalfWalk<HALF> W(n0, n0p);
    const int mk = W.mk;
    const int nlines = cntp * cntq;
    const bool active = qline < qend;
    const int line = min(qline, qend - 1);
    int i1, i2, lid;
    emg::line_of_thread<DIR>(colour, cntp, cntq, line % cntp, line / cntp, i1, i2, lid);
    // entry j lives on component cj at (k + dk, i1 - d1, i2 - d2); entry 4 on component 2.
    // standard blocks carry the transverse edges of node k+1, mirrored blocks those of node k
    const int cj = j == 0 ? 0 : (j <= 2 ? 1 : 2), dk = (j == 0 || HALF) ? 0 : 1, dk4 = HALF ? 0 : 1;
    const int d1 = j == 1 ? 1 : 0, d2 = j == 3 ? 1 : 0;
    T *const ej = A.E(cj) + A.idx(cj, dk, i1 - d1, i2 - d2);
    const long sj = (long)A.idx(cj, dk + 1, i1 - d1, i2 - d2) - (long)A.idx(cj, dk, i1 - d1, i2 - d2);
    T *const e4 = A.E(2) + A.idx(2, dk4, i1, i2);
    const long s4 = (long)A.idx(2, dk4 + 1, i1, i2) - (long)A.idx(2, dk4, i1, i2);
    T *const dslot = dummy + ((threadIdx.x & 63) >> 2) * 5;
    T *const dj = dslot + j, *const d4 = dslot + 4;

    QuadRow<T, FT> ring[QD];
    const LaneAddr<T, HALF, SPLIT, FT, T> LA(fac, lfac, nlines, line, j, V);
    auto fetch = [&](QuadRow<T, FT> &q, int i) {
        q.load(LA, min(max(W.bwd(W.clampi(i)), HALF), n0p - 1));   // a half without blocks still prefetches
    };
    if (!MIDFIRST) {
#pragma unroll
        for (int d = 0; d < QD; ++d) fetch(ring[d], d);
    }
    // coupling to the middle: B_m (top) / U_{m+1} (bottom)
    QuadRow<T> qm;
    qm.load_b(lfac, (size_t)(HALF ? mk + 1 : mk) * nlines + line, j);

    // x_Q: this lane's entry j (xa) and entry 4 (even lanes) / 5 (odd lanes) (xb)
    T xa, xb;
    quad_middle<T, FT, T>(n0, n0p, nlines, line, j, fac, lfac, V, xa, xb);
    if (MIDFIRST) {
        asm volatile("" ::: "memory");       // keep the ring fetch behind the middle block
#pragma unroll
        for (int d = 0; d < QD; ++d) fetch(ring[d], d);
    }
    const T xq0 = quad_bcast<0>(xa), xq4 = quad_bcast<0>(xb), xq5 = quad_bcast<1>(xb);
    if (HALF == 0) {
        // lane j writes entry j of x_Q (E0(m), t(m+1)_1..3), lane 0 also entry 4, lane 1
        // entry 5 (E0(m+1)); t(m+1) are the transverse entries of "standard block m"
        const int dkm = j == 0 ? 0 : 1;
        T *const pm = A.E(cj) + A.idx(cj, mk + dkm, i1 - d1, i2 - d2);
        *(active ? pm : dj) = xa;
        T *const p4 = A.E(2) + A.idx(2, mk + 1, i1, i2);
        T *const p5 = A.E(0) + A.idx(0, mk + 1, i1, i2);
        *((active && j == 0) ? p4 : ((active && j == 1) ? p5 : d4)) = xb;
    }
    // x of the block next to the walk: standard block m / mirrored block m+1 of x_Q
    T x[5];                  // only x[0], x[4] and the own entry are needed by the coupling
    x[0] = HALF ? xq5 : xq0;
    x[4] = xq4;
    const double own0 = j == 0 ? 1.0 : 0.0;
    T xmine = HALF ? emg::mad(own0, xq5, xop::mul(1.0 - own0, xa)) : xa;   // this lane's own entry of x
    double upA = qm.bA, upD = qm.bD, up04 = qm.b04, up44 = qm.d4;          // entries of the coupling block
    // running scatter pointers (surplus quads: the dummy slots, not advanced): no per-lane
    // 64-bit multiply per step
    T *pej = active ? ej + (long)W.bwd(0) * sj : dj;
    T *pe4 = active ? e4 + (long)W.bwd(0) * s4 : d4;
    const long incj = active ? (HALF ? sj : -sj) : 0, inc4 = active ? (HALF ? s4 : -s4) : 0;
    const double nz = j != 0 ? 1.0 : 0.0;
    for (int i0 = 0; i0 < W.steps; i0 += QD) {
#pragma unroll
        for (int d = 0; d < QD; ++d) {
            const int k = W.bwd(i0 + d);
            const QuadRow<T, FT> &q = ring[d];
            T xn, x4;
            quad_backward_step(q, q.v, q.v4, nz, upA, upD, up04, up44, x[0], x[4], xmine, xn, x4);
            upA = q.bA; upD = q.bD; up04 = q.b04; up44 = q.d4;
            const bool real_block = HALF ? k <= n0 - 1 : true;      // uniform over the wave
            T *const oj = real_block ? pej : dj;
            T *const o4 = real_block ? pe4 : d4;
            *oj = xn;
            *o4 = x4;
            pej += incj;
            pe4 += inc4;
            fetch(ring[d], i0 + d + QD);
        }
    }
}

// One block step of a backward half-chain for one right-hand side: (wj, w4) = this lane's entries j and 4 of
// the block's w record, q its factor record, (upA, upD, up04, up44) the coupling to the block solved before;
// updates the carried x_0 / x_4 / own entry and returns the block's x_j and x_4.
template <class T, class Q>
__device__ __forceinline__ void quad_backward_step(const Q &q, const T wj, const T w4, const double nz,
                                                   const double upA, const double upD, const double up04,
                                                   const double up44, T &x0, T &x4, T &xmine, T &xn, T &xn4)
{
    // h = B^T x_prev: h_0 = 0, h_m = B(0,m) x_0 + B(m,m) x_m
    const T hj = emg::mad(xop::mul(upA, nz), x0, xop::mul(xop::mul(upD, nz), xmine));
    const T h4 = emg::mad(up04, x0, xop::mul(up44, x4));
    const T h1 = quad_rot<1>(hj), h2 = quad_rot<2>(hj), h3 = quad_rot<3>(hj);
    const T t0 = emg::widen(q.t[0]), t1 = emg::widen(q.t[1]), t2 = emg::widen(q.t[2]), t3 = emg::widen(q.t[3]),
            t4 = emg::widen(q.t[4]), t44 = emg::widen(q.t44);
    xn = xop::sub(emg::nmad(t4, h4, emg::nmad(t1, h1, emg::nmad(t0, hj, wj))),
                  emg::mad(t3, h3, xop::mul(t2, h2)));
    xn4 = xop::sub(emg::nmad(t44, h4, w4), quad_sum(xop::mul(t4, hj)));
    x0 = quad_bcast<0>(xn);
    x4 = xn4;
    xmine = xn;
}

template <class T, int DIR, int QD>
__global__ __launch_bounds__(64) void k_line_backward(emg::Level<T> L, int colour, int cntp, int cntq, int n0p,
                                                      const T *fac, const double *lfac, const T *vec, T *dummy)
{
    const int gt = blockIdx.x * 64 + threadIdx.x;
    const VecRef<T> V = VecRef<T>::global(const_cast<T *>(vec), cntp * cntq);
    const int nl = cntp * cntq;
    if (blockIdx.y == 0) quad_backward<T, DIR, 0, QD>(L, colour, cntp, cntq, n0p, gt >> 2, nl, gt & 3, fac, lfac, V, dummy);
    else quad_backward<T, DIR, 1, QD>(L, colour, cntp, cntq, n0p, gt >> 2, nl, gt & 3, fac, lfac, V, dummy);
}

constexpr int LC_THREADS = 256;   // workgroup of k_line_colour: 2 chain waves + helper waves for the rhs phase
// One launch per colour for small levels: a workgroup of two waves owns 16 lines -- wave 0
// their top halves, wave 1 their bottom halves -- and runs rhs assembly, forward and
// backward substitution for them back to back. Lines of one colour class are independent,
// so only the workgroup's own records have to be complete between the phases (workgroup
// barriers; both waves sit on one CU and share its L1). On the coarse levels the three
// separate launches are bound by launch latency, not by work.
// VMODE: where the right-hand-side / solution records of the workgroup's 16 lines live
// (VecRef): 0 global scratch; 1 LDS (16 x n0p x 80 B + dummy slots); 2 slots 0..3 in LDS
// (16 x n0p x 64 B), slot 4 in the global scratch -- for lines too long for mode 1 (128
// blocks: 166 KB). The launcher picks the first mode that fits. In LDS the records never leave
// the CU: no HBM/L2 round trips between the three phases.
// FT: storage type of the T records (T, or emg::compact_of<T> on levels with compact line records: the ring holds
// them as loaded, the steps widen them; the right-hand-side / solution records stay T here -- they live in LDS)
template <class T, int DIR, int VMODE, bool BATCH, int QD = emg::LINE_PAD, class FT = T>
__global__ __launch_bounds__(LC_THREADS, BATCH ? 2 : 1) void k_line_colour(emg::Level<T> L, int colour, int cntp, int cntq, int n0p,
                                                            int lpw, const FT *fac, const double *lfac, T *vec,
                                                            T *dummy, size_t vstride)
{
    // BATCH: grid.y = right-hand side (Level::batch): same factors, own field / source / scratch
    // (a separate instantiation: the single-source kernel keeps its register count)
    size_t boff = 0;
    if (BATCH) {
        boff = blockIdx.y * L.bstride;
        vec += blockIdx.y * vstride;
        dummy += blockIdx.y * vstride;
    }
    // lpw = lines per workgroup (16, 8 or 4): the chain phases cost the same however full
    // the two waves are, but the right-hand-side phase is spread over more CUs when the
    // colour class has fewer than 16 x 256 lines
    extern __shared__ double2 lc_smem[];
    const emg::Axes<T, DIR> A(L, boff);
    const int nlines = cntp * cntq;
    const int line0 = blockIdx.x * lpw;
    const int nl = min(lpw, nlines - line0);
    T *const lvec = reinterpret_cast<T *>(lc_smem);
    VecRef<T> V;
    T *dum, *dum4;           // dummy store targets in the address spaces of V.base / V.base4
    if (VMODE == 1) {
        V = VecRef<T>{lvec, lpw, line0, 5, lvec, lpw, line0};
        dum = dum4 = lvec + (size_t)lpw * n0p * 5;
    } else if (VMODE == 2) {
        V = VecRef<T>{lvec, lpw, line0, 4, vec, nlines, 0};
        dum = lvec + (size_t)lpw * n0p * 4;
        dum4 = dummy;
    } else if (VMODE == 3) {
        // lines too long for mode 2: the rows around the middle block in LDS, the outer rows in the
        // global scratch (launch.h: line_split_rows)
        const emg::LineSplit sp = emg::line_split_rows(A.n0(), n0p, lpw, sizeof(T));
        V = VecRef<T>{lvec - (size_t)sp.klo * lpw * 4, lpw, line0, 4, vec, nlines, 0, vec, nlines, sp.klo, sp.khi};
        dum = lvec + (size_t)lpw * (sp.khi - sp.klo) * 4;
        dum4 = dummy;
    } else {
        // (vstride == ~0 in a single-source launch: the debug aliasing of option line_debug)
        V = VecRef<T>::global(vec, (!BATCH && vstride == ~(size_t)0) ? 0 : nlines);
        dum = dum4 = dummy;
    }
    // (1) right-hand sides of the workgroup's lines; x-lines run the lanes along the line. Only the n0 real blocks
    // are computed: with the identity padding rows (n0p - n0 = 2 ... 5 of them, rhs = 0) among the items, 16 lines of
    // 64 blocks were 1 056 items = five rounds of 256 threads instead of four, 8 lines of 32 blocks two instead of one
    // -- and a round is 3 000 (L2) to 8 500 (HBM) cycles of the launch (profiles/r04_line_phase_stamps.txt)
    const int n0r = A.n0();
    for (int i = threadIdx.x; i < nl * n0r; i += LC_THREADS) {
        const int ll = DIR == 0 ? i / n0r : i % nl;
        const int k = DIR == 0 ? i % n0r : i / nl;
        const int lid = line0 + ll;
        int i1, i2, l2;
        emg::line_of_thread<DIR>(colour, cntp, cntq, lid % cntp, lid / cntp, i1, i2, l2);
        T rhs[5];
        emg::line_rhs<T, DIR>(A, k, i1, i2, rhs);
        if (VMODE == 3 && (k < V.klo || k >= V.khi)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *V.pg(k, lid, r) = rhs[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) *V.p(k, lid, r) = rhs[r];
        }
        *V.p4(k, lid) = rhs[4];
    }
    // identity padding blocks behind block n0 - 1: rhs = 0 (dealt from the last thread downwards: the helper waves)
    for (int i = LC_THREADS - 1 - (int)threadIdx.x; i < nl * (n0p - n0r); i += LC_THREADS) {
        const int k = n0r + i / nl, lid = line0 + i % nl;
        if (VMODE == 3 && (k < V.klo || k >= V.khi)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *V.pg(k, lid, r) = emg::zero<T>();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) *V.p(k, lid, r) = emg::zero<T>();
        }
        *V.p4(k, lid) = emg::zero<T>();
    }
    __syncthreads();
    // chain waves: waves 0 / 1 walk the top / bottom halves of the workgroup's first 16 lines; with
    // 32 lines per workgroup waves 2 / 3 do the same for lines 16..31, otherwise they only helped
    // with the right-hand sides and are done
    const int wave = threadIdx.x >> 6;
    if (wave >= 2 && lpw < 32) return;
    const int half = wave & 1;
    const int qline = line0 + (wave >> 1) * 16 + ((threadIdx.x & 63) >> 2), j = threadIdx.x & 3;
    const int qend = line0 + nl;
    constexpr bool SPLIT = VMODE == 3;
    if (half == 0) quad_forward<T, 0, QD, SPLIT, FT>(A.n0(), n0p, nlines, qline, qend, j, fac, lfac, V, dum, dum4);
    else quad_forward<T, 1, QD, SPLIT, FT>(A.n0(), n0p, nlines, qline, qend, j, fac, lfac, V, dum, dum4);
    __syncthreads();
    // the backward pass stores into the FIELD; its dummy slots must be global memory too, or
    // the address select mixes address spaces and the stores become flat instructions
    if (half == 0) quad_backward<T, DIR, 0, QD, BATCH, SPLIT, FT>(L, colour, cntp, cntq, n0p, qline, qend, j, fac, lfac, V, dummy, boff);
    else quad_backward<T, DIR, 1, QD, BATCH, SPLIT, FT>(L, colour, cntp, cntq, n0p, qline, qend, j, fac, lfac, V, dummy, boff);
}

// ---- the colour pass of SMALL levels with short dependent chains: k_line_wide ----------------------------
// (stencil.h, "wide" form.) On the levels where a colour class has fewer lines than the chip has SIMDs a launch of
// k_line_colour costs what one wave needs to issue its chain: ~130 instructions per block step at one instruction
// per ~5 cycles, 64 dependent steps for a 64-block line, and as much again for the right-hand sides, the middle
// block and the set-up of the walks on the shortest lines (profiles/r04_line_phase_stamps.txt). Here the
// recurrences run in four unknowns with the model-only matrices N_k = (T_k C_k)[1..4, 1..4] (k_line_wide_setup:
// one more record of 16 entries per block, behind the T records of the direction):
//   * SIXTEEN lanes per half-line, lane (a, b) holding N_k(a, b): a step is one complex multiply-add per lane and a
//     two-stage sum -- inside the quads (DPP quad_perm) in even steps, across the quads (DPP row_ror) in odd steps
//     with the matrix fetched transposed, so that the result lies where the next step wants it: ~25 instructions;
//   * everything that is not a recurrence -- right-hand sides, g = (T r)[1..4], c = r - C w, w_0, g' = C^T w,
//     x = T (c - h), the scatter -- by ONE THREAD PER BLOCK (waves 0..2) which keeps r / c, T_k and C_k in registers
//     from the first phase to the last; the middle blocks (6 x 6) by the threads of wave 3;
//   * the 4-vectors pass between the block threads and the chain lanes through LDS rows of six entries (four values,
//     a zero that the lanes without a right-hand-side entry add, a dummy that they store to): no select, no branch
//     inside the chains.
// Five phases, four workgroup barriers (LDS only). Same factors T_k as the other line kernels; N_k is one more
// rounding of T_k C_k, so results agree with them to rounding, not bit for bit -- a level runs ONE of the kernels
// for every right-hand side (single source and batch alike), whatever the batch size.
constexpr int LW_THREADS = 256;
constexpr int LW_BLOCK_THREADS = 192;      // waves 0..2: one thread per top / bottom block; wave 3: the middle blocks
constexpr int LW_ROW = 6;                  // entries of an LDS row: values 1..4, a zero, a dummy
// lines per workgroup: at most 8 (two chain waves per half), and every block of them needs a thread
inline int wide_lpw(int n0, int nbthr = LW_BLOCK_THREADS) { return std::max(1, std::min(8, nbthr / std::max(n0 - 2, 1))); }
// an invariant of the k_line_wide launch (launch_line_colour): a line's blocks fit the block threads even with one line
// per workgroup -- it holds because line_wide_capable stops at WIDE_N0_MAX (the dynamic LDS, ~55 KB at most, is checked
// against the opt-in limit at the launch)
static_assert(emg::WIDE_N0_MAX - 2 <= LW_BLOCK_THREADS, "k_line_wide: one thread per block of a line");

template <class T> __global__ __launch_bounds__(256) void k_line_wide_setup(const T *fac, const double *lfac, T *nfac, size_t nrec)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrec) return;
    T Tk[15], N[16];
    double lf[8];
#pragma unroll
    for (int j = 0; j < 15; ++j) Tk[j] = fac[r * 15 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) lf[j] = lfac[r * 8 + j];
    emg::wide_n_record<T>(Tk, lf, N);
#pragma unroll
    for (int j = 0; j < 16; ++j) nfac[r * 16 + j] = N[j];
}

// phase stamps of workgroup 0 (threads 0 and 192), only in the -DEMG_WIDE_STAMPS build of tools/wide_stamps.py
#ifdef EMG_WIDE_STAMPS
__device__ unsigned long long g_wide_stamps[32];
#define WSTAMP(i)                                                                                       \
    do {                                                                                                \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == nbthr))                   \
            g_wide_stamps[(threadIdx.x ? 16 : 0) + (i)] = __builtin_amdgcn_s_memtime();                 \
    } while (0)
#else
#define WSTAMP(i)
#endif

// One chain of a 16-lane group: nst steps v <- acc_i - M_i v, M = N (forward) or N^T (BWD), acc_i from / result to
// the LDS row krow0 + i dk, N from record kmat0 + i dk. v enters (and leaves an even number of steps) with lane
// (a, b) holding entry b.
template <class T, bool BWD>
__device__ __forceinline__ void wide_chain(T *rows, const T *nbase, size_t nrow, unsigned loff, int krow0, int kmat0,
                                           int dk, int nst, int l16, T v)
{
    const int a = l16 >> 2, b = l16 & 3;
    const unsigned e1 = 4 * a + b, e2 = 4 * b + a;
    const unsigned eI = loff + (BWD ? e2 : e1), eII = loff + (BWD ? e1 : e2);
    const int rdI = b == 0 ? a : 4, wrI = b == 0 ? a : 5;
    const int rdII = a == 0 ? b : 4, wrII = a == 0 ? b : 5;
    T nr[4], ac[4];
    auto fetch = [&](int d, int i) {
        const int ic = min(i, nst - 1);
        nr[d] = nbase[(size_t)(kmat0 + ic * dk) * nrow + ((d & 1) ? eII : eI)];
        ac[d] = rows[(krow0 + ic * dk) * LW_ROW + ((d & 1) ? rdII : rdI)];
    };
#pragma unroll
    for (int d = 0; d < 4; ++d) fetch(d, d);
    auto step = [&](int d, int i) {
        T p = emg::nmad(nr[d], v, ac[d]);
        if (!(d & 1)) {
            p = xop::add(p, dpp_move<0x4E>(p));    // quad_perm:[2,3,0,1]
            p = xop::add(p, dpp_move<0xB1>(p));    // quad_perm:[1,0,3,2]
        } else {
            p = xop::add(p, dpp_move<0x128>(p));   // row_ror:8
            p = xop::add(p, dpp_move<0x124>(p));   // row_ror:4
        }
        rows[(krow0 + i * dk) * LW_ROW + ((d & 1) ? wrII : wrI)] = p;
        v = p;
    };
    int i = 0;
    for (; i + 4 <= nst; i += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            step(d, i + d);
            fetch(d, i + d + 4);
        }
    }
    if (i < nst) {
        step(0, i);
        if (i + 1 < nst) {
            step(1, i + 1);
            if (i + 2 < nst) step(2, i + 2);
        }
    }
}

// (LDS written by some lanes of a wave, read by others of the same wave: the hardware runs a wave's LDS operations in
//  order; this keeps the compiler from moving them across)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// row r of T_Q z (two accumulators, summed as wide_middle sums them)
template <class T> __device__ __forceinline__ T wide_row6(const T (&row)[6], const T *z)
{
    const T lo = emg::mad(row[4], z[4], emg::mad(row[2], z[2], emg::mad(row[0], z[0], emg::zero<T>())));
    const T hi = emg::mad(row[5], z[5], emg::mad(row[3], z[3], emg::mad(row[1], z[1], emg::zero<T>())));
    return lo + hi;
}

// The middle blocks (6 x 6) have EIGHT LANES per line (wave 3: lane r = 0..5 owns entry / row r): one thread walking
// wide_middle -- six z entries, six rows of T_Q z, eight h entries: ~2 700 ticks -- was what the workgroup waited for at
// the barrier behind phase C (the block threads need ~1 200, profiles/r05_small_level_experiments.txt). The operations
// and their order are wide_middle's, entry by entry: the same bits.
template <class T, int DIR, bool BATCH>
__global__ __launch_bounds__(LW_THREADS + 64) void k_line_wide(emg::Level<T> L, int colour, int cntp, int cntq, int lpw,
                                                              const T *fac, const double *lfac, const T *nfac, int nbthr)
{
    extern __shared__ double2 lw_smem[];
    WSTAMP(0);
    const size_t boff = BATCH ? blockIdx.y * L.bstride : 0;
    const emg::Axes<T, DIR> A(L, boff);
    const int n0 = A.n0(), mk = emg::line_mid(n0);
    const int nbt = mk, nbb = n0 - mk - 2, nblk = n0 - 2;
    const int nlines = cntp * cntq, line0 = blockIdx.x * lpw, nl = min(lpw, nlines - line0);
    const int rows = n0 + 1;                               // (row n0 of a line: the dummy row)
    T *const GY = reinterpret_cast<T *>(lw_smem);          // [lpw][rows][LW_ROW]: g, then y
    T *const GH = GY + (size_t)lpw * rows * LW_ROW;        //                      g', then h
    T *const RQ = GH + (size_t)lpw * rows * LW_ROW;        // [lpw][2][LW_ROW]: the middle blocks' r_Q -> z, and x_Q
    const int t = threadIdx.x;
    const bool isq = t >= nbthr;                           // (nbthr block threads: 192, or 256 with a fifth wave for the middle blocks)
    const int tq = t - nbthr;
    const int rq = tq & 7;                                 // middle blocks: lane rq of the line's eight
    const bool has = isq ? (tq >> 3) < nl && rq < 6 : t < nl * nblk;
    int ll = 0, j = 0;
    if (isq) ll = tq >> 3;
    else if (has) {
        if (DIR == 0) { ll = emg::fast_div(t, nblk); j = t - ll * nblk; }      // x-lines: the field is contiguous along the line
        else { j = emg::fast_div(t, nl); ll = t - j * nl; }
    }
    int i1 = 0, i2 = 0, k = 0, mir = 0;
    T Tk[15], r[5];                                         // block thread: T_k, r -> c
    double lf[8];                                           // C_k
    T qrow[6];                                              // middle lane: row rq of T_Q
    double cq[4] = {0.0, 0.0, 0.0, 0.0};                    //   its entries of B_m / U_{m+1}: lane 0 / 5 the first row, 1..4 {B(0,b), B(b,b), U(0,b), U(b,b)}
    T *const rqv = RQ + (size_t)ll * 2 * LW_ROW;
    // ---- (A) right-hand sides, factor records, g --------------------------------------------------
    if (has) {
        const int lid = line0 + ll;
        int l2;
        const int tq_ = emg::fast_div(lid, cntp);
        emg::line_of_thread<DIR>(colour, cntp, cntq, lid - tq_ * cntp, tq_, i1, i2, l2);
        if (!isq) {
            const emg::WideBlock wb = emg::wide_block(j, mk);
            k = wb.k; mir = wb.mir;
            T rb[5];
            emg::wide_block_rhs<T, DIR>(A, k, mir, i1, i2, rb);
            // (the factor records behind the right-hand sides in program order: fetched first, the phase got slower --
            //  5.7 -> 6.25 us per launch on 4- and 8-block lines)
            const size_t rec = (size_t)k * nlines + lid;
#pragma unroll
            for (int q = 0; q < 15; ++q) Tk[q] = fac[rec * 15 + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) lf[q] = lfac[rec * 8 + q];
            T g[4];
            emg::wide_g<T, 15>(Tk, rb, g);
            T *const row = GY + ((size_t)ll * rows + k) * LW_ROW;
#pragma unroll
            for (int q = 0; q < 4; ++q) row[q] = g[q];
            row[4] = emg::zero<T>();
            GH[((size_t)ll * rows + k) * LW_ROW + 4] = emg::zero<T>();
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = rb[q];
        } else {
            // every lane its row of T_Q (21 packed entries over the records m and m + 1) and its coupling entries ...
            const size_t rm0 = (size_t)mk * nlines + lid, rm1 = rm0 + nlines;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int q = emg::sym(rq, b);
                qrow[b] = q < 15 ? fac[rm0 * 15 + q] : fac[rm1 * 15 + q - 15];
            }
            if (rq == 0 || rq == 5) {
#pragma unroll
                for (int b = 0; b < 4; ++b) cq[b] = lfac[(rq == 0 ? rm0 : rm1) * 8 + b];
            } else {
                cq[0] = lfac[rm0 * 8 + rq - 1]; cq[1] = lfac[rm0 * 8 + 4 + rq - 1];
                cq[2] = lfac[rm1 * 8 + rq - 1]; cq[3] = lfac[rm1 * 8 + 4 + rq - 1];
            }
            // ... and lane 0 the six right-hand-side entries of the block, for all of them
            if (rq == 0) {
                T rm[5];
                emg::line_rhs<T, DIR>(A, mk, i1, i2, rm);
#pragma unroll
                for (int q = 0; q < 5; ++q) rqv[q] = rm[q];
                rqv[5] = emg::line_rhs_e0<T, DIR>(A, mk + 1, i1, i2);
            }
        }
    }
    WSTAMP(1);
    __syncthreads();
    WSTAMP(2);
    // the chain groups: waves 0 / 1 the top halves of lines 0..3 / 4..7, waves 2 / 3 their bottom halves -- a wave's
    // groups all walk the same number of steps; groups beyond the last line repeat it (identical stores)
    const int wave = t >> 6, lane = t & 63, half = wave >> 1;
    const int cl = min((wave & 1) * 4 + (lane >> 4), nl - 1);
    const bool chain_wave = wave < 4 && (wave & 1) * 4 < nl;
    const T *const nbase = nfac + (size_t)line0 * 16;
    const size_t nrow = (size_t)nlines * 16;
    // ---- (F) forward chains: y_k = g_k - N_k y_kn ---------------------------------------------------
    {
        const int nst = half ? nbb : nbt;
        if (chain_wave && nst > 0)
            wide_chain<T, false>(GY + (size_t)cl * rows * LW_ROW, nbase, nrow, (unsigned)cl * 16, half ? n0 - 1 : 0,
                                 half ? n0 - 1 : 0, half ? -1 : 1, nst, lane & 15, emg::zero<T>());
    }
    WSTAMP(3);
    __syncthreads();
    WSTAMP(4);
    // ---- (C) per block: c = r - C w_kn, w_0, g' = C^T w; the middle blocks -----------------------------
    if (has) {
        if (!isq) {
            const int kn = mir ? k + 1 : k - 1;
            const bool first = mir ? k == n0 - 1 : k == 0;
            const T *const yr = GY + ((size_t)ll * rows + (first ? k : kn)) * LW_ROW;
            const T *const yo = GY + ((size_t)ll * rows + k) * LW_ROW;
            T yp[4], y[4], c[5], rb[5], gp[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { yp[q] = yr[q]; y[q] = yo[q]; }
#pragma unroll
            for (int q = 0; q < 5; ++q) rb[q] = r[q];
            emg::wide_c<T>(lf, rb, yp, c);
            const T w0 = emg::wide_row5<T, 15>(Tk, 0, c);
            emg::wide_gp<T>(lf, w0, y, gp);
            T *const o = GH + ((size_t)ll * rows + (first ? n0 : kn)) * LW_ROW;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = gp[q];
#pragma unroll
            for (int q = 0; q < 5; ++q) r[q] = c[q];
        } else {
            // z entry by entry (wide_middle): z_0 = r_0 - B(0,.) y_T, z_5 = r_5 - U(0,.) y_B, z_b = r_b - B(b,b) y_T,b - U(b,b) y_B,b
            const T *const yt = GY + ((size_t)ll * rows + max(mk - 1, 0)) * LW_ROW;
            const T *const yb = GY + ((size_t)ll * rows + min(mk + 2, n0 - 1)) * LW_ROW;
            T z = rqv[rq];
            if (rq == 0 || rq == 5) {
                const T *const y = rq == 0 ? yt : yb;
                const bool any = rq == 0 ? nbt > 0 : nbb > 0;
                const T y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
                T q = cq[0] * (any ? y0 : emg::zero<T>());
                q = emg::mad(cq[1], any ? y1 : emg::zero<T>(), q);
                q = emg::mad(cq[2], any ? y2 : emg::zero<T>(), q);
                q = emg::mad(cq[3], any ? y3 : emg::zero<T>(), q);
                z = z - q;
            } else {
                const T vt = yt[rq - 1], vb = yb[rq - 1];
                const T yT = nbt > 0 ? vt : emg::zero<T>(), yB = nbb > 0 ? vb : emg::zero<T>();
                z = emg::nmad(cq[3], yB, emg::nmad(cq[1], yT, z));
            }
            rqv[LW_ROW + rq] = z;
        }
    }
    wave_sync();
    T xq = emg::zero<T>();
    if (has && isq) {
        xq = wide_row6<T>(qrow, rqv + LW_ROW);             // x_Q row by row
        rqv[rq] = xq;
    }
    wave_sync();
    if (has && isq) {
        if (rq >= 1 && rq <= 4) {                          // h of the two neighbouring blocks, entry by entry
            const T x0 = rqv[0], x5 = rqv[5];
            GH[((size_t)ll * rows + (nbt > 0 ? mk - 1 : n0)) * LW_ROW + rq - 1] = emg::mad(cq[0], x0, cq[1] * xq);
            GH[((size_t)ll * rows + (nbb > 0 ? mk + 2 : n0)) * LW_ROW + rq - 1] = emg::mad(cq[2], x5, cq[3] * xq);
        }
        // the middle block's solution: entries 0..4 of block m (wide_block_scatter), and E0(m + 1)
        if (rq == 0) A.E(0)[A.idx(0, mk, i1, i2)] = xq;
        else if (rq == 1) A.E(1)[A.idx(1, mk + 1, i1 - 1, i2)] = xq;
        else if (rq == 2) A.E(1)[A.idx(1, mk + 1, i1, i2)] = xq;
        else if (rq == 3) A.E(2)[A.idx(2, mk + 1, i1, i2 - 1)] = xq;
        else if (rq == 4) A.E(2)[A.idx(2, mk + 1, i1, i2)] = xq;
        else A.E(0)[A.idx(0, mk + 1, i1, i2)] = xq;
    }
    WSTAMP(5);
    __syncthreads();
    WSTAMP(6);
    // ---- (B) backward chains: h_k = g'_k - N_kp^T h_kp, outwards from the block next to the middle ------------
    {
        const int nst = (half ? nbb : nbt) - 1;
        if (chain_wave && nst > 0) {
            const int kb0 = half ? mk + 2 : mk - 1, dk = half ? 1 : -1;
            T *const rw = GH + (size_t)cl * rows * LW_ROW;
            const T v0 = rw[kb0 * LW_ROW + (lane & 3)];
            wide_chain<T, true>(rw, nbase, nrow, (unsigned)cl * 16, kb0 + dk, kb0, dk, nst, lane & 15, v0);
        }
    }
    WSTAMP(7);
    __syncthreads();
    WSTAMP(8);
    // ---- (E) per block: x = T (c - h), scatter ------------------------------------------------------
    if (has && !isq) {
        const T *const hr = GH + ((size_t)ll * rows + k) * LW_ROW;
        T h[4], c[5], x[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = hr[q];
#pragma unroll
        for (int q = 0; q < 5; ++q) c[q] = r[q];
        emg::wide_x<T, 15>(Tk, c, h, x);
        emg::wide_block_scatter<T, DIR>(A, k, mir, i1, i2, x);
    }
    WSTAMP(9);
}

// ---- fused colour pass with STREAMED records (the largest levels): k_line_stream -----------------
// k_line_colour runs right-hand sides, forward and backward substitution one after the other, and
// on the levels whose records do not fit the LDS of a CU the right-hand sides make a round trip
// through the global scratch (written by the first phase, read by the second: 160 B per block of
// the pass's ~1450). Here six producer waves PRODUCE the right-hand sides of the
// next R block rows into an LDS ring while the two chain waves CONSUME the current R rows in their
// forward half-chains: the right-hand sides never leave the CU, and their assembly (a bandwidth
// phase) overlaps the forward substitution (a latency chain). In the backward pass the producers
// copy the w records of the next R steps into the same ring (wide coalesced loads by otherwise idle
// waves; the chain quads keep no register ring for them). Same arithmetic, entry by entry, as
// k_line_colour (stencil.h: line_rhs_e0 / line_rhs_t, quad_forward_step, quad_backward_step):
// bit-identical results.
//
// The same workgroup can serve its 16 lines for a GROUP of B <= 4 right-hand sides that share the
// factors -- the sources of one frequency (emg3d/simulations.py:1453-1464): a chain quad holds the
// factor row of a block in registers once and applies it to the B right-hand sides (B independent
// dependency chains: their instructions interleave), the producers fill B rings. With the batch as a
// grid dimension (k_line_colour<BATCH>) every source's workgroups stream the 2 x 304 B of factors per
// block again -- 47 % of the bytes of a level-0 colour pass, and PMC shows no merging in L2 (12-13 GB
// per launch of two sources against 5.2-5.6 GB for one). Measured at 256^3 (profiles/r04_batch_lines_*):
// 0.78-0.79 x the single-source time per source for B = 2, 0.72-0.76 x for B = 4.
//
// LDS: ring [2 buffers][B][2 halves][R rows][16 lines][5 entries] -- item (source, half, step i) holds
// the five right-hand-side (forward) / w (backward) entries of the block that half's chain works on
// at step i, already in the chain's grouping (bottom half = mirrored blocks: entry 0 of record row k,
// entries 1..4 of row k - 1). R = 16 for B <= 2, 8 for B = 3, 4 (160 KB). The w / solution records
// live in the global scratch (with them in LDS only 8 lines of 256 blocks fit a workgroup, and half-
// filled chain waves cost more than the bytes save: measured). Producer waves: four for one source
// (with two the producers, one memory round trip per item, are what the chains wait for), six for
// groups -- the kernel is held to 256 registers by its chain waves' SIMD partners anyway, so two more
// cost nothing and keep more loads in flight (B = 2: 0.84-0.87 -> 0.73-0.76 x per source). One
// workgroup barrier (LDS-only: the chains' factor prefetch stays in flight) per R steps. The producers
// also put the raw right-hand sides of the rows the middle block reads (row m, and entry 0 of row
// m + 1) into the records.
//   RD : depth of the factor prefetch ring in the chain waves (4 for one source; 2 for B >= 2: a step
//        of B sources takes B times as long, so two steps ahead is as far ahead in time).
// lfo != nullptr: also the eight coupling entries of the item's block (stencil.h: line_coupling -- from the zeta
// values the right-hand side needs anyway) into the coupling ring [2 halves][R rows][lpw lines][8]
template <class T, int DIR>
__device__ __forceinline__ void stream_produce(const emg::Axes<T, DIR> &A, int colour, int cntp, int cntq, int n0p,
                                               int line0, int nl, int lpw, T *buf, int R, int chunk, int pt, int np,
                                               double *lfo = nullptr)
{
    const int n0 = A.n0();
    const int items = 2 * R * lpw;
    for (int it = pt; it < items; it += np) {
        // x-lines: the lanes run along the line (the field is contiguous there); else across the lines
        int ll, row, half;
        if (DIR == 0) { row = it % R; ll = (it / R) % lpw; half = it / (R * lpw); }
        else { ll = it % lpw; row = (it / lpw) % R; half = it / (R * lpw); }
        const int i = chunk * R + row;                       // forward step of the half
        const int k = half ? n0p - 1 - i : i;                // its block
        const int k0 = k, kt = half ? k - 1 : k;             // record rows of entry 0 / entries 1..4
        const int lid = line0 + min(ll, nl - 1);
        int i1, i2, l2;
        emg::line_of_thread<DIR>(colour, cntp, cntq, lid % cntp, lid / cntp, i1, i2, l2);
        T rhs[5];
        emg::line_rhs_t<T, DIR>(A, min(max(kt, 0), n0 - 1), i1, i2, rhs);
        rhs[0] = emg::line_rhs_e0<T, DIR>(A, min(max(k0, 0), n0 - 1), i1, i2);
        const double keep0 = (k0 >= 0 && k0 < n0) ? 1.0 : 0.0;          // identity padding blocks: rhs = 0
        const double keept = (kt >= 0 && kt < n0) ? 1.0 : 0.0;
        T *o = buf + ((size_t)(half * R + row) * lpw + ll) * 5;
        o[0] = keep0 * rhs[0];
#pragma unroll
        for (int r = 1; r < 5; ++r) o[r] = keept * rhs[r];
        if (lfo) {
            double c[8];
            emg::line_coupling<T, DIR>(A, k, i1, i2, half != 0, c);
            double *lo = lfo + ((size_t)(half * R + row) * lpw + ll) * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) lo[r] = c[r];
        }
    }
}

constexpr int LS_PROD = 384;                 // producer threads of k_line_stream for groups (6 waves; one source: 4; + 2 chain waves)

// forward half-chain that takes its right-hand sides from the LDS ring, for B right-hand sides
// LFR: the coupling entries come from the coupling ring (lfring), not from the lfac records
// FT / WT: storage types of the T records / the w records (T, or emg::compact_of<T>); vec: the w records of the
// group's first source as WT, source b's b * vstride elements (of WT) behind them
template <class T, int HALF, int RD, int B, bool LFR, class FT = T, class WT = T>
__device__ __forceinline__ void quad_forward_stream(int n0, int n0p, int nlines, int qline, int qend, int line0, int j,
                                                      const FT *fac, const double *lfac, WT *vec, size_t vstride,
                                                      size_t dummy_off, const T *ringbase, int lpw, int R, int nchunks,
                                                      const double *lfring)
{
    const HalfWalk<HALF> W(n0, n0p);
    const bool active = qline < qend;
    const int line = min(qline, qend - 1);
    const int ll = line - line0;
    const VecRef<WT> V = VecRef<WT>::global(vec, nlines);
    WT *const dummy = vec + dummy_off;                           // (every source's scratch has its dummy slots)
    WT *const dslot = dummy + ((threadIdx.x & 63) >> 2) * 5;
    QuadRow<T, FT> ring[RD];
    using LAddr = LaneAddr<T, HALF, false, FT, WT>;
    const LAddr LA(fac, lfac, nlines, line, j, V);
    auto fetch = [&](QuadRow<T, FT> &q, int i) { q.template load<LAddr, false, !LFR>(LA, W.fwd(W.clampi(i))); };
#pragma unroll
    for (int d = 0; d < RD; ++d) fetch(ring[d], d);
    __syncthreads();                                          // chunk 0 of the rings and the middle rows are there
    T wsel[B], w4p[B];
#pragma unroll
    for (int b = 0; b < B; ++b) wsel[b] = w4p[b] = emg::zero<T>();
    const double nz = j != 0 ? 1.0 : 0.0, is0 = 1.0 - nz;
    const size_t srcelems = (size_t)2 * R * lpw * 5;          // ring of one source, one buffer
    const size_t bufelems = (size_t)B * srcelems;
    for (int c = 0; c < nchunks; ++c) {
        const T *const items = ringbase + (size_t)(c & 1) * bufelems + ((size_t)(HALF * R) * lpw + ll) * 5;
        const int iend = min((c + 1) * R, W.steps);
        for (int i0 = c * R; i0 < iend; i0 += RD) {
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                const int k = W.fwd(i0 + d);
                QuadRow<T, FT> &q = ring[d];
                const T *const it = items + (size_t)(i0 + d - c * R) * lpw * 5;
                if constexpr (LFR)
                    q.take_b(lfring + (size_t)(c & 1) * ((size_t)2 * R * lpw * 8) +
                             ((size_t)(HALF * R + (i0 + d - c * R)) * lpw + ll) * 8, j);
                WT *const o4 = active ? LA.pv4(k) : dslot + 4;
                WT *const oj = active ? LA.pvj(k) : dslot + j;
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const T v = it[b * srcelems + j], v4 = it[b * srcelems + 4];
                    T wn, w4;
                    quad_forward_step(q, v, v4, nz, is0, wsel[b], w4p[b], wn, w4);
                    oj[b * vstride] = emg::narrow<WT>(wn);
                    o4[b * vstride] = emg::narrow<WT>(w4);
                }
                fetch(ring[d], i0 + d + RD);
            }
        }
        lds_barrier();                                        // this chunk is consumed, the next one produced
    }
}

// The producers' job in the BACKWARD pass: copy the w records of the next R steps of both half-chains
// into the ring, in the chains' grouping (stream_produce's: entry 0 of record row k, entries 1..4 of
// row k - 1 for a mirrored block) -- sixteen lines x 80 B are contiguous in the scratch, so the idle
// producer waves fetch them as wide coalesced loads and the chain quads need neither a register ring
// nor four scattered 16-byte loads per step for them.
template <class T, int DIR, class WT = T>
__device__ __forceinline__ void stream_produce_w(const WT *vec, int nlines, int n0, int n0p, int line0, int nl, int lpw,
                                                 T *buf, int R, int chunk, int pt, int np,
                                                 const emg::Axes<T, DIR> *A = nullptr, int colour = 0, int cntp = 0,
                                                 int cntq = 0, double *lfo = nullptr)
{
    const int mk = emg::line_mid(n0);
    const int items = 2 * R * lpw;
    for (int it = pt; it < items; it += np) {
        const int ll = it % lpw, row = (it / lpw) % R, half = it / (R * lpw);
        const int steps = half ? n0p - 2 - mk : mk;
        const int ic = max(min(chunk * R + row, steps - 1), 0);
        const int k = min(max(half ? mk + 2 + ic : mk - 1 - ic, half), n0p - 1);     // HalfWalk::bwd
        const int lid = line0 + min(ll, nl - 1);
        const WT *const r0 = vec + ((size_t)k * nlines + lid) * 5;
        const WT *const rt = vec + ((size_t)(half ? k - 1 : k) * nlines + lid) * 5;
        const WT a0 = r0[0], a1 = rt[1], a2 = rt[2], a3 = rt[3], a4 = rt[4];
        T *o = buf + ((size_t)(half * R + row) * lpw + ll) * 5;
        o[0] = emg::widen(a0); o[1] = emg::widen(a1); o[2] = emg::widen(a2); o[3] = emg::widen(a3); o[4] = emg::widen(a4);
        if (lfo) {
            // the coupling entries of block k from four zeta values (32 B) instead of its lfac record (64 B)
            int i1, i2, l2;
            emg::line_of_thread<DIR>(colour, cntp, cntq, lid % cntp, lid / cntp, i1, i2, l2);
            double c[8];
            emg::line_coupling<T, DIR>(*A, k, i1, i2, half != 0, c);
            double *lo = lfo + ((size_t)(half * R + row) * lpw + ll) * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) lo[r] = c[r];
        }
    }
}

// backward substitution of one half for B right-hand sides (quad_backward, MIDFIRST form, per source);
// the w records come from the LDS ring (stream_produce_w)
// vec / vstride as in quad_forward_stream (WT); fdummy: dummy store targets of FIELD type for surplus quads and
// padding blocks (global memory, like the field)
template <class T, int DIR, int HALF, int RD, int B, bool PAIR, bool LFR, class FT = T, class WT = T>
__device__ __forceinline__ void quad_backward_stream(const emg::Level<T> &L, int colour, int cntp, int cntq, int n0p, int qline,
                                                int qend, int line0, int j, const FT *fac, const double *lfac, WT *vec,
                                                size_t vstride, T *fdummy, size_t boff0, const T *ringbase, int lpw, int R,
                                                int nchunks, const double *lfring)
{
    const emg::Axes<T, DIR> A(L, boff0);
    const int n0 = A.n0();
    const HalfWalk<HALF> W(n0, n0p);
    const int mk = W.mk;
    const int nlines = cntp * cntq;
    const bool active = qline < qend;
    const int line = min(qline, qend - 1);
    const int ll = line - line0;
    const size_t bs = L.bstride;
    int i1, i2, lid;
    emg::line_of_thread<DIR>(colour, cntp, cntq, line % cntp, line / cntp, i1, i2, lid);
    const int cj = j == 0 ? 0 : (j <= 2 ? 1 : 2), dk = (j == 0 || HALF) ? 0 : 1, dk4 = HALF ? 0 : 1;
    const int d1 = j == 1 ? 1 : 0, d2 = j == 3 ? 1 : 0;
    T *const ej = A.E(cj) + A.idx(cj, dk, i1 - d1, i2 - d2);
    const long sj = (long)A.idx(cj, dk + 1, i1 - d1, i2 - d2) - (long)A.idx(cj, dk, i1 - d1, i2 - d2);
    T *const e4 = A.E(2) + A.idx(2, dk4, i1, i2);
    const long s4 = (long)A.idx(2, dk4 + 1, i1, i2) - (long)A.idx(2, dk4, i1, i2);
    // dummy store targets: the dummy slots of the group's first scratch (global memory, like the field)
    T *const dslot = fdummy + ((threadIdx.x & 63) >> 2) * 5;
    T *const dj = dslot + j, *const d4 = dslot + 4;
    const size_t fstep = active ? bs : 0;                     // per-source step of the field pointers

    const VecRef<WT> V = VecRef<WT>::global(vec, nlines);
    QuadRow<T, FT> ring[RD];
    using LAddr = LaneAddr<T, HALF, false, FT, WT>;
    const LAddr LA(fac, lfac, nlines, line, j, V);
    auto fetch = [&](QuadRow<T, FT> &q, int i) {
        q.template load<LAddr, false, !LFR>(LA, min(max(W.bwd(W.clampi(i)), HALF), n0p - 1));
    };
    // coupling to the middle: B_m (top) / U_{m+1} (bottom)
    QuadRow<T> qm;
    qm.load_b(lfac, (size_t)(HALF ? mk + 1 : mk) * nlines + line, j);
    T x0[B], x4[B], xmine[B];
    const double own0 = j == 0 ? 1.0 : 0.0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const VecRef<WT> Vb = VecRef<WT>::global(vec + b * vstride, nlines);
        T xa, xb;
        quad_middle<T, FT, WT>(n0, n0p, nlines, line, j, fac, lfac, Vb, xa, xb);
        const T xq0 = quad_bcast<0>(xa), xq4 = quad_bcast<0>(xb), xq5 = quad_bcast<1>(xb);
        if (HALF == 0) {
            const int dkm = j == 0 ? 0 : 1;
            T *const pm = A.E(cj) + A.idx(cj, mk + dkm, i1 - d1, i2 - d2) + b * bs;
            *(active ? pm : dj) = xa;
            T *const p4 = A.E(2) + A.idx(2, mk + 1, i1, i2) + b * bs;
            T *const p5 = A.E(0) + A.idx(0, mk + 1, i1, i2) + b * bs;
            *((active && j == 0) ? p4 : ((active && j == 1) ? p5 : d4)) = xb;
        }
        x0[b] = HALF ? xq5 : xq0;
        x4[b] = xq4;
        xmine[b] = HALF ? emg::mad(own0, xq5, xop::mul(1.0 - own0, xa)) : xa;
    }
    asm volatile("" ::: "memory");           // keep the ring fetch behind the middle blocks
#pragma unroll
    for (int d = 0; d < RD; ++d) fetch(ring[d], d);
    __syncthreads();                                          // chunk 0 of the w ring is there
    double upA = qm.bA, upD = qm.bD, up04 = qm.b04, up44 = qm.d4;          // entries of the coupling block
    T *pej = active ? ej + (long)W.bwd(0) * sj : dj;
    T *pe4 = active ? e4 + (long)W.bwd(0) * s4 : d4;
    const long incj = active ? (HALF ? sj : -sj) : 0, inc4 = active ? (HALF ? s4 : -s4) : 0;
    const double nz = j != 0 ? 1.0 : 0.0;
    const size_t srcelems = (size_t)2 * R * lpw * 5, bufelems = (size_t)B * srcelems;
    constexpr bool PAIRED = PAIR && DIR == 0 && RD % 2 == 0;  // (steps come in pairs: W.steps is a multiple of 4)
    T hold_j[B], hold_4[B];
    T *hold_oj = dj, *hold_o4 = d4;
    size_t hold_step = 0;
    for (int c = 0; c < nchunks; ++c) {
        const T *const items = ringbase + (size_t)(c & 1) * bufelems + ((size_t)(HALF * R) * lpw + ll) * 5;
        const int iend = min((c + 1) * R, W.steps);
        for (int i0 = c * R; i0 < iend; i0 += RD) {
#pragma unroll
            for (int d = 0; d < RD; ++d) {
                const int k = W.bwd(i0 + d);
                QuadRow<T, FT> &q = ring[d];
                const T *const it = items + (size_t)(i0 + d - c * R) * lpw * 5;
                if constexpr (LFR)
                    q.take_b(lfring + (size_t)(c & 1) * ((size_t)2 * R * lpw * 8) +
                             ((size_t)(HALF * R + (i0 + d - c * R)) * lpw + ll) * 8, j);
                const bool real_block = HALF ? k <= n0 - 1 : true;      // uniform over the wave
                T *const oj = real_block ? pej : dj;
                T *const o4 = real_block ? pe4 : d4;
                const size_t ostep = real_block ? fstep : 0;
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const T wj = it[b * srcelems + j], w4 = it[b * srcelems + 4];
                    T xn, xn4;
                    quad_backward_step(q, wj, w4, nz, upA, upD, up04, up44, x0[b], x4[b], xmine[b], xn, xn4);
                    if (PAIRED && (d & 1) == 0) {
                        // x-lines: a lane's results of consecutive blocks are neighbours in memory (16 B each).
                        // Stored step by step a 128-B line is touched eight times, ~1.2 us apart with four
                        // right-hand sides, and the open lines of an XCD's 32 workgroups (5000 each) overflow its
                        // L2: every piece goes to HBM as its own sector write (PMC: 1.27 GB written per
                        // source-launch against 0.79 GB for one source). The even step of a pair is held and
                        // stored together with the odd one: 32 contiguous bytes at a time.
                        hold_j[b] = xn; hold_4[b] = xn4;
                    } else if (PAIRED) {
                        hold_oj[b * hold_step] = hold_j[b];
                        oj[b * ostep] = xn;
                        hold_o4[b * hold_step] = hold_4[b];
                        o4[b * ostep] = xn4;
                    } else {
                        oj[b * ostep] = xn;
                        o4[b * ostep] = xn4;
                    }
                }
                if (PAIRED && (d & 1) == 0) { hold_oj = oj; hold_o4 = o4; hold_step = ostep; }
                upA = q.bA; upD = q.bD; up04 = q.b04; up44 = q.d4;
                pej += incj;
                pe4 += inc4;
                fetch(ring[d], i0 + d + RD);
            }
        }
        lds_barrier();                                        // this chunk is consumed, the next one copied
    }
}

// NPROD: producer threads (four waves for a single source; six for groups -- with the kernel held to 256
// registers by its six waves anyway, two more producer waves cost nothing and keep more loads in flight)
// LFR: the coupling entries (the 8 reals of a block's lfac record) are recomputed by the producers and handed
// over through a second ring [2 buffers][2 halves][R][lpw][8] behind the first: 64 B per block less to fetch in
// the forward pass (the producers hold the zeta values already), 32 B less in the backward pass.
// COMPACT: the T records (`facv`) and the w records (in the scratch) are stored as emg::compact_of<T> -- single
// precision, rounded once when the set-up / the forward pass stores them and widened when they are loaded; the
// right-hand sides, the rings, every operation and the solution stay in T. 120 + 2 x 40 instead of 240 + 2 x 80
// of the ~1 210 B a block costs per colour pass on the levels that live in HBM.
template <class T, int DIR, int B, int RD, int NPROD, bool PAIR, bool LFR, bool COMPACT = false>
__global__ __launch_bounds__(128 + NPROD, 1) void k_line_stream(emg::Level<T> L, int colour, int cntp, int cntq, int n0p,
                                                                    int lpw, int R, const void *facv, const double *lfac,
                                                                    T *vecT, size_t vstrideT, size_t boff0)
{
    // vecT: the scratch of the group's first right-hand side (source b's: b * vstrideT elements of T behind it);
    // boff0: element offset of the group's first source in the field / source buffers
    using FT = typename std::conditional<COMPACT, typename emg::compact_of<T>::type, T>::type;
    using WT = FT;
    const FT *const fac = reinterpret_cast<const FT *>(facv);
    WT *const vec = reinterpret_cast<WT *>(vecT);
    const size_t vstride = vstrideT * (sizeof(T) / sizeof(WT));          // in elements of WT
    const size_t dummy_off = vstrideT - emg::LINE_DUMMY;                  // behind a source's records (either type)
    T *const fdummy = vecT + dummy_off;
    extern __shared__ double2 ls_smem[];
    const int nlines = cntp * cntq;
    const int line0 = blockIdx.x * lpw;
    const int nl = min(lpw, nlines - line0);
    T *const ringbase = reinterpret_cast<T *>(ls_smem);
    const size_t srcelems = (size_t)2 * R * lpw * 5, bufelems = (size_t)B * srcelems;
    double *const lfring = LFR ? reinterpret_cast<double *>(ringbase + 2 * bufelems) : nullptr;
    const size_t lfelems = (size_t)2 * R * lpw * 8;
    const int n0 = DIR == 0 ? L.nx : DIR == 1 ? L.ny : L.nz;
    const int mk = emg::line_mid(n0);
    const int smax = max(mk, n0p - 2 - mk);
    const int nchunks = (smax + R - 1) / R;
    const int wave = threadIdx.x >> 6;
    if (wave >= 2) {
        // ---- producers: right-hand sides for the forward pass, w records for the backward pass
        const int pt = threadIdx.x - 128;
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const emg::Axes<T, DIR> A(L, boff0 + b * L.bstride);
            const VecRef<WT> V = VecRef<WT>::global(vec + b * vstride, nlines);
            for (int ll = pt; ll < nl; ll += NPROD) {
                const int lid = line0 + ll;
                int i1, i2, l2;
                emg::line_of_thread<DIR>(colour, cntp, cntq, lid % cntp, lid / cntp, i1, i2, l2);
                T rhs[5];
                emg::line_rhs<T, DIR>(A, mk, i1, i2, rhs);
#pragma unroll
                for (int r = 0; r < 4; ++r) *V.p(mk, lid, r) = emg::narrow<WT>(rhs[r]);
                *V.p4(mk, lid) = emg::narrow<WT>(rhs[4]);
                *V.p(mk + 1, lid, 0) = emg::narrow<WT>(emg::line_rhs_e0<T, DIR>(A, min(mk + 1, n0 - 1), i1, i2));
            }
            stream_produce<T, DIR>(A, colour, cntp, cntq, n0p, line0, nl, lpw, ringbase + b * srcelems, R, 0, pt, NPROD,
                                   (LFR && b == 0) ? lfring : nullptr);
        }
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) {
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    const emg::Axes<T, DIR> A(L, boff0 + b * L.bstride);
                    stream_produce<T, DIR>(A, colour, cntp, cntq, n0p, line0, nl, lpw,
                                           ringbase + (size_t)((c + 1) & 1) * bufelems + b * srcelems, R, c + 1, pt, NPROD,
                                           (LFR && b == 0) ? lfring + (size_t)((c + 1) & 1) * lfelems : nullptr);
                }
            }
            lds_barrier();
        }
        __syncthreads();                                      // the forward chains are done: all w records are written
        const emg::Axes<T, DIR> A0(L, boff0);
#pragma unroll 1
        for (int b = 0; b < B; ++b)
            stream_produce_w<T, DIR, WT>(vec + b * vstride, nlines, n0, n0p, line0, nl, lpw, ringbase + b * srcelems, R, 0, pt, NPROD,
                                         &A0, colour, cntp, cntq, (LFR && b == 0) ? lfring : nullptr);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) {
#pragma unroll 1
                for (int b = 0; b < B; ++b)
                    stream_produce_w<T, DIR, WT>(vec + b * vstride, nlines, n0, n0p, line0, nl, lpw,
                                                 ringbase + (size_t)((c + 1) & 1) * bufelems + b * srcelems, R, c + 1, pt, NPROD,
                                                 &A0, colour, cntp, cntq,
                                                 (LFR && b == 0) ? lfring + (size_t)((c + 1) & 1) * lfelems : nullptr);
            }
            lds_barrier();
        }
        return;
    }
    const int half = wave & 1;
    const int qline = line0 + ((threadIdx.x & 63) >> 2), j = threadIdx.x & 3;
    const int qend = line0 + nl;
    if (half == 0) quad_forward_stream<T, 0, RD, B, LFR, FT, WT>(n0, n0p, nlines, qline, qend, line0, j, fac, lfac, vec, vstride, dummy_off, ringbase, lpw, R, nchunks, lfring);
    else quad_forward_stream<T, 1, RD, B, LFR, FT, WT>(n0, n0p, nlines, qline, qend, line0, j, fac, lfac, vec, vstride, dummy_off, ringbase, lpw, R, nchunks, lfring);
    __syncthreads();
    if (half == 0) quad_backward_stream<T, DIR, 0, RD, B, PAIR, LFR, FT, WT>(L, colour, cntp, cntq, n0p, qline, qend, line0, j, fac, lfac, vec, vstride, fdummy, boff0, ringbase, lpw, R, nchunks, lfring);
    else quad_backward_stream<T, DIR, 1, RD, B, PAIR, LFR, FT, WT>(L, colour, cntp, cntq, n0p, qline, qend, line0, j, fac, lfac, vec, vstride, fdummy, boff0, ringbase, lpw, R, nchunks, lfring);
}

// End of the spelled-out section: back to the mode the translation unit is compiled with -- -ffp-contract=fast-honor-
// pragmas, which emg3d_amd/_lib.py (HIPCC_FLAGS) passes explicitly, so that this line and the command line cannot
// drift apart. (The pragma can only name a mode; a push / pop of the previous one does not exist on this target:
// `#pragma float_control(push)` is "not supported on this target - ignored" by amdgcn clang, checked with hipcc 7.2.)
#pragma clang fp contract(fast)

// Residual + per-block partial sums of |r|^2. A workgroup walks `zb` consecutive planes: the plane
// above the current one, fetched for the curl, is the next iteration's own plane and is still in
// the CU's cache then -- with one plane per workgroup the three workgroups that need a plane run on
// different XCDs at different times and each fetches it from HBM (PMC, 256^3: 399 B read per cell
// against 144).
template <class T, bool ROLL>
__global__ __launch_bounds__(256) void k_residual(emg::Level<T> L0, T *rx, T *ry, T *rz, double *partial, int nzb, int zb)
{
    // grid.z = plane blocks x right-hand sides; the residual buffers are stacked like the fields, and
    // source b owns the partial sums [b nblk, (b+1) nblk) (blockIdx.z runs over both)
    const int b = blockIdx.z / nzb;
    const emg::Level<T> L = emg::source_level(L0, b);
    if (rx) { rx += b * L0.bstride; ry += b * L0.bstride; rz += b * L0.bstride; }
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int z0 = (blockIdx.z - b * nzb) * zb, z1 = min(z0 + zb, L.nz + 1);
    double acc = 0.0;
    if (ix <= L.nx && iy <= L.ny) {
        if constexpr (ROLL) acc = emg::residual_column<T>(L, rx, ry, rz, ix, iy, z0, z1);
        else for (int iz = z0; iz < z1; ++iz) acc += emg::residual_cell<T>(L, rx, ry, rz, ix, iy, iz);
    }
    if (partial) {
        // wave reduction (64 lanes), then across the 4 waves through LDS
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __shared__ double wsum[4];
        const int tid = threadIdx.y * blockDim.x + threadIdx.x;
        if ((tid & 63) == 0) wsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) {
            const int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            partial[bid] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        }
    }
}

// Deterministic final reduction of the partial sums (single workgroup).
__global__ __launch_bounds__(256) void k_reduce_sum(const double *partial, int n, double *out)
{
    partial += (size_t)blockIdx.x * n;      // one workgroup per right-hand side
    out += blockIdx.x;
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sm[0];
}

template <class T> __global__ __launch_bounds__(256) void k_restrict(emg::Restrict<T> R)
{
    const int cix = blockIdx.x * blockDim.x + threadIdx.x;
    const int ciy = blockIdx.y * blockDim.y + threadIdx.y;
    const int b = blockIdx.z / R.cnzn, ciz = blockIdx.z - b * R.cnzn;    // grid.z = planes x right-hand sides
    if (cix >= R.cnxn || ciy >= R.cnyn) return;
    R.rx += b * R.fstride; R.ry += b * R.fstride; R.rz += b * R.fstride;
    R.crx += b * R.cstride; R.cry += b * R.cstride; R.crz += b * R.cstride;
    if (R.cex) { R.cex += b * R.cstride; R.cey += b * R.cstride; R.cez += b * R.cstride; }
    emg::restrict_node<T>(R, cix, ciy, ciz);
}

template <class T> __global__ __launch_bounds__(256) void k_prolong(emg::Prolong<T> P)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int b = blockIdx.z / (P.nz + 1), iz = blockIdx.z - b * (P.nz + 1);   // planes x right-hand sides
    if (ix > P.nx || iy > P.ny) return;
    P.ex += b * P.fstride; P.ey += b * P.fstride; P.ez += b * P.fstride;
    P.cex += b * P.cstride; P.cey += b * P.cstride; P.cez += b * P.cstride;
    emg::prolong_cell<T>(P, ix, iy, iz);
}

template <class T>
__global__ __launch_bounds__(256) void k_restrict_param(T *out, const T *in, int nx, int ny, int cnx,
                                                        int cny, int fx, int fy, int fz)
{
    const int cix = blockIdx.x * blockDim.x + threadIdx.x;
    const int ciy = blockIdx.y * blockDim.y + threadIdx.y;
    const int ciz = blockIdx.z;
    if (cix >= cnx || ciy >= cny) return;
    emg::restrict_param_cell<T>(out, in, nx, ny, cnx, cny, fx, fy, fz, cix, ciy, ciz);
}

// Zero the tangential components on the PEC faces (reference emg3d/solver.py:349-355).
template <class T>
__global__ __launch_bounds__(256) void k_pec_zero(T *ex, T *ey, T *ez, int nx, int ny, int nz)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int iz = blockIdx.z;
    if (ix > nx || iy > ny) return;
    const bool bx = ix == 0 || ix == nx, by = iy == 0 || iy == ny, bz = iz == 0 || iz == nz;
    if (ix < nx && (by || bz)) ex[ix + nx * (iy + (ny + 1) * iz)] = emg::zero<T>();
    if (iy < ny && (bx || bz)) ey[ix + (nx + 1) * (iy + ny * iz)] = emg::zero<T>();
    if (iz < nz && (bx || by)) ez[ix + (nx + 1) * (iy + (ny + 1) * iz)] = emg::zero<T>();
}

template <class T> __global__ void k_band_solve(T *amat, T *bvec, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) emg::band_solve<T>(amat, bvec, n);
}

template <class T>
__global__ void k_blocks_to_amat(T *amat, T *bvec, const T *middle, const double *left, const T *rhs,
                                 int im, int nc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) emg::blocks_to_amat<T>(amat, bvec, middle, left, rhs, im, nc);
}

// --------------------------------------------------------------------------- launchers --

// How one colour pass of a line direction is launched (decided by line_plan, executed by
// launch_line_colour; exported through emg3d_line_kernel_name so that callers -- bench.py, the tests --
// name the kernel that runs instead of re-deriving the rule).
enum LineKind { LK_SEPARATE = 0, LK_COLOUR = 1, LK_STREAM = 3 };
struct LinePlan {
    int kind;        // LineKind
    int vmode;       // LK_COLOUR: where the records live (k_line_colour's VMODE 0..3)
    int lpw;         // lines per workgroup
    int R;           // LK_STREAM / LK_STREAM: rows per chunk of the right-hand-side ring
    size_t smem;     // dynamic LDS of the launch
    bool shortl;     // the records were laid out with the short granule (lines of <= LINE_SHORT blocks)
    bool batchk;     // LK_COLOUR: the instantiation with the batch as a grid dimension
};
// largest group of right-hand sides one k_line_stream workgroup serves (its chain quads hold a factor
// row once and apply it to all of them)
constexpr int LSB_MAX = 4;
// rows per chunk of the ring of a group of g right-hand sides: 2 buffers x g x 2 halves x R x 16 lines x
// 5 entries must fit the 160 KB of a CU
inline int stream_rows(int g, size_t elem, bool lfr = false)
{
    int R = g_line_stream_r > 0 ? g_line_stream_r : 16;
    while (R > 4 && (size_t)2 * 2 * R * 16 * (g * 5 * elem + (lfr ? 64 : 0)) > (size_t)160 * 1024) R -= 4;
    return R;
}
template <class T> LinePlan line_plan(const emg::LineClass &lc, int batch)
{
    LinePlan P{LK_SEPARATE, 0, 16, 0, 0, emg::line_pad(lc.n0) == emg::LINE_PAD_SHORT, false};
    if (!(g_line_fuse == 1 || (g_line_fuse == 2 && lc.lines <= g_line_fuse_max))) return P;
    P.kind = LK_COLOUR;
    // lines per workgroup: as few as keeps the workgroup count within one per CU
    int lpw = 16;
    if (g_line_lpw > 0) lpw = g_line_lpw;
    else if (cdiv(lc.lines, 4) * batch <= 256) lpw = 4;      // all right-hand sides count
    else if (cdiv(lc.lines, 8) * batch <= 256) lpw = 8;
    // records in LDS if they fit (+ the dummy slots) and every workgroup gets a CU;
    // line_lds = 2: whenever they fit, with fewer lines per workgroup if the lines are too
    // long for 16 (experiment)
    const size_t lds_cu = 160 * 1024;
    auto rec_bytes = [&](int l, int w) { return ((size_t)l * lc.n0p * w + emg::LINE_DUMMY) * sizeof(T); };
    if (g_line_lds == 2 && g_line_lpw == 0) {
        while (lpw > 4 && rec_bytes(lpw, 4) > lds_cu) lpw /= 2;
    }
    P.lpw = lpw;
    const unsigned nwg = cdiv(lc.lines, lpw);
    // (the QD = 4 kernels hold more than 256 registers: one workgroup per CU whatever its LDS use,
    // so their records go to LDS whenever they fit; the short-line kernels share a CU and keep
    // their records in LDS only while every workgroup of the launch gets a CU at once)
    const bool one_per_cu = !P.shortl && batch == 1;
    auto fits = [&](size_t smem) {
        return g_line_lds && smem <= lds_cu &&
               (g_line_lds >= 2 || one_per_cu || (size_t)nwg <= 256 * (lds_cu / smem));
    };
    const size_t smem1 = rec_bytes(lpw, 5);
    const size_t smem2 = rec_bytes(lpw, 4);
    P.batchk = batch > 1 || g_line_occ2;
    // (streaming also the levels whose whole records fit -- 64-block lines and shorter -- measured slower: 84.5
    // against 83.8 ms per config-3 cycle)
    const bool streamable = g_line_stream && !P.shortl && !g_line_occ2 && !(g_line_debug & 1) && lpw <= 16 && !fits(smem1) &&
                            (!fits(smem2) || g_line_stream >= 2);
    // the largest levels of a single-source solve: right-hand sides streamed through LDS
    // (where slots 0..3 of the records fit in LDS -- 128-block lines -- k_line_colour's mode 2 is as fast:
    // 11.34 against 11.40 ms per config-2 cycle, 1.80-1.91 against 1.82-1.88 ms per call at 256 x 128 x 128).
    // (k_line_stream has two chain waves = 16 lines per workgroup at most: with line_lpw = 32 the launch
    // stays with k_line_colour, whose waves 2 / 3 walk lines 16..31; the ring must fit the LDS of a CU)
    if (streamable && batch == 1 && lc.n0 >= 16) {
        P.kind = LK_STREAM;      // (a single source: a group of one)
        return P;
    }
    // several right-hand sides on such a level: groups of up to LSB_MAX of them per workgroup, the
    // factors fetched once per group (k_line_stream)
    // (also where slots 0..3 of the records of ONE source would fit in LDS: with the batch as a grid dimension
    // those launches fetch the factors once per source)
    const bool streamable_b = g_line_stream && !P.shortl && !g_line_occ2 && !(g_line_debug & 1) && lpw == 16 && !fits(smem1);
    if (streamable_b && batch > 1 && g_line_stream_bmin > 0 && lc.n0 >= g_line_stream_bmin) {
        P.kind = LK_STREAM;
        return P;
    }
    if (P.shortl) {              // (records of <= 6 blocks: a few KB, they fit whenever LDS records are on)
        if (fits(smem1)) { P.vmode = 1; P.smem = smem1; }
    } else if (fits(smem1)) { P.vmode = 1; P.smem = smem1; }
    else if (fits(smem2)) { P.vmode = 2; P.smem = smem2; }
    else if (g_line_lds >= 3 && batch == 1 && !g_line_occ2 && lpw == 16) {
        // line_lds = 3 (experiment): lines too long for mode 2 keep the record rows around the
        // middle block in LDS, the outer rows in the global scratch (mode 3)
        const emg::LineSplit sp = emg::line_split_rows(lc.n0, lc.n0p, lpw, sizeof(T));
        if (sp.lds_bytes > 0) { P.vmode = 3; P.smem = sp.lds_bytes; P.batchk = false; }
    }
    return P;
}

// does the direction run k_line_wide on this level? (option line_wide = longest line; the level must hold N records)
inline bool line_wide_used(int dir, int nx, int ny, int nz)
{
    return g_line_wide > 0 && emg::line_n0(dir, nx, ny, nz) <= g_line_wide && emg::line_nfac_elems(dir, nx, ny, nz) > 0;
}

// Does direction `dir` of this level keep COMPACT line factors (k_line_stream<.., COMPACT>)? The level asks for it
// (emg3d_level::flags, or option line_compact = 1), and every colour class of the direction runs the streamed kernel
// for the level's number of right-hand sides (one source, or groups of them) -- the set-up (which stores the T records
// in that form), the size query and the launcher all decide with this one function of the level and the options.
template <class T> bool line_compact_used(const emg::Level<T> &L, int dir)
{
    if (g_line_compact < 0 || !(g_line_compact > 0 || (L.flags & emg::LEVEL_LINE_COMPACT))) return false;
    if (line_wide_used(dir, L.nx, L.ny, L.nz)) return false;
    bool any = false;
    for (int c = 0; c < 4; ++c) {
        const emg::LineClass lc = emg::line_class(dir, L.nx, L.ny, L.nz, c);
        if (lc.lines <= 0) continue;
        // decided on the plan of ONE source (a level that streams for one source streams for groups as well: more
        // right-hand sides never mean fewer lines per workgroup), so that a source sees the same records -- and gives
        // the same bits -- alone and in a batch
        // Also the three-phase kernel on lines of more than LINE_SHORT blocks (its T records only: its w records live
        // in LDS): the levels with 64- ... 128-block lines are bound by the same factor stream. Both plans must run the
        // same kind of kernel -- the streamed one rounds its w records, the three-phase one does not.
        // ... for ANY number of right-hand sides: a class with few lines runs the three-phase kernel with 4 or 8 lines per
        // workgroup for one source and the streamed kernel (16 lines per workgroup, records too large for LDS) for a
        // batch -- such a level keeps fp64 records for everybody (PI: the plan of a very large batch).
        const LinePlan P1 = line_plan<T>(lc, 1), PB = line_plan<T>(lc, L.batch), PI = line_plan<T>(lc, 1 << 10);
        auto ok = [](const LinePlan &P) {
            return P.kind == LK_STREAM || (P.kind == LK_COLOUR && !P.shortl && P.vmode >= 0 && P.vmode <= 2 && g_line_compact_colour);
        };
        if (!ok(P1) || !ok(PB) || !ok(PI) || P1.kind != PB.kind || P1.kind != PI.kind) return false;
        any = true;
    }
    return any;
}

template <class T, int DIR, int B>
void launch_stream_group(const emg::Level<T> &L, int c, const emg::LineClass &lc, const void *f, const double *lf, T *vec,
                         size_t vstride, int b0, int lpw, hipStream_t st, bool compact = false)
{
    // one source: the coupling entries through a second ring (option line_stream_lf, default 1)
    // (for groups of two the second ring leaves room for 8 rows per chunk only: y / z lines 0.83 -> 0.79-0.82 x per
    // source, x-lines 0.82 -> 0.93 x -- measured, not adopted)
    const bool lfr = B == 1 && (g_line_stream_lf != 0 || compact);
    int R = stream_rows(B, sizeof(T), lfr);
    const size_t smem = (size_t)2 * B * 2 * R * lpw * 5 * sizeof(T) + (lfr ? (size_t)2 * 2 * R * lpw * 8 * sizeof(double) : 0);
    // one source: four producer waves and unpaired stores (in a config-3 cycle six waves / paired x-line stores
    // measure the same to 0.5 %: tools/ab_cycle.py); groups: six producer waves, x-line stores in pairs
    constexpr int RD = B >= 2 ? 2 : emg::LINE_PAD;
    constexpr int NPROD = B >= 2 ? LS_PROD : 256;
    const void *kern = lfr ? (const void *)&k_line_stream<T, DIR, B, RD, NPROD, (B >= 2), (B == 1)>
                           : (const void *)&k_line_stream<T, DIR, B, RD, NPROD, (B >= 2), false>;
    int nprod = NPROD;
    size_t smem_c = smem;
    if constexpr (B >= 2) {
        if (compact) kern = (const void *)&k_line_stream<T, DIR, B, RD, NPROD, true, false, true>;
    }
    if constexpr (B == 1) {
        const bool rd8 = g_line_compact_rd == 8 || (g_line_compact_rd == 0 && DIR == 0);
        if (compact) {
            kern = rd8 ? (const void *)&k_line_stream<T, DIR, 1, 8, LS_PROD_COMPACT, false, true, true>
                       : (const void *)&k_line_stream<T, DIR, 1, RD, LS_PROD_COMPACT, false, true, true>;
            nprod = LS_PROD_COMPACT;
        }
        // rows per ring chunk of the compact kernel: 16 for x-lines (their producers read 16 consecutive blocks of a line as
        // one contiguous segment), 8 for y / z lines (same-box A/B at 256^3: x 0.709 -> 0.775, y 0.785 -> 0.764, z 0.780 ->
        // 0.766 ms per launch with 8) unless option line_stream_r names a value
        if (compact && DIR != 0 && g_line_stream_r == 0 && R > 8) {
            R = 8;
            smem_c = (size_t)2 * 2 * R * lpw * 5 * sizeof(T) + (size_t)2 * 2 * R * lpw * 8 * sizeof(double);
        }
    }
    (void)allow_lds(kern, 160 * 1024);
    T *v0 = vec + (size_t)b0 * vstride;
    size_t boff0 = (size_t)b0 * L.bstride;
    const unsigned nwg = cdiv(lc.lines, lpw);
    void *args[] = {(void *)&L, (void *)&c, (void *)&lc.cntp, (void *)&lc.cntq, (void *)&lc.n0p, (void *)&lpw, (void *)&R,
                    (void *)&f, (void *)&lf, (void *)&v0, (void *)&vstride, (void *)&boff0};
    (void)hipLaunchKernel(kern, dim3(nwg), dim3(128 + nprod), args, smem_c, st);
}

template <class T, int DIR>
int launch_line_colour(const emg::Level<T> &L, int c, const T *fac, const double *lfac, T *vec, hipStream_t st)
{
    const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
    if (lc.lines <= 0) return 0;
    const dim3 bb = d3(emg::lineblk_block());
    const dim3 bgp = d3(emg::lineblk_grid(lc, true));
    const T *f = fac + lc.fac_off;
    const double *lf = lfac + lc.lfac_off;
    // compact T records (line_compact_used): the class offset counts records, whatever their element type
    using FT = typename emg::compact_of<T>::type;
    const bool compact = line_compact_used<T>(L, DIR);
    const void *fv = compact ? (const void *)(reinterpret_cast<const FT *>(fac) + lc.fac_off) : (const void *)f;
    const dim3 qb = d3(emg::linequad_block());
    const emg::Dim3 q1 = emg::linequad_grid(lc);
    const dim3 qg2(q1.x, 2, 1);                      // x: 16 lines per wave, y: top / bottom half
    const size_t vstride = emg::line_vec_elems(DIR, L.nx, L.ny, L.nz);    // scratch of one right-hand side
    const size_t dummy_off = vstride - emg::LINE_DUMMY;
    if (line_wide_used(DIR, L.nx, L.ny, L.nz)) {
        const int nbthr = g_line_wide_bt == 256 || (g_line_wide_bt == 0 && wide_lpw(lc.n0, 256) > wide_lpw(lc.n0, LW_BLOCK_THREADS)) ? 256 : LW_BLOCK_THREADS;
        const int lpw = wide_lpw(lc.n0, nbthr);
        const size_t smem = ((size_t)2 * lpw * (lc.n0 + 1) + 2 * lpw) * LW_ROW * sizeof(T);
        const T *nf = fac + emg::line_fac_elems(DIR, L.nx, L.ny, L.nz) + lc.fac_off / 15 * 16;
        const dim3 grid(cdiv(lc.lines, lpw), L.batch);
        if (smem > (size_t)64 * 1024) {
            (void)allow_lds((const void *)&k_line_wide<T, DIR, true>, smem);
            (void)allow_lds((const void *)&k_line_wide<T, DIR, false>, smem);
        }
        if (L.batch > 1)
            hipLaunchKernelGGL((k_line_wide<T, DIR, true>), grid, dim3(nbthr + 64), smem, st, L, c, lc.cntp, lc.cntq, lpw, f, lf, nf, nbthr);
        else
            hipLaunchKernelGGL((k_line_wide<T, DIR, false>), grid, dim3(nbthr + 64), smem, st, L, c, lc.cntp, lc.cntq, lpw, f, lf, nf, nbthr);
        return 0;
    }
    const LinePlan P = line_plan<T>(lc, L.batch);
    if (P.kind == LK_STREAM) {
        // groups of at most LSB_MAX right-hand sides, as even as possible (8 -> 4 + 4, 6 -> 3 + 3, 5 -> 3 + 2)
        const int ng = cdiv(L.batch, LSB_MAX);
        int b0 = 0;
        for (int g = 0; g < ng; ++g) {
            const int gs = L.batch / ng + (g < L.batch % ng ? 1 : 0);
            if (gs == 4) launch_stream_group<T, DIR, 4>(L, c, lc, fv, lf, vec, vstride, b0, P.lpw, st, compact);
            else if (gs == 3) launch_stream_group<T, DIR, 3>(L, c, lc, fv, lf, vec, vstride, b0, P.lpw, st, compact);
            else if (gs == 2) launch_stream_group<T, DIR, 2>(L, c, lc, fv, lf, vec, vstride, b0, P.lpw, st, compact);
            else launch_stream_group<T, DIR, 1>(L, c, lc, fv, lf, vec, vstride, b0, P.lpw, st, compact);
            b0 += gs;
        }
        return 0;
    }
    if (P.kind == LK_COLOUR) {
        const int lpw = P.lpw;
        const unsigned nwg = cdiv(lc.lines, lpw);
        const size_t lds_cu = 160 * 1024;
        constexpr int P4 = emg::LINE_PAD, P2 = emg::LINE_PAD_SHORT;
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, 1, false, P4>), lds_cu);
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, 2, false, P4>), lds_cu);
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, 1, true, P4>), lds_cu);
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, 2, true, P4>), lds_cu);
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, 3, false, P4>), lds_cu);
#define LC_LAUNCH(VM, QDV)                                                                                               \
    do {                                                                                                                 \
        if (L.batch > 1 || (g_line_occ2 && VM == 0))                                                                     \
            hipLaunchKernelGGL((k_line_colour<T, DIR, VM, true, QDV>), dim3(nwg, L.batch), dim3(LC_THREADS), P.smem, st, L, \
                               c, lc.cntp, lc.cntq, lc.n0p, lpw, f, lf, vec, vec + dummy_off, vstride);                  \
        else                                                                                                             \
            hipLaunchKernelGGL((k_line_colour<T, DIR, VM, false, QDV>), dim3(nwg), dim3(LC_THREADS), P.smem, st, L, c,    \
                               lc.cntp, lc.cntq, lc.n0p, lpw, f, lf, vec, vec + dummy_off,                               \
                               (VM == 0 && (g_line_debug & 1)) ? ~(size_t)0 : vstride);                                  \
    } while (0)
        // compact T records (line_compact_used): the same kernel reading single-precision factor rows
#define LC_LAUNCH_C(VM)                                                                                                  \
    do {                                                                                                                 \
        const FT *fc = reinterpret_cast<const FT *>(fv);                                                                 \
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, VM, false, P4, FT>), lds_cu);              \
        (void)allow_lds(reinterpret_cast<const void *>(&k_line_colour<T, DIR, VM, true, P4, FT>), lds_cu);               \
        if (L.batch > 1)                                                                                                 \
            hipLaunchKernelGGL((k_line_colour<T, DIR, VM, true, P4, FT>), dim3(nwg, L.batch), dim3(LC_THREADS), P.smem, st, \
                               L, c, lc.cntp, lc.cntq, lc.n0p, lpw, fc, lf, vec, vec + dummy_off, vstride);              \
        else                                                                                                             \
            hipLaunchKernelGGL((k_line_colour<T, DIR, VM, false, P4, FT>), dim3(nwg), dim3(LC_THREADS), P.smem, st, L, c, \
                               lc.cntp, lc.cntq, lc.n0p, lpw, fc, lf, vec, vec + dummy_off, vstride);                    \
    } while (0)
        if (compact) {
            if (P.shortl || P.vmode > 2 || g_line_occ2 || (g_line_debug & 1)) {
                return fail(EMG3D_ERR_BADARG, "gauss_seidel: compact line factors under options that cannot read them");
            }
            if (P.vmode == 1) LC_LAUNCH_C(1);
            else if (P.vmode == 2) LC_LAUNCH_C(2);
            else LC_LAUNCH_C(0);
            return 0;
        }
        if (P.shortl) {
            if (P.vmode == 1) LC_LAUNCH(1, P2);
            else LC_LAUNCH(0, P2);
        } else if (P.vmode == 1) LC_LAUNCH(1, P4);
        else if (P.vmode == 2) LC_LAUNCH(2, P4);
        else if (P.vmode == 3)
            hipLaunchKernelGGL((k_line_colour<T, DIR, 3, false, P4>), dim3(nwg), dim3(LC_THREADS), P.smem, st, L, c,
                               lc.cntp, lc.cntq, lc.n0p, lpw, f, lf, vec, vec + dummy_off, vstride);
        else LC_LAUNCH(0, P4);
#undef LC_LAUNCH
#undef LC_LAUNCH_C
        return 0;
    }
    // separate launches (option line_fuse = 0): the right-hand sides one after the other
    if (compact) {
        return fail(EMG3D_ERR_BADARG, "gauss_seidel: compact line factors need the fused line kernels (line_fuse)");
    }
    for (int b = 0; b < L.batch; ++b) {
        const emg::Level<T> Lb = emg::source_level(L, b);
        T *const vb = vec + (size_t)b * vstride;
        if (DIR == 0)
            hipLaunchKernelGGL(k_line_rhs_xt<T>, dim3(cdiv(lc.cntp, 16), lc.cntq, cdiv(lc.n0p, 16)), dim3(256), 0, st,
                               Lb, c, lc.cntp, lc.cntq, lc.n0p, vb);
        else
            hipLaunchKernelGGL((k_line_rhs<T, DIR>), bgp, bb, 0, st, Lb, c, lc.cntp, lc.cntq, vb);
        if (emg::line_pad(lc.n0) == emg::LINE_PAD_SHORT) {
            hipLaunchKernelGGL((k_line_forward<T, emg::LINE_PAD_SHORT>), qg2, qb, 0, st, lc.n0, lc.n0p, lc.lines, f, lf, vb,
                               vb + dummy_off);
            hipLaunchKernelGGL((k_line_backward<T, DIR, emg::LINE_PAD_SHORT>), qg2, qb, 0, st, Lb, c, lc.cntp, lc.cntq, lc.n0p,
                               f, lf, (const T *)vb, vb + dummy_off);
        } else {
            hipLaunchKernelGGL((k_line_forward<T, emg::LINE_PAD>), qg2, qb, 0, st, lc.n0, lc.n0p, lc.lines, f, lf, vb,
                               vb + dummy_off);
            hipLaunchKernelGGL((k_line_backward<T, DIR, emg::LINE_PAD>), qg2, qb, 0, st, Lb, c, lc.cntp, lc.cntq, lc.n0p, f,
                               lf, (const T *)vb, vb + dummy_off);
        }
    }
    return 0;
}

template <class T>
int launch_gs(const emg3d_level *lv, int lr, int nu, const void *fac, const double *lfac, void *scratch,
              size_t scratch_bytes, hipStream_t st)
{
    emg::Level<T> L = to_level<T>(lv);
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    if (nx < 2 || ny < 2 || nz < 2) return fail(EMG3D_ERR_BADARG, "gauss_seidel: need >= 2 cells per direction");
    if (lr != 0) {
        if (!fac || !lfac) return fail(EMG3D_ERR_BADARG, "gauss_seidel: line factors missing (emg3d_dev_line_setup)");
        if (!scratch || scratch_bytes < (size_t)L.batch * emg3d_gs_scratch_bytes(lr, nx, ny, nz, lv->is_complex))
            return fail(EMG3D_ERR_SCRATCH, "gauss_seidel: scratch buffer too small");
    }
    const bool tiled = emg::point_tiled(nx, ny, nz, g_point_tile_min);
    // optional eta edge sums (emg3d_dev_point_setup): edge-shaped for the plain kernel, tile-major
    // for the tiled one
    const T *pst = (lr == 0 && !tiled) ? (const T *)fac : nullptr;
    const hipStream_t st_ = st;
    int iback = 0;
    for (int it = 0; it < nu; ++it) {
        iback = 1 - iback;   // first sweep backward (reference emg3d/core.py:301,311)
        if (lr == 0 && tiled) {
            using TB = emg::PointTile;
            using E = emg::EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
            const size_t smem = E::LDS_BYTES;
            // eta sums: tile-major buffer (stored halves when the level's eta are purely imaginary
            // or the field is real), or formed on the fly when fac == NULL
            const int st = !fac ? 0 : pst_mode<T>(L.flags);
            int pf = (g_point_prefetch >= 0 && g_point_prefetch <= 3) ? g_point_prefetch : 0;
            if (st == 0 && pf == 2) pf = 1;      // 24 eta loads per node in flight twice do not fit the registers
#define PT_ROW(B, PF_)                                                                                         \
    {(const void *)&k_gs_point_tile<T, TB, 0, B, PF_>, (const void *)&k_gs_point_tile<T, TB, 2, B, PF_>,          \
     (const void *)&k_gs_point_tile<T, TB, 3, B, PF_>}
            const void *kfn[2][4][3] = {{PT_ROW(false, 0), PT_ROW(false, 1), PT_ROW(false, 2), PT_ROW(false, 3)},
                                        {PT_ROW(true, 0), PT_ROW(true, 1), PT_ROW(true, 2), PT_ROW(true, 3)}};
#undef PT_ROW
            const void *kern = kfn[L.batch > 1 ? 1 : 0][pf][st == 0 ? 0 : st - 1];
            // single-precision eta sums: the instantiations without software prefetch (the default) only
            if (st == emg::PST_HALF_F32)
                kern = L.batch > 1 ? (const void *)&k_gs_point_tile<T, TB, emg::PST_HALF_F32, true, 0>
                                   : (const void *)&k_gs_point_tile<T, TB, emg::PST_HALF_F32, false, 0>;
            if (st == emg::PST_FULL_F32)
                kern = L.batch > 1 ? (const void *)&k_gs_point_tile<T, TB, emg::PST_FULL_F32, true, 0>
                                   : (const void *)&k_gs_point_tile<T, TB, emg::PST_FULL_F32, false, 0>;
            HIP_TRY(allow_lds(kern, smem));
            // A sweep ends with the pair of tile colours the next sweep (opposite direction) starts
            // with, and nothing else runs in between: those tiles do the node colours of BOTH sweeps
            // on one LDS copy -- one load / store of the tile instead of two (1/8 of the traffic of
            // two sweeps) -- minus the node colour that would merely be repeated (skip_repeat).
            const bool fuse_next = g_tile_fuse && it + 1 < nu;
            for (int p = (g_tile_fuse && it > 0) ? 1 : 0; p < 4; ++p) {   // two complementary tile colours per launch
                emg::TilePair P = emg::tile_pair<TB>(nx, ny, nz, iback, p);
                const int gz = P.gz[0] + P.gz[1];
                if (gz <= 0) continue;
                int colours = emg::sweep_colours_packed(iback), nsteps = 4;
                if (p == 3 && fuse_next) {
                    const int next = emg::sweep_colours_packed(1 - iback);
                    // (the first node colour of the next sweep repeats this sweep's last one only under the
                    // mirrored rule, option point_order = 0)
                    const bool repeats = emg::sweep_colour(1 - iback, 0) == emg::sweep_colour(iback, 3);
                    if (g_skip_repeat && repeats) { colours |= (next >> 2) << 8; nsteps = 7; }
                    else { colours |= next << 8; nsteps = 8; }
                }
                const dim3 gb(P.gx[0] > P.gx[1] ? P.gx[0] : P.gx[1], P.gy[0] > P.gy[1] ? P.gy[0] : P.gy[1], gz * L.batch);
                const dim3 tb(TB::THREADS);
                const void *pstv = fac;
                void *args[] = {(void *)&L, (void *)&pstv, (void *)&P, (void *)&colours, (void *)&nsteps};
                HIP_TRY(hipLaunchKernel(kern, gb, tb, args, smem, st_));
            }
            continue;
        }
        if (lr == 0 && g_point_small > 0 && L.batch == 1 && 4 * nu <= 32 &&
            (long long)(nx - 1) * (ny - 1) * (nz - 1) <= g_point_small) {
            // a level for one workgroup: every pass of every sweep in ONE launch (k_gs_point_small)
            if (it > 0) continue;                      // (launched with the first sweep)
            unsigned long long passes = 0;
            int npass = 0, ib = 0;
            for (int s2 = 0; s2 < nu; ++s2) {
                ib = 1 - ib;
                for (int cc = 0; cc < 4; ++cc) {
                    if (g_skip_repeat && s2 > 0 && cc == 0 && emg::sweep_colour(ib, 0) == emg::sweep_colour(1 - ib, 3)) continue;
                    passes |= (unsigned long long)emg::sweep_colour(ib, cc) << (2 * npass);
                    ++npass;
                }
            }
            hipLaunchKernelGGL(k_gs_point_small<T>, dim3(1), dim3(256), 0, st, L, pst, passes, npass);
            continue;
        }
        if (lr == 0) {
            // (same redundancy for the node classes; only the unslabbed schedule is a plain
            // sequence of whole colour passes)
            const bool unslabbed = g_point_slab <= 0 || g_point_slab >= nz - 1;
            const int skip = (g_skip_repeat && it > 0 && unslabbed && emg::sweep_colour(iback, 0) == emg::sweep_colour(1 - iback, 3))
                                 ? emg::sweep_colour(iback, 0) : -1;
            emg::gs_point_schedule(nz, g_point_slab, iback, [&](int c, int iz0, int izn) {
                if (c == skip) return;
                const emg::Dim3 g = emg::gs_point_grid(nx, ny, izn);
                if (g.x > 0 && g.y > 0 && g.z > 0)
                    hipLaunchKernelGGL(k_gs_point<T>, dim3(g.x, g.y, g.z * L.batch), d3(emg::gs_point_block()), 0, st, L,
                                       pst, c, iz0, g.z);
            });
            continue;
        }
        for (int cc = 0; cc < 4; ++cc) {
            const int c = emg::line_sweep_colour(g_line_order, it, cc);
            // A sweep ends with the colour class the next one starts with (both rules of launch.h:
            // line_sweep_colour). A line solve depends on the edges NOT on the line only, and lines of
            // one class do not see each other: solving the class again right away reproduces the same
            // values bit by bit. (The reference's sequential sweeps have the same redundant first line.)
            if (g_skip_repeat && cc == 0 && emg::line_pass_repeats(g_line_order, it)) continue;
            int rc;
            if (lr == 1) rc = launch_line_colour<T, 0>(L, c, (const T *)fac, lfac, (T *)scratch, st);
            else if (lr == 2) rc = launch_line_colour<T, 1>(L, c, (const T *)fac, lfac, (T *)scratch, st);
            else rc = launch_line_colour<T, 2>(L, c, (const T *)fac, lfac, (T *)scratch, st);
            if (rc != 0) return rc;
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T, int DIR>
void launch_line_setup_dir(const emg::Level<T> &L, T *fac, double *lfac, hipStream_t st)
{
    SetupClasses S;
    int gx = 0, gy = 0;
    for (int c = 0; c < 4; ++c) {
        const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
        S.cntp[c] = lc.lines > 0 ? lc.cntp : 0; S.cntq[c] = lc.lines > 0 ? lc.cntq : 0;
        S.fac_off[c] = lc.fac_off; S.lfac_off[c] = lc.lfac_off;
        if (lc.lines > 0) {
            const emg::Dim3 g = emg::line_grid(lc);
            gx = g.x > gx ? g.x : gx; gy = g.y > gy ? g.y : gy;
        }
    }
    if (line_compact_used<T>(L, DIR)) {
        // the T records in compact form (single precision, rounded as they are stored); no N records on such levels
        using FT = typename emg::compact_of<T>::type;
        if (gx > 0 && gy > 0)
            hipLaunchKernelGGL((k_line_setup<T, DIR, FT>), dim3(gx, gy, 4), dim3(128), 0, st, L, S, reinterpret_cast<FT *>(fac), lfac);
        return;
    }
    if (gx > 0 && gy > 0) hipLaunchKernelGGL((k_line_setup<T, DIR>), dim3(gx, gy, 4), dim3(128), 0, st, L, S, fac, lfac);
    // the N records of the wide form, behind the T records (one per block record, from its T and coupling entries)
    const size_t nrec = emg::line_records(DIR, L.nx, L.ny, L.nz);
    if (gx > 0 && gy > 0 && emg::line_nfac_elems(DIR, L.nx, L.ny, L.nz) > 0)
        hipLaunchKernelGGL(k_line_wide_setup<T>, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, (const T *)fac,
                           (const double *)lfac, fac + emg::line_fac_elems(DIR, L.nx, L.ny, L.nz), nrec);
}

template <class T>
int launch_line_setup(const emg3d_level *lv, int lr, void *fac, double *lfac, hipStream_t st)
{
    emg::Level<T> L = to_level<T>(lv);
    if (L.nx < 2 || L.ny < 2 || L.nz < 2) return fail(EMG3D_ERR_BADARG, "line_setup: need >= 2 cells per direction");
    if (lr == 1) launch_line_setup_dir<T, 0>(L, (T *)fac, lfac, st);
    else if (lr == 2) launch_line_setup_dir<T, 1>(L, (T *)fac, lfac, st);
    else launch_line_setup_dir<T, 2>(L, (T *)fac, lfac, st);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
int launch_residual(const emg3d_level *lv, void *rx, void *ry, void *rz, double *ws, size_t ws_len,
                    double *sumsq, hipStream_t st)
{
    emg::Level<T> L = to_level<T>(lv);
    const dim3 block = d3(emg::cell_block());
    dim3 grid = d3(emg::cell_grid(L.nx + 1, L.ny + 1, L.nz + 1));
    // planes per workgroup: 8 where that still leaves every CU several workgroups, else 1. Decided
    // per right-hand side: the blocking fixes the summation order of the norm, and a source must get
    // the same bits whether it is solved alone or in a batch
    const int zb = (size_t)grid.x * grid.y * (grid.z / g_residual_zb) >= 8u * (unsigned)compute_units() ? g_residual_zb : 1;
    grid.z = cdiv((int)grid.z, zb);
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;       // per right-hand side
    if (sumsq && (ws == nullptr || ws_len < nblk * L.batch)) return fail(EMG3D_ERR_SCRATCH, "residual: workspace too small");
    if (zb > 1 && g_residual_roll)
        hipLaunchKernelGGL((k_residual<T, true>), dim3(grid.x, grid.y, grid.z * L.batch), block, 0, st, L, (T *)rx, (T *)ry,
                           (T *)rz, sumsq ? ws : nullptr, (int)grid.z, zb);
    else
        hipLaunchKernelGGL((k_residual<T, false>), dim3(grid.x, grid.y, grid.z * L.batch), block, 0, st, L, (T *)rx, (T *)ry,
                           (T *)rz, sumsq ? ws : nullptr, (int)grid.z, zb);
    if (sumsq) hipLaunchKernelGGL(k_reduce_sum, dim3(L.batch), dim3(256), 0, st, ws, (int)nblk, sumsq);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
int launch_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry, const void *rz,
                    const double *const w[9], int nx, int ny, int nz, int sc_dir, hipStream_t st, int batch = 1,
                    size_t fstride = 0, size_t cstride = 0, void *cex = nullptr, void *cey = nullptr, void *cez = nullptr)
{
    const ScDirs f = sc_flags(sc_dir);
    if ((f.cx && nx % 2) || (f.cy && ny % 2) || (f.cz && nz % 2))
        return fail(EMG3D_ERR_BADARG, "restrict: odd cell count in a coarsened direction");
    emg::Restrict<T> R = emg::make_restrict<T>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir);
    R.batch = batch > 1 ? batch : 1; R.fstride = fstride; R.cstride = cstride;
    R.cex = (T *)cex; R.cey = (T *)cey; R.cez = (T *)cez;
    const emg::Dim3 g = emg::cell_grid(R.cnxn, R.cnyn, R.cnzn);
    hipLaunchKernelGGL(k_restrict<T>, dim3(g.x, g.y, g.z * R.batch), d3(emg::cell_block()), 0, st, R);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
int launch_prolong(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                   const int32_t *ilx, const int32_t *ily, const int32_t *ilz, const double *wx,
                   const double *wy, const double *wz, int nx, int ny, int nz, int sc_dir, hipStream_t st,
                   int batch = 1, size_t fstride = 0, size_t cstride = 0)
{
    emg::Prolong<T> P =
        emg::make_prolong<T>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz, sc_dir);
    P.batch = batch > 1 ? batch : 1; P.fstride = fstride; P.cstride = cstride;
    const emg::Dim3 g = emg::cell_grid(nx + 1, ny + 1, nz + 1);
    hipLaunchKernelGGL(k_prolong<T>, dim3(g.x, g.y, g.z * P.batch), d3(emg::cell_block()), 0, st, P);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
int launch_restrict_param(void *out, const void *in, int nx, int ny, int nz, int sc_dir, hipStream_t st)
{
    const ScDirs f = sc_flags(sc_dir);
    const int fx = f.cx ? 2 : 1, fy = f.cy ? 2 : 1, fz = f.cz ? 2 : 1;
    const int cnx = nx / fx, cny = ny / fy, cnz = nz / fz;
    const dim3 block = d3(emg::cell_block());
    const dim3 grid = d3(emg::cell_grid(cnx, cny, cnz));
    hipLaunchKernelGGL(k_restrict_param<T>, grid, block, 0, st, (T *)out, (const T *)in, nx, ny, cnx, cny, fx, fy, fz);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------- host-flavour plumbing ----

// RAII device buffer filled from / copied back to a host pointer.
struct DevBuf {
    void *d = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (d) (void)hipFree(d); }
    hipError_t up(const void *h, size_t n)
    {
        bytes = n;
        hipError_t e = hipMalloc(&d, n ? n : 1);
        if (e != hipSuccess) return e;
        return n ? hipMemcpy(d, h, n, hipMemcpyHostToDevice) : hipSuccess;
    }
    hipError_t alloc(size_t n)
    {
        bytes = n;
        return hipMalloc(&d, n ? n : 1);
    }
    hipError_t down(void *h) const { return bytes ? hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost) : hipSuccess; }
};

struct Sizes {
    size_t nex, ney, nez, ncc, esz;
    Sizes(int nx, int ny, int nz, int is_complex)
    {
        nex = (size_t)nx * (ny + 1) * (nz + 1);
        ney = (size_t)(nx + 1) * ny * (nz + 1);
        nez = (size_t)(nx + 1) * (ny + 1) * nz;
        ncc = (size_t)nx * ny * nz;
        esz = is_complex ? 16 : 8;
    }
};

// Upload eta_x/eta_y/eta_z preserving aliasing (one device copy per distinct host pointer).
struct EtaUpload {
    DevBuf b[3];
    const void *dptr[3] = {nullptr, nullptr, nullptr};
    hipError_t up(const void *hx, const void *hy, const void *hz, size_t bytes)
    {
        const void *h[3] = {hx, hy, hz};
        for (int i = 0; i < 3; ++i) {
            int alias = -1;
            for (int j = 0; j < i; ++j)
                if (h[j] == h[i]) { alias = j; break; }
            if (alias >= 0) { dptr[i] = dptr[alias]; continue; }
            hipError_t e = b[i].up(h[i], bytes);
            if (e != hipSuccess) return e;
            dptr[i] = b[i].d;
        }
        return hipSuccess;
    }
};

hipError_t upload_inverse(DevBuf &b, const double *h, int n)
{
    double *tmp = new double[n > 0 ? n : 1];
    for (int i = 0; i < n; ++i) tmp[i] = 1.0 / h[i];
    hipError_t e = b.up(tmp, sizeof(double) * (size_t)n);
    delete[] tmp;
    return e;
}

}  // namespace

// ================================================================================ C ABI ==
extern "C" {

int emg3d_version(void) { return EMG3D_AMD_VERSION; }
const char *emg3d_last_error(void) { return g_err.c_str(); }

// run-time options: name -> variable (documented where the variables are declared)
struct OptionEntry { const char *name; int *value; };
static const OptionEntry g_options[] = {
    {"point_slab", &g_point_slab},       {"point_tile_min", &g_point_tile_min}, {"line_fuse", &g_line_fuse},
    {"line_fuse_max", &g_line_fuse_max}, {"skip_repeat", &g_skip_repeat},       {"tile_fuse", &g_tile_fuse},
    {"line_lds", &g_line_lds},           {"point_prefetch", &g_point_prefetch}, {"residual_zb", &g_residual_zb},
    {"line_occ2", &g_line_occ2},         {"point_small", &g_point_small},       {"line_lpw", &g_line_lpw},
    {"line_debug", &g_line_debug},       {"line_stream", &g_line_stream},       {"line_stream_r", &g_line_stream_r},
    {"line_order", &g_line_order},       {"point_order", &emg::point_order_ref()},
    {"line_stream_bmin", &g_line_stream_bmin}, {"line_stream_lf", &g_line_stream_lf}, {"residual_roll", &g_residual_roll},
    {"line_wide", &g_line_wide},         {"line_wide_bt", &g_line_wide_bt},     {"line_compact", &g_line_compact},
    {"line_compact_rd", &g_line_compact_rd}, {"point_compact", &g_point_compact},
    {"line_compact_colour", &g_line_compact_colour},
};
constexpr int N_OPTIONS = sizeof(g_options) / sizeof(g_options[0]);
static int g_options_generation = 0;      // bumped whenever an option changes its value

int emg3d_option_count(void) { return N_OPTIONS; }
const char *emg3d_option_name(int i) { return (i >= 0 && i < N_OPTIONS) ? g_options[i].name : nullptr; }

int emg3d_set_option(const char *name, int value)
{
    if (!name) return fail(EMG3D_ERR_BADARG, "set_option: null name");
    if (!std::strcmp(name, "residual_zb") && value < 1) value = 1;
    if (!std::strcmp(name, "line_lpw") && value != 0 && value != 4 && value != 8 && value != 16 && value != 32)
        return fail(EMG3D_ERR_BADARG, "line_lpw: 0, 4, 8, 16 or 32");
    // line_debug produces WRONG fields by design (timing experiments): only with the environment's consent
    if (!std::strcmp(name, "line_debug") && value != 0 && !std::getenv("EMG3D_AMD_ALLOW_DEBUG"))
        return fail(EMG3D_ERR_BADARG, "line_debug: wrong results by design; set EMG3D_AMD_ALLOW_DEBUG=1 to use it");
    // rows per chunk of k_line_stream's ring: whole register rings of LINE_PAD blocks (a chunk that ends inside a
    // ring pass would be read past its end), and two chunks x two halves x 16 lines must fit the LDS of a CU
    if (!std::strcmp(name, "line_stream_r") && value != 0 && (value < 4 || value > 32 || value % emg::LINE_PAD != 0))
        return fail(EMG3D_ERR_BADARG, "line_stream_r: 0 (= 16) or a multiple of 4 in 4..32");
    // (lines longer than WIDE_N0_MAX hold no N records: a larger value would be a silent no-op)
    if (!std::strcmp(name, "line_wide") && (value < 0 || value > emg::WIDE_N0_MAX)) return fail(EMG3D_ERR_BADARG, "line_wide: 0 .. 64");
    if (!std::strcmp(name, "line_wide_bt") && value != 0 && value != 192 && value != 256) return fail(EMG3D_ERR_BADARG, "line_wide_bt: 0, 192 or 256");
    if (!std::strcmp(name, "line_compact_rd") && value != 0 && value != 4 && value != 8) return fail(EMG3D_ERR_BADARG, "line_compact_rd: 0, 4 or 8");
    if (!std::strcmp(name, "point_compact") && (value < -1 || value > 1)) return fail(EMG3D_ERR_BADARG, "point_compact: -1, 0 or 1");
    if (!std::strcmp(name, "line_compact") && (value < -1 || value > 1)) return fail(EMG3D_ERR_BADARG, "line_compact: -1, 0 or 1");
    if (!std::strcmp(name, "line_order") && (value < 0 || value > 2)) return fail(EMG3D_ERR_BADARG, "line_order: 0, 1 or 2");
    if (!std::strcmp(name, "point_order") && (value < 0 || value > 1)) return fail(EMG3D_ERR_BADARG, "point_order: 0 or 1");
    for (const OptionEntry &o : g_options)
        if (!std::strcmp(name, o.name)) {
            if (*o.value != value) ++g_options_generation;
            *o.value = value;
            return 0;
        }
    return fail(EMG3D_ERR_BADARG, "set_option: unknown option");
}

int emg3d_options_generation(void) { return g_options_generation; }

#ifdef EMG_WIDE_STAMPS
int emg3d_debug_wide_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wide_stamps), sizeof(unsigned long long) * 32);
}
#endif

const char *emg3d_line_kernel_name(int lr, int nx, int ny, int nz, int is_complex, int batch)
{
    if (lr < 1 || lr > 3 || nx < 2 || ny < 2 || nz < 2) return "";
    // the largest colour class (odd, odd) decides, as it does for the scratch size
    const emg::LineClass lc = emg::line_class(lr - 1, nx, ny, nz, 3);
    if (lc.lines <= 0) return "";
    if (line_wide_used(lr - 1, nx, ny, nz)) return "k_line_wide";
    const LinePlan P = is_complex ? line_plan<cplx>(lc, batch > 1 ? batch : 1) : line_plan<double>(lc, batch > 1 ? batch : 1);
    switch (P.kind) {
    case LK_STREAM: return "k_line_stream";
    case LK_COLOUR: return "k_line_colour";
    default: return "k_line_rhs+k_line_forward+k_line_backward";
    }
}

int emg3d_get_option(const char *name)
{
    if (name)
        for (const OptionEntry &o : g_options)
            if (!std::strcmp(name, o.name)) return *o.value;
    return -1;
}

int emg3d_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t emg3d_gs_scratch_bytes(int lr, int nx, int ny, int nz, int is_complex)
{
    if (lr < 1 || lr > 3) return 0;
    return emg::line_vec_elems(lr - 1, nx, ny, nz) * (is_complex ? 16 : 8);
}

size_t emg3d_line_fac_bytes(int lr, int nx, int ny, int nz, int is_complex)
{
    if (lr < 1 || lr > 3) return 0;
    return (emg::line_fac_elems(lr - 1, nx, ny, nz) + emg::line_nfac_elems(lr - 1, nx, ny, nz)) * (is_complex ? 16 : 8);
}

size_t emg3d_line_fac_bytes_lv(const emg3d_level *lv, int lr)
{
    if (!lv || lr < 1 || lr > 3) return 0;
    const bool compact = lv->is_complex ? line_compact_used<cplx>(to_level<cplx>(lv), lr - 1)
                                        : line_compact_used<double>(to_level<double>(lv), lr - 1);
    if (compact) return emg::line_fac_elems(lr - 1, lv->nx, lv->ny, lv->nz) * (lv->is_complex ? 8 : 4);
    return emg3d_line_fac_bytes(lr, lv->nx, lv->ny, lv->nz, lv->is_complex);
}

int emg3d_line_compact_used(const emg3d_level *lv, int lr)
{
    if (!lv || lr < 1 || lr > 3) return 0;
    return lv->is_complex ? (int)line_compact_used<cplx>(to_level<cplx>(lv), lr - 1)
                          : (int)line_compact_used<double>(to_level<double>(lv), lr - 1);
}

size_t emg3d_line_lfac_bytes(int lr, int nx, int ny, int nz)
{
    if (lr < 1 || lr > 3) return 0;
    return emg::line_lfac_elems(lr - 1, nx, ny, nz) * 8;
}

int emg3d_dev_line_setup(const emg3d_level *lv, int lr, void *fac, double *lfac, void *stream)
{
    if (!lv || lr < 1 || lr > 3 || !fac || !lfac) return fail(EMG3D_ERR_BADARG, "line_setup: bad argument");
    return lv->is_complex ? launch_line_setup<cplx>(lv, lr, fac, lfac, (hipStream_t)stream)
                          : launch_line_setup<double>(lv, lr, fac, lfac, (hipStream_t)stream);
}

static size_t point_fac_bytes(int nx, int ny, int nz, int is_complex, int flags)
{
    using TB = emg::PointTile;
    if (emg::point_tiled(nx, ny, nz, g_point_tile_min)) {
        const int mode = is_complex ? pst_mode<cplx>(flags) : pst_mode<double>(flags);
        return emg::tile_pst_elems(nx, ny, nz, TB::BX, TB::BY, TB::BZ) * emg::tile_pst_bytes(mode);
    }
    const Sizes S(nx, ny, nz, is_complex);
    return (S.nex + S.ney + S.nez) * S.esz;
}
size_t emg3d_point_fac_bytes(int nx, int ny, int nz, int is_complex)
{
    return point_fac_bytes(nx, ny, nz, is_complex, 0);      // flags = 0: the larger layout
}
int emg3d_point_compact_used(const emg3d_level *lv)
{
    if (!lv || !emg::point_tiled(lv->nx, lv->ny, lv->nz, g_point_tile_min)) return 0;
    const int mode = lv->is_complex ? pst_mode<cplx>(lv->flags) : pst_mode<double>(lv->flags);
    return mode == emg::PST_HALF_F32 || mode == emg::PST_FULL_F32;
}
size_t emg3d_point_fac_bytes_lv(const emg3d_level *lv)
{
    return lv ? point_fac_bytes(lv->nx, lv->ny, lv->nz, lv->is_complex, lv->flags) : 0;
}

int emg3d_dev_point_setup(const emg3d_level *lv, void *fac, void *stream)
{
    if (!lv || !fac) return fail(EMG3D_ERR_BADARG, "point_setup: bad argument");
    const hipStream_t st = (hipStream_t)stream;
    if (emg::point_tiled(lv->nx, lv->ny, lv->nz, g_point_tile_min)) {
        using TB = emg::PointTile;
        const emg::TileCount n = emg::tile_count<TB>(lv->nx, lv->ny, lv->nz);
        const dim3 g(n.x, n.y, n.z), b(TB::THREADS);
        const int mode = lv->is_complex ? pst_mode<cplx>(lv->flags) : pst_mode<double>(lv->flags);
        if (!lv->is_complex && mode == emg::PST_HALF_F32)
            hipLaunchKernelGGL((k_point_setup_tile<double, TB, emg::PST_HALF_F32>), g, b, 0, st, to_level<double>(lv), fac);
        else if (!lv->is_complex)
            hipLaunchKernelGGL((k_point_setup_tile<double, TB, emg::PST_HALF>), g, b, 0, st, to_level<double>(lv), fac);
        else if (mode == emg::PST_HALF_F32)
            hipLaunchKernelGGL((k_point_setup_tile<cplx, TB, emg::PST_HALF_F32>), g, b, 0, st, to_level<cplx>(lv), fac);
        else if (mode == emg::PST_HALF)
            hipLaunchKernelGGL((k_point_setup_tile<cplx, TB, emg::PST_HALF>), g, b, 0, st, to_level<cplx>(lv), fac);
        else if (mode == emg::PST_FULL_F32)
            hipLaunchKernelGGL((k_point_setup_tile<cplx, TB, emg::PST_FULL_F32>), g, b, 0, st, to_level<cplx>(lv), fac);
        else
            hipLaunchKernelGGL((k_point_setup_tile<cplx, TB, emg::PST_FULL>), g, b, 0, st, to_level<cplx>(lv), fac);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    HIP_TRY(hipMemsetAsync(fac, 0, emg3d_point_fac_bytes_lv(lv), st));
    const dim3 g = d3(emg::cell_grid(lv->nx + 1, lv->ny + 1, lv->nz + 1)), b = d3(emg::cell_block());
    if (lv->is_complex) hipLaunchKernelGGL(k_point_setup<cplx>, g, b, 0, st, to_level<cplx>(lv), (cplx *)fac);
    else hipLaunchKernelGGL(k_point_setup<double>, g, b, 0, st, to_level<double>(lv), (double *)fac);
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_eta_is_imaginary(const emg3d_level *lv, int *result, void *stream)
{
    if (!lv || !result) return fail(EMG3D_ERR_BADARG, "eta_is_imaginary: bad argument");
    *result = 0;
    if (!lv->is_complex) return 0;
    const hipStream_t st = (hipStream_t)stream;
    int *dflag = nullptr;
    HIP_TRY(hipMalloc((void **)&dflag, sizeof(int)));
    hipError_t e = hipMemsetAsync(dflag, 0, sizeof(int), st);
    const size_t n = (size_t)lv->nx * lv->ny * lv->nz;
    const void *done[3] = {nullptr, nullptr, nullptr};
    const void *eta[3] = {lv->eta_x, lv->eta_y, lv->eta_z};
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        if (eta[i] == done[0] || eta[i] == done[1]) continue;       // aliased arrays once
        done[i] = eta[i];
        hipLaunchKernelGGL(k_any_real_part, dim3(1024), dim3(256), 0, st, (const cplx *)eta[i], n, dflag);
        e = hipGetLastError();
    }
    int h = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, dflag, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dflag);
    if (e != hipSuccess) return hipfail(e, "eta_is_imaginary");
    *result = h == 0;
    return 0;
}

int emg3d_dev_gauss_seidel(const emg3d_level *lv, int lr, int nu, const void *fac, const double *lfac,
                           void *scratch, size_t scratch_bytes, void *stream)
{
    if (!lv || lr < 0 || lr > 3 || nu < 0) return fail(EMG3D_ERR_BADARG, "gauss_seidel: bad argument");
    return lv->is_complex ? launch_gs<cplx>(lv, lr, nu, fac, lfac, scratch, scratch_bytes, (hipStream_t)stream)
                          : launch_gs<double>(lv, lr, nu, fac, lfac, scratch, scratch_bytes, (hipStream_t)stream);
}

size_t emg3d_residual_ws_len(int nx, int ny, int nz)
{
    const emg::Dim3 g = emg::cell_grid(nx + 1, ny + 1, nz + 1);
    return (size_t)g.x * g.y * g.z;
}

int emg3d_dev_residual(const emg3d_level *lv, void *rx, void *ry, void *rz, double *ws, size_t ws_len,
                       double *sumsq, void *stream)
{
    if (!lv) return fail(EMG3D_ERR_BADARG, "residual: bad argument");
    return lv->is_complex ? launch_residual<cplx>(lv, rx, ry, rz, ws, ws_len, sumsq, (hipStream_t)stream)
                          : launch_residual<double>(lv, rx, ry, rz, ws, ws_len, sumsq, (hipStream_t)stream);
}

int emg3d_dev_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry, const void *rz,
                       const double *wxl, const double *wx0, const double *wxr, const double *wyl,
                       const double *wy0, const double *wyr, const double *wzl, const double *wz0,
                       const double *wzr, int nx, int ny, int nz, int sc_dir, int is_complex, void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "restrict: sc_dir must be 0..6");
    const double *const w[9] = {wxl, wx0, wxr, wyl, wy0, wyr, wzl, wz0, wzr};
    return is_complex ? launch_restrict<cplx>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream)
                      : launch_restrict<double>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream);
}

int emg3d_dev_prolong(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                      const int32_t *ilx, const int32_t *ily, const int32_t *ilz, const double *wx,
                      const double *wy, const double *wz, int nx, int ny, int nz, int sc_dir,
                      int is_complex, void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "prolong: sc_dir must be 0..6");
    return is_complex ? launch_prolong<cplx>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz,
                                             sc_dir, (hipStream_t)stream)
                      : launch_prolong<double>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz,
                                               sc_dir, (hipStream_t)stream);
}

int emg3d_dev_restrict_batch(void *crx, void *cry, void *crz, const void *rx, const void *ry, const void *rz,
                             const double *wxl, const double *wx0, const double *wxr, const double *wyl,
                             const double *wy0, const double *wyr, const double *wzl, const double *wz0,
                             const double *wzr, int nx, int ny, int nz, int sc_dir, int is_complex, int batch,
                             size_t fine_stride, size_t coarse_stride, void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "restrict: sc_dir must be 0..6");
    const double *const w[9] = {wxl, wx0, wxr, wyl, wy0, wyr, wzl, wz0, wzr};
    return is_complex ? launch_restrict<cplx>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream, batch,
                                              fine_stride, coarse_stride)
                      : launch_restrict<double>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream,
                                                batch, fine_stride, coarse_stride);
}

int emg3d_dev_restrict_clear_batch(void *crx, void *cry, void *crz, void *cex, void *cey, void *cez, const void *rx,
                                   const void *ry, const void *rz, const double *wxl, const double *wx0,
                                   const double *wxr, const double *wyl, const double *wy0, const double *wyr,
                                   const double *wzl, const double *wz0, const double *wzr, int nx, int ny, int nz,
                                   int sc_dir, int is_complex, int batch, size_t fine_stride, size_t coarse_stride,
                                   void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "restrict: sc_dir must be 0..6");
    if (!cex || !cey || !cez) return fail(EMG3D_ERR_BADARG, "restrict_clear: coarse field missing");
    const double *const w[9] = {wxl, wx0, wxr, wyl, wy0, wyr, wzl, wz0, wzr};
    return is_complex ? launch_restrict<cplx>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream, batch,
                                              fine_stride, coarse_stride, cex, cey, cez)
                      : launch_restrict<double>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir, (hipStream_t)stream,
                                                batch, fine_stride, coarse_stride, cex, cey, cez);
}

int emg3d_dev_prolong_batch(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                            const int32_t *ilx, const int32_t *ily, const int32_t *ilz, const double *wx,
                            const double *wy, const double *wz, int nx, int ny, int nz, int sc_dir, int is_complex,
                            int batch, size_t fine_stride, size_t coarse_stride, void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "prolong: sc_dir must be 0..6");
    return is_complex ? launch_prolong<cplx>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz, sc_dir,
                                             (hipStream_t)stream, batch, fine_stride, coarse_stride)
                      : launch_prolong<double>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz, sc_dir,
                                               (hipStream_t)stream, batch, fine_stride, coarse_stride);
}

int emg3d_dev_restrict_param(void *out, const void *in, int nx, int ny, int nz, int sc_dir, int is_complex,
                             void *stream)
{
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "restrict_param: sc_dir must be 0..6");
    return is_complex ? launch_restrict_param<cplx>(out, in, nx, ny, nz, sc_dir, (hipStream_t)stream)
                      : launch_restrict_param<double>(out, in, nx, ny, nz, sc_dir, (hipStream_t)stream);
}

int emg3d_dev_pec_zero(void *ex, void *ey, void *ez, int nx, int ny, int nz, int is_complex, void *stream)
{
    const dim3 block = d3(emg::cell_block());
    const dim3 grid = d3(emg::cell_grid(nx + 1, ny + 1, nz + 1));
    if (is_complex)
        hipLaunchKernelGGL(k_pec_zero<cplx>, grid, block, 0, (hipStream_t)stream, (cplx *)ex, (cplx *)ey, (cplx *)ez, nx, ny, nz);
    else
        hipLaunchKernelGGL(k_pec_zero<double>, grid, block, 0, (hipStream_t)stream, (double *)ex, (double *)ey, (double *)ez, nx, ny, nz);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------- host flavour ----

int emg3d_core_amat_x(void *rx, void *ry, void *rz, const void *ex, const void *ey, const void *ez,
                      const void *eta_x, const void *eta_y, const void *eta_z, const double *zeta,
                      const double *hx, const double *hy, const double *hz, int nx, int ny, int nz,
                      int is_complex)
{
    if (emg3d_device_count() < 1) return fail(EMG3D_ERR_NODEVICE, "no HIP device");
    const Sizes S(nx, ny, nz, is_complex);
    DevBuf drx, dry, drz, dex, dey, dez, dz, dhx, dhy, dhz;
    EtaUpload eta;
    HIP_TRY(drx.up(rx, S.nex * S.esz)); HIP_TRY(dry.up(ry, S.ney * S.esz)); HIP_TRY(drz.up(rz, S.nez * S.esz));
    HIP_TRY(dex.up(ex, S.nex * S.esz)); HIP_TRY(dey.up(ey, S.ney * S.esz)); HIP_TRY(dez.up(ez, S.nez * S.esz));
    HIP_TRY(eta.up(eta_x, eta_y, eta_z, S.ncc * S.esz));
    HIP_TRY(dz.up(zeta, S.ncc * 8));
    HIP_TRY(upload_inverse(dhx, hx, nx)); HIP_TRY(upload_inverse(dhy, hy, ny)); HIP_TRY(upload_inverse(dhz, hz, nz));
    emg3d_level lv = {};
    lv.nx = nx; lv.ny = ny; lv.nz = nz; lv.is_complex = is_complex;
    lv.ex = dex.d; lv.ey = dey.d; lv.ez = dez.d;
    lv.sx = drx.d; lv.sy = dry.d; lv.sz = drz.d;   // r -= A e  ==  r = r_in - A e, in place
    lv.eta_x = eta.dptr[0]; lv.eta_y = eta.dptr[1]; lv.eta_z = eta.dptr[2];
    lv.zeta = (const double *)dz.d;
    lv.ihx = (const double *)dhx.d; lv.ihy = (const double *)dhy.d; lv.ihz = (const double *)dhz.d;
    // core.amat_x only touches the entries of cells 0..n-1; the upper-boundary branch of
    // the kernel rewrites r = r there, i.e. leaves them as they are.
    int rc = emg3d_dev_residual(&lv, drx.d, dry.d, drz.d, nullptr, 0, nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(drx.down(rx)); HIP_TRY(dry.down(ry)); HIP_TRY(drz.down(rz));
    return 0;
}

int emg3d_core_gauss_seidel(int lr, void *ex, void *ey, void *ez, const void *sx, const void *sy,
                            const void *sz, const void *eta_x, const void *eta_y, const void *eta_z,
                            const double *zeta, const double *hx, const double *hy, const double *hz,
                            int nx, int ny, int nz, int nu, int is_complex)
{
    if (emg3d_device_count() < 1) return fail(EMG3D_ERR_NODEVICE, "no HIP device");
    const Sizes S(nx, ny, nz, is_complex);
    DevBuf dsx, dsy, dsz, dex, dey, dez, dz, dhx, dhy, dhz, scr;
    EtaUpload eta;
    HIP_TRY(dsx.up(sx, S.nex * S.esz)); HIP_TRY(dsy.up(sy, S.ney * S.esz)); HIP_TRY(dsz.up(sz, S.nez * S.esz));
    HIP_TRY(dex.up(ex, S.nex * S.esz)); HIP_TRY(dey.up(ey, S.ney * S.esz)); HIP_TRY(dez.up(ez, S.nez * S.esz));
    HIP_TRY(eta.up(eta_x, eta_y, eta_z, S.ncc * S.esz));
    HIP_TRY(dz.up(zeta, S.ncc * 8));
    HIP_TRY(upload_inverse(dhx, hx, nx)); HIP_TRY(upload_inverse(dhy, hy, ny)); HIP_TRY(upload_inverse(dhz, hz, nz));
    const size_t sb = emg3d_gs_scratch_bytes(lr, nx, ny, nz, is_complex);
    HIP_TRY(scr.alloc(sb));
    DevBuf dfac, dlfac;
    emg3d_level lv = {};
    lv.nx = nx; lv.ny = ny; lv.nz = nz; lv.is_complex = is_complex;
    lv.ex = dex.d; lv.ey = dey.d; lv.ez = dez.d;
    lv.sx = dsx.d; lv.sy = dsy.d; lv.sz = dsz.d;
    lv.eta_x = eta.dptr[0]; lv.eta_y = eta.dptr[1]; lv.eta_z = eta.dptr[2];
    lv.zeta = (const double *)dz.d;
    lv.ihx = (const double *)dhx.d; lv.ihy = (const double *)dhy.d; lv.ihz = (const double *)dhz.d;
    int rc = 0;
    if (lr == 0) {
        int imag = 0;
        rc = emg3d_dev_eta_is_imaginary(&lv, &imag, nullptr);
        if (rc) return rc;
        lv.flags = imag ? EMG3D_LEVEL_ETA_IMAG : 0;
    }
    HIP_TRY(dfac.alloc(lr ? emg3d_line_fac_bytes(lr, nx, ny, nz, is_complex) : emg3d_point_fac_bytes_lv(&lv)));
    HIP_TRY(dlfac.alloc(emg3d_line_lfac_bytes(lr, nx, ny, nz)));
    if (lr != 0) rc = emg3d_dev_line_setup(&lv, lr, dfac.d, (double *)dlfac.d, nullptr);
    else rc = emg3d_dev_point_setup(&lv, dfac.d, nullptr);
    if (rc) return rc;
    rc = emg3d_dev_gauss_seidel(&lv, lr, nu, dfac.d, (const double *)dlfac.d, scr.d, sb, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dex.down(ex)); HIP_TRY(dey.down(ey)); HIP_TRY(dez.down(ez));
    return 0;
}

int emg3d_core_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry, const void *rz,
                        const double *wxl, const double *wx0, const double *wxr, const double *wyl,
                        const double *wy0, const double *wyr, const double *wzl, const double *wz0,
                        const double *wzr, int nx, int ny, int nz, int sc_dir, int is_complex)
{
    if (emg3d_device_count() < 1) return fail(EMG3D_ERR_NODEVICE, "no HIP device");
    if (sc_dir < 0 || sc_dir > 6) return fail(EMG3D_ERR_BADARG, "restrict: sc_dir must be 0..6");
    const ScDirs f = sc_flags(sc_dir);
    const int cnx = f.cx ? nx / 2 : nx, cny = f.cy ? ny / 2 : ny, cnz = f.cz ? nz / 2 : nz;
    const Sizes S(nx, ny, nz, is_complex), C(cnx, cny, cnz, is_complex);
    DevBuf drx, dry, drz, dcx, dcy, dcz, w[9];
    HIP_TRY(drx.up(rx, S.nex * S.esz)); HIP_TRY(dry.up(ry, S.ney * S.esz)); HIP_TRY(drz.up(rz, S.nez * S.esz));
    // coarse arrays are uploaded too: entries the kernel does not write keep their value
    HIP_TRY(dcx.up(crx, C.nex * C.esz)); HIP_TRY(dcy.up(cry, C.ney * C.esz)); HIP_TRY(dcz.up(crz, C.nez * C.esz));
    const double *hw[9] = {wxl, wx0, wxr, wyl, wy0, wyr, wzl, wz0, wzr};
    const int wn[3] = {cnx + 1, cny + 1, cnz + 1};
    const int coarsened[3] = {f.cx, f.cy, f.cz};
    const double *dw[9];
    for (int i = 0; i < 9; ++i) {
        dw[i] = nullptr;
        if (coarsened[i / 3]) {
            HIP_TRY(w[i].up(hw[i], sizeof(double) * (size_t)wn[i / 3]));
            dw[i] = (const double *)w[i].d;
        }
    }
    int rc = emg3d_dev_restrict(dcx.d, dcy.d, dcz.d, drx.d, dry.d, drz.d, dw[0], dw[1], dw[2], dw[3], dw[4],
                                dw[5], dw[6], dw[7], dw[8], nx, ny, nz, sc_dir, is_complex, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dcx.down(crx)); HIP_TRY(dcy.down(cry)); HIP_TRY(dcz.down(crz));
    return 0;
}

int emg3d_core_blocks_to_amat(void *amat, void *bvec, const void *middle, const double *left,
                              const void *rhs, int im, int nc, int n, int is_complex)
{
    if (emg3d_device_count() < 1) return fail(EMG3D_ERR_NODEVICE, "no HIP device");
    const size_t e = is_complex ? 16 : 8;
    DevBuf da, db, dm, dl, dr;
    HIP_TRY(da.up(amat, 6 * (size_t)n * e)); HIP_TRY(db.up(bvec, (size_t)n * e));
    HIP_TRY(dm.up(middle, 25 * e)); HIP_TRY(dl.up(left, 25 * 8)); HIP_TRY(dr.up(rhs, 5 * e));
    if (is_complex)
        hipLaunchKernelGGL(k_blocks_to_amat<cplx>, dim3(1), dim3(1), 0, 0, (cplx *)da.d, (cplx *)db.d,
                           (const cplx *)dm.d, (const double *)dl.d, (const cplx *)dr.d, im, nc);
    else
        hipLaunchKernelGGL(k_blocks_to_amat<double>, dim3(1), dim3(1), 0, 0, (double *)da.d, (double *)db.d,
                           (const double *)dm.d, (const double *)dl.d, (const double *)dr.d, im, nc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(da.down(amat)); HIP_TRY(db.down(bvec));
    return 0;
}

int emg3d_core_solve(void *amat, void *bvec, int n, int is_complex)
{
    if (emg3d_device_count() < 1) return fail(EMG3D_ERR_NODEVICE, "no HIP device");
    const size_t e = is_complex ? 16 : 8;
    DevBuf da, db;
    HIP_TRY(da.up(amat, 6 * (size_t)n * e)); HIP_TRY(db.up(bvec, (size_t)n * e));
    if (is_complex)
        hipLaunchKernelGGL(k_band_solve<cplx>, dim3(1), dim3(1), 0, 0, (cplx *)da.d, (cplx *)db.d, n);
    else
        hipLaunchKernelGGL(k_band_solve<double>, dim3(1), dim3(1), 0, 0, (double *)da.d, (double *)db.d, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(da.down(amat)); HIP_TRY(db.down(bvec));
    return 0;
}

}  // extern "C"

#include "receivers.h"
#include "krylov.h"
#include "adjoint.h"

// TEST INFRASTRUCTURE ONLY -- CPU emulation of the HIP launch grids.
//
// Compiled with a plain host compiler (EMG_HD expands to `inline`), this file walks the
// same (grid, block) index spaces as emg3d_amd/csrc/kernels.hip and calls the very same
// per-thread bodies from emg3d_amd/csrc/{stencil,launch}.h on host arrays. It lets the
// `-m "not gpu"` unit tests check the kernel BODIES and their index arithmetic against the
// oracle in the build container, where no GPU exists. It is not a fallback: the product
// package never loads it, and nothing here is reachable from emg3d_amd/.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "../../emg3d_amd/csrc/launch.h"

using emg::cplx;

namespace {

struct LevelArgs {
    int32_t nx, ny, nz, is_complex;
    void *ex, *ey, *ez;
    const void *sx, *sy, *sz;
    const void *eta_x, *eta_y, *eta_z;
    const double *zeta;
    const double *ihx, *ihy, *ihz;
};

int g_point_slab = 0;
int g_point_tile_min = 1 << 20;
int g_line_order = 1;
int g_line_wide = 0;
int g_point_compact = 0;      // 1: the tile-major eta sums stored in single precision (kernels.hip: option point_compact)
int g_line_compact = 0;       // 1: T and w records of the line passes stored in single precision (kernels.hip: compact k_line_stream); 2: T only (k_line_colour)          // 1: the line passes in the wide form (stencil.h: line_wide_ref), where the level allows it

// one sweep of the tiled point-smoother schedule: the eight tile-colour launches of
// kernels.hip (k_gs_point_tile), each workgroup's phases separated like its barriers
// pmode: 0 eta sums formed on the fly, 2 tile-major buffer with full values, 3 with stored halves
template <class T> void gs_point_tiled(const emg::Level<T> &L, const void *pst, int pmode, int iback)
{
    using TB = emg::PointTile;
    using E = emg::EdgesTile<T, TB::BX, TB::BY, TB::BZ>;
    std::vector<T> lds(E::LDS_BYTES / sizeof(T) + 1);
    const int colours = emg::sweep_colours_packed(iback);
    const emg::TileCount n = emg::tile_count<TB>(L.nx, L.ny, L.nz);
    for (int t8 = 0; t8 < 8; ++t8) {
        const int tc = emg::tile_colour_at(iback, t8);
        const emg::Dim3 g = emg::tile_grid<TB>(L.nx, L.ny, L.nz, tc);
        for (int bz = 0; bz < g.z; ++bz)
            for (int by = 0; by < g.y; ++by)
                for (int bx = 0; bx < g.x; ++bx) {
                    const int tx = (tc & 1) + 2 * bx, ty = ((tc >> 1) & 1) + 2 * by, tz = ((tc >> 2) & 1) + 2 * bz;
                    const int x0 = 1 + tx * TB::BX, y0 = 1 + ty * TB::BY, z0 = 1 + tz * TB::BZ;
                    for (auto &v : lds) v = T(1e300);   // LDS is not initialised on the GPU either
                    for (int t = 0; t < TB::THREADS; ++t) emg::tile_load<T, TB>(L, lds.data(), x0, y0, z0, t);
                    for (int cc = 0; cc < 4; ++cc)
                        for (int t = 0; t < TB::THREADS; ++t) {     // the body of k_gs_point_tile's colour loop
                            const int colour = (colours >> (2 * cc)) & 3;
                            int ix, iy, iz;
                            const bool ok = emg::tile_node<TB>(L.nx, L.ny, L.nz, x0, y0, z0, colour, t, ix, iy, iz);
                            const E ed(lds.data(), x0, y0, z0);
                            emg::PointIn<T> in;
                            emg::point_load_zeta<T>(emg::ZetaTile<E>{ed}, ix, iy, iz, in);
                            emg::point_load_source<T>(L, ix, iy, iz, in);
                            if (pmode == 0) emg::point_load_eta<T, false>(L, nullptr, ix, iy, iz, in);
                            else if (pmode == 3) emg::tile_pst_load<T, TB, emg::PST_HALF>(pst, n.x, n.y, tx, ty, tz, colour, t, in);
                            else if (pmode == 4) emg::tile_pst_load<T, TB, emg::PST_HALF_F32>(pst, n.x, n.y, tx, ty, tz, colour, t, in);
                            else if (pmode == 5) emg::tile_pst_load<T, TB, emg::PST_FULL_F32>(pst, n.x, n.y, tx, ty, tz, colour, t, in);
                            else emg::tile_pst_load<T, TB, emg::PST_FULL>(pst, n.x, n.y, tx, ty, tz, colour, t, in);
                            if (ok) emg::point_update<T, E>(L, in, ed, ix, iy, iz);
                        }
                    for (int t = 0; t < TB::THREADS; ++t) emg::tile_store<T, TB>(L, lds.data(), x0, y0, z0, t);
                }
    }
}
template <class T> bool eta_purely_imaginary(const emg::Level<T> &L)
{
    const size_t n = (size_t)L.nx * L.ny * L.nz;
    for (size_t i = 0; i < n; ++i)
        if (emg::real_of(L.eta_x[i]) != 0.0 || emg::real_of(L.eta_y[i]) != 0.0 || emg::real_of(L.eta_z[i]) != 0.0)
            return false;
    return true;
}

template <class T> emg::Level<T> to_level(const LevelArgs *lv)
{
    emg::Level<T> L;
    L.nx = lv->nx; L.ny = lv->ny; L.nz = lv->nz;
    L.ex = (T *)lv->ex; L.ey = (T *)lv->ey; L.ez = (T *)lv->ez;
    L.sx = (const T *)lv->sx; L.sy = (const T *)lv->sy; L.sz = (const T *)lv->sz;
    L.eta_x = (const T *)lv->eta_x; L.eta_y = (const T *)lv->eta_y; L.eta_z = (const T *)lv->eta_z;
    L.zeta = lv->zeta; L.ihx = lv->ihx; L.ihy = lv->ihy; L.ihz = lv->ihz;
    return L;
}

// Walk a (grid x block) launch; f(gx, gy, gz) gets global thread indices.
template <class F> void for_threads(emg::Dim3 g, emg::Dim3 b, F f)
{
    for (int bz = 0; bz < g.z; ++bz)
        for (int by = 0; by < g.y; ++by)
            for (int bx = 0; bx < g.x; ++bx)
                for (int tz = 0; tz < b.z; ++tz)
                    for (int ty = 0; ty < b.y; ++ty)
                        for (int tx = 0; tx < b.x; ++tx)
                            f(bx * b.x + tx, by * b.y + ty, bz * b.z + tz);
}

template <class T, int DIR>
void line_colour(const emg::Level<T> &L, int c, const T *fac, const double *lfac, T *vec)
{
    const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
    if (lc.lines <= 0) return;
    const T *f = fac + lc.fac_off;
    const double *lf = lfac + lc.lfac_off;
    for_threads(emg::lineblk_grid(lc, true), emg::lineblk_block(), [&](int gx, int gy, int gz) {
        emg::line_rhs_thread<T, DIR>(L, c, lc.cntp, lc.cntq, gx, gy, gz, vec);
    });
    // the GPU runs these two with four lanes per line and one wave per half-chain
    // (kernels.hip); the arithmetic per line is that of the reference walks below
    // (walked over the PADDED block count, like the kernels: identity blocks are no-ops)
    for (int lid = 0; lid < lc.lines; ++lid) emg::line_forward_ref<T>(lc.n0, lc.n0p, lc.lines, lid, f, lf, vec);
    for (int lid = 0; lid < lc.lines; ++lid) emg::line_backward_ref<T>(lc.n0, lc.n0p, lc.lines, lid, f, lf, vec);
    for_threads(emg::lineblk_grid(lc, false), emg::lineblk_block(), [&](int gx, int gy, int gz) {
        emg::line_scatter_thread<T, DIR>(L, c, lc.cntp, lc.cntq, gx, gy, gz, (const T *)vec);
    });
}

// one colour pass with COMPACT records (kernels.hip: k_line_stream<.., COMPACT>): the T records rounded to single
// precision by the set-up, the w records rounded when the forward pass stores them -- and with them the raw right-hand
// sides of the rows the middle block reads, which the producer waves put into the same records --, all arithmetic
// and the solution in T
// WT: storage of the w records -- compact_of<T> (the streamed kernel: g_line_compact = 1) or T (the three-phase kernel
// k_line_colour, whose records live in LDS: g_line_compact = 2)
template <class T, int DIR, class WT>
void line_colour_compact(const emg::Level<T> &L, int c, const typename emg::compact_of<T>::type *fac, const double *lfac,
                         T *rhs, T *xout)
{
    using FT = typename emg::compact_of<T>::type;
    const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
    if (lc.lines <= 0) return;
    const FT *f = fac + lc.fac_off;
    const double *lf = lfac + lc.lfac_off;
    for_threads(emg::lineblk_grid(lc, true), emg::lineblk_block(), [&](int gx, int gy, int gz) {
        emg::line_rhs_thread<T, DIR>(L, c, lc.cntp, lc.cntq, gx, gy, gz, rhs);
    });
    std::vector<WT> w((size_t)5 * lc.n0p * lc.lines);
    const int mk = emg::line_mid(lc.n0);
    for (int lid = 0; lid < lc.lines; ++lid) {
        for (int r = 0; r < 5; ++r) w[((size_t)mk * lc.lines + lid) * 5 + r] = emg::narrow<WT>(rhs[((size_t)mk * lc.lines + lid) * 5 + r]);
        w[((size_t)(mk + 1) * lc.lines + lid) * 5] = emg::narrow<WT>(rhs[((size_t)(mk + 1) * lc.lines + lid) * 5]);
    }
    for (int lid = 0; lid < lc.lines; ++lid) emg::line_forward_ref<T, FT, WT>(lc.n0, lc.n0p, lc.lines, lid, f, lf, w.data(), (const T *)rhs);
    for (int lid = 0; lid < lc.lines; ++lid) emg::line_backward_ref<T, FT, WT>(lc.n0, lc.n0p, lc.lines, lid, f, lf, w.data(), xout);
    for_threads(emg::lineblk_grid(lc, false), emg::lineblk_block(), [&](int gx, int gy, int gz) {
        emg::line_scatter_thread<T, DIR>(L, c, lc.cntp, lc.cntq, gx, gy, gz, (const T *)xout);
    });
}

// one colour pass in the wide form (kernels.hip: k_line_wide): one walk per line, in place on the field
template <class T, int DIR>
void line_colour_wide(const emg::Level<T> &L, int c, const T *fac, const double *lfac, const T *nfac)
{
    const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
    if (lc.lines <= 0) return;
    const emg::Axes<T, DIR> A(L);
    for (int lid = 0; lid < lc.lines; ++lid) {
        int i1, i2, l2;
        emg::line_of_thread<DIR>(c, lc.cntp, lc.cntq, lid % lc.cntp, lid / lc.cntp, i1, i2, l2);
        emg::line_wide_ref<T, DIR>(A, i1, i2, lc.lines, lid, fac + lc.fac_off, lfac + lc.lfac_off,
                                   nfac + lc.fac_off / 15 * 16);
    }
}

template <class T, int DIR, class FT = T> void line_setup_all(const emg::Level<T> &L, FT *fac, double *lfac)
{
    for (int c = 0; c < 4; ++c) {
        const emg::LineClass lc = emg::line_class(DIR, L.nx, L.ny, L.nz, c);
        if (lc.lines <= 0) continue;
        for_threads(emg::line_grid(lc), emg::line_block(), [&](int gx, int gy, int) {
            emg::line_setup_thread<T, DIR, FT>(L, c, lc.cntp, lc.cntq, gx, gy, fac + lc.fac_off, lfac + lc.lfac_off);
        });
    }
}

// xfac / xlfac != nullptr: line factors given by the caller (e.g. the records the HIP set-up kernel wrote, copied to the
// host: T records as T, or as compact_of<T> when g_line_compact is set) instead of the CPU set-up's
template <class T> void gs(const LevelArgs *lv, int lr, int nu, const void *xfac = nullptr, const double *xlfac = nullptr)
{
    emg::Level<T> L = to_level<T>(lv);
    const int nx = L.nx, ny = L.ny, nz = L.nz;
    std::vector<T> vec(lr ? emg::line_vec_elems(lr - 1, nx, ny, nz) : 1);
    std::vector<T> fac(lr ? emg::line_fac_elems(lr - 1, nx, ny, nz) : 1);
    std::vector<double> lfac(lr ? emg::line_lfac_elems(lr - 1, nx, ny, nz) : 1);
    if (!xfac && lr == 1) line_setup_all<T, 0>(L, fac.data(), lfac.data());
    if (!xfac && lr == 2) line_setup_all<T, 1>(L, fac.data(), lfac.data());
    if (!xfac && lr == 3) line_setup_all<T, 2>(L, fac.data(), lfac.data());
    // compact form: the same factorisation, its T records rounded to single precision as they are stored
    using FT = typename emg::compact_of<T>::type;
    const bool compact = lr && g_line_compact;
    std::vector<FT> facc(compact ? fac.size() : 1);
    std::vector<T> xvec(compact ? vec.size() : 1);
    if (compact && !xfac && lr == 1) line_setup_all<T, 0, FT>(L, facc.data(), lfac.data());
    if (compact && !xfac && lr == 2) line_setup_all<T, 1, FT>(L, facc.data(), lfac.data());
    if (compact && !xfac && lr == 3) line_setup_all<T, 2, FT>(L, facc.data(), lfac.data());
    if (xfac && lr) {
        if (compact) std::memcpy(facc.data(), xfac, facc.size() * sizeof(FT));
        else std::memcpy(fac.data(), xfac, fac.size() * sizeof(T));
        std::memcpy(lfac.data(), xlfac, lfac.size() * sizeof(double));
    }
    // the N records of the wide form (k_line_wide_setup): one per block record, from its T and coupling entries
    const size_t nrec = lr ? fac.size() / 15 : 0;
    const bool wide = lr && g_line_wide && emg::line_wide_capable(emg::line_n0(lr - 1, nx, ny, nz), nrec);
    std::vector<T> nfac(wide ? nrec * 16 : 1);
    if (wide)
        for (size_t r = 0; r < nrec; ++r) {
            T Tk[15], N[16];
            double lf[8];
            for (int j = 0; j < 15; ++j) Tk[j] = fac[r * 15 + j];
            for (int j = 0; j < 8; ++j) lf[j] = lfac[r * 8 + j];
            emg::wide_n_record<T>(Tk, lf, N);
            for (int j = 0; j < 16; ++j) nfac[r * 16 + j] = N[j];
        }
    // point smoother: odd nu runs with the precomputed eta edge sums (k_point_setup), even
    // nu forms them on the fly -- both forms of point_load get exercised
    std::vector<T> pstv;
    const T *pst = nullptr;
    if (lr == 0 && (nu & 1)) {
        pstv.assign((size_t)nx * (ny + 1) * (nz + 1) + (size_t)(nx + 1) * ny * (nz + 1) +
                    (size_t)(nx + 1) * (ny + 1) * nz, T(0));
        for_threads(emg::cell_grid(nx + 1, ny + 1, nz + 1), emg::cell_block(), [&](int ix, int iy, int iz) {
            if (ix <= nx && iy <= ny) emg::point_setup_cell<T>(L, pstv.data(), ix, iy, iz);
        });
        pst = pstv.data();
    }
    // tiled schedule: the tile-major buffer of k_point_setup_tile, with stored halves when the
    // model allows it (nu = 1, 5, ...), with full values (nu = 3, 7, ...), or no buffer (even nu)
    const bool tiled = lr == 0 && emg::point_tiled(nx, ny, nz, g_point_tile_min);
    std::vector<double> tpst;
    int pmode = 0;
    if (tiled && (nu & 1)) {
        using TB = emg::PointTile;
        const bool half = std::is_same<T, double>::value || ((nu & 3) == 1 && eta_purely_imaginary<T>(L));
        pmode = half ? (g_point_compact ? 4 : 3) : (g_point_compact ? 5 : 2);
        const emg::TileCount n = emg::tile_count<TB>(nx, ny, nz);
        tpst.assign(emg::tile_pst_elems(nx, ny, nz, TB::BX, TB::BY, TB::BZ) * (half ? 1 : 2), 0.0);
        for (int tz = 0; tz < n.z; ++tz)
            for (int ty = 0; ty < n.y; ++ty)
                for (int tx = 0; tx < n.x; ++tx)
                    for (int t = 0; t < TB::THREADS; ++t) {
                        if (pmode == 3) emg::tile_pst_setup<T, TB, emg::PST_HALF>(L, tpst.data(), n.x, n.y, tx, ty, tz, t);
                        else if (pmode == 4) emg::tile_pst_setup<T, TB, emg::PST_HALF_F32>(L, tpst.data(), n.x, n.y, tx, ty, tz, t);
                        else if (pmode == 5) emg::tile_pst_setup<T, TB, emg::PST_FULL_F32>(L, tpst.data(), n.x, n.y, tx, ty, tz, t);
                        else emg::tile_pst_setup<T, TB, emg::PST_FULL>(L, tpst.data(), n.x, n.y, tx, ty, tz, t);
                    }
    }
    int iback = 0;
    for (int it = 0; it < nu; ++it) {
        iback = 1 - iback;
        if (tiled) {
            gs_point_tiled<T>(L, tpst.data(), pmode, iback);
            continue;
        }
        if (lr == 0) {
            emg::gs_point_schedule(nz, g_point_slab, iback, [&](int c, int iz0, int izn) {
                for_threads(emg::gs_point_grid(nx, ny, izn), emg::gs_point_block(),
                            [&](int gx, int gy, int gz) { emg::gs_point_thread<T>(L, pst, c, iz0, gx, gy, gz); });
            });
            continue;
        }
        for (int cc = 0; cc < 4; ++cc) {
            const int c = emg::line_sweep_colour(g_line_order, it, cc);
            if (compact && g_line_compact == 2) {
                if (lr == 1) line_colour_compact<T, 0, T>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                else if (lr == 2) line_colour_compact<T, 1, T>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                else line_colour_compact<T, 2, T>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                continue;
            }
            if (compact) {
                if (lr == 1) line_colour_compact<T, 0, FT>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                else if (lr == 2) line_colour_compact<T, 1, FT>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                else line_colour_compact<T, 2, FT>(L, c, facc.data(), lfac.data(), vec.data(), xvec.data());
                continue;
            }
            if (wide) {
                if (lr == 1) line_colour_wide<T, 0>(L, c, fac.data(), lfac.data(), nfac.data());
                else if (lr == 2) line_colour_wide<T, 1>(L, c, fac.data(), lfac.data(), nfac.data());
                else line_colour_wide<T, 2>(L, c, fac.data(), lfac.data(), nfac.data());
                continue;
            }
            if (lr == 1) line_colour<T, 0>(L, c, fac.data(), lfac.data(), vec.data());
            else if (lr == 2) line_colour<T, 1>(L, c, fac.data(), lfac.data(), vec.data());
            else line_colour<T, 2>(L, c, fac.data(), lfac.data(), vec.data());
        }
    }
}

template <class T> double residual(const LevelArgs *lv, void *rx, void *ry, void *rz)
{
    emg::Level<T> L = to_level<T>(lv);
    double acc = 0.0;
    for_threads(emg::cell_grid(L.nx + 1, L.ny + 1, L.nz + 1), emg::cell_block(), [&](int ix, int iy, int iz) {
        if (ix <= L.nx && iy <= L.ny) acc += emg::residual_cell<T>(L, (T *)rx, (T *)ry, (T *)rz, ix, iy, iz);
    });
    return acc;
}

}  // namespace

extern "C" {

void emu_set_point_slab(int t) { g_point_slab = t; }
void emu_set_point_tile_min(int n) { g_point_tile_min = n; }
void emu_set_line_order(int o) { g_line_order = o; }
void emu_set_line_wide(int w) { g_line_wide = w; }
void emu_set_line_compact(int c) { g_line_compact = c; }
void emu_set_point_compact(int c) { g_point_compact = c; }
void emu_set_point_order(int o) { emg::point_order_ref() = o; }

void emu_gauss_seidel(const LevelArgs *lv, int lr, int nu)
{
    if (lv->is_complex) gs<cplx>(lv, lr, nu); else gs<double>(lv, lr, nu);
}
// line sweeps with the caller's factor records (layout of emg3d_dev_line_setup; compact records if emu_set_line_compact(1))
void emu_gauss_seidel_fac(const LevelArgs *lv, int lr, int nu, const void *fac, const double *lfac)
{
    if (lv->is_complex) gs<cplx>(lv, lr, nu, fac, lfac); else gs<double>(lv, lr, nu, fac, lfac);
}

double emu_residual(const LevelArgs *lv, void *rx, void *ry, void *rz)
{
    return lv->is_complex ? residual<cplx>(lv, rx, ry, rz) : residual<double>(lv, rx, ry, rz);
}

void emu_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry, const void *rz,
                  const double *const *w, int nx, int ny, int nz, int sc_dir, int is_complex)
{
    if (is_complex) {
        auto R = emg::make_restrict<cplx>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir);
        for_threads(emg::cell_grid(R.cnxn, R.cnyn, R.cnzn), emg::cell_block(), [&](int i, int j, int k) {
            if (i < R.cnxn && j < R.cnyn) emg::restrict_node<cplx>(R, i, j, k);
        });
    } else {
        auto R = emg::make_restrict<double>(crx, cry, crz, rx, ry, rz, w, nx, ny, nz, sc_dir);
        for_threads(emg::cell_grid(R.cnxn, R.cnyn, R.cnzn), emg::cell_block(), [&](int i, int j, int k) {
            if (i < R.cnxn && j < R.cnyn) emg::restrict_node<double>(R, i, j, k);
        });
    }
}

void emu_prolong(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                 const int *ilx, const int *ily, const int *ilz, const double *wx, const double *wy,
                 const double *wz, int nx, int ny, int nz, int sc_dir, int is_complex)
{
    if (is_complex) {
        auto P = emg::make_prolong<cplx>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz, sc_dir);
        for_threads(emg::cell_grid(nx + 1, ny + 1, nz + 1), emg::cell_block(), [&](int i, int j, int k) {
            if (i <= nx && j <= ny) emg::prolong_cell<cplx>(P, i, j, k);
        });
    } else {
        auto P = emg::make_prolong<double>(ex, ey, ez, cex, cey, cez, ilx, ily, ilz, wx, wy, wz, nx, ny, nz, sc_dir);
        for_threads(emg::cell_grid(nx + 1, ny + 1, nz + 1), emg::cell_block(), [&](int i, int j, int k) {
            if (i <= nx && j <= ny) emg::prolong_cell<double>(P, i, j, k);
        });
    }
}

void emu_restrict_param(void *out, const void *in, int nx, int ny, int nz, int sc_dir, int is_complex)
{
    const emg::ScDirs f = emg::sc_flags(sc_dir);
    const int fx = f.cx ? 2 : 1, fy = f.cy ? 2 : 1, fz = f.cz ? 2 : 1;
    const int cnx = nx / fx, cny = ny / fy, cnz = nz / fz;
    for (int k = 0; k < cnz; ++k)
        for (int j = 0; j < cny; ++j)
            for (int i = 0; i < cnx; ++i) {
                if (is_complex)
                    emg::restrict_param_cell<cplx>((cplx *)out, (const cplx *)in, nx, ny, cnx, cny, fx, fy, fz, i, j, k);
                else
                    emg::restrict_param_cell<double>((double *)out, (const double *)in, nx, ny, cnx, cny, fx, fy, fz, i, j, k);
            }
}

// The blocks of ONE line system as stencil.h assembles them (complex fields): per block k the
// diagonal dg[5], the strictly lower real part mid[5][5], the coupling to the previous block
// (left0[5], leftd[5]) and the right-hand side rhs[5] -- for numerical experiments with other
// elimination orders (tools/line_reduced_prototype.py) against the kernels' own inputs.
void emu_line_blocks(const LevelArgs *lv, int dir, int i1, int i2, void *dg_, double *mid_, double *left0_,
                     double *leftd_, void *rhs_)
{
    const emg::Level<cplx> L = to_level<cplx>(lv);
    cplx *dgo = (cplx *)dg_, *rhso = (cplx *)rhs_;
    auto run = [&](auto A) {
        const int n0 = A.n0();
        for (int k = 0; k < n0; ++k) {
            cplx dg[5], rhs[5];
            double mid[5][5] = {}, l0[5], ld[5];
            if (dir == 0) {
                emg::line_matrix<cplx, 0>(emg::Axes<cplx, 0>(L), k, i1, i2, dg, mid, l0, ld);
                emg::line_rhs<cplx, 0>(emg::Axes<cplx, 0>(L), k, i1, i2, rhs);
            } else if (dir == 1) {
                emg::line_matrix<cplx, 1>(emg::Axes<cplx, 1>(L), k, i1, i2, dg, mid, l0, ld);
                emg::line_rhs<cplx, 1>(emg::Axes<cplx, 1>(L), k, i1, i2, rhs);
            } else {
                emg::line_matrix<cplx, 2>(emg::Axes<cplx, 2>(L), k, i1, i2, dg, mid, l0, ld);
                emg::line_rhs<cplx, 2>(emg::Axes<cplx, 2>(L), k, i1, i2, rhs);
            }
            for (int r = 0; r < 5; ++r) {
                dgo[k * 5 + r] = dg[r];
                rhso[k * 5 + r] = rhs[r];
                left0_[k * 5 + r] = l0[r];
                leftd_[k * 5 + r] = ld[r];
                for (int m = 0; m < 5; ++m) mid_[(k * 5 + r) * 5 + m] = m < r ? mid[r][m] : 0.0;
            }
        }
    };
    if (dir == 0) run(emg::Axes<cplx, 0>(L)); else if (dir == 1) run(emg::Axes<cplx, 1>(L)); else run(emg::Axes<cplx, 2>(L));
}

// Residual through the column walk of the HIP kernels (stencil.h: residual_column -- operands shared by vertically
// adjacent cells carried from cell to cell), `zb` planes per walk; returns sum |r|^2.
double emu_residual_column(const LevelArgs *lv, void *rx, void *ry, void *rz, int zb)
{
    double acc = 0.0;
    auto run = [&](auto L, auto *px, auto *py, auto *pz) {
        using T = std::remove_pointer_t<decltype(px)>;
        for (int z0 = 0; z0 <= L.nz; z0 += zb)
            for (int iy = 0; iy <= L.ny; ++iy)
                for (int ix = 0; ix <= L.nx; ++ix)
                    acc += emg::residual_column<T>(L, px, py, pz, ix, iy, z0, std::min(z0 + zb, L.nz + 1));
    };
    if (lv->is_complex) run(to_level<cplx>(lv), (cplx *)rx, (cplx *)ry, (cplx *)rz);
    else run(to_level<double>(lv), (double *)rx, (double *)ry, (double *)rz);
    return acc;
}

// The coupling entries k_line_stream's producers recompute (stencil.h: line_coupling) against the lfac records the
// set-up stores for one line: number of the 8 x (records the half-chains read) doubles that differ (bitwise).
int emu_line_coupling_mismatches(const LevelArgs *lv, int dir, int i1, int i2)
{
    const emg::Level<cplx> L = to_level<cplx>(lv);
    int bad = 0;
    auto run = [&](auto A, auto dirtag) {
        constexpr int DIR = decltype(dirtag)::value;
        const int n0 = A.n0(), n0p = emg::line_padded(n0), mk = emg::line_mid(n0);
        std::vector<cplx> fac((size_t)15 * n0p);
        std::vector<double> lfac((size_t)8 * n0p);
        emg::line_setup<cplx, DIR>(L, i1, i2, fac.data(), lfac.data(), 1, 0, n0p);
        for (int k = 0; k < n0p; ++k) {
            if (k == mk || k == mk + 1) continue;          // the middle records are read from lfac by the kernels
            double c[8];
            emg::line_coupling<cplx, DIR>(A, k, i1, i2, k >= mk + 2, c);
            for (int r = 0; r < 8; ++r)
                if (std::memcmp(&c[r], &lfac[(size_t)k * 8 + r], sizeof(double)) != 0 && !(c[r] == 0.0 && lfac[(size_t)k * 8 + r] == 0.0)) ++bad;
        }
    };
    if (dir == 0) run(emg::Axes<cplx, 0>(L), std::integral_constant<int, 0>());
    else if (dir == 1) run(emg::Axes<cplx, 1>(L), std::integral_constant<int, 1>());
    else run(emg::Axes<cplx, 2>(L), std::integral_constant<int, 2>());
    return bad;
}

void emu_solve(void *amat, void *bvec, int n, int is_complex)
{
    if (is_complex) emg::band_solve<cplx>((cplx *)amat, (cplx *)bvec, n);
    else emg::band_solve<double>((double *)amat, (double *)bvec, n);
}

void emu_blocks_to_amat(void *amat, void *bvec, const void *middle, const double *left, const void *rhs,
                        int im, int nc, int is_complex)
{
    if (is_complex)
        emg::blocks_to_amat<cplx>((cplx *)amat, (cplx *)bvec, (const cplx *)middle, left, (const cplx *)rhs, im, nc);
    else
        emg::blocks_to_amat<double>((double *)amat, (double *)bvec, (const double *)middle, left,
                                    (const double *)rhs, im, nc);
}

}  // extern "C"

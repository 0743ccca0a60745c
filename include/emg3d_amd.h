/*
 * emg3d_amd -- C ABI of the MI355X (gfx950) multigrid inner loop.
 *
 * Drop-in boundary for the hot path of emsig/emg3d: every entry point replaces one of the
 * numba kernels of the reference's emg3d/core.py (called from emg3d/solver.py at the
 * sites cited below) or one of the array-level pieces of emg3d/solver.py that have to live
 * on the device once the level hierarchy is device-resident.
 *
 * Two flavours:
 *   emg3d_core_*   HOST pointers, synchronous, exactly the arguments of the reference's
 *                  `core.<fn>` plus explicit sizes -- what a ctypes binding inside the
 *                  reference would call. Data is staged through HBM
 *                  per call; this flavour exists for parity tests and literal drop-in use.
 *   emg3d_dev_*    DEVICE pointers + hipStream_t, asynchronous; used by the device-resident
 *                  multigrid driver (emg3d_amd/solver.py).
 *
 * Conventions
 *   - All 3-D arrays are Fortran-ordered (x fastest), as in the reference
 *     (emg3d/fields.py:201-259): ex (nx,ny+1,nz+1), ey (nx+1,ny,nz+1), ez (nx+1,ny+1,nz),
 *     eta_x/eta_y/eta_z/zeta (nx,ny,nz). nx,ny,nz are CELL counts.
 *   - is_complex = 1: fields and eta are complex128 (interleaved re,im); 0: float64
 *     (Laplace domain, emg3d/fields.py:93-98). zeta and all widths/weights are float64.
 *   - eta_x, eta_y, eta_z may alias each other (isotropic / VTI / HTI models,
 *     emg3d/models.py:693-712); they are never written.
 *   - Return value: 0 on success, otherwise a HIP error code (hipError_t) or a negative
 *     emg3d_amd code; emg3d_last_error() gives the message. Numerical failure (zero
 *     pivot -> inf/nan) is NOT an error here, exactly as in the reference
 *     (emg3d/core.py:1560,1576); it surfaces through the residual norm.
 *   - Smoother ordering: four-colour ordering.
 *     point:  colour = ((ix+iz)&1) | (((iy+iz)&1)<<1);
 *     lines:  colour = (p&1) | ((q&1)<<1), (p,q) the transverse node indices in memory
 *             order: x-lines (iy,iz), y-lines (ix,iz), z-lines (ix,iy).
 *     Point smoother (option "point_order" = 1, the default since round 3): every sweep visits
 *     the node colours in the sequence 0,2,3,1 ("point_order" = 0: a backward sweep -- the
 *     first, third, ... sweep of a call, like the reference's backward-first alternation,
 *     emg3d/core.py:301,311 -- visits them in reverse: the rule of rounds 1-2, which needs up to
 *     a third more cycles). The sweep direction still reverses the TILE order of
 *     the tiled schedule below.
 *     Line smoothers (option "line_order" = 1, the default since round 3): the colour passes of
 *     a call cycle through the classes 1,2,3,0,1,2,3,0,...; sweep number it = 0,1,... of the call
 *     takes positions 3 it .. 3 it + 3 of that sequence (its first pass, for it > 0, repeats the
 *     class the previous sweep ended with: identical values, not launched). Measured on reduced
 *     copies of BASELINE.json's configurations this needs 12-18 % fewer cycles than mirrored
 *     sweeps at the same cost per cycle. "line_order" = 0: the mirrored sweeps
 *     0,2,3,1 / 1,3,2,0 (the lines' order in rounds 1-2). "line_order" = 2: the classes 1,2,3,0 in
 *     every sweep (4 nu launches per call instead of 3 nu + 1; one cycle in ten fewer at full size,
 *     a seventh more per smoothing call: equal in time).
 *     Point smoother on LARGE levels -- (nx-1)(ny-1)(nz-1) >= option "point_tile_min"
 *     (default 2^20) -- the interior nodes are cut into tiles of 32 x 4 x 6 nodes (tile t
 *     along an axis = nodes 1 + t*B .. (t+1)*B) which are coloured
 *     (tx&1)|((ty&1)<<1)|((tz&1)<<2); a forward sweep visits the tile colours in the order
 *     0,7,1,6,2,5,3,4 (backward: reversed; complementary colours are independent of each other
 *     and share a launch) and, inside every tile, the four node colours as above. This is the order in
 *     which one workgroup can keep a tile in LDS for all four node colours (one pass over
 *     the field per sweep instead of four); it is a Gauss-Seidel sweep like the others and
 *     converges alike. The oracle restates all of these orders.
 */
#ifndef EMG3D_AMD_H
#define EMG3D_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMG3D_AMD_VERSION 100 /* 0.1.0 */

#define EMG3D_ERR_BADARG (-1)
#define EMG3D_ERR_NODEVICE (-2)
#define EMG3D_ERR_SCRATCH (-3)

/* One grid level with device-resident arrays (emg3d_dev_* flavour). */
typedef struct emg3d_level {
    int32_t nx, ny, nz;        /* cells */
    int32_t is_complex;        /* 1 complex128, 0 float64 */
    void *ex, *ey, *ez;        /* electric field, updated in place by the smoothers */
    const void *sx, *sy, *sz;  /* source field */
    const void *eta_x, *eta_y, *eta_z;
    const double *zeta;
    const double *ihx, *ihy, *ihz; /* INVERSE cell widths 1/h (device) */
    /* Several right-hand sides that share the model (the sources of one frequency,
     * emg3d/simulations.py:1453-1464): batch > 1 of them are smoothed / reduced / transferred by
     * the same launches. Source b's buffers [ex|ey|ez] and [sx|sy|sz] start b * batch_stride
     * ELEMENTS behind source 0's (the pointers above). 0 or 1: a single source. */
    int32_t batch;
    int32_t flags;             /* EMG3D_LEVEL_* bits, 0 if unknown */
    int64_t batch_stride;
} emg3d_level;

/* emg3d_level::flags. ETA_IMAG: every eta value has a real part of exactly zero -- the
 * diffusive approximation at a real frequency, eta = -i omega mu0 sigma V (emg3d/models.py:
 * 677-678, epsilon_r is None). The tiled point smoother then keeps its eta edge sums as 8-byte
 * values (half the bytes, identical results). Never required; a caller that does not know
 * leaves it 0 or asks emg3d_dev_eta_is_imaginary. */
#define EMG3D_LEVEL_ETA_IMAG 1
/* LINE_COMPACT: the caller's promise that this level solves a CORRECTION equation -- its right-hand side is a
 * residual and its result is added to a field kept in full precision elsewhere (every coarse level of a cycle;
 * the finest level in residual form or as a Krylov preconditioner). The line passes that stream their records
 * through HBM (k_line_stream: lines of ~128 blocks and more) may then keep the inverse blocks T_k of the stored
 * factorisation and the forward pass's w records in SINGLE precision: values are rounded once when they are stored
 * and widened when they are loaded, every operation, every right-hand side and the solution stay fp64. The line
 * solve becomes (A_line (1 + O(eps32 cond(S_k))))^-1 -- a perturbation of the smoother, not of the equation: the
 * iteration converges to the same field at the same rate as long as eps32 x cond of the 5 x 5 blocks (~ 1 / (omega mu
 * sigma h^2)) stays well below the smoothing factor, which is the caller's to check (emg3d_amd.solver: cond <= 3e4).
 * What it buys: 120 + 2 x 40 instead of 240 + 2 x 80 B per block and colour pass of the ~1 210 B such a pass moves,
 * and 184 instead of 304 B per cell and direction of factor memory. Set it BEFORE emg3d_dev_line_setup and keep it:
 * set-up, emg3d_line_fac_bytes_lv and the smoother read it from the level. Unset (0): everything fp64. */
#define EMG3D_LEVEL_LINE_COMPACT 2
/* POINT_COMPACT: the same promise (the level solves a correction equation) for the tiled point smoother: its eta edge
 * sums -- the imaginary (conduction) part of the diagonals of its 6 x 6 systems, emg3d/core.py:377-412, stored per
 * (tile, node colour, entry, thread) -- are kept in single precision: 24 instead of 48 B per node of the ~360 a sweep
 * moves (full values, with epsilon_r: 48 instead of 96). A relative perturbation of 6e-8 of those diagonal parts, i.e.
 * of the smoother, never of the residual the iteration is driven by. emg3d_point_compact_used says whether a level's
 * point smoother runs that way (flag set and the level large enough for the tiled schedule). */
#define EMG3D_LEVEL_POINT_COMPACT 4

int emg3d_version(void);
const char *emg3d_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int emg3d_device_count(void);
/* Tuning knobs; all but "point_tile_min", "line_order", "point_order" (the order of the sweeps) and "line_wide" (the
 * rounding of the short-line solves) never change results. "point_slab": plane-slab thickness of the point
 * smoother's launch schedule (0 = one launch per colour over all planes). "point_tile_min"
 * DOES select the sweep order of the point smoother (see above; <= 0 never tiled). "line_lds":
 * 1 (default) keeps the right-hand-side / solution records of a fused line launch in LDS
 * when they fit, 0 always uses the global scratch. "line_lpw": lines per workgroup of a fused
 * line launch (4, 8, 16, 32; 0 = automatic; with 32 the streamed kernels, whose two chain waves serve
 * 16 lines, are not used). "line_fuse":
 * 0 three launches per colour and line direction (rhs, forward, backward), 1 one fused
 * launch, 2 (default) fused for colour classes with at most "line_fuse_max" lines (default:
 * no limit -- the fused launch is the faster one at every size measured). "skip_repeat": 1
 * (default) does not launch the colour pass that repeats the last colour class of the previous
 * sweep of the same call (it reproduces the same values bit by bit); 0 launches every pass.
 * "tile_fuse": 1 (default) lets the tiles of the tiled point smoother where two consecutive
 * sweeps meet run both sweeps on one LDS copy (same operations, one load / store less).
 * "line_stream": 2 (default) runs the colour passes of lines whose records do not fit the LDS of a CU (~128
 * blocks and more with 16 lines per workgroup) with the right-hand sides and coupling entries (forward pass)
 * and the w records (backward pass) staged through LDS rings by producer waves while the chain waves substitute
 * (k_line_stream: no round trip of the right-hand sides through the scratch, bit-identical results); 1 only
 * where not even slots 0..3 of the records fit (~160 blocks and more); 0 the
 * three-phase kernel everywhere. "line_stream_lf": 1 (default) one source's coupling entries are recomputed by the
 * producers instead of fetched (bit-equal values). "line_stream_r": rows per
 * half of that ring (0 = 16; a multiple of 4 in 4..32, anything else is refused; fewer where several
 * right-hand sides share the LDS). "line_stream_bmin": with several right-hand sides
 * (emg3d_level::batch > 1) such passes on lines of at least this many blocks (default 64; <= 0: never)
 * serve GROUPS of up to four right-hand sides per workgroup, every factor row fetched once per group
 * (k_line_stream<.., B>; per source the same arithmetic: bit-identical to separate solves) -- with the batch
 * as a grid dimension every source's workgroups fetch the factors again.
 * "line_wide": lines of at most this many blocks (default 33; 0 = never; only on levels small enough to hold one more
 * 16-entry record per block) are solved by k_line_wide: the same direct solve of the line system as the other line
 * kernels (emg3d/core.py:1481-1616) with the same factors, but with the block recurrences restated in four unknowns
 * through the model-only matrices N_k = (T_k C_k)[1..4, 1..4], which are one more rounding of T_k C_k -- fields agree
 * with the other kernels' to rounding (~1e-13 per call), not bit for bit. A level runs the same kernel for every
 * right-hand side, alone or in a batch, so batched and separate solves stay bit-identical under any value.
 * "line_order" and "point_order" DO select the order of the sweeps (see above), like
 * "point_tile_min".
 * "line_compact": 0 (default) compact line records where the level asks for them (EMG3D_LEVEL_LINE_COMPACT), 1 on
 * every level whose direction streams (tests, timing), -1 never (the flag is ignored). CHANGES the rounding of the
 * streamed line solves (see EMG3D_LEVEL_LINE_COMPACT).
 * "point_compact": the same three values for the eta sums of the tiled point smoother (EMG3D_LEVEL_POINT_COMPACT).
 * "line_compact_rd" (0 = 8 for x-lines, 4 else; 4 | 8): prefetch depth of the chain waves of the compact streamed
 * kernel; no influence on results.
 * "line_debug" is for timing experiments only (bit 0 aliases the records of a line: WRONG
 * results); a non-zero value is refused unless the environment variable EMG3D_AMD_ALLOW_DEBUG is set.
 * Out-of-range values of "line_order" (0..2) and "point_order" (0..1) are refused (EMG3D_ERR_BADARG). */
int emg3d_set_option(const char *name, int value);
int emg3d_get_option(const char *name);
/* enumeration of the options (for callers that key cached state -- captured graphs, option-
 * dependent buffers -- on the whole option set): names 0 .. emg3d_option_count()-1 */
int emg3d_option_count(void);
const char *emg3d_option_name(int i);
/* a counter that advances whenever emg3d_set_option changes a value: a cheap key for such caches */
int emg3d_options_generation(void);
/* The kernel that runs a colour pass of line direction lr (1/2/3) on a level of (nx,ny,nz) cells with
 * `batch` right-hand sides under the current options -- "k_line_stream" (records staged through an LDS
 * ring; with batch > 1: groups of right-hand sides per factor fetch), "k_line_colour" (three phases in one
 * launch) or the three separate kernels --, decided on the level's largest colour class by the very rule
 * the launcher uses. For profiles and benchmarks; "" on bad input. */
const char *emg3d_line_kernel_name(int lr, int nx, int ny, int nz, int is_complex, int batch);

/* ---------------------------------------------------------------- host flavour ---- */

/* core.amat_x (emg3d/core.py:57-206; called at emg3d/solver.py:695,1060):  r -= A e */
int emg3d_core_amat_x(void *rx, void *ry, void *rz, const void *ex, const void *ey, const void *ez,
                      const void *eta_x, const void *eta_y, const void *eta_z, const double *zeta,
                      const double *hx, const double *hy, const double *hz, int nx, int ny, int nz,
                      int is_complex);

/* core.gauss_seidel / _x / _y / _z (emg3d/core.py:210-1348; solver.py:837-846).
 * lr = 0 point, 1 x-line, 2 y-line, 3 z-line. */
int emg3d_core_gauss_seidel(int lr, void *ex, void *ey, void *ez, const void *sx, const void *sy,
                            const void *sz, const void *eta_x, const void *eta_y, const void *eta_z,
                            const double *zeta, const double *hx, const double *hy, const double *hz,
                            int nx, int ny, int nz, int nu, int is_complex);

/* core.restrict (emg3d/core.py:1620-2001; solver.py:937). Coarse/fine NODE counts are
 * implied: fine cells (nx,ny,nz), coarse cells = fine/2 in every coarsened direction.
 * w?l/w?0/w?r: weights of emg3d/core.py:2004-2076 (length = coarse nodes; ignored for a
 * direction that sc_dir does not coarsen). */
int emg3d_core_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry,
                        const void *rz, const double *wxl, const double *wx0, const double *wxr,
                        const double *wyl, const double *wy0, const double *wyr, const double *wzl,
                        const double *wz0, const double *wzr, int nx, int ny, int nz, int sc_dir,
                        int is_complex);

/* core.blocks_to_amat (emg3d/core.py:1351-1477) and core.solve (emg3d/core.py:1481-1616)
 * on the reference's banded storage A(i,j) -> amat[i+5j]. Inside the line smoothers these
 * never exist (the factorisation is streamed); the entry points are kept for the
 * reference's known-answer tests and run as single-thread device kernels. */
int emg3d_core_blocks_to_amat(void *amat, void *bvec, const void *middle, const double *left,
                              const void *rhs, int im, int nc, int n, int is_complex);
int emg3d_core_solve(void *amat, void *bvec, int n, int is_complex);

/* -------------------------------------------------------------- device flavour ---- */

/* Line relaxation keeps the block factorisation of every line matrix in HBM: it depends
 * on eta, zeta and h only, so it is computed once per level and direction
 * (emg3d_dev_line_setup) and reused by every sweep -- the work core.solve
 * (emg3d/core.py:1481-1616) repeats on every call of the reference's line smoothers.
 * Accuracy: the local systems are as ill-conditioned as 1 / (omega mu sigma h^2) (1e5 in sediments
 * at 1 Hz, 5e9 in air of 1e8 Ohm m), so any two fp64 evaluations of a sweep differ by eps x cond:
 * 1e-14 (marine model) ... 3e-6 (the same model under an air layer, random fields) between these
 * kernels -- point and line smoothers alike -- and the reference's arithmetic. What sets the line
 * smoothers apart is the RESIDUAL a solve leaves: multiplying by the stored inverses of the 5 x 5
 * Schur complements it is eps x cond x |right-hand side|, where a substitution (the point
 * smoother's, the reference's) leaves eps x |right-hand side|. An iteration whose tolerance is below
 * that runs the smoother on the residual equation (emg3d_amd.solve(residual_form=...) on the finest
 * level; coarse levels and preconditioner calls are in that form anyway): the errors then scale
 * with the residual and the iteration converges to round-off.
 * Sizes of the two factor buffers (complex/real part `fac`, real coupling part `lfac`)
 * and of the per-call scratch (right-hand sides / solutions of one colour class): */
size_t emg3d_line_fac_bytes(int lr, int nx, int ny, int nz, int is_complex);
/* ... of `fac` for THIS level: smaller where its direction lr keeps compact records (EMG3D_LEVEL_LINE_COMPACT and the
 * direction streams; emg3d_line_compact_used says whether), else emg3d_line_fac_bytes -- which is always enough */
size_t emg3d_line_fac_bytes_lv(const emg3d_level *lv, int lr);
int emg3d_line_compact_used(const emg3d_level *lv, int lr);
int emg3d_point_compact_used(const emg3d_level *lv);
size_t emg3d_line_lfac_bytes(int lr, int nx, int ny, int nz);
size_t emg3d_gs_scratch_bytes(int lr, int nx, int ny, int nz, int is_complex); /* 0 for lr = 0 */

/* Factorise all lines of direction lr (1/2/3 = x/y/z) of level lv into fac / lfac. */
int emg3d_dev_line_setup(const emg3d_level *lv, int lr, void *fac, double *lfac, void *stream);

/* Point smoother: the eta edge sums of emg3d/core.py:377-390 (one value per edge, arrays
 * shaped like ex|ey|ez) depend on the model only; computed once per level they replace 24
 * scattered eta loads per node by 6. Optional: emg3d_dev_gauss_seidel(lr = 0) forms the sums
 * on the fly when fac == NULL; both give identical bits. */
size_t emg3d_point_fac_bytes(int nx, int ny, int nz, int is_complex);   /* upper bound (flags = 0) */
size_t emg3d_point_fac_bytes_lv(const emg3d_level *lv);                  /* exact, honours lv->flags */
int emg3d_dev_point_setup(const emg3d_level *lv, void *fac, void *stream);
/* On levels that use the tiled schedule (option "point_tile_min", which must not change between
 * setup and use) the sums are laid out tile by tile, node colour by node colour, so that every
 * colour step of a workgroup reads them as one dense stream and every byte once per sweep; each
 * edge sum then exists twice (once per end node): 96 B per node, or 48 B with
 * EMG3D_LEVEL_ETA_IMAG / real fields. Elsewhere: arrays shaped like ex|ey|ez.
 *
 * *result = 1 if all real parts of lv's eta arrays are exactly zero (synchronises `stream`). */
int emg3d_dev_eta_is_imaginary(const emg3d_level *lv, int *result, void *stream);

/* nu sweeps of the smoother lr (0 point, 1/2/3 line along x/y/z) on level lv.
 * fac/lfac: from emg3d_dev_line_setup for the same level and lr; for lr = 0 fac is the
 * buffer of emg3d_dev_point_setup or NULL, lfac is ignored. With lv->batch > 1 all right-hand
 * sides are swept by the same launches (the factors are shared); scratch must then hold
 * batch x emg3d_gs_scratch_bytes. */
int emg3d_dev_gauss_seidel(const emg3d_level *lv, int lr, int nu, const void *fac, const double *lfac,
                           void *scratch, size_t scratch_bytes, void *stream);

/* Number of doubles of workspace emg3d_dev_residual needs for its block partial sums. */
size_t emg3d_residual_ws_len(int nx, int ny, int nz);

/* solver.residual (emg3d/solver.py:1022-1070): r = s - A e for the whole buffer
 * (rx/ry/rz may be NULL: norm only; they may alias lv->s*: in-place core.amat_x form).
 * If sumsq != NULL, *sumsq (device double) receives sum |r|^2 over all entries
 * (deterministic two-stage reduction through ws). With lv->batch > 1: r buffers stacked like
 * the fields (batch_stride), ws of batch x emg3d_residual_ws_len doubles, sumsq[batch]. */
int emg3d_dev_residual(const emg3d_level *lv, void *rx, void *ry, void *rz, double *ws,
                       size_t ws_len, double *sumsq, void *stream);

/* core.restrict on device pointers; fine cells (nx,ny,nz). w arrays are device pointers. */
int emg3d_dev_restrict(void *crx, void *cry, void *crz, const void *rx, const void *ry,
                       const void *rz, const double *wxl, const double *wx0, const double *wxr,
                       const double *wyl, const double *wy0, const double *wyr, const double *wzl,
                       const double *wz0, const double *wzr, int nx, int ny, int nz, int sc_dir,
                       int is_complex, void *stream);

/* solver.prolongation (emg3d/solver.py:947-1019): fine e += P coarse e, interior only.
 * il?/w? (device, length fine nodes): lower coarse node and weight of the upper coarse
 * node per fine node (emg3d/solver.py:1457-1462). */
int emg3d_dev_prolong(void *ex, void *ey, void *ez, const void *cex, const void *cey, const void *cez,
                      const int32_t *ilx, const int32_t *ily, const int32_t *ilz, const double *wx,
                      const double *wy, const double *wz, int nx, int ny, int nz, int sc_dir,
                      int is_complex, void *stream);

/* The two transfers for `batch` right-hand sides stacked like emg3d_level::batch describes:
 * source b's fine buffers start b * fine_stride elements behind source 0's, its coarse buffers
 * b * coarse_stride; one launch serves all of them. */
int emg3d_dev_restrict_batch(void *crx, void *cry, void *crz, const void *rx, const void *ry,
                             const void *rz, const double *wxl, const double *wx0, const double *wxr,
                             const double *wyl, const double *wy0, const double *wyr, const double *wzl,
                             const double *wz0, const double *wzr, int nx, int ny, int nz, int sc_dir,
                             int is_complex, int batch, size_t fine_stride, size_t coarse_stride,
                             void *stream);
/* the same restriction that also sets the coarse FIELD ce* to zero -- the start value of the coarse
 * solve (emg3d/solver.py:941) -- in the same pass over the coarse edges (no separate fill launch) */
int emg3d_dev_restrict_clear_batch(void *crx, void *cry, void *crz, void *cex, void *cey, void *cez,
                                   const void *rx, const void *ry, const void *rz, const double *wxl,
                                   const double *wx0, const double *wxr, const double *wyl,
                                   const double *wy0, const double *wyr, const double *wzl,
                                   const double *wz0, const double *wzr, int nx, int ny, int nz,
                                   int sc_dir, int is_complex, int batch, size_t fine_stride,
                                   size_t coarse_stride, void *stream);
int emg3d_dev_prolong_batch(void *ex, void *ey, void *ez, const void *cex, const void *cey,
                            const void *cez, const int32_t *ilx, const int32_t *ily, const int32_t *ilz,
                            const double *wx, const double *wy, const double *wz, int nx, int ny, int nz,
                            int sc_dir, int is_complex, int batch, size_t fine_stride,
                            size_t coarse_stride, void *stream);

/* solver._restrict_model_parameters (emg3d/solver.py:1667-1718): coarse = sum of the
 * 2/4/8 fine cells; is_complex refers to the parameter dtype (eta: field dtype, zeta: 0). */
int emg3d_dev_restrict_param(void *out, const void *in, int nx, int ny, int nz, int sc_dir,
                             int is_complex, void *stream);

/* Zero the tangential field on the six PEC faces (emg3d/solver.py:349-355). */
int emg3d_dev_pec_zero(void *ex, void *ey, void *ez, int nx, int ny, int nz, int is_complex,
                       void *stream);

/* ---- around the multigrid cycle: Krylov vector work -------------------
 * solver.krylov (emg3d/solver.py:652-784) wraps scipy.sparse.linalg.{bicgstab, cgs} around the
 * multigrid preconditioner; their vector updates and inner products run on the device as
 * instances of one fused step,
 *     y = sum_{i < nterms} c_i xs[i]         c_i = table[slots[i]] * scales[i], or scales[i] alone
 *                                            if slots[i] < 0;  y may be one of the xs; nterms <= 4
 *     table[dslots[k]] = conj(das[k]) . dbs[k]   for k < ndots <= 3, evaluated with the new y
 * followed by up to 8 scalar instructions prog[4 i .. 4 i + 3] = (op, dst, a, b) on the table:
 * op 0 dst = a / b, 1 dst = a * b, 2 dst = -a, 3 dst = a. `table` holds complex scalars as
 * (re, im) pairs of doubles in device memory (real fields use the real parts), so the scalars
 * of the recurrences stay on the GPU; the host reads the table when it has to decide
 * (convergence, breakdown). ws: emg3d_krylov_ws_len() doubles. Vectors have n entries of the
 * field dtype; the arrays of pointers / integers / scales are HOST arrays. */
size_t emg3d_krylov_ws_len(void);
int emg3d_dev_krylov_step(size_t n, int is_complex, void *y, int nterms, const void *const *xs, const int *slots,
                          const double *scales, int ndots, const void *const *das, const void *const *dbs,
                          const int *dslots, int nprog, const int *prog, double *table, double *ws, size_t ws_len,
                          void *stream);
/* The Krylov operator (emg3d/solver.py:686-702): out = A x with x = lv->ex|ey|ez (lv->s* unused);
 * entries core.amat_x never touches (upper boundary) are set to zero. */
int emg3d_dev_apply_operator(const emg3d_level *lv, void *ox, void *oy, void *oz, void *stream);
/* zero fill (a kernel of the library: as a hipMemsetAsync node of a captured graph it was seen to
 * race with the neighbouring kernels) / device-to-device hipMemcpyAsync, on the stream */
int emg3d_dev_zero(void *p, size_t bytes, void *stream);
int emg3d_dev_copy(void *dst, const void *src, size_t bytes, void *stream);

/* ---- after a solve: magnetic field and receiver responses ------------
 * fields.get_magnetic_field / _edge_curl_factor (emg3d/fields.py:617-659, 941-1009):
 * m = curl(e) * (zeta / (s mu0)) averaged over the two cells of a face, on the faces
 * mx (nx+1,ny,nz), my (nx,ny+1,nz), mz (nx,ny,nz+1); boundary faces stay 0. zeta = V / mu_r
 * (nx,ny,nz) doubles, hx/hy/hz cell widths -- all device pointers; s mu0 = smu0_re + i smu0_im
 * (real fields: smu0_im ignored). */
int emg3d_dev_magnetic_field(int nx, int ny, int nz, int is_complex, const void *ex, const void *ey,
                             const void *ez, const double *zeta, const double *hx, const double *hy,
                             const double *hz, double smu0_re, double smu0_im, void *mx, void *my,
                             void *mz, void *stream);

/* fields.get_receiver -> maps.interpolate (emg3d/fields.py:522-614, emg3d/maps.py:232-368).
 * method 'cubic' = maps.interp_spline_3d (maps.py:500-552) = scipy.ndimage.map_coordinates(
 * order=3, mode='constant', cval=nan): emg3d_dev_spline_filter turns the n0 x n1 x n2 array
 * (x fastest) IN PLACE into cubic B-spline coefficients (scipy.ndimage.spline_filter, mirror);
 * emg3d_dev_spline_eval evaluates them at npts index-space coordinates coords[0..npts) = x,
 * [npts..2npts) = y, [2npts..3npts) = z (the host maps metres to index space with the same 1-D
 * cubic interpolant as the reference); points outside [0, n-1] give NaN. */
int emg3d_dev_spline_filter(void *data, int n0, int n1, int n2, int is_complex, void *stream);
int emg3d_dev_spline_eval(const void *coef, int n0, int n1, int n2, int is_complex,
                          const double *coords, int npts, void *out, void *stream);
/* method 'linear' = scipy RegularGridInterpolator(fill_value=nan): idx (3 x npts, int32) is the
 * lower corner of the cell holding each point (any entry < 0: outside, NaN), w (3 x npts) the
 * weights of the upper corner. */
int emg3d_dev_linear_eval(const void *values, int n0, int n1, int n2, int is_complex,
                          const int32_t *idx, const double *w, int npts, void *out, void *stream);

/* ---- adjoint-state gradient ---------------------------------------------
 * Simulation.gradient (emg3d/simulations.py:1041-1063) per source-frequency pair:
 * gfield = real(bfield * s mu0 * efield) on the edges, maps.interp_edges_to_vol_averages
 * (emg3d/maps.py:667-719) to the cells, added to the gradient -- one kernel over the cells with
 * the forward field e* and the back-propagated field b* in HBM. volumes (nx,ny,nz), g* (nx,ny,nz)
 * doubles, accumulated (+=): the three components of the reference's (3, nx, ny, nz) array. */
int emg3d_dev_gradient_accumulate(int nx, int ny, int nz, int is_complex, const void *ex, const void *ey,
                                  const void *ez, const void *bx, const void *by, const void *bz,
                                  double smu0_re, double smu0_im, const double *volumes, double *gx,
                                  double *gy, double *gz, void *stream);

/* ---- before a solve: model re-gridding -------------------------------
 * maps.interp_volume_average (emg3d/maps.py:555-616) behind Model.interpolate_to_grid
 * (emg3d/models.py:322-366). values (nx,ny,nz) -> out (mx,my,mz), doubles, x fastest. Per axis
 * the host supplies maps._volume_average_weights (maps.py:619-664) grouped by output cell:
 * seg* (m+1 offsets), w* (segment lengths), in* (input cell of each segment); new_vol = output
 * cell volumes. Same additions in the same order as the reference: bit-identical. log10_scale = 1:
 * the values are averaged on a log10 scale (maps.interpolate(log=True), emg3d/maps.py:346-358).
 * log10_scale = 2: the ADJOINT of the linear averaging (maps._interp_volume_average_adj,
 * emg3d/maps.py:722-750 -- the gradient's way back from a computational grid): the tables are the
 * transposed ones (grouped by ORIGINAL cell = output cell of this call), `values` and `new_vol`
 * both live on the averaged grid (nx,ny,nz), and out (mx,my,mz) is ACCUMULATED:
 * out_i += sum_o overlap_io / new_vol_o * values_o. */
int emg3d_dev_volume_average(const double *values, int nx, int ny, int nz, const int32_t *segx,
                             const int32_t *segy, const int32_t *segz, const double *wx,
                             const double *wy, const double *wz, const int32_t *inx,
                             const int32_t *iny, const int32_t *inz, const double *new_vol, int mx,
                             int my, int mz, double *out, int log10_scale, void *stream);

/* fields.get_source_field -> _dipole_vector (emg3d/fields.py:386-519, 792-938): the source vector
 * of a dipole / wire through `npoints` points (device, npoints x 3, metres), times the complex
 * factor `scale` (strength x (-s mu0)), ADDED to sx|sy|sz with atomic adds -- one thread walks one
 * straight segment from grid plane to grid plane. nodes_* (n + 1) and h* (n): device doubles. */
int emg3d_dev_source_field(int nx, int ny, int nz, int is_complex, const double *nodes_x, const double *nodes_y,
                           const double *nodes_z, const double *hx, const double *hy, const double *hz,
                           const double *points, int npoints, double scale_re, double scale_im, void *sx,
                           void *sy, void *sz, void *stream);

/* models.VolumeModel (emg3d/models.py:654-691) on the device: eta_{x,y,z} = -s mu0 V (sigma [+ s eps0
 * eps_r]) and zeta = V / mu_r from the model's PROPERTY arrays (nx,ny,nz doubles; property_y /
 * property_z / epsilon_r / mu_r may be NULL) and its mapping (0 Conductivity, 1 Resistivity,
 * 2 LgConductivity, 3 LgResistivity, 4 LnConductivity, 5 LnResistivity; emg3d/maps.py:120-330);
 * hx/hy/hz: cell widths (device); s mu0 and s eps0 as (re, im) (real fields: re only). eta_y / eta_z
 * are written only when property_y / property_z are given -- the caller aliases them to eta_x
 * otherwise, as the reference does (emg3d/models.py:693-712). */
int emg3d_dev_volume_model(int nx, int ny, int nz, int is_complex, const double *property_x,
                           const double *property_y, const double *property_z, const double *epsilon_r,
                           const double *mu_r, int mapping, const double *hx, const double *hy,
                           const double *hz, double smu0_re, double smu0_im, double seps0_re,
                           double seps0_im, void *eta_x, void *eta_y, void *eta_z, double *zeta,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EMG3D_AMD_H */

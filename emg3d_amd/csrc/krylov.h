// Vector kernels of the Krylov solvers around the multigrid preconditioner (SURVEY.md section
// 8f, rank 1; reference emg3d/solver.py:652-784, which calls scipy.sparse.linalg.bicgstab / cgs
// on host vectors). On the device a Krylov iteration is a handful of vector updates and inner
// products between the operator / preconditioner applications; they are all instances of ONE
// fused kernel
//
//     y = sum_i c_i x_i   (up to four terms; y may be one of the x_i)
//     [ d_k = conj(a_k) . b_k  for up to three pairs, evaluated with the NEW y ]
//
// whose coefficients c_i are read from a small table of scalars in device memory, followed by
// a one-workgroup kernel that adds up the per-workgroup partial sums (fixed order:
// deterministic), stores them in the table and runs a few scalar instructions on it (alpha =
// rho / (rt . v), beta = ..., sign changes). The scalars of the recurrences therefore never
// leave the GPU between two decisions of the host (convergence and breakdown tests): BiCGSTAB
// needs two table read-backs per iteration instead of a host round trip per inner product.
// Included at the end of kernels.hip (one translation unit).
#pragma once

namespace {

constexpr int KRY_MAX_TERMS = 4, KRY_MAX_DOTS = 3, KRY_MAX_PROG = 8;
constexpr int KRY_GRID = 2048, KRY_BLOCK = 256;

struct KryArgs {
    int nterms;
    const void *x[KRY_MAX_TERMS];
    int slot[KRY_MAX_TERMS];          // coefficient = table[slot] * scale, or just scale if slot < 0
    double scale[KRY_MAX_TERMS];
    void *y;                          // nullptr: no update, inner products only
    int ndots;
    const void *da[KRY_MAX_DOTS], *db[KRY_MAX_DOTS];
    int dslot[KRY_MAX_DOTS];
};
struct KryProg {
    int n;
    int op[KRY_MAX_PROG], dst[KRY_MAX_PROG], a[KRY_MAX_PROG], b[KRY_MAX_PROG];
};

__device__ __forceinline__ cplx kry_conj_mul(cplx a, cplx b) { return cplx(a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re); }
__device__ __forceinline__ double kry_conj_mul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx kry_coef(const double *table, int slot, double scale, cplx)
{
    return slot < 0 ? cplx(scale, 0.0) : cplx(table[2 * slot] * scale, table[2 * slot + 1] * scale);
}
__device__ __forceinline__ double kry_coef(const double *table, int slot, double scale, double)
{
    return slot < 0 ? scale : table[2 * slot] * scale;
}
__device__ __forceinline__ void kry_acc(double (&acc)[2], cplx v) { acc[0] += v.re; acc[1] += v.im; }
__device__ __forceinline__ void kry_acc(double (&acc)[2], double v) { acc[0] += v; }

// partial[(block * KRY_MAX_DOTS + k) * 2 + {0, 1}]: re / im of workgroup `block`'s share of dot k
template <class T>
__global__ __launch_bounds__(KRY_BLOCK) void k_kry_update(KryArgs A, size_t n, const double *table, double *partial)
{
    T c[KRY_MAX_TERMS];
#pragma unroll
    for (int i = 0; i < KRY_MAX_TERMS; ++i) c[i] = i < A.nterms ? kry_coef(table, A.slot[i], A.scale[i], T()) : emg::zero<T>();
    double acc[KRY_MAX_DOTS][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    T *const y = reinterpret_cast<T *>(A.y);
    for (size_t i = (size_t)blockIdx.x * KRY_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * KRY_BLOCK) {
        T v = emg::zero<T>();
        if (y) {
#pragma unroll
            for (int k = 0; k < KRY_MAX_TERMS; ++k)
                if (k < A.nterms) v = emg::mad(c[k], reinterpret_cast<const T *>(A.x[k])[i], v);
            y[i] = v;
        }
#pragma unroll
        for (int k = 0; k < KRY_MAX_DOTS; ++k) {
            if (k < A.ndots) {
                const T a = (y && A.da[k] == A.y) ? v : reinterpret_cast<const T *>(A.da[k])[i];
                const T b = (y && A.db[k] == A.y) ? v : reinterpret_cast<const T *>(A.db[k])[i];
                kry_acc(acc[k], kry_conj_mul(a, b));
            }
        }
    }
    if (A.ndots == 0) return;
    __shared__ double sm[KRY_BLOCK / 64][KRY_MAX_DOTS][2];
#pragma unroll
    for (int k = 0; k < KRY_MAX_DOTS; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double a = acc[k][p];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k][p] = a;
        }
    __syncthreads();
    if (threadIdx.x < KRY_MAX_DOTS * 2) {
        const int k = threadIdx.x >> 1, p = threadIdx.x & 1;
        partial[((size_t)blockIdx.x * KRY_MAX_DOTS + k) * 2 + p] = (sm[0][k][p] + sm[1][k][p]) + (sm[2][k][p] + sm[3][k][p]);
    }
}

// scalar instructions on table slots (complex): dst = a / b, a * b, -a, a
enum { KRY_DIV = 0, KRY_MUL = 1, KRY_NEG = 2, KRY_COPY = 3 };

__global__ __launch_bounds__(KRY_BLOCK) void k_kry_finish(const double *partial, int nblocks, int ndots, KryArgs A, KryProg P,
                                                          double *table)
{
    __shared__ double sm[KRY_BLOCK];
    for (int q = 0; q < ndots * 2; ++q) {           // q = dot k, part p
        const int k = q >> 1, p = q & 1;
        double a = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += KRY_BLOCK) a += partial[((size_t)b * KRY_MAX_DOTS + k) * 2 + p];
        sm[threadIdx.x] = a;
        __syncthreads();
        for (int s = KRY_BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) table[2 * A.dslot[k] + p] = sm[0];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    for (int i = 0; i < P.n; ++i) {
        const cplx a(table[2 * P.a[i]], table[2 * P.a[i] + 1]);
        const cplx b = P.b[i] >= 0 ? cplx(table[2 * P.b[i]], table[2 * P.b[i] + 1]) : cplx(1.0, 0.0);
        cplx r;
        switch (P.op[i]) {
        case KRY_DIV: {      // (a conj b) / |b|^2, as numpy divides complex numbers up to rounding
            const double d = b.re * b.re + b.im * b.im;
            r = cplx((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
            break;
        }
        case KRY_MUL: r = a * b; break;
        case KRY_NEG: r = -a; break;
        default: r = a; break;
        }
        table[2 * P.dst[i]] = r.re;
        table[2 * P.dst[i] + 1] = r.im;
    }
}

// out = A x for a vector x laid out like a field (the Krylov operator of emg3d/solver.py:686-702:
// core.amat_x into a zero field, negated): the residual kernel with the source switched off.
// A workgroup walks `zb` consecutive planes (like k_residual): a thread carries the operands a cell shares with
// the cell below it in registers (stencil.h: residual_load_roll) -- 28 instead of 53 loads per cell, same values.
template <class T>
__global__ __launch_bounds__(256) void k_apply_operator(emg::Level<T> L0, T *ox, T *oy, T *oz, int nzb, int zb)
{
    const int b = blockIdx.z / nzb;
    emg::Level<T> L = emg::source_level(L0, b);
    ox += b * L0.bstride; oy += b * L0.bstride; oz += b * L0.bstride;
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int z0 = (blockIdx.z - b * nzb) * zb, z1 = min(z0 + zb, L.nz + 1);
    if (ix > L.nx || iy > L.ny) return;
    const emg::Axes<T, 0> A(L);
    // r = 0 - A e on the entries core.amat_x touches, 0 elsewhere; A e = -r
    const bool inx = ix < L.nx, iny = iy < L.ny;
    emg::ResIn<T> in;
    bool have = false;
    for (int iz = z0; iz < z1; ++iz) {
        const bool inz = iz < L.nz;
        if (inx && iny && inz) {
            T rx, ry, rz;
            if (have) emg::residual_load_roll<T>(L, ix, iy, iz, in);
            else emg::residual_load<T>(L, ix, iy, iz, in);
            have = true;
            emg::residual_compute<T, false>(L, in, ix, iy, iz, rx, ry, rz);
            ox[A.iex(ix, iy, iz)] = -rx;
            oy[A.iey(ix, iy, iz)] = -ry;
            oz[A.iez(ix, iy, iz)] = -rz;
        } else {
            if (inx) ox[A.iex(ix, iy, iz)] = emg::zero<T>();
            if (iny) oy[A.iey(ix, iy, iz)] = emg::zero<T>();
            if (inz) oz[A.iez(ix, iy, iz)] = emg::zero<T>();
        }
    }
}

// zero fill as a kernel of this library (a node like any other in a captured graph)
__global__ __launch_bounds__(256) void k_fill_zero(double *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.0;
}

}  // namespace

extern "C" {

size_t emg3d_krylov_ws_len(void) { return (size_t)KRY_GRID * KRY_MAX_DOTS * 2; }

int emg3d_dev_krylov_step(size_t n, int is_complex, void *y, int nterms, const void *const *xs, const int *slots,
                          const double *scales, int ndots, const void *const *das, const void *const *dbs,
                          const int *dslots, int nprog, const int *prog, double *table, double *ws, size_t ws_len,
                          void *stream)
{
    if (nterms < 0 || nterms > KRY_MAX_TERMS || ndots < 0 || ndots > KRY_MAX_DOTS || nprog < 0 || nprog > KRY_MAX_PROG ||
        !table || (nterms > 0 && !y))
        return fail(EMG3D_ERR_BADARG, "krylov_step: bad argument");
    if (ndots > 0 && (!ws || ws_len < emg3d_krylov_ws_len())) return fail(EMG3D_ERR_SCRATCH, "krylov_step: workspace too small");
    KryArgs A = {};
    A.nterms = nterms;
    A.y = nterms > 0 ? y : nullptr;
    for (int i = 0; i < nterms; ++i) { A.x[i] = xs[i]; A.slot[i] = slots[i]; A.scale[i] = scales[i]; }
    A.ndots = ndots;
    for (int k = 0; k < ndots; ++k) { A.da[k] = das[k]; A.db[k] = dbs[k]; A.dslot[k] = dslots[k]; }
    KryProg P = {};
    P.n = nprog;
    for (int i = 0; i < nprog; ++i) { P.op[i] = prog[4 * i]; P.dst[i] = prog[4 * i + 1]; P.a[i] = prog[4 * i + 2]; P.b[i] = prog[4 * i + 3]; }
    const hipStream_t st = (hipStream_t)stream;
    size_t want = (n + KRY_BLOCK - 1) / KRY_BLOCK;
    const int grid = (int)(want < 1 ? 1 : (want > (size_t)KRY_GRID ? (size_t)KRY_GRID : want));
    if (nterms > 0 || ndots > 0) {
        if (is_complex) hipLaunchKernelGGL(k_kry_update<cplx>, dim3(grid), dim3(KRY_BLOCK), 0, st, A, n, (const double *)table, ws);
        else hipLaunchKernelGGL(k_kry_update<double>, dim3(grid), dim3(KRY_BLOCK), 0, st, A, n, (const double *)table, ws);
    }
    if (ndots > 0 || nprog > 0)
        hipLaunchKernelGGL(k_kry_finish, dim3(1), dim3(KRY_BLOCK), 0, st, (const double *)ws, grid, ndots, A, P, table);
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_apply_operator(const emg3d_level *lv, void *ox, void *oy, void *oz, void *stream)
{
    if (!lv || !ox || !oy || !oz) return fail(EMG3D_ERR_BADARG, "apply_operator: bad argument");
    const dim3 block = d3(emg::cell_block());
    dim3 grid = d3(emg::cell_grid(lv->nx + 1, lv->ny + 1, lv->nz + 1));
    const int batch = lv->batch > 1 ? lv->batch : 1;
    // planes per workgroup: as in the residual kernel (8 where that still leaves every CU several workgroups)
    const int zb = (g_residual_roll && (size_t)grid.x * grid.y * (grid.z / g_residual_zb) >= 8u * (unsigned)compute_units())
                       ? g_residual_zb : 1;
    grid.z = cdiv((int)grid.z, zb);
    if (lv->is_complex)
        hipLaunchKernelGGL(k_apply_operator<cplx>, dim3(grid.x, grid.y, grid.z * batch), block, 0, (hipStream_t)stream,
                           to_level<cplx>(lv), (cplx *)ox, (cplx *)oy, (cplx *)oz, (int)grid.z, zb);
    else
        hipLaunchKernelGGL(k_apply_operator<double>, dim3(grid.x, grid.y, grid.z * batch), block, 0, (hipStream_t)stream,
                           to_level<double>(lv), (double *)ox, (double *)oy, (double *)oz, (int)grid.z, zb);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* plain device memory helpers, so that no tensor-library operation touches field-sized data */
int emg3d_dev_zero(void *p, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    if (bytes % 8 != 0 || ((size_t)p & 7) != 0) return fail(EMG3D_ERR_BADARG, "zero: buffer of doubles expected");
    const size_t n = bytes / 8;
    size_t want = (n + 255) / 256;
    const int grid = (int)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(k_fill_zero, dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)p, n);
    HIP_TRY(hipGetLastError());
    return 0;
}
int emg3d_dev_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

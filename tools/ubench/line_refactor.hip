// Bound measurement for a line smoother WITHOUT stored factors: the
// level-0 colour pass of core.gauss_seidel_y/_x/_z (emg3d/core.py:506-1348) with the block
// factorisation recomputed on the fly -- the reference's own arithmetic: assemble M_k, B_k from
// eta / zeta / h (stencil.h: line_matrix), Schur update S_k = M_k - B_k S_{k-1}^{-1} B_k^T, LDL^T
// of S_k, substitution of the right-hand side (core.py:632-772, 1481-1616) -- by ONE LANE PER
// HALF-LINE (two-sided elimination: 2 x 16384 half-lines per colour class at 256^3 = 512 waves).
//
// It is a timing prototype, not a solver: the bottom half is timed as a second top-down chain
// over blocks [n0/2, n0) (same arithmetic as the mirrored recurrence), there is no middle block.
// Phases that can be timed separately (MODE):
//   0  chain only:            matrix assembly + Schur update + LDL^T          (no right-hand side)
//   1  + right-hand side assembly (line_rhs) and its forward substitution    (nothing stored)
//   2  + the LDL^T factors and w_k of every block stored for a backward pass (15 + 5 complex)
//   backward: reads them back, couples with B_{k+1}^T (from zeta), substitutes, scatters to the field
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 line_refactor.hip -o line_refactor && ./line_refactor [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../emg3d_amd/csrc/launch.h"

using emg::cplx;
using T = cplx;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// trailing 4 x 4 block X = (S^{-1})[1:5,1:5] from the LDL^T factors of S: S^{-1} = Z^T D^{-1} Z with
// Z = L^{-1} (unit lower triangular); only the columns 1..4 of Z are needed. 30 complex operations
// (stencil.h: sub_lower_coupling does five full solves, 125).
__device__ __forceinline__ void trailing_inverse(const T (&C)[10], const T (&dinv)[5], T (&X)[10])
{
    using emg::tri;
    // Z(r,m), 1 <= m < r <= 4:  Z(r,m) = -(C(r,m) + sum_{m<k<r} C(r,k) Z(k,m))
    const T z21 = -C[tri(2, 1)];
    const T z32 = -C[tri(3, 2)];
    const T z43 = -C[tri(4, 3)];
    const T z31 = -(emg::mad(C[tri(3, 2)], z21, C[tri(3, 1)]));
    const T z42 = -(emg::mad(C[tri(4, 3)], z32, C[tri(4, 2)]));
    const T z41 = -(emg::mad(C[tri(4, 3)], z31, emg::mad(C[tri(4, 2)], z21, C[tri(4, 1)])));
    // W = D^{-1} Z (rows k, columns m)
    const T w21 = dinv[2] * z21, w31 = dinv[3] * z31, w32 = dinv[3] * z32;
    const T w41 = dinv[4] * z41, w42 = dinv[4] * z42, w43 = dinv[4] * z43;
    // X(a,b) = sum_{k >= a} Z(k,a) W(k,b), a >= b >= 1 (Z(a,a) = 1); packed X[tri(a,b)-?]: index (a-1)(a)/2 + (b-1)
    auto ix = [](int a, int b) { return (a - 1) * a / 2 + (b - 1); };
    X[ix(4, 4)] = dinv[4];
    X[ix(4, 3)] = w43;
    X[ix(4, 2)] = w42;
    X[ix(4, 1)] = w41;
    X[ix(3, 3)] = emg::mad(z43, w43, dinv[3]);
    X[ix(3, 2)] = emg::mad(z43, w42, w32);
    X[ix(3, 1)] = emg::mad(z43, w41, w31);
    X[ix(2, 2)] = emg::mad(z42, w42, emg::mad(z32, w32, dinv[2]));
    X[ix(2, 1)] = emg::mad(z42, w41, emg::mad(z32, w31, w21));
    X[ix(1, 1)] = emg::mad(z41, w41, emg::mad(z31, w31, emg::mad(z21, w21, dinv[1])));
}

// S -= B X B^T for B = e0 l0^T + diag(0, d) (lower triangle of S), X packed as above
__device__ __forceinline__ void schur_update(T (&S)[5][5], const T (&X)[10], const double (&l0)[5], const double (&d)[5])
{
    auto ix = [](int a, int b) { return a >= b ? (a - 1) * a / 2 + (b - 1) : (b - 1) * b / 2 + (a - 1); };
    T u[5];                                 // u = X l0 (entries 1..4)
#pragma unroll
    for (int a = 1; a < 5; ++a) {
        T acc = l0[1] * X[ix(a, 1)];
#pragma unroll
        for (int b = 2; b < 5; ++b) acc = emg::mad(l0[b], X[ix(a, b)], acc);
        u[a] = acc;
    }
    T s00 = l0[1] * u[1];
#pragma unroll
    for (int a = 2; a < 5; ++a) s00 = emg::mad(l0[a], u[a], s00);
    S[0][0] -= s00;
#pragma unroll
    for (int a = 1; a < 5; ++a) {
        S[a][0] = emg::nmad(d[a], u[a], S[a][0]);
#pragma unroll
        for (int b = 1; b <= a; ++b) S[a][b] = emg::nmad(d[a] * d[b], X[ix(a, b)], S[a][b]);
    }
}

// coupling block B_k alone (left0, leftd of stencil.h: line_matrix), from zeta and the widths
template <int DIR>
__device__ __forceinline__ void line_coupling(const emg::Axes<T, DIR> &A, int k, int i1, int i2, double (&l0)[5], double (&ld)[5])
{
    const int i0m = k, i1m = i1 - 1, i2m = i2 - 1;
    const double h00 = A.ih0()[i0m], h10 = A.ih1()[i1m], h11 = A.ih1()[i1], h20 = A.ih2()[i2m], h21 = A.ih2()[i2];
    const double z000 = A.zeta(i0m, i1m, i2m), z010 = A.zeta(i0m, i1, i2m), z001 = A.zeta(i0m, i1m, i2), z011 = A.zeta(i0m, i1, i2);
    const double k00 = 0.5 * h00;
    l0[0] = 0.0;
    l0[1] = 0.5 * h10 * (z001 + z000) * h00;
    l0[2] = -0.5 * h11 * (z011 + z010) * h00;
    l0[3] = 0.5 * h20 * (z010 + z000) * h00;
    l0[4] = -0.5 * h21 * (z011 + z001) * h00;
    ld[0] = 0.0;
    ld[1] = -k00 * (z001 + z000) * h00;
    ld[2] = -k00 * (z011 + z010) * h00;
    ld[3] = -k00 * (z010 + z000) * h00;
    ld[4] = -k00 * (z011 + z001) * h00;
}

template <int DIR, int MODE>
__global__ __launch_bounds__(64) void k_forward(emg::Level<T> L, int colour, int cntp, int cntq, T *store, size_t nhalf, T *sink)
{
    const emg::Axes<T, DIR> A(L);
    int i1, i2, lid;
    const int tp = blockIdx.x * 64 + threadIdx.x;
    const bool valid = emg::line_of_thread<DIR>(colour, cntp, cntq, min(tp, cntp - 1), blockIdx.y, i1, i2, lid);
    (void)valid;
    const int half = blockIdx.z, n0 = A.n0();
    const int kb = half ? n0 / 2 : 0, ke = half ? n0 : n0 / 2;
    const size_t col = (size_t)half * (nhalf / 2) + lid;
    T C[10], dinv[5], w[5];
#pragma unroll
    for (int j = 0; j < 10; ++j) C[j] = emg::zero<T>();
#pragma unroll
    for (int j = 0; j < 5; ++j) { dinv[j] = T(1.0); w[j] = emg::zero<T>(); }
    for (int k = kb; k < ke; ++k) {
        T dg[5];
        double mid[5][5], left0[5], leftd[5];
        emg::line_matrix<T, DIR>(A, k, i1, i2, dg, mid, left0, leftd);
        T S[5][5];
        emg::std_block<T>(dg, mid, S);
        if (k > kb) {
            T X[10];
            trailing_inverse(C, dinv, X);
            schur_update(S, X, left0, leftd);
        }
        emg::ldlt5<T>(S, 5, C, dinv);
        if (MODE >= 1) {
            T rhs[5];
            emg::line_rhs<T, DIR>(A, k, i1, i2, rhs);
            // c = rhs - B w_prev ;  w = S^{-1} c
            T q0 = emg::zero<T>();
#pragma unroll
            for (int m = 1; m < 5; ++m) {
                q0 = emg::mad(left0[m], w[m], q0);
                rhs[m] = emg::nmad(leftd[m], w[m], rhs[m]);
            }
            rhs[0] -= q0;
            emg::ldlt5_solve<T>(C, dinv, rhs);
#pragma unroll
            for (int j = 0; j < 5; ++j) w[j] = rhs[j];
        }
        if (MODE >= 2) {
            T *o = store + ((size_t)k - kb) * 20 * nhalf + col;
#pragma unroll
            for (int j = 0; j < 10; ++j) o[(size_t)j * nhalf] = C[j];
#pragma unroll
            for (int j = 0; j < 5; ++j) o[(size_t)(10 + j) * nhalf] = dinv[j];
#pragma unroll
            for (int j = 0; j < 5; ++j) o[(size_t)(15 + j) * nhalf] = w[j];
        }
    }
    // keep the chain alive
    T acc = dinv[0] + dinv[4] + C[9] + w[0] + w[4];
    if (acc.re == 12345.678) sink[col] = acc;
}

template <int DIR>
__global__ __launch_bounds__(64) void k_backward(emg::Level<T> L, int colour, int cntp, int cntq, const T *store, size_t nhalf)
{
    const emg::Axes<T, DIR> A(L);
    int i1, i2, lid;
    const int tp = blockIdx.x * 64 + threadIdx.x;
    const bool valid = emg::line_of_thread<DIR>(colour, cntp, cntq, min(tp, cntp - 1), blockIdx.y, i1, i2, lid);
    const int half = blockIdx.z, n0 = A.n0();
    const int kb = half ? n0 / 2 : 0, ke = half ? n0 : n0 / 2;
    const size_t col = (size_t)half * (nhalf / 2) + lid;
    T x[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) x[j] = emg::zero<T>();
    double l0[5], ld[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) l0[j] = ld[j] = 0.0;
    for (int k = ke - 1; k >= kb; --k) {
        const T *o = store + ((size_t)k - kb) * 20 * nhalf + col;
        T C[10], dinv[5], w[5];
#pragma unroll
        for (int j = 0; j < 10; ++j) C[j] = o[(size_t)j * nhalf];
#pragma unroll
        for (int j = 0; j < 5; ++j) dinv[j] = o[(size_t)(10 + j) * nhalf];
#pragma unroll
        for (int j = 0; j < 5; ++j) w[j] = o[(size_t)(15 + j) * nhalf];
        // q = B_{k+1}^T x_{k+1}: q_0 = 0, q_m = l0_m x_0 + ld_m x_m
        T q[5];
        q[0] = emg::zero<T>();
#pragma unroll
        for (int m = 1; m < 5; ++m) q[m] = emg::mad(ld[m], x[m], l0[m] * x[0]);
        emg::ldlt5_solve<T>(C, dinv, q);
#pragma unroll
        for (int j = 0; j < 5; ++j) x[j] = w[j] - q[j];
        if (valid && tp < cntp) emg::line_scatter<T, DIR>(A, k, i1, i2, x);
        line_coupling<DIR>(A, k, i1, i2, l0, ld);
    }
}

__global__ void k_init(double *p, size_t n, double lo, double hi, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        p[i] = lo + (hi - lo) * (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}
__global__ void k_init_eta(T *p, size_t n, double scale, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        p[i] = T(0.0, -scale * (0.03 + 3.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0)));
    }
}

template <class F> float timeit(F f, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int DIR> void run_dir(const emg::Level<T> &L, int n, T *store, T *sink)
{
    const int colour = 3;
    const emg::LineClass c = emg::line_class(DIR, n, n, n, colour);
    const size_t nhalf = 2 * (size_t)c.lines;
    const dim3 grid(emg::cdiv(c.cntp, 64), c.cntq, 2), block(64);
    const double blocks = (double)c.lines * c.n0;
    const double alg = 200.0 * blocks;         // algorithmic bytes of one colour pass (tri-axial, complex)
    auto report = [&](const char *what, float ms, double bytes_per_block) {
        printf("  %-58s %7.3f ms  = %5.1f %% of 8 TB/s on the pass's algorithmic bytes; nominal traffic %4.0f B/block -> %5.2f TB/s\n",
               what, ms, 100.0 * alg / (ms * 1e-3) / 8e12, bytes_per_block, bytes_per_block * blocks / (ms * 1e-3) / 1e12);
    };
    printf("%c-lines, %d^3, colour class %d: %d lines x %d blocks, %u workgroups of one wave (lane = half-line)\n",
           "xyz"[DIR], n, colour, c.lines, c.n0, grid.x * grid.y * grid.z);
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_forward<DIR, 2>)));
    printf("  registers: forward (mode 2) %d", fa.numRegs);
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_backward<DIR>)));
    printf(", backward %d; scratch %zu B\n", fa.numRegs, (size_t)fa.localSizeBytes);
    const float t0 = timeit([&] { hipLaunchKernelGGL((k_forward<DIR, 0>), grid, block, 0, 0, L, colour, c.cntp, c.cntq, store, nhalf, sink); });
    report("forward, chain only (assembly + Schur update + LDL^T)", t0, 224);
    const float t1 = timeit([&] { hipLaunchKernelGGL((k_forward<DIR, 1>), grid, block, 0, 0, L, colour, c.cntp, c.cntq, store, nhalf, sink); });
    report("forward + right-hand side + substitution, nothing stored", t1, 224 + 80 + 192);
    const float t2 = timeit([&] { hipLaunchKernelGGL((k_forward<DIR, 2>), grid, block, 0, 0, L, colour, c.cntp, c.cntq, store, nhalf, sink); });
    report("forward, factors + w stored (320 B per block)", t2, 224 + 80 + 192 + 320);
    const float t3 = timeit([&] { hipLaunchKernelGGL((k_backward<DIR>), grid, block, 0, 0, L, colour, c.cntp, c.cntq, store, nhalf); });
    report("backward (factors + w read, coupling from zeta, scatter)", t3, 320 + 32 + 80);
    report("forward (stored) + backward = one colour pass", t2 + t3, 224 + 80 + 192 + 320 + 320 + 32 + 80);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const size_t ncell = (size_t)n * n * n;
    const size_t nex = (size_t)n * (n + 1) * (n + 1);
    T *e, *s, *eta[3], *store, *sink;
    double *zeta, *ih;
    CK(hipMalloc(&e, 3 * nex * sizeof(T)));
    CK(hipMalloc(&s, 3 * nex * sizeof(T)));
    for (int c = 0; c < 3; ++c) CK(hipMalloc(&eta[c], ncell * sizeof(T)));
    CK(hipMalloc(&zeta, ncell * 8));
    CK(hipMalloc(&ih, 3 * n * 8));
    const size_t lines = (size_t)(n / 2) * (n / 2);
    CK(hipMalloc(&store, lines * 2 * (n / 2) * 20 * sizeof(T)));
    CK(hipMalloc(&sink, lines * 2 * sizeof(T)));
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(e), 6 * nex, -1.0, 1.0, 1u);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(s), 6 * nex, -1.0, 1.0, 2u);
    // h = 25 m (stretched a little), zeta = V, eta = -i omega mu0 sigma V at 1 Hz
    hipLaunchKernelGGL(k_init, dim3(16), dim3(256), 0, 0, ih, (size_t)3 * n, 1.0 / 40.0, 1.0 / 25.0, 3u);
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, zeta, ncell, 15625.0, 64000.0, 4u);
    for (int c = 0; c < 3; ++c)
        hipLaunchKernelGGL(k_init_eta, dim3(4096), dim3(256), 0, 0, eta[c], ncell, 7.9e-6 * 30000.0 / (1.0 + c), 5u + c);
    CK(hipDeviceSynchronize());
    emg::Level<T> L;
    L.nx = L.ny = L.nz = n;
    L.ex = e; L.ey = e + nex; L.ez = e + 2 * nex;
    L.sx = s; L.sy = s + nex; L.sz = s + 2 * nex;
    L.eta_x = eta[0]; L.eta_y = eta[1]; L.eta_z = eta[2];
    L.zeta = zeta;
    L.ihx = ih; L.ihy = ih + n; L.ihz = ih + 2 * n;
    run_dir<1>(L, n, store, sink);
    run_dir<2>(L, n, store, sink);
    run_dir<0>(L, n, store, sink);
    return 0;
}

// Issue cost of the chain loops' factor fetches (one wave64, data resident in L2): the packed-record pattern of the
// library (lane j of quad q reads six 16-byte entries of record q, records 240 B apart, + four doubles of a
// 64-byte coupling record) against a wave-major layout (64 consecutive 16-byte entries per load instruction).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/load_rates.hip -o tools/ubench/load_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define NSTEP 1024
template <int MODE> __global__ void k(const double2 *fac, const double *lfac, double *out, unsigned long long *cyc, int nlines)
{
    const int t = threadIdx.x, j = t & 3, q = t >> 2;
    double2 acc = {0.0, 0.0};
    double accd = 0.0;
    const size_t frow = (size_t)nlines * 15, lrow = (size_t)nlines * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k0 = 0; k0 < NSTEP; k0 += 4) {
        double2 r[4][6];
        double l[4][4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = (k0 + d) & 63;            // 64 block rows: 64 x 16 lines x 240 B = 245 KB, L2-resident
            if (MODE == 0) {
                const double2 *f = fac + (size_t)k * frow + q * 15;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int m = (j + e) & 3; r[d][e] = f[j >= m ? j * (j + 1) / 2 + m : m * (m + 1) / 2 + j]; }
                r[d][4] = f[10 + j];
                r[d][5] = f[14];
                const double *lf = lfac + (size_t)k * lrow + q * 8;
                l[d][0] = lf[j == 0 ? 3 : j - 1]; l[d][1] = lf[4 + (j > 1 ? j : 1) - 1]; l[d][2] = lf[3]; l[d][3] = lf[7];
            } else if (MODE == 1) {
                const double2 *f = fac + (size_t)k * 64 * 6;
#pragma unroll
                for (int e = 0; e < 6; ++e) r[d][e] = f[e * 64 + t];
                const double *lf = lfac + (size_t)k * 64 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) l[d][e] = lf[e * 64 + t];
            } else {      // wave-major, coupling entries as two 16-byte loads
                const double2 *f = fac + (size_t)k * 64 * 8;
#pragma unroll
                for (int e = 0; e < 6; ++e) r[d][e] = f[e * 64 + t];
                const double2 a = f[6 * 64 + t], b = f[7 * 64 + t];
                l[d][0] = a.x; l[d][1] = a.y; l[d][2] = b.x; l[d][3] = b.y;
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int e = 0; e < 6; ++e) { acc.x += r[d][e].x; acc.y += r[d][e].y; }
#pragma unroll
            for (int e = 0; e < 4; ++e) accd += l[d][e];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[t] = acc.x + acc.y + accd;
    if (t == 0) cyc[0] = t1 - t0;
}
int main()
{
    const int nlines = 16;
    double2 *fac; double *lfac, *out; unsigned long long *cyc, h;
    (void)hipMalloc(&fac, 64 * 64 * 8 * 16 * 2); (void)hipMalloc(&lfac, 64 * 64 * 8 * 8); (void)hipMalloc(&out, 512); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(fac, 0, 64 * 64 * 8 * 16 * 2); (void)hipMemset(lfac, 0, 64 * 64 * 8 * 8);
    const char *names[3] = {"packed records (library layout)", "wave-major, 6 x 16 B + 4 x 8 B", "wave-major, 8 x 16 B"};
    for (int m = 0; m < 3; ++m) {
        for (int r = 0; r < 3; ++r) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, fac, lfac, out, cyc, nlines);
            else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, fac, lfac, out, cyc, nlines);
            else hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, fac, lfac, out, cyc, nlines);
            (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-34s %7.1f ticks per step (10 / 10 / 8 load instructions + 16 fp64 adds)\n", names[m], (double)h / NSTEP);
    }
    return 0;
}

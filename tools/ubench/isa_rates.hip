// Micro-benchmark: cycles per wave64 instruction on gfx950 for the instruction classes of the
// line-smoother block step, one wave per workgroup (as in k_line_colour's chain waves).
//   hipcc --offload-arch=gfx950 -O3 isa_rates.hip -o isa_rates && ./isa_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 256
template <int MODE> __global__ void k(double *out, unsigned long long *cyc, double a_, double b_, int active)
{
    if ((int)threadIdx.x >= active) return;      // partly filled wave: does the SIMD skip the empty quarter-passes?
    // per-lane operands: with wave-uniform (SGPR) operands a VOP3 instruction may need extra
    // moves (one constant-bus read per instruction on gfx9) and the count is off
    const double a = a_ + 1e-13 * threadIdx.x, b = b_ + 1e-13 * threadIdx.x;
    double x0 = a + threadIdx.x, x1 = b + threadIdx.x, x2 = a * 2 + threadIdx.x, x3 = b * 3 + threadIdx.x;
    double y0 = x0 + 1, y1 = x1 + 1, y2 = x2 + 1, y3 = x3 + 1;
    const unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) {          // 8 dependent fma (one chain)
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = x0 * a + b;
        } else if (MODE == 1) {   // 8 independent fma (8 chains)
            x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
            y0 = y0 * a + b; y1 = y1 * a + b; y2 = y2 * a + b; y3 = y3 * a + b;
        } else if (MODE == 2) {   // 8 dependent dpp moves of a double (16 v_mov_b32_dpp)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int lo = __builtin_amdgcn_mov_dpp(__double2loint(x0), 0x39, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x0), 0x39, 0xf, 0xf, true);
                x0 = __hiloint2double(hi, lo);
            }
        } else if (MODE == 3) {   // 8 independent dpp moves of doubles (16 v_mov_b32_dpp)
            double *p[8] = {&x0, &x1, &x2, &x3, &y0, &y1, &y2, &y3};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int lo = __builtin_amdgcn_mov_dpp(__double2loint(*p[u]), 0x39, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_mov_dpp(__double2hiint(*p[u]), 0x39, 0xf, 0xf, true);
                *p[u] = __hiloint2double(hi, lo);
            }
        } else if (MODE == 4) {   // fma -> dpp -> fma -> dpp dependent (4 fma + 8 dpp movs)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0 = x0 * a + b;
                int lo = __builtin_amdgcn_mov_dpp(__double2loint(x0), 0x39, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x0), 0x39, 0xf, 0xf, true);
                x0 = __hiloint2double(hi, lo);
            }
        } else if (MODE == 5) {   // 8 dependent mul_f64
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = x0 * a;
        } else if (MODE == 6) {   // 8 dependent add_f64
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = x0 + a;
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int per_iter, int active = 64)
{
    double *out; unsigned long long *cyc, h[4];
    hipMalloc(&out, 4 * 64 * 8); hipMalloc(&cyc, 4 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(4), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9, active);
    hipLaunchKernelGGL(k<MODE>, dim3(4), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9, active);
    hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-52s %2d lanes %7.2f clock64 ticks per instruction\n", name, active, (double)h[0] / (N * per_iter));
    hipFree(out); hipFree(cyc);
}

__global__ void kcal(unsigned long long *o)
{
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    double x = threadIdx.x;
    for (int i = 0; i < 200000; ++i) x = x * 1.0000001 + 1e-9;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; o[2] = (unsigned long long)x; }
}

int main()
{
    {   // clock64 ticks per second, from the constant 100 MHz wall clock
        unsigned long long *d, h[3];
        (void)hipMalloc(&d, 24);
        hipLaunchKernelGGL(kcal, dim3(1), dim3(64), 0, 0, d);
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("clock64: %.1f MHz (%llu ticks in %llu wall ticks of 10 ns); the 200000 dependent v_fma_f64 took %.2f ticks each\n",
               (double)h[0] / ((double)h[1] * 1e-8) / 1e6, h[0], h[1], (double)h[0] / 200000.0);
    }
    run<0>("v_fma_f64, dependent chain", 8);
    run<1>("v_fma_f64, 8 independent chains", 8);
    run<5>("v_mul_f64, dependent chain", 8);
    run<6>("v_add_f64, dependent chain", 8);
    run<2>("v_mov_b32_dpp quad_perm, dependent (per mov)", 16);
    run<3>("v_mov_b32_dpp quad_perm, independent (per mov)", 16);
    run<4>("fma -> dpp(lo,hi) -> fma ... (per instruction)", 12);
    for (int active : {32, 16}) {
        run<0>("v_fma_f64, dependent chain", 8, active);
        run<1>("v_fma_f64, 8 independent chains", 8, active);
        run<3>("v_mov_b32_dpp quad_perm, independent (per mov)", 16, active);
        run<4>("fma -> dpp(lo,hi) -> fma ... (per instruction)", 12, active);
    }
    return 0;
}

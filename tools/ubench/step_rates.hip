// Issue time of the chain-step FUNCTIONS alone (no memory traffic): one wave64, operands in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/step_rates.hip -o tools/ubench/step_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../emg3d_amd/csrc/cplx.h"
using emg::cplx;
#pragma clang fp contract(off)
namespace xop {
__device__ __forceinline__ double mul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx mul(double a, cplx b) { return cplx(a * b.re, a * b.im); }
__device__ __forceinline__ cplx mul(cplx a, cplx b) { return cplx(__builtin_fma(-a.im, b.im, a.re * b.re), __builtin_fma(a.im, b.re, a.re * b.im)); }
__device__ __forceinline__ cplx add(cplx a, cplx b) { return cplx(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx sub(cplx a, cplx b) { return cplx(a.re - b.re, a.im - b.im); }
}
template <int CTRL> __device__ __forceinline__ double dpp_move(double x)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ cplx dpp_move(cplx x) { return cplx(dpp_move<CTRL>(x.re), dpp_move<CTRL>(x.im)); }
template <int LN, class T> __device__ __forceinline__ T quad_bcast(T x) { return dpp_move<LN * 0x55>(x); }
template <class T> __device__ __forceinline__ T quad_sum(T x)
{
    x = xop::add(x, dpp_move<0xB1>(x));
    x = xop::add(x, dpp_move<0x4E>(x));
    return x;
}
template <int R, class T> __device__ __forceinline__ T quad_rot(T x) { return dpp_move<R == 1 ? 0x39 : (R == 2 ? 0x4E : 0x93)>(x); }
struct Row { cplx t[5], t44; double bA, bD, b04, d4; };
__device__ __forceinline__ void fwd_step(const Row &q, const cplx v, const cplx v4, const double nz, const double is0, cplx &wsel, cplx &w4p, cplx &wn, cplx &w4)
{
    const cplx rowsum = quad_sum(xop::mul(q.bA, wsel));
    const cplx cj = emg::nmad(xop::mul(q.bD, nz), wsel, emg::nmad(is0, rowsum, v));
    const cplx c4 = emg::nmad(q.d4, w4p, v4);
    const cplx c1 = quad_rot<1>(cj), c2 = quad_rot<2>(cj), c3 = quad_rot<3>(cj);
    wn = xop::add(emg::mad(q.t[4], c4, emg::mad(q.t[1], c1, xop::mul(q.t[0], cj))), emg::mad(q.t[3], c3, xop::mul(q.t[2], c2)));
    w4 = emg::mad(q.t44, c4, quad_sum(xop::mul(q.t[4], cj)));
    wsel = emg::mad(nz, wn, xop::mul(is0, w4));
    w4p = w4;
}
__device__ __forceinline__ void bwd_step(const Row &q, const cplx wj, const cplx w4, const double nz, const double upA, const double upD, const double up04, const double up44, cplx &x0, cplx &x4, cplx &xmine, cplx &xn, cplx &xn4)
{
    const cplx hj = emg::mad(xop::mul(upA, nz), x0, xop::mul(xop::mul(upD, nz), xmine));
    const cplx h4 = emg::mad(up04, x0, xop::mul(up44, x4));
    const cplx h1 = quad_rot<1>(hj), h2 = quad_rot<2>(hj), h3 = quad_rot<3>(hj);
    xn = xop::sub(emg::nmad(q.t[4], h4, emg::nmad(q.t[1], h1, emg::nmad(q.t[0], hj, wj))), emg::mad(q.t[3], h3, xop::mul(q.t[2], h2)));
    xn4 = xop::sub(emg::nmad(q.t44, h4, w4), quad_sum(xop::mul(q.t[4], hj)));
    x0 = quad_bcast<0>(xn);
    x4 = xn4;
    xmine = xn;
}
#define NSTEP 512
template <int MODE> __global__ void k(const double *in, double *out, unsigned long long *cyc)
{
    Row q[4];
    const int t = threadIdx.x;
    for (int d = 0; d < 4; ++d) {
        for (int r = 0; r < 5; ++r) q[d].t[r] = cplx(in[(d * 8 + r) * 64 + t] * 1e-3, in[(d * 8 + r + 1) * 64 + t] * 1e-3);
        q[d].t44 = cplx(in[(d * 8 + 6) * 64 + t] * 1e-3, 0.1);
        q[d].bA = in[(d * 8 + 7) * 64 + t] * 1e-3; q[d].bD = q[d].bA * 0.5; q[d].b04 = q[d].bA * 0.25; q[d].d4 = q[d].bA * 0.125;
    }
    cplx v(in[t], in[64 + t]), v4(in[128 + t], in[192 + t]);
    const int j = t & 3;
    const double nz = j != 0 ? 1.0 : 0.0, is0 = 1.0 - nz;
    cplx a = cplx(0.0, 0.0), b = cplx(0.0, 0.0), c = cplx(0.1, 0.2), o1, o2, acc(0.0, 0.0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < NSTEP; i += 4) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (MODE == 0) fwd_step(q[d], v, v4, nz, is0, a, b, o1, o2);
            else bwd_step(q[d], v, v4, nz, q[d].bA, q[d].bD, q[d].b04, q[d].d4, a, b, c, o1, o2);
            acc = xop::add(acc, xop::add(o1, o2));      // (keeps the outputs alive: 4 more fp64 per step)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[t] = acc.re + acc.im + a.re + b.im + c.re;
    if (t == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *in, *out; unsigned long long *cyc, h;
    (void)hipMalloc(&in, 64 * 64 * 8); (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    double hin[64 * 64];
    for (int i = 0; i < 64 * 64; ++i) hin[i] = 0.3 + 0.001 * (i % 97);
    (void)hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, in, out, cyc); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("forward step : %.1f ticks per step (incl. 4 fp64 of the harness)\n", (double)h / NSTEP);
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, in, out, cyc); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("backward step: %.1f ticks per step (incl. 4 fp64 of the harness)\n", (double)h / NSTEP);
    return 0;
}

// fp64 issue rates of one wave64 on gfx950 as a function of WHERE the operands live (register banks, SGPR,
// inline constants):  hipcc --offload-arch=gfx950 -O3 tools/ubench/isa_rates.hip -o tools/ubench/isa_rates
// Explicit registers: a VGPR pair v[2k:2k+1] starts in bank (2k) & 3 = 0 or 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
// 16 instructions per body line, REP bodies
#define K(NAME, BODY)                                                                                         \
    __global__ void NAME(unsigned long long *cyc, double *out)                                                \
    {                                                                                                         \
        unsigned long long t0, t1;                                                                            \
        asm volatile("v_cvt_f64_u32 v[4:5], v0\n v_cvt_f64_u32 v[6:7], v0\n v_cvt_f64_u32 v[8:9], v0\n"       \
                     "v_cvt_f64_u32 v[10:11], v0\n v_cvt_f64_u32 v[12:13], v0\n v_cvt_f64_u32 v[14:15], v0\n" \
                     "v_cvt_f64_u32 v[16:17], v0\n v_cvt_f64_u32 v[18:19], v0\n v_cvt_f64_u32 v[20:21], v0\n" \
                     "v_cvt_f64_u32 v[22:23], v0\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0x80000000\n" ::: "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s20", "s21"); \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));               \
        _Pragma("unroll") for (int i = 0; i < REP; ++i) asm volatile(BODY ::: "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"); \
        asm volatile("s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));                   \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                               \
        double r;                                                                                             \
        asm volatile("v_add_f64 %0, v[12:13], v[14:15]" : "=v"(r));                                           \
        out[threadIdx.x] = r;                                                                                 \
    }
#define X8(a) a a a a a a a a
// independent instructions (4 destinations round-robin), different source-bank layouts
K(mul_b02, X8("v_mul_f64 v[12:13], v[4:5], v[6:7]\n v_mul_f64 v[14:15], v[4:5], v[6:7]\n"))          // sources in banks 0 and 2
K(mul_b00, X8("v_mul_f64 v[12:13], v[4:5], v[8:9]\n v_mul_f64 v[14:15], v[4:5], v[8:9]\n"))          // both in bank 0
K(add_b02, X8("v_add_f64 v[12:13], v[4:5], v[6:7]\n v_add_f64 v[14:15], v[4:5], v[6:7]\n"))
K(add_b00, X8("v_add_f64 v[12:13], v[4:5], v[8:9]\n v_add_f64 v[14:15], v[4:5], v[8:9]\n"))
K(fma_vvv, X8("v_fma_f64 v[12:13], v[4:5], v[6:7], v[8:9]\n v_fma_f64 v[14:15], v[4:5], v[6:7], v[10:11]\n"))
K(fma_vvs, X8("v_fma_f64 v[12:13], v[4:5], v[6:7], s[20:21]\n v_fma_f64 v[14:15], v[4:5], v[6:7], s[20:21]\n"))   // = mul (addend -0.0 in an SGPR pair)
K(fma_vv0, X8("v_fma_f64 v[12:13], v[4:5], v[6:7], 0\n v_fma_f64 v[14:15], v[4:5], v[6:7], 0\n"))
K(fma_v1v, X8("v_fma_f64 v[12:13], v[4:5], 1.0, v[6:7]\n v_fma_f64 v[14:15], v[4:5], 1.0, v[6:7]\n"))             // = add
K(fma_v1v_b00, X8("v_fma_f64 v[12:13], v[4:5], 1.0, v[8:9]\n v_fma_f64 v[14:15], v[4:5], 1.0, v[8:9]\n"))
K(fma_vvs_b00, X8("v_fma_f64 v[12:13], v[4:5], v[8:9], s[20:21]\n v_fma_f64 v[14:15], v[4:5], v[8:9], s[20:21]\n"))
K(fmac_b02, X8("v_fmac_f64 v[12:13], v[4:5], v[6:7]\n v_fmac_f64 v[14:15], v[4:5], v[6:7]\n"))       // dst = src2: banks 0,2 + dst bank 0 / 2
K(fma_vvv_same, X8("v_fma_f64 v[12:13], v[4:5], v[6:7], v[4:5]\n v_fma_f64 v[14:15], v[4:5], v[6:7], v[6:7]\n"))  // a repeated operand
K(mov_dpp, X8("v_mov_b32_dpp v12, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v14, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
K(mul_f32, X8("v_mul_f32 v12, v4, v6\n v_mul_f32 v14, v4, v6\n"))
// dependent chains
K(dep_mul, X8("v_mul_f64 v[12:13], v[12:13], v[6:7]\n v_mul_f64 v[12:13], v[12:13], v[6:7]\n"))
K(dep_fma_vvs, X8("v_fma_f64 v[12:13], v[12:13], v[6:7], s[20:21]\n v_fma_f64 v[12:13], v[12:13], v[6:7], s[20:21]\n"))
K(dep_fma_vvv, X8("v_fma_f64 v[12:13], v[4:5], v[6:7], v[12:13]\n v_fma_f64 v[12:13], v[4:5], v[6:7], v[12:13]\n"))

template <class F> void run(const char *name, F kern)
{
    unsigned long long *d, h = 0;
    double *o;
    (void)hipMalloc(&d, 8);
    (void)hipMalloc(&o, 64 * 8);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, o);
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%-14s %8.2f ticks per instruction (%llu ticks for %d)\n", name, (double)h / (REP * 16), h, REP * 16);
    (void)hipFree(d); (void)hipFree(o);
}
int main()
{
#define R(n) run(#n, n)
    R(mul_b02); R(mul_b00); R(add_b02); R(add_b00); R(fma_vvv); R(fma_vvs); R(fma_vv0); R(fma_v1v); R(fma_v1v_b00);
    R(fma_vvs_b00); R(fmac_b02); R(fma_vvv_same); R(mov_dpp); R(mul_f32); R(dep_mul); R(dep_fma_vvs); R(dep_fma_vvv);
    return 0;
}

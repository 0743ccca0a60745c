// Scalar types for the fp64 / complex-fp64 instantiations of every kernel.
//
// The reference runs its kernels either on complex128 (frequency domain) or on
// float64 (Laplace domain) arrays (reference emg3d/fields.py:93-98). `cplx` is a
// plain {re, im} pair laid out exactly like numpy complex128 / C99 double complex,
// so device buffers can be filled from either without conversion.
//
// EMG_HD expands to __host__ __device__ under hipcc and to nothing under a plain
// host compiler; the latter is used ONLY by the CPU emulation harness in
// tests/emu/ (unit tests of the kernel bodies without a GPU), never by the product.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EMG_HD __host__ __device__ __forceinline__
#else
#define EMG_HD inline
#endif

namespace emg {

struct cplx {
    double re, im;
    EMG_HD cplx() {}
    EMG_HD cplx(double r) : re(r), im(0.0) {}
    EMG_HD cplx(double r, double i) : re(r), im(i) {}
};

EMG_HD cplx operator+(cplx a, cplx b) { return cplx(a.re + b.re, a.im + b.im); }
EMG_HD cplx operator-(cplx a, cplx b) { return cplx(a.re - b.re, a.im - b.im); }
EMG_HD cplx operator-(cplx a) { return cplx(-a.re, -a.im); }
EMG_HD cplx operator*(cplx a, cplx b)
{
    return cplx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
EMG_HD cplx operator*(double a, cplx b) { return cplx(a * b.re, a * b.im); }
EMG_HD cplx operator*(cplx a, double b) { return cplx(a.re * b, a.im * b); }
EMG_HD cplx operator+(cplx a, double b) { return cplx(a.re + b, a.im); }
EMG_HD cplx operator+(double a, cplx b) { return cplx(a + b.re, b.im); }
EMG_HD cplx operator-(cplx a, double b) { return cplx(a.re - b, a.im); }
EMG_HD cplx &operator+=(cplx &a, cplx b) { a.re += b.re; a.im += b.im; return a; }
EMG_HD cplx &operator-=(cplx &a, cplx b) { a.re -= b.re; a.im -= b.im; return a; }
EMG_HD cplx &operator+=(cplx &a, double b) { a.re += b; return a; }
EMG_HD cplx &operator*=(cplx &a, cplx b) { a = a * b; return a; }
EMG_HD cplx &operator*=(cplx &a, double b) { a.re *= b; a.im *= b; return a; }

// Compact (single-precision) STORAGE types of the line factors and w records ("compact line
// factors"): values are rounded once when they are stored and widened when they are loaded; every arithmetic
// operation stays fp64. compact_of<T>::type: cplx -> cplxf, double -> float.
struct cplxf {
    float re, im;
    EMG_HD cplxf() {}
    EMG_HD cplxf(float r, float i) : re(r), im(i) {}
};
template <class T> struct compact_of;
template <> struct compact_of<double> { using type = float; };
template <> struct compact_of<cplx> { using type = cplxf; };
EMG_HD double widen(double a) { return a; }
EMG_HD double widen(float a) { return (double)a; }
EMG_HD cplx widen(cplx a) { return a; }
EMG_HD cplx widen(cplxf a) { return cplx((double)a.re, (double)a.im); }
// value of type T as the storage type S (S = T: unchanged; S = compact_of<T>: rounded to nearest)
template <class S> struct narrow_to;
template <> struct narrow_to<double> { static EMG_HD double of(double a) { return a; } };
template <> struct narrow_to<float> { static EMG_HD float of(double a) { return (float)a; } };
template <> struct narrow_to<cplx> { static EMG_HD cplx of(cplx a) { return a; } };
template <> struct narrow_to<cplxf> { static EMG_HD cplxf of(cplx a) { return cplxf((float)a.re, (float)a.im); } };
template <class S, class T> EMG_HD S narrow(T a) { return narrow_to<S>::of(a); }

// 1/z with one real division: conj(z) / |z|^2.
EMG_HD cplx recip(cplx a)
{
    double d = 1.0 / (a.re * a.re + a.im * a.im);
    return cplx(a.re * d, -a.im * d);
}
EMG_HD double recip(double a) { return 1.0 / a; }

// Reciprocal for the pivots of the point smoother's 6 x 6 systems: hardware estimate
// (v_rcp_f64) + two Newton steps, 5 instructions and within an ulp or two of 1/a, instead of
// the ~12-instruction correctly rounded IEEE division sequence -- six pivots per node.
EMG_HD double recip_fast(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return r;
#else
    return 1.0 / a;
#endif
}
EMG_HD cplx recip_fast(cplx a)
{
    const double d = recip_fast(a.re * a.re + a.im * a.im);
    return cplx(a.re * d, -a.im * d);
}

EMG_HD double abs2(cplx a) { return a.re * a.re + a.im * a.im; }
EMG_HD double abs2(double a) { return a * a; }

EMG_HD double imag_of(cplx a) { return a.im; }
EMG_HD double imag_of(double a) { return a; }
EMG_HD double real_of(cplx a) { return a.re; }
EMG_HD double real_of(double) { return 0.0; }      // a real field has no separate "real part of eta"
// value whose stored half is v: a purely imaginary complex number, or the real number itself
template <class T> EMG_HD T from_stored(double v);
template <> EMG_HD double from_stored<double>(double v) { return v; }
template <> EMG_HD cplx from_stored<cplx>(double v) { return cplx(0.0, v); }

template <class T> EMG_HD T zero();
template <> EMG_HD double zero<double>() { return 0.0; }
template <> EMG_HD cplx zero<cplx>() { return cplx(0.0, 0.0); }

// c + a*b and c - a*b with explicit fused multiply-adds: four per complex product. Written
// with operators, `c - a * b` costs six instructions per complex product (2 mul + 2 fma for
// the product, 2 add for the difference: contraction may not re-associate the sum), and the
// sequential recurrences of the smoothers are bound by the number of fp64 instructions.
EMG_HD double mad(double a, double b, double c) { return __builtin_fma(a, b, c); }
EMG_HD double nmad(double a, double b, double c) { return __builtin_fma(-a, b, c); }
EMG_HD cplx mad(double a, cplx b, cplx c) { return cplx(__builtin_fma(a, b.re, c.re), __builtin_fma(a, b.im, c.im)); }
EMG_HD cplx nmad(double a, cplx b, cplx c)
{
    return cplx(__builtin_fma(-a, b.re, c.re), __builtin_fma(-a, b.im, c.im));
}
EMG_HD cplx mad(cplx a, cplx b, cplx c)
{
    return cplx(__builtin_fma(-a.im, b.im, __builtin_fma(a.re, b.re, c.re)),
                __builtin_fma(a.im, b.re, __builtin_fma(a.re, b.im, c.im)));
}
EMG_HD cplx nmad(cplx a, cplx b, cplx c)
{
    return cplx(__builtin_fma(a.im, b.im, __builtin_fma(-a.re, b.re, c.re)),
                __builtin_fma(-a.im, b.re, __builtin_fma(-a.re, b.im, c.im)));
}

// a*b+c helpers (the compiler contracts these into v_fma_f64)
EMG_HD double fmadd(double a, double b, double c) { return a * b + c; }
EMG_HD cplx fmadd(double a, cplx b, cplx c) { return cplx(a * b.re + c.re, a * b.im + c.im); }
EMG_HD cplx fmadd(cplx a, cplx b, cplx c)
{
    return cplx(a.re * b.re - a.im * b.im + c.re, a.re * b.im + a.im * b.re + c.im);
}

}  // namespace emg

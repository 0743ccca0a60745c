// What follows a solve: the magnetic field of an electric
// field and the responses at receiver positions, on the device -- the reference does both
// with NumPy/SciPy on the host (emg3d/fields.py:522-659, 941-1009; emg3d/maps.py:232-368,
// 500-552), where the cubic-spline prefilter of one 128^3 component alone costs more than
// a whole solve here. Included at the end of kernels.hip (one translation unit).
//
//   k_edge_curl      fields._edge_curl_factor: H = curl E / (s mu), face values
//   k_spline_filter  scipy.ndimage.spline_filter(order=3, mode='mirror'), one axis, in place:
//                    one thread per line (recursive filter, pole sqrt(3)-2, exact mirror
//                    initialisation over the whole line: ndimage/src/ni_splines.c)
//   k_spline_eval    scipy.ndimage.map_coordinates(order=3, mode='constant', cval=nan),
//                    prefiltered coefficients in, one thread per point
//   k_linear_eval    RegularGridInterpolator(method='linear', fill_value=nan), cell index and
//                    weights from the host (np.searchsorted), one thread per point
// and what precedes it when the computational grid differs from the model grid (rank 3):
//   k_volume_average maps.interp_volume_average (emg3d/maps.py:555-664), one thread per output cell
#pragma once

namespace {

template <class T>
__global__ __launch_bounds__(256) void k_edge_curl(int nx, int ny, int nz, const T *ex, const T *ey, const T *ez,
                                                   const double *zeta, const double *hx, const double *hy,
                                                   const double *hz, T inv_smu0, T *mx, T *my, T *mz)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    const int iz = blockIdx.z;
    if (ix >= nx || iy >= ny) return;
    const int ixm = max(0, ix - 1), iym = max(0, iy - 1), izm = max(0, iz - 1);
    const int ixp = ix + 1, iyp = iy + 1, izp = iz + 1;
#define EX(i, j, k) ex[(size_t)(i) + (size_t)nx * ((j) + (size_t)(ny + 1) * (k))]
#define EY(i, j, k) ey[(size_t)(i) + (size_t)(nx + 1) * ((j) + (size_t)ny * (k))]
#define EZ(i, j, k) ez[(size_t)(i) + (size_t)(nx + 1) * ((j) + (size_t)(ny + 1) * (k))]
#define ZT(i, j, k) zeta[(size_t)(i) + (size_t)nx * ((j) + (size_t)ny * (k))]
    // nabla x E (fields.py:980-986)
    const T fx = (EZ(ix, iyp, iz) - EZ(ix, iy, iz)) * (1.0 / hy[iy]) - (EY(ix, iy, izp) - EY(ix, iy, iz)) * (1.0 / hz[iz]);
    const T fy = (EX(ix, iy, izp) - EX(ix, iy, iz)) * (1.0 / hz[iz]) - (EZ(ixp, iy, iz) - EZ(ix, iy, iz)) * (1.0 / hx[ix]);
    const T fz = (EY(ixp, iy, iz) - EY(ix, iy, iz)) * (1.0 / hx[ix]) - (EX(ix, iyp, iz) - EX(ix, iy, iz)) * (1.0 / hy[iy]);
    // zeta / (s mu0) averaged over the two cells of the face (fields.py:988-996)
    const double dx = hx[ixm] + hx[ix], dy = hy[iym] + hy[iy], dz = hz[izm] + hz[iz];
    const double z0 = ZT(ix, iy, iz);
    const T zx = (ZT(ixm, iy, iz) + z0) * inv_smu0, zy = (ZT(ix, iym, iz) + z0) * inv_smu0;
    const T zz = (ZT(ix, iy, izm) + z0) * inv_smu0;
    if (ix != 0) mx[(size_t)ix + (size_t)(nx + 1) * (iy + (size_t)ny * iz)] = fx * zx * (1.0 / (dx * hy[iy] * hz[iz]));
    if (iy != 0) my[(size_t)ix + (size_t)nx * (iy + (size_t)(ny + 1) * iz)] = fy * zy * (1.0 / (hx[ix] * dy * hz[iz]));
    if (iz != 0) mz[(size_t)ix + (size_t)nx * (iy + (size_t)ny * iz)] = fz * zz * (1.0 / (hx[ix] * hy[iy] * dz));
#undef EX
#undef EY
#undef EZ
#undef ZT
}

// One thread per line; the host describes the lines of a pass: line l = u + nu * v starts at
// element u * su + v * sv, has n elements, `s` apart (n >= 2; shorter axes are skipped by the
// host).
template <class T>
__global__ __launch_bounds__(64) void k_spline_filter(T *c, long nlines, long nu, long su, long sv, int n, long s_)
{
    const long l = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    const size_t first = (size_t)((l % nu) * su + (l / nu) * sv), s = (size_t)s_;
    T *p = c + first;
    const double z = -0.26794919243112270647;     // sqrt(3) - 2
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    // gain, then the causal initial value of the mirrored, periodic extension (exact sum)
    double z_n_1 = 1.0;
    for (int i = 0; i < n - 1; ++i) z_n_1 *= z;
    T c0 = gain * p[0] + z_n_1 * (gain * p[(size_t)(n - 1) * s]);
    double z_i = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 = c0 + z_i * (gain * p[(size_t)i * s] + z_n_1 * (gain * p[(size_t)(n - 1 - i) * s]));
        z_i *= z;
    }
    T prev = c0 * (1.0 / (1.0 - z_n_1 * z_n_1)), prev2 = prev;
    p[0] = prev;
    for (int i = 1; i < n; ++i) {                    // causal
        prev2 = prev;
        prev = gain * p[(size_t)i * s] + z * prev;
        p[(size_t)i * s] = prev;
    }
    // the last two causal values from registers (not re-read right behind their stores)
    T nxt = (z * prev2 + prev) * (z / (z * z - 1.0));
    p[(size_t)(n - 1) * s] = nxt;
    for (int i = n - 2; i >= 0; --i) {               // anticausal
        nxt = z * (nxt - p[(size_t)i * s]);
        p[(size_t)i * s] = nxt;
    }
}

__device__ __forceinline__ int spline_mirror(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i >= n ? p - i : i;
}
__device__ __forceinline__ void spline_weights(double x, int &start, double (&w)[4])
{
    const double fl = floor(x), t = x - fl;
    start = (int)fl - 1;
    w[0] = (1.0 - t) * (1.0 - t) * (1.0 - t) / 6.0;
    w[1] = (3.0 * t * t * t - 6.0 * t * t + 4.0) / 6.0;
    w[2] = (-3.0 * t * t * t + 3.0 * t * t + 3.0 * t + 1.0) / 6.0;
    w[3] = t * t * t / 6.0;
}
template <class T> __device__ __forceinline__ T nan_value();
template <> __device__ __forceinline__ double nan_value<double>() { return __builtin_nan(""); }
template <> __device__ __forceinline__ cplx nan_value<cplx>() { return cplx(__builtin_nan(""), 0.0); }

template <class T>
__global__ __launch_bounds__(64) void k_spline_eval(const T *c, int n0, int n1, int n2, const double *coords, int npts,
                                                    T *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    const double x = coords[p], y = coords[npts + p], zc = coords[2 * (size_t)npts + p];
    // !(a >= 0 && a <= n - 1) also catches NaN coordinates
    if (!(x >= 0.0 && x <= n0 - 1.0 && y >= 0.0 && y <= n1 - 1.0 && zc >= 0.0 && zc <= n2 - 1.0)) {
        out[p] = nan_value<T>();
        return;
    }
    int sx, sy, sz;
    double wx[4], wy[4], wz[4];
    spline_weights(x, sx, wx);
    spline_weights(y, sy, wy);
    spline_weights(zc, sz, wz);
    T acc = emg::zero<T>();
    for (int a = 0; a < 4; ++a) {
        const int ia = spline_mirror(sx + a, n0);
        for (int b = 0; b < 4; ++b) {
            const int ib = spline_mirror(sy + b, n1);
            for (int d = 0; d < 4; ++d) {
                const int id = spline_mirror(sz + d, n2);
                acc = acc + (wx[a] * wy[b] * wz[d]) * c[(size_t)ia + (size_t)n0 * (ib + (size_t)n1 * id)];
            }
        }
    }
    out[p] = acc;
}

template <class T>
__global__ __launch_bounds__(64) void k_linear_eval(const T *v, int n0, int n1, int n2, const int32_t *idx,
                                                    const double *w, int npts, T *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    const int i = idx[p], j = idx[npts + p], k = idx[2 * (size_t)npts + p];
    if (i < 0 || j < 0 || k < 0) {
        out[p] = nan_value<T>();
        return;
    }
    const double wx = w[p], wy = w[npts + p], wz = w[2 * (size_t)npts + p];
    T acc = emg::zero<T>();
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int d = 0; d < 2; ++d) {
                const double wt = (a ? wx : 1.0 - wx) * (b ? wy : 1.0 - wy) * (d ? wz : 1.0 - wz);
                acc = acc + wt * v[(size_t)(i + a) + (size_t)n0 * ((j + b) + (size_t)n1 * (k + d))];
            }
    out[p] = acc;
}

// Model re-gridding by volume averaging (maps.interp_volume_average, emg3d/maps.py:555-616):
// every axis is cut into segments by the union of input and output nodes (host, tiny:
// maps._volume_average_weights); the segments of output cell o along an axis are
// [seg[o], seg[o+1]) with length w and input cell in. One thread per OUTPUT cell gathers
// sum (wz wy) wx v in the reference's z, y, x order -- the same additions in the same order
// as its scatter loop -- and divides by the output cell volume.
__global__ __launch_bounds__(256) void k_volume_average(const double *v, int nx, int ny, const int32_t *sx,
                                                        const int32_t *sy, const int32_t *sz, const double *wx,
                                                        const double *wy, const double *wz, const int32_t *inx,
                                                        const int32_t *iny, const int32_t *inz, const double *vol,
                                                        int mx, int my, int mz, double *out, int logscale)
{
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y * blockDim.y + threadIdx.y;
    const int oz = blockIdx.z;
    if (ox >= mx || oy >= my) return;
    // logscale == 2: the ADJOINT of the (linear) averaging, for the gradient's way back from a
    // computational grid (reference maps._interp_volume_average_adj, emg3d/maps.py:722-750): the
    // tables are the transposed ones (segments grouped by the cell of the ORIGINAL grid, which is
    // the output here), `v` lives on the averaged grid together with ITS cell volumes `vol`, and
    // the result is ADDED to `out`:  out_i += sum_o (overlap_io / vol_o) v_o
    const bool adjoint = logscale == 2;
    double acc = 0.0;
    for (int a = sz[oz]; a < sz[oz + 1]; ++a)
        for (int b = sy[oy]; b < sy[oy + 1]; ++b) {
            const double w_zy = wz[a] * wy[b];
            const size_t roff = (size_t)nx * (iny[b] + (size_t)ny * inz[a]);
            const double *row = v + roff;
            // logscale == 1: the average of log10(value), returned as 10 ** average (maps.py:346-358)
            if (adjoint) {
                for (int c = sx[ox]; c < sx[ox + 1]; ++c) acc += w_zy * wx[c] * (row[inx[c]] / vol[roff + inx[c]]);
            } else {
                for (int c = sx[ox]; c < sx[ox + 1]; ++c) acc += w_zy * wx[c] * (logscale ? log10(row[inx[c]]) : row[inx[c]]);
            }
        }
    const size_t o = (size_t)ox + (size_t)mx * (oy + (size_t)my * oz);
    if (adjoint) out[o] += acc;
    else out[o] = logscale ? pow(10.0, acc / vol[o]) : acc / vol[o];
}

}  // namespace

extern "C" {

int emg3d_dev_volume_average(const double *values, int nx, int ny, int nz, const int32_t *segx, const int32_t *segy,
                             const int32_t *segz, const double *wx, const double *wy, const double *wz,
                             const int32_t *inx, const int32_t *iny, const int32_t *inz, const double *new_vol,
                             int mx, int my, int mz, double *out, int log10_scale, void *stream)
{
    if (!values || !segx || !segy || !segz || !wx || !wy || !wz || !inx || !iny || !inz || !new_vol || !out ||
        nx < 1 || ny < 1 || nz < 1 || mx < 1 || my < 1 || mz < 1)
        return fail(EMG3D_ERR_BADARG, "volume_average: bad argument");
    const dim3 block(64, 4, 1), grid((mx + 63) / 64, (my + 3) / 4, mz);
    hipLaunchKernelGGL(k_volume_average, grid, block, 0, (hipStream_t)stream, values, nx, ny, segx, segy, segz, wx, wy,
                       wz, inx, iny, inz, new_vol, mx, my, mz, out, log10_scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_magnetic_field(int nx, int ny, int nz, int is_complex, const void *ex, const void *ey, const void *ez,
                             const double *zeta, const double *hx, const double *hy, const double *hz, double smu0_re,
                             double smu0_im, void *mx, void *my, void *mz, void *stream)
{
    if (nx < 1 || ny < 1 || nz < 1 || !ex || !ey || !ez || !zeta || !hx || !hy || !hz || !mx || !my || !mz)
        return fail(EMG3D_ERR_BADARG, "magnetic_field: bad argument");
    const hipStream_t st = (hipStream_t)stream;
    const size_t esz = is_complex ? 16 : 8;
    HIP_TRY(hipMemsetAsync(mx, 0, (size_t)(nx + 1) * ny * nz * esz, st));
    HIP_TRY(hipMemsetAsync(my, 0, (size_t)nx * (ny + 1) * nz * esz, st));
    HIP_TRY(hipMemsetAsync(mz, 0, (size_t)nx * ny * (nz + 1) * esz, st));
    const dim3 block(64, 4, 1), grid((nx + 63) / 64, (ny + 3) / 4, nz);
    if (is_complex)
        hipLaunchKernelGGL(k_edge_curl<cplx>, grid, block, 0, st, nx, ny, nz, (const cplx *)ex, (const cplx *)ey,
                           (const cplx *)ez, zeta, hx, hy, hz, emg::recip(cplx(smu0_re, smu0_im)), (cplx *)mx, (cplx *)my,
                           (cplx *)mz);
    else
        hipLaunchKernelGGL(k_edge_curl<double>, grid, block, 0, st, nx, ny, nz, (const double *)ex, (const double *)ey,
                           (const double *)ez, zeta, hx, hy, hz, 1.0 / smu0_re, (double *)mx, (double *)my, (double *)mz);
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_spline_filter(void *data, int n0, int n1, int n2, int is_complex, void *stream)
{
    if (!data || n0 < 1 || n1 < 1 || n2 < 1) return fail(EMG3D_ERR_BADARG, "spline_filter: bad argument");
    const hipStream_t st = (hipStream_t)stream;
    const long plane = (long)n0 * n1;
    // lines of the three passes: (nlines, nu, su, sv, n, s)
    const long pass[3][6] = {{(long)n1 * n2, n1, n0, plane, n0, 1},
                             {(long)n0 * n2, n0, 1, plane, n1, n0},
                             {plane, plane, 1, 0, n2, plane}};
    for (int axis = 0; axis < 3; ++axis) {
        const long *q = pass[axis];
        if (q[4] < 2) continue;                      // a single sample is its own coefficient
        const dim3 grid((unsigned)((q[0] + 63) / 64));
        if (is_complex)
            hipLaunchKernelGGL(k_spline_filter<cplx>, grid, dim3(64), 0, st, (cplx *)data, q[0], q[1], q[2], q[3], (int)q[4], q[5]);
        else
            hipLaunchKernelGGL(k_spline_filter<double>, grid, dim3(64), 0, st, (double *)data, q[0], q[1], q[2], q[3], (int)q[4], q[5]);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_spline_eval(const void *coef, int n0, int n1, int n2, int is_complex, const double *coords, int npts,
                          void *out, void *stream)
{
    if (!coef || !coords || !out || npts < 0) return fail(EMG3D_ERR_BADARG, "spline_eval: bad argument");
    if (npts == 0) return 0;
    const dim3 grid((npts + 63) / 64);
    if (is_complex)
        hipLaunchKernelGGL(k_spline_eval<cplx>, grid, dim3(64), 0, (hipStream_t)stream, (const cplx *)coef, n0, n1, n2,
                           coords, npts, (cplx *)out);
    else
        hipLaunchKernelGGL(k_spline_eval<double>, grid, dim3(64), 0, (hipStream_t)stream, (const double *)coef, n0, n1,
                           n2, coords, npts, (double *)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int emg3d_dev_linear_eval(const void *values, int n0, int n1, int n2, int is_complex, const int32_t *idx,
                          const double *w, int npts, void *out, void *stream)
{
    if (!values || !idx || !w || !out || npts < 0) return fail(EMG3D_ERR_BADARG, "linear_eval: bad argument");
    if (npts == 0) return 0;
    const dim3 grid((npts + 63) / 64);
    if (is_complex)
        hipLaunchKernelGGL(k_linear_eval<cplx>, grid, dim3(64), 0, (hipStream_t)stream, (const cplx *)values, n0, n1, n2,
                           idx, w, npts, (cplx *)out);
    else
        hipLaunchKernelGGL(k_linear_eval<double>, grid, dim3(64), 0, (hipStream_t)stream, (const double *)values, n0, n1,
                           n2, idx, w, npts, (double *)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

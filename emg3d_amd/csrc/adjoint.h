// Adjoint-state gradient: what the reference does on the host
// after the forward and the back-propagated solve of a source-frequency pair
// (emg3d/simulations.py:1041-1063):
//     gfield = real(bfield * s mu0 * efield)                         on the edges
//     maps.interp_edges_to_vol_averages(gfield, volumes, grad)       (emg3d/maps.py:667-719)
//     gradient += grad
// as ONE gather kernel over the cells while both fields are still in HBM: a cell adds up its
// four x-, y- and z-edges (each edge value times volume / 4) in the order the reference's
// scatter loop reaches them (iz outer, iy, ix inner), so the sums carry the same bits. The
// reference's loop adds a boundary edge twice to its boundary cell; those edges are tangential
// to the PEC boundary and exactly zero in both fields, so nothing is lost by adding them once.
// Included at the end of kernels.hip (one translation unit).
#pragma once

namespace {

__device__ __forceinline__ double grad_edge(cplx b, cplx e, cplx smu0)
{
    const cplx t = b * smu0;                       // numpy evaluates (bfield * smu0) * efield
    return t.re * e.re - t.im * e.im;
}
__device__ __forceinline__ double grad_edge(double b, double e, double smu0) { return b * smu0 * e; }

template <class T>
__global__ __launch_bounds__(256) void k_gradient_accumulate(int nx, int ny, int nz, const T *ex, const T *ey, const T *ez,
                                                             const T *bx, const T *by, const T *bz, T smu0,
                                                             const double *vol, double *gx, double *gy, double *gz)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y, iz = blockIdx.z;
    if (ix >= nx || iy >= ny) return;
    const size_t c = (size_t)ix + (size_t)nx * (iy + (size_t)ny * iz);
    const double v = vol[c];
#define IX(i, j, k) ((size_t)(i) + (size_t)nx * ((j) + (size_t)(ny + 1) * (k)))
#define IY(i, j, k) ((size_t)(i) + (size_t)(nx + 1) * ((j) + (size_t)ny * (k)))
#define IZ(i, j, k) ((size_t)(i) + (size_t)(nx + 1) * ((j) + (size_t)(ny + 1) * (k)))
#define TERM(B, E, I) (v * grad_edge(B[I], E[I], smu0) / 4)
    {   // x-edges (ix; iy | iy+1; iz | iz+1): order of the reference's loop = z outer, y inner
        double a = gx[c];
        a += TERM(bx, ex, IX(ix, iy, iz));
        a += TERM(bx, ex, IX(ix, iy + 1, iz));
        a += TERM(bx, ex, IX(ix, iy, iz + 1));
        a += TERM(bx, ex, IX(ix, iy + 1, iz + 1));
        gx[c] = a;
    }
    {   // y-edges (ix | ix+1; iy; iz | iz+1): x inner
        double a = gy[c];
        a += TERM(by, ey, IY(ix, iy, iz));
        a += TERM(by, ey, IY(ix + 1, iy, iz));
        a += TERM(by, ey, IY(ix, iy, iz + 1));
        a += TERM(by, ey, IY(ix + 1, iy, iz + 1));
        gy[c] = a;
    }
    {   // z-edges (ix | ix+1; iy | iy+1; iz)
        double a = gz[c];
        a += TERM(bz, ez, IZ(ix, iy, iz));
        a += TERM(bz, ez, IZ(ix + 1, iy, iz));
        a += TERM(bz, ez, IZ(ix, iy + 1, iz));
        a += TERM(bz, ez, IZ(ix + 1, iy + 1, iz));
        gz[c] = a;
    }
#undef IX
#undef IY
#undef IZ
#undef TERM
}

}  // namespace

extern "C" {

int emg3d_dev_gradient_accumulate(int nx, int ny, int nz, int is_complex, const void *ex, const void *ey, const void *ez,
                                  const void *bx, const void *by, const void *bz, double smu0_re, double smu0_im,
                                  const double *volumes, double *gx, double *gy, double *gz, void *stream)
{
    if (!ex || !bx || !volumes || !gx || !gy || !gz) return fail(EMG3D_ERR_BADARG, "gradient_accumulate: bad argument");
    const dim3 block(64, 4, 1), grid(cdiv(nx, 64), cdiv(ny, 4), nz);
    if (is_complex)
        hipLaunchKernelGGL(k_gradient_accumulate<cplx>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, (const cplx *)ex,
                           (const cplx *)ey, (const cplx *)ez, (const cplx *)bx, (const cplx *)by, (const cplx *)bz,
                           cplx(smu0_re, smu0_im), volumes, gx, gy, gz);
    else
        hipLaunchKernelGGL(k_gradient_accumulate<double>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, (const double *)ex,
                           (const double *)ey, (const double *)ez, (const double *)bx, (const double *)by,
                           (const double *)bz, smu0_re, volumes, gx, gy, gz);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

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

// ---- before a solve: the source vector of a dipole or wire on the
// device. The reference (fields._dipole_vector, emg3d/fields.py:792-938) walks, per straight
// segment, over the cells of its bounding box and clips the segment against each cell; here one
// thread walks along its segment from grid plane to grid plane (the crossing parameters of the
// three axes are three increasing sequences, merged on the fly) and deposits every piece -- its
// x / y / z extent on the four edges of that direction of the cell holding the piece's midpoint,
// weighted bilinearly in the two transverse coordinates of the midpoint -- with atomic adds.
namespace {

__device__ __forceinline__ int src_cell(const double *nodes, int n, double v)
{   // index of the cell [nodes[i], nodes[i+1]) that holds v, clamped to 0 .. n-1 (n cells)
    int lo = 0, hi = n + 1;                      // upper_bound over the n + 1 nodes
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nodes[mid] <= v) lo = mid + 1; else hi = mid;
    }
    const int i = lo - 1;
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
template <class T> __device__ __forceinline__ void src_add(T *p, double v, cplx scale);
template <> __device__ __forceinline__ void src_add<cplx>(cplx *p, double v, cplx scale)
{
    atomicAdd(&p->re, v * scale.re);
    atomicAdd(&p->im, v * scale.im);
}
template <> __device__ __forceinline__ void src_add<double>(double *p, double v, cplx scale) { atomicAdd(p, v * scale.re); }

template <class T>
__global__ __launch_bounds__(64) void k_source_segments(int nx, int ny, int nz, const double *nodes_x, const double *nodes_y,
                                                        const double *nodes_z, const double *hx, const double *hy,
                                                        const double *hz, const double *points, int nseg, cplx scale,
                                                        T *sx, T *sy, T *sz)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const double *nodes[3] = {nodes_x, nodes_y, nodes_z};
    const double *h[3] = {hx, hy, hz};
    const int n[3] = {nx, ny, nz};
    double p0[3], d[3];
    int k[3], kend[3], step[3];
    for (int a = 0; a < 3; ++a) {
        p0[a] = points[3 * s + a];
        d[a] = points[3 * (s + 1) + a] - p0[a];
        // grid planes crossed strictly inside the segment, in the order of the walk
        const double lo = d[a] > 0 ? p0[a] : p0[a] + d[a], hi = d[a] > 0 ? p0[a] + d[a] : p0[a];
        int first = 0, last = n[a];
        while (first <= n[a] && !(nodes[a][first] > lo)) ++first;       // first node > lo
        while (last >= 0 && !(nodes[a][last] < hi)) --last;              // last node < hi
        if (d[a] == 0.0 || first > last) { k[a] = 0; kend[a] = -1; step[a] = 1; }
        else if (d[a] > 0) { k[a] = first; kend[a] = last; step[a] = 1; }
        else { k[a] = last; kend[a] = first; step[a] = -1; }
    }
    auto remaining = [&](int a) { return kend[a] >= 0 && (step[a] > 0 ? k[a] <= kend[a] : k[a] >= kend[a]); };
    auto tcross = [&](int a) { return (nodes[a][k[a]] - p0[a]) / d[a]; };
    double t0 = 0.0;
    for (;;) {
        double t1 = 1.0;
        for (int a = 0; a < 3; ++a)
            if (remaining(a)) { const double t = tcross(a); if (t < t1) t1 = t; }
        for (int a = 0; a < 3; ++a)             // crossings that coincide (a corner, an edge) count once
            while (remaining(a) && tcross(a) <= t1 + 1e-14) k[a] += step[a];
        if (t1 - t0 > 1e-14) {
            int c[3];
            double r[3];
            for (int a = 0; a < 3; ++a) {
                const double mid = p0[a] + 0.5 * (t0 + t1) * d[a];
                c[a] = src_cell(nodes[a], n[a], mid);
                r[a] = (mid - nodes[a][c[a]]) / h[a][c[a]];
            }
            const double w = t1 - t0;
            const int ix = c[0], iy = c[1], iz = c[2];
            const double rx = r[0], ry = r[1], rz = r[2];
            if (d[0] != 0.0) {
                const double m = w * d[0];
                src_add<T>(sx + ((size_t)ix + (size_t)nx * (iy + (size_t)(ny + 1) * iz)), m * (1 - ry) * (1 - rz), scale);
                src_add<T>(sx + ((size_t)ix + (size_t)nx * (iy + 1 + (size_t)(ny + 1) * iz)), m * ry * (1 - rz), scale);
                src_add<T>(sx + ((size_t)ix + (size_t)nx * (iy + (size_t)(ny + 1) * (iz + 1))), m * (1 - ry) * rz, scale);
                src_add<T>(sx + ((size_t)ix + (size_t)nx * (iy + 1 + (size_t)(ny + 1) * (iz + 1))), m * ry * rz, scale);
            }
            if (d[1] != 0.0) {
                const double m = w * d[1];
                src_add<T>(sy + ((size_t)ix + (size_t)(nx + 1) * (iy + (size_t)ny * iz)), m * (1 - rx) * (1 - rz), scale);
                src_add<T>(sy + ((size_t)ix + 1 + (size_t)(nx + 1) * (iy + (size_t)ny * iz)), m * rx * (1 - rz), scale);
                src_add<T>(sy + ((size_t)ix + (size_t)(nx + 1) * (iy + (size_t)ny * (iz + 1))), m * (1 - rx) * rz, scale);
                src_add<T>(sy + ((size_t)ix + 1 + (size_t)(nx + 1) * (iy + (size_t)ny * (iz + 1))), m * rx * rz, scale);
            }
            if (d[2] != 0.0) {
                const double m = w * d[2];
                src_add<T>(sz + ((size_t)ix + (size_t)(nx + 1) * (iy + (size_t)(ny + 1) * iz)), m * (1 - rx) * (1 - ry), scale);
                src_add<T>(sz + ((size_t)ix + 1 + (size_t)(nx + 1) * (iy + (size_t)(ny + 1) * iz)), m * rx * (1 - ry), scale);
                src_add<T>(sz + ((size_t)ix + (size_t)(nx + 1) * (iy + 1 + (size_t)(ny + 1) * iz)), m * (1 - rx) * ry, scale);
                src_add<T>(sz + ((size_t)ix + 1 + (size_t)(nx + 1) * (iy + 1 + (size_t)(ny + 1) * iz)), m * rx * ry, scale);
            }
        }
        if (t1 >= 1.0) break;
        t0 = t1;
    }
}

}  // namespace

extern "C" {

int emg3d_dev_source_field(int nx, int ny, int nz, int is_complex, const double *nodes_x, const double *nodes_y,
                           const double *nodes_z, const double *hx, const double *hy, const double *hz,
                           const double *points, int npoints, double scale_re, double scale_im, void *sx, void *sy,
                           void *sz, void *stream)
{
    if (!nodes_x || !points || !sx || !sy || !sz || npoints < 2) return fail(EMG3D_ERR_BADARG, "source_field: bad argument");
    const int nseg = npoints - 1;
    const dim3 grid(cdiv(nseg, 64)), block(64);
    if (is_complex)
        hipLaunchKernelGGL(k_source_segments<cplx>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, nodes_x, nodes_y, nodes_z,
                           hx, hy, hz, points, nseg, cplx(scale_re, scale_im), (cplx *)sx, (cplx *)sy, (cplx *)sz);
    else
        hipLaunchKernelGGL(k_source_segments<double>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, nodes_x, nodes_y, nodes_z,
                           hx, hy, hz, points, nseg, cplx(scale_re, 0.0), (double *)sx, (double *)sy, (double *)sz);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- before a solve: the volume-integrated model on the device. models.VolumeModel
// (emg3d/models.py:654-691): eta = -s mu0 V (sigma [+ s eps0 eps_r]), zeta = V / mu_r with
// V = hx hy hz, evaluated in the reference's order of operations, from the PROPERTY arrays and the
// model's mapping (emg3d/maps.py:120-330: conductivity = backward(property)).
namespace {

__device__ __forceinline__ double vm_conductivity(double p, int mapping)
{
    switch (mapping) {
    case 1: return 1.0 / p;                 // Resistivity
    case 2: return pow(10.0, p);            // LgConductivity
    case 3: return pow(10.0, -p);           // LgResistivity
    case 4: return exp(p);                  // LnConductivity
    case 5: return exp(-p);                 // LnResistivity
    default: return p;                      // Conductivity
    }
}
__device__ __forceinline__ cplx vm_eta(cplx nsmu0, double vol, double sigma, bool has_eps, cplx seps)
{
    const cplx base = nsmu0 * vol;                               // (-s mu0) * V
    if (!has_eps) return base * sigma;
    return base * (cplx(sigma, 0.0) + seps);                     // (-s mu0 V) * (sigma + s eps0 eps_r)
}
__device__ __forceinline__ double vm_eta(double nsmu0, double vol, double sigma, bool has_eps, double seps)
{
    return has_eps ? (nsmu0 * vol) * (sigma + seps) : (nsmu0 * vol) * sigma;
}

template <class T>
__global__ __launch_bounds__(256) void k_volume_model(int nx, int ny, int nz, const double *px, const double *py,
                                                      const double *pz, const double *eps_r, const double *mu_r,
                                                      int mapping, const double *hx, const double *hy, const double *hz,
                                                      T nsmu0, T seps0, T *eta_x, T *eta_y, T *eta_z, double *zeta)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x, iy = blockIdx.y * blockDim.y + threadIdx.y, iz = blockIdx.z;
    if (ix >= nx || iy >= ny) return;
    const size_t c = (size_t)ix + (size_t)nx * (iy + (size_t)ny * iz);
    const double vol = (hx[ix] * hy[iy]) * hz[iz];
    const bool has_eps = eps_r != nullptr;
    const T seps = has_eps ? seps0 * eps_r[c] : emg::zero<T>();
    eta_x[c] = vm_eta(nsmu0, vol, vm_conductivity(px[c], mapping), has_eps, seps);
    if (py) eta_y[c] = vm_eta(nsmu0, vol, vm_conductivity(py[c], mapping), has_eps, seps);
    if (pz) eta_z[c] = vm_eta(nsmu0, vol, vm_conductivity(pz[c], mapping), has_eps, seps);
    zeta[c] = mu_r ? vol / mu_r[c] : vol;
}

}  // namespace

extern "C" {

int emg3d_dev_volume_model(int nx, int ny, int nz, int is_complex, const double *property_x, const double *property_y,
                           const double *property_z, const double *epsilon_r, const double *mu_r, int mapping,
                           const double *hx, const double *hy, const double *hz, double smu0_re, double smu0_im,
                           double seps0_re, double seps0_im, void *eta_x, void *eta_y, void *eta_z, double *zeta,
                           void *stream)
{
    if (!property_x || !hx || !hy || !hz || !eta_x || !zeta || (property_y && !eta_y) || (property_z && !eta_z) ||
        mapping < 0 || mapping > 5)
        return fail(EMG3D_ERR_BADARG, "volume_model: bad argument");
    const dim3 block(64, 4, 1), grid(cdiv(nx, 64), cdiv(ny, 4), nz);
    if (is_complex)
        hipLaunchKernelGGL(k_volume_model<cplx>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, property_x, property_y,
                           property_z, epsilon_r, mu_r, mapping, hx, hy, hz, cplx(-smu0_re, -smu0_im),
                           cplx(seps0_re, seps0_im), (cplx *)eta_x, (cplx *)eta_y, (cplx *)eta_z, zeta);
    else
        hipLaunchKernelGGL(k_volume_model<double>, grid, block, 0, (hipStream_t)stream, nx, ny, nz, property_x, property_y,
                           property_z, epsilon_r, mu_r, mapping, hx, hy, hz, -smu0_re, seps0_re, (double *)eta_x,
                           (double *)eta_y, (double *)eta_z, zeta);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

// Micro-benchmark: what this box's memory system streams -- the ceiling the smoothers' real
// traffic is measured against: a pure read (sum of double2), a copy and a
// fill over 2 GiB, 16-byte lanes, for several grid sizes.
//   hipcc --offload-arch=gfx950 -O3 hbm_rates.hip -o hbm_rates && ./hbm_rates
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const double2 *a, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;     // never true: keeps the loads
}
__global__ __launch_bounds__(256) void k_read4(const double2 *a, size_t n, double *out)
{
    double s = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += (v0.x + v0.y) + (v1.x + v1.y) + (v2.x + v2.y) + (v3.x + v3.y);
    }
    for (; i < n; i += stride) s += a[i].x;
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const double2 *a, double2 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_fill(double2 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = double2{1.0, 2.0};
}

// the same buffer read `passes` times in one launch: footprints that fit the 256 MiB Infinity Cache
// (but not the 32 MiB of L2) show what that cache level delivers to a streaming read
__global__ __launch_bounds__(256) void k_reread(const double2 *a, size_t n, int passes, double *out)
{
    double s = 0;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const double2 v = a[i];
            s += v.x + v.y;
        }
    if (s == 12345.678) out[0] = s;
}
// write a footprint, then read it back in reverse order (last written, first read: the forward /
// backward pattern of a line solve's records)
__global__ __launch_bounds__(256) void k_write_then_read(double2 *a, size_t n, double *out)
{
    double s = 0;
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < n; i += stride) a[i] = double2{1.0 + i, 2.0};
    size_t last = t0 + (n - 1 - t0) / stride * stride;
    for (size_t i = last + stride; i >= stride + t0; ) {
        i -= stride;
        const double2 v = a[i];
        s += v.x + v.y;
        if (i < stride) break;
    }
    if (s == 12345.678) out[0] = s;
}

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    double2 *a, *b; double *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int grid : {256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
        const float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
        const float r4 = timeit([&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, a, n, out); });
        const float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
        const float f = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, b, n); });
        printf("grid %6d: read %6.2f TB/s  read x4 %6.2f TB/s  copy %6.2f TB/s (read + write)  fill %6.2f TB/s\n", grid,
               bytes / r / 1e9, bytes / r4 / 1e9, 2.0 * bytes / c / 1e9, bytes / f / 1e9);
    }
    for (size_t mb : {16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t nn = (mb << 20) / 16;
        const int passes = (int)(4096 / mb) > 1 ? (int)(4096 / mb) : 1;
        const float r = timeit([&] { hipLaunchKernelGGL(k_reread, dim3(2048), dim3(256), 0, 0, a, nn, passes, out); });
        const float w = timeit([&] { hipLaunchKernelGGL(k_write_then_read, dim3(2048), dim3(256), 0, 0, a, nn, out); });
        printf("footprint %5zu MiB: re-read x%-3d %6.2f TB/s   write then read back in reverse %6.2f TB/s (write + read)\n", mb,
               passes, (double)(mb << 20) * passes / r / 1e9, 2.0 * (double)(mb << 20) / w / 1e9);
    }
    const float m = timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpyAsync device-to-device: %6.2f TB/s (read + write)\n", 2.0 * bytes / m / 1e9);
    return 0;
}

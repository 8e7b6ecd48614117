// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for 8-byte-per-lane coalesced accesses (the access width
// of the GEMM staging loads) on gfx950: reads 1 GiB and writes 256 MiB of doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const double *x, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
    if (s == 123.456) out[0] = s;
}
__global__ void write8(double *y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.0;
}
int main()
{
    const size_t n = (size_t)1 << 27;       // 1 GiB of doubles (> 256 MiB Infinity Cache)
    double *x, *o; (void)hipMalloc(&x, n * 8); (void)hipMalloc(&o, 64);
    (void)hipMemset(x, 0, n * 8);
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, x, n, o);
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, x, n / 4);
    (void)hipDeviceSynchronize();
    printf("read8: %zu bytes, write8: %zu bytes\n", n * 8, n / 4 * 8);
    return 0;
}

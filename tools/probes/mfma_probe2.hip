// FP64 MFMA probes with the operand/accumulator pattern of the GEMM kernels (distinct registers), no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k444(int iters, double *out)
{
    double acc[64], a[4], b[16];
    for (int i = 0; i < 64; i++) acc[i] = 0;
    for (int i = 0; i < 4; i++) a[i] = 1.0 + (threadIdx.x + i) * 1e-9;
    for (int i = 0; i < 16; i++) b[i] = 1.0 - (threadIdx.x + i) * 1e-9;
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[j * 4 + i] = __builtin_amdgcn_mfma_f64_4x4x4f64(b[j], a[i], acc[j * 4 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] += 1e-12;      // keep the loop from being hoisted
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < 64; j++) s += acc[j];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0) / (64.0 * iters); out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0; }
}

__global__ __launch_bounds__(256) void k16(int iters, double *out)
{
    v4d acc[16]; double a[4], b[4];
    for (int i = 0; i < 16; i++) acc[i] = (v4d){0, 0, 0, 0};
    for (int i = 0; i < 4; i++) { a[i] = 1.0 + (threadIdx.x + i) * 1e-9; b[i] = 1.0 - (threadIdx.x + i) * 1e-9; }
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[j * 4 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[j * 4 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] += 1e-12;
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < 16; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0) / (16.0 * iters); out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0; }
}

template <class K>
void run(const char *name, K kern, int grid, int iters, double flop_per_wave_iter)
{
    double *d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, 10, d);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double h[3]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double tf = flop_per_wave_iter * grid * 4.0 * iters / (ms * 1e-3) * 1e-12;
    printf("%-40s grid %5d: %8.2f TFLOP/s  cycles/MFMA (per wave) %7.1f  clock %6.0f MHz\n", name, grid, tf, h[1], h[2]);
    (void)hipFree(d);
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int cu = p.multiProcessorCount;
    for (int w = 1; w <= 4; w *= 2) {
        char nm[64];
        snprintf(nm, sizeof nm, "4x4x4  64 acc, 4+16 operands, %d wave/SIMD", w);
        run(nm, k444, cu * w, 2000, 64.0 * 512);
        snprintf(nm, sizeof nm, "16x16x4 16 acc, 4+4 operands, %d wave/SIMD", w);
        run(nm, k16, cu * w, 2000, 16.0 * 2048);
    }
    return 0;
}

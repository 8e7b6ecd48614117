// FP64 MFMA issue-rate probes for gfx950 (results recorded in profiles/).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(int iters, double *out)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < NACC; j++) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0) / ((double)NACC * iters); out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k4(int iters, double *out)
{
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = 0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[j], 0, 0, 0);
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < NACC; j++) s += acc[j];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0) / ((double)NACC * iters); out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0; }
}

// plain FP64 FMA (vector) for comparison: v_fma_f64, 64 lanes
template <int NACC>
__global__ __launch_bounds__(256) void kfma(int iters, double *out)
{
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9 * threadIdx.x;
    uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_fma(acc[j], x, y);
    }
    uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int j = 0; j < NACC; j++) s += acc[j];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(c1 - c0) / ((double)NACC * iters); out[2] = (double)(c1 - c0) / (double)(r1 - r0) * 100.0; }
}

template <class F>
void run(const char *name, F launch, int grid, int block, int iters, double flop_per_thread_iter_total)
{
    double *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(grid, block, 100, d);
    hipEventRecord(e0, 0);
    launch(grid, block, iters, d);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double h[3]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double tf = flop_per_thread_iter_total * iters / (ms * 1e-3) * 1e-12;
    printf("%-44s grid %5d block %4d : %8.2f TFLOP/s  cycles/instr(per wave) %7.1f  clock %6.0f MHz  (%.2f ms)\n", name, grid, block, tf, h[1], h[2], ms);
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", p.gcnArchName, cu, p.clockRate / 1000);
    const int it = 20000;
#define L16(N) [](int g, int b, int i, double *d) { hipLaunchKernelGGL(k16<N>, dim3(g), dim3(b), 0, 0, i, d); }
#define L4(N) [](int g, int b, int i, double *d) { hipLaunchKernelGGL(k4<N>, dim3(g), dim3(b), 0, 0, i, d); }
#define LF(N) [](int g, int b, int i, double *d) { hipLaunchKernelGGL(kfma<N>, dim3(g), dim3(b), 0, 0, i, d); }
    // waves per launch = grid * block/64 ; flops = waves * NACC * 2048 per iter
    run("mfma_f64_16x16x4, 1 wave/SIMD, 4 acc", L16(4), cu, 256, it, (double)cu * 4 * 4 * 2048);
    run("mfma_f64_16x16x4, 1 wave/SIMD, 8 acc", L16(8), cu, 256, it, (double)cu * 4 * 8 * 2048);
    run("mfma_f64_16x16x4, 1 wave/SIMD, 1 acc (dependent)", L16(1), cu, 256, it, (double)cu * 4 * 1 * 2048);
    run("mfma_f64_16x16x4, 2 waves/SIMD, 4 acc", L16(4), cu * 2, 256, it, (double)cu * 2 * 4 * 4 * 2048);
    run("mfma_f64_16x16x4, 4 waves/SIMD, 4 acc", L16(4), cu * 4, 256, it, (double)cu * 4 * 4 * 4 * 2048);
    run("mfma_f64_16x16x4, 1 wave on the chip, 8 acc", L16(8), 1, 64, it, 1.0 * 8 * 2048);
    run("mfma_f64_4x4x4 (4 blocks), 1 wave/SIMD, 8 acc", L4(8), cu, 256, it, (double)cu * 4 * 8 * 512);
    run("mfma_f64_4x4x4 (4 blocks), 2 waves/SIMD, 8 acc", L4(8), cu * 2, 256, it, (double)cu * 2 * 4 * 8 * 512);
    run("v_fma_f64, 2 waves/SIMD, 8 acc", LF(8), cu * 2, 256, it, (double)cu * 2 * 4 * 8 * 128);
    run("v_fma_f64, 4 waves/SIMD, 8 acc", LF(8), cu * 4, 256, it, (double)cu * 4 * 4 * 8 * 128);
    return 0;
}

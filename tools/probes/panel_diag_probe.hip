// Probe (round 3): where the time of k_panel_diag goes.  One 256 x 256 window inside a tall panel (lda rows), the kernel
// from the product's own source with clock stamps at its phase boundaries (thread 0, s_memrealtime: 100 MHz), launched
// `iters` times on fresh copies of the same SPD block.  Also the four k_potrf64 launches that factorize the same four
// diagonal blocks one by one (what the per-block chain paid for them), for comparison.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I suitesparse_amd/csrc tools/probes/panel_diag_probe.hip -o /tmp/pdp && /tmp/pdp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ unsigned long long *g_stamps;
#define MI355_PDIAG_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#include "schedule.hpp"
#include "kernels.hip.hpp"
using namespace mi355;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int lda = argc > 1 ? atoi(argv[1]) : 20000, ncols = argc > 2 ? atoi(argv[2]) : 256, iters = 20;
    std::vector<double> A((size_t)lda * ncols, 0.0);
    uint64_t x = 88172645463325252ull;
    for (int j = 0; j < ncols; j++)
        for (int i = j; i < ncols; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            A[i + (size_t)j * lda] = (i == j) ? 2.0 * ncols : (double)(x >> 11) / 9007199254740992.0 - 0.5;
        }
    double *dA, *dA0, *dD; int *dinfo; unsigned long long *dst; PanelGroup *dg; PotrfTask *dp;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dA0, A.size() * 8)); CK(hipMalloc(&dD, 4 * 64 * 64 * 8)); CK(hipMalloc(&dinfo, 16));
    CK(hipMalloc(&dst, 64 * 8)); CK(hipMalloc(&dg, sizeof(PanelGroup))); CK(hipMalloc(&dp, 4 * sizeof(PotrfTask)));
    CK(hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dinfo, 0, 16)); CK(hipMemset(dst, 0, 64 * 8));
    PanelGroup g; g.a_off = 0; for (int b = 0; b < 4; b++) g.dinv_off[b] = b * 4096; g.wg0 = 0; g.lda = lda; g.ncols = ncols; g.s = 0; g.col0 = 0; g.below = lda - ncols; g.pad = 0;
    CK(hipMemcpy(dg, &g, sizeof g, hipMemcpyHostToDevice));
    PotrfTask pt[4];
    for (int b = 0; b < 4; b++) { pt[b] = PotrfTask(); pt[b].a_off = 64 * b + (int64_t)64 * b * lda; pt[b].dinv_off = b * 4096; pt[b].lda = lda; pt[b].nb = 64; pt[b].s = 0; pt[b].col0 = 64 * b; }
    CK(hipMemcpy(dp, pt, sizeof pt, hipMemcpyHostToDevice));
    unsigned long long *null_st = nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9, sum = 0; std::vector<unsigned long long> st(64), acc(64, 0);
    for (int it = 0; it < iters; it++) {
        CK(hipMemcpy(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof dst));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_panel_diag, dim3(1), dim3(256), 0, 0, (const PanelGroup *)dg, dA, dD, dinfo);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { best = std::min(best, (double)ms); sum += ms; }
        CK(hipMemcpy(st.data(), dst, 64 * 8, hipMemcpyDeviceToHost));
        if (it >= 2) for (int i = 0; i < 24; i++) acc[i] += st[i] - st[0];
    }
    printf("k_panel_diag lda %d ncols %d: best %.1f us, mean %.1f us (events around one launch)\n", lda, ncols, best * 1e3, sum / (iters - 2) * 1e3);
    const char *nm[6] = {"block start", "strips loaded", "updates done", "block in LDS", "potrf core done", "trsm stored"};
    for (int i = 0; i < 6 * ((ncols + 63) / 64); i++)
        printf("  block %d  %-16s  +%7.2f us\n", i / 6, nm[i % 6], (double)acc[i] / (iters - 2) * 0.01);
    // the four diagonal blocks by k_potrf64, one launch each (values irrelevant: timing only)
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &null_st, sizeof null_st));
    best = 1e9;
    for (int it = 0; it < iters; it++) {
        CK(hipMemcpy(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int b = 0; b < 4; b++) hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(256), 0, 0, (const PotrfTask *)(dp + b), dA, dD, dinfo, (const int32_t *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms);
    }
    printf("4 x k_potrf64 (64 x 64 each, dependent launches): best %.1f us\n", best * 1e3);
    int info = 0; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    printf("info %d\n", info);
    return 0;
}

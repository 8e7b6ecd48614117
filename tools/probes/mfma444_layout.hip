// Empirical lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: for every (la, lb) pair one wave runs
// the instruction with A one-hot at lane la and B one-hot at lane lb and reports which lane of D is 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int *out)
{
    const int pair = blockIdx.x, la = pair / 64, lb = pair % 64, lane = threadIdx.x;
    double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[pair] = lane + 64 * (int)d;   // d should be exactly 1
}
int main()
{
    int *d; hipMalloc(&d, 4096 * sizeof(int)); hipMemset(d, 0xff, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
    std::vector<int> h(4096); hipMemcpy(h.data(), d, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // for each la: list of (lb -> ld)
    for (int la = 0; la < 64; la++) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf(" B%d->D%d", lb, h[la * 64 + lb] % 64);
        printf("\n");
    }
    return 0;
}

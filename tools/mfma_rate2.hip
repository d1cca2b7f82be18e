// mfma_rate2.hip -- 4x4x1 chain with A from NW distinct AGPRs (pinned) or VGPRs, B from 8 VGPRs, D in VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int WAVES, int NW, bool PIN>
__global__ __launch_bounds__(WAVES * 64) void kreal(const float *src, float *out, int iters) {
    float w[NW];
    float hv[8];
    for (int i = 0; i < NW; ++i) w[i] = src[(i * 64 + threadIdx.x) & 4095];
    for (int i = 0; i < 8; ++i) hv[i] = src[(i * 7 + threadIdx.x) & 4095];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it) {
        if (PIN) {
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("" : "+a"(w[i]));
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) asm volatile("" : "+v"(w[i]));
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], hv[i & 7], acc[i & 3], 0, 0, 0);
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(src, 0, 4096 * 4);
    const int iters = 4000, grid = 128;
    float a = timeit([&] { hipLaunchKernelGGL((kreal<4, 256, true>), dim3(grid), dim3(256), 0, 0, src, out, iters); });
    float b = timeit([&] { hipLaunchKernelGGL((kreal<8, 128, true>), dim3(grid), dim3(512), 0, 0, src, out, iters); });
    float c = timeit([&] { hipLaunchKernelGGL((kreal<4, 128, false>), dim3(grid), dim3(256), 0, 0, src, out, iters); });
    float d = timeit([&] { hipLaunchKernelGGL((kreal<4, 128, true>), dim3(grid), dim3(256), 0, 0, src, out, iters); });
    printf("per iteration: 4 waves x 256 MFMA (A in AGPR) %.3f us | 8 waves x 128 MFMA (A in AGPR) %.3f us | 4 waves x 128 (A in VGPR) %.3f us | 4 waves x 128 (A in AGPR) %.3f us\n",
           a * 1e3 / iters, b * 1e3 / iters, c * 1e3 / iters, d * 1e3 / iters);
    return 0;
}

// mfma_rate.hip -- issue-rate probe for the f32 MFMA forms (one wave per SIMD, 4 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k4x4(float *out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 256 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void k16(float *out, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 256 * 4);
    const int iters = 4000;
    for (int grid : {1, 128, 256}) {
        float t1 = timeit([&] { hipLaunchKernelGGL(k4x4<1>, dim3(grid), dim3(256), 0, 0, out, iters); });
        float t2 = timeit([&] { hipLaunchKernelGGL(k4x4<2>, dim3(grid), dim3(256), 0, 0, out, iters); });
        float t4 = timeit([&] { hipLaunchKernelGGL(k4x4<4>, dim3(grid), dim3(256), 0, 0, out, iters); });
        float t8 = timeit([&] { hipLaunchKernelGGL(k4x4<8>, dim3(grid), dim3(256), 0, 0, out, iters); });
        float t16 = timeit([&] { hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("grid %3d: 256x mfma_4x4x1 per iter: 1 acc %.3f us, 2 acc %.3f us, 4 acc %.3f us, 8 acc %.3f us | 64x mfma_16x16x4 (same MACs): %.3f us per iter\n",
               grid, t1 * 1e3 / iters, t2 * 1e3 / iters, t4 * 1e3 / iters, t8 * 1e3 / iters, t16 * 1e3 / iters);
    }
    return 0;
}

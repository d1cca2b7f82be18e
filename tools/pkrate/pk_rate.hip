// Is packed f32 (v_pk_fma_f32: two f32 per lane and instruction) issued at the rate of v_fma_f32 on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/pkrate/pk_rate tools/pkrate/pk_rate.hip && tools/pkrate/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    v2f a[8]; float s[16];
    for (int i = 0; i < 8; ++i) { a[i] = v2f{threadIdx.x * 1e-3f + i, 1.0f + i}; }
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 1e-3f + i;
    const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);      // 64 v_pk_fma_f32 = 128 fma
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = __builtin_fmaf(s[i], 1.0001f, 0.5f);       // 64 v_fma_f32
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) t += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pk = 0; pk < 2; ++pk) {
        const int iters = 20000, blocks = 2048;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters * 64;   // wave-instructions
        printf("%s: %.2f ms, %.2f G wave-instr/s, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32", ms, instr / ms / 1e6,
               instr * 64 * 2 * (pk ? 2 : 1) / ms / 1e9);
    }
    return 0;
}

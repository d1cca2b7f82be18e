// Do matrix (MFMA) and vector (VALU) instructions of the two waves that share a SIMD execute concurrently on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/coexec_probe tools/coexec_probe.hip && tools/coexec_probe
// One 512-thread workgroup per CU (waves w and w + 4 share a SIMD).  Waves 0-3 issue a chain of v_mfma_f32_16x16x32_f16,
// waves 4-7 a chain of vector instructions (v_fma_f32, or v_exp_f32 for the transcendental unit); each role is timed alone
// and together.  together == max(alone) -> the pipes overlap; together == sum -> they share the issue port for the duration.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int VKIND>
__global__ __launch_bounds__(512) void probe(float *out, int iters, int run_mfma, int run_valu) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    float v0 = threadIdx.x * 1e-3f, v1 = 1.0f + v0, v2 = 0.5f, v3 = 0.25f;
    if (wave < 4) {
        if (run_mfma)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {   // two independent accumulator chains
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc1, 0, 0, 0);
                }
            }
    } else if (run_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {       // four independent chains
                if (VKIND == 0) {
                    v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
                    v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
                } else {
                    v0 = __builtin_amdgcn_exp2f(v0) * 0.5f; v1 = __builtin_amdgcn_exp2f(v1) * 0.5f;
                    v2 = __builtin_amdgcn_exp2f(v2) * 0.5f; v3 = __builtin_amdgcn_exp2f(v3) * 0.5f;
                }
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + v0 + v1 + v2 + v3;
}

template <int VKIND>
static float run(float *d, int iters, int m, int v) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<VKIND>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VKIND>, dim3(256), dim3(512), 0, 0, d, iters, m, v);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        const float m = kind ? run<1>(d, iters, 1, 0) : run<0>(d, iters, 1, 0);
        const float v = kind ? run<1>(d, iters, 0, 1) : run<0>(d, iters, 0, 1);
        const float both = kind ? run<1>(d, iters, 1, 1) : run<0>(d, iters, 1, 1);
        const double mf = 32.0 * iters, vf = (kind ? 256.0 : 128.0) * iters;   // instructions per wave (kind 1: 128 v_exp_f32 + 128 v_mul_f32 per iteration)
        printf("{\"vector_kind\": \"%s\", \"mfma_only_ms\": %.3f, \"valu_only_ms\": %.3f, \"together_ms\": %.3f, \"sum_ms\": %.3f, "
               "\"mfma_per_wave\": %.0f, \"valu_instr_per_wave\": %.0f}\n",
               kind ? "v_exp_f32 + v_mul_f32" : "v_fma_f32", m, v, both, m + v, mf, vf);
    }
    return 0;
}

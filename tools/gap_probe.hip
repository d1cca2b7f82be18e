// How many vector instructions fit in the gap of a v_mfma_f32_16x16x32_f16 issued by the SAME wave (one wave per SIMD)?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/gap_probe tools/gap_probe.hip && tools/gap_probe
// One 256-thread workgroup per CU; every wave runs [1 MFMA + K independent v_fma_f32 (or K v_exp_f32)] x 64 per iteration, the
// stream pinned with sched_barrier(0).  Reported: shader cycles per gap (s_memtime) for K = 0 .. 16, for MFMAs on independent
// accumulators and on ONE dependent accumulator chain, weights (A operand) in AGPRs as in lstm_rec16h_kernel; the same for
// v_mfma_f32_32x32x16_f16 (twice the FLOP in 32 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int K, int KIND, bool CHAIN, bool BIG>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big[2] = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    asm volatile("" : "+a"(a));
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            asm volatile("" : "+a"(a));
            if constexpr (BIG) {
                f32x16 &c = big[CHAIN ? 0 : (g & 1)];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
            } else {
                f32x4 &c = acc[CHAIN ? 0 : (g & 3)];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
                else v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s + big[0][0] + big[1][5];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int KIND, bool CHAIN, bool BIG>
static double run(float *d, unsigned long long *c, int iters) {
    hipLaunchKernelGGL((probe<K, KIND, CHAIN, BIG>), dim3(256), dim3(256), 0, 0, d, c, iters);
    hipLaunchKernelGGL((probe<K, KIND, CHAIN, BIG>), dim3(256), dim3(256), 0, 0, d, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    return m / 256.0 / (64.0 * iters);
}

template <int KIND, bool CHAIN, bool BIG>
static void sweep(float *d, unsigned long long *c, int iters, const char *name) {
    printf("{\"mfma\": \"%s\", \"vector\": \"%s\", \"accumulators\": \"%s\", \"cycles_per_gap_by_K_0_1_2_3_4_6_8_12_16\": [%.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f]}\n",
           BIG ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", name, CHAIN ? "one dependent chain" : "independent",
           run<0, KIND, CHAIN, BIG>(d, c, iters), run<1, KIND, CHAIN, BIG>(d, c, iters), run<2, KIND, CHAIN, BIG>(d, c, iters), run<3, KIND, CHAIN, BIG>(d, c, iters),
           run<4, KIND, CHAIN, BIG>(d, c, iters), run<6, KIND, CHAIN, BIG>(d, c, iters), run<8, KIND, CHAIN, BIG>(d, c, iters), run<12, KIND, CHAIN, BIG>(d, c, iters),
           run<16, KIND, CHAIN, BIG>(d, c, iters));
}

int main() {
    float *d;
    unsigned long long *c;
    (void)hipMalloc(&d, 256 * 256 * sizeof(float));
    (void)hipMalloc(&c, 256 * sizeof(unsigned long long));
    const int iters = 2000;
    sweep<0, false, false>(d, c, iters, "v_fma_f32");
    sweep<0, true, false>(d, c, iters, "v_fma_f32");
    sweep<1, false, false>(d, c, iters, "v_exp_f32");
    sweep<0, false, true>(d, c, iters, "v_fma_f32");
    sweep<0, true, true>(d, c, iters, "v_fma_f32");
    sweep<1, false, true>(d, c, iters, "v_exp_f32");
    return 0;
}

// Does a v_mfma_f32_16x16x32_f16 stream keep its 16-cycle issue rate when every MFMA reads a DIFFERENT A (and B) register tuple,
// from AGPRs or from VGPRs?  (lstm_rec16h_kernel reads 32-64 resident weight tuples in turn; tools/gap_probe.hip reuses one.)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/opnd_probe tools/opnd_probe.hip && tools/opnd_probe
// One wave per SIMD (256-thread workgroups, one per CU), 64 MFMAs per iteration, shader cycles per MFMA by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// NA: distinct A tuples cycled through; AGPR: A tuples live in AGPRs; NB: distinct B tuples; NC: accumulators (1 = one dependent chain)
template <int NA, bool AGPR, int NB, int NC>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[NC];
    for (int i = 0; i < NC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a[NA], b[NB];
    for (int t = 0; t < NA; ++t)
        for (int i = 0; i < 8; ++i) a[t][i] = (_Float16)(threadIdx.x * 0.001f + i + t);
    for (int t = 0; t < NB; ++t)
        for (int i = 0; i < 8; ++i) b[t][i] = (_Float16)(1.0f + i * 0.01f + t);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            if (AGPR) asm volatile("" : "+a"(a[t]));
            else asm volatile("" : "+v"(a[t]));
        }
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            acc[g % NC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g % NA], b[(g / 2) % NB], acc[g % NC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NA, bool AGPR, int NB, int NC>
static void run(float *d, unsigned long long *c, int iters) {
    hipLaunchKernelGGL((probe<NA, AGPR, NB, NC>), dim3(256), dim3(256), 0, 0, d, c, iters);
    hipLaunchKernelGGL((probe<NA, AGPR, NB, NC>), dim3(256), dim3(256), 0, 0, d, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    printf("{\"A_tuples\": %d, \"A_in\": \"%s\", \"B_tuples\": %d, \"accumulators\": %d, \"cycles_per_mfma\": %.1f}\n", NA, AGPR ? "AGPR" : "VGPR", NB, NC,
           m / 256.0 / (64.0 * iters));
}

int main() {
    float *d;
    unsigned long long *c;
    (void)hipMalloc(&d, 256 * 256 * sizeof(float));
    (void)hipMalloc(&c, 256 * sizeof(unsigned long long));
    const int iters = 2000;
    run<1, true, 1, 4>(d, c, iters);
    run<16, true, 1, 4>(d, c, iters);
    run<32, true, 1, 4>(d, c, iters);
    run<32, true, 8, 4>(d, c, iters);
    run<32, true, 8, 2>(d, c, iters);
    run<32, true, 8, 1>(d, c, iters);
    run<1, false, 1, 4>(d, c, iters);
    run<16, false, 1, 4>(d, c, iters);
    run<16, false, 8, 4>(d, c, iters);
    run<16, false, 8, 1>(d, c, iters);
    run<1, true, 8, 4>(d, c, iters);
    return 0;
}

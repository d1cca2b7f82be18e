// What does a global store cost the MFMA stream of the wave that issues it (one wave per SIMD, as in gemm_f16p_ws_kernel)?
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/store_probe tools/store_probe.hip && tools/store_probe
// One 256-thread workgroup per CU; every wave runs [2 x v_mfma_f32_32x32x16_f16 (+ one store every Nth group)] x 64 per iteration (stream
// pinned with sched_barrier(0)); N = 1, 2, 4, 8, 16 (gemm_f16p_ws_kernel: one 16-byte store per 8 groups).  Store forms: global_store_dword / dwordx2 / dwordx4, plain and non-temporal, each lane writing
// consecutive 4 / 8 / 16 bytes (whole 128-byte lines per 32 / 16 / 8 lanes), into a ring of `ring_kb` KiB per wave (small: stays in L2;
// large: streams to HBM).  Reported: the extra shader cycles ONE store adds to the wave's stream at each density (a group of two MFMAs alone is 64).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int K, int WIDTH, bool NT, int EVERY>
__global__ __launch_bounds__(256, 1) void probe(float *buf, size_t ring_floats, unsigned long long *cyc, float *out, int iters) {
    f32x16 acc[2] = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    asm volatile("" : "+a"(a));
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *ring = buf + (size_t)wave_global * ring_floats + lane * WIDTH;
    f32x4 v = {1.f + lane, 2.f, 3.f, 4.f};
    size_t pos = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            asm volatile("" : "+a"(a));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < ((g % EVERY) == 0 ? K : 0); ++k) {
                float *p = ring + pos;
                if (WIDTH == 4) { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); else *reinterpret_cast<f32x4 *>(p) = v; }
                if (WIDTH == 2) { if (NT) __builtin_nontemporal_store(f32x2{v.x, v.y}, reinterpret_cast<f32x2 *>(p)); else *reinterpret_cast<f32x2 *>(p) = f32x2{v.x, v.y}; }
                if (WIDTH == 1) { if (NT) __builtin_nontemporal_store(v.x, p); else *p = v.x; }
                pos += 64 * WIDTH;
                if (pos >= ring_floats) pos = 0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][5];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int WIDTH, bool NT, int EVERY>
static double run(float *buf, size_t ring_floats, unsigned long long *c, float *out, int iters) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<K, WIDTH, NT, EVERY>), dim3(256), dim3(256), 0, 0, buf, ring_floats, c, out, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 256; ++i) m += (double)h[i];
    return m / 256.0 / (64.0 * iters);
}

template <int WIDTH, bool NT>
static void line(float *buf, size_t ring_kb, unsigned long long *c, float *out, int iters) {
    const size_t rf = ring_kb * 256;
    const double base = run<0, WIDTH, NT, 1>(buf, rf, c, out, iters);
    const double e1 = run<1, WIDTH, NT, 1>(buf, rf, c, out, iters), e2 = run<1, WIDTH, NT, 2>(buf, rf, c, out, iters), e4 = run<1, WIDTH, NT, 4>(buf, rf, c, out, iters),
                 e8 = run<1, WIDTH, NT, 8>(buf, rf, c, out, iters), e16 = run<1, WIDTH, NT, 16>(buf, rf, c, out, iters);
    printf("{\"store\": \"global_store_dword%s%s\", \"ring_KiB_per_wave\": %zu, \"cycles_per_2_mfma_group_without_stores\": %.1f, "
           "\"extra_cycles_per_store_at_one_store_every_1_2_4_8_16_groups\": [%.1f, %.1f, %.1f, %.1f, %.1f]}\n",
           WIDTH == 4 ? "x4" : WIDTH == 2 ? "x2" : "", NT ? " nt" : "", ring_kb, base, (e1 - base) * 1, (e2 - base) * 2, (e4 - base) * 4, (e8 - base) * 8, (e16 - base) * 16);
}

int main() {
    float *buf, *out;
    unsigned long long *c;
    const size_t big_kb = 1024;   // 1 MiB per wave x 1024 waves = 1 GiB: streams to HBM
    (void)hipMalloc(&buf, big_kb * 1024 * 1024);
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&c, 256 * sizeof(unsigned long long));
    const int iters = 200;
    for (size_t kb : {(size_t)16, big_kb}) {
        line<4, false>(buf, kb, c, out, iters);
        line<4, true>(buf, kb, c, out, iters);
        line<2, true>(buf, kb, c, out, iters);
        line<1, true>(buf, kb, c, out, iters);
    }
    return 0;
}

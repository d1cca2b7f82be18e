// A matrix-pipe burner to run beside the feature kernel: does ANY MFMA-dense neighbour corrupt fbank_kernel's output, or only
// the 128x128-tile GEMM?   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libburner.so tools/mfma_burner.hip
#include <hip/hip_runtime.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int NACC, int LDS_KB>
__global__ __launch_bounds__(256) void burn(float *out, int iters) {
    __shared__ float lds[LDS_KB * 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        a[0] = (_Float16)lds[(threadIdx.x + it) & 255];
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// vector-only and barrier-free variants
template <bool USE_LDS>
__global__ __launch_bounds__(256) void burn_valu(float *out, int iters) {
    __shared__ float lds[USE_LDS ? 40 * 256 : 1];
    float v0 = threadIdx.x * 1e-3f, v1 = 1.f + v0, v2 = 0.5f, v3 = 0.25f;
    if (USE_LDS) { lds[threadIdx.x] = v0; __syncthreads(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
            v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
        }
        if (USE_LDS) { v0 += lds[(threadIdx.x + it) & 255]; __syncthreads(); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
}
template <int NACC>
__global__ __launch_bounds__(256) void burn_nolds(float *out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LDS_KB, bool BAR, bool RD>
__global__ __launch_bounds__(256) void burn_mix(float *out, int iters) {
    __shared__ float lds[LDS_KB * 256];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.01f); }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if (RD) a[0] = (_Float16)lds[(threadIdx.x + it) & 255];
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// memory-only neighbours: every block streams its own 4 MiB slice of `out` (iters passes): kind 10 writes it, kind 11 reads it
template <bool WRITE>
__global__ __launch_bounds__(256) void burn_mem(float *out, int iters) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    f32x4 *p = reinterpret_cast<f32x4 *>(out) + (size_t)blockIdx.x * (4u << 20) / 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
        for (int i = threadIdx.x; i < (4 << 20) / 16; i += 256) {
            if (WRITE) __builtin_nontemporal_store(f32x4{(float)it, 1.f, 2.f, 3.f}, p + i);
            else acc += __builtin_nontemporal_load(p + i);
        }
    if (!WRITE && acc[0] == 12345.678f) out[0] = acc[1];
}

extern "C" int burner_launch(int kind, float *out, int blocks, int iters, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL((burn<4, 28>), dim3(blocks), dim3(256), 0, s, out, iters);        // 64 accumulator registers, 28 KiB (the 128x64 GEMM's shape)
    else if (kind == 1) hipLaunchKernelGGL((burn<8, 40>), dim3(blocks), dim3(256), 0, s, out, iters);   // 128 accumulator registers, 40 KiB (the 128x128 GEMM's)
    else if (kind == 2) hipLaunchKernelGGL((burn<8, 28>), dim3(blocks), dim3(256), 0, s, out, iters);
    else if (kind == 3) hipLaunchKernelGGL((burn_valu<true>), dim3(blocks), dim3(256), 0, s, out, iters);    // no MFMA, LDS + barriers
    else if (kind == 4) hipLaunchKernelGGL((burn_valu<false>), dim3(blocks), dim3(256), 0, s, out, iters);   // no MFMA, no LDS
    else if (kind == 5) hipLaunchKernelGGL((burn_nolds<4>), dim3(blocks), dim3(256), 0, s, out, iters);      // MFMA only
    else if (kind == 6) hipLaunchKernelGGL((burn_mix<28, true, false>), dim3(blocks), dim3(256), 0, s, out, iters);    // MFMA + barrier, LDS allocated but idle
    else if (kind == 7) hipLaunchKernelGGL((burn_mix<28, false, true>), dim3(blocks), dim3(256), 0, s, out, iters);    // MFMA + LDS reads, no barrier
    else if (kind == 8) hipLaunchKernelGGL((burn_mix<28, false, false>), dim3(blocks), dim3(256), 0, s, out, iters);   // MFMA, LDS allocated, nothing else
    else if (kind == 10) hipLaunchKernelGGL((burn_mem<true>), dim3(blocks), dim3(256), 0, s, out, iters);     // HBM writes only (out: blocks x 4 MiB)
    else if (kind == 11) hipLaunchKernelGGL((burn_mem<false>), dim3(blocks), dim3(256), 0, s, out, iters);    // HBM reads only
    else hipLaunchKernelGGL((burn_mix<1, true, true>), dim3(blocks), dim3(256), 0, s, out, iters);                     // MFMA + barrier + reads, 1 KiB of LDS
    return (int)hipGetLastError();
}

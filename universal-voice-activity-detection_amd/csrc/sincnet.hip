// sincnet.hip -- SincNet front end of PyanNet (reference: src/models/blocks/sincnet.py:33-103, called from
// PyanNet.forward, src/models/segmentation/PyanNet.py:177).  waveform (B, S) -> features (B, frames, 60):
//
//   wav_norm1d (InstanceNorm1d(1))                                   -> wav_stats_kernel   (scale, shift per b)
//   sinc filter bank conv (80 x 251, stride 10), |.|, MaxPool1d(3)   -> conv_pool_kernel<3, true>
//   InstanceNorm1d(80) + leaky_relu                                  -> norm_finalize_kernel (scale, shift per (b, c)),
//                                                                       applied when the NEXT conv stages its input
//   Conv1d(80, 60, 5), MaxPool1d(3)                                  -> conv_pool_kernel<2, false>
//   InstanceNorm1d(60) + leaky_relu
//   Conv1d(60, 60, 5), MaxPool1d(3)                                  -> conv_pool_kernel<2, false>
//   InstanceNorm1d(60) + leaky_relu, "b f t -> b t f"                -> sinc_out_kernel
//
// The convolutions are implicit GEMMs on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32 products and
// a k-ordered accumulation): a workgroup of 3 waves owns 96 consecutive conv positions (= 32 pooled outputs)
// x all output channels; the whole transposed filter matrix ([K/2][N][2] so that one ds_read_b32 per lane
// feeds the B operand without bank conflicts) stays in LDS across the tiles a workgroup walks, the input
// window is staged with the previous layer's normalisation + leaky_relu folded in, and the epilogue does
// bias, |.|, the 3:1 max pool and the per-tile (sum, M2) statistics the instance norm of the next stage
// needs -- written as per-tile partials and reduced in tile order, so results do not depend on scheduling.
#include "uvad_internal.h"

namespace uvad {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CT = 96;        // conv positions per tile (3 waves x 32 MFMA rows) = 32 pooled outputs
constexpr int YS = 97;        // LDS row stride of the conv-output staging tile

__global__ __launch_bounds__(256) void wav_stats_kernel(const float *wav, long long S, long long row_stride, const float *gamma,
                                                        const float *beta, float eps, float *scale, float *shift) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *x = wav + (size_t)b * row_stride;
    double s = 0.0, ss = 0.0;
    for (long long i = tid; i < S; i += 256) {
        const double v = x[i];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    __shared__ double red[2][4];
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = s / (double)S;
        double var = ss / (double)S - mean * mean;
        if (var < 0.0) var = 0.0;
        const double sc = (double)gamma[0] / sqrt(var + (double)eps);
        scale[b] = (float)sc;
        shift[b] = (float)((double)beta[0] - mean * sc);
    }
}

template <int NT, bool CIN1>
__global__ __launch_bounds__(192) void conv_pool_kernel(SincConvArgs a) {
    constexpr int NW = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wt = smem;                                  // [Kp/2][NW][2]
    float *xy = smem + (size_t)a.Kp * NW;              // input window [rows][XW] / conv-output tile [NW][YS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int b = blockIdx.y;
    const int XW = (CT - 1) * a.stride + a.Kw;
    const int rows = a.Cin + (a.Kp > a.Ktot);          // one zero row when the padded K reads past the last channel

    {   // filter matrix -> LDS once per workgroup (float4; Kp*NW is a multiple of 64)
        const float4 *src = reinterpret_cast<const float4 *>(a.Wt2);
        float4 *dst = reinterpret_cast<float4 *>(wt);
        const int n4 = a.Kp * NW / 4;
        for (int i = tid; i < n4; i += 192) dst[i] = src[i];
    }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int c0 = tile * CT;
        const long long x0 = (long long)c0 * a.stride;
        __syncthreads();   // weights staged / previous tile's pooled reads of xy are complete
        for (int ci = wave; ci < rows; ci += 3) {
            const bool real = ci < a.Cin;
            const float sc = real ? a.in_scale[(size_t)b * a.Cin + ci] : 0.f;
            const float sh = real ? a.in_shift[(size_t)b * a.Cin + ci] : 0.f;
            const float *src = a.in + (size_t)b * a.in_bstride + (size_t)(real ? ci : 0) * a.Lin;
            for (int x = lane; x < XW + 1; x += 64) {
                const long long gx = x0 + x;
                float v = 0.f;
                if (real && x < XW && gx < a.Lin) {
                    v = __builtin_fmaf(src[gx], sc, sh);
                    if (a.in_lrelu) v = v >= 0.f ? v : v * a.slope;
                }
                if (x < XW || ci == rows - 1) xy[(size_t)ci * XW + x] = v;   // the one-past element exists only after the last row
            }
        }
        __syncthreads();

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // K loop, software pipelined over two register sets: the operands of step ks+1 are read from LDS
        // before the MFMAs of step ks are issued (sched_group_barrier pins that order), so the ~100-cycle LDS
        // latency hides under the 64-cycle MFMAs instead of between them.  Kp is a multiple of 4 => ksteps even.
        const float *bp = wt + (size_t)li * 2 + kk;
        int aoff = (wave * 32 + li) * a.stride + kk;
        int kw = kk;
        const int ksteps = a.Kp >> 1;
        float a0, a1, b0[NT], b1[NT];
#define UVAD_SN_LOAD(AV, BV, ks_)                                                          \
    {                                                                                      \
        AV = xy[aoff];                                                                     \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) BV[t] = bp[(size_t)(ks_) * (NW * 2) + t * 64]; \
        if (CIN1) {                                                                        \
            aoff += 2;                                                                     \
        } else {                                                                           \
            kw += 2;                                                                       \
            aoff += 2;                                                                     \
            if (kw >= a.Kw) { kw -= a.Kw; aoff += XW - a.Kw; }                             \
        }                                                                                  \
    }
#define UVAD_SN_MFMA(AV, BV) \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV[t], acc[t], 0, 0, 0);
        UVAD_SN_LOAD(a0, b0, 0)
        for (int ks = 0; ks + 2 < ksteps; ks += 2) {
            UVAD_SN_LOAD(a1, b1, ks + 1)
            UVAD_SN_MFMA(a0, b0)
            UVAD_SN_LOAD(a0, b0, ks + 2)
            UVAD_SN_MFMA(a1, b1)
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);   // DS reads of set 1
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);       // MFMAs of set 0
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);   // DS reads of set 0 (next pair)
            __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);       // MFMAs of set 1
        }
        UVAD_SN_LOAD(a1, b1, ksteps - 1)
        UVAD_SN_MFMA(a0, b0)
        UVAD_SN_MFMA(a1, b1)
#undef UVAD_SN_LOAD
#undef UVAD_SN_MFMA
        __syncthreads();   // all A reads of the input window are done: the region becomes the output tile

#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = t * 32 + li;
            const float bias = a.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                float v = acc[t][r] + bias;
                if (a.do_abs) v = __builtin_fabsf(v);
                xy[(size_t)n * YS + wave * 32 + row] = v;
            }
        }
        __syncthreads();

        const int p0 = tile * (CT / 3);
        for (int idx = tid; idx < NW * 32; idx += 192) {
            const int n = idx >> 5, p = idx & 31;
            const float *y = xy + (size_t)n * YS + 3 * p;
            const float m = __builtin_fmaxf(__builtin_fmaxf(y[0], y[1]), y[2]);
            const bool valid = n < a.Cout && p0 + p < a.Lpool;
            if (valid) a.out[((size_t)b * a.Cout + n) * a.Lpool + p0 + p] = m;
            // per-tile (sum, M2 about the tile mean): combined in tile order by norm_finalize_kernel (Chan's
            // update), which keeps the variance accurate when it is small against the mean (short inputs)
            float s = valid ? m : 0.f;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            const int nt = a.Lpool - p0 < 32 ? a.Lpool - p0 : 32;
            const float d = valid ? m - s / (float)nt : 0.f;
            float m2 = d * d;
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o);
            if (p == 0) {
                float *pp = a.partials + (((size_t)b * a.ntiles + tile) * NW + n) * 2;
                pp[0] = s;
                pp[1] = m2;
            }
        }
    }
}

// per-tile (sum, M2) partials, combined in tile order -> per (b, c) affine of the instance norm:
//   y = (x - mean) / sqrt(var + eps) * gamma + beta = x * scale + shift      (biased variance, torch InstanceNorm1d)
__global__ __launch_bounds__(128) void norm_finalize_kernel(const float *partials, int ntiles, int NW, int C, int L, const float *gamma,
                                                            const float *beta, float eps, float *scale, float *shift) {
    const int b = blockIdx.x, n = threadIdx.x;
    if (n >= C) return;
    double mean = 0.0, M2 = 0.0, cnt = 0.0;
    for (int t = 0; t < ntiles; ++t) {
        const float *pp = partials + (((size_t)b * ntiles + t) * NW + n) * 2;
        const double nt = (double)(L - 32 * t < 32 ? L - 32 * t : 32);
        const double mt = (double)pp[0] / nt, delta = mt - mean, tot = cnt + nt;
        M2 += (double)pp[1] + delta * delta * cnt * nt / tot;
        mean += delta * nt / tot;
        cnt = tot;
    }
    const double var = M2 / (double)L;
    const double sc = (double)gamma[n] / sqrt(var + (double)eps);
    scale[(size_t)b * C + n] = (float)sc;
    shift[(size_t)b * C + n] = (float)((double)beta[n] - mean * sc);
}

// last norm + leaky_relu and the "batch feature frames -> batch frames feature" rearrange (PyanNet.py:179)
__global__ __launch_bounds__(256) void sinc_out_kernel(const float *P, const float *scale, const float *shift, int B, int C, int L, float slope,
                                                       float *feats, int ldf) {
    const long long n = (long long)B * L * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const int t = (int)(bt % L), b = (int)(bt / L);
        float v = __builtin_fmaf(P[((size_t)b * C + c) * L + t], scale[(size_t)b * C + c], shift[(size_t)b * C + c]);
        v = v >= 0.f ? v : v * slope;
        feats[((size_t)b * L + t) * ldf + c] = v;
    }
}

}  // namespace

size_t sinc_conv_lds_bytes(const SincConvArgs &a, int NT) {
    const int NW = NT * 32;
    const int XW = (CT - 1) * a.stride + a.Kw;
    const size_t rows = (size_t)a.Cin + (a.Kp > a.Ktot);
    size_t xy = rows * XW + 8;
    if (xy < (size_t)NW * YS) xy = (size_t)NW * YS;
    return ((size_t)a.Kp * NW + xy) * sizeof(float);
}

hipError_t launch_wav_stats(const float *wav, int B, long long S, long long row_stride, const float *gamma, const float *beta, float eps,
                            float *scale, float *shift, hipStream_t s) {
    hipLaunchKernelGGL(wav_stats_kernel, dim3(B), dim3(256), 0, s, wav, S, row_stride, gamma, beta, eps, scale, shift);
    return hipGetLastError();
}

hipError_t launch_sinc_conv(const SincConvArgs &a, hipStream_t s) {
    const int NT = (a.Cout + 31) / 32;
    if (NT != 2 && NT != 3) return hipErrorInvalidValue;
    const size_t lds = sinc_conv_lds_bytes(a, NT);
    // enough workgroups to fill 256 CUs (one per CU: the filter matrix takes most of the LDS), each walking
    // several tiles of one utterance so the filter matrix is staged once per walk
    int gx = (1024 + a.B - 1) / a.B;
    if (gx > a.ntiles) gx = a.ntiles;
    if (gx < 1) gx = 1;
    const dim3 grid(gx, a.B), block(192);
    const bool cin1 = a.Cin == 1;
#define UVAD_SINC_LAUNCH(NT_, C1_)                                                                                         \
    {                                                                                                                      \
        auto k = conv_pool_kernel<NT_, C1_>;                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                                     \
    }
    if (NT == 3 && cin1) UVAD_SINC_LAUNCH(3, true)
    else if (NT == 3) UVAD_SINC_LAUNCH(3, false)
    else if (cin1) UVAD_SINC_LAUNCH(2, true)
    else UVAD_SINC_LAUNCH(2, false)
#undef UVAD_SINC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_norm_finalize(const float *partials, int B, int ntiles, int NW, int C, int L, const float *gamma, const float *beta, float eps,
                                float *scale, float *shift, hipStream_t s) {
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(B), dim3(128), 0, s, partials, ntiles, NW, C, L, gamma, beta, eps, scale, shift);
    return hipGetLastError();
}

hipError_t launch_sinc_out(const float *P, const float *scale, const float *shift, int B, int C, int L, float slope, float *feats, int ldf,
                           hipStream_t s) {
    const long long n = (long long)B * L * C;
    if (n <= 0) return hipSuccess;
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(sinc_out_kernel, dim3((int)g), dim3(256), 0, s, P, scale, shift, B, C, L, slope, feats, ldf);
    return hipGetLastError();
}

}  // namespace uvad

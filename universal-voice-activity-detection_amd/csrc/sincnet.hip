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
// x all output channels; the whole transposed filter matrix ([K][N]: one conflict-free ds_read_b32 per lane
// feeds the B operand) stays in LDS across the tiles a workgroup walks; the input window of the NEXT tile is
// fetched into registers while the matrix cores work on the current one and is written to LDS with the
// previous layer's normalisation + leaky_relu folded in; the epilogue does
// bias, |.|, the 3:1 max pool and the per-tile (sum, M2) statistics the instance norm of the next stage
// needs -- written as per-tile partials and reduced in tile order, so results do not depend on scheduling.
#include "uvad_internal.h"

namespace uvad {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Sum over each 32-lane half of the wave, returned in every lane of the half (DPP adds inside the 16-lane rows,
// row_bcast:15 into the odd rows, then the totals sit in lanes 31 and 63).  Inactive half-waves read as zero.
#define UVAD_SN_DPP_ADD(V, CTRL, ROW_MASK) \
    V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, ROW_MASK, 0xF, true))
__device__ __forceinline__ float half_sum(float v, bool upper) {
    UVAD_SN_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    UVAD_SN_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    UVAD_SN_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
    UVAD_SN_DPP_ADD(v, 0x140, 0xF);   // row_mirror: every lane holds its row total
    UVAD_SN_DPP_ADD(v, 0x142, 0xA);   // row_bcast:15 into rows 1 and 3
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return upper ? hi : lo;
}

constexpr int YS = 97;        // LDS row stride of the conv-output staging tile

__global__ __launch_bounds__(1024) void wav_stats_kernel(const float *wav, long long S, long long row_stride, const float *gamma,
                                                         const float *beta, float eps, float *scale, float *shift) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *x = wav + (size_t)b * row_stride;
    double s = 0.0, ss = 0.0;
    long long i0 = 0;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // 16-byte loads, 1024 threads: the whole row is in flight at once
        const long long n4 = S >> 2;
        for (long long i = tid; i < n4; i += 1024) {
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            ss += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        }
        i0 = n4 << 2;
    }
    for (long long i = i0 + tid; i < S; i += 1024) {
        const double v = x[i];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    __shared__ double red[2][16];
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        s = 0.0; ss = 0.0;
        for (int w = 0; w < 16; ++w) { s += red[0][w]; ss += red[1][w]; }
        const double mean = s / (double)S;
        double var = ss / (double)S - mean * mean;
        if (var < 0.0) var = 0.0;
        const double sc = (double)gamma[0] / sqrt(var + (double)eps);
        scale[b] = (float)sc;
        shift[b] = (float)((double)beta[0] - mean * sc);
    }
}

template <int NT, bool CIN1, int EPT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv_pool_kernel(SincConvArgs a) {
    constexpr int NW = NT * 32;
    constexpr int NTHR = WAVES * 64;
    constexpr int CTW = WAVES * 32;            // conv positions computed per tile (one 32-row MFMA block per wave)
    constexpr int TA = CTW / 3 * 3;            // positions the tile advances by (pool windows never straddle tiles)
    constexpr int PT = TA / 3;                 // pooled outputs per tile
    constexpr int PHASES = (WAVES + 2) / 3;    // the epilogue runs in groups of three waves = 96 positions = 32 pooled
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (scale, shift) of the current utterance's input norm: a separate LDS object, so the compiler knows the window
    // stores below cannot alias it (with one array every store waited for the previous (scale, shift) read)
    __shared__ float2 nrm[96];
    float *wt = smem;                                  // [Kp][NW]: W^T, zero padded
    float *xy = smem + (size_t)a.Kp * NW;              // input window [Cin][XW] (+ zero pad, + 1 dump slot) / conv-output tile [NW][YS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int XW = (CTW - 1) * a.stride + a.Kw;
    const int nelem = a.Cin * XW;                      // real elements of the window
    const int npad = 8;                                // the Kp - Ktot <= 7 zero-weight padded k read taps >= Kw: <= 7 elements past the window
    const int dump = nelem + npad;                     // where the staging stores of threads past the window go

    // Persistent workgroups (one per CU: the filter matrix takes most of the LDS): workgroup w owns the contiguous
    // range [g_begin, g_end) of the B*ntiles (utterance, tile) pairs, so the filter matrix is staged once per launch.
    const long long total = (long long)a.B * a.ntiles;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long g_begin = per * blockIdx.x;
    const long long g_end = g_begin + per < total ? g_begin + per : total;
    if (g_begin >= g_end) return;

    {   // filter matrix -> LDS (Kp*NW is a multiple of 128 floats), 8 independent 16-byte loads in flight per thread
        const float4 *src = reinterpret_cast<const float4 *>(a.Wt2);
        float4 *dst = reinterpret_cast<float4 *>(wt);
        const int n4 = a.Kp * NW / 4;
        for (int i0 = 0; i0 < n4; i0 += NTHR * 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = i0 + j * NTHR + tid; v[j] = src[i < n4 ? i : 0]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int i = i0 + j * NTHR + tid; if (i < n4) dst[i] = v[j]; }
        }
    }

    // Window element e of this thread is linear index idx = tid + NTHR*e of [Cin][XW] -> (ci, x), walked incrementally.
    const int ci0 = tid / XW, xs0 = tid - ci0 * XW;
    const int qstep = NTHR / XW, rstep = NTHR - qstep * XW;
    float pre[EPT];
#define UVAD_SN_PREFETCH(g_)                                                               \
    {   /* 32-bit element offsets from the utterance's (wave-uniform) base: one address VGPR per load */ \
        const int b_ = (int)((g_) / a.ntiles), tile_ = (int)((g_) - (long long)b_ * a.ntiles); \
        const float *inb_ = a.in + (size_t)b_ * a.in_bstride;                              \
        const int x0_ = tile_ * TA * a.stride;                                             \
        int x_ = xs0;                                                                      \
        unsigned off_ = (unsigned)ci0 * (unsigned)a.Lin + (unsigned)(x0_ + xs0);           \
        _Pragma("unroll") for (int e = 0; e < EPT; ++e) {                                  \
            const bool ok_ = tid + NTHR * e < nelem && x0_ + x_ < a.Lin;                   \
            pre[e] = inb_[ok_ ? off_ : 0u];                                                \
            x_ += rstep; off_ += (unsigned)qstep * (unsigned)a.Lin + (unsigned)rstep;      \
            if (x_ >= XW) { x_ -= XW; off_ += (unsigned)(a.Lin - XW); }                    \
        }                                                                                  \
    }
    UVAD_SN_PREFETCH(g_begin)

    int cur_b = -1;
    for (long long g = g_begin; g < g_end; ++g) {
        const int b = (int)(g / a.ntiles), tile = (int)(g - (long long)b * a.ntiles);
        __syncthreads();   // filter matrix staged / previous tile's pooled reads of xy are complete
        if (b != cur_b) {  // (scale, shift) of this utterance's input norm
            cur_b = b;
            if (tid < a.Cin) nrm[tid] = make_float2(a.in_scale[(size_t)b * a.Cin + tid], a.in_shift[(size_t)b * a.Cin + tid]);
            __syncthreads();
        }
        {   // registers -> LDS with the previous stage's instance norm (+ leaky_relu) folded in; branch-free: threads
            // past the window store into a dump slot
            const int x0 = tile * TA * a.stride;
            int ci = ci0, x = xs0;
#pragma unroll
            for (int e0 = 0; e0 < EPT; e0 += 8) {   // eight (scale, shift) reads in flight, then eight stores
                float2 ns[8];
                bool in[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    ns[j] = nrm[ci < a.Cin ? ci : 0];
                    in[j] = x0 + x < a.Lin;
                    ci += qstep; x += rstep;
                    if (x >= XW) { x -= XW; ++ci; }
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the eight reads together (the scheduler pairs them otherwise)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = tid + NTHR * (e0 + j);
                    float v = __builtin_fmaf(pre[e0 + j], ns[j].x, ns[j].y);
                    if (a.in_lrelu) v = v >= 0.f ? v : v * a.slope;
                    xy[idx < nelem ? idx : dump] = in[j] ? v : 0.f;
                }
            }
            for (int i = tid; i < npad; i += NTHR) xy[nelem + i] = 0.f;   // what the zero-weight padded K steps read
        }
        __syncthreads();
        // the next tile's window is fetched now and lands while the matrix cores work on this one
        if (g + 1 < g_end) UVAD_SN_PREFETCH(g + 1)

        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // K loop, software pipelined over four register sets: the operands of step ks+2 are read from LDS
        // before the MFMAs of step ks are issued (sched_group_barrier pins that order), i.e. every ds_read has
        // 2*NT MFMAs (256 / 384 cycles) to land.  Kp is a multiple of 8 => ksteps is a multiple of 4.
        // K order: single-channel stage k = tap; multi-channel stages k = tap*Cin + channel (channel-minor, Cin even), so
        // the two lane halves (k, k+1) are two channels at one tap and the walk through the window is wave-uniform:
        // soff is scalar arithmetic, no per-lane wrap logic between the MFMAs (that cost 25 % of the K loop).
        const float *bp = wt + (size_t)kk * NW + li;
        const int abase = (wave * 32 + li) * a.stride + (CIN1 ? kk : kk * XW);
        int soff = 0, cpair = 0;
        const int chalf = a.Cin >> 1;
        const int ksteps = a.Kp >> 1;
        float a0, a1, a2, a3, b0[NT], b1[NT], b2[NT], b3[NT];
#define UVAD_SN_LOAD(AV, BV, ks_)                                                          \
    {                                                                                      \
        AV = xy[abase + soff];                                                             \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) BV[t] = bp[(size_t)(ks_) * (NW * 2) + t * 32]; \
        if (CIN1) {                                                                        \
            soff += 2;                                                                     \
        } else {                                                                           \
            soff += 2 * XW;                                                                \
            if (++cpair == chalf) { cpair = 0; soff += 1 - a.Cin * XW; }                   \
        }                                                                                  \
    }
#define UVAD_SN_MFMA(AV, BV) \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV[t], acc[t], 0, 0, 0);
#define UVAD_SN_GROUPS()                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0); /* DS reads */ \
    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);     /* MFMAs    */
        UVAD_SN_LOAD(a0, b0, 0)
        UVAD_SN_LOAD(a1, b1, 1)
        int ks = 0;
        for (; ks + 4 < ksteps; ks += 4) {
            UVAD_SN_LOAD(a2, b2, ks + 2)
            UVAD_SN_MFMA(a0, b0)
            UVAD_SN_LOAD(a3, b3, ks + 3)
            UVAD_SN_MFMA(a1, b1)
            UVAD_SN_LOAD(a0, b0, ks + 4)
            UVAD_SN_MFMA(a2, b2)
            UVAD_SN_LOAD(a1, b1, ks + 5)
            UVAD_SN_MFMA(a3, b3)
            UVAD_SN_GROUPS() UVAD_SN_GROUPS() UVAD_SN_GROUPS() UVAD_SN_GROUPS()
        }
        UVAD_SN_LOAD(a2, b2, ks + 2)
        UVAD_SN_MFMA(a0, b0)
        UVAD_SN_LOAD(a3, b3, ks + 3)
        UVAD_SN_MFMA(a1, b1)
        UVAD_SN_MFMA(a2, b2)
        UVAD_SN_MFMA(a3, b3)
#undef UVAD_SN_GROUPS
#undef UVAD_SN_LOAD
#undef UVAD_SN_MFMA
        __syncthreads();   // all A reads of the input window are done: the region becomes the output tile

        // Epilogue in groups of three waves (96 positions = 32 pooled outputs, the size of the LDS output tile): bias,
        // |.|, accumulators -> LDS; 3:1 max pool; coalesced store; and the group's (sum, M2 about its own mean) per
        // channel -- combined in order by norm_finalize_kernel (Chan's update), which keeps the variance accurate when
        // it is small against the mean.  Each 32-lane half owns one channel per pass (DPP half-wave sums).
#pragma unroll
        for (int ph = 0; ph < PHASES; ++ph) {
            if (PHASES > 1 && ph > 0) __syncthreads();   // the previous group's pooled reads are complete
            if (wave / 3 == ph) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = t * 32 + li;
                    const float bias = a.bias[n];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                        float v = acc[t][r] + bias;
                        if (a.do_abs) v = __builtin_fabsf(v);
                        xy[(size_t)n * YS + (wave % 3) * 32 + row] = v;
                    }
                }
            }
            __syncthreads();
            const int cap = ph == PHASES - 1 ? PT - 32 * (PHASES - 1) : 32;
            const int p0 = tile * PT + 32 * ph;
            int nt = a.Lpool - p0 < cap ? a.Lpool - p0 : cap;
            if (nt < 0) nt = 0;
            const float inv_nt = nt > 0 ? 1.0f / (float)nt : 0.f;
            constexpr int PASSES = (NW + 2 * WAVES - 1) / (2 * WAVES);
#pragma unroll 4
            for (int j = 0; j < PASSES; ++j) {
                const int n = j * 2 * WAVES + wave * 2 + kk, p = li;
                const int nc = n < NW ? n : NW - 1;
                const float *y = xy + (size_t)nc * YS + 3 * p;
                const float m = __builtin_fmaxf(__builtin_fmaxf(y[0], y[1]), y[2]);
                const bool valid = n < a.Cout && p < nt;
                if (valid) a.out[((size_t)b * a.Cout + n) * a.Lpool + p0 + p] = m;
                const float s = half_sum(valid ? m : 0.f, kk != 0);
                const float d = valid ? m - s * inv_nt : 0.f;
                const float m2 = half_sum(d * d, kk != 0);
                if (p == 0 && n < NW) {
                    float *pp = a.partials + (((size_t)b * a.ntiles + tile) * PHASES + ph) * (NW * 2) + n * 2;
                    pp[0] = s;
                    pp[1] = m2;
                }
            }
        }
    }
}

// per-tile (sum, M2) partials, combined in tile order -> per (b, c) affine of the instance norm:
//   y = (x - mean) / sqrt(var + eps) * gamma + beta = x * scale + shift      (biased variance, torch InstanceNorm1d)
__global__ __launch_bounds__(128) void norm_finalize_kernel(const float *partials, int ntiles, int pt, int phases, int NW, int C, int L,
                                                            const float *gamma, const float *beta, float eps, float *scale, float *shift) {
    const int b = blockIdx.x, n = threadIdx.x;
    if (n >= C) return;
    double mean = 0.0, M2 = 0.0, cnt = 0.0;
    for (int t = 0; t < ntiles; ++t)
        for (int ph = 0; ph < phases; ++ph) {   // the statistics groups of conv_pool_kernel, in position order
            const int start = t * pt + 32 * ph, cap = ph == phases - 1 ? pt - 32 * (phases - 1) : 32;
            const int ni = L - start < cap ? L - start : cap;
            if (ni <= 0) continue;
            const float *pp = partials + (((size_t)b * ntiles + t) * phases + ph) * (NW * 2) + n * 2;
            const double nt = (double)ni;
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

static int sinc_window(const SincConvArgs &a, int waves) { return (waves * 32 - 1) * a.stride + a.Kw; }

size_t sinc_conv_lds_bytes(const SincConvArgs &a, int NT, int waves) {
    const int NW = NT * 32;
    size_t xy = (size_t)a.Cin * sinc_window(a, waves) + 8 + 1;
    if (xy < (size_t)NW * YS) xy = (size_t)NW * YS;
    return ((size_t)a.Kp * NW + xy) * sizeof(float) + 96 * sizeof(float2);   // + the static (scale, shift) table
}

// elements of the input window each thread carries in registers between tiles
int sinc_conv_ept(const SincConvArgs &a, int waves) { return (a.Cin * sinc_window(a, waves) + waves * 64 - 1) / (waves * 64); }

// Workgroup shape of a stage: 8 waves (two per SIMD, 255 positions = 85 pooled outputs per tile) when the filter
// matrix + window fit the LDS and the window fits the register staging -- the single-channel sinc stage --
// 4 waves for the multi-channel stages under the same conditions, else 3 waves (96 positions = 32 pooled outputs).
SincConvPlan sinc_conv_plan(const SincConvArgs &a) {
    const int NT = (a.Cout + 31) / 32;
    SincConvPlan p;
    const bool wide = a.Cin == 1 && sinc_conv_lds_bytes(a, NT, 8) <= (size_t)160 * 1024 && sinc_conv_ept(a, 8) <= 8;
    // multi-channel stages: 4 waves (one per SIMD, 126 positions = 42 pooled outputs) when the wider window still fits
    const bool four = a.Cin > 1 && sinc_conv_lds_bytes(a, NT, 4) <= (size_t)160 * 1024 && sinc_conv_ept(a, 4) <= 48;
    p.waves = wide ? 8 : four ? 4 : 3;
    p.pt = p.waves * 32 / 3;
    p.phases = (p.waves + 2) / 3;
    return p;
}

hipError_t launch_wav_stats(const float *wav, int B, long long S, long long row_stride, const float *gamma, const float *beta, float eps,
                            float *scale, float *shift, hipStream_t s) {
    hipLaunchKernelGGL(wav_stats_kernel, dim3(B), dim3(1024), 0, s, wav, S, row_stride, gamma, beta, eps, scale, shift);
    return hipGetLastError();
}

hipError_t launch_sinc_conv(const SincConvArgs &a, hipStream_t s) {
    const int NT = (a.Cout + 31) / 32;
    if (NT != 2 && NT != 3) return hipErrorInvalidValue;
    const SincConvPlan plan = sinc_conv_plan(a);
    if (a.ntiles != (a.Lpool + plan.pt - 1) / plan.pt) return hipErrorInvalidValue;   // the caller sized the partials with the same plan
    const size_t lds = sinc_conv_lds_bytes(a, NT, plan.waves);
    // persistent workgroups, one per CU (the LDS-resident filter matrix allows no more), each walking a contiguous
    // range of the (utterance, tile) pairs
    const long long total = (long long)a.B * a.ntiles;
    const int ncu = a.n_cu > 0 ? a.n_cu : 256;
    const dim3 grid((unsigned)(total < ncu ? total : ncu)), block(plan.waves * 64);
    const bool cin1 = a.Cin == 1;
    const int ept = sinc_conv_ept(a, plan.waves);
    if (ept > 48 || (cin1 && ept > 8)) return hipErrorInvalidValue;   // uvad_sincnet_configure rejects these geometries
#define UVAD_SINC_LAUNCH(NT_, C1_, EPT_, W_)                                                                               \
    {                                                                                                                      \
        auto k = conv_pool_kernel<NT_, C1_, EPT_, W_>;                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                     \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                                     \
    }
    if (cin1 && plan.waves == 8) {
        if (NT == 3) UVAD_SINC_LAUNCH(3, true, 8, 8) else UVAD_SINC_LAUNCH(2, true, 8, 8)
    } else if (cin1) {
        if (NT == 3) UVAD_SINC_LAUNCH(3, true, 8, 3) else UVAD_SINC_LAUNCH(2, true, 8, 3)
    } else if (plan.waves == 4) {
        if (ept <= 32) { if (NT == 3) UVAD_SINC_LAUNCH(3, false, 32, 4) else UVAD_SINC_LAUNCH(2, false, 32, 4) }
        else { if (NT == 3) UVAD_SINC_LAUNCH(3, false, 48, 4) else UVAD_SINC_LAUNCH(2, false, 48, 4) }
    } else if (ept <= 32) {
        if (NT == 3) UVAD_SINC_LAUNCH(3, false, 32, 3) else UVAD_SINC_LAUNCH(2, false, 32, 3)
    } else {
        if (NT == 3) UVAD_SINC_LAUNCH(3, false, 48, 3) else UVAD_SINC_LAUNCH(2, false, 48, 3)
    }
#undef UVAD_SINC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_norm_finalize(const float *partials, int B, int ntiles, int pt, int phases, int NW, int C, int L, const float *gamma,
                                const float *beta, float eps, float *scale, float *shift, hipStream_t s) {
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(B), dim3(128), 0, s, partials, ntiles, pt, phases, NW, C, L, gamma, beta, eps, scale, shift);
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

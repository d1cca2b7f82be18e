// gemm.hip -- exact-f32 MFMA GEMM for the time-parallel contractions of the classifier:
//   * LSTM input projections  x_t * W_ih^T + (b_ih + b_hh)   (reference: nn.LSTM inside
//     PyanNet2.forward, src/models/segmentation/PyanNet2.py:169-172)
//   * feed-forward layers     leaky_relu(x * W^T + b)         (PyanNet2.py:183-185)
//
// gfx950 design: v_mfma_f32_32x32x2_f32 (bit-exact k-ordered fmaf chain, 64 FLOP/clk/SIMD).
// 128x128 output tile per 256-thread workgroup, 2x2 waves, each wave a 64x64 sub-tile held as
// 2x2 accumulators of 32x32; K is walked in steps of 32 staged through LDS with register
// prefetch of the next K-step.  LDS rows are padded to 36 floats so the ds_read_b128 fragment
// reads (16-lane groups, 64-bank rule) are conflict-free.  Block ids are remapped so that the
// N-tiles that share one A row-panel run on the same XCD (private L2) back to back.
#include "uvad_internal.h"

namespace uvad {

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Row m of A.  Rows past M (or padded sequences, b >= B) are clamped to row 0: every load in the
// kernel is unconditional (no exec-masked branches in the K loop); what such rows produce is
// either never stored (row >= M) or belongs to a padded sequence nobody reads.
__device__ __forceinline__ const float *a_row_ptr(const GemmArgs &a, int m, int r0, int rend) {
    if (m >= rend) return a.A + (a.a_mode == 1 ? (size_t)0 : (size_t)r0 * a.lda);
    if (a.a_mode != 1) return a.A + (size_t)m * a.lda;
    const int per_tile = a.T * SEQ_TILE;
    const int tile = m / per_tile, rem = m - tile * per_tile;
    const int t = rem / SEQ_TILE, j = rem - t * SEQ_TILE;
    const int b = tile * SEQ_TILE + j;
    if (b >= a.B) return a.A;
    return a.A + ((size_t)b * a.T + t) * a.lda;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a, int mt, int nt) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch, speed only).
    if (a.gate && (*a.gate != 0) != (a.gate_run_if_set != 0)) return;   // device-side kernel selection (see GemmArgs)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int m_tile = (idx / nt) * 8 + xcd, n_tile = idx % nt;
    if (m_tile >= mt) return;
    const int R0 = m_tile * BM, Rend = a.M;
    const int C0 = n_tile * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging assignment: 8 float4 per 32-float row, 32 rows per pass, 4 passes per operand
    const int srow = tid >> 3, skq = tid & 7;
    const float *ap0 = a_row_ptr(a, R0 + srow, R0, Rend) + skq * 4;
    const float *ap1 = a_row_ptr(a, R0 + srow + 32, R0, Rend) + skq * 4;
    const float *ap2 = a_row_ptr(a, R0 + srow + 64, R0, Rend) + skq * 4;
    const float *ap3 = a_row_ptr(a, R0 + srow + 96, R0, Rend) + skq * 4;
    // W rows are zero-padded to ldw >= nk*BK; rows past N are clamped to row 0 (never stored)
    const int n0 = C0 + srow;
    const float *bp0 = a.W + (size_t)(n0 < a.N ? n0 : 0) * a.ldw + skq * 4;
    const float *bp1 = a.W + (size_t)(n0 + 32 < a.N ? n0 + 32 : 0) * a.ldw + skq * 4;
    const float *bp2 = a.W + (size_t)(n0 + 64 < a.N ? n0 + 64 : 0) * a.ldw + skq * 4;
    const float *bp3 = a.W + (size_t)(n0 + 96 < a.N ? n0 + 96 : 0) * a.ldw + skq * 4;

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    // A columns past K (only in the last K-step when K % 32 != 0): the load re-reads column block 0 (unconditional,
    // in bounds) and the registers are zeroed (a non-finite value there must not meet the zero-padded weights).
#define UVAD_GLOAD(k0)                                                        \
    {                                                                         \
        const bool kin_ = (k0) + skq * 4 < a.K;                               \
        const int ka_ = kin_ ? (k0) : -skq * 4;                               \
        const float4 z4_ = make_float4(0.f, 0.f, 0.f, 0.f);                   \
        ra0 = *reinterpret_cast<const float4 *>(ap0 + ka_);                   \
        ra1 = *reinterpret_cast<const float4 *>(ap1 + ka_);                   \
        ra2 = *reinterpret_cast<const float4 *>(ap2 + ka_);                   \
        ra3 = *reinterpret_cast<const float4 *>(ap3 + ka_);                   \
        if (!kin_) { ra0 = z4_; ra1 = z4_; ra2 = z4_; ra3 = z4_; }            \
        rb0 = *reinterpret_cast<const float4 *>(bp0 + (k0));                  \
        rb1 = *reinterpret_cast<const float4 *>(bp1 + (k0));                  \
        rb2 = *reinterpret_cast<const float4 *>(bp2 + (k0));                  \
        rb3 = *reinterpret_cast<const float4 *>(bp3 + (k0));                  \
    }
    float *as_w = &As[srow * LDS_LD + skq * 4], *bs_w = &Bs[srow * LDS_LD + skq * 4];
#define UVAD_LSTORE()                                                         \
    {                                                                         \
        *reinterpret_cast<float4 *>(as_w) = ra0;                              \
        *reinterpret_cast<float4 *>(as_w + 32 * LDS_LD) = ra1;                \
        *reinterpret_cast<float4 *>(as_w + 64 * LDS_LD) = ra2;                \
        *reinterpret_cast<float4 *>(as_w + 96 * LDS_LD) = ra3;                \
        *reinterpret_cast<float4 *>(bs_w) = rb0;                              \
        *reinterpret_cast<float4 *>(bs_w + 32 * LDS_LD) = rb1;                \
        *reinterpret_cast<float4 *>(bs_w + 64 * LDS_LD) = rb2;                \
        *reinterpret_cast<float4 *>(bs_w + 96 * LDS_LD) = rb3;                \
    }

    const int nk = (a.K + BK - 1) / BK;
    UVAD_GLOAD(0)
    UVAD_LSTORE()
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const float *as_r = &As[(wr * 64 + fr) * LDS_LD + fh * 4], *bs_r = &Bs[(wc * 64 + fr) * LDS_LD + fh * 4];
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) UVAD_GLOAD((kt + 1) * BK)
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            const float4 a0 = *reinterpret_cast<const float4 *>(as_r + kc * 8);
            const float4 a1 = *reinterpret_cast<const float4 *>(as_r + 32 * LDS_LD + kc * 8);
            const float4 b0 = *reinterpret_cast<const float4 *>(bs_r + kc * 8);
            const float4 b1 = *reinterpret_cast<const float4 *>(bs_r + 32 * LDS_LD + kc * 8);
            // the four accumulators are visited round-robin so consecutive MFMAs are independent
#define UVAD_MFMA_E(E)                                                               \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b0.E, acc00, 0, 0, 0);        \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b1.E, acc01, 0, 0, 0);        \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b0.E, acc10, 0, 0, 0);        \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b1.E, acc11, 0, 0, 0);
            UVAD_MFMA_E(x)
            UVAD_MFMA_E(y)
            UVAD_MFMA_E(z)
            UVAD_MFMA_E(w)
        }
        __syncthreads();
        if (kt + 1 < nk) {
            UVAD_LSTORE()
            __syncthreads();
        }
    }

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool full = R0 + BM <= Rend && C0 + BN <= a.N;   // block-uniform: interior tiles store unguarded
#define UVAD_EPILOGUE(ACC, I, J)                                                                     \
    {                                                                                                \
        const int col = C0 + wc * 64 + (J) * 32 + fr;                                                \
        const int rbase = R0 + wr * 64 + (I) * 32 + 4 * fh;                                          \
        const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;                                \
        /* c_blocked: tile-blocked gate matrix (g_index); the 16 rows of a lane stay inside one 128-row tile */ \
        float *crow = a.c_blocked ? a.C + g_index(rbase, col < a.N ? col : 0, a.N) : a.C + (size_t)rbase * a.ldc + col; \
        const size_t cstride = a.c_blocked ? 64 : (size_t)a.ldc;                                     \
        float v[16];                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
            v[r] = ACC[r] + bias;                                                                    \
            if (a.act == 1) v[r] = v[r] >= 0.f ? v[r] : a.leaky_slope * v[r];                        \
        }                                                                                            \
        if (full) {                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                           \
                crow[(size_t)((r & 3) + 8 * (r >> 2)) * cstride] = v[r];                             \
        } else {                                                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                const int dr = (r & 3) + 8 * (r >> 2);                                               \
                if (rbase + dr < Rend && col < a.N) crow[(size_t)dr * cstride] = v[r];                \
            }                                                                                        \
        }                                                                                            \
    }
    UVAD_EPILOGUE(acc00, 0, 0)
    UVAD_EPILOGUE(acc01, 0, 1)
    UVAD_EPILOGUE(acc10, 1, 0)
    UVAD_EPILOGUE(acc11, 1, 1)
}

}  // namespace

int gemm_padded_k(int K) { return (K + BK - 1) / BK * BK; }

hipError_t launch_gemm(const GemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (a.ldw < gemm_padded_k(a.K)) return hipErrorInvalidValue;
    const int mt = (a.M + BM - 1) / BM;
    const int nt = (a.N + BN - 1) / BN;
    if (mt <= 0) return hipSuccess;
    const int grid = ((mt + 7) / 8) * 8 * nt;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(grid), dim3(256), 0, s, a, mt, nt);
    return hipGetLastError();
}

}  // namespace uvad

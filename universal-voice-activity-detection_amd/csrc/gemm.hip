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

__device__ __forceinline__ const float *a_row_ptr(const GemmArgs &a, int m) {
    if (m >= a.M) return nullptr;
    if (a.a_mode == 0) return a.A + (size_t)m * a.lda;
    const int per_tile = a.T * SEQ_TILE;
    const int tile = m / per_tile, rem = m - tile * per_tile;
    const int t = rem / SEQ_TILE, j = rem - t * SEQ_TILE;
    const int b = tile * SEQ_TILE + j;
    if (b >= a.B) return nullptr;
    return a.A + ((size_t)b * a.T + t) * a.lda;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a, int mt, int nt) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch, speed only).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int m_tile = (idx / nt) * 8 + xcd, n_tile = idx % nt;
    if (m_tile >= mt) return;
    const int R0 = m_tile * BM, C0 = n_tile * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // staging assignment: 8 float4 per 32-float row, 32 rows per pass, 4 passes per operand
    const int srow = tid >> 3, skq = tid & 7;
    const float *ap[4];
    const float *bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ap[i] = a_row_ptr(a, R0 + srow + 32 * i);
        const int n = C0 + srow + 32 * i;
        bp[i] = n < a.N ? a.W + (size_t)n * a.K : nullptr;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
        const int k = k0 + skq * 4;
        const bool kin = k < a.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (kin && ap[i]) ? *reinterpret_cast<const float4 *>(ap[i] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (kin && bp[i]) ? *reinterpret_cast<const float4 *>(bp[i] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(&As[(srow + 32 * i) * LDS_LD + skq * 4]) = ra[i];
            *reinterpret_cast<float4 *>(&Bs[(srow + 32 * i) * LDS_LD + skq * 4]) = rb[i];
        }
    };

    const int nk = (a.K + BK - 1) / BK;
    gload(0);
    lstore();
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = *reinterpret_cast<const float4 *>(&As[(wr * 64 + i * 32 + fr) * LDS_LD + kc * 8 + fh * 4]);
                bv[i] = *reinterpret_cast<const float4 *>(&Bs[(wc * 64 + i * 32 + fr) * LDS_LD + kc * 8 + fh * 4]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            lstore();
            __syncthreads();
        }
    }

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = C0 + wc * 64 + j * 32 + fr;
        if (col >= a.N) continue;
        const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = R0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < a.M) {
                    float v = acc[i][j][r] + bias;
                    if (a.act == 1) v = v >= 0.f ? v : a.leaky_slope * v;
                    a.C[(size_t)row * a.ldc + col] = v;
                }
            }
        }
    }
}

}  // namespace

hipError_t launch_gemm(const GemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    const int grid = ((mt + 7) / 8) * 8 * nt;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(grid), dim3(256), 0, s, a, mt, nt);
    return hipGetLastError();
}

}  // namespace uvad

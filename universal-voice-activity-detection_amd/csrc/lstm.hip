// lstm.hip -- the sequential half of nn.LSTM (reference: the time loop inside
// `self.lstm(outputs)`, src/models/segmentation/PyanNet2.py:169-172; gate order i,f,g,o,
// zero initial state, eval mode).  The time-parallel half (x_t*W_ih^T + biases) is gemm.hip.
//
// gfx950 design (persistent-RNN): one 256-thread workgroup owns SEQ_TILE = 4 sequences of one
// direction for ALL T steps, so no workgroup ever talks to another.  W_hh (4H x H f32 = 256 KiB
// at H = 128) does not fit the 160 KiB LDS, but it fits the CU's 512 KiB register file: each of
// the 4 waves (one per SIMD, 512-register budget) keeps the 128 rows of its 32 hidden units as
// 256 resident VGPR/AGPRs per lane, laid out as the A operand of v_mfma_f32_4x4x1_16B_f32:
//      block b (16 per instruction) = hidden unit, A rows = its 4 gates (i,f,g,o), K = 1,
//      B = h_{t-1}[k] for the 4 sequences (identical in every block), D[gate][seq].
// A lane therefore ends the 128-deep chain holding all four gate pre-activations of ONE
// (unit, sequence) pair: the cell update is lane-local, no shuffles.  h_t is exchanged
// between the waves through a double-buffered 2 KiB LDS tile (one barrier per step) and the
// next steps' gate pre-activations are prefetched from HBM PD steps ahead.
// Exact f32: the MFMA is a k-ordered fmaf chain (bit-exact f32).
#include "uvad_internal.h"

namespace uvad {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int PD = 4;  // gate prefetch depth (steps)

__device__ __forceinline__ float sigmoid_f(float x) {
    // 1/(1+2^(-x*log2e)); v_exp_f32 + v_rcp_f32 (1 ulp each).  |x| <= 16 => abs err < 3e-7.
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return __builtin_fmaf(2.0f, sigmoid_f(2.0f * x), -1.0f);
}

template <int H>
__global__ __launch_bounds__(256) void lstm_rec_kernel(LstmArgs a) {
    constexpr int UW = H / 4;   // hidden units per wave
    constexpr int RB = UW / 16; // 16-unit MFMA row blocks per wave
    constexpr int HS = H + 4;   // LDS row stride (floats): the 4 sequence rows land on disjoint banks
    static_assert(RB >= 1 && UW % 16 == 0, "H must be a multiple of 64");

    __shared__ __attribute__((aligned(16))) float hbuf[2][SEQ_TILE][HS];

    const int tile = blockIdx.x, dir = blockIdx.y;
    const bool reverse = dir == 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jb = lane & 3;    // sequence within the tile (B / D operand column), gate index of the A operand
    const int blk = lane >> 2;  // MFMA block = hidden unit within the row block

    // ---- resident recurrent weights -------------------------------------------------------
    float w[RB][H];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(a.Whh_packed + (size_t)dir * 4 * H * H);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int kq = 0; kq < H / 4; ++kq) {
                const float4 v = wp[(size_t)((wave * RB + rb) * (H / 4) + kq) * 64 + lane];
                w[rb][4 * kq + 0] = v.x; w[rb][4 * kq + 1] = v.y;
                w[rb][4 * kq + 2] = v.z; w[rb][4 * kq + 3] = v.w;
            }
    }

    // ---- state ----------------------------------------------------------------------------
    const int seq = tile * SEQ_TILE + jb;                 // padded batch index
    const int nseq = a.tiles * SEQ_TILE;
    float c[RB];
    int unit[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        unit[rb] = wave * UW + rb * 16 + blk;
        const size_t so = ((size_t)dir * nseq + seq) * H + unit[rb];
        c[rb] = a.c0 ? a.c0[so] : 0.0f;
        hbuf[0][jb][unit[rb]] = a.h0 ? a.h0[so] : 0.0f;
    }
    __syncthreads();

    // row of (t, jb) in the tile-major activation matrices
    const size_t row0 = (size_t)tile * a.T * SEQ_TILE + jb;
    const float *gbase = a.G + (size_t)dir * 4 * H;
    float *ybase = a.Y + (size_t)dir * H;

    f32x4 gq[PD][RB];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int t = reverse ? a.T - 1 - p : p;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (p < a.T)
                gq[p][rb] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)t * SEQ_TILE) * a.ldg + unit[rb] * 4);
            else
                gq[p][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    float hlast[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) hlast[rb] = 0.0f;

    for (int s = 0; s < a.T; ++s) {
        const int t = reverse ? a.T - 1 - s : s;
        f32x4 acc[RB][2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            acc[rb][0] = gq[0][rb];
            acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // rotate the prefetch ring and issue the load for step s + PD
#pragma unroll
        for (int p = 0; p + 1 < PD; ++p)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) gq[p][rb] = gq[p + 1][rb];
        {
            const int sp = s + PD;
            if (sp < a.T) {
                const int tp = reverse ? a.T - 1 - sp : sp;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    gq[PD - 1][rb] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)tp * SEQ_TILE) * a.ldg + unit[rb] * 4);
            }
        }

        // ---- gates += W_hh * h_{t-1}: 2*H MFMAs (4x4x1, 16 blocks) per wave -----------------
        const float *hb = &hbuf[s & 1][jb][0];
#pragma unroll
        for (int kq = 0; kq < H / 4; ++kq) {
            const float4 hv = *reinterpret_cast<const float4 *>(hb + 4 * kq);
            const int half = kq >= H / 8 ? 1 : 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb][half] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 0], hv.x, acc[rb][half], 0, 0, 0);
                acc[rb][half] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 1], hv.y, acc[rb][half], 0, 0, 0);
                acc[rb][half] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 2], hv.z, acc[rb][half], 0, 0, 0);
                acc[rb][half] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 3], hv.w, acc[rb][half], 0, 0, 0);
            }
        }

        // ---- lane-local cell update ---------------------------------------------------------
        float *hn = &hbuf[(s + 1) & 1][jb][0];
        float *yrow = ybase + (row0 + (size_t)t * SEQ_TILE) * a.ldy;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const f32x4 g = acc[rb][0] + acc[rb][1];
            const float ig = sigmoid_f(g[0]);
            const float fg = sigmoid_f(g[1]);
            const float gg = tanh_f(g[2]);
            const float og = sigmoid_f(g[3]);
            c[rb] = __builtin_fmaf(fg, c[rb], ig * gg);
            const float h = og * tanh_f(c[rb]);
            hlast[rb] = h;
            hn[unit[rb]] = h;
            yrow[unit[rb]] = h;
        }
        __syncthreads();
    }

    if (a.hN) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const size_t so = ((size_t)dir * nseq + seq) * H + unit[rb];
            a.hN[so] = hlast[rb];
            a.cN[so] = c[rb];
        }
    }
}

}  // namespace

size_t whh_packed_elems(int H) { return (size_t)4 * H * H; }

void pack_whh(const float *w_hh, int H, float *out) {
    const int UW = H / 4, RB = UW / 16;
    for (int wave = 0; wave < 4; ++wave)
        for (int rb = 0; rb < RB; ++rb)
            for (int kq = 0; kq < H / 4; ++kq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int gate = lane & 3, unit = wave * UW + rb * 16 + (lane >> 2);
                        out[((size_t)((wave * RB + rb) * (H / 4) + kq) * 64 + lane) * 4 + e] =
                            w_hh[(size_t)(gate * H + unit) * H + 4 * kq + e];
                    }
}

hipError_t launch_lstm(const LstmArgs &a, hipStream_t s) {
    if (a.tiles <= 0 || a.T <= 0) return hipSuccess;
    dim3 grid(a.tiles, a.dirs);
    if (a.H == 128)
        hipLaunchKernelGGL(lstm_rec_kernel<128>, grid, dim3(256), 0, s, a);
    else if (a.H == 64)
        hipLaunchKernelGGL(lstm_rec_kernel<64>, grid, dim3(256), 0, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace uvad

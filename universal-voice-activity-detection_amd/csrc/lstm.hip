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

#include <cstdlib>
#include <cstring>

namespace uvad {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifndef UVAD_LSTM_PD
#define UVAD_LSTM_PD 4   // 8 fits (252 of 256 registers) but measured the same on one box: 6.11 vs 6.11 ms per step, 1.36 vs 1.36 ms per launch
#endif
constexpr int PD = 4;  // gate prefetch depth (steps) of the variant kernels; the main kernel uses UVAD_LSTM_PD (below)

__device__ __forceinline__ float sigmoid_f(float x) {
    // 1/(1+2^(-x*log2e)); v_exp_f32 + v_rcp_f32 (1 ulp each).  |x| <= 16 => abs err < 3e-7.
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return __builtin_fmaf(2.0f, sigmoid_f(2.0f * x), -1.0f);
}

// The cell update of one (unit, sequence) pair, cut into short stages so that the stages of row
// block rb can be placed, in program order, between the MFMA groups of row block rb+1 (the
// matrix pipe and the VALU issue from the same in-order stream; a stage fits the 32-cycle shadow
// of a group of four 2-pass MFMAs).
struct GateState {
    f32x4 a0, a1, a2, a3, g;
    float e0, e1, e2, e3, t, h;
};
constexpr float NL2E = -1.4426950408889634f;
constexpr int GATE_STAGES = 10;
template <int ST>
__device__ __forceinline__ void gate_stage(GateState &S, float &c) {
#ifdef UVAD_ABL_NOGATE   // diagnostic build (tools/lstm_ablate.hip): cell update reduced to one add
    if constexpr (ST == 0) S.g = (S.a0 + S.a1) + (S.a2 + S.a3);
    if constexpr (ST == 9) { S.h = S.g[0] * 1e-3f + S.g[1] * 1e-3f; c = S.g[2] + S.g[3]; }
    return;
#endif
    if constexpr (ST == 0) {
        S.g = (S.a0 + S.a1) + (S.a2 + S.a3);
    } else if constexpr (ST == 1) {
        S.e0 = NL2E * S.g[0]; S.e1 = NL2E * S.g[1]; S.e2 = (2.0f * NL2E) * S.g[2]; S.e3 = NL2E * S.g[3];
    } else if constexpr (ST == 2) {
        S.e0 = __builtin_amdgcn_exp2f(S.e0); S.e1 = __builtin_amdgcn_exp2f(S.e1);
        S.e2 = __builtin_amdgcn_exp2f(S.e2); S.e3 = __builtin_amdgcn_exp2f(S.e3);
    } else if constexpr (ST == 3) {
        S.e0 += 1.0f; S.e1 += 1.0f; S.e2 += 1.0f; S.e3 += 1.0f;
    } else if constexpr (ST == 4) {
        S.e0 = __builtin_amdgcn_rcpf(S.e0); S.e1 = __builtin_amdgcn_rcpf(S.e1);     // i, f
        S.e2 = __builtin_amdgcn_rcpf(S.e2); S.e3 = __builtin_amdgcn_rcpf(S.e3);     // sigma(2g), o
    } else if constexpr (ST == 5) {
        const float gg = __builtin_fmaf(2.0f, S.e2, -1.0f);                          // tanh(g)
        c = __builtin_fmaf(S.e1, c, S.e0 * gg);
        S.t = (2.0f * NL2E) * c;
    } else if constexpr (ST == 6) {
        S.t = __builtin_amdgcn_exp2f(S.t);
    } else if constexpr (ST == 7) {
        S.t += 1.0f;
    } else if constexpr (ST == 8) {
        S.t = __builtin_amdgcn_rcpf(S.t);
    } else if constexpr (ST == 9) {
        S.h = S.e3 * __builtin_fmaf(2.0f, S.t, -1.0f);                               // o * tanh(c)
    }
}
template <int ST>
__device__ __forceinline__ void gate_stages_upto(GateState &S, float &c) {   // stages [0, ST]
    if constexpr (ST > 0) gate_stages_upto<ST - 1>(S, c);
    gate_stage<ST>(S, c);
}

// Gate prefetch: plain loads into a ring of PD register slots (the time loop is unrolled by PD so
// every slot is a fixed register).  An inline-asm load with hand-counted vmcnt was tried and
// measured no faster, and it is fragile (hipcc may reuse an asm load's destination before the data
// lands), so the compiler's own waitcnt bookkeeping is kept.
__device__ __forceinline__ void gq_load(f32x4 &dst, const float *p) {
#ifdef UVAD_ABL_NOGMEM   // diagnostic: no global traffic inside the time loop
    dst = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
    (void)p;
#else
    dst = *reinterpret_cast<const f32x4 *>(p);
#endif
}
#ifdef UVAD_ABL_NOGMEM
#define UVAD_YSTORE(ptr, v) asm volatile("" ::"v"(ptr), "v"(v))
#else
#define UVAD_YSTORE(ptr, v) (*(ptr) = (v))
#endif

template <int H, int WAVES, bool HAS_G2>
__global__ __launch_bounds__(WAVES * 64) void lstm_rec_kernel(LstmArgs a) {
    constexpr int UW = H / WAVES;   // hidden units per wave
    constexpr int RB = UW / 16;     // 16-unit MFMA row blocks per wave
    constexpr int HS = H + 4;   // LDS row stride (floats): the 4 sequence rows land on disjoint banks
    constexpr int PD = HAS_G2 ? 4 : UVAD_LSTM_PD;   // gate prefetch depth (steps)
    static_assert(RB >= 1 && UW % 16 == 0 && UW * WAVES == H, "H must split into 16-unit blocks over the waves");

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

    // steps [s_begin, s_end) of the recurrence (a chunk when the host overlaps the next layer's projection)
    const int s_begin = a.s_begin, s_end = a.s_count > 0 ? a.s_begin + a.s_count : a.T;
    const float *gbase2 = HAS_G2 ? a.G2 + (size_t)dir * 4 * H : nullptr;
    f32x4 gq[PD][RB], gq2[HAS_G2 ? PD : 1][RB];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int sp = s_begin + p < s_end ? s_begin + p : s_end - 1;
        const int t = reverse ? a.T - 1 - sp : sp;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            gq_load(gq[p][rb], gbase + (row0 + (size_t)t * SEQ_TILE) * a.ldg + unit[rb] * 4);
            if constexpr (HAS_G2) gq_load(gq2[p][rb], gbase2 + (row0 + (size_t)t * SEQ_TILE) * a.ldg + unit[rb] * 4);
        }
    }

    float hlast[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) hlast[rb] = 0.0f;
#ifdef UVAD_STAMP   // diagnostic build (tools/lstm_ablate.hip): per-wave cycle shares of a step
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
#define UVAD_STAMP_AT(i)                                                                          \
    {                                                                                             \
        unsigned long long t_;                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (st_prev) st_acc[i] += t_ - st_prev;                                                   \
        st_prev = t_;                                                                             \
    }
#else
#define UVAD_STAMP_AT(i)
#endif

    // Time loop unrolled by the prefetch depth: step s uses ring slot s % PD and refills it with the
    // gates of step s + PD, so every load lands in the register it is consumed from PD steps later
    // (no register rotation => the compiler can wait with a counted vmcnt instead of vmcnt(0), and
    // the h stores of the last steps stay in flight).
    for (int s0 = s_begin; s0 < s_end; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s >= s_end) break;   // wave-uniform
        const int t = reverse ? a.T - 1 - s : s;
        // Pin the resident weights in the accumulator half of the unified register file: MFMA reads
        // A operands straight from AGPRs, and VALU-addressable VGPRs stay free for h / gates.
        // (Zero instructions: the constraint only tells the allocator where the values live here.)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int k = 0; k < H; ++k) asm volatile("" : "+a"(w[rb][k]));

        // h_{t-1} of this lane's sequence, all H values (broadcast reads: 4 distinct addresses per wave)
        const float *hb = &hbuf[(s - s_begin) & 1][jb][0];
        // RB > 1: every row block re-uses the values, keep them all; RB == 1: stream them (the register
        // budget of a two-waves-per-SIMD workgroup is 256: 128 for W_hh, the rest for everything else)
        // With RB == 1 they are streamed through a ring of HR slots, refilled right after use, so that
        // HR LDS reads stay in flight ahead of the MFMA groups (LDS latency under 8 reading waves is
        // several MFMA groups long; a read issued one group ahead stalls the matrix pipe).
        constexpr int HR = RB > 1 ? H / 4 : 16;
        float4 hv[HR];
#ifdef UVAD_ABL_NOLDSREAD   // diagnostic: no h reads at all (results meaningless)
#pragma unroll
        for (int kq = 0; kq < HR; ++kq) hv[kq] = make_float4(c[0], hlast[0], c[0] * 0.5f, hlast[0] * 0.5f);
#else
#pragma unroll
        for (int kq = 0; kq < HR; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(hb + 4 * kq);
#endif

        float *hn = &hbuf[(s - s_begin + 1) & 1][jb][0];
        float *yrow = ybase + (row0 + (size_t)t * SEQ_TILE) * a.ldy;
        // Row blocks one after the other, each as 4 independent accumulation chains (k mod 4):
        // dependent MFMAs are 4 issues apart.  Everything that does not depend on the running chain
        // is written BETWEEN its MFMA groups: the prefetch-ring rotation and the next gate load inside
        // row block 0's chain, the cell update of row block rb inside row block rb+1's chain.
        GateState S[RB];
        UVAD_STAMP_AT(3)   // [3] = barrier wait + loop top
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            S[rb].a0 = gq[u][rb];
            if constexpr (HAS_G2) S[rb].a0 += gq2[u][rb];
            S[rb].a1 = S[rb].a2 = S[rb].a3 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (rb > 0) __builtin_amdgcn_sched_barrier(0);   // row block rb's chain starts after rb-1's ended
#pragma unroll
            for (int kq = 0; kq < H / 4; ++kq) {
                const float4 hq = hv[kq % HR];
#ifndef UVAD_ABL_NOLDSREAD
                if (RB == 1 && kq + HR < H / 4) hv[kq % HR] = *reinterpret_cast<const float4 *>(hb + 4 * (kq + HR));
#endif
#ifndef UVAD_ABL_NOMFMA
                S[rb].a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 0], hq.x, S[rb].a0, 0, 0, 0);
                S[rb].a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 1], hq.y, S[rb].a1, 0, 0, 0);
                S[rb].a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 2], hq.z, S[rb].a2, 0, 0, 0);
                S[rb].a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[rb][4 * kq + 3], hq.w, S[rb].a3, 0, 0, 0);
#else
                S[rb].a0[0] += w[rb][4 * kq] * hq.x;   // keeps W and h alive without the matrix pipe
#endif
                if (rb == 0 && kq == H / 8) {
                    // refill this ring slot with the gates of step min(s + PD, T - 1) (branch-free: a
                    // redundant re-load of the last row is harmless)
                    const int sp = s + PD < s_end ? s + PD : s_end - 1;
                    const int tp = reverse ? a.T - 1 - sp : sp;
#pragma unroll
                    for (int r2 = 0; r2 < RB; ++r2) {
                        gq_load(gq[u][r2], gbase + (row0 + (size_t)tp * SEQ_TILE) * a.ldg + unit[r2] * 4);
                        if constexpr (HAS_G2) gq_load(gq2[u][r2], gbase2 + (row0 + (size_t)tp * SEQ_TILE) * a.ldg + unit[r2] * 4);
                    }
                }
                if (rb > 0 && kq <= GATE_STAGES) __builtin_amdgcn_sched_barrier(0);   // pin: MFMA group | stage | MFMA group ...
                if (rb > 0) {   // cell update of the previous row block, one stage per MFMA group
                    if (kq == 0) gate_stage<0>(S[rb - 1], c[rb - 1]);
                    if (kq == 1) gate_stage<1>(S[rb - 1], c[rb - 1]);
                    if (kq == 2) gate_stage<2>(S[rb - 1], c[rb - 1]);
                    if (kq == 3) gate_stage<3>(S[rb - 1], c[rb - 1]);
                    if (kq == 4) gate_stage<4>(S[rb - 1], c[rb - 1]);
                    if (kq == 5) gate_stage<5>(S[rb - 1], c[rb - 1]);
                    if (kq == 6) gate_stage<6>(S[rb - 1], c[rb - 1]);
                    if (kq == 7) gate_stage<7>(S[rb - 1], c[rb - 1]);
                    if (kq == 8) gate_stage<8>(S[rb - 1], c[rb - 1]);
                    if (kq == 9) {
                        gate_stage<9>(S[rb - 1], c[rb - 1]);
                        hlast[rb - 1] = S[rb - 1].h;
                        hn[unit[rb - 1]] = S[rb - 1].h;
                        UVAD_YSTORE(&yrow[unit[rb - 1]], S[rb - 1].h);
                    }
                }
            }
        }
        if constexpr (RB == 1) {
            // keep HR LDS reads in flight: [HR reads] then [4 MFMA + 1 read] per group (without this the
            // scheduler sinks every read to one group before its use and the chain stalls on LDS latency)
            __builtin_amdgcn_sched_group_barrier(0x100, HR, 0);
#pragma unroll
            for (int i = 0; i < H / 4 - HR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * HR, 0);
        }
        UVAD_STAMP_AT(0)   // [0] = h reads + MFMA chains
        gate_stages_upto<GATE_STAGES - 1>(S[RB - 1], c[RB - 1]);   // the exposed tail: last row block
        hlast[RB - 1] = S[RB - 1].h;
        hn[unit[RB - 1]] = S[RB - 1].h;
        UVAD_YSTORE(&yrow[unit[RB - 1]], S[RB - 1].h);
        UVAD_STAMP_AT(1)   // [1] = cell update + h write/store
#ifndef UVAD_ABL_NOSYNC
        __syncthreads();
#endif
      }
    }

#ifdef UVAD_STAMP
    if (lane == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.hN) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 4;
        for (int i = 0; i < 4; ++i) o[i] = st_acc[i];
    }
    return;
#endif
    if (a.hN) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const size_t so = ((size_t)dir * nseq + seq) * H + unit[rb];
            a.hN[so] = hlast[rb];
            a.cN[so] = c[rb];
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Skewed variant (H = 128, 8 waves): the production kernel for the reference geometry.
//
// Same data layout, same arithmetic and the same per-wave MFMA chains as lstm_rec_kernel<128, 8>,
// but each step is cut in two at k = H/2 and the two wave groups (G0 = waves 0-3 = units 0..63,
// G1 = waves 4-7 = units 64..127; one wave of each group per SIMD) run HALF A STEP APART:
//
//      barrier #2t            barrier #2t+1            barrier #2t+2
//   G0:  | B(t) + cell update(t)   |  A(t+1)                 |  B(t+1) + cell update ...
//   G1:  | A(t)                    |  B(t) + cell update(t)  |  A(t+1)
//
// A(t) = the 64 MFMAs that consume h_{t-1} of G0's units, B(t) = the 64 that consume G1's.  In every
// interval one wave of a SIMD runs a chain plus its (VALU, LDS) cell update while its partner runs a
// bare chain, so the matrix pipe -- the bound of this kernel -- keeps issuing during the cell update
// and the LDS hand-off instead of idling through them as it does when all waves move in lock step.
// Every wave executes the same program; G1 simply passes one extra barrier before the loop (and G0 one
// after it).  Double buffering of the h tile makes the half-step skew safe: h_t of a group is written
// one full interval before anyone reads it and overwritten two steps later.
template <int H>
__global__ __launch_bounds__(512) void lstm_rec_skew_kernel(LstmArgs a) {
    constexpr int HS = H + 4, NQ = H / 8;   // NQ float4 reads per half chain
    static_assert(H == 128, "one 16-unit MFMA row block per wave, 8 waves");
    __shared__ __attribute__((aligned(16))) float hbuf[2][SEQ_TILE][HS];

    const int tile = blockIdx.x, dir = blockIdx.y;
    const bool reverse = dir == 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    const int jb = lane & 3, blk = lane >> 2;

    float w[H];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(a.Whh_packed + (size_t)dir * 4 * H * H);
#pragma unroll
        for (int kq = 0; kq < H / 4; ++kq) {
            const float4 v = wp[(size_t)(wave * (H / 4) + kq) * 64 + lane];
            w[4 * kq + 0] = v.x; w[4 * kq + 1] = v.y; w[4 * kq + 2] = v.z; w[4 * kq + 3] = v.w;
        }
    }
    const int seq = tile * SEQ_TILE + jb, nseq = a.tiles * SEQ_TILE;
    const int unit = wave * 16 + blk;
    const size_t so = ((size_t)dir * nseq + seq) * H + unit;
    float c = a.c0 ? a.c0[so] : 0.0f;
    float hlast = 0.0f;
    hbuf[0][jb][unit] = a.h0 ? a.h0[so] : 0.0f;
    __syncthreads();

    const size_t row0 = (size_t)tile * a.T * SEQ_TILE + jb;
    const float *gbase = a.G + (size_t)dir * 4 * H + unit * 4;
    float *ybase = a.Y + (size_t)dir * H + unit;

    f32x4 gq[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int sp = p < a.T ? p : a.T - 1;
        const int t = reverse ? a.T - 1 - sp : sp;
        gq[p] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)t * SEQ_TILE) * a.ldg);
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the half-step skew (wave-uniform branch)

    for (int s0 = 0; s0 < a.T; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s >= a.T) break;   // wave-uniform
        const int t = reverse ? a.T - 1 - s : s;
#pragma unroll
        for (int k = 0; k < H; ++k) asm volatile("" : "+a"(w[k]));   // W_hh stays in AGPRs (constraint only)

        const float *hb = &hbuf[s & 1][jb][0];
        GateState S;
        S.a0 = gq[u];
        S.a1 = S.a2 = S.a3 = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 hv[NQ];
        // ---- A(s): k in [0, H/2) -- h_{s-1} of group 0's units ------------------------------------
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(hb + 4 * kq);
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) {
            S.a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 0], hv[kq].x, S.a0, 0, 0, 0);
            S.a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 1], hv[kq].y, S.a1, 0, 0, 0);
            S.a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 2], hv[kq].z, S.a2, 0, 0, 0);
            S.a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 3], hv[kq].w, S.a3, 0, 0, 0);
            if (kq == NQ / 2) {   // refill this ring slot with the gates of step min(s + PD, T - 1)
                const int sp = s + PD < a.T ? s + PD : a.T - 1;
                const int tp = reverse ? a.T - 1 - sp : sp;
                gq[u] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)tp * SEQ_TILE) * a.ldg);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NQ, 0);       // all NQ reads in flight, then the chain
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ, 0);
        __syncthreads();                                            // barrier #2s (G0) / #2s+1 (G1)

        // ---- B(s): k in [H/2, H) -- h_{s-1} of group 1's units; then this wave's cell update --------
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(hb + H / 2 + 4 * kq);
        __builtin_amdgcn_s_setprio(1);   // this wave has the long interval: its MFMAs go first
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) {
            S.a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[H / 2 + 4 * kq + 0], hv[kq].x, S.a0, 0, 0, 0);
            S.a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[H / 2 + 4 * kq + 1], hv[kq].y, S.a1, 0, 0, 0);
            S.a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[H / 2 + 4 * kq + 2], hv[kq].z, S.a2, 0, 0, 0);
            S.a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[H / 2 + 4 * kq + 3], hv[kq].w, S.a3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, NQ, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ, 1);
        __builtin_amdgcn_s_setprio(0);
        gate_stages_upto<GATE_STAGES - 1>(S, c);
        hlast = S.h;
        hbuf[(s + 1) & 1][jb][unit] = S.h;
        ybase[(row0 + (size_t)t * SEQ_TILE) * a.ldy] = S.h;
        __syncthreads();                                            // barrier #2s+1 (G0) / #2s+2 (G1)
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // G0 balances the barrier count
    if (a.hN) {
        a.hN[so] = hlast;
        a.cN[so] = c;
    }
}


// ---------------------------------------------------------------------------------------------
// Throughput variant for large batches (H = 128): 16 sequences per workgroup on v_mfma_f32_16x16x4_f32.
//
// When tiles*dirs exceeds the CU count the recurrence is no longer latency- but throughput-bound, and
// the 2-pass 4x4x1 MFMA is the wrong instruction: it holds the vector issue port for its whole 8
// cycles, so the cell updates are serialised behind the chains.  The 16x16x4 form does
// the same MACs per cycle but issues once per 32 cycles, leaving 24 issue slots per MFMA to the VALU /
// LDS work of both waves of the SIMD.  It needs N = 16 columns = 16 sequences per workgroup:
//   A (16 rows x 4 k)  = W_hh rows of 4 units x 4 gates,   k-step ks covers k = 32*kk + ks (kk = lane>>4)
//   B (4 k x 16 seqs)  = h_{t-1},  D: lane (q = lane>>4, j = lane&15) holds gates i,f,g,o (regs 0..3) of unit
//   16*wave + 4*rb + q for sequence j  -> the cell update is lane-local again.
// 8 waves x (4 row blocks x 32 k-steps) = 1024 MFMAs of 32 cycles per step and workgroup; W_hh stays in 128
// AGPRs per lane.  Rows keep the SEQ_TILE = 4 layout (a workgroup owns 4 consecutive tiles), so GEMMs and
// the classifier are unchanged.  No state carry / chunking (the callers that need those run small batches).
template <int H>
__global__ __launch_bounds__(512) void lstm_rec16_kernel(LstmArgs a) {
    static_assert(H == 128, "written for H = 128");
    constexpr int NS = 16, RB = 4, KS = H / 4, HSK = 36, PD16 = 2;
    __shared__ __attribute__((aligned(16))) float hbuf[2][4][NS][HSK];   // [buffer][kk][sequence][32 (+4 pad)]

    const int tile16 = blockIdx.x, dir = blockIdx.y;
    const bool reverse = dir == 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;

    float w[RB * KS];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(a.Whh_packed16 + (size_t)dir * 4 * H * H);
#pragma unroll
        for (int i = 0; i < RB * KS / 4; ++i) {
            const float4 v = wp[(size_t)(wave * (RB * KS / 4) + i) * 64 + lane];
            w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    }
    // this lane's sequence: tile (of 4) and row offset; lanes of missing tiles in the last workgroup are clamped
    // for loads and masked for stores
    const int t4 = tile16 * 4 + (j >> 2);
    const bool live = t4 < a.tiles;
    const int t4c = live ? t4 : a.tiles - 1;
    const size_t row0 = (size_t)t4c * a.T * SEQ_TILE + (j & 3);
    const int ubase = wave * 16 + q;   // unit of row block rb: ubase + 4*rb
    const float *gbase = a.G + (size_t)dir * 4 * H + (size_t)ubase * 4;
    float *ybase = a.Y + (size_t)dir * H + ubase;

    float c[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        c[rb] = 0.0f;
        const int u = ubase + 4 * rb;
        hbuf[0][u >> 5][j][u & 31] = 0.0f;
    }
    __syncthreads();

    f32x4 gq[PD16][RB];
#pragma unroll
    for (int p = 0; p < PD16; ++p) {
        const int sp = p < a.T ? p : a.T - 1;
        const int t = reverse ? a.T - 1 - sp : sp;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            gq[p][rb] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)t * SEQ_TILE) * a.ldg + 16 * rb);
    }

    for (int s0 = 0; s0 < a.T; s0 += PD16) {
#pragma unroll
      for (int u2 = 0; u2 < PD16; ++u2) {
        const int s = s0 + u2;
        if (s >= a.T) break;   // wave-uniform
        const int t = reverse ? a.T - 1 - s : s;
#pragma unroll
        for (int k = 0; k < RB * KS; ++k) asm volatile("" : "+a"(w[k]));   // W_hh stays in AGPRs (constraint only)

        // B operands: h_{s-1}[seq j][32*q + ks], ks = 0..31, as 8 x 16-byte reads
        float4 hv[KS / 4];
        const float *hb = &hbuf[s & 1][q][j][0];
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) hv[i] = *reinterpret_cast<const float4 *>(hb + 4 * i);

        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = gq[u2][rb];
        {   // refill this ring slot with the gates of step min(s + PD16, T - 1)
            const int sp = s + PD16 < a.T ? s + PD16 : a.T - 1;
            const int tp = reverse ? a.T - 1 - sp : sp;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                gq[u2][rb] = *reinterpret_cast<const f32x4 *>(gbase + (row0 + (size_t)tp * SEQ_TILE) * a.ldg + 16 * rb);
        }
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[rb * KS + 4 * i + 0], hv[i].x, acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[rb * KS + 4 * i + 1], hv[i].y, acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[rb * KS + 4 * i + 2], hv[i].z, acc[rb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[rb * KS + 4 * i + 3], hv[i].w, acc[rb], 0, 0, 0);
        }
        float *yrow = ybase + (row0 + (size_t)t * SEQ_TILE) * a.ldy;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            GateState S;
            S.a0 = acc[rb];
            S.a1 = S.a2 = S.a3 = f32x4{0.f, 0.f, 0.f, 0.f};
            gate_stages_upto<GATE_STAGES - 1>(S, c[rb]);
            const int u = ubase + 4 * rb;
            hbuf[(s + 1) & 1][u >> 5][j][u & 31] = S.h;
            if (live) yrow[4 * rb] = S.h;
        }
        __syncthreads();
      }
    }
}

}  // namespace

size_t whh_packed_elems(int H) { return (size_t)4 * H * H; }

int lstm_waves(int H) { return H == 128 ? 8 : 4; }

void pack_whh(const float *w_hh, int H, float *out) {
    const int WAVES = lstm_waves(H), UW = H / WAVES, RB = UW / 16;
    for (int wave = 0; wave < WAVES; ++wave)
        for (int rb = 0; rb < RB; ++rb)
            for (int kq = 0; kq < H / 4; ++kq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int gate = lane & 3, unit = wave * UW + rb * 16 + (lane >> 2);
                        out[((size_t)((wave * RB + rb) * (H / 4) + kq) * 64 + lane) * 4 + e] =
                            w_hh[(size_t)(gate * H + unit) * H + 4 * kq + e];
                    }
}

// register image of lstm_rec16_kernel: [wave 8][rb 4][ks 32 (as 8 float4)][lane 64]
void pack_whh16(const float *w_hh, int H, float *out) {
    const int KS = H / 4;
    for (int wave = 0; wave < 8; ++wave)
        for (int rb = 0; rb < 4; ++rb)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, kk = lane >> 4;
                    const int gate = i & 3, unit = wave * 16 + 4 * rb + (i >> 2);
                    const int idx = rb * KS + ks;   // element index in the lane's w[] array
                    out[((size_t)(wave * (4 * KS / 4) + idx / 4) * 64 + lane) * 4 + (idx & 3)] =
                        w_hh[(size_t)(gate * H + unit) * H + 32 * kk + ks];
                }
}

hipError_t launch_lstm(const LstmArgs &a, hipStream_t s) {
    if (a.tiles <= 0 || a.T <= 0) return hipSuccess;
    dim3 grid(a.tiles, a.dirs);
    // large batches: 16 sequences per workgroup (throughput variant); UVAD_LSTM=tile16 / tile4 force either
    static const int force16 = [] { const char *e = std::getenv("UVAD_LSTM"); return !e ? 0 : std::strcmp(e, "tile16") == 0 ? 1 : std::strcmp(e, "tile4") == 0 ? -1 : 0; }();
    const bool can16 = a.H == 128 && a.Whh_packed16 && !a.G2 && a.s_count == 0 && !a.h0 && !a.hN;
    if (can16 && (force16 > 0 || (force16 == 0 && a.tiles * a.dirs >= 512))) {
        hipLaunchKernelGGL(lstm_rec16_kernel<128>, dim3((a.tiles + 3) / 4, a.dirs), dim3(512), 0, s, a);
        return hipGetLastError();
    }
    static const bool plain = [] { const char *e = std::getenv("UVAD_LSTM"); return !(e && std::strcmp(e, "skew") == 0); }();   // UVAD_LSTM=skew selects the half-step variant (measured 2 % slower)
    if (a.H == 128 && !plain && !a.G2 && a.s_count == 0)
        hipLaunchKernelGGL(lstm_rec_skew_kernel<128>, grid, dim3(512), 0, s, a);
    else if (a.H == 128 && a.G2)
        hipLaunchKernelGGL((lstm_rec_kernel<128, 8, true>), grid, dim3(512), 0, s, a);
    else if (a.H == 128)
        hipLaunchKernelGGL((lstm_rec_kernel<128, 8, false>), grid, dim3(512), 0, s, a);
    else if (a.H == 64 && a.G2)
        hipLaunchKernelGGL((lstm_rec_kernel<64, 4, true>), grid, dim3(256), 0, s, a);
    else if (a.H == 64)
        hipLaunchKernelGGL((lstm_rec_kernel<64, 4, false>), grid, dim3(256), 0, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace uvad

// lstm.hip -- the sequential half of nn.LSTM (reference: the time loop inside
// `self.lstm(outputs)`, src/models/segmentation/PyanNet2.py:169-172; gate order i,f,g,o,
// zero initial state, eval mode).  The time-parallel half (x_t*W_ih^T + biases) is the GEMM.
//
// gfx950 design (persistent-RNN): one workgroup owns SEQ_TILE = 4 sequences of one direction for
// ALL T steps, so no workgroup ever talks to another.  W_hh (4H x H f32 = 256 KiB at H = 128) does
// not fit the 160 KiB LDS, but it fits the CU's 512 KiB register file: each of the 8 waves (two per
// SIMD, 256-register budget each) keeps the 64 rows (16 hidden units x 4 gates) of its units as
// 128 resident AGPRs per lane, laid out as the A operand of v_mfma_f32_4x4x1_16B_f32:
//      block b (16 per instruction) = hidden unit, A rows = its 4 gates (i,f,g,o), K = 1,
//      B = h_{t-1}[k] for the 4 sequences (identical in every block), D[gate][seq].
// A lane therefore ends the 128-deep chain holding all four gate pre-activations of ONE
// (unit, sequence) pair: the cell update is lane-local, no shuffles.  h_t is exchanged
// between the waves through a double-buffered 2 KiB LDS tile (one barrier per step) and the
// next steps' gate pre-activations are prefetched from HBM PD steps ahead.  H = 64 runs the same
// kernel with 4 waves.  Exact f32: the MFMA is a k-ordered fmaf chain (bit-exact f32).
//
// Gate functions: the recurrence feeds every rounding error back into (h, c) 4 x T times, so the
// sigmoid / tanh here are held to ~1 ulp of ABSOLUTE error at their output scale (v_exp_f32 +
// v_rcp_f32 refined by one Newton step; tanh as (1 - e)/(1 + e) with e = exp(-2|x|), which has no
// cancellation against a rounded 2*sigmoid), see DESIGN.md section 4 for the measured effect.
#include "uvad_internal.h"

namespace uvad {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifndef UVAD_LSTM_PD
#define UVAD_LSTM_PD 4   // gate prefetch depth (steps); 8 fits (252 of 256 registers) and measured the same
#endif

#ifndef UVAD_LSTM_HR
#define UVAD_LSTM_HR 8    // h reads kept in flight ahead of the MFMA groups (float4 each).  8: 76 VGPRs + 128 AGPRs = 208 registers per wave, which
                          // leaves room for one split-f16 GEMM wave (96) beside the two recurrent waves of a SIMD; measured 1 % faster alone than 16
#endif

constexpr float L2E = 1.4426950408889634f;

// 1 / d for d in [1, 2^127): v_rcp_f32 (1 ulp) + one Newton step -> <= 0.5 ulp + 2^-46
__device__ __forceinline__ float rcp_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float sigmoid_f(float x) {
    // exponent clamped so that 1 + e stays finite (sigmoid(-87) = 1.6e-38 either way)
    const float e = __builtin_amdgcn_exp2f(__builtin_fminf(-L2E * x, 126.0f));
    return rcp_nr(1.0f + e);
}
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __builtin_amdgcn_exp2f((-2.0f * L2E) * __builtin_fabsf(x));   // (0, 1]
    const float n = 1.0f - e, d = 1.0f + e;                                       // n exact for e >= 0.5 (Sterbenz)
    const float r = __builtin_amdgcn_rcpf(d);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);                           // correctly rounded n / d up to 2^-46
    return __builtin_copysignf(q, x);
}

// The cell update of one (unit, sequence) pair: g = gate pre-activations (i, f, g, o), c updated in place, returns h.
__device__ __forceinline__ float lstm_cell(const f32x4 g, float &c) {
    const float ig = sigmoid_f(g[0]), fg = sigmoid_f(g[1]), gg = tanh_f(g[2]), og = sigmoid_f(g[3]);
    c = __builtin_fmaf(fg, c, ig * gg);
    return og * tanh_f(c);
}

// The cell update of the 16-sequence kernel, which is bound by vector-instruction issue (four cells per lane and step): two of the five
// reciprocals per cell disappear by dividing once per PRODUCT,
//     sigmoid(i) * tanh(g) = sign(g) (1 - eg) / ((1 + ei)(1 + eg)),   sigmoid(o) * tanh(c) = sign(c) (1 - ec) / ((1 + eo)(1 + ec))
// with e* = exp2 of the clamped negated pre-activation as in sigmoid_f / tanh_f (the denominators stay below 2^127: ei <= 2^125,
// eg <= 1).  Each quotient is v_rcp_f32 + one Newton step on the quotient, i.e. correctly rounded up to 2^-46 like tanh_f's (without the
// Newton step the launch is SLOWER -- its fmas fill the transcendentals' issue slots -- and the x4 logit error 1.3 - 2.2 x larger: DESIGN.md 5b-3).
// Up to round 4 two cells were updated per packed f32 instruction (lstm_cell2); since round 5 row block rb is updated right behind that row
// block's MFMAs, one cell at a time with the same operations (every packed instruction was two independent IEEE operations: the logits are
// bit-identical), so that the update of row blocks 0 .. 2 is issued among the MFMAs of the next one and only ONE cell update -- not a pair --
// is left exposed behind the last MFMA of a step: 1.815 -> 1.802 ms per launch (A/B, 3 x 4 layers each).
__device__ __forceinline__ float quot_1(float n, float d) {   // n / d, d in [1, 2^127)
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ void lstm_cell1(const f32x4 g, float &c, float &h) {
    const float ai = __builtin_fminf(g[0] * -L2E, 125.0f), af = __builtin_fminf(g[1] * -L2E, 125.0f), ao = __builtin_fminf(g[3] * -L2E, 125.0f);
    const float ag = (-2.0f * L2E) * __builtin_fabsf(g[2]);
    const float ei = __builtin_amdgcn_exp2f(ai), ef = __builtin_amdgcn_exp2f(af), eg = __builtin_amdgcn_exp2f(ag), eo = __builtin_amdgcn_exp2f(ao);
    float itg = quot_1(1.0f - eg, (1.0f + ei) * (1.0f + eg));                        // sigmoid(i) * |tanh(g)|
    itg = __builtin_copysignf(itg, g[2]);
    const float fg = quot_1(1.0f, 1.0f + ef);
    c = __builtin_fmaf(fg, c, itg);
    const float ec = __builtin_amdgcn_exp2f((-2.0f * L2E) * __builtin_fabsf(c));
    h = __builtin_copysignf(quot_1(1.0f - ec, (1.0f + eo) * (1.0f + ec)), c);      // sigmoid(o) * |tanh(c)|
}

// Gate prefetch: plain loads into a ring of PD register slots (the time loop is unrolled by PD so
// every slot is a fixed register).  An inline-asm load with hand-counted vmcnt was tried and
// measured no faster, and it is fragile (hipcc may reuse an asm load's destination before the data
// lands), so the compiler's own waitcnt bookkeeping is kept.
__device__ __forceinline__ void gq_load(f32x4 &dst, const float *p) {
    dst = *reinterpret_cast<const f32x4 *>(p);
}

// h_t -> the two f16 planes the next layer's f16p GEMM reads (gemm_f16p.hip: a ~= hi + lo * 2^-11; |h| < 1, so both pieces are
// far inside the f16 range)
__device__ __forceinline__ void store_planes(unsigned short *ph, unsigned short *pl, float h) {
    const _Float16 hh = (_Float16)h;
    const _Float16 hl = (_Float16)((h - (float)hh) * 2048.0f);
    *ph = __builtin_bit_cast(unsigned short, hh);
    *pl = __builtin_bit_cast(unsigned short, hl);
}

template <int H, int WAVES, bool PLANES>
__global__ __launch_bounds__(WAVES * 64) void lstm_rec_kernel(LstmArgs a) {
    constexpr int HS = H + 4;   // LDS row stride (floats): the 4 sequence rows land on disjoint banks
    constexpr int PD = UVAD_LSTM_PD;
    static_assert(H == WAVES * 16, "one 16-unit MFMA row block per wave");

    __shared__ __attribute__((aligned(16))) float hbuf[2][SEQ_TILE][HS];

    const int tile = blockIdx.x, dir = blockIdx.y;
    const bool reverse = dir == 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jb = lane & 3;    // sequence within the tile (B / D operand column), gate index of the A operand
    const int blk = lane >> 2;  // MFMA block = hidden unit within the wave's row block

    // ---- resident recurrent weights -------------------------------------------------------
    float w[H];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(a.Whh_packed + (size_t)dir * 4 * H * H);
#pragma unroll
        for (int kq = 0; kq < H / 4; ++kq) {
            const float4 v = wp[(size_t)(wave * (H / 4) + kq) * 64 + lane];
            w[4 * kq + 0] = v.x; w[4 * kq + 1] = v.y; w[4 * kq + 2] = v.z; w[4 * kq + 3] = v.w;
        }
    }

    // ---- state ----------------------------------------------------------------------------
    const int seq = tile * SEQ_TILE + jb;                 // padded batch index
    const int nseq = a.tiles * SEQ_TILE;
    const int unit = wave * 16 + blk;
    const size_t so = ((size_t)dir * nseq + seq) * H + unit;
    float c = a.c0 ? a.c0[so] : 0.0f;
    hbuf[0][jb][unit] = a.h0 ? a.h0[so] : 0.0f;
    __syncthreads();

    // Row of (t, jb) in the tile-major activation matrices = rowu + 4*t + jb, rowu wave-uniform.  The blocked layouts
    // (g_index / plane_index) split into a wave-uniform part that changes with t (scalar unit) and a per-lane constant:
    //   G: the wave's 16 units x 4 gates are exactly one 64-column tile, the 4 sequence rows adjacent -> 1 KiB contiguous per step
    //   Y planes: the wave's 16 units are exactly one 16-column block, 4 rows adjacent -> 128 contiguous bytes per plane and step
    const size_t rowu = (size_t)tile * a.T * SEQ_TILE;
    const size_t row0 = rowu + jb;
    const size_t g_wave = (size_t)(dir * (4 * H / 64) + wave) * (128 * 64);
    const unsigned g_lane = (unsigned)(jb * 64 + blk * 4);
    const size_t g_tile = (size_t)(a.ldg / 64) * (128 * 64);          // floats per 128-row tile of G
    auto g_ptr = [&](int t) {
        const size_t R = rowu + (size_t)t * SEQ_TILE;
        return a.G + (R >> 7) * g_tile + g_wave + (R & 127) * 64 + g_lane;
    };
    const size_t ycol = (size_t)dir * H + unit;
    const size_t y_wave = (size_t)(dir * (H / 16) + wave) * (PLANE_TILE * 16);
    const unsigned y_lane = (unsigned)(jb * 16 + blk);
    const size_t y_tile = (size_t)(a.ldy / 16) * (PLANE_TILE * 16);   // elements per 128-row tile of a Y plane

    // the steps of this launch: all T frames, or a chunk of them continuing from the carried state (LstmArgs::steps)
    const int nst = a.steps > 0 ? a.steps : a.T;
    const int t_lo = a.steps > 0 ? a.t_begin[dir] : 0;
    auto t_of = [&](int s) { return reverse ? t_lo + nst - 1 - s : t_lo + s; };
    f32x4 gq[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) gq_load(gq[p], g_ptr(t_of(p < nst ? p : nst - 1)));

    float hlast = 0.0f;

    // Time loop unrolled by the prefetch depth: step s uses ring slot s % PD and refills it with the
    // gates of step s + PD, so every load lands in the register it is consumed from PD steps later
    // (no register rotation => the compiler can wait with a counted vmcnt instead of vmcnt(0), and
    // the h stores of the last steps stay in flight).
    for (int s0 = 0; s0 < nst; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s >= nst) break;   // wave-uniform
        const int t = t_of(s);
        // Pin the resident weights in the accumulator half of the unified register file: MFMA reads
        // A operands straight from AGPRs, and VALU-addressable VGPRs stay free for h / gates.
        // (Zero instructions: the constraint only tells the allocator where the values live here.)
#pragma unroll
        for (int k = 0; k < H; ++k) asm volatile("" : "+a"(w[k]));

        // h_{t-1} of this lane's sequence, all H values (broadcast reads: 4 distinct addresses per wave),
        // streamed through a ring of HR slots refilled right after use, so that HR LDS reads stay in
        // flight ahead of the MFMA groups (LDS latency under 8 reading waves is several MFMA groups
        // long; a read issued one group ahead stalls the matrix pipe).
        const float *hb = &hbuf[s & 1][jb][0];
        constexpr int HR = UVAD_LSTM_HR;
        float4 hv[HR];
#pragma unroll
        for (int kq = 0; kq < HR; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(hb + 4 * kq);
        // 4 independent accumulation chains (k mod 4): dependent MFMAs are 4 issues apart
        f32x4 a0 = gq[u], a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < H / 4; ++kq) {
            const float4 hq = hv[kq % HR];
            if (kq + HR < H / 4) hv[kq % HR] = *reinterpret_cast<const float4 *>(hb + 4 * (kq + HR));
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 0], hq.x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 1], hq.y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 2], hq.z, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 3], hq.w, a3, 0, 0, 0);
            if (kq == H / 8) {
                // refill this ring slot with the gates of step min(s + PD, T - 1) (branch-free: a
                // redundant re-load of the last row is harmless)
                gq_load(gq[u], g_ptr(t_of(s + PD < nst ? s + PD : nst - 1)));
            }
        }
        // keep HR LDS reads in flight: [HR reads] then [4 MFMA + 1 read] per group (without this the
        // scheduler sinks every read to one group before its use and the chain stalls on LDS latency)
        __builtin_amdgcn_sched_group_barrier(0x100, HR, 0);
#pragma unroll
        for (int i = 0; i < H / 4 - HR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * HR, 0);
        hlast = lstm_cell((a0 + a1) + (a2 + a3), c);
        hbuf[(s + 1) & 1][jb][unit] = hlast;
        if constexpr (PLANES) {
            // K-blocked plane (uvad_internal.h plane_index): the 4 sequence rows x 16 units of a wave are 128 contiguous bytes
            const size_t R = rowu + (size_t)t * SEQ_TILE;
            const size_t yo = (R >> 7) * y_tile + y_wave + (R & 127) * 16 + y_lane;
            store_planes(a.Yh + yo, a.Yl + yo, hlast);
        } else {
            *(a.Y + (row0 + (size_t)t * SEQ_TILE) * a.ldy + ycol) = hlast;
        }
        __syncthreads();
      }
    }

    if (a.hN) {
        a.hN[so] = hlast;
        a.cN[so] = c;
    }
}


// ---------------------------------------------------------------------------------------------
// Throughput form (H = 128): 16 sequences per workgroup, W_hh * h on the f16 matrix cores with the SAME operand split as
// gemm_f16p.hip -- W_hh scaled by a power of two and split into three f16 planes that reproduce it exactly (a rounded
// static weight would be a different network), h_t split into two f16 planes (22 bits, noise) -- four
// v_mfma_f32_16x16x32_f16 products per term set in two f32 accumulators:
//     hi += P0*h1;   lo += P1*h1 + P0*h2 + P2*h1;   W_hh*h = (hi + lo * 2^-11) * 2^-S
// 64 MFMAs of 16 cycles per wave and step instead of the 128 x 32 cycles of the f32 form (v_mfma_f32_16x16x4_f32): the
// matrix pipe stops being the bound and the cell updates (4 per lane and step) run in the issue slots the MFMAs leave free.
//   A (16 rows x 32 k) = W_hh rows of 4 units x 4 gates: lane (r = lane & 15, kq = lane >> 4) holds k = 32 ks + 8 kq .. + 7
//   B (32 k x 16 seqs) = h_{t-1}: lane (j = lane & 15, kq) holds the same 8 k of sequence j
//   D: lane (q = lane >> 4, j) holds gates i,f,g,o (regs 0..3) of unit 16 wave + 4 rb + q for sequence j -> lane-local cell update
// W_hh residency: P0 and P1 of a wave's 64 rows are 128 AGPRs per lane (as the f32 image was); P2 (128 KiB per direction)
// does not fit beside them and lives in LDS, 16 KiB per wave, read as 16 conflict-free ds_read_b128 per step.
// h_t is exchanged through a double-buffered LDS image of its two f16 planes ([plane][sequence][128 + 8 pad]: rows 272 bytes
// apart, so the 16 sequence rows of a B-fragment read fall on disjoint banks).
constexpr int R16_HP = 136;                          // f16 elements per h row in LDS (128 + pad)
constexpr int R16_P2_ELEMS = 8 * 16 * 64 * 8;        // P2 image of one direction
constexpr int R16_HB_ELEMS = 2 * 2 * 16 * R16_HP;    // [buffer][plane][sequence][R16_HP]
constexpr int R16_P2Q_ELEMS = 8 * 4 * 64 * 16;       // the same image as bf8 (E5M2) bytes, counted in u16: [wave][row block][lane][32 bytes]
constexpr int R16_HQ = 144;                          // bytes per sequence row of the fp8 image of h (128 + pad, a multiple of 16)
constexpr int R16_HQ_ELEMS = 2 * 16 * R16_HQ / 2;    // [buffer][sequence][R16_HQ] in u16
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using i32x8 = __attribute__((ext_vector_type(8))) int;


// NPROD = 3: without the P2 x h1 product (LstmArgs::products): no P2 image in LDS (17 KiB instead of 145), 48 MFMAs per wave and step
// P2Q (NPROD = 4 only): the P2 x h1 product on the 8-bit matrix pipe.  P2 is the residue of two round-to-nearest f16 splits of a 24-bit
// significand -- 0, +-1 or +-2 units at a per-element exponent, at most two significant bits -- so, shifted by one power of two, it IS a bf8
// (E5M2) number (pack_whh16h_p2q verifies every element; E5M2 is the upper byte of an f16) and the WEIGHTS STAY EXACT; h enters that
// product rounded to fp8 (E4M3: a 2^-4 relative change of a term that is 2^-22 of the sum).  One v_mfma_scale_f32_16x16x128_f8f6f4 over
// the whole K = 128 (32 cycles; operand layout and the power-of-two scale operand checked with exact integers by tools/mx_probe.hip)
// replaces the four f16 MFMAs of a row block (64 cycles): 896 instead of 1 024 matrix cycles per wave and step, the P2 image 64 instead of
// 128 KiB.  The k order of that product is the kernel's own: byte 16 w + 4 q + rb of a sequence's fp8 row is unit 16 w + 4 rb + q, so the
// four cells a lane updates are one aligned dword of the image (one ds_write_b32); the host packs P2's columns in the same order.
template <bool PLANES, int NPROD, bool P2Q>
__global__ __launch_bounds__(512) void lstm_rec16h_kernel(LstmArgs a) {
    constexpr int H = 128, RB = 4, KST = 4;
    static_assert(!P2Q || NPROD == 4, "the fp8 form of the P2 product exists for four products only");
    extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
    unsigned short *p2 = sm16, *hb = sm16 + (NPROD == 4 ? (P2Q ? R16_P2Q_ELEMS : R16_P2_ELEMS) : 0);
    unsigned char *hq = reinterpret_cast<unsigned char *>(hb + R16_HB_ELEMS);   // (P2Q) [buffer][sequence][R16_HQ] fp8 image of h

    const int tile16 = blockIdx.x, dir = blockIdx.y;
    const bool reverse = dir == 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;

    // ---- resident P0 / P1: register r = ((plane * 16 + rb * 4 + ks) * 4 + v) holds elements 2v, 2v+1 of that fragment
    u32x4 w[32];   // fragment (plane, rb, ks) = w[plane * 16 + rb * 4 + ks]: one aligned 4-register tuple each
    {
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(a.Whh16h_regs + (size_t)dir * (8 * 128 * 64));
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = wp[(size_t)(wave * 32 + i) * 64 + lane];
    }
    // ---- P2 image of this direction -> LDS (128 KiB, once), h buffers zeroed (h_{-1} = 0)
    {
        uint4 *dst = reinterpret_cast<uint4 *>(p2);
        if constexpr (NPROD == 4 && P2Q) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.Whh16h_p2q + (size_t)dir * R16_P2Q_ELEMS);
            for (int i = threadIdx.x; i < R16_P2Q_ELEMS / 8; i += 512) dst[i] = src[i];
        } else if constexpr (NPROD == 4) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.Whh16h_p2 + (size_t)dir * R16_P2_ELEMS);
            for (int i = threadIdx.x; i < R16_P2_ELEMS / 8; i += 512) dst[i] = src[i];
        }
        for (int i = threadIdx.x; i < (R16_HB_ELEMS + (P2Q ? R16_HQ_ELEMS : 0)) / 2; i += 512) reinterpret_cast<unsigned *>(hb)[i] = 0u;
    }
    // this lane's sequence: tile (of 4) and row offset; lanes of missing tiles in the last workgroup are clamped
    // for loads and masked for stores
    const int t4 = tile16 * 4 + (j >> 2);
    const bool live = t4 < a.tiles;
    const int t4c = live ? t4 : a.tiles - 1;
    const size_t row0 = (size_t)t4c * a.T * SEQ_TILE + (j & 3);
    const int ubase = wave * 16 + q;   // unit of row block rb: ubase + 4*rb
    // Blocked layouts (g_index / plane_index): the wave's 64 gate columns are one 64-column tile of G and its 16 units one
    // 16-column block of Y, so row block rb is simply 16 floats (G) / 4 elements (Y) further
    const size_t g_wave = (size_t)(dir * 8 + wave) * (128 * 64) + 4 * q, g_tile = (size_t)(a.ldg / 64) * (128 * 64);
    const size_t y_tile = (size_t)(a.ldy / 16) * (PLANE_TILE * 16);
    const size_t ycol = (size_t)dir * H + ubase;
    const float wscale = a.whh16h_scale[dir];
    // Row pointers are stepped, not recomputed (the kernel is bound by vector-instruction issue): a step moves 4 rows inside a
    // 128-row tile of the blocked layouts (g_index / plane_index) or, on a wrap, to the neighbouring tile.
    const int sgn = reverse ? -1 : 1, t_first = reverse ? a.T - 1 : 0;
    const int g_in = 4 * 64 * sgn, g_wrap = ((int)g_tile - 124 * 64) * sgn;
    const int y_in = 4 * 16 * sgn, y_wrap = ((int)y_tile - 124 * 16) * sgn;
    auto advance = [&](auto *&ptr, int &rl, int d_in, int d_wrap) {
        rl += 4 * sgn;
        const bool wrap = (unsigned)rl >= 128u;
        rl &= 127;
        ptr += (ptrdiff_t)(wrap ? d_wrap : d_in);
    };
    int g_rl;
    const float *g_p;
    {
        const size_t R = row0 + (size_t)t_first * SEQ_TILE;
        g_rl = (int)(R & 127);
        g_p = a.G + (R >> 7) * g_tile + g_wave + (size_t)g_rl * 64;
    }

    float c[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) c[rb] = 0.0f;
    f32x4 gq[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) gq[rb] = *reinterpret_cast<const f32x4 *>(g_p + 16 * rb);
    __syncthreads();

    // Cooperative store of h_t as the two K-blocked f16 planes: the whole workgroup's output of a step is 8 KiB (2 planes x 16
    // sequences x 128 units), laid out in HBM as 64 runs of 128 bytes (4 sequence rows x 16 units); after the barrier it sits
    // complete in the LDS image the next step reads its B fragments from, and every thread moves ONE 16-byte piece of it
    // (per-lane 2-byte stores straight from the cell update cost 37 % of the kernel).
    // thread -> (plane, tile of 4 sequences, 16-unit block, row in tile, 16-byte half)
    int c_rl = 0, c_src = 0;
    unsigned short *c_p = nullptr;
    bool clive = false;
    if constexpr (PLANES) {
        const int tid = threadIdx.x;
        const int cp = tid >> 8, ctt = (tid >> 6) & 3, ckb = (tid >> 3) & 7, cjb = (tid >> 1) & 3, chh = tid & 1;
        const int ct4 = tile16 * 4 + ctt;
        clive = ct4 < a.tiles;
        const size_t R = (size_t)(clive ? ct4 : a.tiles - 1) * a.T * SEQ_TILE + cjb + (size_t)t_first * SEQ_TILE;
        c_rl = (int)(R & 127);
        c_p = (cp ? a.Yl : a.Yh) + (R >> 7) * y_tile + (size_t)(dir * 8 + ckb) * (PLANE_TILE * 16) + chh * 8 + (size_t)c_rl * 16;
        c_src = cp * (16 * R16_HP) + (ctt * 4 + cjb) * R16_HP + ckb * 16 + chh * 8;
    }
    auto coop_store = [&](const unsigned short *himg, bool step) {   // stores the frame c_p points at, then (step) moves on to the next
        if constexpr (PLANES) {
            const uint4 v = *reinterpret_cast<const uint4 *>(himg + c_src);
            if (clive) *reinterpret_cast<uint4 *>(c_p) = v;
            if (step) advance(c_p, c_rl, y_in, y_wrap);
        }
    };

    const unsigned short *p2w = p2 + (size_t)(wave * 16) * 512 + lane * 8;   // fragment (rb, ks) at + (rb*4 + ks) * 512
    // (P2Q) fragment rb at + rb * 2048, 32 bytes per lane.  A lane's two 16-byte halves sit in the order that makes BOTH ds_read_b128 of the
    // fragment bank-conflict-free: with the halves in k order, lanes l and l + 16 (+ 8 in the instruction's 16-lane groups) meet on one
    // 16-byte slot in each read (2-way: 52 % of the kernel's LDS-active cycles were conflicts); lanes 16 .. 31 and 48 .. 63 therefore keep
    // their SECOND half first (pack_whh16h_p2q stores it that way) and read at + 16 first.
    const int p2q_swap = 16 * ((lane >> 4) & 1);
    const unsigned char *p2qw = reinterpret_cast<const unsigned char *>(p2) + ((size_t)(wave * 4) * 64 + lane) * 32;
    const int hfrag = j * R16_HP + 8 * q;                                   // + 32 ks inside a plane
    const int hq_rd = j * R16_HQ + 32 * q, hq_wr = j * R16_HQ + 16 * wave + 4 * q;   // (P2Q) this lane's B fragment / the dword of its four cells
    const int p2q_scale = P2Q ? a.p2q_scale : 127;
    for (int s = 0; s < a.T; ++s) {
        const int t = reverse ? a.T - 1 - s : s;
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("" : "+a"(w[k]));   // P0 / P1 stay in AGPRs (constraint only)

        if (s + 1 < a.T) advance(g_p, g_rl, g_in, g_wrap);   // the gates of the next step (the last step re-reads its own)
        const float *gnext = g_p;
        const unsigned short *hcur = hb + (s & 1) * (2 * 16 * R16_HP);
        unsigned short *hnxt = hb + ((s + 1) & 1) * (2 * 16 * R16_HP);
        f16x8 h1[KST], h2[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            h1[ks] = *reinterpret_cast<const f16x8 *>(hcur + hfrag + 32 * ks);
            h2[ks] = *reinterpret_cast<const f16x8 *>(hcur + 16 * R16_HP + hfrag + 32 * ks);
        }
        i32x8 hqf = {};
        if constexpr (P2Q) {
            const unsigned char *hqc = hq + (s & 1) * (16 * R16_HQ) + hq_rd;
            const u32x4 lo4 = *reinterpret_cast<const u32x4 *>(hqc), hi4 = *reinterpret_cast<const u32x4 *>(hqc + 16);
            hqf = i32x8{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        }
        const size_t yrow = row0 + (size_t)t * SEQ_TILE;
        // Row blocks one after the other (16 MFMAs each); the P2 fragment of MFMA group f + 1 is requested from LDS before group f.
        // The kernel is bound by vector-instruction issue (~290 per wave and step: four cell updates per lane), not by the matrix
        // pipe (40 % busy), so scheduling is left to the compiler: staging the cell update of row block rb between the MFMA
        // groups of rb + 1 behind sched_barriers measured 10 % slower, a forced [4 MFMA, 1 LDS read] cadence 3 % slower.
        float hnew[RB];
        f32x4 gpre[RB];
        f16x8 w2n = {};
        if constexpr (NPROD == 4 && !P2Q) w2n = *reinterpret_cast<const f16x8 *>(p2w);
        auto p2q_frag = [&](int rb) {   // (P2Q) the bf8 fragment of row block rb: 32 bytes per lane
            const u32x4 lo4 = *reinterpret_cast<const u32x4 *>(p2qw + p2q_swap + rb * 2048), hi4 = *reinterpret_cast<const u32x4 *>(p2qw + (16 - p2q_swap) + rb * 2048);
            return i32x8{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        };
        i32x8 wqn = {};
        if constexpr (P2Q) wqn = p2q_frag(0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
            const i32x8 wq = wqn;
            if constexpr (P2Q)
                if (rb + 1 < RB) wqn = p2q_frag(rb + 1);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                const int f = rb * 4 + ks;
                const f16x8 w2 = w2n;
                if constexpr (NPROD == 4 && !P2Q)
                    if (f + 1 < 16) w2n = *reinterpret_cast<const f16x8 *>(p2w + (f + 1) * 512);
                const f16x8 w0 = __builtin_bit_cast(f16x8, w[f]);
                const f16x8 w1 = __builtin_bit_cast(f16x8, w[16 + f]);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, h1[ks], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h1[ks], hi, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, h2[ks], lo, 0, 0, 0);
                if constexpr (NPROD == 4 && !P2Q) lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, h1[ks], lo, 0, 0, 0);
            }
            // (P2Q) P2 (bf8, scaled back by 2^(p2q_scale - 127)) x h (fp8) over the whole K in one instruction
            if constexpr (P2Q) lo = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq, hqf, lo, 1, 0, 0, p2q_scale, 0, 127);
            gpre[rb] = __builtin_elementwise_fma(__builtin_elementwise_fma(lo, f32x4{0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f}, hi),
                                                 f32x4{wscale, wscale, wscale, wscale}, gq[rb]);
            gq[rb] = *reinterpret_cast<const f32x4 *>(gnext + 16 * rb);   // next step's gates of this row block: a whole step to arrive
            lstm_cell1(gpre[rb], c[rb], hnew[rb]);   // behind its own row block's MFMAs: overlaps the next row block's, one cell exposed at the end
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int u = ubase + 4 * rb;
            const _Float16 hh = (_Float16)hnew[rb];
            const _Float16 hl = (_Float16)((hnew[rb] - (float)hh) * 2048.0f);
            hnxt[j * R16_HP + u] = __builtin_bit_cast(unsigned short, hh);
            hnxt[16 * R16_HP + j * R16_HP + u] = __builtin_bit_cast(unsigned short, hl);
            if constexpr (!PLANES) {
                if (live) a.Y[yrow * a.ldy + ycol + 4 * rb] = hnew[rb];   // exact-f32 output: per lane (the planes in LDS hold 22 bits)
            }
        }
        if constexpr (P2Q) {   // the lane's four cells (row blocks 0 .. 3 = bytes 0 .. 3) as fp8, one dword of the next step's image
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(hnew[0], hnew[1], 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(hnew[2], hnew[3], pk, true);
            *reinterpret_cast<int *>(hq + ((s + 1) & 1) * (16 * R16_HQ) + hq_wr) = pk;
        }
        // h of the PREVIOUS step (complete in LDS since the last barrier, not overwritten before the next one) goes out now,
        // when the fragment registers of this step are dead
        if (s > 0) coop_store(hcur, true);
        __syncthreads();
    }
    if (a.T > 0) coop_store(hb + (a.T & 1) * (2 * 16 * R16_HP), false);
}


// ---------------------------------------------------------------------------------------------
// Any other hidden size (the reference constructor takes any, PyanNet2.py:82-95; its configurations use 128): the same recurrence
// without resident weights.  One workgroup = 4 sequences of one direction; a thread owns hidden units u = tid, tid + 256, ... (H <= 1024:
// at most 4) for all 4 sequences and, per step, forms their four gate pre-activations as plain f32 fmaf chains over k (the order a
// torch CPU / the oracle's row dot product has), W_hh rows (torch layout [4H][H]) streamed from L2 every step, h_{t-1} broadcast from
// LDS.  Slow by design -- the weight matrix crosses the CU once per step -- and exact f32; it exists so that no constructor argument of
// the reference is refused, the fast forms are the H = 128 / 64 kernels above.
template <bool PLANES>
__global__ __launch_bounds__(256) void lstm_rec_any_kernel(LstmArgs a) {
    constexpr int MAXU = 4;
    extern __shared__ __attribute__((aligned(16))) float hs[];   // [2][SEQ_TILE][HS]
    const int H = a.H, HS = H + 4;
    const int tile = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const bool reverse = dir == 1;
    const int nseq = a.tiles * SEQ_TILE;
    const float *W = a.Whh_packed + (size_t)dir * 4 * H * H;     // this form reads the plain torch matrix
    const int nst = a.steps > 0 ? a.steps : a.T;
    const int t_lo = a.steps > 0 ? a.t_begin[dir] : 0;
    float c[MAXU][SEQ_TILE], hl[MAXU][SEQ_TILE];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = tid + 256 * i;
#pragma unroll
        for (int j = 0; j < SEQ_TILE; ++j) {
            const size_t so = ((size_t)dir * nseq + tile * SEQ_TILE + j) * H + u;
            c[i][j] = (u < H && a.c0) ? a.c0[so] : 0.0f;
            hl[i][j] = (u < H && a.h0) ? a.h0[so] : 0.0f;
            if (u < H) hs[j * HS + u] = hl[i][j];
        }
    }
    __syncthreads();
    const size_t rowu = (size_t)tile * a.T * SEQ_TILE;
    for (int s = 0; s < nst; ++s) {
        const int t = reverse ? t_lo + nst - 1 - s : t_lo + s;
        const float *hc = hs + (s & 1) * (SEQ_TILE * HS);
        float *hn = hs + ((s + 1) & 1) * (SEQ_TILE * HS);
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i;
            if (u >= H) break;
            float acc[4][SEQ_TILE];
#pragma unroll
            for (int j = 0; j < SEQ_TILE; ++j) {
                const float4 g = *reinterpret_cast<const float4 *>(a.G + g_index(rowu + (size_t)t * SEQ_TILE + j, dir * 4 * H + u * 4, a.ldg));
                acc[0][j] = g.x; acc[1][j] = g.y; acc[2][j] = g.z; acc[3][j] = g.w;
            }
            for (int k = 0; k < H; k += 4) {
                float4 w4[4], h4[SEQ_TILE];
#pragma unroll
                for (int g = 0; g < 4; ++g) w4[g] = *reinterpret_cast<const float4 *>(W + ((size_t)g * H + u) * H + k);
#pragma unroll
                for (int j = 0; j < SEQ_TILE; ++j) h4[j] = *reinterpret_cast<const float4 *>(hc + j * HS + k);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < SEQ_TILE; ++j) {
                        acc[g][j] = __builtin_fmaf(w4[g].x, h4[j].x, acc[g][j]);
                        acc[g][j] = __builtin_fmaf(w4[g].y, h4[j].y, acc[g][j]);
                        acc[g][j] = __builtin_fmaf(w4[g].z, h4[j].z, acc[g][j]);
                        acc[g][j] = __builtin_fmaf(w4[g].w, h4[j].w, acc[g][j]);
                    }
            }
#pragma unroll
            for (int j = 0; j < SEQ_TILE; ++j) {
                const float h = lstm_cell(f32x4{acc[0][j], acc[1][j], acc[2][j], acc[3][j]}, c[i][j]);
                hl[i][j] = h;
                hn[j * HS + u] = h;
                const size_t row = rowu + (size_t)t * SEQ_TILE + j;
                if constexpr (PLANES) {
                    const size_t yo = plane_index(row, dir * H + u, a.ldy);
                    store_planes(a.Yh + yo, a.Yl + yo, h);
                } else {
                    a.Y[row * a.ldy + (size_t)dir * H + u] = h;
                }
            }
        }
        __syncthreads();
    }
    if (a.hN) {
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + 256 * i;
            if (u >= H) break;
#pragma unroll
            for (int j = 0; j < SEQ_TILE; ++j) {
                const size_t so = ((size_t)dir * nseq + tile * SEQ_TILE + j) * H + u;
                a.hN[so] = hl[i][j];
                a.cN[so] = c[i][j];
            }
        }
    }
}

}  // namespace

constexpr double REC16_ROUND_COST = 1.6;   // time of one round of the 16-sequence form / one round of the 4-sequence form (1.94 vs 1.33 ms per layer at T = 1000; 2.2 ms with every CU busy)

size_t whh_packed_elems(int H) { return (size_t)4 * H * H; }
size_t whh16h_regs_elems() { return (size_t)8 * 128 * 64; }        // u32 per direction
size_t whh16h_p2_elems() { return (size_t)R16_P2_ELEMS; }          // f16 per direction
size_t whh16h_p2q_elems() { return (size_t)R16_P2Q_ELEMS; }        // u16 per direction (64 KiB of bf8 bytes)

int lstm_waves(int H) { return H == 128 ? 8 : 4; }

// register image of lstm_rec_kernel: [wave][kq = k/4][lane 64][4 k] with lane = (unit within the wave) * 4 + gate
void pack_whh(const float *w_hh, int H, float *out) {
    if (H != 128 && H != 64) {   // lstm_rec_any_kernel reads the torch matrix as it is
        for (size_t i = 0; i < (size_t)4 * H * H; ++i) out[i] = w_hh[i];
        return;
    }
    const int WAVES = lstm_waves(H);
    for (int wave = 0; wave < WAVES; ++wave)
        for (int kq = 0; kq < H / 4; ++kq)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int gate = lane & 3, unit = wave * 16 + (lane >> 2);
                    out[((size_t)(wave * (H / 4) + kq) * 64 + lane) * 4 + e] = w_hh[(size_t)(gate * H + unit) * H + 4 * kq + e];
                }
}

// Images of lstm_rec16h_kernel for one direction, from torch w_hh [4H][H] (rows i,f,g,o), H = 128:
//   regs: P0 / P1 fragments as [wave 8][i 32][lane 64][4 x u32]; the lane's register r = 4 i + e = ((plane * 16 + rb * 4 + ks) * 4 + v)
//         holds elements 2v, 2v+1 (low, high half) of fragment (plane, rb, ks): row lane & 15 -> unit 16 wave + 4 rb + (row >> 2),
//         gate row & 3; k = 32 ks + 8 (lane >> 4) + element
//   p2:   P2 fragments as [wave 8][rb * 4 + ks 16][lane 64][8 f16]
//   *wscale = 2^-S, the power-of-two scale that put max|w| into [2^13, 2^14) before the exact three-way split (gemm_f16p.hip)
bool pack_whh16h(const float *w_hh, unsigned *regs, unsigned short *p2, float *wscale) {
    const int H = 128;
    float amax = 0.0f;
    bool finite = true;
    for (size_t i = 0; i < (size_t)4 * H * H; ++i) {
        const float v = __builtin_fabsf(w_hh[i]);
        if (!(v <= 3.0e38f)) finite = false;
        if (v > amax) amax = v;
    }
    int S = 0;
    if (finite && amax > 0.0f) {
        int e;
        (void)__builtin_frexpf(amax, &e);
        S = 14 - e;
        if (S > 100) S = 100;
        if (S < -100) S = -100;
    }
    const float up = __builtin_ldexpf(1.0f, S);
    *wscale = __builtin_ldexpf(1.0f, -S);
    auto piece = [&](int gate, int unit, int k, int plane) -> unsigned short {
        const float ws = finite ? w_hh[(size_t)(gate * H + unit) * H + k] * up : 0.0f;
        const _Float16 p0 = (_Float16)ws;
        const float t2 = (ws - (float)p0) * 2048.0f;
        const _Float16 p1 = (_Float16)t2;
        const _Float16 p2v = (_Float16)(t2 - (float)p1);
        const _Float16 v = plane == 0 ? p0 : plane == 1 ? p1 : p2v;
        unsigned short b;
        __builtin_memcpy(&b, &v, 2);
        return b;
    };
    for (int wave = 0; wave < 8; ++wave)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 15, kq = lane >> 4;
            for (int rb = 0; rb < 4; ++rb)
                for (int ks = 0; ks < 4; ++ks) {
                    const int unit = 16 * wave + 4 * rb + (row >> 2), gate = row & 3, f = rb * 4 + ks;
                    for (int e = 0; e < 8; ++e) {
                        const int k = 32 * ks + 8 * kq + e;
                        for (int plane = 0; plane < 2; ++plane) {
                            const int r = (plane * 16 + f) * 4 + (e >> 1);
                            unsigned &dst = regs[((size_t)(wave * 32 + (r >> 2)) * 64 + lane) * 4 + (r & 3)];
                            const unsigned bits = piece(gate, unit, k, plane);
                            dst = (e & 1) ? ((dst & 0x0000ffffu) | (bits << 16)) : ((dst & 0xffff0000u) | bits);
                        }
                        p2[((size_t)(wave * 16 + f) * 64 + lane) * 8 + e] = piece(gate, unit, k, 2);
                    }
                }
        }
    return finite;
}

// P2 (the third plane of pack_whh16h's split) as bf8 for v_mfma_scale_f32_16x16x128_f8f6f4: [wave 8][rb 4][lane 64][32 bytes] (the 16-byte
// halves of lanes 16 .. 31 and 48 .. 63 swapped: see the kernel's p2q_swap); lane
// (row = lane & 15, kq = lane >> 4) holds row -> (unit 16 wave + 4 rb + (row >> 2), gate row & 3), byte jj -> column k' = 32 kq + jj, where k'
// names the SOURCE unit 16 (k' >> 4) + 4 (k' & 3) + ((k' >> 2) & 3) (the kernel's fp8 image of h keeps a lane's four cells in one dword).
// E5M2 is the upper byte of an f16, so an element is exactly representable iff the low byte of (P2 x 2^13 as f16) is zero: P2 is 0, +-1 or
// +-2 units of 2^(e - 12) (e = the weight's exponent after scaling, <= 13), i.e. <= 4 in magnitude and at most two significant bits.
bool pack_whh16h_p2q(const float *w_hh, unsigned short *p2q, int *scale) {
    const int H = 128, SHIFT = 13;
    float amax = 0.0f;
    for (size_t i = 0; i < (size_t)4 * H * H; ++i) {
        const float v = __builtin_fabsf(w_hh[i]);
        if (!(v <= 3.0e38f)) return false;
        if (v > amax) amax = v;
    }
    int S = 0;
    if (amax > 0.0f) {
        int e;
        (void)__builtin_frexpf(amax, &e);
        S = 14 - e;
        if (S > 100) S = 100;
        if (S < -100) S = -100;
    }
    const float up = __builtin_ldexpf(1.0f, S);
    *scale = 127 - SHIFT;
    unsigned char *out = reinterpret_cast<unsigned char *>(p2q);
    bool exact = true;
    for (int wave = 0; wave < 8; ++wave)
        for (int rb = 0; rb < 4; ++rb)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = lane & 15, kq = lane >> 4;
                const int unit = 16 * wave + 4 * rb + (row >> 2), gate = row & 3;
                for (int jj = 0; jj < 32; ++jj) {
                    const int kp = 32 * kq + jj, src = 16 * (kp >> 4) + 4 * (kp & 3) + ((kp >> 2) & 3);
                    const float ws = w_hh[(size_t)(gate * H + unit) * H + src] * up;
                    const _Float16 p0 = (_Float16)ws;
                    const float t2 = (ws - (float)p0) * 2048.0f;
                    const _Float16 p1 = (_Float16)t2;
                    const _Float16 p2v = (_Float16)(t2 - (float)p1);
                    const _Float16 sh = (_Float16)((float)p2v * 8192.0f);
                    unsigned short bits;
                    __builtin_memcpy(&bits, &sh, 2);
                    if ((bits & 0xffu) != 0 || (float)sh != (float)p2v * 8192.0f) exact = false;
                    // (lanes 16 .. 31 and 48 .. 63: the two 16-byte halves swapped -- the kernel's conflict-free read order)
                    out[((size_t)((wave * 4 + rb) * 64 + lane)) * 32 + (jj ^ (16 * ((lane >> 4) & 1)))] = (unsigned char)(bits >> 8);
                }
            }
    return exact;
}

int lstm_auto_tile(int tiles, int dirs, int H, int n_cu) {
    if (H != 128) return 4;
    const int ncu = n_cu > 0 ? n_cu : 256;
    const long rounds4 = ((long)tiles * dirs + ncu - 1) / ncu, rounds16 = ((long)((tiles + 3) / 4) * dirs + ncu - 1) / ncu;
    return (double)rounds16 * REC16_ROUND_COST < (double)rounds4 ? 16 : 4;
}

hipError_t launch_lstm(const LstmArgs &a, hipStream_t s, int *tile_used) {
    if (tile_used) *tile_used = 0;
    if (a.tiles <= 0 || a.T <= 0) return hipSuccess;
    if (a.H != 128 && a.H != 64) {   // no register-resident form: the generic kernel (plain torch W_hh in Whh_packed)
        if (a.H < 4 || a.H % 4 || a.H > 1024 || a.tile_mode == 16 || !a.Whh_packed) return hipErrorInvalidValue;
        if (tile_used) *tile_used = 4;
        const size_t lds = (size_t)2 * SEQ_TILE * (a.H + 4) * sizeof(float);
        const dim3 grid(a.tiles, a.dirs);
        if (a.Y == nullptr) {
            if (!a.Yh || !a.Yl) return hipErrorInvalidValue;
            hipLaunchKernelGGL((lstm_rec_any_kernel<true>), grid, dim3(256), lds, s, a);
        } else {
            hipLaunchKernelGGL((lstm_rec_any_kernel<false>), grid, dim3(256), lds, s, a);
        }
        return hipGetLastError();
    }
    const bool can16 = a.H == 128 && a.Whh16h_regs && a.Whh16h_p2 && a.whh16h_scale && !a.h0 && !a.hN;
    if (a.tile_mode == 16 && !can16) return hipErrorInvalidValue;
    if (a.steps > 0 && (a.tile_mode != 4 || a.t_begin[0] < 0 || a.t_begin[0] + a.steps > a.T ||
                        (a.dirs > 1 && (a.t_begin[1] < 0 || a.t_begin[1] + a.steps > a.T))))
        return hipErrorInvalidValue;   // chunks exist for the 4-sequence form only
    const bool planes = a.Y == nullptr;
    if (planes && (!a.Yh || !a.Yl)) return hipErrorInvalidValue;
    const int pick = a.tile_mode ? a.tile_mode : lstm_auto_tile(a.tiles, a.dirs, a.H, a.n_cu);
    if (can16 && pick == 16) {
        if (tile_used) *tile_used = 16;
        const dim3 grid16((a.tiles + 3) / 4, a.dirs);
        if (a.products != 0 && a.products != 3 && a.products != 4) return hipErrorInvalidValue;
        const bool three = a.products == 3;
        const bool q8 = !three && a.Whh16h_p2q != nullptr;   // the P2 product on the 8-bit matrix pipe (weights verified exact as bf8)
        // 145 KiB with the f16 P2 image, 86 with the bf8 one, 17 without: one workgroup per CU either way (registers)
        const size_t lds = (size_t)((three ? 0 : q8 ? R16_P2Q_ELEMS : R16_P2_ELEMS) + R16_HB_ELEMS + (q8 ? R16_HQ_ELEMS : 0)) * sizeof(unsigned short);
        const void *fn = planes ? (three ? reinterpret_cast<const void *>(lstm_rec16h_kernel<true, 3, false>)
                                         : q8 ? reinterpret_cast<const void *>(lstm_rec16h_kernel<true, 4, true>) : reinterpret_cast<const void *>(lstm_rec16h_kernel<true, 4, false>))
                                : (three ? reinterpret_cast<const void *>(lstm_rec16h_kernel<false, 3, false>)
                                         : q8 ? reinterpret_cast<const void *>(lstm_rec16h_kernel<false, 4, true>) : reinterpret_cast<const void *>(lstm_rec16h_kernel<false, 4, false>));
        {   // the attribute belongs to the (function, device) pair and a process may own contexts on several GPUs (include/uvad.h), so it
            // is set for the CURRENT device on every launch (as launch_fbank / launch_sinc_conv do; no process-global "done" flag)
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        LstmArgs args = a;
        void *params[] = {&args};
        const hipError_t e = hipLaunchKernel(fn, grid16, dim3(512), params, lds, s);
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (tile_used) *tile_used = 4;
    const dim3 grid(a.tiles, a.dirs);
    if (a.H == 128 && planes)
        hipLaunchKernelGGL((lstm_rec_kernel<128, 8, true>), grid, dim3(512), 0, s, a);
    else if (a.H == 128)
        hipLaunchKernelGGL((lstm_rec_kernel<128, 8, false>), grid, dim3(512), 0, s, a);
    else if (a.H == 64 && planes)
        hipLaunchKernelGGL((lstm_rec_kernel<64, 4, true>), grid, dim3(256), 0, s, a);
    else if (a.H == 64)
        hipLaunchKernelGGL((lstm_rec_kernel<64, 4, false>), grid, dim3(256), 0, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace uvad


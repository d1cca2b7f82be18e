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
#ifdef UVAD_FAST_GATES   // diagnostic A/B (tools/err_probe.py): the round-1 forms
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-L2E * x));
#else
    // exponent clamped so that 1 + e stays finite (sigmoid(-87) = 1.6e-38 either way)
    const float e = __builtin_amdgcn_exp2f(__builtin_fminf(-L2E * x, 126.0f));
    return rcp_nr(1.0f + e);
#endif
}
__device__ __forceinline__ float tanh_f(float x) {
#ifdef UVAD_FAST_GATES
    return __builtin_fmaf(2.0f, sigmoid_f(2.0f * x), -1.0f);
#else
    const float e = __builtin_amdgcn_exp2f((-2.0f * L2E) * __builtin_fabsf(x));   // (0, 1]
    const float n = 1.0f - e, d = 1.0f + e;                                       // n exact for e >= 0.5 (Sterbenz)
    const float r = __builtin_amdgcn_rcpf(d);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);                           // correctly rounded n / d up to 2^-46
    return __builtin_copysignf(q, x);
#endif
}

// The cell update of one (unit, sequence) pair: g = gate pre-activations (i, f, g, o), c updated in place, returns h.
__device__ __forceinline__ float lstm_cell(const f32x4 g, float &c) {
#ifdef UVAD_ABL_NOGATE   // diagnostic build (tools/lstm_ablate.hip): cell update reduced to a few adds
    const float h = g[0] * 1e-3f + g[1] * 1e-3f;
    c = g[2] + g[3];
    return h;
#else
    const float ig = sigmoid_f(g[0]), fg = sigmoid_f(g[1]), gg = tanh_f(g[2]), og = sigmoid_f(g[3]);
    c = __builtin_fmaf(fg, c, ig * gg);
    return og * tanh_f(c);
#endif
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

    f32x4 gq[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
        const int sp = p < a.T ? p : a.T - 1;
        const int t = reverse ? a.T - 1 - sp : sp;
        gq_load(gq[p], g_ptr(t));
    }

    float hlast = 0.0f;
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
    for (int s0 = 0; s0 < a.T; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s >= a.T) break;   // wave-uniform
        const int t = reverse ? a.T - 1 - s : s;
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
#ifdef UVAD_ABL_NOLDSREAD   // diagnostic: no h reads at all (results meaningless)
#pragma unroll
        for (int kq = 0; kq < HR; ++kq) hv[kq] = make_float4(c, hlast, c * 0.5f, hlast * 0.5f);
#else
#pragma unroll
        for (int kq = 0; kq < HR; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(hb + 4 * kq);
#endif

        UVAD_STAMP_AT(3)   // [3] = barrier wait + loop top
        // 4 independent accumulation chains (k mod 4): dependent MFMAs are 4 issues apart
        f32x4 a0 = gq[u], a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < H / 4; ++kq) {
            const float4 hq = hv[kq % HR];
#ifndef UVAD_ABL_NOLDSREAD
            if (kq + HR < H / 4) hv[kq % HR] = *reinterpret_cast<const float4 *>(hb + 4 * (kq + HR));
#endif
#ifndef UVAD_ABL_NOMFMA
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 0], hq.x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 1], hq.y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 2], hq.z, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 3], hq.w, a3, 0, 0, 0);
#else
            a0[0] += w[4 * kq] * hq.x;   // keeps W and h alive without the matrix pipe
#endif
            if (kq == H / 8) {
                // refill this ring slot with the gates of step min(s + PD, T - 1) (branch-free: a
                // redundant re-load of the last row is harmless)
                const int sp = s + PD < a.T ? s + PD : a.T - 1;
                const int tp = reverse ? a.T - 1 - sp : sp;
                gq_load(gq[u], g_ptr(tp));
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
        UVAD_STAMP_AT(0)   // [0] = h reads + MFMA chains
        hlast = lstm_cell((a0 + a1) + (a2 + a3), c);
        hbuf[(s + 1) & 1][jb][unit] = hlast;
        if constexpr (PLANES) {
            // K-blocked plane (uvad_internal.h plane_index): the 4 sequence rows x 16 units of a wave are 128 contiguous bytes
#ifndef UVAD_ABL_NOGMEM
            const size_t R = rowu + (size_t)t * SEQ_TILE;
            const size_t yo = (R >> 7) * y_tile + y_wave + (R & 127) * 16 + y_lane;
            store_planes(a.Yh + yo, a.Yl + yo, hlast);
#endif
        } else {
            UVAD_YSTORE(a.Y + (row0 + (size_t)t * SEQ_TILE) * a.ldy + ycol, hlast);
        }
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
template <int H, bool PLANES>
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
    const int gcol = dir * 4 * H + ubase * 4;   // G is tile-blocked (g_index); row block rb is 16 columns further
    const size_t ycol = (size_t)dir * H + ubase;

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
            gq[p][rb] = *reinterpret_cast<const f32x4 *>(a.G + g_index(row0 + (size_t)t * SEQ_TILE, gcol + 16 * rb, a.ldg));
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

        // k-major over four independent accumulators (row blocks).  A variant with the row blocks one after the other and the
        // cell update of block rb staged between the MFMA groups of block rb + 1 measured the same 5.7 us per step at
        // B = 4096 (the kernel is bound by the f32 matrix pipe, 76 % busy), so the simple form stays.
        f32x4 acc[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = gq[u2][rb];
        {   // refill this ring slot with the gates of step min(s + PD16, T - 1)
            const int sp = s + PD16 < a.T ? s + PD16 : a.T - 1;
            const int tp = reverse ? a.T - 1 - sp : sp;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                gq[u2][rb] = *reinterpret_cast<const f32x4 *>(a.G + g_index(row0 + (size_t)tp * SEQ_TILE, gcol + 16 * rb, a.ldg));
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
        const size_t yrow = row0 + (size_t)t * SEQ_TILE;
        float hnew[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) hnew[rb] = lstm_cell(acc[rb], c[rb]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int u = ubase + 4 * rb;
            hbuf[(s + 1) & 1][u >> 5][j][u & 31] = hnew[rb];
            if (live) {
                if constexpr (PLANES) {
                    const size_t yo = plane_index(yrow, (int)ycol + 4 * rb, a.ldy);
                    store_planes(a.Yh + yo, a.Yl + yo, hnew[rb]);
                } else {
                    a.Y[yrow * a.ldy + ycol + 4 * rb] = hnew[rb];
                }
            }
        }
        __syncthreads();
      }
    }
}

}  // namespace

constexpr double REC16_ROUND_COST = 3.0;   // time of one round of the 16-sequence form / one round of the 4-sequence form

size_t whh_packed_elems(int H) { return (size_t)4 * H * H; }

int lstm_waves(int H) { return H == 128 ? 8 : 4; }

// register image of lstm_rec_kernel: [wave][kq = k/4][lane 64][4 k] with lane = (unit within the wave) * 4 + gate
void pack_whh(const float *w_hh, int H, float *out) {
    const int WAVES = lstm_waves(H);
    for (int wave = 0; wave < WAVES; ++wave)
        for (int kq = 0; kq < H / 4; ++kq)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int gate = lane & 3, unit = wave * 16 + (lane >> 2);
                    out[((size_t)(wave * (H / 4) + kq) * 64 + lane) * 4 + e] = w_hh[(size_t)(gate * H + unit) * H + 4 * kq + e];
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

// tile_mode 0: the form with the smaller estimated time.  One workgroup per CU at a time for both forms (registers), so a
// launch takes ceil(workgroups / CUs) rounds; a round of the 16-sequence form costs REC16_ROUND_COST rounds of the
// 4-sequence form (measured at B = 1024 ... 4096, profiles/README.md).  4 / 16 force either.  *tile_used reports the choice.
int lstm_auto_tile(int tiles, int dirs, int H, int n_cu) {
    if (H != 128) return 4;
    const int ncu = n_cu > 0 ? n_cu : 256;
    const long rounds4 = ((long)tiles * dirs + ncu - 1) / ncu, rounds16 = ((long)((tiles + 3) / 4) * dirs + ncu - 1) / ncu;
    return (double)rounds16 * REC16_ROUND_COST < (double)rounds4 ? 16 : 4;
}

hipError_t launch_lstm(const LstmArgs &a, hipStream_t s, int *tile_used) {
    if (tile_used) *tile_used = 0;
    if (a.tiles <= 0 || a.T <= 0) return hipSuccess;
    const bool can16 = a.H == 128 && a.Whh_packed16 && !a.h0 && !a.hN;
    if (a.tile_mode == 16 && !can16) return hipErrorInvalidValue;
    const bool planes = a.Y == nullptr;
    if (planes && (!a.Yh || !a.Yl)) return hipErrorInvalidValue;
    const bool pick16 = lstm_auto_tile(a.tiles, a.dirs, a.H, a.n_cu) == 16;
    if (can16 && (a.tile_mode == 16 || (a.tile_mode == 0 && pick16))) {
        if (tile_used) *tile_used = 16;
        const dim3 grid16((a.tiles + 3) / 4, a.dirs);
        if (planes) hipLaunchKernelGGL((lstm_rec16_kernel<128, true>), grid16, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((lstm_rec16_kernel<128, false>), grid16, dim3(512), 0, s, a);
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

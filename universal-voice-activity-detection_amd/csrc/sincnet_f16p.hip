// sincnet_f16p.hip -- the three convolution stages of the SincNet front end (reference: src/models/blocks/sincnet.py:44-103, called from
// PyanNet.forward, src/models/segmentation/PyanNet.py:174-177) on the f16 matrix cores with f32-equivalent arithmetic: the numerics of
// gemm_f16p.hip (weights scaled by a power of two and split into THREE f16 planes that add up to the f32 value exactly, activations
// into TWO planes = 22 bits, four v_mfma_f32_16x16x32_f16 products per term set in two f32 accumulators).  sincnet.hip keeps the exact
// f32 form (v_mfma_f32_32x32x2_f32) for GEMM modes 0 / 2 and for geometries this file does not take.
//
// Every stage is ONE strided-Hankel contraction  out[p][n] = sum_k W[n][k] * w[s * p + k]:
//   stage 1 (sinc bank 80 x 251, stride 10):  w = the normalised waveform, s = 10, K = 256 (taps 251 .. 255 carry zero weights);
//   stage 2 (Conv1d 80 -> 60, 5 taps):        w = the window stored position-major / channel-minor [x][80], s = 80, k = tap * 80 + channel,
//                                             K = 416 (13 k-steps; k >= 400 carry zero weights);
//   stage 3 (Conv1d 60 -> 60, 5 taps):        w = [x][80] rows holding 64 channels (60 real), k = tap * 64 + channel, K = 320 -- the byte
//                                             offset of a k-step is a compile-time immediate (tap * 160 + 64 * (ks & 1)).
// That is what the intermediate layout is chosen for: a stage writes its pooled output as [b][x][channel] (channel-minor), so the next
// stage's window is one contiguous run of HBM and an A fragment (8 consecutive k of one position) is one aligned ds_read_b128.
//
// Work split (256 threads = 4 waves, one per SIMD, one workgroup per CU, persistent over a contiguous range of (utterance, tile) pairs):
//   * WEIGHT-STATIONARY: wave w keeps the three planes of channel tile w (16 output channels) for the whole K in registers as the B
//     operands of the MFMAs (96 / 156 / 120 registers); stage 1 has 80 = 5 x 16 channels: every wave also holds tile 4 (+ 96 registers)
//     and runs it for a quarter of the tile's positions -- 1.25 units of work per SIMD, none idle, no padded channel.
//   * a tile = 192 conv positions = 64 pooled outputs, in 4 groups of 3 MFMA row blocks (48 positions).  The A operand's row -> position map
//     is free (every lane reads its own row from the LDS window), and it is chosen so that the accumulator layout of the 16x16 MFMA
//     (lane quarter q holds rows 4 q .. 4 q + 3) leaves each lane with 12 CONSECUTIVE positions of one channel after the three row blocks of a
//     group: bias, |.|, MaxPool1d(3) and the instance-norm statistics happen in registers, and a store instruction writes four 64-byte
//     runs of the channel-minor output.  No LDS round trip in the epilogue (sincnet.hip: 20 % of a tile).
//   * which 12 positions a lane quarter gets (quarter slot sigma(group, quarter) of the tile's 16) is picked so that every ds_read_b128 of
//     an A fragment is bank-conflict-free: stages 2 / 3 (row stride 160 bytes) with sigma = 4 g + q; stage 1 needs 16-byte-aligned
//     fragments of a stride-20-byte sequence, so its window is kept as FOUR copies shifted by 0 / 12 / 8 / 4 bytes (position p reads copy
//     p mod 4) at region offsets 0 / 160 / 80 / 0 bytes mod 256 with sigma from a search (tools/sinc_bank_search.py).
//   * statistics: per (tile, channel) -- per (tile, wave, channel) for stage 1's shared channel tile -- (count, sum, M2 about that set's own
//     mean) in f32, combined in a fixed order in double by norm_finalize_f16p_kernel (Chan's update): deterministic, independent of batch
//     neighbours and of scheduling, as in sincnet.hip.
#include "uvad_internal.h"

namespace uvad {

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TILE_POS = 192;     // conv positions per tile
constexpr int TILE_POOL = 64;     // pooled outputs per tile
constexpr int NGROUP = 4;         // groups of 3 row blocks (48 positions, 16 pooled) per tile

template <int ST> struct Stage;
template <> struct Stage<1> {
    static constexpr int KS = 8, NT = 5, CST = 80, ROWB = 20;                 // k-steps of 32, channel tiles of 16, output row width, bytes per position step
    static constexpr int REGION = 4608, LO_OFF = 4 * REGION, WIN = 2176;      // bytes per window copy, lo planes behind the four hi copies, samples per window
    static constexpr int LDS_BYTES = 8 * REGION;
    static constexpr unsigned long long SIGMA = 0xFDB9ECA875316420ull;        // sigma(g, q) = nibble 4 g + q: (0,2,4,6) (1,3,5,7) (8,10,12,14) (9,11,13,15)
    __device__ static constexpr int kimm(int ks) { return 64 * ks; }
};
template <> struct Stage<2> {
    static constexpr int KS = 13, NT = 4, CST = 64, ROWB = 160;
    static constexpr int ROWS = 197, LO_OFF = 31744, IN_CST = 80;             // window rows (192 + 4 taps + the zero-weight k >= 400 row), lo plane offset
    static constexpr int LDS_BYTES = 2 * LO_OFF + 2 * 80 * 4;                 // + the (scale, shift) table of the input norm
    static constexpr unsigned long long SIGMA = 0xFEDCBA9876543210ull;
    __device__ static constexpr int kimm(int ks) { return 64 * ks; }
};
template <> struct Stage<3> {
    static constexpr int KS = 10, NT = 4, CST = 64, ROWB = 160;
    static constexpr int ROWS = 196, LO_OFF = 31744, IN_CST = 64;
    static constexpr int LDS_BYTES = 2 * LO_OFF + 2 * 80 * 4;
    static constexpr unsigned long long SIGMA = 0xFEDCBA9876543210ull;
    __device__ static constexpr int kimm(int ks) { return 160 * (ks >> 1) + 64 * (ks & 1); }
};

__device__ __forceinline__ f16x8 lds_frag(const unsigned char *p) { return *reinterpret_cast<const f16x8 *>(p); }

// a ~= hi + lo * 2^-11 (22 bits), both planes round-to-nearest
__device__ __forceinline__ void split2(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

template <int ST>
__global__ __launch_bounds__(256, 1) void sinc_conv_f16p_kernel(SincF16Args a) {
    using S = Stage<ST>;
    constexpr int KS = S::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;            // accumulator layout: channel n of the wave's tile, row quarter q
    const int ar = lane & 15, akq = lane >> 4;         // A operand: MFMA row ar (quarter ar >> 2, position ar & 3 inside it), k quarter akq

    const long long total = (long long)a.B * a.ntiles;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long g_begin = per * blockIdx.x;
    const long long g_end = g_begin + per < total ? g_begin + per : total;
    if (g_begin >= g_end) return;

    // ---- the wave's weight fragments for the whole K: lane (n, kq) holds W[16 t + n][32 ks + 8 kq .. + 8] of each plane
    f16x8 w0[KS], w1[KS], w2[KS];
    f16x8 x0[ST == 1 ? KS : 1], x1[ST == 1 ? KS : 1], x2[ST == 1 ? KS : 1];   // stage 1: channel tile 4, shared by the four waves
    {
        const f16x8 *wf = reinterpret_cast<const f16x8 *>(a.Wfrag) + ((size_t)wave * 3 * KS) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            w0[ks] = wf[(size_t)(0 * KS + ks) * 64];
            w1[ks] = wf[(size_t)(1 * KS + ks) * 64];
            w2[ks] = wf[(size_t)(2 * KS + ks) * 64];
        }
        if constexpr (ST == 1) {
            const f16x8 *xf = reinterpret_cast<const f16x8 *>(a.Wfrag) + ((size_t)4 * 3 * KS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                x0[ks] = xf[(size_t)(0 * KS + ks) * 64];
                x1[ks] = xf[(size_t)(1 * KS + ks) * 64];
                x2[ks] = xf[(size_t)(2 * KS + ks) * 64];
            }
        }
    }
    const float bias_own = a.bias[16 * wave + n], bias_x = ST == 1 ? a.bias[64 + n] : 0.f;
    const float wscale = a.wscale;

    // ---- per-lane byte address of the A fragment of (group 0 slot 0, row block 0, k-step 0) in the hi plane / copy
    unsigned a_base;
    if constexpr (ST == 1) {
        const int j = ar & 3;   // the window copy of this row: position = 12 sigma + 4 b + j, so position mod 4 = j
        const int rj = j == 1 ? 160 : j == 2 ? 80 : 0, cb = (16 - 4 * j) & 15;
        a_base = (unsigned)(j * S::REGION + rj + cb + 20 * j + 16 * akq);
    } else {
        a_base = (unsigned)(S::ROWB * (ar & 3) + 16 * akq);
    }
    const int a_rho = ar >> 2;

    // ---- the window of a tile travels global -> registers (prefetch: issued BEFORE the previous tile's MFMAs, so HBM / L2 latency hides
    //      under them) -> LDS (stage: norm, leaky_relu and the f16 split, after them).  Stage 1: chunks of 8 samples, thread t takes chunk t
    //      and, for t < 16, chunk 256 + t.  Stages 2 / 3: 16-byte units of the contiguous [row][channel] window, RPP whole rows per pass: thread
    //      t < THR owns the SAME four channels c4 = t mod U in every pass (its (scale, shift) are read from the LDS table once per tile, not once
    //      per unit: 32 dependent LDS reads and their latency per tile were the largest part of a 5 300-cycle staging) and rows t / U + RPP i.
    constexpr int SIN_CST = Stage<ST == 1 ? 2 : ST>::IN_CST, SU = SIN_CST / 4, RPP = 256 / SU, THR = RPP * SU, SROWS = Stage<ST == 1 ? 2 : ST>::ROWS;
    constexpr int NPRE = ST == 1 ? 4 : (SROWS + RPP - 1) / RPP;
    float4 pre[NPRE];
    // Stages 2 / 3: where the window of tile gi starts and how many of its rows exist; pf_load(i) fetches this thread's unit of pass i.  Every
    // load is UNCONDITIONAL (a thread past THR, a row past the window or past the utterance re-reads unit 0: stage() ignores it / stages zeros),
    // so a load is one address select and one instruction, and the loads of the NEXT tile can be issued one at a time between the k-steps of the
    // current one (pf_slot): issued as one burst in front of the MFMAs they held the wave for 2 100 - 2 600 cycles per tile -- the CU's vector
    // memory path takes 17 x 4 KiB from four waves at once -- with the matrix pipe idle.
    const float *pf_src = a.in;
    int pf_rows = 0;
    auto pf_setup = [&](long long gi) __attribute__((always_inline)) {
        const int b = (int)(gi / a.ntiles), tile = (int)(gi - (long long)b * a.ntiles);
        const long long x0s = (long long)tile * TILE_POS, left = (long long)a.Lin - x0s;
        pf_src = a.in + ((size_t)b * a.Lin + x0s) * (ST == 1 ? 1 : SIN_CST);
        pf_rows = left < SROWS ? (int)left : SROWS;
    };
    auto pf_load = [&](int i) __attribute__((always_inline)) {
        const int r0 = tid / SU;
        const bool ok = tid < THR && r0 + RPP * i < pf_rows;
        pre[i] = *reinterpret_cast<const float4 *>(pf_src + (size_t)(ok ? (tid + THR * i) * 4 : 0));
    };
    // the k-step slot (group * (KS - 1) + ks - 1, ks = 1 .. KS - 1) in which unit i of the next tile is fetched: spread evenly over the tile
    auto pf_slot = [](int i) { return (i * NGROUP * (KS - 1)) / NPRE; };
    auto prefetch = [&](long long gi) __attribute__((always_inline)) {
        const int b = (int)(gi / a.ntiles), tile = (int)(gi - (long long)b * a.ntiles);
        if constexpr (ST == 1) {
            const float *src = a.in + (size_t)b * a.in_bstride;
            const long long x0s = (long long)tile * (TILE_POS * 10);
            const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;   // (the tile origin is a multiple of 1920 samples)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = tid + 256 * r;
                const long long x = x0s + 8 * c;
                float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo4;
                if (c < S::WIN / 8) {
                    if (vec && x + 8 <= a.Lin) {
                        lo4 = *reinterpret_cast<const float4 *>(src + x);
                        hi4 = *reinterpret_cast<const float4 *>(src + x + 4);
                    } else {
                        float e[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) e[k] = x + k < a.Lin ? src[x + k] : __builtin_nanf("");   // NaN marks "past the end": staged as zero
                        lo4 = make_float4(e[0], e[1], e[2], e[3]);
                        hi4 = make_float4(e[4], e[5], e[6], e[7]);
                    }
                }
                pre[2 * r] = lo4;
                pre[2 * r + 1] = hi4;
            }
        } else {
            pf_setup(gi);
#pragma unroll
            for (int i = 0; i < NPRE; ++i) pf_load(i);
        }
    };
    auto stage = [&](int b, int tile) __attribute__((always_inline)) {
        if constexpr (ST == 1) {
            const float sc = a.in_scale[b], sh = a.in_shift[b];
            const long long x0s = (long long)tile * (TILE_POS * 10);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = tid + 256 * r;
                if (c < S::WIN / 8) {
                    const float e[8] = {pre[2 * r].x, pre[2 * r].y, pre[2 * r].z, pre[2 * r].w, pre[2 * r + 1].x, pre[2 * r + 1].y, pre[2 * r + 1].z, pre[2 * r + 1].w};
                    _Float16 h[8], l[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float v = x0s + 8 * c + k < a.Lin ? __builtin_fmaf(e[k], sc, sh) : 0.f;
                        split2(v, h[k], l[k]);
                    }
                    uint4 ph, pl;
                    __builtin_memcpy(&ph, h, 16);
                    __builtin_memcpy(&pl, l, 16);
                    unsigned char *d = smem + 16 * c;
                    // copy 0: aligned; copy 1: +160 + 12; copy 2: +80 + 8; copy 3: +0 + 4 (4-, 8- and 4-byte aligned)
                    *reinterpret_cast<uint4 *>(d) = ph;
                    *reinterpret_cast<uint4 *>(d + S::LO_OFF) = pl;
                    unsigned *p1 = reinterpret_cast<unsigned *>(d + 1 * S::REGION + 160 + 12), *p3 = reinterpret_cast<unsigned *>(d + 3 * S::REGION + 4);
                    unsigned *q1 = reinterpret_cast<unsigned *>(d + 1 * S::REGION + 160 + 12 + S::LO_OFF), *q3 = reinterpret_cast<unsigned *>(d + 3 * S::REGION + 4 + S::LO_OFF);
                    p1[0] = ph.x; p1[1] = ph.y; p1[2] = ph.z; p1[3] = ph.w;
                    q1[0] = pl.x; q1[1] = pl.y; q1[2] = pl.z; q1[3] = pl.w;
                    p3[0] = ph.x; p3[1] = ph.y; p3[2] = ph.z; p3[3] = ph.w;
                    q3[0] = pl.x; q3[1] = pl.y; q3[2] = pl.z; q3[3] = pl.w;
                    uint2 *p2 = reinterpret_cast<uint2 *>(d + 2 * S::REGION + 80 + 8), *q2 = reinterpret_cast<uint2 *>(d + 2 * S::REGION + 80 + 8 + S::LO_OFF);
                    p2[0] = make_uint2(ph.x, ph.y); p2[1] = make_uint2(ph.z, ph.w);
                    q2[0] = make_uint2(pl.x, pl.y); q2[1] = make_uint2(pl.z, pl.w);
                }
            }
        } else {
            const float *nrm = reinterpret_cast<const float *>(smem + 2 * S::LO_OFF);
            const int r0 = tid / SU, c4 = tid - r0 * SU;
            const float4 sc = *reinterpret_cast<const float4 *>(nrm + 4 * c4), sh = *reinterpret_cast<const float4 *>(nrm + 80 + 4 * c4);   // (c4 < 20: inside the table for every thread)
            unsigned char *d = smem + r0 * S::ROWB + 8 * c4;
            const long long left = (long long)a.Lin - (long long)tile * TILE_POS;
            const int rows = left < SROWS ? (int)left : SROWS;                   // rows of this window that exist (the others: the unit-0 re-read, staged as zero)
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                if (tid < THR && (i + 1 < NPRE || r0 + RPP * i < SROWS)) {
                    float4 v = pre[i];
                    if (rows < SROWS && r0 + RPP * i >= rows) v = make_float4(0.f, 0.f, 0.f, 0.f);   // (the utterance's last tile only)
                    const float e[4] = {__builtin_fmaf(v.x, sc.x, sh.x), __builtin_fmaf(v.y, sc.y, sh.y), __builtin_fmaf(v.z, sc.z, sh.z), __builtin_fmaf(v.w, sc.w, sh.w)};
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = __builtin_fmaxf(e[k], e[k] * a.slope);   // leaky_relu of the previous stage (0 <= slope < 1; NaN stays NaN)
                        split2(t, h[k], l[k]);
                    }
                    uint2 ph, pl;
                    __builtin_memcpy(&ph, h, 8);
                    __builtin_memcpy(&pl, l, 8);
                    *reinterpret_cast<uint2 *>(d + i * (RPP * S::ROWB)) = ph;
                    *reinterpret_cast<uint2 *>(d + i * (RPP * S::ROWB) + S::LO_OFF) = pl;
                }
            }
        }
    };

    int cur_b = -1;
    prefetch(g_begin);
    for (long long gi = g_begin; gi < g_end; ++gi) {
        const int b = (int)(gi / a.ntiles), tile = (int)(gi - (long long)b * a.ntiles);
        __syncthreads();   // the previous tile's fragment reads are complete
        if constexpr (ST != 1) {
            if (b != cur_b) {   // [80] scale, [80] shift of this utterance's input norm (0, 0 past the real channels)
                float *nrm = reinterpret_cast<float *>(smem + 2 * S::LO_OFF);
                cur_b = b;
                if (tid < 160) {
                    const int c = tid < 80 ? tid : tid - 80;
                    const float *t = tid < 80 ? a.in_scale : a.in_shift;
                    nrm[tid] = c < a.n_in ? t[(size_t)b * a.n_in + c] : 0.f;
                }
                __syncthreads();
            }
        }
        stage(b, tile);
        __syncthreads();
        if constexpr (ST == 1) {
            if (gi + 1 < g_end) prefetch(gi + 1);
        } else {
            pf_setup(gi + 1 < g_end ? gi + 1 : gi);   // (the range's last tile re-reads its own window: no branch in the MFMA stream)
        }

        // ---------------- one group (48 positions) against one channel tile: 3 row blocks x KS k-steps x 4 products, then the epilogue.
        // The A fragments of k-step ks + 1 are read while the twelve MFMAs of k-step ks run (two fragment sets, the order pinned with
        // sched_barrier), and the last k-step reads the first fragments of the NEXT group (nh / nl), whose latency then hides under this
        // group's epilogue.
        auto frag_ptr = [&](int g) __attribute__((always_inline)) {
            const int sig_a = (int)((S::SIGMA >> (16 * g + 4 * a_rho)) & 15ull);
            return smem + a_base + (unsigned)(12 * S::ROWB) * (unsigned)sig_a;
        };
        f16x8 nh[3], nl[3];
        auto load_first = [&](const unsigned char *ap) __attribute__((always_inline)) {
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                nh[rb] = lds_frag(ap + 4 * S::ROWB * rb + S::kimm(0));
                nl[rb] = lds_frag(ap + 4 * S::ROWB * rb + S::kimm(0) + S::LO_OFF);
            }
        };
        // pooled outputs of this tile: whole (every one of its 64 below Lpool: plain stores, plain sums) or the utterance's last, ragged one
        const int valid_tile = a.Lpool - tile * TILE_POOL < TILE_POOL ? a.Lpool - tile * TILE_POOL : TILE_POOL;
        const bool full = valid_tile == TILE_POOL;
        float *out_tile = a.out + ((size_t)b * a.Lpool + (size_t)tile * TILE_POOL) * S::CST;
        auto run_group = [&](int g, const unsigned char *ap, const unsigned char *ap_next, const f16x8(&p0)[KS], const f16x8(&p1)[KS], const f16x8(&p2)[KS],
                             float bias, int ctile, float(&m)[4]) __attribute__((always_inline)) {
            f32x4 hi[3], lo[3];
            f16x8 fh[2][3], fl[2][3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                hi[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                fh[0][rb] = nh[rb]; fl[0][rb] = nl[rb];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cs = ks & 1, ns = cs ^ 1;
                // the six reads of the NEXT k-step (of the next group at the end) are issued first ...
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) {
                    if (ks + 1 < KS) {
                        fh[ns][rb] = lds_frag(ap + 4 * S::ROWB * rb + S::kimm(ks + 1 < KS ? ks + 1 : 0));
                        fl[ns][rb] = lds_frag(ap + 4 * S::ROWB * rb + S::kimm(ks + 1 < KS ? ks + 1 : 0) + S::LO_OFF);
                    } else {
                        nh[rb] = lds_frag(ap_next + 4 * S::ROWB * rb + S::kimm(0));
                        nl[rb] = lds_frag(ap_next + 4 * S::ROWB * rb + S::kimm(0) + S::LO_OFF);
                    }
                }
                // ... beside this k-step's twelve MFMAs, which use fragments read one k-step ago: "two MFMAs, one read" six times (the k-step is
                // fenced, so the reads the scheduler can pick are the next k-step's; left alone hipcc sinks every read to just in front of its use)
                // (product-major order: the three products of one `lo` accumulator are two other MFMAs apart)
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) hi[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cs][rb], p0[ks], hi[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) lo[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cs][rb], p1[ks], lo[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) lo[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cs][rb], p0[ks], lo[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) lo[rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cs][rb], p2[ks], lo[rb], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST != 1) {   // this k-step's share of the next tile's window (own channel tiles only: ctile < NGROUP)
                    if (ctile < NGROUP && ks >= 1) {
#pragma unroll
                        for (int i = 0; i < NPRE; ++i)
                            if (pf_slot(i) == g * (KS - 1) + ks - 1) pf_load(i);
                    }
                }
            }
            // lane (n, q): positions 12 sigma(g, q) + 4 rb + i, i = 0 .. 3.  u = hi + lo * 2^-11; the stage's value is u * 2^-S + bias (stage 1: |u| * 2^-S,
            // the sinc bank has no bias), a non-decreasing map of u (of |u|), so MaxPool1d(3) is taken on u and the affine map applied to the
            // four maxima: the same bits as pooling the mapped values (rounding is monotone), 8 vector instructions fewer per group.
            float v[12];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = __builtin_fmaf(lo[rb][i], 0.00048828125f, hi[rb][i]);
                    v[4 * rb + i] = ST == 1 ? __builtin_fabsf(u) : u;
                }
            const int sig_q = (int)((S::SIGMA >> (16 * g + 4 * q)) & 15ull);
            float *o = out_tile + (4 * sig_q) * S::CST + 16 * ctile + n;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mx = __builtin_fmaxf(__builtin_fmaxf(v[3 * j], v[3 * j + 1]), v[3 * j + 2]);
                m[j] = ST == 1 ? mx * wscale : __builtin_fmaf(mx, wscale, bias);
            }
            if (full) {   // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j * S::CST] = m[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * sig_q + j < valid_tile) o[j * S::CST] = m[j];
            }
            return sig_q;
        };
        // statistics of `cnt` pooled outputs of one channel held by the four lane quarters (vals[k] valid where ok[k]): (count, sum, M2
        // about their own mean) -> partial slot `part` of this tile.  The mean uses v_rcp_f32: M2 about a value 1 ulp off the true mean
        // differs from the true M2 in the second order only, and norm_finalize_f16p_kernel recomputes the mean from (sum, count) in double.
        auto put_stats = [&](float sl, int cnt, auto &&sq_dev, int part, int ctile) __attribute__((always_inline)) {
            float s = sl + __shfl_xor(sl, 16);
            s += __shfl_xor(s, 32);
            const float mean = cnt > 0 ? s * __builtin_amdgcn_rcpf((float)cnt) : 0.f;
            float d2 = sq_dev(mean);
            d2 += __shfl_xor(d2, 16);
            d2 += __shfl_xor(d2, 32);
            if (q == 0) {
                float *pp = a.partials + ((((size_t)b * a.ntiles + tile) * NGROUP + part) * 3) * S::CST + 16 * ctile + n;
                pp[0] = (float)cnt;
                pp[S::CST] = s;
                pp[2 * S::CST] = d2;
            }
        };
        // the weight fragments stay where the MFMAs read them (AGPRs): a constraint, no instruction
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" : "+a"(w0[ks]), "+a"(w1[ks]), "+a"(w2[ks]));
            if constexpr (ST == 1) asm volatile("" : "+a"(x0[ks]), "+a"(x1[ks]), "+a"(x2[ks]));
        }
        load_first(frag_ptr(0));
        float pm[NGROUP][4];
        int sq[NGROUP];
#pragma unroll
        for (int g = 0; g < NGROUP; ++g) {
            // what the last k-step prefetches: the next group's first fragments (stage 1: then the shared tile's group; at the very end a
            // harmless re-read of this group's)
            const int gn = g + 1 < NGROUP ? g + 1 : ST == 1 ? wave : g;
            sq[g] = run_group(g, frag_ptr(g), frag_ptr(gn), w0, w1, w2, bias_own, wave, pm[g]);
        }
        {   // the wave's own 16 channels: one statistics partial per TILE (slot 0): a lane holds 16 of the tile's 64 pooled outputs
            float sl = 0.f;
#pragma unroll
            for (int g = 0; g < NGROUP; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) sl += (full || 4 * sq[g] + j < valid_tile) ? pm[g][j] : 0.f;
            put_stats(sl, valid_tile, [&](float mean) __attribute__((always_inline)) {
                float d2 = 0.f;
#pragma unroll
                for (int g = 0; g < NGROUP; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = pm[g][j] - mean;
                        d2 += (full || 4 * sq[g] + j < valid_tile) ? d * d : 0.f;
                    }
                return d2;
            }, 0, wave);
        }
        if constexpr (ST == 1) {   // channel tile 4, this wave's quarter of the positions: one partial per (tile, wave)
            float xm[4];
            const int sx = run_group(wave, frag_ptr(wave), frag_ptr(wave), x0, x1, x2, bias_x, 4, xm);
            int cnt = 0;   // (wave-uniform: the four quarter slots of group `wave`)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int left = valid_tile - 4 * (int)((S::SIGMA >> (16 * wave + 4 * qq)) & 15ull);
                cnt += left < 0 ? 0 : left > 4 ? 4 : left;
            }
            float sl = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sl += (full || 4 * sx + j < valid_tile) ? xm[j] : 0.f;
            put_stats(sl, cnt, [&](float mean) __attribute__((always_inline)) {
                float d2 = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = xm[j] - mean;
                    d2 += (full || 4 * sx + j < valid_tile) ? d * d : 0.f;
                }
                return d2;
            }, wave, 4);
        }
    }
}

// per (tile, group) (count, sum, M2 about the group mean) partials -> per (b, c) affine of the instance norm:
//   y = (x - mean) / sqrt(var + eps) * gamma + beta = x * scale + shift   (biased variance, torch InstanceNorm1d)
// One workgroup per utterance; the partials of a channel are cut into NCH contiguous chunks, each combined in order by one thread (Chan's
// update, double), and thread 0 of the channel combines the chunk results in chunk order: a fixed tree, so the result does not depend on
// batch neighbours or scheduling.  (One thread per channel walking all 4 x ntiles partials took 0.14 ms of dependent loads per launch.)
constexpr int FIN_NCH = 12, FIN_CP = 80;
// Channels below `split_from` have ONE partial per tile (slot 0 of its NGROUP slots), the others (stage 1's shared channel tile) NGROUP.
__global__ __launch_bounds__(FIN_NCH * FIN_CP) void norm_finalize_f16p_kernel(const float *partials, int ntiles, int split_from, int CST, int C, int L,
                                                                           const float *gamma, const float *beta, float eps, float *scale, float *shift) {
    __shared__ double red[FIN_NCH][3][FIN_CP];
    const int b = blockIdx.x, n = threadIdx.x % FIN_CP, ch = threadIdx.x / FIN_CP;
    const int slots = n >= split_from ? NGROUP : 1;
    const int per = (ntiles + FIN_NCH - 1) / FIN_NCH, t0 = ch * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    double mean = 0.0, M2 = 0.0, cnt = 0.0;
    if (n < C) {
        for (int t = t0; t < t1; ++t) {
            const float *pp = partials + ((size_t)b * ntiles + t) * NGROUP * 3 * CST + n;
            for (int k = 0; k < slots; ++k, pp += 3 * CST) {
                const double nt = (double)pp[0];
                if (nt <= 0.0) continue;
                const double mt = (double)pp[CST] / nt, delta = mt - mean, tot = cnt + nt;
                M2 += (double)pp[2 * CST] + delta * delta * cnt * nt / tot;
                mean += delta * nt / tot;
                cnt = tot;
            }
        }
    }
    red[ch][0][n] = cnt; red[ch][1][n] = mean; red[ch][2][n] = M2;
    __syncthreads();
    if (ch != 0 || n >= C) return;
    for (int k = 1; k < FIN_NCH; ++k) {
        const double nt = red[k][0][n];
        if (nt <= 0.0) continue;
        const double delta = red[k][1][n] - mean, tot = cnt + nt;
        M2 += red[k][2][n] + delta * delta * cnt * nt / tot;
        mean += delta * nt / tot;
        cnt = tot;
    }
    const double var = M2 / (double)L;
    const double sc = (double)gamma[n] / sqrt(var + (double)eps);
    scale[(size_t)b * C + n] = (float)sc;
    shift[(size_t)b * C + n] = (float)((double)beta[n] - mean * sc);
}

// last norm + leaky_relu; the stage output is already "batch frames feature" (PyanNet.py:179), CST floats per row
__global__ __launch_bounds__(256) void sinc_out_f16p_kernel(const float *P, const float *scale, const float *shift, int B, int C, int CST, int L, float slope,
                                                            float *feats, int ldf) {
    const long long total = (long long)B * L * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const int b = (int)(bt / L);
        float v = __builtin_fmaf(P[(size_t)bt * CST + c], scale[(size_t)b * C + c], shift[(size_t)b * C + c]);
        v = v >= 0.f ? v : v * slope;
        feats[(size_t)bt * ldf + c] = v;
    }
}

}  // namespace

int sinc_f16p_ksteps(int stage) { return stage == 0 ? Stage<1>::KS : stage == 1 ? Stage<2>::KS : Stage<3>::KS; }
int sinc_f16p_cst(int stage) { return stage == 0 ? Stage<1>::CST : stage == 1 ? Stage<2>::CST : Stage<3>::CST; }
int sinc_f16p_tiles16(int stage) { return stage == 0 ? Stage<1>::NT : stage == 1 ? Stage<2>::NT : Stage<3>::NT; }
int sinc_f16p_ntiles(long long Lpool) { return (int)((Lpool + TILE_POOL - 1) / TILE_POOL); }
size_t sinc_f16p_partial_floats(int stage, int B, int ntiles) { return (size_t)B * ntiles * NGROUP * 3 * sinc_f16p_cst(stage); }
size_t sinc_f16p_wfrag_elems(int stage) { return (size_t)sinc_f16p_tiles16(stage) * 3 * sinc_f16p_ksteps(stage) * 64 * 8; }

// The reference's geometry and what is a change of size only (fewer filters / channels): sinc bank of <= 80 filters x <= 256 taps, stride 10;
// Conv1d(80 -> <= 64, 5 taps); Conv1d(<= 64 -> <= 64, 5 taps).  Everything else runs the exact-f32 kernels of sincnet.hip.
bool sinc_f16p_supported(int n_filters, int kernel_size, int stride, int c2, int k2, int c3, int k3) {
    return stride == 10 && kernel_size <= 256 && n_filters == 80 && k2 == 5 && c2 <= 64 && (c2 & 3) == 0 && k3 == 5 && c3 <= 64;
}

// W[n][k] (row-major, ldk floats per row, K order of the stage as in the header comment; n < nrows) -> the B-operand register image
// [16-channel tile][plane][k-step][lane][8 f16] of w * 2^S split into three exact f16 planes; *wscale = 2^-S.  false: a weight is not finite.
bool sinc_f16p_pack_weights(int stage, const float *w, int nrows, int ldk, unsigned short *out, float *wscale) {
    const int KS = sinc_f16p_ksteps(stage), NT = sinc_f16p_tiles16(stage);
    float amax = 0.0f;
    bool finite = true;
    for (size_t i = 0; i < (size_t)nrows * ldk; ++i) {
        const float v = __builtin_fabsf(w[i]);
        if (!(v <= 3.0e38f)) finite = false;
        if (v > amax) amax = v;
    }
    int S = 0;
    if (finite && amax > 0.0f) {
        int e;
        (void)__builtin_frexpf(amax, &e);
        S = 14 - e;               // amax * 2^S in [2^13, 2^14): no piece that matters is an f16 subnormal (gemm_f16p.hip)
        if (S > 100) S = 100;
        if (S < -100) S = -100;
    }
    const float up = __builtin_ldexpf(1.0f, S);
    *wscale = __builtin_ldexpf(1.0f, -S);
    const size_t total = sinc_f16p_wfrag_elems(stage);
    for (size_t i = 0; i < total; ++i) out[i] = 0;
    for (int t = 0; t < NT; ++t)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 16 * t + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
                    if (n >= nrows || k >= ldk || !finite) continue;
                    const float ws = w[(size_t)n * ldk + k] * up;               // exact (power of two)
                    const _Float16 p0 = (_Float16)ws;
                    const float t2 = (ws - (float)p0) * 2048.0f;                // exact
                    const _Float16 p1 = (_Float16)t2;
                    const _Float16 p2 = (_Float16)(t2 - (float)p1);             // exact: at most 24 - 22 significant bits are left
                    const _Float16 pl[3] = {p0, p1, p2};
                    for (int pi = 0; pi < 3; ++pi) {
                        const size_t o = ((((size_t)t * 3 + pi) * KS + ks) * 64 + lane) * 8 + j;
                        __builtin_memcpy(&out[o], &pl[pi], 2);
                    }
                }
    return finite;
}

hipError_t launch_sinc_conv_f16p(int stage, const SincF16Args &a, hipStream_t s) {
    const long long total = (long long)a.B * a.ntiles;
    if (total <= 0) return hipSuccess;
    const int ncu = a.n_cu > 0 ? a.n_cu : 256;
    const dim3 grid((unsigned)(total < ncu ? total : ncu)), block(256);
#define UVAD_SF_LAUNCH(ST_)                                                                                                          \
    {                                                                                                                                \
        auto k = sinc_conv_f16p_kernel<ST_>;                                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Stage<ST_>::LDS_BYTES); \
        if (e != hipSuccess) return e;                                                                                               \
        hipLaunchKernelGGL(k, grid, block, Stage<ST_>::LDS_BYTES, s, a);                                                             \
    }
    if (stage == 0) UVAD_SF_LAUNCH(1) else if (stage == 1) UVAD_SF_LAUNCH(2) else if (stage == 2) UVAD_SF_LAUNCH(3) else return hipErrorInvalidValue;
#undef UVAD_SF_LAUNCH
    return hipGetLastError();
}

hipError_t launch_norm_finalize_f16p(int stage, const float *partials, int B, int ntiles, int C, int L, const float *gamma, const float *beta, float eps,
                                     float *scale, float *shift, hipStream_t s) {
    const int CST = sinc_f16p_cst(stage), split_from = stage == 0 ? 64 : CST;
    hipLaunchKernelGGL(norm_finalize_f16p_kernel, dim3(B), dim3(FIN_NCH * FIN_CP), 0, s, partials, ntiles, split_from, CST, C, L, gamma, beta, eps, scale, shift);
    return hipGetLastError();
}

hipError_t launch_sinc_out_f16p(const float *P, const float *scale, const float *shift, int B, int C, int CST, int L, float slope, float *feats, int ldf,
                                hipStream_t s) {
    const long long n = (long long)B * L * C;
    if (n <= 0) return hipSuccess;
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(sinc_out_f16p_kernel, dim3((int)g), dim3(256), 0, s, P, scale, shift, B, C, CST, L, slope, feats, ldf);
    return hipGetLastError();
}

}  // namespace uvad

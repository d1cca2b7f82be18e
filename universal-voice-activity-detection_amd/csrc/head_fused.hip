// head_fused.hip -- the feed-forward stack and the classifier of PyanNet2.forward in ONE kernel:
//     z1 = leaky_relu(y W1^T + b1);  z2 = leaky_relu(z1 W2^T + b2);  logit = z2 . w + b;  prob = sigmoid(logit)
//   (reference: for linear in self.linear: outputs = F.leaky_relu(linear(outputs)); self.activation(self.classifier(outputs)),
//    src/models/segmentation/PyanNet2.py:183-187)
// for the default head (two feed-forward layers of 128 units, H x directions = 256 or 128 inputs) in split-f16 GEMM mode.
//
// Why.  As three launches (two gemm_f16p_kernel + classifier_kernel) the head moves ~800 MB through HBM per cfg-2 step for 0.1
// PFLOP-equivalent of matrix work: it reads the LSTM output planes (262 MB), writes and re-reads the planes of z1 (131 + 131 MB),
// writes z2 as f32 (131 MB) and reads it again for a 128-term dot product -- 0.19 ms, bound by those round trips.  Here a workgroup
// keeps BOTH weight matrices in registers (the weight-stationary scheme of gemm_f16p_ws.hip), streams 64-row tiles of the LSTM
// output planes through an LDS-DMA ring, hands z1 to the second contraction through LDS (as the two f16 planes the split-f16
// arithmetic wants) and reduces z2 against the classifier row in registers: 262 MB in, 2 MB out.
//
// Numerics: the two contractions issue the same MFMA products in the same order per accumulator as gemm_f16p_kernel (three exact f16
// weight planes x two activation planes, hi / lo accumulator sets), so z1 and z2 are bit-identical to the unfused path; the final
// dot product is summed in a different order than classifier_kernel's (16 columns per lane, then 2 x 4 partial sums), i.e. equal
// to rounding of an f32 sum of 128 terms.
//
// Structure (one wave per SIMD, everything else is gemm_f16p_ws.hip's):
//   * 4 waves; wave w owns output columns [32 w, 32 w + 32) of BOTH layers: W1 fragments 3 x K1/16 x 4 registers (192 at K1 = 256,
//     in AGPRs), W2 fragments 3 x 8 x 4 = 96;
//   * tile = 64 rows (two 32-row MFMA blocks: hi / lo accumulators = 64 registers); a pipeline step = two 16-deep k-blocks = 8 KiB
//     of the hi and lo planes by LDS-DMA (two 1 KiB pieces per wave), ring of 8 steps in flight across barriers and tiles, counted
//     vmcnt, raw s_barrier, fragment reads one step ahead by inline asm with hand-counted lgkmcnt;
//   * MFMAs with the WEIGHT fragment as the A operand (transposed 32 x 32 blocks): a lane holds 4 consecutive columns of one row, which
//     is what the LDS image of z1 (row-major k-blocks, the A operand of layer 2) and the in-lane classifier dot product want;
//   * persistent workgroups pull tiles from one counter, two tiles ahead (a launch that finds part of the chip busy still balances).
// Every LDS access is a 32- or 128-bit operation (see fbank.hip on 64-bit LDS operations beside MFMA neighbours).
#include "uvad_internal.h"

#include <type_traits>
#include <utility>

namespace uvad {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int TM = 64;                 // rows per tile
constexpr int HSLAB = TM * 16;         // f16 elements of one k-block of one plane (2 KiB)
constexpr int HSTEP = 4 * HSLAB;       // a pipeline step: k-block 2 s (hi, lo), k-block 2 s + 1 (hi, lo) = 8 KiB
constexpr int HNST = 8;                // ring depth (steps)
constexpr int LH = 128;                // units of both feed-forward layers

template <class F, int... I> __device__ __forceinline__ void hf_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void hf_static_for(F &&f) { hf_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int N> __device__ __forceinline__ void hf_wait_vm_lgkm0() {   // s_waitcnt vmcnt(N) lgkmcnt(0)
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 0xf) | ((N >> 4) << 14));
}

// NKS: pipeline steps of layer 1 (K1 = 32 * NKS)
template <int NKS, int NPROD>   // NPROD: 4 = all four products (weights exact), 3 = without the P2 planes (GemmArgs::products)
__global__ __launch_bounds__(256, 1) void head_fused_kernel(HeadArgs a, int mt) {
    constexpr int NKB1 = 2 * NKS, NKB2 = LH / 16;
    static_assert(NKS % HNST == 0 || HNST % NKS == 0, "the stage of a step must not depend on the tile");
    constexpr int NST = NKS < HNST ? NKS : HNST;
    static_assert(NKS % NST == 0 && NST >= 4, "ring geometry");
    // ALL LDS in one array: [NST steps], the z1 image (8 k-blocks x (hi slab, lo slab)), 4 x 64 partial sums, the tile-queue word
    constexpr int Z1_OFF = NST * HSTEP, PS_OFF = Z1_OFF + NKB2 * 2 * HSLAB, Q_OFF = PS_OFF + 4 * TM * 2;
    __shared__ __attribute__((aligned(16))) unsigned short lds[Q_OFF + 16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    int *qword = reinterpret_cast<int *>(lds + Q_OFF);
    const unsigned q_addr = (unsigned)(size_t)(lptr_t)qword;

    // ---- tile queue (gemm_f16p_ws.hip): one counter, two tiles known ahead
    if (tid == 0) {
        qword[0] = (int)atomicAdd(a.counter, 1u);
        qword[2] = (int)atomicAdd(a.counter, 1u);   // (not adjacent: two 32-bit LDS operations, never one 64-bit one)
    }
    __syncthreads();
    int c_cur = __builtin_amdgcn_readfirstlane(qword[0]), c_nxt = __builtin_amdgcn_readfirstlane(qword[2]);
    if ((unsigned)c_cur >= (unsigned)mt) return;   // (queue indices are compared unsigned: no word that is not a tile index passes)

    // ---- weight fragments: lane (fr, fh) holds W[32 wave + fr][16 kb + 8 fh .. + 8] of each of the three planes
    f16x8 w1a[NKB1], w1b[NKB1], w1c[NPROD == 4 ? NKB1 : 1], w2a[NKB2], w2b[NKB2], w2c[NPROD == 4 ? NKB2 : 1];
    {
        const size_t p1 = (size_t)NKB1 * (128 * 16), p2 = (size_t)NKB2 * (128 * 16);   // plane sizes (N = 128: one 128-row tile)
        const unsigned short *b1 = a.W1 + (wave * 32 + fr) * 16 + fh * 8, *b2 = a.W2 + (wave * 32 + fr) * 16 + fh * 8;
#pragma unroll
        for (int kb = 0; kb < NKB1; ++kb) {
            w1a[kb] = *reinterpret_cast<const f16x8 *>(b1 + (size_t)kb * 2048);
            w1b[kb] = *reinterpret_cast<const f16x8 *>(b1 + p1 + (size_t)kb * 2048);
            if constexpr (NPROD == 4) w1c[kb] = *reinterpret_cast<const f16x8 *>(b1 + 2 * p1 + (size_t)kb * 2048);
        }
#pragma unroll
        for (int kb = 0; kb < NKB2; ++kb) {
            w2a[kb] = *reinterpret_cast<const f16x8 *>(b2 + (size_t)kb * 2048);
            w2b[kb] = *reinterpret_cast<const f16x8 *>(b2 + p2 + (size_t)kb * 2048);
            if constexpr (NPROD == 4) w2c[kb] = *reinterpret_cast<const f16x8 *>(b2 + 2 * p2 + (size_t)kb * 2048);
        }
    }
    // per accumulator register r (column 8 (r >> 2) + 4 fh + (r & 3) of the wave's 32): biases and the classifier weight
    float bias1[16], bias2[16], wcls[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int col = wave * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
        bias1[r] = a.b1[col]; bias2[r] = a.b2[col]; wcls[r] = a.wc[col];
    }
    const float s1 = a.w1scale, s2 = a.w2scale, slope = a.slope, bcls = a.bc[0];

    // ---- DMA plan: a step is 8 pieces of 1 KiB (32 rows x 32 bytes); wave w moves both halves of slab w of the step
    //      (slab 0 / 1 = hi / lo plane of the even k-block, 2 / 3 = of the odd one).  Source swizzle as in gemm_f16p.hip.
    const int prow = lane >> 1;
    const unsigned poff = (unsigned)((prow * 2 + ((lane & 1) ^ ((prow >> 3) & 1))) * 8);
    const unsigned short *src_plane = (wave & 1) ? a.Yl : a.Yh;
    const int nkb_src = NKB1;   // k-blocks per 128-row tile of the source planes
    auto issue = [&](int stage, int c, int step) {
        // source: plane_index(row, col, K1) with row = 64 c, col = 16 (2 step + (wave >> 1))
        const size_t row0 = (size_t)c * TM;
        const size_t slab = ((row0 >> 7) * nkb_src + 2 * step + (wave >> 1)) * (128 * 16) + (row0 & 127) * 16;
        unsigned short *img = lds + stage * HSTEP + wave * HSLAB;
        __builtin_amdgcn_global_load_lds((gptr_t)(src_plane + slab + poff), (lptr_t)(img), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src_plane + slab + poff + 512), (lptr_t)(img + 512), 16, 0, 0);
    };
    auto clampc = [&](int c, int fallback) { return (unsigned)c < (unsigned)mt ? c : fallback; };

    // fragment reads (inline asm, hand-counted lgkmcnt: see gemm_f16p_ws.hip): lane (fr, fh) of row block i reads row 32 i + fr, chunk
    // fh ^ ((fr >> 3) & 1) of a slab
    const unsigned f_base = (unsigned)(size_t)(lptr_t)lds + (unsigned)((fr * 16 + ((fh ^ ((fr >> 3) & 1)) * 8)) * 2);
    f16x8 ah[2][2][2], al[2][2][2];   // [set][k-block of the step][row block]
#define HF_READ(dst, byte_off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(f_base), "n"(byte_off))
#define HF_READ_STEP(set, stage)                                                                                            \
    {                                                                                                                       \
        constexpr int b_ = (stage) * (HSTEP * 2);                                                                           \
        HF_READ(ah[set][0][0], b_ + 0 * 2048 + 0); HF_READ(ah[set][0][1], b_ + 0 * 2048 + 1024);                            \
        HF_READ(al[set][0][0], b_ + 1 * 2048 + 0); HF_READ(al[set][0][1], b_ + 1 * 2048 + 1024);                            \
        HF_READ(ah[set][1][0], b_ + 2 * 2048 + 0); HF_READ(ah[set][1][1], b_ + 2 * 2048 + 1024);                            \
        HF_READ(al[set][1][0], b_ + 3 * 2048 + 0); HF_READ(al[set][1][1], b_ + 3 * 2048 + 1024);                            \
    }

    // ---- prologue
#pragma unroll
    for (int v = 0; v < NST - 1; ++v) issue(v, v < NKS ? c_cur : clampc(c_nxt, c_cur), v % NKS);
    __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * (NST - 2)) & 0xf) | (((2 * (NST - 2)) >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    HF_READ_STEP(0, 0)

    f32x16 hi[2], lo[2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // z1 image write address of this lane: row (32 i + fr), k-block 2 wave + (q >> 1), chunk (q & 1) ^ ((fr >> 3) & 1), 4 halves at + 4 fh
    // (byte addresses of the two chunk slots of row fr, + 8 fh: the lane's four halves inside the chunk; k-block / plane / row block by immediate)
    const unsigned z1w_b0 = (unsigned)(size_t)(lptr_t)(lds + Z1_OFF) + (unsigned)(2 * wave * (2 * HSLAB * 2)) + (unsigned)((fr * 16 + (((fr >> 3) & 1)) * 8 + 4 * fh) * 2);
    const unsigned z1w_b1 = (unsigned)(size_t)(lptr_t)(lds + Z1_OFF) + (unsigned)(2 * wave * (2 * HSLAB * 2)) + (unsigned)((fr * 16 + (((fr >> 3) & 1) ^ 1) * 8 + 4 * fh) * 2);
#define Z1B(i, q) (((q) >> 1) * (2 * HSLAB * 2) + (i) * (32 * 16 * 2))   /* + the wave's two k-blocks, in the base */
    const unsigned z1r = (unsigned)(size_t)(lptr_t)(lds + Z1_OFF) + (unsigned)((fr * 16 + ((fh ^ ((fr >> 3) & 1)) * 8)) * 2);
    float *psum = reinterpret_cast<float *>(lds + PS_OFF);
#define HF_SB __builtin_amdgcn_sched_barrier(0)

    auto tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < NKB1; ++kb) {
            asm volatile("" : "+a"(w1a[kb]), "+a"(w1b[kb]));
            if constexpr (NPROD == 4) asm volatile("" : "+a"(w1c[kb]));
        }
        const int c_n = clampc(c_nxt, c_cur);
        unsigned pulled = 0;
        // ================= layer 1: NKS steps of two k-blocks =================
        hf_static_for<NKS>([&](auto st_tag) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value;
            constexpr int s = st & 1, sn = (st + 1) & 1, stn = (st + 1) % NST;
            HF_SB;
            hf_wait_vm_lgkm0<2 * (NST - 3)>();   // own pieces of step st + 1 landed (the logit stores of wave 0 are not counted: stricter, never looser)
            __builtin_amdgcn_s_barrier();
            HF_SB;
            constexpr int v = st - 1 + NST;
            const int vc = v < NKS ? c_cur : c_n;
            const size_t row0 = (size_t)vc * TM;
            const size_t slab = ((row0 >> 7) * nkb_src + 2 * (v % NKS) + (wave >> 1)) * (128 * 16) + (row0 & 127) * 16;
            unsigned short *img = lds + ((st + NST - 1) % NST) * HSTEP + wave * HSLAB;
            auto M = [&](auto j_tag) __attribute__((always_inline)) {   // MFMA j of the step: k-block j >> 3, product (j >> 1) & 3, row block j & 1
                constexpr int j = decltype(j_tag)::value, kk = j >> 3, p = (j >> 1) & 3, i = j & 1, kb = 2 * st + kk;
                constexpr bool first = st == 0 && kk == 0;
                if constexpr (p == 0) hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1a[kb], ah[s][kk][i], first ? zero : hi[i], 0, 0, 0);
                if constexpr (p == 1) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1b[kb], ah[s][kk][i], first ? zero : lo[i], 0, 0, 0);
                if constexpr (p == 2) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1a[kb], al[s][kk][i], lo[i], 0, 0, 0);
                if constexpr (p == 3 && NPROD == 4) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1c[kb], ah[s][kk][i], lo[i], 0, 0, 0);
            };
#define HF_M(j) M(std::integral_constant<int, (j)>{})
            constexpr int fb = stn * (HSTEP * 2);
            HF_M(0); HF_M(1); HF_SB;
            __builtin_amdgcn_global_load_lds((gptr_t)(src_plane + slab + poff), (lptr_t)(img), 16, 0, 0);
            HF_SB; HF_M(2); HF_M(3); HF_SB;
            __builtin_amdgcn_global_load_lds((gptr_t)(src_plane + slab + poff + 512), (lptr_t)(img + 512), 16, 0, 0);
            HF_SB;
            if constexpr (st == 0) {   // thread 0 draws the tile after the next one (result read at st = NST - 2, behind that step's wait)
                unsigned long long sv;
                asm volatile("v_cmp_eq_u32_e32 vcc, 0, %2\n\ts_and_saveexec_b64 %1, vcc\n\tglobal_atomic_add %0, %3, %4, off sc0\n\ts_mov_b64 exec, %1"
                             : "=&v"(pulled), "=&s"(sv) : "v"(tid), "v"(a.counter), "v"(1u) : "vcc", "memory");
                HF_SB;
            }
            if constexpr (st == NST - 2) {
                unsigned long long sv;
                asm volatile("v_cmp_eq_u32_e32 vcc, 0, %1\n\ts_and_saveexec_b64 %0, vcc\n\tds_write_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %0"
                             : "=&s"(sv) : "v"(tid), "v"(q_addr), "v"(pulled) : "vcc", "memory");
                HF_SB;
            }
            HF_M(4); HF_M(5); HF_SB;
            HF_READ(ah[sn][0][0], fb + 0 * 2048 + 0); HF_READ(ah[sn][0][1], fb + 0 * 2048 + 1024); HF_SB;
            HF_M(6); HF_M(7); HF_SB;
            HF_READ(al[sn][0][0], fb + 1 * 2048 + 0); HF_READ(al[sn][0][1], fb + 1 * 2048 + 1024); HF_SB;
            HF_M(8); HF_M(9); HF_SB;
            HF_READ(ah[sn][1][0], fb + 2 * 2048 + 0); HF_READ(ah[sn][1][1], fb + 2 * 2048 + 1024); HF_SB;
            HF_M(10); HF_M(11); HF_SB;
            HF_READ(al[sn][1][0], fb + 3 * 2048 + 0); HF_READ(al[sn][1][1], fb + 3 * 2048 + 1024); HF_SB;
            HF_M(12); HF_M(13); HF_M(14); HF_M(15); HF_SB;
        });
        // ================= z1 = leaky_relu((hi + lo 2^-11) 2^-S1 + b1) as two f16 planes into the LDS image =================
        (void)z1w_b0; (void)z1w_b1;
        hf_static_for<8>([&](auto iq_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(iq_tag)::value >> 2, q = decltype(iq_tag)::value & 3;
            unsigned hbits[2], lbits[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                unsigned short hh[2], ll[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 4 * q + 2 * jj + e;
                    float z = __builtin_fmaf(__builtin_fmaf(lo[i][r], 0.00048828125f, hi[i][r]), s1, bias1[r]);
                    z = z >= 0.f ? z : slope * z;
                    const _Float16 h = (_Float16)z;
                    hh[e] = __builtin_bit_cast(unsigned short, h);
                    ll[e] = __builtin_bit_cast(unsigned short, (_Float16)((z - (float)h) * 2048.0f));
                }
                hbits[jj] = hh[0] | ((unsigned)hh[1] << 16);
                lbits[jj] = ll[0] | ((unsigned)ll[1] << 16);
            }
            // k-block 2 wave + (q >> 1); its hi slab, then its lo slab; row 32 i + fr; chunk slot (q & 1) ^ ((fr >> 3) & 1).  Inline asm:
            // hipcc merges adjacent 32-bit LDS stores into 64-bit ones (ds_write2st64_b64), the class this library keeps out of
            // kernels that run beside MFMA loops of other workgroups
            const unsigned dstb = (q & 1) ? z1w_b1 : z1w_b0;
#define HF_ZWRITE(data, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(dstb), "v"(data), "n"(off) : "memory")
            HF_ZWRITE(hbits[0], Z1B(i, q)); HF_ZWRITE(hbits[1], Z1B(i, q) + 4);
            HF_ZWRITE(lbits[0], Z1B(i, q) + HSLAB * 2); HF_ZWRITE(lbits[1], Z1B(i, q) + HSLAB * 2 + 4);
        });
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the image is written
        __builtin_amdgcn_s_barrier();
        // ================= layer 2: 8 k-blocks out of the image =================
        {
            f16x8 zh[2][2], zl[2][2];   // [set][row block]
#define HF_ZREAD(dst, byte_off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(z1r), "n"(byte_off))
            HF_ZREAD(zh[0][0], 0); HF_ZREAD(zh[0][1], 1024); HF_ZREAD(zl[0][0], 2048); HF_ZREAD(zl[0][1], 2048 + 1024);
            hf_static_for<NKB2>([&](auto kb_tag) __attribute__((always_inline)) {
                constexpr int kb = decltype(kb_tag)::value, s = kb & 1, sn = (kb + 1) & 1;
                HF_SB;
                if constexpr (kb + 1 < NKB2) {
                    HF_ZREAD(zh[sn][0], (kb + 1) * 4096 + 0); HF_ZREAD(zh[sn][1], (kb + 1) * 4096 + 1024);
                    HF_ZREAD(zl[sn][0], (kb + 1) * 4096 + 2048); HF_ZREAD(zl[sn][1], (kb + 1) * 4096 + 2048 + 1024);
                    __builtin_amdgcn_s_waitcnt(0xc07f | (4 << 8));   // lgkmcnt(4): this k-block's fragments
                } else {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                }
                HF_SB;
#pragma unroll
                for (int i = 0; i < 2; ++i) hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2a[kb], zh[s][i], kb == 0 ? zero : hi[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2b[kb], zh[s][i], kb == 0 ? zero : lo[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2a[kb], zl[s][i], lo[i], 0, 0, 0);
                if constexpr (NPROD == 4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2c[kb], zh[s][i], lo[i], 0, 0, 0);
                }
            });
        }
        HF_SB;
        // ================= z2, the classifier row, the sigmoid =================
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float z = __builtin_fmaf(__builtin_fmaf(lo[i][r], 0.00048828125f, hi[i][r]), s2, bias2[r]);
                z = z >= 0.f ? z : slope * z;
                acc = __builtin_fmaf(z, wcls[r], acc);
            }
            acc += __shfl_xor(acc, 32);                           // the other 16 columns of the wave's 32
            if (fh == 0) psum[wave * TM + i * 32 + fr] = acc;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (wave == 0) {   // 64 lanes, 64 rows
            const float logit = ((psum[lane] + psum[TM + lane]) + (psum[2 * TM + lane] + psum[3 * TM + lane])) + bcls;
            const long long m = (long long)c_cur * TM + lane;      // tile-major row: (seq tile * T + t) * 4 + j
            const long long per_tile = (long long)a.T * SEQ_TILE;
            if (m < (long long)a.tiles * per_tile) {
                const int tl = (int)(m / per_tile);
                const int rem = (int)(m - (long long)tl * per_tile);
                const int t = rem / SEQ_TILE, b = tl * SEQ_TILE + (rem - t * SEQ_TILE);
                if (b < a.B) {
                    const size_t o = (size_t)b * a.ld_out + t;
                    if (a.logits) a.logits[o] = logit;
                    if (a.probs) a.probs[o] = 1.0f / (1.0f + expf(-logit));
                }
            }
        }
        int c_nn;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c_nn) : "v"(q_addr) : "memory");
        c_cur = c_nxt;
        c_nxt = __builtin_amdgcn_readfirstlane(c_nn);
        // the next tile's step-0 fragments were read during the last step of layer 1 (set NKS & 1 = 0): still in registers
    };
    tile();
    for (int done = 1; done < mt && (unsigned)c_cur < (unsigned)mt; ++done) tile();
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): no LDS-DMA may be in flight when the LDS is handed on
}

}  // namespace

bool head_fused_supported(int K1, int lin_hidden, int lin_layers, long long M, int n_cu) {
    if (lin_layers != 2 || lin_hidden != LH || (K1 != 256 && K1 != 128)) return false;
    (void)n_cu;
    return M > 0;   // small launches too (a streaming step of 512 feeds = 16 tiles): one launch instead of three dependent ones
}

hipError_t launch_head_fused(const HeadArgs &a, int n_cu, hipStream_t s) {
    if (!a.Yh || !a.Yl || !a.W1 || !a.W2 || !a.b1 || !a.b2 || !a.wc || !a.bc || !a.counter || a.M <= 0) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.counter, 0, sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    const int mt = (int)((a.M + TM - 1) / TM);
    const int ncu = n_cu > 0 ? n_cu : 256;
    const int grid = mt < ncu ? mt : ncu;
    if (a.products != 0 && a.products != 3 && a.products != 4) return hipErrorInvalidValue;
    const bool three = a.products == 3;
    if (a.K1 == 256 && three) hipLaunchKernelGGL((head_fused_kernel<8, 3>), dim3(grid), dim3(256), 0, s, a, mt);
    else if (a.K1 == 256) hipLaunchKernelGGL((head_fused_kernel<8, 4>), dim3(grid), dim3(256), 0, s, a, mt);
    else if (a.K1 == 128 && three) hipLaunchKernelGGL((head_fused_kernel<4, 3>), dim3(grid), dim3(256), 0, s, a, mt);
    else if (a.K1 == 128) hipLaunchKernelGGL((head_fused_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a, mt);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace uvad

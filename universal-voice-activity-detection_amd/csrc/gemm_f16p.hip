// gemm_f16p.hip -- f32-accurate GEMM on the f16 matrix cores, operands PRE-SPLIT into f16 planes ("f16p").
//   C = act(A * W^T + b) for the time-parallel contractions of the classifier: the four LSTM input projections
//   (x_t * W_ih^T + b_ih + b_hh, nn.LSTM inside PyanNet2.forward, src/models/segmentation/PyanNet2.py:169-172) and the
//   feed-forward layers (leaky_relu(x * W^T + b), PyanNet2.py:183-185).
//
// Numerics
//   Weights are static, so their representation must be EXACT: a rounded weight is not noise but a slightly different
//   network, and the near-chaotic x4 test network turns a 2^-23 relative weight perturbation into a mean logit error as large
//   as the whole fp32 CPU path's (measured).  Each weight matrix is scaled by a power of two (2^S, so
//   that max|w| lands in [2^13, 2^14): no piece of a weight that matters falls into the f16 subnormals) and split on the
//   host into THREE f16 planes that add up to the f32 value exactly (11 + 11 + 2 mantissa bits):
//       w * 2^S = P0 + (P1 + P2) * 2^-11,   P0 = f16(ws),  P1 = f16((ws - P0) * 2^11),  P2 = f16((ws - P0) * 2^11 - P1)
//   Activations vary, so their rounding IS noise; they arrive as TWO planes written by the kernel that produced them
//   (the recurrent kernel's epilogue, the feature split kernel, this kernel's own epilogue):
//       a ~= a1 + a2 * 2^-11,   a1 = f16(a), a2 = f16((a - a1) * 2^11)            (|residual| <= 2^-22 |a|, 22 of 24 bits)
//   The product keeps four terms in two f32 accumulator sets (f16 x f16 products are exact in f32 and
//   v_mfma_f32_32x32x16_f16 accumulates in f32):
//       hi += a1*P0;   lo += a1*P1 + a2*P0 + a1*P2;   a*w = (hi + lo * 2^-11) * 2^-S   (dropped: a2*P1, a2*P2 <= 2^-22 relative)
//   Operand range: |a| < 65504.  h of an LSTM is in (-1, 1); the feed-forward activations are bounded by the weights
//   (uvad_finalize checks the bound); caller-supplied features are checked on the device (split_features_kernel sets a flag
//   and the exact-f32 kernel of gemm.hip runs the first projection instead, GemmArgs::gate).
//
// gfx950 design (HBM-bound on paper: 1.05 GB of gate pre-activations written per K = 256 projection at cfg 2)
//   * 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = four 32x32 MFMA tiles x two accumulator
//     sets = 128 accumulator registers): ~162 VGPRs, so THREE workgroups share a CU and hide each other's barriers and
//     memory latency -- occupancy instead of a deep software pipeline.  The kernel is bound by the operand stream through the
//     vector memory path (L2 -> LDS), not by the matrix pipe: the first version's 128 x 64 tile moved 14 KiB per 32 MFMAs,
//     this one 20 KiB per 64 (ablations and counters: DESIGN.md section 3.3).
//   * All five operand planes of a K-step go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
//     VALU), double buffered: one barrier per K-step, the next step's loads are in flight during this step's MFMAs.
//   * Planes live in HBM in a K-BLOCKED layout (plane_index() in uvad_internal.h): [row tile][16-column block][row][16], so
//     the slab a workgroup needs for one 16-deep k-block of one plane is ONE contiguous 4 KiB run (128 rows) and every DMA
//     wave-instruction reads eight full 128-byte lines.  (Row-major planes were measured first: a K-step then touches
//     32-byte pieces of rows 512 bytes apart, 32 L2 requests per wave-instruction, and the kernel is bound by the L2
//     request rate -- TA busy 77 %, MFMA busy 21 %, profiles/README.md.)
//   * The LDS image is dense (a DMA wave-instruction writes 1 KiB contiguously); bank conflicts of the ds_read_b128
//     fragment reads are removed by an XOR swizzle of the 16-byte chunk index applied to the DMA's per-lane SOURCE
//     address and to the read address (cdna_hip_programming.md section 5.4 rule 21).
//   * Epilogue: the tile is assembled in LDS (over the operand stages) and leaves as unmasked 16-byte stores into its
//     contiguous run of the blocked gate matrix / of the K-blocked output planes (dword stores straight from the 32x32 C
//     layout reached 2.9 TB/s and did not overlap the K loops of the other workgroups).
//   * Block ids are remapped so that the N-tiles sharing one A row panel run on the same XCD (private L2) back to back.
#include "uvad_internal.h"

namespace uvad {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int BM = PLANE_TILE, BN = 128, GT = 64;   // GT: column width of a tile of the blocked gate matrix (g_index)

// LDS image of one 16-deep k-block (f16 elements): 128 rows each of A hi, A lo, W P0, W P1, W P2; a row is 16 elements = two
// 16-byte chunks, chunk c of row r sits at slot c ^ ((r >> 3) & 1) (rows r and r + 8 share a bank line)
constexpr int SLAB = 128 * 16;
constexpr int KB_AHI = 0, KB_ALO = SLAB, KB_W0 = 2 * SLAB, KB_W1 = 3 * SLAB, KB_W2 = 4 * SLAB, KB_ELEMS = 5 * SLAB;
static_assert(BM == 128 && BN == 128, "the DMA plan moves five 128-row slabs per k-block, one 32-row piece per wave each");

#ifndef UVAD_F16P_NST
#define UVAD_F16P_NST 2
#endif
#ifndef UVAD_F16P_OCC
#define UVAD_F16P_OCC 3
#endif

template <bool OUT_PLANES, int NSTAGES>
__global__ __launch_bounds__(256, NSTAGES == 2 ? (OUT_PLANES ? 2 : UVAD_F16P_OCC) : 1) void gemm_f16p_kernel(GemmArgs a, int mt, int nt) {
    // LDS stages of one 16-deep k-block each (20 KiB).  NSTAGES = 2, the throughput instance: a plain double buffer, three
    // workgroups per CU hide each other's waits (rings of 3 and 4 stages with counted vmcnt waits measured the same or slower,
    // also for the N = 128 feed-forward layers).  NSTAGES = 4, the latency instance for launches that do not fill the chip
    // (the per-chunk steps of uvad_stream_step: 64 workgroups at 512 feeds): three k-blocks in flight per workgroup, because
    // there the K loop is a chain of L2 / HBM round trips with nothing else on the CU to cover them.
    constexpr int STAGE = KB_ELEMS, NST = NSTAGES;
    constexpr bool EPI_HALVES = NST * STAGE < BM * BN * 2;   // the f32 output tile of the blocked epilogue aliases the stages: whole, or one 128 x 64 half at a time
    constexpr int EPI_ELEMS = EPI_HALVES ? BM * GT * 2 : BM * BN * 2;
    constexpr int LDS_ELEMS = NST * STAGE > EPI_ELEMS ? NST * STAGE : EPI_ELEMS;
    __shared__ __attribute__((aligned(16))) unsigned short lds[LDS_ELEMS];

    if (a.gate && (*a.gate != 0) != (a.gate_run_if_set != 0)) return;   // device-side kernel selection (see GemmArgs)
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch, speed only)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int m_tile = (idx / nt) * 8 + xcd, n_tile = idx % nt;
    if (m_tile >= mt) return;
    const int R0 = m_tile * BM, C0 = n_tile * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- DMA plan.  Per k-block FIVE wave-instructions per wave: rows [32 wave, 32 wave + 32) of each of the five slabs.  A lane
    //      always fetches the chunk that belongs at its LDS slot (slot s of row r holds chunk s ^ ((r >> 3) & 1)): bases are
    //      wave-uniform, the lane offset is one constant.
    const int nkb = a.K / 16;                                   // k-blocks of the whole contraction (K is a multiple of 32)
    const size_t wplane = (size_t)((a.N + BN - 1) / BN) * nkb * SLAB;   // elements per W plane (N padded to whole 128-row tiles)
    const int ra = tid >> 1;
    const unsigned off = (unsigned)((ra * 2 + ((tid & 1) ^ ((ra >> 3) & 1))) * 8);
    const unsigned short *a_hi = a.Ah + (size_t)m_tile * nkb * SLAB + off;
    const unsigned short *a_lo = a.Al + (size_t)m_tile * nkb * SLAB + off;
    const unsigned short *w_0 = a.Wsplit16 + (size_t)n_tile * nkb * SLAB + off;
    const unsigned short *w_1 = w_0 + wplane, *w_2 = w_0 + 2 * wplane;
    auto issue = [&](int stage, int kb) {
        unsigned short *img = lds + stage * STAGE + wave * 512;
        const size_t k = (size_t)kb * SLAB;
        __builtin_amdgcn_global_load_lds((gptr_t)(a_hi + k), (lptr_t)(img + KB_AHI), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a_lo + k), (lptr_t)(img + KB_ALO), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(w_0 + k), (lptr_t)(img + KB_W0), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(w_1 + k), (lptr_t)(img + KB_W1), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(w_2 + k), (lptr_t)(img + KB_W2), 16, 0, 0);
    };

    // accumulators of the wave's 64 x 64 tile: [row block][column block], hi and lo sets
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { hi[i][j][r] = 0.f; lo[i][j][r] = 0.f; }

    // fragment addresses (f16 elements inside a slab): lane = (row fr of the 32-row MFMA tile, k-half fh)
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    int fa[2], fw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ar = wr * 64 + i * 32 + fr, wrw = wc * 64 + i * 32 + fr;
        fa[i] = ar * 16 + ((fh ^ ((ar >> 3) & 1)) * 8);
        fw[i] = wrw * 16 + ((fh ^ ((wrw >> 3) & 1)) * 8);
    }

    // ---- k loop: one barrier per k-block; the DMA of the next k-block(s) is in flight during the MFMAs of this one.
    //      Per wave and k-block: 10 ds_read_b128 and 16 MFMAs (4 products x 2 x 2 tiles); per workgroup 20 KiB through the
    //      vector memory path for 64 MFMAs (the 128 x 64 tile of the first version moved 14 KiB per 32 MFMAs and was bound by
    //      exactly that: TA busy 77-86 %, matrix pipe 65 % inside the k loop).
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < nkb) issue(p, p);
    for (int kt = 0; kt < nkb; ++kt) {
        if constexpr (NST == 2) {
            __syncthreads();   // (hipcc waits vmcnt(0) here) k-block kt has landed for every wave; everyone is done reading k-block kt-1
        } else {
            // k-blocks kt .. kt+NST-2 are in flight (5 loads each, fewer at the tail): wait for the oldest only
            if (kt + NST - 2 < nkb) __builtin_amdgcn_s_waitcnt(0x0f70 | (5 * (NST - 2) & 0xf) | (((5 * (NST - 2)) >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_barrier();
        }
        if (kt + NST - 1 < nkb) issue((kt + NST - 1) % NST, kt + NST - 1);
        const unsigned short *st = lds + (kt % NST) * STAGE;
        f16x8 ah[2], al[2], w0[2], w1[2], w2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const f16x8 *>(st + KB_AHI + fa[i]);
            w0[i] = *reinterpret_cast<const f16x8 *>(st + KB_W0 + fw[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            w1[i] = *reinterpret_cast<const f16x8 *>(st + KB_W1 + fw[i]);
            al[i] = *reinterpret_cast<const f16x8 *>(st + KB_ALO + fa[i]);
            w2[i] = *reinterpret_cast<const f16x8 *>(st + KB_W2 + fw[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], w0[j], hi[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], w1[j], lo[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], w0[j], lo[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], w2[j], lo[i][j], 0, 0, 0);
    }

    // ---- epilogue: (hi + lo * 2^-11) * 2^-S + bias, activation.  C layout of the 32x32 MFMA: register r of lane (fr, fh) =
    //      row 8*(r>>2) + 4*fh + (r&3), column fr.
    if (!OUT_PLANES && a.c_blocked) {
        // Tile-blocked gate matrix: this workgroup's 128 x 128 tile is two adjacent 128 x 64 tiles of G = ONE contiguous 64 KiB
        // run (whole tiles exist for the padding rows too), so the tile goes through LDS and leaves as unmasked 16-byte
        // stores, 1 KiB per wave-instruction (dword stores straight from the accumulators measured 2.9 TB/s and did not
        // overlap with the K loops).
        float *Ct = reinterpret_cast<float *>(lds);
        float *dst = a.C + ((size_t)m_tile * (a.N / GT) + (size_t)n_tile * (BN / GT)) * (BM * GT);
#pragma unroll
        for (int half = 0; half < (EPI_HALVES ? 2 : 1); ++half) {
            __syncthreads();   // every wave is done reading the last stage / storing the previous half
            if (!EPI_HALVES || wc == half) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int lc = wc * 64 + j * 32 + fr;                    // column inside the 128-wide tile
                    const float bias = a.bias ? a.bias[C0 + lc] : 0.f;
                    float *Cj = Ct + (EPI_HALVES ? 0 : (lc / GT) * (BM * GT)) + (lc % GT);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lr = wr * 64 + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                            Cj[lr * GT] = __builtin_fmaf(__builtin_fmaf(lo[i][j][r], 0.00048828125f, hi[i][j][r]), a.wscale, bias);
                        }
                }
            }
            __syncthreads();
            constexpr int N4 = (EPI_HALVES ? BM * GT : BM * BN) / 4;
#pragma unroll
            for (int j = 0; j < N4 / 256; ++j) {
                const int q = tid + 256 * j;
                *reinterpret_cast<float4 *>(dst + (size_t)half * (BM * GT) + (size_t)q * 4) = *reinterpret_cast<const float4 *>(Ct + q * 4);
            }
        }
        return;
    }
    if constexpr (OUT_PLANES) {
        if (a.ldc % BN == 0) {
            // Whole 128-column tiles of K-blocked planes: the tile's image of ONE plane -- 8 column blocks x 128 rows x 16 -- is a
            // contiguous 32 KiB run (plane_index), rows of the padding tile included; it is assembled in LDS, one plane after the
            // other, and leaves as 16-byte stores (2-byte stores straight from the accumulators: 0.19 ms for the 65 MB of a
            // 256 -> 128 layer at cfg 2).
            unsigned short *Pt = lds;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                __syncthreads();   // every wave is done reading the last stage / storing the other plane
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int lc = wc * 64 + j * 32 + fr, col = C0 + lc;
                    const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
                    unsigned short *Pj = Pt + (lc >> 4) * (BM * 16) + (lc & 15);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lr = wr * 64 + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                            float v = __builtin_fmaf(__builtin_fmaf(lo[i][j][r], 0.00048828125f, hi[i][j][r]), a.wscale, bias);
                            if (a.act == 1) v = v >= 0.f ? v : a.leaky_slope * v;
                            if (col >= a.N) v = 0.f;   // padding columns of the consumer's K must read as zero
                            const _Float16 h = (_Float16)v;
                            const _Float16 piece = pl == 0 ? h : (_Float16)((v - (float)h) * 2048.0f);
                            Pj[lr * 16] = __builtin_bit_cast(unsigned short, piece);
                        }
                }
                __syncthreads();
                unsigned short *dstp = (pl == 0 ? a.Ch : a.Cl) + ((size_t)m_tile * (a.ldc / 16) + (size_t)n_tile * (BN / 16)) * (BM * 16);
#pragma unroll
                for (int k = 0; k < BM * BN / 8 / 256; ++k) {
                    const int q = tid + 256 * k;
                    *reinterpret_cast<uint4 *>(dstp + (size_t)q * 8) = *reinterpret_cast<const uint4 *>(Pt + q * 8);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = C0 + wc * 64 + j * 32 + fr;
        const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = R0 + wr * 64 + i * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                float v = __builtin_fmaf(__builtin_fmaf(lo[i][j][r], 0.00048828125f, hi[i][j][r]), a.wscale, bias);
                if (a.act == 1) v = v >= 0.f ? v : a.leaky_slope * v;
                if constexpr (OUT_PLANES) {
                    // consumer = another f16p GEMM whose K is ldc (N rounded up to 32): its padding columns must read as zero
                    if (row < a.M && col < a.ldc) {
                        if (col >= a.N) v = 0.f;
                        const _Float16 h = (_Float16)v;
                        const _Float16 l = (_Float16)((v - (float)h) * 2048.0f);
                        const size_t o = plane_index(row, col, a.ldc);
                        a.Ch[o] = __builtin_bit_cast(unsigned short, h);
                        a.Cl[o] = __builtin_bit_cast(unsigned short, l);
                    }
                } else {
                    if (row < a.M && col < a.N) a.C[(size_t)row * a.ldc + col] = v;
                }
            }
    }
}

// Canonical f32 features [B][T][F] -> the two f16 planes (K-blocked, Fp columns) in tile-major row order (row m = (tile*T + t)*4 + j
// holds sequence b = 4*tile + j at frame t; rows of padding sequences and the columns [F, Fp) are zero), and *flag = 1 if any
// value is non-finite or outside the f16 range (the caller then runs the exact-f32 projection, see GemmArgs::gate).
__global__ __launch_bounds__(256) void split_features_kernel(const float *x, int B, int T, int F, int Fp, int tiles, unsigned short *xh,
                                                             unsigned short *xl, int *flag) {
    const int q4 = Fp / 4;
    const long long n = (long long)tiles * T * SEQ_TILE * q4;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / q4;
        const int c = (int)(i - m * q4) * 4;
        const long long per_tile = (long long)T * SEQ_TILE;
        const int tile = (int)(m / per_tile);
        const int rem = (int)(m - (long long)tile * per_tile);
        const int t = rem / SEQ_TILE, b = tile * SEQ_TILE + (rem - t * SEQ_TILE);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && c < F) v = *reinterpret_cast<const float4 *>(x + ((size_t)b * T + t) * F + c);   // F % 4 == 0
        bad |= !(__builtin_fabsf(v.x) < 65504.0f) | !(__builtin_fabsf(v.y) < 65504.0f) | !(__builtin_fabsf(v.z) < 65504.0f) | !(__builtin_fabsf(v.w) < 65504.0f);
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const _Float16 hh = (_Float16)e[k];
            const _Float16 ll = (_Float16)((e[k] - (float)hh) * 2048.0f);
            h[k] = __builtin_bit_cast(unsigned short, hh);
            l[k] = __builtin_bit_cast(unsigned short, ll);
        }
        const size_t o = plane_index(m, c, Fp);   // c % 4 == 0: the four columns stay inside one 16-column block
        *reinterpret_cast<uint2 *>(xh + o) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2 *>(xl + o) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
    if (flag && __builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

int gemm_f16p_padded_k(int K) { return (K + 31) / 32 * 32; }

// host: f32 [N][ldw] (rows zero-padded to ldw, a multiple of 32) -> three f16 planes of w * 2^S that add up to it EXACTLY (see the
// header), each in the K-blocked layout with 128-row tiles (N padded to whole tiles with zero rows): out holds
// 3 * weight_plane_elems(N, ldw) elements; *wscale = 2^-S.  false: a weight is non-finite (the matrix cannot be represented).
size_t weight_plane_elems(int N, int ldw) { return (size_t)((N + BN - 1) / BN) * BN * ldw; }

bool split_weights_f16x3(const float *w, int N, int ldw, unsigned short *out, float *wscale) {
    const size_t n = (size_t)N * ldw, pe = weight_plane_elems(N, ldw);
    float amax = 0.0f;
    bool finite = true;
    for (size_t i = 0; i < n; ++i) {
        const float a = __builtin_fabsf(w[i]);
        if (!(a <= 3.0e38f)) finite = false;
        if (a > amax) amax = a;
    }
    int S = 0;
    if (finite && amax > 0.0f) {
        int e;
        (void)__builtin_frexpf(amax, &e);   // amax = m * 2^e, m in [0.5, 1)
        S = 14 - e;                           // amax * 2^S in [2^13, 2^14)
        if (S > 100) S = 100;
        if (S < -100) S = -100;
    }
    const float up = __builtin_ldexpf(1.0f, S);
    *wscale = __builtin_ldexpf(1.0f, -S);
    for (size_t i = 0; i < 3 * pe; ++i) out[i] = 0;
    const int nkb = ldw / 16;
    for (int r = 0; r < N; ++r)
        for (int k = 0; k < ldw; ++k) {
            const float ws = finite ? w[(size_t)r * ldw + k] * up : 0.0f;   // exact (power of two)
            const _Float16 p0 = (_Float16)ws;
            const float t2 = (ws - (float)p0) * 2048.0f;            // exact
            const _Float16 p1 = (_Float16)t2;
            const float r2 = t2 - (float)p1;                        // exact; at most 24 - 22 significant bits are left
            const _Float16 p2 = (_Float16)r2;
            const size_t o = ((size_t)(r / BN) * nkb + k / 16) * (BN * 16) + (size_t)(r % BN) * 16 + k % 16;
            __builtin_memcpy(&out[o], &p0, 2);
            __builtin_memcpy(&out[pe + o], &p1, 2);
            __builtin_memcpy(&out[2 * pe + o], &p2, 2);
        }
    return finite;
}

hipError_t launch_gemm_f16p(const GemmArgs &a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if (!a.Ah || !a.Al || !a.Wsplit16 || a.K <= 0 || (a.K & 31) || a.ldw != a.K) return hipErrorInvalidValue;
    if (a.out_planes ? (!a.Ch || !a.Cl) : !a.C) return hipErrorInvalidValue;
    if (a.c_blocked && (a.out_planes || a.act != 0 || a.N % BN != 0)) return hipErrorInvalidValue;   // the gate matrix: whole 128-column tiles, no activation
    const int mt = (a.M + BM - 1) / BM, nt = ((a.out_planes ? a.ldc : a.N) + BN - 1) / BN;
    const int grid = ((mt + 7) / 8) * 8 * nt;
    const bool small = (long)mt * nt <= 512;   // fewer than two workgroups per CU of an MI355X: latency instance
    if (a.out_planes && small) hipLaunchKernelGGL((gemm_f16p_kernel<true, 4>), dim3(grid), dim3(256), 0, s, a, mt, nt);
    else if (a.out_planes) hipLaunchKernelGGL((gemm_f16p_kernel<true, UVAD_F16P_NST>), dim3(grid), dim3(256), 0, s, a, mt, nt);
    else if (small) hipLaunchKernelGGL((gemm_f16p_kernel<false, 4>), dim3(grid), dim3(256), 0, s, a, mt, nt);
    else hipLaunchKernelGGL((gemm_f16p_kernel<false, UVAD_F16P_NST>), dim3(grid), dim3(256), 0, s, a, mt, nt);
    return hipGetLastError();
}

hipError_t launch_split_features(const float *x, int B, int T, int F, int Fp, int tiles, unsigned short *xh, unsigned short *xl, int *flag,
                                 hipStream_t s) {
    if (flag) {
        hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    const long long n = (long long)tiles * T * SEQ_TILE * (Fp / 4);
    if (n <= 0) return hipSuccess;
    const long long want = (n + 255) / 256;
    const int grid = (int)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(split_features_kernel, dim3(grid), dim3(256), 0, s, x, B, T, F, Fp, tiles, xh, xl, flag);
    return hipGetLastError();
}

}  // namespace uvad

// gemm_f16p_ws.hip -- the LSTM input projections (x_t * W_ih^T + b_ih + b_hh, nn.LSTM inside PyanNet2.forward,
// src/models/segmentation/PyanNet2.py:169-172) as a WEIGHT-STATIONARY, PERSISTENT kernel on the f16 matrix cores.
// Same numerics and the same instruction order per accumulator as gemm_f16p.hip (three exact f16 weight planes, two activation
// planes, four v_mfma_f32_32x32x16_f16 products per 16-deep k-block in two f32 accumulator sets): the gate matrix it writes is
// BIT-IDENTICAL to gemm_f16p_kernel's.  What changes is where the operands live.
//
// Why.  gemm_f16p_kernel streams all five operand planes of every 128 x 128 tile through L2 -> LDS: 20 KiB per 64 MFMAs, 5.2 GB
// per K = 256 launch for 0.28 GB of operands, and its matrix pipe is 40 % busy (round-2 review item 2).  The
// weights of a projection are tiny (N = 1024 rows x K = 256: 1.5 MB in three planes) and the row count is huge (256 000), so
// here the weights never move:
//   * a workgroup owns ONE 128-column tile of W for its whole life; each of its 4 waves keeps the three planes of its 32
//     columns for the WHOLE K in registers (3 x K/16 fragments of 4 registers = 192 at K = 256; they are the B operands of the
//     MFMAs and are read in place);
//   * only the two activation planes of a 128-row tile stream through LDS: 8 KiB per 64 MFMAs (2.5 x less than before), by
//     LDS-DMA into a ring of NST stages that stays in flight ACROSS the k-block barriers and across tiles (counted vmcnt, raw
//     s_barrier, all LDS in one array: cdna_hip_programming.md section 5 "Pipelining across barriers");
//   * every wave multiplies all 128 rows by its 32 columns: 4 row blocks x (hi, lo) accumulators = 128 registers, A fragments
//     read one k-block ahead into a second register set, so the 16 MFMAs of a k-block never wait for LDS;
//   * one wave per SIMD (about 400 registers), one workgroup per CU: the workgroups are PERSISTENT and pull 128-row tiles from
//     per-(group, column tile) counters in HBM, two tiles ahead, so a launch that finds part of the chip busy (other steps in
//     flight) still balances, and the 8 column tiles of one row tile are pulled by workgroups of the same XCD group around
//     the same time (the A panel is fetched from HBM once, the other 7 reads hit that XCD's L2);
//   * the MFMAs are issued with the WEIGHT fragment as the A operand and the activation fragment as B, i.e. they compute the
//     transposed 32 x 32 block: a lane then holds four CONSECUTIVE gate columns of one row per accumulator quad, which it
//     writes with one ds_write_b128 into the wave's own staging image of the finished tile ([128 rows][32 columns] f32, rows
//     padded to 144 bytes);
//   * that image leaves during the NEXT tile's k-blocks: 16 pieces of 8 rows x 128 bytes, each one ds_read_b128 and one
//     16-byte store per lane, so every store instruction writes eight FULL 128-byte lines, one or two per k-block, between
//     MFMAs, with the non-temporal hint (the gate matrix is read next by another kernel, and 1 GB of it passing through the
//     L2 evicts the activation panels the other column tiles of the XCD are about to read: K = 64 launch 0.35 -> 0.22 ms).
//     (Stores straight from the accumulators were measured first: 16-byte pieces of 32 different rows per instruction
//     -- 0.185 ms of a 0.477 ms K = 256 launch was the store path, nothing of it overlapped.)
// LDS: NST x 8 KiB ring (64 KiB at K = 256) + 4 x 18 KiB staging.  Launches that do not fill the chip, feed-forward layers (f16-plane output) and K outside
// {64, 96, 128, 256} stay on gemm_f16p_kernel.
#include "uvad_internal.h"

#include <type_traits>
#include <utility>

namespace uvad {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int SLAB = 128 * 16;      // f16 elements of one 16-deep k-block of one plane of a 128-row tile (4 KiB)
constexpr int STAGE = 2 * SLAB;     // hi plane, lo plane
constexpr int GROUPS = 8;           // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only, never correctness)

// ring depth: a divisor of NKB (the stage of a k-block is then the same in every tile, so the unrolled tile body addresses LDS
// with constants) with NST - 1 <= NKB (the DMA runs at most into the NEXT tile)
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant expression
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int NKB> struct Ring { static constexpr int NST = NKB % 8 == 0 ? 8 : NKB % 6 == 0 ? 6 : 4; };

// s_waitcnt vmcnt(N) only.  simm16: vmcnt = {[15:14], [3:0]}, expcnt [6:4] = 7 (no wait), lgkmcnt [11:8] = 15 (no wait)
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 0xf) | ((N >> 4) << 14));
}

// The finished tile leaves as NPIECE pieces during the next tile: pieces [piece_begin(kb), piece_begin(kb + 1)) are read from the
// staging image during k-block kb (kb <= NKB - 2) and stored at the top of k-block kb + 1.
constexpr int NPIECE = 16;
template <int NKB> constexpr int piece_begin(int kb) { return kb >= NKB - 1 ? NPIECE : kb <= 0 ? 0 : (NPIECE * kb) / (NKB - 1); }
// 16-byte stores issued at the top of k-blocks [from, to) of a tile that has a predecessor
template <int NKB> constexpr int stores_in(int from, int to) {
    int n = 0;
    for (int j = from < 1 ? 1 : from; j < to; ++j) n += piece_begin<NKB>(j) - piece_begin<NKB>(j - 1);
    return n;
}

// s_waitcnt vmcnt(N) lgkmcnt(0)
template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0070 | (N & 0xf) | ((N >> 4) << 14));
}

// NKB: 16-deep k-blocks of the contraction (K = 16 * NKB)
// NPROD: MFMA products per f32-equivalent product: 4 = the numerics above (weights exact); 3 = without P2 x a_hi (weights rounded to
// 22 bits, the P2 fragments are not loaded: 64 registers fewer at K = 256): uvad_set_gemm_mode(3)
template <int NKB, int NPROD>
__global__ __launch_bounds__(256, 1) void gemm_f16p_ws_kernel(GemmArgs a, int mt, int nt, unsigned *counters) {
    constexpr int NST = Ring<NKB>::NST;
    static_assert(NKB % 2 == 0 && NKB % NST == 0 && NST - 1 <= NKB && NST >= 4, "ring / fragment double-buffer geometry");
    // ALL LDS in one array (a second __shared__ object beside an LDS-DMA ring makes hipcc wait vmcnt(0) before LDS reads):
    // [NST stages][hi slab | lo slab], the four waves' staging images of the finished tile, the tile-queue word
    constexpr int SG_LD = 36;                    // floats per staged row: 32 + 4 of padding (ds_write_b128 of 8 consecutive rows: no bank conflict)
    constexpr int SG_WAVE = 128 * SG_LD * 2;     // ushort elements of one wave's image (18 KiB)
    __shared__ __attribute__((aligned(16))) unsigned short lds[NST * STAGE + 4 * SG_WAVE + 16];
    int *qword = reinterpret_cast<int *>(lds + NST * STAGE + 4 * SG_WAVE);

    if (a.gate && (*a.gate != 0) != (a.gate_run_if_set != 0)) return;   // device-side kernel selection (see GemmArgs)
    const int bid = blockIdx.x, grp = bid & (GROUPS - 1);
    const int n_tile = (bid >> 3) % nt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // ---- tile queue: row tile m = grp + 8 * c, c drawn from this (group, column tile)'s counter.  Two tiles are known ahead.
    unsigned *ctr = counters + (size_t)(grp * nt + n_tile) * 32;   // one 128-byte line per counter
    // Which 128-row tiles: all of them (queue index i = row tile i), or the list of this column tile's direction (GemmArgs::ws_tiles: the
    // time-chunked projections hand each direction the row tiles its recurrence reaches next).  Scalar loads (constant address space):
    // nothing here may enter the vector-memory counter, whose waits below are counted by hand.
    typedef const __attribute__((address_space(4))) int *cint_t;
    cint_t lst = nullptr;
    if (a.ws_tiles) {
        const int d = (a.ws_dirs == 2 && n_tile >= nt / 2) ? 1 : 0;
        lst = (cint_t)(unsigned long long)(a.ws_tiles + a.ws_off[d]);
        mt = a.ws_len[d];
    }
    auto rowtile = [&](int c) { const int i = grp + GROUPS * c; return lst ? lst[i] : i; };
    const int my_tiles = mt > grp ? (mt - grp + GROUPS - 1) / GROUPS : 0;
    if (tid == 0) {
        qword[0] = (int)atomicAdd(ctr, 1u);
        qword[1] = (int)atomicAdd(ctr, 1u);
    }
    __syncthreads();
    int c_cur = __builtin_amdgcn_readfirstlane(qword[0]), c_nxt = __builtin_amdgcn_readfirstlane(qword[1]);
    // (queue indices are compared UNSIGNED everywhere: a word that is not a tile index -- negative included -- can never reach rowtile())
    if ((unsigned)c_cur >= (unsigned)my_tiles) return;   // nothing left for this workgroup (no DMA issued yet)
    // the row tiles behind the two queue indices (a queue index past the end reads as the current tile: its DMAs re-read that tile)
    int r_cur = __builtin_amdgcn_readfirstlane(rowtile(c_cur));
    int r_nxt = (unsigned)c_nxt < (unsigned)my_tiles ? __builtin_amdgcn_readfirstlane(rowtile(c_nxt)) : r_cur;

    // ---- the wave's W fragments for the whole K: lane (fr, fh) holds W[n_tile * 128 + wave * 32 + fr][16 kb + 8 fh .. + 8] of each plane
    static_assert(NPROD == 3 || NPROD == 4, "three or four products");
    f16x8 w0[NKB], w1[NKB], w2[NPROD == 4 ? NKB : 1];
    {
        const size_t wplane = (size_t)((a.N + 127) / 128) * NKB * SLAB;
        const unsigned short *wb = a.Wsplit16 + (size_t)n_tile * NKB * SLAB + (wave * 32 + fr) * 16 + fh * 8;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            w0[kb] = *reinterpret_cast<const f16x8 *>(wb + (size_t)kb * SLAB);
            w1[kb] = *reinterpret_cast<const f16x8 *>(wb + wplane + (size_t)kb * SLAB);
            if constexpr (NPROD == 4) w2[kb] = *reinterpret_cast<const f16x8 *>(wb + 2 * wplane + (size_t)kb * SLAB);
        }
    }
    // Transposed MFMA blocks (see (4)): accumulator register 4 q + j of lane (fr, fh) = row fr of the row block, column 8 q + 4 fh + j of
    // the wave's 32 columns.  bias per accumulator register:
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = a.bias ? a.bias[n_tile * 128 + wave * 32 + 8 * (r >> 2) + 4 * fh + (r & 3)] : 0.f;
    const float wscale = a.wscale;
    const unsigned q_addr = (unsigned)(size_t)(lptr_t)qword;   // LDS byte address of the tile-queue word

    // ---- DMA plan: per k-block TWO wave-instructions per wave, rows [32 wave, 32 wave + 32) of the hi and of the lo slab.  A lane
    //      fetches the chunk that belongs at its LDS slot (slot s of row r holds chunk s ^ ((r >> 3) & 1): the fragment reads are
    //      then conflict-free, as in gemm_f16p.hip)
    const int ra = tid >> 1;
    const unsigned off = (unsigned)((ra * 2 + ((tid & 1) ^ ((ra >> 3) & 1))) * 8);
    const unsigned short *a_hi = a.Ah + off, *a_lo = a.Al + off;
    auto issue = [&](int stage, int rt, int kb) {
        const size_t t = (size_t)rt * NKB + kb;   // slab index of (row tile, k-block)
        unsigned short *img = lds + stage * STAGE + wave * 512;
        __builtin_amdgcn_global_load_lds((gptr_t)(a_hi + t * SLAB), (lptr_t)(img), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(a_lo + t * SLAB), (lptr_t)(img + SLAB), 16, 0, 0);
    };
    // (a queue index past the end still gets its DMAs -- the vmcnt counts below are static --: r_nxt then names the current tile)

    // Fragment reads.  Lane (fr, fh) of row block i reads row 32 i + fr, chunk fh ^ ((fr >> 3) & 1): one per-lane base address, the
    // rest is an immediate offset (stage, plane, row block).  They are INLINE ASM with hand-counted lgkmcnt waits: beside LDS-DMA
    // hipcc waits lgkmcnt(0) before every use of an LDS result -- i.e. also for the reads it has just issued for the NEXT k-block,
    // whose latency then sits in front of the MFMAs once per k-block.  (Nothing else in the loop touches LDS except the tile-queue
    // word, whose statements drain the counter themselves.)
    const unsigned f_base = (unsigned)(size_t)(lptr_t)lds + (unsigned)((fr * 16 + ((fh ^ ((fr >> 3) & 1)) * 8)) * 2);
    f16x8 ah[2][4], al[2][4];
#define UVAD_WS_READ(dst, byte_off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(f_base), "n"(byte_off))
    // (a macro: clang does not capture variables that a generic lambda names only in inline-asm operands)
#define UVAD_WS_READ_FRAGS(set, stage)                                                                                       \
    {                                                                                                                      \
        constexpr int base_ = (stage) * (STAGE * 2);                                                                       \
        UVAD_WS_READ(ah[set][0], base_ + 0 * 1024); UVAD_WS_READ(ah[set][1], base_ + 1 * 1024);                            \
        UVAD_WS_READ(ah[set][2], base_ + 2 * 1024); UVAD_WS_READ(ah[set][3], base_ + 3 * 1024);                            \
        UVAD_WS_READ(al[set][0], base_ + SLAB * 2 + 0 * 1024); UVAD_WS_READ(al[set][1], base_ + SLAB * 2 + 1 * 1024);      \
        UVAD_WS_READ(al[set][2], base_ + SLAB * 2 + 2 * 1024); UVAD_WS_READ(al[set][3], base_ + SLAB * 2 + 3 * 1024);      \
    }
    static_assert((NST - 1) * STAGE * 2 + SLAB * 2 + 3 * 1024 < 65536, "DS offset field is 16 bits");

    // ---- prologue: k-blocks 0 .. NST-2 of the stream (the current tile, then the next) in flight; k-block 0 in registers
#pragma unroll
    for (int v = 0; v < NST - 1; ++v) issue(v, v < NKB ? r_cur : r_nxt, v % NKB);
    wait_vm<2 * (NST - 2)>();
    __builtin_amdgcn_s_barrier();
    UVAD_WS_READ_FRAGS(0, 0)

    f32x16 hi[4], lo[4];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- the finished tile.  Staging (per wave): lane (fr, fh) writes accumulator quad q of row block i at row 32 i + fr, floats
    //      8 q + 4 fh .. + 3.  Piece p (p = 0 .. 15) = rows 8 p .. 8 p + 7: lane l reads row 8 p + (l >> 3), floats 4 (l & 7) .. + 3 and
    //      stores them at the same place of the gate matrix: [128-row tile][64-column tile][128][64] f32 (g_index), this wave's 32
    //      columns being one half of a 64-column tile -- eight full 128-byte lines per store instruction.
    float *sg_w = reinterpret_cast<float *>(lds + NST * STAGE + wave * SG_WAVE) + fr * SG_LD + 4 * fh;
    const unsigned sg_r = (unsigned)(size_t)(lptr_t)(lds + NST * STAGE + wave * SG_WAVE) + (unsigned)(((lane >> 3) * SG_LD + (lane & 7) * 4) * 4);
    const size_t g_lane = (size_t)(lane >> 3) * 64 + (wave & 1) * 32 + (lane & 7) * 4;
    const size_t g_ntile = (size_t)n_tile * 2 + (wave >> 1);
    const int n64 = a.N / 64;
    float *gprev = nullptr;                       // where the staged tile goes
    f32x4 pc[NPIECE];                            // pieces in flight between their ds_read (k-block kb) and their store (k-block kb + 1)
#define UVAD_WS_READ_PIECE(p) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pc[p]) : "v"(sg_r), "n"((p) * 8 * SG_LD * 4))
    static_assert(15 * 8 * SG_LD * 4 < 65536, "DS offset field is 16 bits");
    static_assert(piece_begin<NKB>(1) <= 6 && NPIECE - piece_begin<NKB>(NKB - 2) <= 6, "at most six pieces per k-block (UVAD_WS_PIECES)");

    // One 128-row tile.  FIRST: the workgroup's first tile (no stores of a previous tile among the outstanding operations).
    auto tile = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // the weight fragments live in AGPRs (the MFMAs read them in place) and nothing else does: constraint only, no instruction
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            asm volatile("" : "+a"(w0[kb]), "+a"(w1[kb]));
            if constexpr (NPROD == 4) asm volatile("" : "+a"(w2[kb]));
        }
        float *gout = a.C + ((size_t)r_cur * n64 + g_ntile) * (128 * 64) + g_lane;
        unsigned pulled = 0;
        (void)pc; (void)sg_r;   // (named outside inline asm too, so that the generic lambdas below capture them)
        static_for<NKB>([&](auto kb_tag) __attribute__((always_inline)) {
            constexpr int kb = decltype(kb_tag)::value;
            constexpr int s = kb & 1, sn = (kb + 1) & 1, stn = (kb + 1) % NST;   // fragment set in use / being filled, stage being read
            // One wave per SIMD issues everything itself, and the matrix pipe idles while it issues anything else for long.  So the
            // k-block is ONE pinned instruction stream (every group is fenced with sched_barrier(0); hipcc otherwise sinks the reads to the
            // end and hoists MFMAs over the waits): wait + barrier, then eight MFMA pairs with a GAP after each that takes the wave's
            // other work: gap 0 the stores of the previous tile's pieces and the first DMA piece, gap 1 the second DMA piece, gaps 1 - 4
            // two fragment reads each, gap 5 the reads of the staged pieces.  (Assigning the stores and DMA pieces to different gaps
            // per wave, so that the four waves do not hit the CU's memory path at the same instant, was measured: 0.56 ms against 0.47.)
            // The MFMAs use the fragment set read during the PREVIOUS k-block; the DMA refills the slot of k-block kb - 1; the reads
            // fetch k-block kb + 1 (the next tile's first one at the end).
            // MFMA j of the k-block: A operand = the weight fragment, B operand = the activation fragment, so the 32 x 32 block comes
            // out transposed (a lane holds four consecutive columns of one row); same products and k order as gemm_f16p_kernel:
            //   j = 0..3  hi[i] += P0 . a_hi     4..7  lo[i] += P1 . a_hi     8..11  lo[i] += P0 . a_lo     12..15  lo[i] += P2 . a_hi
            // The last k-block of a tile runs them row block by row block (j = i, 4 + i, 8 + i, 12 + i) so that a finished row block can be
            // staged while the next ones are multiplied.
            constexpr bool LAST = kb == NKB - 1;
            auto M = [&](auto j_tag) __attribute__((always_inline)) {
                constexpr int j = decltype(j_tag)::value, i = j & 3, p = j >> 2;
                if constexpr (p == 0) hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kb], ah[s][i], kb == 0 ? zero : hi[i], 0, 0, 0);
                if constexpr (p == 1) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kb], ah[s][i], kb == 0 ? zero : lo[i], 0, 0, 0);
                if constexpr (p == 2) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kb], al[s][i], lo[i], 0, 0, 0);
                if constexpr (p == 3 && NPROD == 4) lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[kb], ah[s][i], lo[i], 0, 0, 0);   // (NPROD == 3: these slots of the stream stay empty)
            };
            // (hi + lo * 2^-11) * 2^-S + bias into the wave's staging image, four ds_write_b128 per row block.  (The image's previous
            // content was read during k-blocks 0 .. NKB - 2 of this tile: one wave's LDS operations execute in order.)
            auto stage_block = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 o;
                    o.x = __builtin_fmaf(__builtin_fmaf(lo[i][4 * q + 0], 0.00048828125f, hi[i][4 * q + 0]), wscale, bias[4 * q + 0]);
                    o.y = __builtin_fmaf(__builtin_fmaf(lo[i][4 * q + 1], 0.00048828125f, hi[i][4 * q + 1]), wscale, bias[4 * q + 1]);
                    o.z = __builtin_fmaf(__builtin_fmaf(lo[i][4 * q + 2], 0.00048828125f, hi[i][4 * q + 2]), wscale, bias[4 * q + 2]);
                    o.w = __builtin_fmaf(__builtin_fmaf(lo[i][4 * q + 3], 0.00048828125f, hi[i][4 * q + 3]), wscale, bias[4 * q + 3]);
                    *reinterpret_cast<float4 *>(sg_w + i * (32 * SG_LD) + 8 * q) = o;
                }
            };
#define UVAD_WS_M(j) M(std::integral_constant<int, (j)>{})
#define UVAD_WS_SB __builtin_amdgcn_sched_barrier(0)
#define UVAD_WS_PIECE_IF(kb_, t_) \
    if constexpr (piece_begin<NKB>(kb_) + (t_) < piece_begin<NKB>((kb_) + 1)) UVAD_WS_READ_PIECE(piece_begin<NKB>(kb_) + (t_));
#define UVAD_WS_STORE_IF(t_)                                                                            \
    if constexpr (piece_begin<NKB>(kb - 1) + (t_) < piece_begin<NKB>(kb))                               \
        __builtin_nontemporal_store(pc[piece_begin<NKB>(kb - 1) + (t_)], reinterpret_cast<f32x4 *>(gprev + (size_t)(piece_begin<NKB>(kb - 1) + (t_)) * (8 * 64)));
            // (1) the wave's own pieces of k-block kb + 1 have landed: everything but what was issued after them may still be in flight --
            //     the DMAs of the NST - 3 k-blocks in between and the stores of the previous tile's pieces issued in THIS tile's k-blocks
            //     in that window (stores of the tile before are not counted, nor is anything a wave issues later in the k-block of the
            //     awaited DMA: the wait is then stricter than needed, never looser).  lgkmcnt(0): the fragments and pieces read during
            //     the previous k-block.
            UVAD_WS_SB;
            wait_vm_lgkm0<2 * (NST - 3) + (FIRST ? 0 : stores_in<NKB>(kb - (NST - 3), kb))>();
            __builtin_amdgcn_s_barrier();   // ... and everyone's; every wave has k-block kb's fragments in registers, so the slot of kb - 1 is free
            UVAD_WS_SB;
            constexpr int v = kb - 1 + NST;                       // (2) the k-block of the stream that refills that slot
            const int vr = v < NKB ? r_cur : r_nxt;
            const size_t vt = ((size_t)vr * NKB + v % NKB) * SLAB;
            unsigned short *img = lds + ((kb + NST - 1) % NST) * STAGE + wave * 512;
            constexpr int fb = stn * (STAGE * 2);                 // (3) where the fragments of k-block kb + 1 are
            // what goes into gap g (after MFMA pair g)
#define UVAD_WS_GAP(g)                                                                                                           \
    {                                                                                                                            \
        if constexpr (!FIRST && kb >= 1 && (g) == 0) {   /* the pieces read during k-block kb - 1 leave */                          \
            UVAD_WS_STORE_IF(0) UVAD_WS_STORE_IF(1) UVAD_WS_STORE_IF(2) UVAD_WS_STORE_IF(3) UVAD_WS_STORE_IF(4) UVAD_WS_STORE_IF(5)          \
        }                                                                                                                        \
        if constexpr ((g) == 0) __builtin_amdgcn_global_load_lds((gptr_t)(a_hi + vt), (lptr_t)(img), 16, 0, 0);                   \
        if constexpr ((g) == 1) __builtin_amdgcn_global_load_lds((gptr_t)(a_lo + vt), (lptr_t)(img + SLAB), 16, 0, 0);            \
        if constexpr ((g) == 1) { UVAD_WS_READ(ah[sn][0], fb + 0 * 1024); UVAD_WS_READ(ah[sn][1], fb + 1 * 1024); }              \
        if constexpr ((g) == 2) { UVAD_WS_READ(ah[sn][2], fb + 2 * 1024); UVAD_WS_READ(ah[sn][3], fb + 3 * 1024); }              \
        if constexpr ((g) == 3) { UVAD_WS_READ(al[sn][0], fb + SLAB * 2 + 0 * 1024); UVAD_WS_READ(al[sn][1], fb + SLAB * 2 + 1 * 1024); } \
        if constexpr ((g) == 4) { UVAD_WS_READ(al[sn][2], fb + SLAB * 2 + 2 * 1024); UVAD_WS_READ(al[sn][3], fb + SLAB * 2 + 3 * 1024); } \
        if constexpr ((g) == 5 && !FIRST && !LAST) {   /* this k-block's share of the previous tile's staged pieces */              \
            UVAD_WS_PIECE_IF(kb, 0) UVAD_WS_PIECE_IF(kb, 1) UVAD_WS_PIECE_IF(kb, 2) UVAD_WS_PIECE_IF(kb, 3) UVAD_WS_PIECE_IF(kb, 4) UVAD_WS_PIECE_IF(kb, 5) \
        }                                                                                                                        \
        if constexpr (LAST && (g) == 2) stage_block(0);                                                                          \
        if constexpr (LAST && (g) == 4) stage_block(1);                                                                          \
        if constexpr (LAST && (g) == 6) stage_block(2);                                                                          \
        if constexpr (LAST && (g) == 7) stage_block(3);                                                                          \
        UVAD_WS_SB;                                                                                                              \
    }
            // Thread 0 draws the tile after the next one.  Inline asm with the EXEC mask set inside the statement (no branch around it, so
            // no phi / copy of its result register): the compiler's vmcnt bookkeeping does not see the atomic and never waits for it.
            // Its result is read at kb = NST - 2, behind that k-block's vmcnt wait.
            if (kb == 0) {
                unsigned long long sv;
                asm volatile("v_cmp_eq_u32_e32 vcc, 0, %2\n\ts_and_saveexec_b64 %1, vcc\n\tglobal_atomic_add %0, %3, %4, off sc0\n\ts_mov_b64 exec, %1"
                             : "=&v"(pulled), "=&s"(sv) : "v"(tid), "v"(ctr), "v"(1u) : "vcc", "memory");
                UVAD_WS_SB;
            }
            if (kb == NST - 2) {
                // the atomic was issued before the DMAs of k-blocks 1 .. NST-3 of this tile, which the wait above has left as the only
                // operations that may still be in flight: its result is in the register.  Published by the barrier of kb + 1.
                unsigned long long sv;
                asm volatile("v_cmp_eq_u32_e32 vcc, 0, %1\n\ts_and_saveexec_b64 %0, vcc\n\tds_write_b32 %2, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %0"
                             : "=&s"(sv) : "v"(tid), "v"(q_addr), "v"(pulled) : "vcc", "memory");
                UVAD_WS_SB;
            }
            if constexpr (!LAST) {
                UVAD_WS_M(0); UVAD_WS_M(1); UVAD_WS_SB; UVAD_WS_GAP(0)
                UVAD_WS_M(2); UVAD_WS_M(3); UVAD_WS_SB; UVAD_WS_GAP(1)
                UVAD_WS_M(4); UVAD_WS_M(5); UVAD_WS_SB; UVAD_WS_GAP(2)
                UVAD_WS_M(6); UVAD_WS_M(7); UVAD_WS_SB; UVAD_WS_GAP(3)
                UVAD_WS_M(8); UVAD_WS_M(9); UVAD_WS_SB; UVAD_WS_GAP(4)
                UVAD_WS_M(10); UVAD_WS_M(11); UVAD_WS_SB; UVAD_WS_GAP(5)
                UVAD_WS_M(12); UVAD_WS_M(13); UVAD_WS_SB; UVAD_WS_GAP(6)
                UVAD_WS_M(14); UVAD_WS_M(15); UVAD_WS_SB; UVAD_WS_GAP(7)
            } else {
                UVAD_WS_M(0); UVAD_WS_M(4); UVAD_WS_SB; UVAD_WS_GAP(0)
                UVAD_WS_M(8); UVAD_WS_M(12); UVAD_WS_SB; UVAD_WS_GAP(1)
                UVAD_WS_M(1); UVAD_WS_M(5); UVAD_WS_SB; UVAD_WS_GAP(2)
                UVAD_WS_M(9); UVAD_WS_M(13); UVAD_WS_SB; UVAD_WS_GAP(3)
                UVAD_WS_M(2); UVAD_WS_M(6); UVAD_WS_SB; UVAD_WS_GAP(4)
                UVAD_WS_M(10); UVAD_WS_M(14); UVAD_WS_SB; UVAD_WS_GAP(5)
                UVAD_WS_M(3); UVAD_WS_M(7); UVAD_WS_SB; UVAD_WS_GAP(6)
                UVAD_WS_M(11); UVAD_WS_M(15); UVAD_WS_SB; UVAD_WS_GAP(7)
            }
        });
        // the queue word written at kb = NST - 2 was published by the barrier of kb = NST - 1 <= NKB - 1
        int c_nn;   // (a generic-pointer read would be a FLAT load, which hipcc guards with vmcnt(0): the DMA ring would drain once per tile)
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c_nn) : "v"(q_addr) : "memory");
        gprev = gout;
        c_cur = c_nxt;
        r_cur = r_nxt;
        c_nxt = __builtin_amdgcn_readfirstlane(c_nn);
        r_nxt = (unsigned)c_nxt < (unsigned)my_tiles ? __builtin_amdgcn_readfirstlane(rowtile(c_nxt)) : r_cur;
    };
    tile(std::true_type{});
    // (the counter hands every tile index out once, so a workgroup can never run more than my_tiles tiles: the explicit bound is
    // the exit condition every wave reaches whatever the queue words hold)
    for (int done = 1; done < my_tiles && (unsigned)c_cur < (unsigned)my_tiles; ++done) tile(std::false_type{});
    // the last tile's image
    static_for<NPIECE>([&](auto t) __attribute__((always_inline)) {
        constexpr int p = decltype(t)::value;
        f32x4 o;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(o) : "v"(sg_r), "n"(p * 8 * SG_LD * 4) : "memory");
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(gprev + (size_t)p * (8 * 64)));
    });
    wait_vm<0>();   // no LDS-DMA may still be in flight when the workgroup's LDS is handed to another one
}

}  // namespace

size_t gemm_f16p_ws_counter_bytes() { return (size_t)GROUPS * 8 * 32 * sizeof(unsigned); }

// true if the shape is one this kernel handles and the launch is large enough to be worth a persistent grid
bool gemm_f16p_ws_supported(const GemmArgs &a, int n_cu) {
    if (a.out_planes || !a.c_blocked || a.act != 0 || a.N % 128 != 0 || a.N / 128 > 8) return false;
    if (a.K != 64 && a.K != 96 && a.K != 128 && a.K != 256) return false;
    const long mt = (a.M + 127) / 128;
    return mt * (a.N / 128) >= 2L * (n_cu > 0 ? n_cu : 256);
}

hipError_t launch_gemm_f16p_ws(const GemmArgs &a, unsigned *counters, int n_cu, hipStream_t s) {
    if (!gemm_f16p_ws_supported(a, n_cu) || !counters || !a.Ah || !a.Al || !a.Wsplit16 || !a.C || a.ldw != a.K) return hipErrorInvalidValue;
    const int mt = (a.M + 127) / 128, nt = a.N / 128;
    if (a.ws_tiles) {
        if ((a.ws_dirs != 1 && a.ws_dirs != 2) || (a.ws_dirs == 2 && nt % 2 != 0)) return hipErrorInvalidValue;
        for (int d = 0; d < a.ws_dirs; ++d)
            if (a.ws_off[d] < 0 || a.ws_len[d] < 0 || a.ws_len[d] > mt) return hipErrorInvalidValue;
    }
    hipError_t e = hipMemsetAsync(counters, 0, gemm_f16p_ws_counter_bytes(), s);
    if (e != hipSuccess) return e;
    // one workgroup per CU; every (group, column tile) pair must own at least one workgroup: a multiple of 8 * nt
    const int per = GROUPS * nt;
    int grid = (n_cu > 0 ? n_cu : 256) / per * per;
    if (grid < per) grid = per;
    if (a.products != 0 && a.products != 3 && a.products != 4) return hipErrorInvalidValue;
    const bool three = a.products == 3;
#define UVAD_WS_LAUNCH(NKB_)                                                                                                          \
    if (three) hipLaunchKernelGGL((gemm_f16p_ws_kernel<NKB_, 3>), dim3(grid), dim3(256), 0, s, a, mt, nt, counters);                   \
    else hipLaunchKernelGGL((gemm_f16p_ws_kernel<NKB_, 4>), dim3(grid), dim3(256), 0, s, a, mt, nt, counters);
    switch (a.K / 16) {
        case 4: UVAD_WS_LAUNCH(4) break;
        case 6: UVAD_WS_LAUNCH(6) break;
        case 8: UVAD_WS_LAUNCH(8) break;
        case 16: UVAD_WS_LAUNCH(16) break;
        default: return hipErrorInvalidValue;
    }
#undef UVAD_WS_LAUNCH
    return hipGetLastError();
}

}  // namespace uvad

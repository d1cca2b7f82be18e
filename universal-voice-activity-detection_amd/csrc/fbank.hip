// fbank.hip -- fused Kaldi-style log-mel front end, one launch, no intermediate in HBM.
// Stands behind lhotse's Fbank.extract_batch as the reference calls it
// (src/datasets/ami/utils.py:153,157-163; src/utils/helper.py:120-130): reflect-padded framing
// (snip_edges=False), per-frame DC removal, pre-emphasis, window, zero-pad to 512, rFFT,
// power, triangular mel projection, log(max(., eps)).  (The torch-op original materialises the
// (B,T,512) frame tensor and the complex spectrum in HBM: >5x the algorithmic traffic.)
//
// gfx950 design
//   * a workgroup (4 waves) owns FR_WG = 24 consecutive frames of one utterance: the
//     23*160+400 PCM samples they cover are loaded ONCE (16-byte coalesced loads, reflection
//     handled at the utterance edges) into an LDS tile; neighbouring workgroups re-read only the
//     240-sample overlap (6 %, absorbed by L2).
//   * a wave transforms TWO frames at a time as the real and imaginary part of one 512-point
//     complex FFT, 8 points per lane, as three radix-8 passes (512 = 8*8*8) held in registers;
//     the two re-distributions between passes go through a padded per-wave LDS scratch with
//     wave-local ordering only (no workgroup barrier in the loop).
//   * the two spectra are separated with the conjugate-symmetry identity, |.|^2 of frame A and of frame B go to LDS
//     (over the transpose scratch), and lane m accumulates mel filter m over its (start,len) band, two bins per
//     iteration; 64 lanes write 256 contiguous bytes per frame.
//   * every LDS access is a 32-bit operation (real and imaginary parts live in separate arrays): see the note at the
//     per-wave scratch in the kernel.
//   The kernel is bound by vector-instruction issue, not by HBM: ~650 instructions per lane and frame pair
//   (three radix-8 passes, two transposes, framing, mel) put the ceiling near 1.5 G frames/s = 17 % of the
//   8.9 G frames/s the HBM roofline would allow.
//   Algorithmic HBM bytes per frame: 640 read (320 for int16 PCM) + 4*n_mels written.
#include "uvad_internal.h"

namespace uvad {

namespace {

constexpr int NFFT = 512;
#ifndef UVAD_FB_PAIRS
#define UVAD_FB_PAIRS 3    // frame pairs per wave: 24 frames per workgroup (16 KiB PCM tile; with the scratch ~43 KiB of LDS -> 3 workgroups per CU)
#endif
#ifndef UVAD_FB_MINWAVES
#define UVAD_FB_MINWAVES 4   // <= 128 VGPRs (no spills): register occupancy never the limiter
#endif
constexpr int PAIRS_PER_WAVE = UVAD_FB_PAIRS;
constexpr int FR_WG = 4 * 2 * PAIRS_PER_WAVE;  // frames per workgroup
constexpr int ZB_LD = 9;                       // padded row (8 complex + 1) of the transpose scratch
constexpr int ZB_ELEMS = 64 * ZB_LD;           // 576 complex >= 512
constexpr int PB_ELEMS = 0;                    // the 257 (powerA, powerB) pairs reuse the first 264 slots of the transpose scratch

__device__ __forceinline__ void wave_lds_fence() {
    // Hand-off through the wave's own LDS scratch.  The hardware executes one wave's DS operations in
    // issue order, so a ds_read issued after a ds_write of the same wave sees the written data; all that
    // is needed is that the COMPILER keeps that order.  (A release/acquire fence pair here, even at
    // wavefront scope, made hipcc drain vmcnt(0) -- i.e. wait for the feature stores of the previous
    // frame pair to reach memory -- at every hand-off.)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Wave-wide sum as a wave-uniform value, on the VALU (DPP), not through the LDS crossbar: four
// xor-style steps give every lane its 16-lane row total, two row broadcasts accumulate the rows into
// lane 63, a readlane returns it.  (A ds_bpermute butterfly costs an LDS round trip per step.)
#define UVAD_DPP_ADD(V, CTRL, ROW_MASK)                                                                   \
    V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, ROW_MASK, 0xF, true))
__device__ __forceinline__ float wave_sum(float v) {
    UVAD_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    UVAD_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    UVAD_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
    UVAD_DPP_ADD(v, 0x140, 0xF);   // row_mirror: every lane holds its row total
    UVAD_DPP_ADD(v, 0x142, 0xA);   // row_bcast:15 into rows 1 and 3
    UVAD_DPP_ADD(v, 0x143, 0xC);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// forward 8-point DFT, in place: out[q] = sum_r in[r] * exp(-2*pi*i*r*q/8)
__device__ __forceinline__ void dft8(float (&re)[8], float (&im)[8]) {
    constexpr float R = 0.70710678118654752f;
    float ar[4], ai[4], br[4], bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ar[j] = re[j] + re[j + 4];
        ai[j] = im[j] + im[j + 4];
        br[j] = re[j] - re[j + 4];
        bi[j] = im[j] - im[j + 4];
    }
    // b_j *= W8^j
    {
        const float t1r = (br[1] + bi[1]) * R, t1i = (bi[1] - br[1]) * R;  // * (1 - i)/sqrt2
        br[1] = t1r; bi[1] = t1i;
        const float t2r = bi[2], t2i = -br[2];                              // * (-i)
        br[2] = t2r; bi[2] = t2i;
        const float t3r = (bi[3] - br[3]) * R, t3i = -(br[3] + bi[3]) * R; // * (-1 - i)/sqrt2
        br[3] = t3r; bi[3] = t3i;
    }
    // 4-point DFTs: even outputs from a, odd outputs from b
    {
        const float s0r = ar[0] + ar[2], s0i = ai[0] + ai[2];
        const float s1r = ar[0] - ar[2], s1i = ai[0] - ai[2];
        const float s2r = ar[1] + ar[3], s2i = ai[1] + ai[3];
        const float s3r = ai[1] - ai[3], s3i = -(ar[1] - ar[3]);           // (a1 - a3) * (-i)
        re[0] = s0r + s2r; im[0] = s0i + s2i;
        re[4] = s0r - s2r; im[4] = s0i - s2i;
        re[2] = s1r + s3r; im[2] = s1i + s3i;
        re[6] = s1r - s3r; im[6] = s1i - s3i;
    }
    {
        const float s0r = br[0] + br[2], s0i = bi[0] + bi[2];
        const float s1r = br[0] - br[2], s1i = bi[0] - bi[2];
        const float s2r = br[1] + br[3], s2i = bi[1] + bi[3];
        const float s3r = bi[1] - bi[3], s3i = -(br[1] - br[3]);
        re[1] = s0r + s2r; im[1] = s0i + s2i;
        re[5] = s0r - s2r; im[5] = s0i - s2i;
        re[3] = s1r + s3r; im[3] = s1i + s3i;
        re[7] = s1r - s3r; im[7] = s1i - s3i;
    }
}

template <bool I16>
__device__ __forceinline__ float pcm_at(const void *row, int64_t i) {
    if (I16) return (float)reinterpret_cast<const int16_t *>(row)[i] * (1.0f / 32768.0f);
    return reinterpret_cast<const float *>(row)[i];
}


template <bool I16>
__global__ __launch_bounds__(256, UVAD_FB_MINWAVES) void fbank_kernel(FbankArgs a, const float2 *__restrict__ tw512) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.frame_len, sh = a.frame_shift, F = a.n_mels;
    const int raw_pad = (((FR_WG - 1) * sh + L) + 3) & ~3;
    const int mel_pairs = (a.tab.mel_stride + 1) / 2;          // bins per filter rounded up to pairs (zero weights in the padding)
    const int melw_pad = (2 * mel_pairs * F + 3) & ~3;
    float *raw = smem;
    float *melw = raw + raw_pad;
    float2 *wscr = reinterpret_cast<float2 *>(melw + melw_pad);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * FR_WG;
    const int nfr = (int)((a.T - t0) < FR_WG ? (a.T - t0) : FR_WG);
    if (b >= a.B) {   // plane output only (the grid covers whole sequence tiles): rows of a padding sequence read as zero
        for (int i = tid; i < nfr * a.plane_w; i += 256) {
            const size_t row = ((size_t)(b >> 2) * a.T + (size_t)(t0 + i / a.plane_w)) * SEQ_TILE + (b & 3);
            const size_t o = plane_index(row, i % a.plane_w, a.plane_w);
            a.plane_hi[o] = 0;
            a.plane_lo[o] = 0;
        }
        return;
    }
    const int n_left = a.snip_edges ? 0 : (L - sh) / 2;
    const int64_t s0 = t0 * sh - n_left;
    const int need = (nfr - 1) * sh + L;
    const size_t rstride = a.row_stride ? (size_t)a.row_stride : (size_t)a.S;
    const void *xrow = I16 ? (const void *)(reinterpret_cast<const int16_t *>(a.pcm) + (size_t)b * rstride)
                           : (const void *)(reinterpret_cast<const float *>(a.pcm) + (size_t)b * rstride);

    // streaming rows are virtual (FbankArgs::vs_*): sample p of row b's [tail | chunk] image, as stream_stage_kernel would have staged it
    auto vs_abs = [&](int p) -> float {
        if (p >= a.vs_tail) return a.vs_chunk[(size_t)b * a.vs_chunk_len + (p - a.vs_tail)];
        if (a.vs_first) {   // absolute sample index p - tail in [-tail, 0): mirror of the chunk's head (including the edge sample), zero beyond
            const int m = a.vs_tail - p;
            return (m <= a.vs_n_left && m - 1 < a.vs_chunk_len) ? a.vs_chunk[(size_t)b * a.vs_chunk_len + (m - 1)] : 0.0f;
        }
        return a.vs_tail_in[(size_t)b * a.vs_tail + p];
    };
    const bool virt = !I16 && a.vs_chunk != nullptr;
    if (virt && blockIdx.x == 0)   // the tail the next step starts from
        for (int q = tid; q < a.vs_tail; q += 256) a.vs_tail_out[(size_t)b * a.vs_tail + q] = vs_abs(a.vs_chunk_len + q);

    // ---- stage the PCM tile (reflect at the utterance edges) --------------------------------
    // All loads of the tile are issued before the first LDS write (ST_IT chunks of 4 samples per
    // thread in flight): a load -> wait -> store loop would pay the HBM latency once per chunk.
    constexpr int ST_IT = ((FR_WG - 1) * 160 + 400 + 1023) / 1024;   // one round covers the tile at the reference geometry (25 ms / 10 ms)
    for (int base = 0; base < need; base += ST_IT * 1024) {
        float4 v[ST_IT];
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            const int i = base + it * 1024 + tid * 4;
            const int64_t g = s0 + i;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < need) {
                bool fast = !virt && g >= 0 && g + 3 < a.S;
                if (I16) {
                    const int16_t *p = reinterpret_cast<const int16_t *>(xrow) + g;
                    fast = fast && (reinterpret_cast<uintptr_t>(p) & 7) == 0;
                    if (fast) {
                        const short4 q = *reinterpret_cast<const short4 *>(p);
                        v[it] = make_float4((float)q.x * (1.0f / 32768.0f), (float)q.y * (1.0f / 32768.0f),
                                            (float)q.z * (1.0f / 32768.0f), (float)q.w * (1.0f / 32768.0f));
                    }
                } else {
                    const float *p = reinterpret_cast<const float *>(xrow) + g;
                    fast = fast && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
                    if (fast) v[it] = *reinterpret_cast<const float4 *>(p);
                }
                if (!fast) {
                    float e4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int64_t idx = g + e;
                        if (idx < 0) idx = -idx - 1;
                        if (idx >= a.S) idx = 2 * a.S - 1 - idx;
                        if (idx < 0) idx = 0;
                        if (idx >= a.S) idx = a.S - 1;
                        e4[e] = virt ? vs_abs((int)idx + a.vs_offset) : pcm_at<I16>(xrow, idx);
                    }
                    v[it] = make_float4(e4[0], e4[1], e4[2], e4[3]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            const int i = base + it * 1024 + tid * 4;
            if (i < need) *reinterpret_cast<float4 *>(raw + i) = v[it];   // raw is padded to a multiple of 4
        }
    }
    // mel weights as [bin-in-band][filter] (transposed and zero-padded to whole pairs on the host), so lane m reads conflict-free
    for (int i = tid; i < 2 * mel_pairs * F; i += 256) melw[i] = a.tab.mel_wt[i];

    // ---- per-lane constants -------------------------------------------------------------------
    float win[8];
    float2 tw1[8], tw2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        win[r] = n < L ? a.tab.window[n] : 0.0f;
        tw1[r] = tw512[(lane * r) & (NFFT - 1)];
        tw2[r] = tw512[(8 * (lane & 7) * r) & (NFFT - 1)];
    }
    const int nfilt_pass = (F + 63) / 64;
    // band starts of this lane's filters (passes 0 and 1 cover n_mels <= 128): loaded ONCE per workgroup.
    // Inside the frame loop a global load would also make hipcc drain vmcnt(0), i.e. wait for the feature
    // stores of the previous frame pair, every pair.
    const int mst0 = a.tab.mel_start[lane < F ? lane : F - 1];
    int mst1 = a.tab.mel_start[lane + 64 < F ? lane + 64 : F - 1];
    int mst0v = mst0;
    // Retire every global load issued so far BEFORE the frame loop: a register that is still "in flight"
    // at loop entry makes hipcc put s_waitcnt vmcnt(0) at its first use INSIDE the loop, where it then
    // also waits for the feature stores of the previous frame pair on every iteration.
#pragma unroll
    for (int r = 0; r < 8; ++r)
        asm volatile("" : "+v"(win[r]), "+v"(tw1[r].x), "+v"(tw1[r].y), "+v"(tw2[r].x), "+v"(tw2[r].y));
    asm volatile("" : "+v"(mst0v), "+v"(mst1));
    __syncthreads();
    // Per-wave scratch as TWO 4-byte arrays (real / imaginary parts of the transform; later the power of frame A / frame B), never
    // as float2 pairs: every LDS instruction of this kernel is a 32-bit one (ds_read_b32 / ds_read2_b32 / ds_write2_b32 ...).
    // With 64-bit LDS operations (ds_read_b64, ds_write2st64_b64, ...) the kernel returned wrong frames whenever its waves
    // shared a CU with waves of another kernel that loop over MFMAs, an LDS read and s_barrier (a GEMM of another step in
    // flight) -- the same happens to stock rocFFT, and the 32-bit form is immune (DESIGN.md section 3.3,
    // profiles/r02_concurrency_corruption.json).  It is also 3 % faster.
    float *zr = reinterpret_cast<float *>(wscr + (size_t)wave * (ZB_ELEMS + PB_ELEMS)), *zi = zr + ZB_ELEMS;
#define ZB_PUT(idx, a, b) { zr[idx] = (a); zi[idx] = (b); }
#define ZB_GET(idx) make_float2(zr[idx], zi[idx])
    const float inv_len = 1.0f / (float)L;

    for (int q = 0; q < PAIRS_PER_WAVE; ++q) {
        const int fa = 2 * (q * 4 + wave);
        if (fa >= nfr) break;  // wave-uniform
        const bool has_b = fa + 1 < nfr;
        const float *xa = raw + fa * sh;
        const float *xb = raw + (has_b ? fa + 1 : fa) * sh;

        // ---- framing, DC removal, pre-emphasis, window: lane p owns n = p + 64 r ---------------
        float re[8], im[8], pa[8], pbv[8];
        float suma = 0.f, sumb = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // unconditional LDS reads (index clamped into the frame); samples past the frame are
            // zeroed by a select for the mean and by the zero window tap for the spectrum
            const int n = lane + 64 * r;
            const bool in = n < L;
            const int nc = in ? n : L - 1;
            const int np = nc > 0 ? nc - 1 : 0;
            const float va = xa[nc], vpa = xa[np], vb = xb[nc], vpb = xb[np];
            re[r] = in ? va : 0.f;
            pa[r] = vpa;
            im[r] = in ? vb : 0.f;
            pbv[r] = vpb;
            suma += re[r];
            sumb += im[r];
        }
        float mua = 0.f, mub = 0.f;
        if (a.remove_dc) {
            mua = wave_sum(suma) * inv_len;
            mub = wave_sum(sumb) * inv_len;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            re[r] = ((re[r] - mua) - a.preemph * (pa[r] - mua)) * win[r];
            im[r] = has_b ? ((im[r] - mub) - a.preemph * (pbv[r] - mub)) * win[r] : 0.f;
        }
        // ---- pass 1: DFT over n1 (n = 64 n1 + p), twiddle W512^(p k1) ----------------------------
        dft8(re, im);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float r0 = re[k] * tw1[k].x - im[k] * tw1[k].y;
            const float i0 = re[k] * tw1[k].y + im[k] * tw1[k].x;
            re[k] = r0; im[k] = i0;
        }
        // transpose: lane p = 8a + b -> element k1 goes to row (k1, b), column a
        {
            const int aa = lane >> 3, bb = lane & 7;
#pragma unroll
            for (int k = 0; k < 8; ++k) ZB_PUT((k * 8 + bb) * ZB_LD + aa, re[k], im[k]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = ZB_GET(lane * ZB_LD + k);
            re[k] = v.x; im[k] = v.y;
        }
        // ---- pass 2: lane (k1, b): DFT over a -> c, twiddle W64^(b c) -----------------------------
        dft8(re, im);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float r0 = re[k] * tw2[k].x - im[k] * tw2[k].y;
            const float i0 = re[k] * tw2[k].y + im[k] * tw2[k].x;
            re[k] = r0; im[k] = i0;
        }
        wave_lds_fence();  // all reads of the scratch are issued before it is rewritten
        {
            const int k1 = lane >> 3, bb = lane & 7;
#pragma unroll
            for (int c = 0; c < 8; ++c) ZB_PUT((k1 * 8 + c) * ZB_LD + bb, re[c], im[c]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = ZB_GET(lane * ZB_LD + k);
            re[k] = v.x; im[k] = v.y;
        }
        // ---- pass 3: lane (k1, c): DFT over b -> d; Z[k1 + 8c + 64d] -------------------------------
        dft8(re, im);
        wave_lds_fence();
        {
            const int lam = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
            for (int d = 0; d < 8; ++d) ZB_PUT(lam + 64 * d, re[d], im[d]);
        }
        wave_lds_fence();
        // ---- split the two real spectra, power: A = (Z[k] + conj Z[N-k])/2, B = (Z[k] - conj Z[N-k])/(2i)
        // (all reads first, then the power pairs overwrite the scratch)
        {
            float2 pw[5];
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                const int k = lane + 64 * d;
                const int kc = d < 4 ? k : (lane == 0 ? 256 : 0);
                const float2 z = ZB_GET(kc);
                const float2 w = ZB_GET((NFFT - kc) & (NFFT - 1));
                const float ar = z.x + w.x, ai = z.y - w.y;
                const float br = z.y + w.y, bi = z.x - w.x;
                pw[d] = make_float2(0.25f * (ar * ar + ai * ai), 0.25f * (br * br + bi * bi));
            }
            wave_lds_fence();
#pragma unroll
            for (int d = 0; d < 5; ++d)
                if (d < 4 || lane == 0) ZB_PUT(lane + 64 * d, pw[d].x, pw[d].y);
        }
        wave_lds_fence();
        // ---- mel band sums + log; lane = filter ----------------------------------------------------
        // Uniform trip count (the longest band, zero-padded weights), two bins per LDS instruction: the weights of bins i, i+1
        // are F floats apart (ds_read2_b32), their power pairs adjacent (ds_read2_b64, no alignment requirement).  No clamp of
        // the bin index: a band that runs past bin 256 reads this pair's own spectrum values from the scratch (finite) against
        // zero weights.
        for (int ps = 0; ps < nfilt_pass; ++ps) {
            const int m = lane + 64 * ps;
            const int mm = m < F ? m : F - 1;
            const int st = ps == 0 ? mst0v : mst1;   // n_mels <= 128 (checked by uvad_create): no global load in this loop
            float ea = 0.f, eb = 0.f, ea2 = 0.f, eb2 = 0.f;
            const float *wp = melw + mm;
            for (int i = 0; i < mel_pairs; ++i) {
                const float w0 = wp[(2 * i) * F], w1 = wp[(2 * i + 1) * F];
                const float2 p0 = ZB_GET(st + 2 * i), p1 = ZB_GET(st + 2 * i + 1);
                ea = __builtin_fmaf(w0, p0.x, ea);
                eb = __builtin_fmaf(w0, p0.y, eb);
                ea2 = __builtin_fmaf(w1, p1.x, ea2);
                eb2 = __builtin_fmaf(w1, p1.y, eb2);
            }
            ea += ea2;
            eb += eb2;
            // ocml logf (<= 1 ulp), not the 2-ulp-of-log2 __logf: two per lane and frame pair, nothing next to the FFT
            if (a.plane_hi) {
                // straight into the operand planes of the first projection GEMM (gemm_f16p.hip: x ~= hi + lo * 2^-11, K-blocked, rows in
                // tile-major order m = (tile * T + t) * 4 + j): the f32 feature tensor never exists.  Columns [F, plane_w) are zero.
                if (m < a.plane_w) {
                    const float va = m < F ? logf(fmaxf(ea, a.log_floor)) : 0.f, vb = m < F ? logf(fmaxf(eb, a.log_floor)) : 0.f;
                    const size_t row = ((size_t)(b >> 2) * a.T + (size_t)(t0 + fa)) * SEQ_TILE + (b & 3);
                    const size_t oa = plane_index(row, m, a.plane_w);
                    const _Float16 ha = (_Float16)va;
                    a.plane_hi[oa] = __builtin_bit_cast(unsigned short, ha);
                    a.plane_lo[oa] = __builtin_bit_cast(unsigned short, (_Float16)((va - (float)ha) * 2048.0f));
                    if (has_b) {
                        const size_t ob = plane_index(row + SEQ_TILE, m, a.plane_w);
                        const _Float16 hb = (_Float16)vb;
                        a.plane_hi[ob] = __builtin_bit_cast(unsigned short, hb);
                        a.plane_lo[ob] = __builtin_bit_cast(unsigned short, (_Float16)((vb - (float)hb) * 2048.0f));
                    }
                }
            } else if (m < F) {
                float *o = a.feats + ((size_t)b * a.T + t0 + fa) * F + m;
                o[0] = logf(fmaxf(ea, a.log_floor));
                if (has_b) o[F] = logf(fmaxf(eb, a.log_floor));
            }
        }
        wave_lds_fence();
    }
}

}  // namespace

__global__ __launch_bounds__(256) void stream_stage_kernel(const float *chunk_pcm, int B, int chunk, int tail, int n_left,
                                                           int first_step, const float *tail_in, float *tail_out, float *staging) {
    const int W = tail + chunk;
    const long long n = (long long)B * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / W), p = (int)(i - (long long)b * W);
        float v;
        if (p >= tail) {
            v = chunk_pcm[(size_t)b * chunk + (p - tail)];
        } else if (first_step) {
            const int idx = p - tail;   // absolute sample index in [-tail, 0): mirror (including the edge sample)
            v = (-idx <= n_left && -idx - 1 < chunk) ? chunk_pcm[(size_t)b * chunk + (-idx - 1)] : 0.0f;
        } else {
            v = tail_in[(size_t)b * tail + p];
        }
        staging[i] = v;
        if (p >= chunk) tail_out[(size_t)b * tail + (p - chunk)] = v;
    }
}

size_t fbank_lds_bytes(const FbankArgs &a) {
    const size_t raw_pad = (size_t)((((FR_WG - 1) * a.frame_shift + a.frame_len) + 3) & ~3);
    const size_t melw_pad = (size_t)((2 * ((a.tab.mel_stride + 1) / 2) * a.n_mels + 3) & ~3);
    return (raw_pad + melw_pad) * sizeof(float) + 4 * (size_t)(ZB_ELEMS + PB_ELEMS) * sizeof(float2);
}

hipError_t launch_stream_stage(const float *chunk_pcm, int B, int chunk, int tail, int n_left, int first_step,
                               const float *tail_in, float *tail_out, float *staging, hipStream_t s) {
    const long long n = (long long)B * (tail + chunk);
    long long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(stream_stage_kernel, dim3((int)g), dim3(256), 0, s, chunk_pcm, B, chunk, tail, n_left, first_step,
                       tail_in, tail_out, staging);
    return hipGetLastError();
}

hipError_t launch_fbank(const FbankArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.T <= 0) return hipSuccess;
    const size_t lds = fbank_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.plane_hi && (!a.plane_lo || a.plane_w < a.n_mels || a.plane_w % 16 || a.plane_w > 128)) return hipErrorInvalidValue;
    const unsigned rows = a.plane_hi ? (unsigned)((a.B + SEQ_TILE - 1) / SEQ_TILE * SEQ_TILE) : (unsigned)a.B;
    if (rows > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.T + FR_WG - 1) / FR_WG), rows);
    const float2 *tw = reinterpret_cast<const float2 *>(a.tab.tw512);
    hipError_t e;
    if (a.pcm_is_i16) {
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(fbank_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(fbank_kernel<true>, grid, dim3(256), lds, s, a, tw);
    } else {
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(fbank_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(fbank_kernel<false>, grid, dim3(256), lds, s, a, tw);
    }
    return hipGetLastError();
}

}  // namespace uvad

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
#include "fbank_pair.h"

namespace uvad {

namespace {

using fbp::NFFT;
using fbp::ZB_ELEMS;
using fbp::wave_lds_fence;
#ifndef UVAD_FB_PAIRS
#define UVAD_FB_PAIRS 3    // frame pairs per wave: 24 frames per workgroup (16 KiB PCM tile; with the scratch ~43 KiB of LDS -> 3 workgroups per CU)
#endif
#ifndef UVAD_FB_MINWAVES
#define UVAD_FB_MINWAVES 4   // <= 128 VGPRs (no spills): register occupancy never the limiter
#endif
constexpr int PAIRS_PER_WAVE = UVAD_FB_PAIRS;
constexpr int FR_WG = 4 * 2 * PAIRS_PER_WAVE;  // frames per workgroup
constexpr int PB_ELEMS = 0;                    // the 257 (powerA, powerB) pairs reuse the first 264 slots of the transpose scratch


template <bool I16>
__device__ __forceinline__ float pcm_at(const void *row, int64_t i) {
    if (I16) return (float)reinterpret_cast<const int16_t *>(row)[i] * (1.0f / 32768.0f);
    return reinterpret_cast<const float *>(row)[i];
}


// NORMAL_FLOOR: the energy floor is a normal float (chosen by launch_fbank from FbankArgs::log_floor; see fbp::log_floored)
template <bool I16, bool NORMAL_FLOOR>
__global__ __launch_bounds__(256, UVAD_FB_MINWAVES) void fbank_kernel(FbankArgs a, const float2 *__restrict__ tw512) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.frame_len, sh = a.frame_shift, F = a.n_mels;
    const int raw_pad = (((FR_WG - 1) * sh + L) + 3) & ~3;
    const int mel_quads = a.tab.mel_stride / 4;                // the band loop's trip count: four bins per iteration (zero weights in the padding)
    const int melw_pad = (int)mel_image_floats(a.tab.mel_stride, F);
    float *raw = smem;
    float *melw = raw + raw_pad;
    float2 *wscr = reinterpret_cast<float2 *>(melw + melw_pad);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // A tile that lies wholly inside the row, on an aligned address, needs none of the per-lane range / alignment tests below: one
    // workgroup-uniform decision (scalar arithmetic) instead of four 64-bit compares per 16-byte piece.  All but the first and last
    // tiles of a row are of this kind.
    const bool interior = !virt && s0 >= 0 && s0 + (int64_t)((need + 3) & ~3) <= a.S &&
                          ((I16 ? reinterpret_cast<uintptr_t>(reinterpret_cast<const int16_t *>(xrow) + s0) & 7
                                : reinterpret_cast<uintptr_t>(reinterpret_cast<const float *>(xrow) + s0) & 15) == 0);
    for (int base = 0; base < need; base += ST_IT * 1024) {
        float4 v[ST_IT];
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
            const int i = base + it * 1024 + tid * 4;
            const int64_t g = s0 + i;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < need) {
                bool fast = interior || (!virt && g >= 0 && g + 3 < a.S);
                if (I16) {
                    const int16_t *p = reinterpret_cast<const int16_t *>(xrow) + g;
                    fast = interior || (fast && (reinterpret_cast<uintptr_t>(p) & 7) == 0);
                    if (fast) {
                        const short4 q = *reinterpret_cast<const short4 *>(p);
                        v[it] = make_float4((float)q.x * (1.0f / 32768.0f), (float)q.y * (1.0f / 32768.0f),
                                            (float)q.z * (1.0f / 32768.0f), (float)q.w * (1.0f / 32768.0f));
                    }
                } else {
                    const float *p = reinterpret_cast<const float *>(xrow) + g;
                    fast = interior || (fast && (reinterpret_cast<uintptr_t>(p) & 15) == 0);
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
    // mel weights as [bin-in-band][filter] (transposed and zero-padded on the host: mel_image_floats), so lane m reads conflict-free
    for (int i = tid; i < melw_pad; i += 256) melw[i] = a.tab.mel_wt[i];

    // ---- per-lane constants (fbank_pair.h: window taps, twiddles, band starts; retired before the frame loop) ----------------
    fbp::PairConsts kc;
    fbp::load_pair_consts(kc, a.tab, tw512, lane, L, F);
    const int nfilt_pass = (F + 63) / 64;
    __syncthreads();
    // Per-wave scratch as TWO 4-byte arrays (real / imaginary parts of the transform; later the power of frame A / frame B), never
    // as float2 pairs: every LDS instruction of this kernel is a 32-bit one (ds_read_b32 / ds_read2_b32 / ds_write2_b32 ...).
    // With 64-bit LDS operations (ds_read_b64, ds_write2st64_b64, ...) the kernel returned wrong frames whenever its waves
    // shared a CU with waves of another kernel that loop over MFMAs, an LDS read and s_barrier (a GEMM of another step in
    // flight) -- the same happens to stock rocFFT, and the 32-bit form is immune (DESIGN.md section 3.3,
    // profiles/r02_concurrency_corruption.json).  It is also 3 % faster.
    float *zr = reinterpret_cast<float *>(wscr + (size_t)wave * (ZB_ELEMS + PB_ELEMS)), *zi = zr + ZB_ELEMS;
    const float inv_len = 1.0f / (float)L;
    constexpr bool floor_normal = NORMAL_FLOOR;

    for (int q = 0; q < PAIRS_PER_WAVE; ++q) {
        const int fa = 2 * (q * 4 + wave);
        if (fa >= nfr) break;  // wave-uniform
        const bool has_b = fa + 1 < nfr;
        const float *xa = raw + fa * sh;
        const float *xb = raw + (has_b ? fa + 1 : fa) * sh;

        // ---- the pair's transform, spectrum split, power and mel band sums (fbank_pair.h); per filter pass every lane gets its two band energies
        fbp::fbank_pair(xa, xb, has_b, kc, zr, zi, melw, mel_quads, a.tab.nyquist != 0, nfilt_pass, F, L, a.preemph, a.remove_dc != 0, inv_len, lane, [&](int m, float ea, float eb) {

            // (fbp::log_floored: ocml logf's value, without its subnormal-argument handling when the floor rules those out)
            if (a.plane_hi) {
                // straight into the operand planes of the first projection GEMM (gemm_f16p.hip: x ~= hi + lo * 2^-11, K-blocked, rows in
                // tile-major order m = (tile * T + t) * 4 + j): the f32 feature tensor never exists.  Columns [F, plane_w) are zero.
                if (m < a.plane_w) {
                    const float va = m < F ? fbp::log_floored(ea, a.log_floor, floor_normal) : 0.f, vb = m < F ? fbp::log_floored(eb, a.log_floor, floor_normal) : 0.f;
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
                o[0] = fbp::log_floored(ea, a.log_floor, floor_normal);
                if (has_b) o[F] = fbp::log_floored(eb, a.log_floor, floor_normal);
            }
        });
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
    const size_t melw_pad = mel_image_floats(a.tab.mel_stride, a.n_mels);
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
    const bool normal_floor = a.log_floor >= 1.17549435e-38f;   // FLT_MIN
    const void *fn = a.pcm_is_i16 ? (normal_floor ? (const void *)fbank_kernel<true, true> : (const void *)fbank_kernel<true, false>)
                                  : (normal_floor ? (const void *)fbank_kernel<false, true> : (const void *)fbank_kernel<false, false>);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    FbankArgs args = a;
    void *params[] = {&args, &tw};
    const hipError_t e = hipLaunchKernel(fn, grid, dim3(256), params, lds, s);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace uvad

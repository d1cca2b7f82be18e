// fbank.hip -- fused Kaldi-style log-mel front end, one launch, no intermediate in HBM.
// Stands behind lhotse's Fbank.extract_batch as the reference calls it
// (src/datasets/ami/utils.py:153,157-163; src/utils/helper.py:120-130): reflect-padded framing
// (snip_edges=False), per-frame DC removal, pre-emphasis, window, zero-pad to 512, rFFT,
// power, triangular mel projection, log(max(., eps)).  (The torch-op original materialises the
// (B,T,512) frame tensor and the complex spectrum in HBM: >5x the algorithmic traffic.)
//
// gfx950 design
//   * a workgroup (4 waves) owns FR_WG = 40 consecutive frames of one utterance: the
//     39*160+400 PCM samples they cover are loaded ONCE (16-byte coalesced loads, reflection
//     handled at the utterance edges) into an LDS tile; neighbouring workgroups re-read only the
//     240-sample overlap (4 %).
//   * a wave transforms TWO frames at a time as the real and imaginary part of one 512-point
//     complex FFT, 8 points per lane, as three radix-8 passes (512 = 8*8*8) held in registers;
//     the two re-distributions between passes go through a padded per-wave LDS scratch with
//     wave-local ordering only (no workgroup barrier in the loop).
//   * the two spectra are separated with the conjugate-symmetry identity, |.|^2 goes to LDS as
//     (frameA, frameB) pairs, and lane m accumulates mel filter m over its (start,len) band with
//     one ds_read_b64 per bin serving both frames; 64 lanes write 256 contiguous bytes per frame.
//   Algorithmic HBM bytes per frame: 640 read (320 for int16 PCM) + 4*n_mels written.
#include "uvad_internal.h"

namespace uvad {

namespace {

constexpr int NFFT = 512;
constexpr int PAIRS_PER_WAVE = 5;
constexpr int FR_WG = 4 * 2 * PAIRS_PER_WAVE;  // frames per workgroup
constexpr int ZB_LD = 9;                       // padded row (8 complex + 1) of the transpose scratch
constexpr int ZB_ELEMS = 64 * ZB_LD;           // 576 complex >= 512
constexpr int PB_ELEMS = 264;                  // 257 (powerA, powerB) pairs, padded

__device__ __forceinline__ void wave_lds_fence() {
    // order this wave's LDS traffic (hardware executes one wave's DS ops in order; this stops
    // the compiler from moving them across the hand-off point)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
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
__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs a, const float2 *__restrict__ tw512) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.frame_len, sh = a.frame_shift, F = a.n_mels;
    const int raw_pad = (((FR_WG - 1) * sh + L) + 3) & ~3;
    const int melw_pad = (a.tab.mel_stride * F + 3) & ~3;
    float *raw = smem;
    float *melw = raw + raw_pad;
    float2 *wscr = reinterpret_cast<float2 *>(melw + melw_pad);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * FR_WG;
    const int nfr = (int)((a.T - t0) < FR_WG ? (a.T - t0) : FR_WG);
    const int n_left = a.snip_edges ? 0 : (L - sh) / 2;
    const int64_t s0 = t0 * sh - n_left;
    const int need = (nfr - 1) * sh + L;
    const size_t rstride = a.row_stride ? (size_t)a.row_stride : (size_t)a.S;
    const void *xrow = I16 ? (const void *)(reinterpret_cast<const int16_t *>(a.pcm) + (size_t)b * rstride)
                           : (const void *)(reinterpret_cast<const float *>(a.pcm) + (size_t)b * rstride);

    // ---- stage the PCM tile (reflect at the utterance edges) --------------------------------
    for (int i = tid * 4; i < need; i += 256 * 4) {
        const int64_t g = s0 + i;
        bool fast = g >= 0 && g + 3 < a.S && i + 3 < need;
        if (fast) {
            if (I16) {
                const int16_t *p = reinterpret_cast<const int16_t *>(xrow) + g;
                fast = (reinterpret_cast<uintptr_t>(p) & 7) == 0;
                if (fast) {
                    const short4 v = *reinterpret_cast<const short4 *>(p);
                    raw[i + 0] = (float)v.x * (1.0f / 32768.0f);
                    raw[i + 1] = (float)v.y * (1.0f / 32768.0f);
                    raw[i + 2] = (float)v.z * (1.0f / 32768.0f);
                    raw[i + 3] = (float)v.w * (1.0f / 32768.0f);
                }
            } else {
                const float *p = reinterpret_cast<const float *>(xrow) + g;
                fast = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
                if (fast) *reinterpret_cast<float4 *>(raw + i) = *reinterpret_cast<const float4 *>(p);
            }
        }
        if (!fast) {
            for (int e = 0; e < 4 && i + e < need; ++e) {
                int64_t idx = g + e;
                if (idx < 0) idx = -idx - 1;
                if (idx >= a.S) idx = 2 * a.S - 1 - idx;
                if (idx < 0) idx = 0;
                if (idx >= a.S) idx = a.S - 1;
                raw[i + e] = pcm_at<I16>(xrow, idx);
            }
        }
    }
    // mel weights transposed to [bin-in-band][filter] so lane m reads conflict-free
    for (int i = tid; i < a.tab.mel_stride * F; i += 256) {
        const int m = i % F, r = i / F;
        melw[i] = a.tab.mel_w[(size_t)m * a.tab.mel_stride + r];
    }

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
    __syncthreads();

    float2 *zb = wscr + (size_t)wave * (ZB_ELEMS + PB_ELEMS);
    float2 *pb = zb + ZB_ELEMS;
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
            const int n = lane + 64 * r;
            const bool in = n < L;
            const int np = n > 0 ? n - 1 : 0;
            re[r] = in ? xa[n] : 0.f;
            pa[r] = in ? xa[np] : 0.f;
            im[r] = in ? xb[n] : 0.f;
            pbv[r] = in ? xb[np] : 0.f;
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
            for (int k = 0; k < 8; ++k) zb[(k * 8 + bb) * ZB_LD + aa] = make_float2(re[k], im[k]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = zb[lane * ZB_LD + k];
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
            for (int c = 0; c < 8; ++c) zb[(k1 * 8 + c) * ZB_LD + bb] = make_float2(re[c], im[c]);
        }
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 v = zb[lane * ZB_LD + k];
            re[k] = v.x; im[k] = v.y;
        }
        // ---- pass 3: lane (k1, c): DFT over b -> d; Z[k1 + 8c + 64d] -------------------------------
        dft8(re, im);
        wave_lds_fence();
        {
            const int lam = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
            for (int d = 0; d < 8; ++d) zb[lam + 64 * d] = make_float2(re[d], im[d]);
        }
        wave_lds_fence();
        // ---- split the two real spectra, power: A = (Z[k] + conj Z[N-k])/2, B = (Z[k] - conj Z[N-k])/(2i)
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const int k = lane + 64 * d;
            if (d < 4 || lane == 0) {
                const float2 z = zb[k];
                const float2 w = zb[(NFFT - k) & (NFFT - 1)];
                const float ar = z.x + w.x, ai = z.y - w.y;
                const float br = z.y + w.y, bi = z.x - w.x;
                pb[k] = make_float2(0.25f * (ar * ar + ai * ai), 0.25f * (br * br + bi * bi));
            }
        }
        wave_lds_fence();
        // ---- mel band sums + log; lane = filter ----------------------------------------------------
        for (int ps = 0; ps < nfilt_pass; ++ps) {
            const int m = lane + 64 * ps;
            if (m < F) {
                const int st = a.tab.mel_start[m], ln = a.tab.mel_len[m];
                float ea = 0.f, eb = 0.f;
                for (int i = 0; i < ln; ++i) {
                    const float wv = melw[i * F + m];
                    const float2 pv = pb[st + i];
                    ea = __builtin_fmaf(wv, pv.x, ea);
                    eb = __builtin_fmaf(wv, pv.y, eb);
                }
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
    const size_t melw_pad = (size_t)((a.tab.mel_stride * a.n_mels + 3) & ~3);
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
    const dim3 grid((unsigned)((a.T + FR_WG - 1) / FR_WG), (unsigned)a.B);
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

// fbank_pair.h -- the per-wave part of the log-mel front end, shared by fbank_kernel (fbank.hip) and the streaming step's one-launch
// kernel (lstm_stack.hip): two frames at a time as the real and imaginary part of one 512-point complex FFT, 8 points per lane,
// three radix-8 passes in registers, two re-distributions through a padded per-wave LDS scratch, spectrum split, |.|^2, mel band
// sums.  See fbank.hip for the design notes; every LDS access here is a 32-bit operation.
#pragma once
#include "uvad_internal.h"
#include <type_traits>

namespace uvad {
namespace fbp {

constexpr int NFFT = 512;
constexpr int ZB_LD = 9;                       // padded row (8 complex + 1) of the transpose scratch
constexpr int ZB_ELEMS = 64 * ZB_LD;           // 576 complex >= 512 (the 257 (powerA, powerB) pairs reuse its first slots)

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

// log(max(e, floor)) of a band energy.  ocml's logf (<= 1 ulp; the 2-ulp-of-log2 __logf is not used) is v_log_f32 and a
// double-float product with ln 2 -- and, around it, a rescaling of subnormal arguments (compare, select, ldexp before; select,
// subtract after) plus two canonicalising v_max in front of the fmaxf.  With a floor that is itself a normal number (the
// reference's is FLT_EPSILON) the argument is never subnormal: the same four operations on the v_log_f32 result then give ocml's
// value BIT FOR BIT (tests/test_gpu_parity.py compares the two paths) in 8 instead of 15 vector instructions.  A floor below
// FLT_MIN takes ocml's logf.
__device__ __forceinline__ float log_floored(float e, float floor, bool floor_is_normal) {
    if (!floor_is_normal) return logf(fmaxf(e, floor));
    float x;
    asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(e), "v"(floor));   // (fmaxf: a NaN energy reads as the floor, as with ocml)
    const float y = __builtin_amdgcn_logf(x);                     // log2
    constexpr float C = 0x1.62e42ep-1f, C_LO = 0x1.efa39ep-25f;   // ln 2 = C + C_LO (0x3f317217, 0x3377d1cf)
    const float t = C * y;
    const float r = __builtin_fmaf(C, y, __builtin_fmaf(C_LO, y, __builtin_fmaf(y, C, -t)));
    return __builtin_fabsf(y) < __builtin_inff() ? r : y;
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

// per-lane constants of a wave: window taps, twiddles of the two twiddled passes, band starts of the lane's filters (passes 0 and 1
// cover n_mels <= 128).  Loaded ONCE, and retired before any loop that stores to global memory: a register that is still "in
// flight" at loop entry makes hipcc put s_waitcnt vmcnt(0) at its first use INSIDE the loop, where it then also waits for the
// stores of the previous iteration.
struct PairConsts {
    float win[8];
    float2 tw1[8], tw2[8];
    int mst0, mst1;
};
__device__ __forceinline__ void load_pair_consts(PairConsts &k, const FbankTables &tab, const float2 *__restrict__ tw512, int lane, int L, int F) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        k.win[r] = n < L ? tab.window[n] : 0.0f;
        k.tw1[r] = tw512[(lane * r) & (NFFT - 1)];
        k.tw2[r] = tw512[(8 * (lane & 7) * r) & (NFFT - 1)];
    }
    k.mst0 = tab.mel_start[lane < F ? lane : F - 1];
    k.mst1 = tab.mel_start[lane + 64 < F ? lane + 64 : F - 1];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        asm volatile("" : "+v"(k.win[r]), "+v"(k.tw1[r].x), "+v"(k.tw1[r].y), "+v"(k.tw2[r].x), "+v"(k.tw2[r].y));
    asm volatile("" : "+v"(k.mst0), "+v"(k.mst1));
}

// One frame pair.  xa / xb: the frames' first samples in LDS (xb == xa when there is no second frame); zr / zi: the wave's scratch
// (ZB_ELEMS floats each, zi == zr + ZB_ELEMS); melw: the [bin-in-band][mel_image_ld(F)] weight image in LDS (mel_image_floats).  emit(m, ea, eb) is called by every lane once per
// filter pass with its filter index m = lane + 64 * pass (possibly >= F) and the two band energies (before the log).
template <class Emit>
__device__ __forceinline__ void fbank_pair(const float *xa, const float *xb, bool has_b, const PairConsts &kc, float *zr, float *zi,
                                           const float *melw, int mel_quads, bool nyquist, int nfilt_pass, int F, int L, float preemph, bool remove_dc,
                                           float inv_len, int lane, Emit &&emit) {
#define ZB_PUT(idx, a, b) { zr[idx] = (a); zi[idx] = (b); }
#define ZB_GET(idx) make_float2(zr[idx], zi[idx])
    const float (&win)[8] = kc.win;
    const float2 (&tw1)[8] = kc.tw1;
    const float2 (&tw2)[8] = kc.tw2;
    const int mst0v = kc.mst0, mst1 = kc.mst1;
        // ---- framing, DC removal, pre-emphasis, window: lane p owns n = p + 64 r ---------------
        // The frame length L is a launch argument.  Of the eight groups of 64 samples a lane's points come from, L / 64 lie wholly
        // inside the frame (six at the reference geometry, L = 400), at most one is cut by the frame's end, the rest are empty.  The
        // loads are instantiated per count of whole groups and selected by one wave-uniform switch: a whole group is read at base +
        // immediate offsets with no index clamp and no select, only the cut group clamps its index into the frame and zeroes the
        // samples past it, an empty group loads nothing.  Nothing outside [0, L) of either frame is read.
        float re[8], im[8], pa[8], pbv[8];
        {
            const float *va = xa + lane, *vb = xb + lane;
            const int p0 = lane > 0 ? -1 : 0;   // x[-1] := x[0] (replicate-padded pre-emphasis)
            auto load = [&](auto whole) {
                constexpr int NW = decltype(whole)::value;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < NW) {
                        re[r] = va[64 * r];
                        im[r] = vb[64 * r];
                        pa[r] = r == 0 ? va[p0] : va[64 * r - 1];
                        pbv[r] = r == 0 ? vb[p0] : vb[64 * r - 1];
                    } else if (r == NW) {
                        const int n = lane + 64 * r;
                        const bool in = n < L;
                        const int nc = in ? n : L - 1;
                        const int np = nc > 0 ? nc - 1 : 0;
                        const float qa = xa[nc], qb = xb[nc];
                        re[r] = in ? qa : 0.f;
                        im[r] = in ? qb : 0.f;
                        pa[r] = xa[np];
                        pbv[r] = xb[np];
                    } else {
                        re[r] = im[r] = pa[r] = pbv[r] = 0.f;
                    }
                }
            };
            switch (L >> 6) {   // wave-uniform; 1 <= L <= 512 (uvad_create)
                case 0: load(std::integral_constant<int, 0>{}); break;
                case 1: load(std::integral_constant<int, 1>{}); break;
                case 2: load(std::integral_constant<int, 2>{}); break;
                case 3: load(std::integral_constant<int, 3>{}); break;
                case 4: load(std::integral_constant<int, 4>{}); break;
                case 5: load(std::integral_constant<int, 5>{}); break;
                case 6: load(std::integral_constant<int, 6>{}); break;
                case 7: load(std::integral_constant<int, 7>{}); break;
                default: load(std::integral_constant<int, 8>{}); break;
            }
        }
        float suma = 0.f, sumb = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            suma += re[r];
            sumb += im[r];
        }
        // (x[n] - mu) - p (x[n-1] - mu) = x[n] - p x[n-1] - (1 - p) mu: the mean enters as one constant per frame.  Without a second
        // frame xb == xa and the imaginary input is a copy of the real one: finite, and its band energies are never stored.
        float ca = 0.f, cb = 0.f;
        if (remove_dc) {
            const float mua = wave_sum(suma) * inv_len, mub = wave_sum(sumb) * inv_len;
            ca = __builtin_fmaf(-preemph, mua, mua);
            cb = __builtin_fmaf(-preemph, mub, mub);
        }
        (void)has_b;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            re[r] = (__builtin_fmaf(-preemph, pa[r], re[r]) - ca) * win[r];
            im[r] = (__builtin_fmaf(-preemph, pbv[r], im[r]) - cb) * win[r];
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
        // (all reads first, then the power pairs overwrite the scratch).  Bins 0 .. 255 are four rounds of 64 lanes; bin 256 is a fifth
        // round for lane 0's sake, run only if some filter weighs it (FbankTables::nyquist; otherwise slot 256 keeps a finite spectrum
        // value, which the band loop may read against a zero weight).
        {
            float2 pw[5];
            auto power = [&](int kc) {
                const float2 z = ZB_GET(kc);
                const float2 w = ZB_GET((NFFT - kc) & (NFFT - 1));
                const float ar = z.x + w.x, ai = z.y - w.y;
                const float br = z.y + w.y, bi = z.x - w.x;
                return make_float2(ar * ar + ai * ai, br * br + bi * bi);   // 4 x the power: the image's weights carry the 1/4 (exact)
            };
#pragma unroll
            for (int d = 0; d < 4; ++d) pw[d] = power(lane + 64 * d);
            if (nyquist) pw[4] = power(lane == 0 ? 256 : 0);   // wave-uniform branch
            wave_lds_fence();
#pragma unroll
            for (int d = 0; d < 4; ++d) ZB_PUT(lane + 64 * d, pw[d].x, pw[d].y);
            if (nyquist && lane == 0) ZB_PUT(256, pw[4].x, pw[4].y);
        }
        wave_lds_fence();
        // ---- mel band sums + log; lane = filter ----------------------------------------------------
        // Uniform trip count (the longest band rounded up to four bins, zero-padded weights), eight bins per iteration (+ one group of
        // four when the count is odd): their weights are rows of a compile-time stride apart (immediate offsets of ONE lane address),
        // their power values adjacent, and all twelve reads of an iteration are issued before its first sum -- a wave waits for LDS once
        // per eight bins of two frames.  (Reading one iteration ahead into a second register set was tried first: hipcc keeps the
        // loop-carried set in copies, ten v_mov per iteration, and waits for the new loads at the copies.)
        // No clamp of the bin index: a band that runs past bin 256 reads finite scratch values against zero weights.  Two chains per
        // frame (even / odd bins), each summed in increasing bin order.
        auto bands = [&](auto ld_tag) {
            constexpr int LD = decltype(ld_tag)::value;
            for (int ps = 0; ps < nfilt_pass; ++ps) {
                const int m = lane + 64 * ps;
                const int st = ps == 0 ? mst0v : mst1;   // n_mels <= 128 (checked by uvad_create): no global load in this loop
                float ea = 0.f, eb = 0.f, ea2 = 0.f, eb2 = 0.f;
                const float *wp = melw + m;              // (columns [F, LD) of the image are zero)
                const float *pr = zr + st;
                // (zi == zr + ZB_ELEMS) as a base register of its own -- the OFFSET is made opaque, not the pointer (which would lose its
                // LDS address space and turn the reads into flat loads): otherwise hipcc rebuilds the address per read pair, four adds per iteration
                int zoff = ZB_ELEMS;
                asm volatile("" : "+v"(zoff));
                const float *pi = pr + zoff;
                float w0[4], a0[4], b0[4], w1[4], a1[4], b1[4];
#define UVAD_MEL_LOAD(W, A, B, q)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
        W[j] = wp[((q) * 4 + j) * LD];                                                                                 \
        A[j] = pr[(q) * 4 + j];                                                                                        \
        B[j] = pi[(q) * 4 + j];                                                                                        \
    }
#define UVAD_MEL_SUM(W, A, B)                                                                                          \
    ea = __builtin_fmaf(W[0], A[0], ea);   eb = __builtin_fmaf(W[0], B[0], eb);                                        \
    ea2 = __builtin_fmaf(W[1], A[1], ea2); eb2 = __builtin_fmaf(W[1], B[1], eb2);                                      \
    ea = __builtin_fmaf(W[2], A[2], ea);   eb = __builtin_fmaf(W[2], B[2], eb);                                        \
    ea2 = __builtin_fmaf(W[3], A[3], ea2); eb2 = __builtin_fmaf(W[3], B[3], eb2);
                int i = 0;
                for (; i + 2 <= mel_quads; i += 2) {
                    UVAD_MEL_LOAD(w0, a0, b0, 0)
                    UVAD_MEL_LOAD(w1, a1, b1, 1)
                    UVAD_MEL_SUM(w0, a0, b0)
                    UVAD_MEL_SUM(w1, a1, b1)
                    wp += 8 * LD; pr += 8; pi += 8;
                }
                if (i < mel_quads) {
                    UVAD_MEL_LOAD(w0, a0, b0, 0)
                    UVAD_MEL_SUM(w0, a0, b0)
                }
#undef UVAD_MEL_LOAD
#undef UVAD_MEL_SUM
                ea += ea2;
                eb += eb2;
                emit(m, ea, eb);
            }
        };
        if (F <= 64) bands(std::integral_constant<int, 64>{});
        else bands(std::integral_constant<int, 128>{});
#undef ZB_PUT
#undef ZB_GET
}

}  // namespace fbp
}  // namespace uvad

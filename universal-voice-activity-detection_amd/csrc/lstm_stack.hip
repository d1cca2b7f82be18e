// lstm_stack.hip -- the whole LSTM stack of a CAUSAL model for a few new frames, one launch (uvad_stream_step, BASELINE cfg 5).
// Reference: the same `self.lstm(outputs)` of PyanNet2.forward (src/models/segmentation/PyanNet2.py:169-172) with
// bidirectional=False, called chunk after chunk with the (h, c) of every layer carried -- what a real-time feed does.
//
// Why.  A streaming step of 512 feeds x 20 ms is 2 frames per feed: as per-layer kernels it was 8 dependent launches (projection +
// recurrence per layer) of a few microseconds of work each, and the step's latency was the launches (0.104 ms for 11 of them).
// In a causal stack nothing couples two sequences: layer l + 1 of a sequence needs layer l of THAT sequence only.  So one
// workgroup takes 4 sequences through ALL layers for the step's T <= 8 frames and no workgroup ever waits for another:
//   per layer   phase A: W_ih's register image -> the wave's 128 resident registers (the A operand of v_mfma_f32_4x4x1_16B_f32, as
//                        in lstm_rec_kernel); gates_t = b + W_ih x_t for the T frames, x_t read from LDS (the features, then the
//                        previous layer's h_t); a lane ends a chain holding the four gate pre-activations of ONE (unit, sequence)
//                        pair, kept in registers;
//               phase B: W_hh's image into the same registers; T recurrence steps exactly as lstm_rec_kernel's (h through a
//                        double-buffered LDS tile, one barrier per step), h_t also written where the next layer reads its input.
// Exact f32 throughout (the MFMA is a k-ordered fmaf chain); the weights (about 2 MB per workgroup at 4 x 128 units) come from L2.
// The head (feed-forward layers on the same chains, classifier row, sigmoid) and the feature stage (fbank_pair.h: the workgroup frames
// its 4 rows from the chunk and the carried PCM tail and transforms its <= 8 frame pairs, one per wave) run in the same launch when the
// model allows: a streaming step is then this one kernel.  Otherwise the features come from fbank_kernel and the last layer's h_t
// leaves as the two K-blocked f16 planes the fused head reads (or f32 rows).
#include "uvad_internal.h"
#include "fbank_pair.h"

namespace uvad {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr float L2E = 1.4426950408889634f;

// gate functions: the ones of lstm.hip (held to ~1 ulp of absolute error at their output scale)
__device__ __forceinline__ float rcp_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float sigmoid_f(float x) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fminf(-L2E * x, 126.0f));
    return rcp_nr(1.0f + e);
}
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __builtin_amdgcn_exp2f((-2.0f * L2E) * __builtin_fabsf(x));
    const float n = 1.0f - e, d = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(d);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_copysignf(q, x);
}
__device__ __forceinline__ float lstm_cell(const f32x4 g, float &c) {
    const float ig = sigmoid_f(g[0]), fg = sigmoid_f(g[1]), gg = tanh_f(g[2]), og = sigmoid_f(g[3]);
    c = __builtin_fmaf(fg, c, ig * gg);
    return og * tanh_f(c);
}

constexpr int H = 128, WAVES = 8, HS = H + 4;   // LDS row stride (floats): the 4 sequence rows land on disjoint banks

// init + sum_k w[k] * v[k] over 4 KQ values of the lane's sequence row `vrow` (LDS): four independent chains (k mod 4), HR LDS reads in flight
// HR: LDS reads kept in flight ahead of the MFMA groups (lstm_rec_kernel keeps 8; here gq[] needs the registers: 4, or 2 where 4 would spill)
template <int KQ, int HR>
__device__ __forceinline__ f32x4 chain(float (&w)[H], const float *vrow, f32x4 init) {
    static_assert(KQ >= HR, "ring depth");
#pragma unroll
    for (int k = 0; k < 4 * KQ; ++k) asm volatile("" : "+a"(w[k]));   // resident in the accumulator half of the register file (constraint only)
    float4 hv[HR];
#pragma unroll
    for (int kq = 0; kq < HR; ++kq) hv[kq] = *reinterpret_cast<const float4 *>(vrow + 4 * kq);
    f32x4 a0 = init, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        const float4 hq = hv[kq % HR];
        if (kq + HR < KQ) hv[kq % HR] = *reinterpret_cast<const float4 *>(vrow + 4 * (kq + HR));
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 0], hq.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 1], hq.y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 2], hq.z, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[4 * kq + 3], hq.w, a3, 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, HR, 0);
#pragma unroll
    for (int i = 0; i < KQ - HR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * HR, 0);
    return (a0 + a1) + (a2 + a3);
}

template <int KQ>
__device__ __forceinline__ void load_image(float (&w)[H], const float *img, int wave, int lane) {
    const float4 *wp = reinterpret_cast<const float4 *>(img);
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        const float4 v = wp[(size_t)(wave * KQ + kq) * 64 + lane];
        w[4 * kq + 0] = v.x; w[4 * kq + 1] = v.y; w[4 * kq + 2] = v.z; w[4 * kq + 3] = v.w;
    }
}

// KIN0: feature width (columns of layer 0's W_ih): 64 or 80 (other widths: the per-layer kernels)
// FB: the feature stage runs in this launch (LstmStackArgs::fb; only with the head in the launch too, so PLANES is false then)
template <int KIN0, bool PLANES, bool FB>
__global__ __launch_bounds__(WAVES * 64) void lstm_stack_kernel(LstmStackArgs a) {
    static_assert(!(FB && PLANES), "the feature stage comes with the in-launch head: no plane output");
    constexpr int TMAX = LSTM_STACK_TMAX;
    constexpr int HR = KIN0 == 64 ? 4 : 2;
    __shared__ __attribute__((aligned(16))) float xbuf[TMAX][SEQ_TILE][HS];   // input of the current layer, frame by frame
    __shared__ __attribute__((aligned(16))) float hbuf[2][SEQ_TILE][HS];

    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jb = lane & 3;    // sequence within the tile (B / D operand column), gate index of the A operand
    const int blk = lane >> 2;  // MFMA block = hidden unit within the wave's row block
    const int unit = wave * 16 + blk;
    const int T = a.T;

    if constexpr (FB) {
        // ---- the feature stage in this launch (fbank_pair.h): the step's rows are virtual -- [carried tail | this step's chunk], as in
        //      fbank_kernel's streaming form (FbankArgs::vs_*) -- and belong to this workgroup alone: its 4 sequences x T <= 4 frames are
        //      8 frame pairs at most, one per wave (row = wave / 2, pair = wave % 2), written straight into xbuf.
        extern __shared__ __attribute__((aligned(16))) float fbs[];
        const FbankArgs &f = a.fb;
        const int L = f.frame_len, sh = f.frame_shift, F = KIN0;
        const int need = (T - 1) * sh + L, raw_ld = (need + 3) & ~3;
        const int mel_quads = f.tab.mel_stride / 4, melw_n = (int)mel_image_floats(f.tab.mel_stride, F);
        float *raw = fbs, *melw = raw + SEQ_TILE * raw_ld, *scr = melw + melw_n;
        auto vs_abs = [&](int b, int p) -> float {   // sample p of row b's [tail | chunk] image (see fbank.hip)
            if (b >= a.B) return 0.0f;
            if (p >= f.vs_tail) return f.vs_chunk[(size_t)b * f.vs_chunk_len + (p - f.vs_tail)];
            if (f.vs_first) {
                const int m = f.vs_tail - p;
                return (m <= f.vs_n_left && m - 1 < f.vs_chunk_len) ? f.vs_chunk[(size_t)b * f.vs_chunk_len + (m - 1)] : 0.0f;
            }
            return f.vs_tail_in[(size_t)b * f.vs_tail + p];
        };
        for (int i = tid; i < SEQ_TILE * need; i += WAVES * 64) {
            const int j = i / need, q = i - j * need;
            raw[j * raw_ld + q] = vs_abs(tile * SEQ_TILE + j, q + f.vs_offset);
        }
        for (int i = tid; i < SEQ_TILE * f.vs_tail; i += WAVES * 64) {   // the tail the next step starts from
            const int j = i / f.vs_tail, q = i - j * f.vs_tail, b = tile * SEQ_TILE + j;
            if (b < a.B) f.vs_tail_out[(size_t)b * f.vs_tail + q] = vs_abs(b, f.vs_chunk_len + q);
        }
        for (int i = tid; i < melw_n; i += WAVES * 64) melw[i] = f.tab.mel_wt[i];
        fbp::PairConsts kc;
        fbp::load_pair_consts(kc, f.tab, reinterpret_cast<const float2 *>(f.tab.tw512), lane, L, F);
        __syncthreads();
        const int j = wave >> 1, fa = 2 * (wave & 1);
        const bool floor_normal = f.log_floor >= 1.17549435e-38f;
        if (fa < T) {   // wave-uniform
            const bool has_b = fa + 1 < T;
            float *zr = scr + (size_t)wave * (2 * fbp::ZB_ELEMS), *zi = zr + fbp::ZB_ELEMS;
            fbp::fbank_pair(raw + j * raw_ld + fa * sh, raw + j * raw_ld + (has_b ? fa + 1 : fa) * sh, has_b, kc, zr, zi, melw, mel_quads, f.tab.nyquist != 0,
                            (F + 63) / 64, F, L, f.preemph, f.remove_dc != 0, 1.0f / (float)L, lane, [&](int m, float ea, float eb) {
                                if (m < F) {
                                    xbuf[fa][j][m] = fbp::log_floored(ea, f.log_floor, floor_normal);
                                    if (has_b) xbuf[fa + 1][j][m] = fbp::log_floored(eb, f.log_floor, floor_normal);
                                }
                            });
        }
        // (the __syncthreads() in front of layer 0's chains publishes xbuf)
    } else
    // ---- the features of the tile's 4 sequences (canonical [B][T][KIN0]; rows of padding sequences read as zero)
    for (int i = tid; i < T * SEQ_TILE * (KIN0 / 4); i += WAVES * 64) {
        const int q = i % (KIN0 / 4), j = (i / (KIN0 / 4)) % SEQ_TILE, t = i / (KIN0 / 4 * SEQ_TILE);
        const int b = tile * SEQ_TILE + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < a.B) v = *reinterpret_cast<const float4 *>(a.feats + ((size_t)b * T + t) * KIN0 + 4 * q);
        *reinterpret_cast<float4 *>(&xbuf[t][j][4 * q]) = v;
    }

    // (32-bit offsets: a step's outputs and a layer's state are far below 2^31 elements; launch_lstm_stack checks)
    const unsigned so = (unsigned)((tile * SEQ_TILE + jb) * H + unit);   // this lane's (sequence, unit) in a layer's state
    const unsigned rowu = (unsigned)(tile * T * SEQ_TILE);               // row of (t, jb) in the tile-major outputs = rowu + 4 t + jb
    const unsigned y_wave = (unsigned)(wave * (PLANE_TILE * 16));
    const unsigned y_lane = (unsigned)(jb * 16 + blk);
    const unsigned y_tile = (unsigned)((a.ldy / 16) * (PLANE_TILE * 16));

    float w[H];
    for (int l = 0; l < a.n_layers; ++l) {
        const bool last = l + 1 == a.n_layers;
        // ---- phase A: gate pre-activations of the T frames from the layer's input
        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias[l] + unit * 4);
        const f32x4 bias = {b4.x, b4.y, b4.z, b4.w};
        f32x4 gq[TMAX];
        if (l == 0) {
            load_image<KIN0 / 4>(w, a.wih[0], wave, lane);
            __syncthreads();   // the features are in LDS
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) gq[t] = chain<KIN0 / 4, HR>(w, &xbuf[t][jb][0], bias);
        } else {
            load_image<H / 4>(w, a.wih[l], wave, lane);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) gq[t] = chain<H / 4, HR>(w, &xbuf[t][jb][0], bias);
        }
        // ---- phase B: the recurrence over the T frames, carried state in / out
        load_image<H / 4>(w, a.whh[l], wave, lane);
        const size_t lo = (size_t)l * a.layer_stride + so;   // (scalar part + the lane's 32-bit offset)
        float c = a.c[lo];
        float hlast = a.h[lo];
        hbuf[0][jb][unit] = hlast;
        __syncthreads();   // every wave has read its input frames (xbuf is rewritten below) and hbuf[0] is complete
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {
                hlast = lstm_cell(chain<H / 4, HR>(w, &hbuf[t & 1][jb][0], gq[t]), c);
                hbuf[(t + 1) & 1][jb][unit] = hlast;
                if (!last || a.logits) {
                    xbuf[t][jb][unit] = hlast;             // the next layer's (or the head's) input
                } else if constexpr (PLANES) {
                    const unsigned R = rowu + (unsigned)t * SEQ_TILE;
                    const unsigned yo = (R >> 7) * y_tile + y_wave + (R & 127) * 16 + y_lane;
                    const _Float16 hh = (_Float16)hlast;
                    a.Yh[yo] = __builtin_bit_cast(unsigned short, hh);
                    a.Yl[yo] = __builtin_bit_cast(unsigned short, (_Float16)((hlast - (float)hh) * 2048.0f));
                } else {
                    a.Y[(size_t)(rowu + jb + (unsigned)t * SEQ_TILE) * a.ldy + unit] = hlast;
                }
                __syncthreads();
            }
        }
        a.h[lo] = hlast;
        a.c[lo] = c;
    }
    if (!a.logits) return;

    // ---- the head on the tile's T x 4 rows: feed-forward layers (reference: PyanNet2.py:183-185) with the SAME chain machinery -- a
    //      128 x 128 matrix is 32 MFMA blocks of 4 output units, i.e. the row blocks of waves 0 and 1; a lane ends a chain holding 4
    //      consecutive output units of (frame t, sequence jb) -- then the classifier row and the sigmoid (PyanNet2.py:186-187).
    for (int j = 0; j < a.n_lin; ++j) {
        f32x4 z[TMAX];
        if (wave < 2) {
            load_image<H / 4>(w, a.lin_w[j], wave, lane);
            const float4 b4 = *reinterpret_cast<const float4 *>(a.lin_b[j] + unit * 4);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
                    z[t] = chain<H / 4, HR>(w, &xbuf[t][jb][0], f32x4{b4.x, b4.y, b4.z, b4.w});
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[t][e] = z[t][e] >= 0.f ? z[t][e] : a.slope * z[t][e];
                }
        }
        __syncthreads();   // every chain has read its input rows
        if (wave < 2) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) *reinterpret_cast<float4 *>(&xbuf[t][jb][unit * 4]) = make_float4(z[t][0], z[t][1], z[t][2], z[t][3]);
        }
        __syncthreads();
    }
    if (tid < T * SEQ_TILE) {
        const int t = tid / SEQ_TILE, j = tid % SEQ_TILE, b = tile * SEQ_TILE + j;
        float acc = a.cls_b[0];
        for (int k = 0; k < H; ++k) acc = __builtin_fmaf(a.cls_w[k], xbuf[t][j][k], acc);
        if (b < a.B) {
            a.logits[(size_t)b * a.ld_out + t] = acc;
            if (a.probs) a.probs[(size_t)b * a.ld_out + t] = 1.0f / (1.0f + expf(-acc));
        }
    }
}

}  // namespace

size_t lstm_image_elems(int K) { return (size_t)4 * H * K; }
size_t fc_image_elems() { return (size_t)H * H; }

// register image of a 128 x 128 nn.Linear weight for the head phase of lstm_stack_kernel: waves 0 and 1, MFMA block blk of wave w =
// output units 4 (16 w + blk) .. + 3 (the A operand's four rows): [wave 2][kq = k/4][lane 64][4 k], lane = blk * 4 + row
void pack_fc_image(const float *wm, float *out) {
    for (int wave = 0; wave < 2; ++wave)
        for (int kq = 0; kq < H / 4; ++kq)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int o = (wave * 16 + (lane >> 2)) * 4 + (lane & 3);
                    out[((size_t)(wave * (H / 4) + kq) * 64 + lane) * 4 + e] = wm[(size_t)o * H + 4 * kq + e];
                }
}

// register image of a [4H][K] torch matrix (rows i,f,g,o blocks of H) for lstm_stack_kernel / lstm_rec_kernel:
// [wave][kq = k/4][lane 64][4 k] with lane = (unit within the wave) * 4 + gate
void pack_lstm_image(const float *wm, int K, float *out) {
    for (int wave = 0; wave < WAVES; ++wave)
        for (int kq = 0; kq < K / 4; ++kq)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int gate = lane & 3, unit = wave * 16 + (lane >> 2);
                    out[((size_t)(wave * (K / 4) + kq) * 64 + lane) * 4 + e] = wm[(size_t)(gate * H + unit) * K + 4 * kq + e];
                }
}

size_t lstm_stack_fb_lds_bytes(const FbankArgs &fb, int T) {
    if (!fb.vs_chunk || !fb.tab.mel_wt || !fb.tab.tw512 || T < 1 || T > LSTM_STACK_TMAX || fb.n_mels > 128) return 0;
    const size_t need = (size_t)(T - 1) * fb.frame_shift + fb.frame_len, raw_ld = (need + 3) & ~(size_t)3;
    const size_t melw = mel_image_floats(fb.tab.mel_stride, fb.n_mels);
    const size_t bytes = (SEQ_TILE * raw_ld + melw + (size_t)WAVES * 2 * fbp::ZB_ELEMS) * sizeof(float);
    return bytes <= 120 * 1024 ? bytes : 0;
}

bool lstm_stack_supported(int hidden, int dirs, int in_dim, int T, int n_layers) {
    return hidden == H && dirs == 1 && (in_dim == 64 || in_dim == 80) && T >= 1 && T <= LSTM_STACK_TMAX && n_layers >= 1 &&
           n_layers <= LSTM_STACK_MAX_LAYERS;
}

hipError_t launch_lstm_stack(const LstmStackArgs &a, hipStream_t s) {
    if (a.tiles <= 0 || a.T <= 0) return hipSuccess;
    if (!lstm_stack_supported(H, 1, a.kin0, a.T, a.n_layers) || (!a.feats && !a.fb_on) || !a.h || !a.c) return hipErrorInvalidValue;
    if ((long long)a.tiles * SEQ_TILE * a.T * (a.ldy > H ? a.ldy : H) >= (1LL << 31)) return hipErrorInvalidValue;   // 32-bit offsets in the kernel
    const bool planes = a.Y == nullptr;
    if (!a.logits && planes && (!a.Yh || !a.Yl)) return hipErrorInvalidValue;
    if (a.logits && (a.n_lin < 0 || a.n_lin > LSTM_STACK_MAX_LIN || !a.cls_w || !a.cls_b || a.ld_out < a.T)) return hipErrorInvalidValue;
    size_t lds = 0;
    if (a.fb_on) {
        lds = lstm_stack_fb_lds_bytes(a.fb, a.T);
        if (!lds || a.fb.n_mels != a.kin0 || !a.logits) return hipErrorInvalidValue;
    }
    const dim3 grid(a.tiles), block(WAVES * 64);
    // (the dynamic-LDS attribute belongs to the (function, device) pair: set for the current device on every launch that needs it)
#define UVAD_STACK_LAUNCH(K_)                                                                                                              \
    if (a.fb_on) {                                                                                                                         \
        if (lds > 40 * 1024) {                                                                                                             \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_stack_kernel<K_, false, true>),                   \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            if (e != hipSuccess) return e;                                                                                                 \
        }                                                                                                                                  \
        hipLaunchKernelGGL((lstm_stack_kernel<K_, false, true>), grid, block, lds, s, a);                                                  \
    } else if (planes) hipLaunchKernelGGL((lstm_stack_kernel<K_, true, false>), grid, block, 0, s, a);                                     \
    else hipLaunchKernelGGL((lstm_stack_kernel<K_, false, false>), grid, block, 0, s, a);
    switch (a.kin0) {
        case 64: UVAD_STACK_LAUNCH(64) break;
        case 80: UVAD_STACK_LAUNCH(80) break;
        default: return hipErrorInvalidValue;
    }
#undef UVAD_STACK_LAUNCH
    return hipGetLastError();
}

}  // namespace uvad

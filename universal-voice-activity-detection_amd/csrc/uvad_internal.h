// Internal launcher declarations shared by the translation units of libuvad.so.
// Everything here is gfx950-only HIP; no torch types, no CPU fallbacks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uvad {

// Sequences are processed in tiles of SEQ_TILE batch entries: activation rows are ordered
//   m = (tile * T + t) * SEQ_TILE + j,   b = tile * SEQ_TILE + j
// so that the SEQ_TILE rows one recurrent workgroup needs at step t are adjacent in HBM.
constexpr int SEQ_TILE = 4;

// f16 activation planes (the operands of gemm_f16p.hip) are K-BLOCKED: rows in tiles of PLANE_TILE, columns in blocks of 16,
//   element (row, col) of a plane with `width` columns (a multiple of 16) lives at plane_index(row, col, width),
// so the PLANE_TILE x 16 slab one GEMM workgroup needs per k-block is one contiguous 4 KiB run.  A plane of M rows
// occupies plane_rows(M) * width elements.
constexpr int PLANE_TILE = 128;
__host__ __device__ inline size_t plane_index(size_t row, int col, int width) {
    return ((row / PLANE_TILE) * (size_t)(width / 16) + (size_t)(col / 16)) * (PLANE_TILE * 16) + (row % PLANE_TILE) * 16 + (size_t)(col % 16);
}
inline size_t plane_rows(size_t M) { return (M + PLANE_TILE - 1) / PLANE_TILE * PLANE_TILE; }

// The gate pre-activations G (output of the projection GEMMs, input of the recurrent kernels) are TILE-BLOCKED:
// [128-row tile][64-column tile][128][64] f32, so a GEMM workgroup's 128 x 128 output tile (two adjacent tiles) is one contiguous 64 KiB run
// (row-major G made every workgroup write 128 pieces of 256 bytes 4 KiB apart: measured 2.6 TB/s, not overlapped with
// the K loops) and the 4 rows x 64 gate columns (16 units x 4 gates) a recurrent wave reads per step are one contiguous KiB.
// ncols (= 4 * hidden * directions) is a multiple of 64; a G of M rows occupies plane_rows(M) * ncols floats.
__host__ __device__ inline size_t g_index(size_t row, int col, int ncols) {
    return ((row / 128) * (size_t)(ncols / 64) + (size_t)(col / 64)) * (128 * 64) + (row % 128) * 64 + (size_t)(col % 64);
}

// ---- gemm.hip / gemm_f16p.hip -------------------------------------------------------------
// C[M][ldc] (cols [0,N)) = act( A[M][K] * W[N][K]^T + bias[N] ).
//   gemm.hip       exact f32 on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain); A and C are f32.
//   gemm_f16p.hip  f32-accurate on the f16 matrix cores; A arrives (and C may leave) as two f16 PLANES
//                  x ~= hi + lo * 2^-11 written by the producing kernel, W as three exact f16 planes (see gemm_f16p.hip).
struct GemmArgs {
    // ---- gemm.hip operands
    const float *A;      // activations
    const float *W;      // [N][ldw] row-major (torch Linear / LSTM weight layout, rows possibly permuted),
                         // each row zero-padded to ldw = gemm_padded_k(K) floats
    // ---- gemm_f16p.hip operands
    const unsigned short *Ah, *Al;    // K-blocked f16 planes of the activations, K columns (columns past the true width zero)
    const unsigned short *Wsplit16;   // w * 2^S as three K-blocked f16 planes (128-row tiles) that add up to it exactly
    float wscale;                     // 2^-S
    unsigned short *Ch, *Cl;          // out_planes: K-blocked f16 planes of the result, ldc columns ([N, ldc) written as zero)
    int out_planes;
    int products;        // gemm_f16p_ws.hip: 4 (0 reads as 4) = all four products above, weights exact; 3 = without P2 x a_hi (weights rounded to 22 bits)
    // ---- common
    int c_blocked;       // C is the tile-blocked gate matrix (g_index, ncols = N) instead of row-major [M][ldc]
    int ldw;
    const float *bias;   // [N] or nullptr
    float *C;
    int M, N, K;         // gemm_f16p.hip: K = the padded width (multiple of 32)
    int lda, ldc;
    // a_mode 0: row m of A is A + m*lda.
    // a_mode 1 (gemm.hip only): A is canonical [B][T][K]; row m = (tile*T + t)*SEQ_TILE + j reads sequence
    //           b = tile*SEQ_TILE + j at frame t (zeros when b >= B).
    //           (rows of padded sequences and rows past the tile's range are clamped, never stored)
    int a_mode, B, T;
    float leaky_slope;   // act: v >= 0 ? v : slope*v when act == 1
    int act;
    // Device-side kernel selection for caller-supplied features (uvad_classify): when `gate` is set the kernel
    // returns at once unless (*gate != 0) == gate_run_if_set.  *gate is written by launch_split_features() earlier on
    // the same stream, so the f16 split never feeds an operand outside the f16 range to the matrix cores and no host
    // sync is needed.
    const int *gate;
    int gate_run_if_set;
    // gemm_f16p_ws.hip, optional (time-chunked projections, uvad_api.hip): instead of every row tile 0 .. ceil(M / 128) - 1, the column
    // tiles of direction d (ws_dirs = 2: d = 0 for columns [0, N / 2), d = 1 for the rest; ws_dirs = 1: d = 0 for all) process the
    // 128-row tiles ws_tiles[ws_off[d] .. ws_off[d] + ws_len[d]) -- device memory, read with scalar loads.  nullptr: all row tiles.
    const int *ws_tiles;
    int ws_off[2], ws_len[2], ws_dirs;
};
hipError_t launch_gemm(const GemmArgs &a, hipStream_t s);
int gemm_padded_k(int K);        // gemm.hip: K rounded up to its K-step (weight row padding)
hipError_t launch_gemm_f16p(const GemmArgs &a, hipStream_t s);
int gemm_f16p_padded_k(int K);   // gemm_f16p.hip: plane row width for a true width K (multiple of 32)
// gemm_f16p_ws.hip: the same contraction for the tile-blocked gate matrix (c_blocked, no activation) as a weight-stationary persistent
// kernel; bit-identical output.  `counters` = gemm_f16p_ws_counter_bytes() of device memory (zeroed by the launcher on the stream).
bool gemm_f16p_ws_supported(const GemmArgs &a, int n_cu);
size_t gemm_f16p_ws_counter_bytes();
hipError_t launch_gemm_f16p_ws(const GemmArgs &a, unsigned *counters, int n_cu, hipStream_t s);
size_t weight_plane_elems(int N, int ldw);
bool split_weights_f16x3(const float *w, int N, int ldw, unsigned short *out /*3 * weight_plane_elems*/, float *wscale);   // false: not representable
// canonical f32 features [B][T][F] -> tile-major K-blocked f16 planes (tiles*T*4 rows, Fp columns); *flag (optional) = 1 if a value is non-finite or
// outside the f16 range
hipError_t launch_split_features(const float *x, int B, int T, int F, int Fp, int tiles, unsigned short *xh, unsigned short *xl, int *flag,
                                 hipStream_t s);

// ---- lstm.hip -----------------------------------------------------------------------------
// One layer, all directions: grid (tiles, dirs).  G holds x_t*W_ih^T + b_ih + b_hh with column
// dir*4H + u*4 + gate, tile-blocked (g_index with ldg columns); Y gets h_t at column dir*H + u.  Rows as above.
struct LstmArgs {
    const float *G; int ldg;
    const float *Whh_packed;     // per dir: register image, see pack_whh()
    // 16-sequence form (H = 128 only, pack_whh16h): per dir the P0 / P1 register image, the P2 LDS image, and 2^-S
    const unsigned *Whh16h_regs; const unsigned short *Whh16h_p2; const float *whh16h_scale;
    // ... and (optional) the P2 image as bf8 bytes for the 8-bit matrix pipe (pack_whh16h_p2q; nullptr: the f16 image is used) with the
    // E8M0 scale operand that undoes its power-of-two shift
    const unsigned short *Whh16h_p2q; int p2q_scale;
    float *Y; int ldy;           // f32 output (exact-f32 GEMM mode), or
    unsigned short *Yh, *Yl;     // the two K-blocked f16 planes (ldy columns) h ~= hi + lo * 2^-11 the f16p GEMM of the next layer reads (Y == nullptr)
    int products;                // 16-sequence form: 4 (0 reads as 4) or 3 (no P2 plane: see GemmArgs)
    int tiles, T, H, dirs;
    // 4-sequence form, optional: run `steps` time steps only (0 = all T): direction 0 frames t_begin[0] .. t_begin[0] + steps - 1 in
    // ascending order, direction 1 frames t_begin[1] .. t_begin[1] + steps - 1 in descending order, from / to the carried state
    // (h0, c0 -> hN, cN).  Rows are addressed with the whole sequence length T.
    int steps, t_begin[2];
    int tile_mode;               // sequences per workgroup: 0 = by estimated time, 4, 16 (see launch_lstm)
    int n_cu;                    // compute units of the device (0 = 256)
    // optional carried state (streaming): [dirs][tiles*SEQ_TILE][H], nullptr = zeros / discard
    const float *h0, *c0; float *hN, *cN;
};
hipError_t launch_lstm(const LstmArgs &a, hipStream_t s, int *tile_used = nullptr);
int lstm_auto_tile(int tiles, int dirs, int H, int n_cu);   // what tile_mode 0 picks (4 or 16)
// elements of the packed W_hh image for one direction
size_t whh_packed_elems(int H);
int lstm_waves(int H);   // waves per recurrent workgroup (8 at H = 128: two per SIMD)
// host-side packer: torch w_hh [4H][H] (rows i,f,g,o) -> register image
void pack_whh(const float *w_hh, int H, float *out);
size_t whh16h_regs_elems();
size_t whh16h_p2_elems();
bool pack_whh16h(const float *w_hh, unsigned *regs, unsigned short *p2, float *wscale);   // false: a weight is non-finite
size_t whh16h_p2q_elems();
// the P2 image as bf8 (E5M2) bytes, columns in the kernel's k order; false if some element is not exactly representable (the f16 image
// then stays in use); *scale = the E8M0 scale operand (127 - shift)
bool pack_whh16h_p2q(const float *w_hh, unsigned short *p2q, int *scale);

// ---- head.hip -----------------------------------------------------------------------------
// logit = Z[m][:K] . w + b ; prob = sigmoid(logit); written at canonical [b][t] (b < B only).
struct ClsArgs {
    const float *Z; int ldz, K;
    const float *w, *b;
    float *logits, *probs;   // either may be nullptr
    int tiles, T, B;
    int ld_out;              // row stride of logits / probs (>= T)
};
hipError_t launch_classifier(const ClsArgs &a, hipStream_t s);
// rows (tile-major) -> canonical [B][T][W] copy, for the parity taps (src_lo != nullptr: src / src_lo are f16 planes)
hipError_t launch_untile(const void *src, const void *src_lo, int lds_, int W, float *dst, int tiles, int T, int B, hipStream_t s);
// threshold 0.5 + binary median
hipError_t launch_median(const float *probs, int B, int T, int kernel, uint8_t *labels, hipStream_t s);
// 0/1 label rows -> ordered (start frame, first non-speech frame) pairs per row + the number of runs
hipError_t launch_runs(const uint8_t *labels, int B, int T, int max_runs, int *runs, int *counts, hipStream_t s);
// per-row {false alarm, missed detection} frame counts of 0/1 label rows
hipError_t launch_der(const uint8_t *pred, const uint8_t *gt, int B, int T, uint32_t *counts, hipStream_t s);

// one wave that busy-waits `ticks` of the 100 MHz constant clock, then (optionally) stores the waited ticks
hipError_t launch_spin(unsigned long long ticks, unsigned long long *sink, hipStream_t s);

// ---- head_fused.hip: leaky_relu(y W1^T + b1) -> leaky_relu(. W2^T + b2) -> . w + b -> sigmoid in one kernel (two 128-unit layers, split-f16 mode) ----
struct HeadArgs {
    const unsigned short *Yh, *Yl;   // K-blocked f16 planes of the LSTM output (K1 columns), tile-major rows
    long long M; int K1;             // rows (tiles * T * SEQ_TILE), input width (256 or 128)
    const unsigned short *W1, *W2;   // three exact f16 planes each (split_weights_f16x3: N = 128 rows, K1 / 128 columns)
    float w1scale, w2scale;
    const float *b1, *b2, *wc, *bc;  // biases [128], classifier row [128] and bias [1]
    float slope;
    float *logits, *probs;           // canonical [b][t] (b < B only), row stride ld_out; either may be nullptr
    int tiles, T, B, ld_out;
    unsigned *counter;               // one word of device memory (zeroed by the launcher)
    int products;                    // 4 (0 reads as 4) or 3 (no P2 plane: see GemmArgs)
};
bool head_fused_supported(int K1, int lin_hidden, int lin_layers, long long M, int n_cu);
hipError_t launch_head_fused(const HeadArgs &a, int n_cu, hipStream_t s);

// ---- fbank.hip ----------------------------------------------------------------------------
struct FbankTables {            // device pointers owned by the ctx
    const float *window;        // [frame_len]
    const int *mel_start;       // [n_mels] first bin with non-zero weight
    const int *mel_len;         // [n_mels] number of bins
    const float *mel_w;         // [n_mels][mel_stride] weights (zero padded)
    const float *mel_wt;        // the same transposed, the LDS image of the mel stage: [mel_stride][mel_image_ld(n_mels)] (zero padded), x 1/4
    int mel_stride;             // the uniform trip count of the band loop: the longest band rounded up to a multiple of 4 bins
    const float *tw512;         // [512][2] (cos, -sin)(2*pi*j/512): forward FFT twiddles
    int nyquist;                // 1 if any filter weighs bin n_fft / 2 (kaldi-style tables, the reference's, carry a zero column there: the
                                // power of that bin -- a fifth round of the spectrum split for one lane's sake -- is then not computed)
};
// The mel stage's weight image (fbank_pair.h): one row of mel_image_ld floats per bin-in-band, a compile-time row stride in the kernels
// (the weights a lane needs per iteration sit at immediate offsets).
__host__ __device__ inline int mel_image_ld(int n_mels) { return n_mels <= 64 ? 64 : 128; }
__host__ __device__ inline size_t mel_image_floats(int mel_stride, int n_mels) { return (size_t)mel_stride * mel_image_ld(n_mels); }
struct FbankArgs {
    const void *pcm; int pcm_is_i16;
    int B; int64_t S; int64_t T;
    int64_t row_stride;         // elements between rows of pcm (0 = S)
    int frame_len, frame_shift, n_mels;
    float preemph, log_floor; int remove_dc, snip_edges;
    float *feats;               // [B][T][n_mels]
    // Instead of feats (plane_hi != nullptr): the two K-blocked f16 planes of the features (plane_w columns, a multiple of 16, the
    // columns [n_mels, plane_w) written as zero) with rows in the classifier's tile-major order, padding sequences zeroed: the A
    // operand of the first projection GEMM (gemm_f16p.hip), bit-identical to what launch_split_features makes of feats
    unsigned short *plane_hi, *plane_lo; int plane_w;
    // Streaming (uvad_stream_step): instead of pcm the rows are VIRTUAL -- row b = [tail of the previous steps (vs_tail samples; on the
    // first step the reflection of the chunk's head) | this step's chunk], read from vs_offset on -- and the workgroups of the first
    // tile also write the next tail (the last vs_tail samples of that row) to vs_tail_out.  vs_chunk == nullptr: plain pcm rows.
    const float *vs_chunk, *vs_tail_in; float *vs_tail_out;
    int vs_tail, vs_chunk_len, vs_first, vs_n_left, vs_offset;
    FbankTables tab;
};
hipError_t launch_fbank(const FbankArgs &a, hipStream_t s);
size_t fbank_lds_bytes(const FbankArgs &a);
// streaming: staging[b] = [tail (frame_len samples, reflection-filled on the first step) | chunk];
// new tail = last `tail` samples of staging
hipError_t launch_stream_stage(const float *chunk_pcm, int B, int chunk, int tail, int n_left, int first_step,
                               const float *tail_in, float *tail_out, float *staging, hipStream_t s);

// ---- lstm_stack.hip: every layer of a causal (one-direction, H = 128) stack for T <= LSTM_STACK_TMAX new frames in ONE launch, carried
//      (h, c) updated in place: the streaming step (uvad_stream_step).  Exact f32.
constexpr int LSTM_STACK_TMAX = 4, LSTM_STACK_MAX_LAYERS = 8, LSTM_STACK_MAX_LIN = 4;
struct LstmStackArgs {
    const float *feats; int kin0;                 // canonical [B][T][kin0] f32 features
    const float *wih[LSTM_STACK_MAX_LAYERS];      // register images of W_ih (pack_lstm_image, K = kin0 for layer 0, 128 after)
    const float *whh[LSTM_STACK_MAX_LAYERS];      // register images of W_hh (pack_whh)
    const float *bias[LSTM_STACK_MAX_LAYERS];     // [4H] b_ih + b_hh in (unit, gate) order
    int n_layers;
    float *h, *c; size_t layer_stride;            // carried state [layer][tiles * SEQ_TILE][128] (layer_stride floats apart)
    float *Y; int ldy;                            // last layer's output: f32 rows (tile-major), or
    unsigned short *Yh, *Yl;                      // its two K-blocked f16 planes of ldy columns (Y == nullptr)
    int tiles, T, B;
    // optional head in the same launch (logits != nullptr): n_lin feed-forward layers of 128 units (leaky_relu) as register images
    // (pack_fc_image), the classifier row and bias; logits / probs at canonical [b][t] (b < B), row stride ld_out
    const float *lin_w[LSTM_STACK_MAX_LIN], *lin_b[LSTM_STACK_MAX_LIN]; int n_lin;
    const float *cls_w, *cls_b; float slope;
    float *logits, *probs; int ld_out;
    // optional feature stage in the same launch (fb_on): the step's virtual rows (fb.vs_*: chunk + carried tail) are framed and
    // transformed by the workgroup that consumes them (fbank_pair.h) and `feats` is not read; fb.tab, fb.frame_len, ... as in FbankArgs
    FbankArgs fb; int fb_on;
};
// LDS the feature stage of lstm_stack_kernel needs beside the kernel's static arrays (0 if it cannot run for these arguments)
size_t lstm_stack_fb_lds_bytes(const FbankArgs &fb, int T);
bool lstm_stack_supported(int hidden, int dirs, int in_dim, int T, int n_layers);
hipError_t launch_lstm_stack(const LstmStackArgs &a, hipStream_t s);
size_t lstm_image_elems(int K);
void pack_lstm_image(const float *w /*[4 * 128][K], torch row order*/, int K, float *out);
size_t fc_image_elems();                                                   // a 128 x 128 feed-forward matrix as a register image
void pack_fc_image(const float *w /*[128][128], torch nn.Linear.weight*/, float *out);

// ---- sincnet.hip: SincNet front end of PyanNet (conv + |.| + maxpool(3) + instance-norm statistics) ----------
struct SincConvArgs {
    const float *in; long long in_bstride; int Cin, Lin;   // in[b*in_bstride + ci*Lin + x]
    const float *in_scale, *in_shift;                      // [B][Cin]: x*scale + shift applied while staging (previous norm)
    int in_lrelu; float slope;                             // then leaky_relu (layers after the first)
    const float *Wt2;                                      // [Kp][NW]: W^T, zero padded (NW = 32*ceil(Cout/32))
    const float *bias;                                     // [NW] zero padded
    int Kw, stride, Ktot, Kp, Cout, do_abs;                // Ktot = Cin*Kw, Kp = Ktot rounded up to a multiple of 8
    int Lconv, Lpool, ntiles;                              // ntiles = ceil(Lpool / plan.pt)
    float *out;                                            // [B][Cout][Lpool] pooled, before the norm
    float *partials;                                       // [B][ntiles][plan.phases][NW][2] (sum, M2 about the group mean) per statistics group
    int B;
    int n_cu;                                              // compute units of the device (persistent grid size); 0 = 256
};
hipError_t launch_wav_stats(const float *wav, int B, long long S, long long row_stride, const float *gamma, const float *beta, float eps,
                            float *scale, float *shift, hipStream_t s);
struct SincConvPlan { int waves, pt, phases; };            // waves per workgroup, pooled outputs per tile, statistics groups per tile
SincConvPlan sinc_conv_plan(const SincConvArgs &a);        // needs Cin, Cout, Kw, stride, Kp
hipError_t launch_sinc_conv(const SincConvArgs &a, hipStream_t s);
size_t sinc_conv_lds_bytes(const SincConvArgs &a, int NT, int waves);
int sinc_conv_ept(const SincConvArgs &a, int waves);   // window elements per thread held in registers (<= 8 single-channel, <= 48 otherwise)
hipError_t launch_norm_finalize(const float *partials, int B, int ntiles, int pt, int phases, int NW, int C, int L, const float *gamma,
                                const float *beta, float eps, float *scale, float *shift, hipStream_t s);
hipError_t launch_sinc_out(const float *P, const float *scale, const float *shift, int B, int C, int L, float slope, float *feats, int ldf,
                           hipStream_t s);

// ---- sincnet_f16p.hip: the same three stages on the f16 matrix cores (f32-equivalent split arithmetic), channel-minor intermediates ----
struct SincF16Args {
    const float *in; long long in_bstride; int Lin;        // stage 0: in[b * in_bstride + x] (the waveform); stages 1, 2: in[(b * Lin + x) * cst_in + c]
    const float *in_scale, *in_shift; int n_in;            // [B][n_in]: x * scale + shift applied while staging (n_in = 1, or the real channel count)
    float slope;                                           // leaky_relu of the previous stage (stages 1, 2)
    const unsigned short *Wfrag; float wscale;             // sinc_f16p_pack_weights
    const float *bias;                                     // [cst] zero padded
    int Lpool, ntiles;                                     // ntiles = sinc_f16p_ntiles(Lpool)
    float *out;                                            // [B][Lpool][cst] pooled, before the norm (channels past Cout written as zero)
    float *partials;                                       // [B][ntiles][4 slots][3][cst]: (count, sum, M2 about the set's own mean); slot 0 per tile,
                                                           // stage 0's channels 64 .. 79 one slot per wave
    int B, n_cu;
};
bool sinc_f16p_supported(int n_filters, int kernel_size, int stride, int c2, int k2, int c3, int k3);
int sinc_f16p_ksteps(int stage);       // 32-deep k-steps of the stage's contraction (K = 32 * ksteps: the packed weight row length)
int sinc_f16p_cst(int stage);          // floats per output row (channels padded to whole 16-channel tiles)
int sinc_f16p_ntiles(long long Lpool);
size_t sinc_f16p_partial_floats(int stage, int B, int ntiles);
size_t sinc_f16p_wfrag_elems(int stage);
bool sinc_f16p_pack_weights(int stage, const float *w, int nrows, int ldk, unsigned short *out, float *wscale);
hipError_t launch_sinc_conv_f16p(int stage, const SincF16Args &a, hipStream_t s);
hipError_t launch_norm_finalize_f16p(int stage, const float *partials, int B, int ntiles, int C, int L, const float *gamma, const float *beta, float eps,
                                     float *scale, float *shift, hipStream_t s);
hipError_t launch_sinc_out_f16p(const float *P, const float *scale, const float *shift, int B, int C, int CST, int L, float slope, float *feats, int ldf,
                                hipStream_t s);

}  // namespace uvad

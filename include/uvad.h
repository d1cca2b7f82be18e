/*
 * uvad.h -- C ABI of libuvad.so: MI355X (gfx950) voice-activity hot path.
 *
 *   16 kHz PCM --fbank--> log-mel (B,T,F) --classify--> per-frame logit / probability
 *
 * The reference (arnavsshah/universal-voice-activity-detection) is pure Python and has no
 * FFI for this path; its boundary is a set of torch.nn.Module / lhotse call contracts.
 * Each entry point below names the reference interface it stands behind (paths relative to
 * the reference root).  The Python host in universal-voice-activity-detection_amd/ binds
 * these with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - return 0 on success, negative on error: UVAD_E_ARG bad argument, UVAD_E_HIP HIP runtime
 *     error (no GPU, launch failure), UVAD_E_STATE wrong call order (not finalized ...),
 *     UVAD_E_WORKSPACE workspace too small, UVAD_E_UNSUPPORTED configuration outside what the
 *     kernels implement.  uvad_last_error() gives the text.
 *   - every pointer named d_* is a DEVICE pointer owned by the caller; the library never
 *     allocates or frees caller tensors.  It owns only the weights / tables inside uvad_ctx.
 *   - compute calls are ASYNCHRONOUS on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and perform no allocation or synchronisation => hipGraph-capturable
 *     (exception: uvad_stream_step, see there).  Every call makes the context's device current
 *     (hipSetDevice) before it enqueues, so a multi-GPU process may interleave contexts freely.
 *   - one ctx per (device, model); a ctx is NOT thread-safe (the reference drives the model
 *     from a single thread: Trainer(devices=1), src/scripts/predict.py:79-85).
 *   - there is NO CPU fallback: on a machine without a gfx950 device uvad_create fails.
 */
#ifndef UVAD_H
#define UVAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVAD_OK             0
#define UVAD_E_ARG         -1
#define UVAD_E_HIP         -2
#define UVAD_E_STATE       -3
#define UVAD_E_WORKSPACE   -4
#define UVAD_E_UNSUPPORTED -5

#define UVAD_ABI_VERSION 4

typedef struct uvad_ctx uvad_ctx; /* opaque */

/* Feature-stage configuration = lhotse FbankConfig as constructed at
 * src/datasets/ami/utils.py:153 and src/utils/helper.py:120 (all defaults but sampling_rate). */
typedef struct {
    int sample_rate;   /* 16000 */
    int frame_len;     /* samples per frame: 400 (25 ms) */
    int frame_shift;   /* hop: 160 (10 ms) */
    int n_fft;         /* 512 (next power of two of frame_len); only 512 is implemented */
    int n_mels;        /* 80 in the reference (config/config.py:33), 64 in BASELINE cfg 2 */
    float preemph;     /* 0.97 */
    float low_hz;      /* informational (the mel matrix is uploaded by uvad_set_tables) */
    float high_hz;     /* informational */
    float log_floor;   /* FLT_EPSILON */
    int remove_dc;     /* 1 */
    int snip_edges;    /* 0 (reflect-padded, T = (S + shift/2) / shift) */
} uvad_fbank_cfg;

/* Classifier configuration = constructor arguments of PyanNet2,
 * src/models/segmentation/PyanNet2.py:60-90 (LSTM_DEFAULTS / LINEAR_DEFAULTS / encoding_dim). */
typedef struct {
    int in_dim;        /* encoding_dim */
    int hidden;        /* lstm.hidden_size: 128 (and 64) run the register-resident recurrent kernels; any other size with hidden x
                        * directions a multiple of 32 (<= 1024) runs a generic recurrence (correct to the same bound, slow: W_hh is
                        * streamed from L2 every step); other sizes: UVAD_E_UNSUPPORTED */
    int num_layers;    /* lstm.num_layers: 4 */
    int bidirectional; /* lstm.bidirectional: 1 */
    int lin_hidden;    /* linear.hidden_size: 128 */
    int lin_layers;    /* linear.num_layers: 2 */
    float leaky_slope; /* F.leaky_relu default 0.01 (PyanNet2.py:185) */
} uvad_model_cfg;

/* ABI version of the loaded library (== UVAD_ABI_VERSION of the header it was built from). */
int uvad_abi_version(void);

/* Replaces: PyanNet2.__init__/build (PyanNet2.py:69-152) + Fbank(FbankConfig(...)) construction
 * (ami/utils.py:153).  Either cfg may be NULL if that stage is not used. */
int uvad_create(int device, const uvad_fbank_cfg *fb, const uvad_model_cfg *model, uvad_ctx **out);

/* Host tables for the feature stage: window[frame_len]; mel[n_mels][n_fft/2+1] row-major
 * (any banded non-negative matrix; the library converts it to per-filter (start,len,weights)).
 * Replaces the window / filterbank buffers lhotse builds inside Fbank (third party). */
int uvad_set_tables(uvad_ctx *, const float *window, const float *mel);

/* One tensor of the PyanNet2 state_dict, by its torch key ("lstm.weight_ih_l0_reverse",
 * "linear.0.weight", "classifier.bias", ...; an optional "model." Lightning prefix is stripped),
 * host pointer, row-major f32.  Replaces nn.Module.load_state_dict / VadModel.load_from_checkpoint
 * (src/scripts/predict.py:77). */
int uvad_set_weight(uvad_ctx *, const char *torch_key, const float *host, const int64_t *shape, int ndim /* 1..3 */);

/* Checks that every tensor is present, repacks into kernel layouts and uploads.  May be called again after
 * further uvad_set_weight calls (weight hot-swap): it waits for the device to go idle, frees the previous
 * upload and replaces it -- so every hipGraph captured from this context earlier (it bakes the old device pointers of the
 * weights into its kernel nodes) is INVALID afterwards and must be captured again. */
int uvad_finalize(uvad_ctx *);
/* Contexts of one process that are finalized with identical tensors, model / SincNet configuration and device SHARE the packed weights on the
 * device (read-only there; repacked and uploaded once: a pipeline of twelve contexts pays the ~45 ms of host-side packing once, not twelve times).
 * Transparent: a context that swaps a weight and finalizes again gets a block of its own, the others keep theirs; a block is freed with its last
 * context.  Returns how many contexts currently use this context's block (1 = not shared; 0 before uvad_finalize; negative on error). */
int uvad_weights_shared_by(const uvad_ctx *);

/* T for S samples (lhotse framing; data/test_data.py:23 pins T = S/160 for 5 s cuts). */
int64_t uvad_num_frames(const uvad_ctx *, int64_t S);

/* Bytes of caller-provided device workspace uvad_classify / uvad_forward need for B sequences of
 * T frames (uvad_forward: pass T = uvad_num_frames(S)). */
size_t uvad_workspace_bytes(const uvad_ctx *, int B, int64_t T);

/* Replaces: Fbank.extract_batch (lhotse; call sites ami/utils.py:157-163, helper.py:122-130).
 * d_pcm [B][S] f32 in [-1,1]  ->  d_feats [B][T][n_mels] f32. */
int uvad_fbank(uvad_ctx *, const float *d_pcm, int B, int64_t S, float *d_feats, void *stream);

/* Same with int16 PCM (wav ingest; halves the HBM read).  Samples are scaled by 1/32768. */
int uvad_fbank_i16(uvad_ctx *, const int16_t *d_pcm, int B, int64_t S, float *d_feats, void *stream);

/* Replaces: PyanNet2.forward (PyanNet2.py:154-187) = VadModel.forward (vad_engine.py:69-80).
 * d_feats [B][T][in_dim] -> d_logits [B][T] (pre-sigmoid, may be NULL) and d_probs [B][T]
 * (what forward returns, viewed as (B,T,1); may be NULL). */
int uvad_classify(uvad_ctx *, const float *d_feats, int B, int T, float *d_logits, float *d_probs,
                  void *d_workspace, size_t ws_bytes, void *stream);

/* uvad_fbank + uvad_classify without returning the features (they stay in the workspace). */
int uvad_forward(uvad_ctx *, const float *d_pcm, int B, int64_t S, float *d_logits, float *d_probs,
                 void *d_workspace, size_t ws_bytes, void *stream);

/* The same from 16-bit PCM as read from a wav file (samples scaled by 1/32768): what the reference's predict flow does per batch
 * (decode audio -> features -> model, src/scripts/predict.py:98 with the offline feature step of ami/utils.py:153-163 folded in). */
int uvad_forward_i16(uvad_ctx *, const int16_t *d_pcm, int B, int64_t S, float *d_logits, float *d_probs,
                     void *d_workspace, size_t ws_bytes, void *stream);

/* Debug / parity taps: copy of the last LSTM layer output [B][T][hidden*dirs] and of the last
 * feed-forward activation [B][T][lin_hidden] from the most recent uvad_classify on this
 * workspace (async on stream).  Either pointer may be NULL.  Where the fused head ran (two 128-unit feed-forward layers, large
 * launch) the feed-forward tap is recomputed from the LSTM output with the per-layer kernels -- the same bits in GEMM modes 1 / 2;
 * in mode 3 (three products) no kernel can reproduce the fused head's activation and d_lin_out != NULL returns UVAD_E_UNSUPPORTED. */
int uvad_get_taps(uvad_ctx *, int B, int T, float *d_lstm_out, float *d_lin_out,
                  const void *d_workspace, void *stream);

/* Streaming (BASELINE cfg 5; the reference has no streaming mode): causal model
 * (bidirectional = 0) with carried state, B streams advancing in lockstep.  Semantics: the logits of
 * frame t are EXACTLY those of the offline path (uvad_forward on the whole signal) because features use
 * the same centred framing; a frame is emitted once its last sample (t*shift - (len-shift)/2 + len) has
 * arrived, i.e. with a look-ahead of 280 samples at the reference geometry.  Each call consumes
 * d_pcm_chunk [B][chunk] and writes the newly complete frames to d_logits [B][ld_logits] (row b, columns
 * 0..k-1); the return value is k >= 0 (same for every stream) or a negative error.  d_state is caller-owned
 * device memory of uvad_stream_state_bytes(ctx, B) bytes holding the PCM tail and (h, c) of every layer;
 * uvad_stream_reset (re)starts all B streams.  The right-edge reflection of the offline path needs the end
 * of the signal and is therefore never produced (streams are open-ended).
 * uvad_stream_step is asynchronous but NOT replay-safe by itself under hipGraph capture: the number of complete frames, the
 * PCM-tail ping-pong parity and the first-chunk reflection are host-side counters baked into the launch arguments at
 * enqueue time.  Replay is offered explicitly: uvad_stream_peek says what the NEXT step will bake in -- two steps with the
 * same (B, chunk, k, offset, parity, first) and the same buffers enqueue identical work, so a graph captured around one
 * uvad_stream_step can be replayed for the other, followed by uvad_stream_advance, which moves the counters exactly as the
 * step would have and returns its k (VadRuntime.stream_step does this; at the reference geometry a stream group settles
 * into two graphs, one per parity). */
size_t uvad_stream_state_bytes(const uvad_ctx *, int B);
size_t uvad_stream_workspace_bytes(const uvad_ctx *, int B, int chunk);
int uvad_stream_reset(uvad_ctx *, void *d_state, int B, void *stream);
int uvad_stream_step(uvad_ctx *, const float *d_pcm_chunk, int B, int chunk, void *d_state,
                     float *d_logits, int ld_logits, void *d_workspace, size_t ws_bytes, void *stream);
int uvad_stream_peek(const uvad_ctx *, const void *d_state, int chunk, int *k, int64_t *offset, int *parity, int *first);
int uvad_stream_advance(uvad_ctx *, void *d_state, int chunk);

/* Replaces: median_filter (src/utils/helper.py:66-97) as used by VadModel.predict_step
 * (vad_engine.py:204-211): threshold 0.5 then odd `kernel`-tap median, zero padded edges.
 * d_probs [B][T] -> d_labels [B][T] uint8 (0/1). */
int uvad_median_filter(uvad_ctx *, const float *d_probs, int B, int T, int kernel, uint8_t *d_labels,
                       void *stream);

/* Replaces: the per-frame run-length walk of get_new_cuts (src/scripts/predict.py:472-490) on 0/1 frame rows
 * (the output of uvad_median_filter): run i of row b is d_runs[b][i] = {first speech frame, first non-speech frame
 * after it (T if the run is open at the end)}, in order; d_counts[b] = number of runs in the row (runs beyond
 * max_runs are counted but not stored; a row of T frames has at most (T+1)/2).  The host turns frames into seconds
 * with the reference's rounding (round(k*shift, 2), drop empty intervals). */
int uvad_label_runs(uvad_ctx *, const uint8_t *d_labels, int B, int T, int max_runs, int32_t *d_runs /*[B][max_runs][2]*/,
                    int32_t *d_counts /*[B]*/, void *stream);

/* Replaces: get_false_alarm / get_missed_detection (src/scripts/predict.py:666-673) on 0/1 frame rows:
 * d_counts [B][2] uint32 = {#(gt == 0 and pred == 1), #(gt == 1 and pred == 0)} per row; the reference's
 * FA / MD / DER are these counts divided by the row length (DER = FA + MD, vad_engine.py:102-105). */
int uvad_der_counts(uvad_ctx *, const uint8_t *d_pred, const uint8_t *d_gt, int B, int T, uint32_t *d_counts,
                    void *stream);

/* ---- SincNet front end (PyanNet) ------------------------------------------------------------
 * Constructor arguments of SincNet (src/models/blocks/sincnet.py:33-70) as PyanNet builds it
 * (src/models/segmentation/PyanNet.py:62, 91-95: stride 10). */
typedef struct {
    int stride;        /* hop of the sinc filter bank: SINCNET_DEFAULTS["stride"] = 10 */
    int n_filters;     /* 80 (40 cos + 40 sin band-pass filters) */
    int kernel_size;   /* 251 */
    int c2, k2;        /* Conv1d(80, 60, 5) */
    int c3, k3;        /* Conv1d(60, 60, 5); c3 must equal the classifier's encoding_dim */
    float leaky_slope; /* F.leaky_relu default 0.01 */
    float eps;         /* InstanceNorm1d eps 1e-5 */
} uvad_sincnet_cfg;

/* Replaces: SincNet.__init__ (sincnet.py:33-70).  Adds the stage to a context created with a model
 * configuration; its tensors go through uvad_set_weight under their state_dict names
 *   sincnet.wav_norm1d.{weight,bias} [1]      sincnet.norm1d.{0,1,2}.{weight,bias} [C]
 *   sincnet.conv1d.{1,2}.weight [Cout][Cin][k]  sincnet.conv1d.{1,2}.bias [Cout]
 * plus the MATERIALISED first-layer filter bank  sincnet.conv1d.0.filters [n_filters][kernel_size]
 * (what asteroid_filterbanks.ParamSincFB.filters() returns from low_hz_ / band_hz_; the host computes it,
 * see universal-voice-activity-detection_amd/sincnet.py), and uvad_finalize packs them. */
int uvad_sincnet_configure(uvad_ctx *, const uvad_sincnet_cfg *);

/* Output frames for S samples: three (conv, MaxPool1d(3)) stages; 80000 -> 293
 * (src/datasets/custom_vad.py:47, src/utils/receptive_field.py:165-193). */
int64_t uvad_sincnet_num_frames(const uvad_ctx *, int64_t S);

/* Device workspace of uvad_sincnet (pooled activations + norm statistics) for B waveforms of S samples;
 * uvad_forward_wav needs this plus uvad_workspace_bytes(ctx, B, frames). */
size_t uvad_sincnet_workspace_bytes(const uvad_ctx *, int B, int64_t S);

/* Replaces: SincNet.forward (sincnet.py:72-103) followed by the rearrange of PyanNet.forward (PyanNet.py:179).
 * d_wav [B][S] f32 (the reference's (batch, 1, samples) tensor) -> d_feats [B][frames][c3]. */
int uvad_sincnet(uvad_ctx *, const float *d_wav, int B, int64_t S, float *d_feats, void *d_workspace,
                 size_t ws_bytes, void *stream);

/* Replaces: PyanNet.forward (PyanNet.py:162-195): uvad_sincnet + uvad_classify; d_logits / d_probs [B][frames]. */
int uvad_forward_wav(uvad_ctx *, const float *d_wav, int B, int64_t S, float *d_logits, float *d_probs,
                     void *d_workspace, size_t ws_bytes, void *stream);

/* Which kernel runs the time-parallel contractions (input projections, feed-forward layers):
 *   0  exact f32: v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain, bit-compatible with f32 FMA arithmetic;
 *   1  (default) f32-accurate on the f16 matrix cores: weights scaled by a power of two and split on the host into THREE
 *      f16 planes that reproduce them exactly, activations into two planes (22 bits) by the kernel that produces them;
 *      four v_mfma_f32_32x32x16_f16 products per term set in two f32 accumulator sets (dropped terms <= 2^-22 relative,
 *      below the rounding noise of the f32 accumulation).
 *      Needs operands inside the f16 range (|x| < 65504).  The library guarantees that without the caller's help:
 *      weights (and the bound they put on the feed-forward activations) are checked by uvad_finalize and a context
 *      that fails runs mode 0; features handed to uvad_classify are checked on the device and a batch that fails runs
 *      its first projection in mode 0 (no host synchronisation: both kernels are enqueued, a device flag picks one).
 *   2  the arithmetic of mode 1 with the tile-streaming kernel (gemm_f16p_kernel) for every GEMM: mode 1 runs the input
 *      projections of large launches in a weight-stationary persistent kernel (gemm_f16p_ws_kernel: the weight planes of a
 *      128-column tile stay in registers, only the activation planes stream through LDS) whose output is BIT-IDENTICAL;
 *      mode 2 exists for A/B measurements and as the reference of that identity test.
 *   3  mode 1 with THREE products per term set in the kernels of large launches (the weight-stationary projection, the fused head,
 *      the 16-sequence recurrence): the P2 x a_hi product is dropped, i.e. the weights are rounded to their two leading f16 planes
 *      (22 bits, the precision the activations already have).  The matrix cores then do 25 % less work, and because a step in
 *      flight runs at the socket's power limit that is time: +9 % frames/s at BASELINE cfg 2, recurrence launch
 *      1.85 -> 1.60 ms.  Logit error on contractive networks as in mode 1 (3e-7 at weights x2); on the near-chaotic x4 test
 *      network twice mode 1's distance from the float64 truth (a rounded weight is a slightly different network), which is why
 *      it is not the default.  Small launches (streaming steps) run mode 1's kernels: outputs of different launch sizes then
 *      differ in the last bits.
 * The SincNet front end (uvad_sincnet, uvad_forward_wav) follows the same selector: modes 1 and 3 run its three convolution stages on
 * the f16 matrix cores with the split arithmetic of mode 1 (exact three-plane weights in registers, sincnet_f16p.hip) when the geometry
 * is the reference's (sinc bank of 80 filters x <= 256 taps at stride 10; Conv1d(80 -> <= 64, 5); Conv1d(<= 64 -> <= 64, 5)) and every
 * stage input provably fits the f16 range (|gamma| * sqrt(length) + |beta| < 60000 for the instance norm in front of it); modes 0 and 2,
 * other geometries and inputs outside that bound run the exact-f32 stages (v_mfma_f32_32x32x2_f32, sincnet.hip).  uvad_get_sincnet_form
 * tells which.
 * All are held to the same 1e-4 logit bound by the tests.  Replaces nothing in the reference (torch picks its GEMM). */
int uvad_set_gemm_mode(uvad_ctx *, int mode);
/* What the most recent uvad_sincnet / uvad_forward_wav call of this context ran: 1 = the split-f16 stages, 0 = the exact-f32 stages
 * (also before the first call); negative on error. */
int uvad_get_sincnet_form(const uvad_ctx *);

/* How many sequences one recurrent workgroup owns (the time loop of nn.LSTM, PyanNet2.py:169-172):
 *   4   latency form (v_mfma_f32_4x4x1): B/4 x directions workgroups, the right one up to a few hundred sequences;
 *   16  throughput form (hidden_size 128 only): W_hh * h as four matrix-core products on the same exact three-plane
 *       split as GEMM mode 1 (three v_mfma_f32_16x16x32_f16; the fourth -- the residue plane, exactly bf8 -- on
 *       v_mfma_scale_f32_16x16x128_f8f6f4: uvad_get_p2_on_fp8); a quarter of the workgroups, each 1.35 x as long: a third of the CU-time per
 *       sequence.  The right one for B >= 1024 and for callers that keep several batches in flight on separate contexts;
 *   0   (default) chosen per call: the form with fewer estimated rounds of workgroups over the CUs (uvad_recurrent_tile_for).
 * uvad_get_recurrent_tile returns what the most recent uvad_classify / uvad_forward* call launched (4 or 16; 0 before
 * the first call).  Results agree to rounding between the two (tests/test_gpu_parity.py). */
int uvad_set_recurrent_tile(uvad_ctx *, int sequences);

/* Time chunks of a layer (the time loop of nn.LSTM and the x_t W_ih^T product in front of it, PyanNet2.py:169-172).  One batch alone
 * on the GPU cannot overlap its layers (layer l + 1 needs the backward pass of layer l to its last step), but inside a layer the
 * forward pass at frame t needs the gate rows up to t only and the backward pass those from t on: with n chunks the projection of
 * chunk i + 1 runs on a stream the library owns, on the CUs the 4-sequence recurrence leaves idle, beside the recurrence of chunk
 * i on the caller's stream (events fork and join the two; the call stays asynchronous and capturable once the stream pair has been
 * used outside a capture).  Same kernels and arithmetic per row: outputs are bit-identical to the unchunked call.
 *   0  (default) automatic: T / 96 chunks, at most 6, of geometrically growing length (only the first one's projection is exposed),
 *      when the 4-sequence recurrence is the form in use, it leaves at least a quarter of the CUs idle, the GEMM mode is 1 or 3 and
 *      a side stream concurrent with the caller's was found;
 *   1  off;   2 .. 64  that many chunks wherever the chunked form can run.
 * uvad_get_time_chunks: what the most recent uvad_classify / uvad_forward* call ran (1 = not chunked).
 * Replaces nothing in the reference (its nn.LSTM is one cuDNN / MIOpen call per layer stack). */
int uvad_set_time_chunks(uvad_ctx *, int chunks);
int uvad_get_time_chunks(const uvad_ctx *);
int uvad_get_recurrent_tile(const uvad_ctx *);
/* 1 if the throughput form runs its fourth product (P2 x h, the residue plane of the exact three-way split of W_hh) on the 8-bit
 * matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4): decided by uvad_finalize, which checks that EVERY element of every layer's P2 plane is
 * exactly a bf8 (E5M2) number after one power-of-two shift, so the weights stay exact (the residue of two round-to-nearest f16
 * splits of a 24-bit significand always is: at most two significant bits, within bf8's exponent range of the f16 value it replaces);
 * 0 in GEMM mode 2 (the kernel set kept for comparisons reads P2 from its f16 image, as every mode did before round 4), if a check
 * failed, or if the model has no hidden_size-128 layers.  h enters that one product rounded to fp8 (E4M3). */
int uvad_get_p2_on_fp8(const uvad_ctx *);
/* What mode 0 would launch for a batch of B sequences (4 or 16).  A sweep sharded over n GPUs that wants every utterance to
 * get the same bits as the unsharded run pins all ranks to uvad_recurrent_tile_for(ctx, GLOBAL batch) (tools/run_cfg4.py
 * --reproducible); left alone each rank picks the faster form for its own shard and results agree to rounding. */
int uvad_recurrent_tile_for(const uvad_ctx *, int B);

/* Do two HIP streams run concurrently?  HIP maps streams onto a small pool of hardware queues and two streams that share
 * a queue serialise, which silently defeats "two batches in flight" (ForwardPipeline).  The probe enqueues a 3 ms
 * spinning wave on stream_a and an empty kernel on stream_b and reports whether b's retired while a's was still running:
 * 1 = concurrent, 0 = serialised, negative = error.  Synchronises both streams (call it at set-up time, not per batch).
 * Replaces nothing in the reference (its inference is one batch at a time, src/scripts/predict.py:98). */
int uvad_streams_overlap(uvad_ctx *, void *stream_a, void *stream_b);

/* Per-stage device timing of the most recent uvad_forward/uvad_classify made with timing enabled
 * (uvad_set_timing(ctx, 1) inserts hipEvents on the caller's stream; not graph-capturable while
 * enabled).  ms[0..4] = fbank, input projections, recurrences, feed-forward+classifier, total.
 * Synchronises on the recorded events. */
int uvad_set_timing(uvad_ctx *, int enabled);
int uvad_get_timing(uvad_ctx *, float ms[5]);
/* The same timed call layer by layer: ms[2k] = input projection of LSTM layer k (x * W_ih^T, nn.LSTM inside PyanNet2.forward,
 * PyanNet2.py:169-172), ms[2k+1] = its recurrence; n = capacity of ms in floats (>= 2 * num_layers).  Returns the number of
 * floats written.  bench.py reads the K = 256 projection and one recurrent launch from here for its roofline object. */
int uvad_get_layer_timing(uvad_ctx *, float *ms, int n);

const char *uvad_last_error(const uvad_ctx *);
void uvad_destroy(uvad_ctx *);

#ifdef __cplusplus
}
#endif
#endif /* UVAD_H */

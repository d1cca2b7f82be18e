/*
 * uvad_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the hot path named in BASELINE.json:north_star:
 *   16 kHz PCM -> Kaldi-style log-mel (lhotse Fbank defaults) -> PyanNet2
 *   (N-layer (bi)LSTM -> leaky-relu feed-forward -> Linear(.,1) -> sigmoid).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (libuvad.so) never links,
 * loads or calls anything in oracle/.
 *
 * What each function follows in the reference (paths relative to /root/reference):
 *   orc_fbank      : lhotse Fbank(FbankConfig(sampling_rate=16000)) as called at
 *                    src/datasets/ami/utils.py:153,157-163 and src/utils/helper.py:120-130.
 *                    lhotse is an UN-VENDORED third-party dependency (requirements.txt:13,
 *                    editable sibling checkout, no pinned version) -> the arithmetic below
 *                    restates the published Kaldi/lhotse algorithm.
 *                    PARITY UNPINNED for this function: the reference holds no fixture or
 *                    test for it; only T = S/160 (data/test_data.py:23) and F = 80
 *                    (config/config.py:33) are pinned.
 *   orc_lstm_layer : torch.nn.LSTM as constructed at src/models/segmentation/PyanNet2.py:95
 *                    and run at PyanNet2.py:169-172 (gate order i,f,g,o; zero initial state;
 *                    eval mode => dropout is identity).
 *   orc_classify   : PyanNet2.forward, src/models/segmentation/PyanNet2.py:154-187
 *                    (LSTM -> [leaky_relu(Linear)]* -> Sigmoid(Linear(.,1))).
 *                    PINNED by tests/golden/pyannet2_*.npz, produced by running the
 *                    reference's own PyanNet2 class on CPU (tools/gen_golden.py).
 *   orc_median_filter / orc_intervals :
 *                    src/utils/helper.py:66-97 (threshold 0.5 + scipy.signal.medfilt, zero
 *                    padded edges) and src/scripts/predict.py:472-490 (run-length -> intervals).
 *
 * Precision: dot products are accumulated in double and rounded to float once per
 * output, states are stored as float -- i.e. an fp32 model with (slightly) better
 * than fp32 summation, which is what "CPU reference within 1e-4" is judged against.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ features */

/* SURVEY.md App.A step 1: T = (S + shift/2) / shift when snip_edges == 0,
 * Kaldi snip_edges: 1 + (S - frame_len)/shift. */
int64_t orc_num_frames(int64_t S, int frame_len, int shift, int snip_edges) {
    if (snip_edges) return S < frame_len ? 0 : 1 + (S - frame_len) / shift;
    return (S + shift / 2) / shift;
}

/* kind: 0 povey (hann_sym^0.85), 1 hamming (0.54/0.46 symmetric), 2 hanning, 3 rectangular */
void orc_window(int kind, int n, float *out) {
    for (int i = 0; i < n; ++i) {
        double a = 2.0 * M_PI * i / (double)(n - 1);
        double w;
        switch (kind) {
        case 0: w = pow(0.5 - 0.5 * cos(a), 0.85); break;
        case 1: w = 0.54 - 0.46 * cos(a); break;
        case 2: w = 0.5 - 0.5 * cos(a); break;
        default: w = 1.0; break;
        }
        out[i] = (float)w;
    }
}

static double mel_of(double hz) { return 1127.0 * log(1.0 + hz / 700.0); }

/* SURVEY.md App.A step 4. out is [n_mels][n_fft/2+1] row-major; last bin (Nyquist) is 0.
 * high_hz <= 0 means nyquist + high_hz (lhotse default -400). */
void orc_mel_banks(int n_mels, int n_fft, float sample_rate, float low_hz, float high_hz, float *out) {
    int nb = n_fft / 2 + 1;
    double hi = high_hz;
    if (hi <= 0.0) hi = sample_rate / 2.0 + hi;
    double ml = mel_of(low_hz), mh = mel_of(hi);
    double d = (mh - ml) / (n_mels + 1);
    memset(out, 0, sizeof(float) * (size_t)n_mels * nb);
    for (int m = 0; m < n_mels; ++m) {
        double l = ml + m * d, c = l + d, r = c + d;
        for (int k = 0; k < n_fft / 2; ++k) {
            double mk = mel_of((double)k * sample_rate / n_fft);
            if (mk > l && mk < r) {
                double w = mk <= c ? (mk - l) / (c - l) : (r - mk) / (r - c);
                out[(size_t)m * nb + k] = (float)w;
            }
        }
    }
}

typedef struct {
    int sample_rate, frame_len, frame_shift, n_fft, n_mels;
    float preemph, low_hz, high_hz, log_floor;
    int remove_dc, snip_edges;
} orc_fbank_cfg;

/* index into the reflect-padded signal (App.A step 1): padded index p covers
 * original sample p - n_left; out-of-range indices mirror INCLUDING the edge sample. */
static float padded_sample(const float *x, int64_t S, int64_t i) {
    if (i < 0) i = -i - 1;
    if (i >= S) i = 2 * S - 1 - i;
    if (i < 0) i = 0;      /* degenerate very short inputs */
    if (i >= S) i = S - 1;
    return x[i];
}

/* pcm [B][S] -> feats [B][T][n_mels]; window [frame_len]; mel [n_mels][n_fft/2+1] */
void orc_fbank(const float *pcm, int B, int64_t S, const orc_fbank_cfg *cfg,
               const float *window, const float *mel, float *feats) {
    const int L = cfg->frame_len, N = cfg->n_fft, nb = N / 2 + 1, F = cfg->n_mels;
    const int64_t T = orc_num_frames(S, L, cfg->frame_shift, cfg->snip_edges);
    const int n_left = cfg->snip_edges ? 0 : (L - cfg->frame_shift) / 2;
    double *ct = (double *)malloc(sizeof(double) * N), *st = (double *)malloc(sizeof(double) * N);
    float *f = (float *)malloc(sizeof(float) * L);
    float *y = (float *)malloc(sizeof(float) * L);
    double *pw = (double *)malloc(sizeof(double) * nb);
    for (int i = 0; i < N; ++i) { ct[i] = cos(2.0 * M_PI * i / N); st[i] = sin(2.0 * M_PI * i / N); }
    for (int b = 0; b < B; ++b) {
        const float *x = pcm + (size_t)b * S;
        for (int64_t t = 0; t < T; ++t) {
            int64_t start = t * cfg->frame_shift - n_left;
            double mu = 0.0;
            for (int n = 0; n < L; ++n) { f[n] = padded_sample(x, S, start + n); mu += f[n]; }
            /* step 2: DC removal, pre-emphasis (replicating sample 0), window */
            float muf = cfg->remove_dc ? (float)(mu / L) : 0.0f;
            for (int n = 0; n < L; ++n) f[n] -= muf;
            for (int n = 0; n < L; ++n) {
                float prev = f[n > 0 ? n - 1 : 0];
                y[n] = (f[n] - cfg->preemph * prev) * window[n];
            }
            /* step 3: power spectrum of the zero-padded frame (direct DFT, double) */
            for (int k = 0; k < nb; ++k) {
                double re = 0.0, im = 0.0;
                for (int n = 0; n < L; ++n) {
                    int idx = (int)(((int64_t)k * n) % N);
                    re += y[n] * ct[idx];
                    im -= y[n] * st[idx];
                }
                pw[k] = (double)(float)(re * re + im * im);
            }
            /* steps 4-5: mel projection, floor, log */
            float *o = feats + ((size_t)b * T + t) * F;
            for (int m = 0; m < F; ++m) {
                double acc = 0.0;
                const float *w = mel + (size_t)m * nb;
                for (int k = 0; k < nb; ++k) acc += pw[k] * w[k];
                float v = (float)acc;
                if (v < cfg->log_floor) v = cfg->log_floor;
                o[m] = logf(v);
            }
        }
    }
    free(ct); free(st); free(f); free(y); free(pw);
}

/* The same feature stage in float64 THROUGHOUT (f32 samples, f32 window / mel tables as uploaded to the GPU, every
 * operation and the output in double): the "truth" the fp32 feature stages -- the HIP kernel and the torch-CPU rfft
 * restatement alike -- are measured against (tests/test_gpu_scale.py, bench.py).  Same algorithm as orc_fbank above
 * (SURVEY.md Appendix A), same PARITY UNPINNED status.  pcm [B][S] -> feats [B][T][n_mels] double. */
void orc_fbank_f64(const float *pcm, int B, int64_t S, const orc_fbank_cfg *cfg,
                   const float *window, const float *mel, double *feats) {
    const int L = cfg->frame_len, N = cfg->n_fft, nb = N / 2 + 1, F = cfg->n_mels;
    const int64_t T = orc_num_frames(S, L, cfg->frame_shift, cfg->snip_edges);
    const int n_left = cfg->snip_edges ? 0 : (L - cfg->frame_shift) / 2;
    double *ct = (double *)malloc(sizeof(double) * N), *st = (double *)malloc(sizeof(double) * N);
    double *f = (double *)malloc(sizeof(double) * L), *y = (double *)malloc(sizeof(double) * L);
    double *pw = (double *)malloc(sizeof(double) * nb);
    for (int i = 0; i < N; ++i) { ct[i] = cos(2.0 * M_PI * i / N); st[i] = sin(2.0 * M_PI * i / N); }
    for (int b = 0; b < B; ++b) {
        const float *x = pcm + (size_t)b * S;
        for (int64_t t = 0; t < T; ++t) {
            int64_t start = t * cfg->frame_shift - n_left;
            double mu = 0.0;
            for (int n = 0; n < L; ++n) { f[n] = padded_sample(x, S, start + n); mu += f[n]; }
            mu = cfg->remove_dc ? mu / L : 0.0;
            for (int n = 0; n < L; ++n) f[n] -= mu;
            for (int n = 0; n < L; ++n) y[n] = (f[n] - (double)cfg->preemph * f[n > 0 ? n - 1 : 0]) * (double)window[n];
            for (int k = 0; k < nb; ++k) {
                double re = 0.0, im = 0.0;
                for (int n = 0; n < L; ++n) {
                    int idx = (int)(((int64_t)k * n) % N);
                    re += y[n] * ct[idx];
                    im -= y[n] * st[idx];
                }
                pw[k] = re * re + im * im;
            }
            double *o = feats + ((size_t)b * T + t) * F;
            for (int m = 0; m < F; ++m) {
                double acc = 0.0;
                const float *w = mel + (size_t)m * nb;
                for (int k = 0; k < nb; ++k) acc += pw[k] * (double)w[k];
                if (acc < (double)cfg->log_floor) acc = (double)cfg->log_floor;
                o[m] = log(acc);
            }
        }
    }
    free(ct); free(st); free(f); free(y); free(pw);
}

/* ---------------------------------------------------------------- classifier */

static float sigmoidf_(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

/* One LSTM layer, one direction.  torch layout: w_ih [4H][in], w_hh [4H][H],
 * b_ih/b_hh [4H], row blocks i,f,g,o.  x [B][T][in] -> y [B][T][ystride] written at
 * column offset yoff (so fwd/rev halves interleave as torch concatenates them).
 * h0/c0 (may be NULL => zeros) and hN/cN (may be NULL) are [B][H] for streaming tests. */
void orc_lstm_layer(const float *x, int B, int T, int in, int H, int reverse,
                    const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                    float *y, int ystride, int yoff,
                    const float *h0, const float *c0, float *hN, float *cN) {
    float *h = (float *)calloc((size_t)H, sizeof(float));
    float *c = (float *)calloc((size_t)H, sizeof(float));
    float *g = (float *)malloc(sizeof(float) * 4 * (size_t)H);
    for (int b = 0; b < B; ++b) {
        for (int u = 0; u < H; ++u) {
            h[u] = h0 ? h0[(size_t)b * H + u] : 0.0f;
            c[u] = c0 ? c0[(size_t)b * H + u] : 0.0f;
        }
        for (int s = 0; s < T; ++s) {
            int t = reverse ? T - 1 - s : s;
            const float *xt = x + ((size_t)b * T + t) * in;
            for (int r = 0; r < 4 * H; ++r) {
                double acc = (double)b_ih[r] + (double)b_hh[r];
                const float *wi = w_ih + (size_t)r * in;
                for (int k = 0; k < in; ++k) acc += (double)wi[k] * xt[k];
                const float *wh = w_hh + (size_t)r * H;
                for (int k = 0; k < H; ++k) acc += (double)wh[k] * h[k];
                g[r] = (float)acc;
            }
            float *yt = y + ((size_t)b * T + t) * ystride + yoff;
            for (int u = 0; u < H; ++u) {
                float ig = sigmoidf_(g[u]), fg = sigmoidf_(g[H + u]);
                float gg = (float)tanh((double)g[2 * H + u]), og = sigmoidf_(g[3 * H + u]);
                c[u] = fg * c[u] + ig * gg;
                h[u] = og * (float)tanh((double)c[u]);
            }
            for (int u = 0; u < H; ++u) yt[u] = h[u];
        }
        if (hN) memcpy(hN + (size_t)b * H, h, sizeof(float) * H);
        if (cN) memcpy(cN + (size_t)b * H, c, sizeof(float) * H);
    }
    free(h); free(c); free(g);
}

typedef struct {
    int in_dim, hidden, num_layers, bidirectional, lin_hidden, lin_layers;
    float leaky_slope;
} orc_model_cfg;

/* Flat weight blob, torch state_dict order for PyanNet2 (SURVEY.md 8b "Weight naming"):
 *   for layer k, for dir d in (fwd[,rev]): w_ih, w_hh, b_ih, b_hh
 *   for linear j: weight [out][in], bias [out]
 *   classifier.weight [1][in], classifier.bias [1]                                 */
size_t orc_weight_count(const orc_model_cfg *m) {
    int D = m->bidirectional ? 2 : 1, H = m->hidden;
    size_t n = 0;
    for (int k = 0; k < m->num_layers; ++k) {
        int in = k == 0 ? m->in_dim : H * D;
        n += (size_t)D * ((size_t)4 * H * in + (size_t)4 * H * H + 8 * (size_t)H);
    }
    int prev = H * D;
    for (int j = 0; j < m->lin_layers; ++j) { n += (size_t)m->lin_hidden * prev + m->lin_hidden; prev = m->lin_hidden; }
    n += (size_t)prev + 1;
    return n;
}

/* feats [B][T][F] -> optional taps lstm_out [B][T][H*D], lin_out [B][T][lin_hidden],
 * logits [B][T], probs [B][T] (any may be NULL). */
void orc_classify(const orc_model_cfg *m, const float *weights, const float *feats, int B, int T,
                  float *lstm_out, float *lin_out, float *logits, float *probs) {
    int D = m->bidirectional ? 2 : 1, H = m->hidden, W = H * D;
    size_t rows = (size_t)B * T;
    /* both ping-pong buffers hold the widest activation of the network (features, LSTM output, feed-forward) */
    size_t wide = (size_t)(m->in_dim > W ? m->in_dim : W);
    if (m->lin_layers > 0 && (size_t)m->lin_hidden > wide) wide = (size_t)m->lin_hidden;
    float *cur = (float *)malloc(sizeof(float) * rows * wide);
    float *nxt = (float *)malloc(sizeof(float) * rows * wide);
    memcpy(cur, feats, sizeof(float) * rows * m->in_dim);
    const float *p = weights;
    int in = m->in_dim;
    for (int k = 0; k < m->num_layers; ++k) {
        for (int d = 0; d < D; ++d) {
            const float *w_ih = p; p += (size_t)4 * H * in;
            const float *w_hh = p; p += (size_t)4 * H * H;
            const float *b_ih = p; p += 4 * H;
            const float *b_hh = p; p += 4 * H;
            orc_lstm_layer(cur, B, T, in, H, d, w_ih, w_hh, b_ih, b_hh, nxt, W, d * H, 0, 0, 0, 0);
        }
        float *tmp = cur; cur = nxt; nxt = tmp;
        in = W;
    }
    if (lstm_out) memcpy(lstm_out, cur, sizeof(float) * rows * W);
    int prev = W;
    for (int j = 0; j < m->lin_layers; ++j) {
        int out = m->lin_hidden;
        const float *w = p; p += (size_t)out * prev;
        const float *bias = p; p += out;
        for (size_t r = 0; r < rows; ++r) {
            const float *xr = cur + r * prev;
            float *yr = nxt + r * out;
            for (int o = 0; o < out; ++o) {
                double acc = bias[o];
                const float *wr = w + (size_t)o * prev;
                for (int k = 0; k < prev; ++k) acc += (double)wr[k] * xr[k];
                float v = (float)acc;
                yr[o] = v >= 0.0f ? v : m->leaky_slope * v;
            }
        }
        float *tmp = cur; cur = nxt; nxt = tmp;
        prev = out;
    }
    if (lin_out) memcpy(lin_out, cur, sizeof(float) * rows * prev);
    const float *wc = p; p += prev;
    float bc = *p;
    for (size_t r = 0; r < rows; ++r) {
        double acc = bc;
        const float *xr = cur + r * prev;
        for (int k = 0; k < prev; ++k) acc += (double)wc[k] * xr[k];
        float lg = (float)acc;
        if (logits) logits[r] = lg;
        if (probs) probs[r] = sigmoidf_(lg);
    }
    free(cur); free(nxt);
}

/* ---------------------------------------------------------------- classifier, float64 throughout
 * Same network (PyanNet2.py:154-187) with EVERY intermediate in double: gate pre-activations, (h, c), layer outputs,
 * feed-forward activations.  Weights and features are the f32 values.  This is the "truth" the fp32 implementations
 * (torch CPU, the HIP path, orc_classify above, whose state is f32) are measured against on the near-chaotic x4 test
 * network, where the distance between two fp32 implementations says little. */
static void lstm_layer_f64(const double *x, int T, int in, int H, int reverse, const float *w_ih, const float *w_hh,
                           const float *b_ih, const float *b_hh, double *y, int ystride, int yoff) {
    double *h = (double *)calloc((size_t)H, sizeof(double)), *c = (double *)calloc((size_t)H, sizeof(double));
    double *g = (double *)malloc(sizeof(double) * 4 * (size_t)H);
    for (int s = 0; s < T; ++s) {
        int t = reverse ? T - 1 - s : s;
        const double *xt = x + (size_t)t * in;
        for (int r = 0; r < 4 * H; ++r) {
            double acc = (double)b_ih[r] + (double)b_hh[r];
            const float *wi = w_ih + (size_t)r * in, *wh = w_hh + (size_t)r * H;
            for (int k = 0; k < in; ++k) acc += (double)wi[k] * xt[k];
            for (int k = 0; k < H; ++k) acc += (double)wh[k] * h[k];
            g[r] = acc;
        }
        double *yt = y + (size_t)t * ystride + yoff;
        for (int u = 0; u < H; ++u) {
            double ig = 1.0 / (1.0 + exp(-g[u])), fg = 1.0 / (1.0 + exp(-g[H + u]));
            double gg = tanh(g[2 * H + u]), og = 1.0 / (1.0 + exp(-g[3 * H + u]));
            c[u] = fg * c[u] + ig * gg;
            h[u] = og * tanh(c[u]);
            yt[u] = h[u];
        }
    }
    free(h); free(c); free(g);
}

/* feats [B][T][F] f32 -> logits [B][T] double (same weight blob as orc_classify) */
void orc_classify_f64(const orc_model_cfg *m, const float *weights, const float *feats, int B, int T, double *logits) {
    int D = m->bidirectional ? 2 : 1, H = m->hidden, W = H * D;
    size_t wide = (size_t)(m->in_dim > W ? m->in_dim : W);
    if (m->lin_layers > 0 && (size_t)m->lin_hidden > wide) wide = (size_t)m->lin_hidden;
    double *cur = (double *)malloc(sizeof(double) * (size_t)T * wide), *nxt = (double *)malloc(sizeof(double) * (size_t)T * wide);
    for (int b = 0; b < B; ++b) {
        for (size_t i = 0; i < (size_t)T * m->in_dim; ++i) cur[i] = feats[(size_t)b * T * m->in_dim + i];
        const float *p = weights;
        int in = m->in_dim;
        for (int k = 0; k < m->num_layers; ++k) {
            for (int d = 0; d < D; ++d) {
                const float *w_ih = p; p += (size_t)4 * H * in;
                const float *w_hh = p; p += (size_t)4 * H * H;
                const float *b_ih = p; p += 4 * H;
                const float *b_hh = p; p += 4 * H;
                lstm_layer_f64(cur, T, in, H, d, w_ih, w_hh, b_ih, b_hh, nxt, W, d * H);
            }
            double *tmp = cur; cur = nxt; nxt = tmp;
            in = W;
        }
        int prev = W;
        for (int j = 0; j < m->lin_layers; ++j) {
            int out = m->lin_hidden;
            const float *w = p; p += (size_t)out * prev;
            const float *bias = p; p += out;
            for (int t = 0; t < T; ++t)
                for (int o = 0; o < out; ++o) {
                    double acc = bias[o];
                    for (int k = 0; k < prev; ++k) acc += (double)w[(size_t)o * prev + k] * cur[(size_t)t * prev + k];
                    nxt[(size_t)t * out + o] = acc >= 0.0 ? acc : (double)m->leaky_slope * acc;
                }
            double *tmp = cur; cur = nxt; nxt = tmp;
            prev = out;
        }
        const float *wc = p; p += prev;
        for (int t = 0; t < T; ++t) {
            double acc = *p;
            for (int k = 0; k < prev; ++k) acc += (double)wc[k] * cur[(size_t)t * prev + k];
            logits[(size_t)b * T + t] = acc;
        }
    }
    free(cur); free(nxt);
}

/* ------------------------------------------------------------ post-processing */

/* helper.py:66-97: x>=0.5 -> 1 else 0, then odd-length median with zero-padded edges
 * (scipy.signal.medfilt semantics).  probs [B][T] -> labels [B][T] (0/1 as uint8). */
void orc_median_filter(const float *probs, int B, int T, int kernel, uint8_t *labels) {
    int half = kernel / 2;
    for (int b = 0; b < B; ++b) {
        const float *p = probs + (size_t)b * T;
        for (int t = 0; t < T; ++t) {
            int ones = 0;
            for (int j = t - half; j <= t + half; ++j)
                if (j >= 0 && j < T && !(p[j] < 0.5f)) ++ones;
            labels[(size_t)b * T + t] = (uint8_t)(ones > half);
        }
    }
}

/* predict.py:472-490: walk frames; a 0->1 edge at frame k opens an interval at k*shift,
 * a 1->0 edge at frame k closes it at (k-1)*shift; values rounded to 2 decimals; an
 * interval is kept only if end - start > 0.  A run still open at the end is closed at
 * (T-1)*shift.  Returns the number of intervals written (<= max_out). */
int orc_intervals(const uint8_t *labels, int T, double shift, double *starts, double *ends, int max_out) {
    int n = 0, open = 0;
    double s = 0.0;
    for (int k = 0; k < T; ++k) {
        if (labels[k] && !open) { open = 1; s = round(k * shift * 100.0) / 100.0; }
        else if (!labels[k] && open) {
            open = 0;
            double e = round((k - 1) * shift * 100.0) / 100.0;
            if (e - s > 0 && n < max_out) { starts[n] = s; ends[n] = e; ++n; }
        }
    }
    if (open) {
        double e = round((T - 1) * shift * 100.0) / 100.0;
        if (e - s > 0 && n < max_out) { starts[n] = s; ends[n] = e; ++n; }
    }
    return n;
}

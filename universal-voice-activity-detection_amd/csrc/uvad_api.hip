// uvad_api.hip -- C ABI of libuvad.so (see include/uvad.h): context, weight repacking into
// kernel layouts, workspace carving and the launch sequence of the hot path.  Host code only;
// every compute call enqueues kernels on the caller's stream and returns (no allocation, no
// synchronisation), so the whole sequence can be captured into a hipGraph by the caller.
#include "../../include/uvad.h"
#include "uvad_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <tuple>
#include <string>
#include <vector>

using namespace uvad;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct LayerDev {
    float *w_ih = nullptr;   // [dirs*4H (permuted: dir, unit, gate)][in]
    unsigned short *w_ih_split16 = nullptr; // the same, scaled by a power of two, as three exact f16 planes (gemm_f16p.hip)
    float w_ih_scale = 1.0f;
    float *bias = nullptr;   // [dirs*4H] b_ih + b_hh, same permutation
    float *w_hh = nullptr;   // [dirs][packed register image]
    float *w_ih_img = nullptr;   // causal H = 128 models: W_ih as a register image (lstm_stack.hip), else nullptr
    unsigned *w_hh16_regs = nullptr;        // 16-sequence kernel (H = 128): [dirs][P0 / P1 register image]
    unsigned short *w_hh16_p2 = nullptr;    // [dirs][P2 LDS image]
    unsigned short *w_hh16_p2q = nullptr;   // [dirs][the same as bf8 bytes] (nullptr: not every element is exactly representable)
    int w_hh16_p2q_scale = 127;
    float *w_hh16_scale = nullptr;          // [dirs] 2^-S
    bool w_hh16_ok = false;
    int in = 0;
};

// Everything uvad_finalize produces from the host tensors: the device buffers (owned here, freed with the last owner) and the values derived
// while packing.  Contexts that are finalized with IDENTICAL host tensors, model / SincNet configuration and device share one of these through a
// process-wide cache (find_or_insert_packed): the twelve slots of a ForwardPipeline, or the three of predict_vad, repack and upload the 6 MB of
// weights once instead of once per context (host-side packing is ~45 ms per context: it was 0.18 of the 0.26 s predict_vad spends on a one-hour
// recording).  Read-only on the device, so sharing needs no synchronisation; a weight hot-swap gives the swapping context a block of its own.
struct PackedWeights {
    int device = 0;
    std::vector<void *> allocs;
    std::vector<LayerDev> layers;
    bool f16_ok = true;
    std::vector<float *> lin_w, lin_b, lin_w_img;
    std::vector<unsigned short *> lin_w_split16;
    std::vector<float> lin_w_scale;
    float *cls_w = nullptr, *cls_b = nullptr;
    bool sinc_ready = false, sinc_f16 = false;
    float *sn_wav_g = nullptr, *sn_wav_b = nullptr;
    float *sn_wt[3] = {nullptr, nullptr, nullptr}, *sn_bias[3] = {nullptr, nullptr, nullptr}, *sn_g[3] = {nullptr, nullptr, nullptr}, *sn_b[3] = {nullptr, nullptr, nullptr};
    unsigned short *sn_wfrag[3] = {nullptr, nullptr, nullptr};
    float sn_wscale[3] = {1.f, 1.f, 1.f}, *sn_bias16[3] = {nullptr, nullptr, nullptr};
    float sn_in_gmax[3] = {0.f, 0.f, 0.f}, sn_in_bmax[3] = {0.f, 0.f, 0.f};
    ~PackedWeights() {
        if (allocs.empty()) return;
        int cur = -1;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        for (void *p : allocs) (void)hipFree(p);
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};

// Row tiles of the tile-major activation matrices (row = (tile * T + t) * 4 + j, 128-row tiles) by the time chunk that needs them first:
// direction 0 walks t upwards, direction 1 downwards.  list[off[d][i] .. + len[d][i]) = the row tiles of chunk i of direction d.
struct ChunkPlan {
    int chunks = 1;
    std::vector<int> bound;   // chunk i = frames [bound[i], bound[i + 1]) of the forward pass, [T - bound[i + 1], T - bound[i]) of the backward pass
    int *d_list = nullptr;
    std::vector<int> off[2], len[2];
    unsigned long long last_use = 0;   // stamp of the context's use counter: the least recently used plan is evicted (MAX_CHUNK_PLANS)
    bool pinned = false;               // handed to a stream capture: a graph may replay launches that read d_list, so it is never evicted
};
constexpr size_t MAX_CHUNK_PLANS = 16;      // distinct (batch, T) shapes whose row-tile lists stay on the device
constexpr int SIDE_RETRY_AFTER = 256;       // calls after which a caller stream that found no concurrent side stream is probed again

struct StreamCounters { int64_t n_samples = 0, n_frames = 0, n_steps = 0; };
struct StreamState { float *h = nullptr, *c = nullptr; size_t layer_stride = 0; };

struct uvad_ctx {
    std::map<void *, StreamCounters> streams;   // host mirror of the lock-step stream groups, keyed by d_state
    int device = 0, n_cu = 256;
    bool has_fb = false, has_model = false, finalized = false, tables_set = false;
    uvad_fbank_cfg fb{};
    uvad_model_cfg mc{};
    std::string err;
    std::map<std::string, HostTensor> host_w;
    // device
    float *d_window = nullptr, *d_mel_w = nullptr, *d_mel_wt = nullptr, *d_tw512 = nullptr;
    int *d_mel_start = nullptr, *d_mel_len = nullptr;
    int mel_stride = 0, mel_nyquist = 1;
    std::vector<LayerDev> layers;
    std::vector<float *> lin_w, lin_b;
    std::vector<unsigned short *> lin_w_split16;
    std::vector<float *> lin_w_img;   // 128 x 128 layers as register images (lstm_stack.hip), else nullptr
    std::vector<float> lin_w_scale;
    bool f16_ok = true;   // every GEMM operand the weights determine fits the f16 range (gemm mode 1 is usable)
    int gemm_mode = 1;    // 0: exact f32 MFMA (gemm.hip); 1: split-f16 x3 (gemm_f16p.hip)
    int rec_tile_mode = 0, rec_tile_used = 0;   // sequences per recurrent workgroup: requested (0 = by batch size) / last launched
    float *cls_w = nullptr, *cls_b = nullptr;
    // SincNet front end (sincnet.hip)
    bool has_sinc = false, sinc_ready = false;
    uvad_sincnet_cfg sc{};
    float *sn_wav_g = nullptr, *sn_wav_b = nullptr;
    float *sn_wt[3] = {nullptr, nullptr, nullptr}, *sn_bias[3] = {nullptr, nullptr, nullptr};
    float *sn_g[3] = {nullptr, nullptr, nullptr}, *sn_b[3] = {nullptr, nullptr, nullptr};
    // ... and for the split-f16 form of the stages (sincnet_f16p.hip; GEMM modes 1 / 3): B-operand register images, 2^-S, padded biases,
    // and the largest |gamma| / |beta| of the norm in FRONT of each stage (the f16 range guard of sincnet_impl)
    bool sinc_f16 = false, sinc_f16_used = false;   // packed and usable / what the most recent uvad_sincnet ran
    unsigned short *sn_wfrag[3] = {nullptr, nullptr, nullptr};
    float sn_wscale[3] = {1.f, 1.f, 1.f}, *sn_bias16[3] = {nullptr, nullptr, nullptr};
    float sn_in_gmax[3] = {0.f, 0.f, 0.f}, sn_in_bmax[3] = {0.f, 0.f, 0.f};
    std::vector<void *> allocs;          // feature tables, twiddles: live as long as the context
    std::vector<void *> weight_allocs;   // what the uvad_finalize in progress has uploaded so far (moved into `packed` when it succeeds)
    std::shared_ptr<PackedWeights> packed;   // the finalized weights this context uses (possibly shared with other contexts: PackedWeights)
    // timing
    bool timing = false;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    std::vector<hipEvent_t> layer_ev;
    // time-chunked layers (uvad_set_time_chunks): the projection of chunk i + 1 on a side stream beside the recurrence of chunk i
    int chunk_mode = 0;                    // 0 = automatic, 1 = off, n > 1 = n chunks wherever the chunked form can run
    int chunks_used = 0;                   // chunks of the most recent classify / forward (1 = not chunked)
    hipStream_t side = nullptr;            // the library's own stream for the chunk projections
    hipStream_t side_for = nullptr;        // the caller stream `side` was PROVEN concurrent with (nullptr: not yet probed / not concurrent)
    bool side_probed_for_null = false;     // (a null caller stream is a valid key: remember that it was probed)
    std::map<hipStream_t, int> side_failed;   // caller streams no side stream was found concurrent with -> calls left until the next probe
                                              // (a probe that ran while other contexts kept the GPU busy can read "serialised" falsely)
    unsigned long long plan_clock = 0;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_chunk;
    std::map<std::tuple<int, int, int, int>, ChunkPlan> chunk_plans;   // (tiles, T, dirs, chunks) -> row-tile lists on the device
};

namespace {

int fail(uvad_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(uvad_ctx *c, hipError_t e, const char *what) {
    return fail(c, UVAD_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, call)                                        \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hip_fail((c), e_, #call); \
    } while (0)

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

template <typename T>
int dev_upload(uvad_ctx *c, const T *host, size_t n, T **out, bool weight = false) {
    void *p = nullptr;
    HIPCHK(c, hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : sizeof(T)));
    (weight ? c->weight_allocs : c->allocs).push_back(p);
    if (n) HIPCHK(c, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<T *>(p);
    return UVAD_OK;
}

// Workspace carving for B sequences of T frames (all offsets in bytes, 256-B aligned).
struct WsLayout {
    int tiles = 0, D = 0, Wd = 0, Fp = 0, Zw = 0;
    size_t M = 0;
    size_t off_G = 0, off_Y[2] = {0, 0}, off_Z[2] = {0, 0}, off_feats = 0, off_fplanes = 0, off_flag = 0, off_ctr = 0, off_hc = 0, total = 0;
};
// Activation buffers hold EITHER f32 rows OR two f16 planes of the same row width (hi plane, then the lo plane): same bytes.
WsLayout carve(const uvad_ctx *c, int B, int64_t T) {
    WsLayout w;
    const uvad_model_cfg &m = c->mc;
    w.tiles = (B + SEQ_TILE - 1) / SEQ_TILE;
    w.D = m.bidirectional ? 2 : 1;
    w.Wd = m.hidden * w.D;
    w.M = (size_t)w.tiles * SEQ_TILE * (size_t)T;
    w.Fp = gemm_f16p_padded_k(m.in_dim);
    w.Zw = m.lin_layers > 0 ? gemm_f16p_padded_k(m.lin_hidden) : 0;
    size_t o = 0;
    w.off_G = o; o += align_up(plane_rows(w.M) * 4 * m.hidden * w.D * sizeof(float));   // tile-blocked: whole 128-row tiles
    const size_t Mp = plane_rows(w.M);   // K-blocked planes come in whole 128-row tiles
    for (int i = 0; i < 2; ++i) { w.off_Y[i] = o; o += align_up(Mp * w.Wd * sizeof(float)); }
    for (int i = 0; i < 2; ++i) { w.off_Z[i] = o; o += align_up(Mp * (size_t)w.Zw * sizeof(float)); }
    w.off_feats = o; o += align_up((size_t)B * T * (size_t)(c->has_fb && c->fb.n_mels > m.in_dim ? c->fb.n_mels : m.in_dim) * sizeof(float));
    w.off_fplanes = o; o += align_up(Mp * (size_t)w.Fp * sizeof(float));   // f16 planes of the features (split-f16 GEMM mode)
    w.off_flag = o; o += align_up(sizeof(int));   // device-side "features outside the f16 range" flag (uvad_classify)
    w.off_ctr = o; o += align_up(gemm_f16p_ws_counter_bytes());   // tile-queue counters of the weight-stationary projection kernel
    w.off_hc = o; o += 2 * align_up((size_t)w.D * w.tiles * SEQ_TILE * m.hidden * sizeof(float));   // (h, c) carried between the time chunks of a layer
    w.total = o;
    return w;
}

// SincNet stage geometry and workspace for B waveforms of S samples.
struct SincLayout {
    int Cin[3], Cout[3], Kw[3], stride[3], NW[3], pt[3], phases[3];
    int64_t Lin[3], Lconv[3], Lpool[3];
    int ntiles[3];
    bool f16 = false; int cst[3] = {0, 0, 0}, ntiles16[3] = {0, 0, 0};   // split-f16 form (sincnet_f16p.hip): floats per output row, tiles of 64 pooled outputs
    size_t off_s0 = 0, off_P[3] = {0, 0, 0}, off_part[3] = {0, 0, 0}, off_sc[3] = {0, 0, 0}, total = 0;
    bool ok = false;
};
SincLayout sinc_carve(const uvad_ctx *c, int B, int64_t S) {
    SincLayout l;
    const uvad_sincnet_cfg &q = c->sc;
    const int cin[3] = {1, q.n_filters, q.c2}, cout[3] = {q.n_filters, q.c2, q.c3}, kw[3] = {q.kernel_size, q.k2, q.k3};
    int64_t L = S;
    l.ok = true;
    for (int i = 0; i < 3; ++i) {
        l.Cin[i] = cin[i]; l.Cout[i] = cout[i]; l.Kw[i] = kw[i]; l.stride[i] = i == 0 ? q.stride : 1;
        l.NW[i] = (cout[i] + 31) / 32 * 32;
        l.Lin[i] = L;
        l.Lconv[i] = L >= kw[i] ? (L - kw[i]) / l.stride[i] + 1 : 0;
        l.Lpool[i] = l.Lconv[i] / 3;
        {   // workgroup shape of the stage (sincnet.hip): pooled outputs per tile and statistics groups per tile
            SincConvArgs a{};
            a.Cin = cin[i]; a.Cout = cout[i]; a.Kw = kw[i]; a.stride = l.stride[i]; a.Ktot = cin[i] * kw[i]; a.Kp = (a.Ktot + 7) / 8 * 8;
            const SincConvPlan plan = sinc_conv_plan(a);
            l.pt[i] = plan.pt; l.phases[i] = plan.phases;
        }
        l.ntiles[i] = (int)((l.Lpool[i] + l.pt[i] - 1) / l.pt[i]);
        if (l.Lpool[i] <= 0) l.ok = false;
        L = l.Lpool[i];
    }
    l.f16 = sinc_f16p_supported(q.n_filters, q.kernel_size, q.stride, q.c2, q.k2, q.c3, q.k3);
    size_t o = 0;
    l.off_s0 = o; o += align_up((size_t)2 * B * sizeof(float));
    for (int i = 0; i < 3; ++i) {
        // the buffers are sized for whichever form a call runs (the GEMM mode can change between calls)
        size_t pooled = (size_t)B * l.Cout[i] * (size_t)(l.ok ? l.Lpool[i] : 0);
        size_t part = (size_t)B * (size_t)(l.ok ? l.ntiles[i] : 0) * l.phases[i] * l.NW[i] * 2;
        if (l.f16) {
            l.cst[i] = sinc_f16p_cst(i);
            l.ntiles16[i] = l.ok ? sinc_f16p_ntiles(l.Lpool[i]) : 0;
            pooled = std::max(pooled, (size_t)B * l.cst[i] * (size_t)(l.ok ? l.Lpool[i] : 0));
            part = std::max(part, sinc_f16p_partial_floats(i, B, l.ntiles16[i]));
        }
        l.off_P[i] = o; o += align_up(pooled * sizeof(float));
        l.off_part[i] = o; o += align_up(part * sizeof(float));
        l.off_sc[i] = o; o += align_up((size_t)2 * B * l.Cout[i] * sizeof(float));
    }
    l.total = o;
    return l;
}

std::string strip_prefix(const char *key) {
    std::string k(key);
    if (k.rfind("model.", 0) == 0) k = k.substr(6);
    // ModuleList-of-LSTMs variant (monolithic=False, PyanNet2.py:103-118): lstm.{k}.weight_ih_l0[_reverse]
    if (k.rfind("lstm.", 0) == 0 && k.size() > 5 && isdigit((unsigned char)k[5])) {
        size_t dot = k.find('.', 5);
        if (dot != std::string::npos) {
            const std::string layer = k.substr(5, dot - 5);
            std::string rest = k.substr(dot + 1);
            const size_t l0 = rest.find("_l0");
            if (l0 != std::string::npos) {
                rest.replace(l0, 3, "_l" + layer);
                k = "lstm." + rest;
            }
        }
    }
    return k;
}

bool expect_shape(const HostTensor &t, std::initializer_list<int64_t> want) {
    if (t.shape.size() != want.size()) return false;
    size_t i = 0;
    for (int64_t w : want)
        if (t.shape[i++] != w) return false;
    return true;
}

}  // namespace

extern "C" {

int uvad_abi_version(void) { return UVAD_ABI_VERSION; }

int uvad_create(int device, const uvad_fbank_cfg *fb, const uvad_model_cfg *model, uvad_ctx **out) {
    if (!out) return UVAD_E_ARG;
    *out = nullptr;
    uvad_ctx *c = new uvad_ctx();
    *out = c;  // returned even on failure so uvad_last_error() can be read; caller destroys it
    c->device = device;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(c, UVAD_E_HIP, std::string("no HIP device available (libuvad has no CPU fallback): ") +
                                       (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= ndev) return fail(c, UVAD_E_ARG, "device index out of range");
    HIPCHK(c, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, UVAD_E_HIP, std::string("libuvad is built for gfx950 only; device is ") + prop.gcnArchName);
    if (fb) {
        c->fb = *fb;
        c->has_fb = true;
        if (fb->n_fft != 512) return fail(c, UVAD_E_UNSUPPORTED, "only n_fft = 512 is implemented");
        if (fb->frame_len < 2 || fb->frame_len > 512 || fb->frame_shift < 1 || fb->n_mels < 1 || fb->n_mels > 128)
            return fail(c, UVAD_E_ARG, "bad fbank configuration");
        // forward-FFT twiddles (cos, -sin)(2 pi j / 512), computed in double
        std::vector<float> tw(2 * 512);
        for (int j = 0; j < 512; ++j) {
            const double a = 2.0 * M_PI * j / 512.0;
            tw[2 * j] = (float)std::cos(a);
            tw[2 * j + 1] = (float)(-std::sin(a));
        }
        int r = dev_upload(c, tw.data(), tw.size(), &c->d_tw512);
        if (r) return r;
    }
    if (model) {
        c->mc = *model;
        c->has_model = true;
        // hidden sizes: 128 and 64 have the register-resident recurrent kernels; any other size whose gate matrix comes in whole
        // 128-column tiles runs the generic recurrence (lstm_rec_any_kernel: correct, slow)
        if (model->hidden < 4 || model->hidden > 1024 || (4 * model->hidden * (model->bidirectional ? 2 : 1)) % 128 != 0)
            return fail(c, UVAD_E_UNSUPPORTED, "lstm hidden_size x directions must be a multiple of 32, hidden_size <= 1024");
        if (model->in_dim < 4 || model->in_dim % 4 != 0)
            return fail(c, UVAD_E_UNSUPPORTED, "encoding_dim must be a positive multiple of 4");
        if (model->num_layers < 1 || model->lin_layers < 0 || (model->lin_layers > 0 && (model->lin_hidden < 4 || model->lin_hidden % 4)))
            return fail(c, UVAD_E_ARG, "bad model configuration");
    }
    for (auto &ev : c->ev) HIPCHK(c, hipEventCreate(&ev));
    return UVAD_OK;
}

int uvad_set_tables(uvad_ctx *c, const float *window, const float *mel) {
    if (!c || !window || !mel) return UVAD_E_ARG;
    if (!c->has_fb) return fail(c, UVAD_E_STATE, "context was created without a fbank configuration");
    HIPCHK(c, hipSetDevice(c->device));
    const int nb = c->fb.n_fft / 2 + 1, F = c->fb.n_mels;
    std::vector<int> st(F), ln(F);
    int maxlen = 1;
    for (int m = 0; m < F; ++m) {
        int lo = nb, hi = -1;
        for (int k = 0; k < nb; ++k)
            if (mel[(size_t)m * nb + k] != 0.0f) { if (k < lo) lo = k; hi = k; }
        st[m] = hi < 0 ? 0 : lo;
        ln[m] = hi < 0 ? 0 : hi - lo + 1;
        if (ln[m] > maxlen) maxlen = ln[m];
    }
    // LDS bank spreading of the mel stage.  In fbank_pair() lane m walks its band's power values at scratch index st[m] + i with the SAME
    // i in every lane, so two lanes of a 32-lane group whose band starts are equal mod 32 hit one bank on every read (the 64-filter
    // table of the bench: up to 4 lanes per bank in the upper half of the filters -- a third of the kernel's LDS cycles were bank
    // conflicts).  Every lane runs the same trip count (the longest band, rounded up to four bins), so a shorter band has slack: its start
    // may be moved down by up to trip - len bins (zero weights in front) without costing an iteration.  A bipartite matching per
    // 32-lane group (filters -> bank residues, augmenting paths, smallest shift first) picks shifts that make the starts distinct
    // mod 32 wherever the slack allows; filters it cannot place keep their start.  Same sums up to the order of their terms.
    {
        const int trip = 4 * ((maxlen + 3) / 4);
        for (int g0 = 0; g0 < F; g0 += 32) {   // lanes g0 % 64 .. + 31 of filter pass g0 / 64: one LDS lane group
            const int n = std::min(32, F - g0);
            std::vector<int> owner(32, -1), shift_of(n, 0);   // bank residue -> filter of the group; chosen shift per filter
            // residue reached by filter j with shift sh; candidates in order of increasing shift
            auto max_shift = [&](int j) { const int m = g0 + j; return ln[m] > 0 ? std::min(trip - ln[m], st[m]) : 0; };
            std::function<bool(int, std::vector<char> &)> place = [&](int j, std::vector<char> &seen) -> bool {
                const int m = g0 + j;
                for (int sh = 0; sh <= max_shift(j); ++sh) {
                    const int res = ((st[m] - sh) % 32 + 32) % 32;
                    if (seen[res]) continue;
                    seen[res] = 1;
                    if (owner[res] < 0 || place(owner[res], seen)) {
                        owner[res] = j;
                        shift_of[j] = sh;
                        return true;
                    }
                }
                return false;
            };
            std::vector<int> order(n);
            for (int j = 0; j < n; ++j) order[j] = j;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return max_shift(a) < max_shift(b); });   // the constrained ones first
            for (int j : order) {
                std::vector<char> seen(32, 0);
                if (!place(j, seen)) shift_of[j] = 0;   // no free bank within its slack: stays where it is (conflicts with one other lane)
            }
            for (int j = 0; j < n; ++j) {
                const int m = g0 + j;
                st[m] -= shift_of[j];
                ln[m] += ln[m] > 0 ? shift_of[j] : 0;
            }
        }
        // (maxlen is unchanged: every shifted band still fits the trip count)
        for (int m = 0; m < F; ++m)
            if (ln[m] > trip) return fail(c, UVAD_E_STATE, "internal: mel band shift exceeded the trip count");
        maxlen = trip;
    }
    {   // the feature kernel keeps the whole weight image, a PCM tile and the transform scratch in LDS: refuse here, by name, what a launch could not take
        FbankArgs probe{};
        probe.frame_len = c->fb.frame_len; probe.frame_shift = c->fb.frame_shift; probe.n_mels = F;
        probe.tab.mel_stride = maxlen;
        if (fbank_lds_bytes(probe) > 160 * 1024)
            return fail(c, UVAD_E_UNSUPPORTED, "mel matrix: the longest band (" + std::to_string(maxlen) + " bins) makes the feature kernel's weight image (" +
                                               std::to_string(mel_image_floats(maxlen, F) * 4) + " bytes) exceed the 160 KiB of LDS with this frame geometry");
    }
    c->mel_stride = maxlen;
    c->mel_nyquist = 0;
    for (int m = 0; m < F; ++m)
        if (mel[(size_t)m * nb + nb - 1] != 0.0f) c->mel_nyquist = 1;
    std::vector<float> w((size_t)F * maxlen, 0.0f);
    for (int m = 0; m < F; ++m)
        for (int i = 0; i < ln[m]; ++i) w[(size_t)m * maxlen + i] = mel[(size_t)m * nb + st[m] + i];
    int r;
    if ((r = dev_upload(c, window, (size_t)c->fb.frame_len, &c->d_window))) return r;
    if ((r = dev_upload(c, st.data(), st.size(), &c->d_mel_start))) return r;
    if ((r = dev_upload(c, ln.data(), ln.size(), &c->d_mel_len))) return r;
    if ((r = dev_upload(c, w.data(), w.size(), &c->d_mel_w))) return r;
    const int ld_t = mel_image_ld(F);
    std::vector<float> wt(mel_image_floats(maxlen, F), 0.0f);
    for (int m = 0; m < F; ++m)
        for (int i = 0; i < ln[m]; ++i) wt[(size_t)i * ld_t + m] = 0.25f * w[(size_t)m * maxlen + i];   // the spectrum split leaves 4 |X|^2 (fbank_pair.h): exact
    if ((r = dev_upload(c, wt.data(), wt.size(), &c->d_mel_wt))) return r;
    c->tables_set = true;
    return UVAD_OK;
}

int uvad_set_weight(uvad_ctx *c, const char *torch_key, const float *host, const int64_t *shape, int ndim) {
    if (!c || !torch_key || !host || !shape || ndim < 1 || ndim > 3) return UVAD_E_ARG;
    if (!c->has_model) return fail(c, UVAD_E_STATE, "context was created without a model configuration");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) return fail(c, UVAD_E_ARG, "non-positive dimension");
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign(host, host + n);
    c->host_w[strip_prefix(torch_key)] = std::move(t);
    c->finalized = false;
    return UVAD_OK;
}

extern "C++" {
namespace {
void free_weights(uvad_ctx *c) {
    for (void *p : c->weight_allocs) (void)hipFree(p);   // (leftovers of a uvad_finalize that failed half-way)
    c->weight_allocs.clear();
    c->packed.reset();                                   // the shared block goes with its last owner
    for (auto &ev : c->layer_ev)
        if (ev) (void)hipEventDestroy(ev);
    c->layer_ev.clear();
    c->layers.clear();
    c->lin_w.clear(); c->lin_b.clear(); c->lin_w_split16.clear(); c->lin_w_scale.clear(); c->lin_w_img.clear();
    c->cls_w = c->cls_b = nullptr;
    c->sn_wav_g = c->sn_wav_b = nullptr;
    for (int i = 0; i < 3; ++i) {
        c->sn_wt[i] = c->sn_bias[i] = c->sn_g[i] = c->sn_b[i] = c->sn_bias16[i] = nullptr;
        c->sn_wfrag[i] = nullptr;
    }
    c->sinc_ready = c->sinc_f16 = false;
    c->finalized = false;
}
}  // namespace
}  // extern "C++"

extern "C++" {
namespace {
// 128-bit digest of everything the packed weights depend on: device, model / SincNet configuration, every host tensor (name, shape, bytes)
struct WeightKey {
    unsigned long long h[2];
    bool operator<(const WeightKey &o) const { return h[0] != o.h[0] ? h[0] < o.h[0] : h[1] < o.h[1]; }
};
inline void mix(WeightKey &k, const void *p, size_t n) {
    const unsigned char *b = reinterpret_cast<const unsigned char *>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        unsigned long long w;
        memcpy(&w, b + i, 8);
        k.h[0] = (k.h[0] ^ w) * 0x9E3779B97F4A7C15ull; k.h[0] ^= k.h[0] >> 29;
        k.h[1] = (k.h[1] + w) * 0xC2B2AE3D27D4EB4Full; k.h[1] ^= k.h[1] >> 31;
    }
    for (; i < n; ++i) {
        k.h[0] = (k.h[0] ^ b[i]) * 0x100000001B3ull;
        k.h[1] = (k.h[1] + b[i]) * 0x9E3779B97F4A7C15ull; k.h[1] ^= k.h[1] >> 27;
    }
}
WeightKey weight_key(const uvad_ctx *c) {
    WeightKey k{{0xcbf29ce484222325ull, 0x84222325cbf29ce4ull}};
    mix(k, &c->device, sizeof c->device);
    mix(k, &c->mc, sizeof c->mc);
    const int hs = c->has_sinc ? 1 : 0;
    mix(k, &hs, sizeof hs);
    if (c->has_sinc) mix(k, &c->sc, sizeof c->sc);
    for (const auto &kv : c->host_w) {   // std::map: key order
        mix(k, kv.first.data(), kv.first.size());
        const size_t nd = kv.second.shape.size();
        mix(k, &nd, sizeof nd);
        mix(k, kv.second.shape.data(), nd * sizeof(int64_t));
        mix(k, kv.second.data.data(), kv.second.data.size() * sizeof(float));
    }
    return k;
}
std::mutex packed_mu;
std::map<WeightKey, std::weak_ptr<PackedWeights>> packed_cache;

// the context's view of a block: plain copies of the pointers and flags (the launch code reads them from the context as before)
void adopt_packed(uvad_ctx *c, const std::shared_ptr<PackedWeights> &pw) {
    c->packed = pw;
    c->layers = pw->layers; c->f16_ok = pw->f16_ok;
    c->lin_w = pw->lin_w; c->lin_b = pw->lin_b; c->lin_w_img = pw->lin_w_img; c->lin_w_split16 = pw->lin_w_split16; c->lin_w_scale = pw->lin_w_scale;
    c->cls_w = pw->cls_w; c->cls_b = pw->cls_b;
    c->sinc_ready = pw->sinc_ready; c->sinc_f16 = pw->sinc_f16;
    c->sn_wav_g = pw->sn_wav_g; c->sn_wav_b = pw->sn_wav_b;
    for (int i = 0; i < 3; ++i) {
        c->sn_wt[i] = pw->sn_wt[i]; c->sn_bias[i] = pw->sn_bias[i]; c->sn_g[i] = pw->sn_g[i]; c->sn_b[i] = pw->sn_b[i];
        c->sn_wfrag[i] = pw->sn_wfrag[i]; c->sn_wscale[i] = pw->sn_wscale[i]; c->sn_bias16[i] = pw->sn_bias16[i];
        c->sn_in_gmax[i] = pw->sn_in_gmax[i]; c->sn_in_bmax[i] = pw->sn_in_bmax[i];
    }
}
std::shared_ptr<PackedWeights> snapshot_packed(uvad_ctx *c) {
    auto pw = std::make_shared<PackedWeights>();
    pw->device = c->device;
    pw->allocs.swap(c->weight_allocs);
    pw->layers = c->layers; pw->f16_ok = c->f16_ok;
    pw->lin_w = c->lin_w; pw->lin_b = c->lin_b; pw->lin_w_img = c->lin_w_img; pw->lin_w_split16 = c->lin_w_split16; pw->lin_w_scale = c->lin_w_scale;
    pw->cls_w = c->cls_w; pw->cls_b = c->cls_b;
    pw->sinc_ready = c->sinc_ready; pw->sinc_f16 = c->sinc_f16;
    pw->sn_wav_g = c->sn_wav_g; pw->sn_wav_b = c->sn_wav_b;
    for (int i = 0; i < 3; ++i) {
        pw->sn_wt[i] = c->sn_wt[i]; pw->sn_bias[i] = c->sn_bias[i]; pw->sn_g[i] = c->sn_g[i]; pw->sn_b[i] = c->sn_b[i];
        pw->sn_wfrag[i] = c->sn_wfrag[i]; pw->sn_wscale[i] = c->sn_wscale[i]; pw->sn_bias16[i] = c->sn_bias16[i];
        pw->sn_in_gmax[i] = c->sn_in_gmax[i]; pw->sn_in_bmax[i] = c->sn_in_bmax[i];
    }
    return pw;
}
}  // namespace
}  // extern "C++"

int uvad_finalize(uvad_ctx *c) {
    if (!c) return UVAD_E_ARG;
    if (!c->has_model) return fail(c, UVAD_E_STATE, "no model configuration");
    HIPCHK(c, hipSetDevice(c->device));
    // Idempotent: a second call (e.g. after swapping weights with uvad_set_weight) replaces the previous upload.
    // Kernels of earlier calls may still be reading the old buffers.
    if (c->packed || !c->weight_allocs.empty()) HIPCHK(c, hipDeviceSynchronize());
    free_weights(c);
    const uvad_model_cfg &m = c->mc;
    const int H = m.hidden, D = m.bidirectional ? 2 : 1;
    // another context of this process already holds these very weights in kernel layouts on this device: share them
    const WeightKey wkey = weight_key(c);
    {
        std::lock_guard<std::mutex> lk(packed_mu);
        auto it = packed_cache.find(wkey);
        if (it != packed_cache.end()) {
            if (std::shared_ptr<PackedWeights> pw = it->second.lock()) {
                adopt_packed(c, pw);
                c->layer_ev.assign((size_t)2 * m.num_layers + 2, nullptr);
                for (auto &ev : c->layer_ev) HIPCHK(c, hipEventCreate(&ev));
                c->finalized = true;
                return UVAD_OK;
            }
            packed_cache.erase(it);
        }
    }
    auto get = [&](const std::string &k) -> const HostTensor * {
        auto it = c->host_w.find(k);
        return it == c->host_w.end() ? nullptr : &it->second;
    };
    c->layers.assign(m.num_layers, LayerDev());
    c->f16_ok = true;
    for (int k = 0; k < m.num_layers; ++k) {
        const int in = k == 0 ? m.in_dim : H * D;
        const int inp = gemm_padded_k(in);   // rows zero-padded to the GEMM's K-step
        std::vector<float> wp((size_t)D * 4 * H * inp, 0.0f), bp((size_t)D * 4 * H), hh((size_t)D * whh_packed_elems(H));
        std::vector<unsigned> hh16r(H == 128 ? (size_t)D * whh16h_regs_elems() : 0, 0u);
        std::vector<unsigned short> hh16p(H == 128 ? (size_t)D * whh16h_p2_elems() : 0, 0);
        std::vector<unsigned short> hh16q(H == 128 ? (size_t)D * whh16h_p2q_elems() : 0, 0);
        bool hh16q_ok = H == 128;
        int hh16q_scale = 127;
        std::vector<float> hh16s(D, 1.0f);
        bool hh16ok = H == 128;
        std::vector<float> ih_img;
        for (int d = 0; d < D; ++d) {
            const std::string suf = "_l" + std::to_string(k) + (d ? "_reverse" : "");
            const HostTensor *wih = get("lstm.weight_ih" + suf), *whh = get("lstm.weight_hh" + suf);
            const HostTensor *bih = get("lstm.bias_ih" + suf), *bhh = get("lstm.bias_hh" + suf);
            if (!wih || !whh || !bih || !bhh) return fail(c, UVAD_E_STATE, "missing LSTM tensor for suffix " + suf);
            if (!expect_shape(*wih, {4 * H, in}) || !expect_shape(*whh, {4 * H, H}) ||
                !expect_shape(*bih, {4 * H}) || !expect_shape(*bhh, {4 * H}))
                return fail(c, UVAD_E_ARG, "LSTM tensor shape mismatch for suffix " + suf);
            // torch rows are gate-major (i,f,g,o blocks of H); kernels want (unit, gate) interleaved
            for (int u = 0; u < H; ++u)
                for (int g = 0; g < 4; ++g) {
                    const size_t dst = (size_t)d * 4 * H + (size_t)u * 4 + g, src = (size_t)g * H + u;
                    std::memcpy(&wp[dst * inp], &wih->data[src * in], sizeof(float) * in);
                    bp[dst] = bih->data[src] + bhh->data[src];
                }
            pack_whh(whh->data.data(), H, &hh[(size_t)d * whh_packed_elems(H)]);
            if (D == 1 && H == 128 && in % 4 == 0) {   // the streaming step's one-launch stack (lstm_stack.hip)
                ih_img.resize(lstm_image_elems(in));
                pack_lstm_image(wih->data.data(), in, ih_img.data());
            }
            if (H == 128 && !pack_whh16h(whh->data.data(), &hh16r[(size_t)d * whh16h_regs_elems()], &hh16p[(size_t)d * whh16h_p2_elems()], &hh16s[d]))
                hh16ok = false;
            if (H == 128 && !pack_whh16h_p2q(whh->data.data(), &hh16q[(size_t)d * whh16h_p2q_elems()], &hh16q_scale)) hh16q_ok = false;
        }
        LayerDev &L = c->layers[k];
        L.in = in;
        int r;
        if ((r = dev_upload(c, wp.data(), wp.size(), &L.w_ih, true))) return r;
        {
            std::vector<unsigned short> sp(3 * weight_plane_elems(D * 4 * H, inp));
            if (!split_weights_f16x3(wp.data(), D * 4 * H, inp, sp.data(), &L.w_ih_scale)) c->f16_ok = false;
            if ((r = dev_upload(c, sp.data(), sp.size(), &L.w_ih_split16, true))) return r;
        }
        if ((r = dev_upload(c, bp.data(), bp.size(), &L.bias, true))) return r;
        if ((r = dev_upload(c, hh.data(), hh.size(), &L.w_hh, true))) return r;
        if (!ih_img.empty() && (r = dev_upload(c, ih_img.data(), ih_img.size(), &L.w_ih_img, true))) return r;
        if (H == 128) {
            if ((r = dev_upload(c, hh16r.data(), hh16r.size(), &L.w_hh16_regs, true))) return r;
            if ((r = dev_upload(c, hh16p.data(), hh16p.size(), &L.w_hh16_p2, true))) return r;
            L.w_hh16_p2q = nullptr;
            L.w_hh16_p2q_scale = hh16q_scale;
            if (hh16q_ok && (r = dev_upload(c, hh16q.data(), hh16q.size(), &L.w_hh16_p2q, true))) return r;
            if ((r = dev_upload(c, hh16s.data(), hh16s.size(), &L.w_hh16_scale, true))) return r;
        }
        L.w_hh16_ok = hh16ok;
    }
    c->lin_w.assign(m.lin_layers, nullptr);
    c->lin_b.assign(m.lin_layers, nullptr);
    c->lin_w_split16.assign(m.lin_layers, nullptr);
    c->lin_w_img.assign(m.lin_layers, nullptr);
    c->lin_w_scale.assign(m.lin_layers, 1.0f);
    int prev = H * D;
    // Static bound on what the feed-forward GEMMs can be fed: |h| < 1 out of the LSTM, so |z_j| <= sum_k |w_jk| * amax + |b_j|
    // (leaky_relu does not grow magnitudes for slopes in [-1, 1]).  If that can leave the f16 range the split-f16 GEMM is not used.
    double amax = 1.0;
    for (int j = 0; j < m.lin_layers; ++j) {
        const HostTensor *w = get("linear." + std::to_string(j) + ".weight"), *b = get("linear." + std::to_string(j) + ".bias");
        if (!w || !b) return fail(c, UVAD_E_STATE, "missing linear." + std::to_string(j));
        if (!expect_shape(*w, {m.lin_hidden, prev}) || !expect_shape(*b, {m.lin_hidden}))
            return fail(c, UVAD_E_ARG, "linear." + std::to_string(j) + " shape mismatch");
        int r;
        const int prevp = gemm_padded_k(prev);
        std::vector<float> wpad((size_t)m.lin_hidden * prevp, 0.0f);
        double zmax = 0.0;
        for (int o = 0; o < m.lin_hidden; ++o) {
            std::memcpy(&wpad[(size_t)o * prevp], &w->data[(size_t)o * prev], sizeof(float) * prev);
            double l1 = 0.0;
            for (int k = 0; k < prev; ++k) l1 += std::fabs((double)w->data[(size_t)o * prev + k]);
            const double z = l1 * amax + std::fabs((double)b->data[o]);
            if (!(z <= zmax)) zmax = z;   // NaN-propagating max
        }
        if ((r = dev_upload(c, wpad.data(), wpad.size(), &c->lin_w[j], true))) return r;
        {
            std::vector<unsigned short> sp(3 * weight_plane_elems(m.lin_hidden, prevp));
            if (!split_weights_f16x3(wpad.data(), m.lin_hidden, prevp, sp.data(), &c->lin_w_scale[j])) c->f16_ok = false;
            if ((r = dev_upload(c, sp.data(), sp.size(), &c->lin_w_split16[j], true))) return r;
        }
        if ((r = dev_upload(c, b->data.data(), b->data.size(), &c->lin_b[j], true))) return r;
        if (m.lin_hidden == 128 && prev == 128) {   // the streaming step's in-launch head (lstm_stack.hip)
            std::vector<float> img(fc_image_elems());
            pack_fc_image(w->data.data(), img.data());
            if ((r = dev_upload(c, img.data(), img.size(), &c->lin_w_img[j], true))) return r;
        }
        amax = zmax * std::fmax(1.0, std::fabs((double)m.leaky_slope));
        if (j + 1 < m.lin_layers && !(amax < 65504.0)) c->f16_ok = false;   // the next feed-forward GEMM would see it
        prev = m.lin_hidden;
    }
    const HostTensor *cw = get("classifier.weight"), *cb = get("classifier.bias");
    if (!cw || !cb) return fail(c, UVAD_E_STATE, "missing classifier tensors");
    if (!expect_shape(*cw, {1, prev}) || !expect_shape(*cb, {1})) return fail(c, UVAD_E_ARG, "classifier shape mismatch");
    int r;
    if ((r = dev_upload(c, cw->data.data(), cw->data.size(), &c->cls_w, true))) return r;
    if ((r = dev_upload(c, cb->data.data(), cb->data.size(), &c->cls_b, true))) return r;
    c->layer_ev.assign((size_t)2 * m.num_layers + 2, nullptr);
    for (auto &ev : c->layer_ev) HIPCHK(c, hipEventCreate(&ev));
    if (c->has_sinc && get("sincnet.conv1d.0.filters")) {   // the stage is optional: packed only when its tensors were given
        const uvad_sincnet_cfg &q = c->sc;
        const HostTensor *wg = get("sincnet.wav_norm1d.weight"), *wb = get("sincnet.wav_norm1d.bias");
        if (!wg || !wb || !expect_shape(*wg, {1}) || !expect_shape(*wb, {1})) return fail(c, UVAD_E_STATE, "missing / misshaped sincnet.wav_norm1d tensors");
        if ((r = dev_upload(c, wg->data.data(), 1, &c->sn_wav_g, true))) return r;
        if ((r = dev_upload(c, wb->data.data(), 1, &c->sn_wav_b, true))) return r;
        const int cin[3] = {1, q.n_filters, q.c2}, cout[3] = {q.n_filters, q.c2, q.c3}, kw[3] = {q.kernel_size, q.k2, q.k3};
        for (int i = 0; i < 3; ++i) {
            const std::string id = std::to_string(i);
            const HostTensor *w = get(i == 0 ? std::string("sincnet.conv1d.0.filters") : "sincnet.conv1d." + id + ".weight");
            const HostTensor *b = i == 0 ? nullptr : get("sincnet.conv1d." + id + ".bias");
            const HostTensor *g = get("sincnet.norm1d." + id + ".weight"), *be = get("sincnet.norm1d." + id + ".bias");
            if (!w || (i > 0 && !b) || !g || !be) return fail(c, UVAD_E_STATE, "missing sincnet tensors of stage " + id);
            const bool wshape = i == 0 ? (expect_shape(*w, {cout[0], kw[0]}) || expect_shape(*w, {cout[0], 1, kw[0]})) : expect_shape(*w, {cout[i], cin[i], kw[i]});
            if (!wshape || (b && !expect_shape(*b, {cout[i]})) || !expect_shape(*g, {cout[i]}) || !expect_shape(*be, {cout[i]}))
                return fail(c, UVAD_E_ARG, "sincnet stage " + id + " tensor shape mismatch");
            // W[n][ci][tap] -> W^T [k = tap*Cin + ci][NW], zero padded in n and k (sincnet.hip B-operand layout)
            const int Ktot = cin[i] * kw[i], Kp = (Ktot + 7) / 8 * 8, NW = (cout[i] + 31) / 32 * 32;
            std::vector<float> wt((size_t)Kp * NW, 0.0f), bias((size_t)NW, 0.0f);
            for (int n = 0; n < cout[i]; ++n) {
                for (int ci = 0; ci < cin[i]; ++ci)
                    for (int t = 0; t < kw[i]; ++t)   // kernel K order: tap-major, channel-minor
                        wt[(size_t)(t * cin[i] + ci) * NW + n] = w->data[((size_t)n * cin[i] + ci) * kw[i] + t];
                if (b) bias[n] = b->data[n];
            }
            if ((r = dev_upload(c, wt.data(), wt.size(), &c->sn_wt[i], true))) return r;
            if ((r = dev_upload(c, bias.data(), bias.size(), &c->sn_bias[i], true))) return r;
            if ((r = dev_upload(c, g->data.data(), g->data.size(), &c->sn_g[i], true))) return r;
            if ((r = dev_upload(c, be->data.data(), be->data.size(), &c->sn_b[i], true))) return r;
            // the norm in FRONT of stage i + 1 (stage 0's is the waveform norm): bounds for the f16 range guard
            if (i < 2) {
                float gm = 0.f, bm = 0.f;
                for (float v : g->data) gm = std::max(gm, std::fabs(v));
                for (float v : be->data) bm = std::max(bm, std::fabs(v));
                c->sn_in_gmax[i + 1] = gm; c->sn_in_bmax[i + 1] = bm;
            }
        }
        c->sn_in_gmax[0] = std::fabs(wg->data[0]); c->sn_in_bmax[0] = std::fabs(wb->data[0]);
        // the split-f16 form of the stages (sincnet_f16p.hip): W[n][k] in the stage's K order -> three exact f16 planes as register images
        c->sinc_f16 = sinc_f16p_supported(q.n_filters, q.kernel_size, q.stride, q.c2, q.k2, q.c3, q.k3);
        for (int i = 0; i < 3 && c->sinc_f16; ++i) {
            const std::string id = std::to_string(i);
            const HostTensor *w = get(i == 0 ? std::string("sincnet.conv1d.0.filters") : "sincnet.conv1d." + id + ".weight");
            const HostTensor *b = i == 0 ? nullptr : get("sincnet.conv1d." + id + ".bias");
            const int ldk = 32 * sinc_f16p_ksteps(i), cst = sinc_f16p_cst(i);
            const int cpad = i == 1 ? cin[1] : 64;            // k = tap * cpad + channel (stage 1: 80 channels per tap, stage 2: 64 with 60 real)
            std::vector<float> wn((size_t)cout[i] * ldk, 0.0f), bias((size_t)cst, 0.0f);
            for (int n = 0; n < cout[i]; ++n) {
                if (i == 0)
                    for (int t = 0; t < kw[0]; ++t) wn[(size_t)n * ldk + t] = w->data[(size_t)n * kw[0] + t];
                else
                    for (int ci = 0; ci < cin[i]; ++ci)
                        for (int t = 0; t < kw[i]; ++t) wn[(size_t)n * ldk + t * cpad + ci] = w->data[((size_t)n * cin[i] + ci) * kw[i] + t];
                if (b) bias[n] = b->data[n];
            }
            std::vector<unsigned short> frag(sinc_f16p_wfrag_elems(i));
            if (!sinc_f16p_pack_weights(i, wn.data(), cout[i], ldk, frag.data(), &c->sn_wscale[i])) { c->sinc_f16 = false; break; }   // a non-finite weight: exact kernels
            if ((r = dev_upload(c, frag.data(), frag.size(), &c->sn_wfrag[i], true))) return r;
            if ((r = dev_upload(c, bias.data(), bias.size(), &c->sn_bias16[i], true))) return r;
        }
        c->sinc_ready = true;
    }
    {   // the uploads become a block other contexts with the same weights can share
        std::shared_ptr<PackedWeights> pw = snapshot_packed(c);
        c->packed = pw;
        std::lock_guard<std::mutex> lk(packed_mu);
        for (auto it = packed_cache.begin(); it != packed_cache.end();)   // (entries whose block is gone: a process that cycles through weight sets)
            it = it->second.expired() ? packed_cache.erase(it) : std::next(it);
        packed_cache[wkey] = pw;
    }
    c->finalized = true;
    return UVAD_OK;
}

int uvad_sincnet_configure(uvad_ctx *c, const uvad_sincnet_cfg *q) {
    if (!c || !q) return UVAD_E_ARG;
    if (!c->has_model) return fail(c, UVAD_E_STATE, "uvad_sincnet_configure: context was created without a model configuration");
    if (q->stride < 1 || q->kernel_size < 3 || q->k2 < 3 || q->k3 < 3 || q->n_filters < 1 || q->c2 < 1 || q->c3 < 1)
        return fail(c, UVAD_E_ARG, "bad SincNet configuration");
    const int cout[3] = {q->n_filters, q->c2, q->c3};
    if ((q->n_filters & 1) || (q->c2 & 1)) return fail(c, UVAD_E_UNSUPPORTED, "SincNet input channel counts of the conv stages must be even");
    for (int i = 0; i < 3; ++i)
        if (cout[i] <= 32 || cout[i] > 96) return fail(c, UVAD_E_UNSUPPORTED, "SincNet channel counts must be in 33..96 (two or three 32-wide MFMA column tiles)");
    if (q->c3 != c->mc.in_dim) return fail(c, UVAD_E_ARG, "SincNet output channels != classifier encoding_dim");
    // LDS budget of the widest stage (filter matrix + staging window), 160 KiB per CU
    const int cin[3] = {1, q->n_filters, q->c2}, kw[3] = {q->kernel_size, q->k2, q->k3};
    for (int i = 0; i < 3; ++i) {
        SincConvArgs a{};
        a.Cin = cin[i]; a.Kw = kw[i]; a.stride = i == 0 ? q->stride : 1; a.Ktot = cin[i] * kw[i]; a.Kp = (a.Ktot + 7) / 8 * 8;
        a.Cout = cout[i];
        if (sinc_conv_lds_bytes(a, (cout[i] + 31) / 32, 3) > (size_t)160 * 1024)
            return fail(c, UVAD_E_UNSUPPORTED, "SincNet stage does not fit the 160 KiB LDS (filter matrix is LDS-resident)");
        if (sinc_conv_ept(a, 3) > (i == 0 ? 8 : 48))
            return fail(c, UVAD_E_UNSUPPORTED, "SincNet stage input window too large for the register-prefetched staging");
    }
    c->sc = *q;
    c->has_sinc = true;
    c->sinc_ready = false;
    c->finalized = false;
    return UVAD_OK;
}

int64_t uvad_sincnet_num_frames(const uvad_ctx *c, int64_t S) {
    if (!c || !c->has_sinc || S < 0) return -1;
    const SincLayout l = sinc_carve(c, 1, S);
    return l.ok ? l.Lpool[2] : 0;
}

size_t uvad_sincnet_workspace_bytes(const uvad_ctx *c, int B, int64_t S) {
    if (!c || !c->has_sinc || B <= 0 || S <= 0) return 0;
    return sinc_carve(c, B, S).total;
}

static int sincnet_impl(uvad_ctx *c, const float *d_wav, int B, int64_t S, float *d_feats, void *ws, size_t ws_bytes, hipStream_t s) {
    if (!c->has_sinc) return fail(c, UVAD_E_STATE, "uvad_sincnet: uvad_sincnet_configure has not been called");
    if (!c->finalized || !c->sinc_ready) return fail(c, UVAD_E_STATE, "uvad_sincnet: SincNet tensors not set / uvad_finalize not called");
    const SincLayout l = sinc_carve(c, B, S);
    if (!l.ok) return fail(c, UVAD_E_ARG, "uvad_sincnet: waveform too short for one output frame");
    if (l.Lconv[0] > 0x7fffffff / 4) return fail(c, UVAD_E_UNSUPPORTED, "uvad_sincnet: waveform too long");
    if (ws_bytes < l.total) return fail(c, UVAD_E_WORKSPACE, "workspace too small: need " + std::to_string(l.total) + " bytes");
    char *base = reinterpret_cast<char *>(ws);
    const uvad_sincnet_cfg &q = c->sc;
    float *s0 = reinterpret_cast<float *>(base + l.off_s0);
    HIPCHK(c, launch_wav_stats(d_wav, B, S, S, c->sn_wav_g, c->sn_wav_b, q.eps, s0, s0 + B, s));
    const float *in = d_wav, *in_scale = s0, *in_shift = s0 + B;
    // Split-f16 form (GEMM modes 1 / 3) when the geometry is the reference's and every stage input provably fits the f16 range: an
    // instance-normalised value is at most sqrt(L - 1) in magnitude, so |gamma| * sqrt(L) + |beta| bounds what the staging converts.
    bool f16 = l.f16 && c->sinc_f16 && (c->gemm_mode == 1 || c->gemm_mode == 3);
    for (int i = 0; i < 3 && f16; ++i)
        if (!(c->sn_in_gmax[i] * std::sqrt((double)l.Lin[i]) + c->sn_in_bmax[i] < 60000.0)) f16 = false;
    c->sinc_f16_used = f16;
    if (f16) {
        for (int i = 0; i < 3; ++i) {
            float *P = reinterpret_cast<float *>(base + l.off_P[i]);
            float *part = reinterpret_cast<float *>(base + l.off_part[i]);
            float *sc = reinterpret_cast<float *>(base + l.off_sc[i]);
            SincF16Args a{};
            a.in = in; a.in_bstride = S; a.Lin = (int)l.Lin[i];
            a.in_scale = in_scale; a.in_shift = in_shift; a.n_in = l.Cin[i]; a.slope = q.leaky_slope;
            a.Wfrag = c->sn_wfrag[i]; a.wscale = c->sn_wscale[i]; a.bias = c->sn_bias16[i];
            a.Lpool = (int)l.Lpool[i]; a.ntiles = l.ntiles16[i];
            a.out = P; a.partials = part; a.B = B; a.n_cu = c->n_cu;
            HIPCHK(c, launch_sinc_conv_f16p(i, a, s));
            HIPCHK(c, launch_norm_finalize_f16p(i, part, B, l.ntiles16[i], l.Cout[i], (int)l.Lpool[i], c->sn_g[i], c->sn_b[i], q.eps, sc,
                                                sc + (size_t)B * l.Cout[i], s));
            in = P; in_scale = sc; in_shift = sc + (size_t)B * l.Cout[i];
        }
        HIPCHK(c, launch_sinc_out_f16p(in, in_scale, in_shift, B, l.Cout[2], l.cst[2], (int)l.Lpool[2], q.leaky_slope, d_feats, l.Cout[2], s));
        return UVAD_OK;
    }
    for (int i = 0; i < 3; ++i) {
        float *P = reinterpret_cast<float *>(base + l.off_P[i]);
        float *part = reinterpret_cast<float *>(base + l.off_part[i]);
        float *sc = reinterpret_cast<float *>(base + l.off_sc[i]);
        SincConvArgs a{};
        a.in = in; a.in_bstride = (long long)l.Cin[i] * l.Lin[i]; a.Cin = l.Cin[i]; a.Lin = (int)l.Lin[i];
        a.in_scale = in_scale; a.in_shift = in_shift; a.in_lrelu = i > 0; a.slope = q.leaky_slope;
        a.Wt2 = c->sn_wt[i]; a.bias = c->sn_bias[i];
        a.Kw = l.Kw[i]; a.stride = l.stride[i]; a.Ktot = l.Cin[i] * l.Kw[i]; a.Kp = (a.Ktot + 7) / 8 * 8; a.Cout = l.Cout[i]; a.do_abs = i == 0;
        a.Lconv = (int)l.Lconv[i]; a.Lpool = (int)l.Lpool[i]; a.ntiles = l.ntiles[i];
        a.out = P; a.partials = part; a.B = B; a.n_cu = c->n_cu;
        HIPCHK(c, launch_sinc_conv(a, s));
        HIPCHK(c, launch_norm_finalize(part, B, l.ntiles[i], l.pt[i], l.phases[i], l.NW[i], l.Cout[i], (int)l.Lpool[i], c->sn_g[i], c->sn_b[i], q.eps, sc,
                                       sc + (size_t)B * l.Cout[i], s));
        in = P; in_scale = sc; in_shift = sc + (size_t)B * l.Cout[i];
    }
    HIPCHK(c, launch_sinc_out(in, in_scale, in_shift, B, l.Cout[2], (int)l.Lpool[2], q.leaky_slope, d_feats, l.Cout[2], s));
    return UVAD_OK;
}

int uvad_sincnet(uvad_ctx *c, const float *d_wav, int B, int64_t S, float *d_feats, void *ws, size_t ws_bytes, void *stream) {
    if (!c) return UVAD_E_ARG;
    if (!d_wav || !d_feats || B <= 0 || S <= 0 || !ws) return fail(c, UVAD_E_ARG, "uvad_sincnet: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    return sincnet_impl(c, d_wav, B, S, d_feats, ws, ws_bytes, (hipStream_t)stream);
}

static int classify_impl(uvad_ctx *c, const float *d_feats, int B, int T, float *d_logits, float *d_probs,
                         void *ws, size_t ws_bytes, hipStream_t s, bool record_start, bool check_range,
                         const StreamState *ss, int ld_out, bool feats_in_planes, const FbankArgs *fused_fb);

int uvad_forward_wav(uvad_ctx *c, const float *d_wav, int B, int64_t S, float *d_logits, float *d_probs,
                     void *ws, size_t ws_bytes, void *stream) {
    if (!c) return UVAD_E_ARG;
    if (!d_wav || B <= 0 || S <= 0 || !ws) return fail(c, UVAD_E_ARG, "uvad_forward_wav: bad argument");
    if (!c->has_sinc) return fail(c, UVAD_E_STATE, "uvad_forward_wav: uvad_sincnet_configure has not been called");
    if (!c->finalized) return fail(c, UVAD_E_STATE, "uvad_forward_wav: uvad_finalize has not been called");
    const int64_t T = uvad_sincnet_num_frames(c, S);
    if (T <= 0 || T > 0x7fffffff) return fail(c, UVAD_E_ARG, "uvad_forward_wav: waveform too short for one output frame");
    const WsLayout w = carve(c, B, T);
    const size_t sn = sinc_carve(c, B, S).total;
    if (ws_bytes < w.total + sn) return fail(c, UVAD_E_WORKSPACE, "workspace too small: need " + std::to_string(w.total + sn) + " bytes");
    char *base = reinterpret_cast<char *>(ws);
    float *feats = reinterpret_cast<float *>(base + w.off_feats);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    int r = sincnet_impl(c, d_wav, B, S, feats, base + w.total, ws_bytes - w.total, s);
    if (r) return r;
    return classify_impl(c, feats, B, (int)T, d_logits, d_probs, ws, w.total, s, false, true, nullptr, 0, false, nullptr);
}

int64_t uvad_num_frames(const uvad_ctx *c, int64_t S) {
    if (!c || !c->has_fb || S < 0) return -1;
    if (c->fb.snip_edges) return S < c->fb.frame_len ? 0 : 1 + (S - c->fb.frame_len) / c->fb.frame_shift;
    return (S + c->fb.frame_shift / 2) / c->fb.frame_shift;
}

size_t uvad_workspace_bytes(const uvad_ctx *c, int B, int64_t T) {
    if (!c || !c->has_model || B <= 0 || T <= 0) return 0;
    return carve(c, B, T).total;
}

static int fbank_impl(uvad_ctx *c, const void *d_pcm, int is_i16, int B, int64_t S, float *d_feats, void *stream,
                      unsigned short *plane_hi = nullptr, unsigned short *plane_lo = nullptr, int plane_w = 0) {
    if (!c || !d_pcm || (!d_feats && !plane_hi) || B <= 0 || S <= 0) return fail(c, UVAD_E_ARG, "uvad_fbank: bad argument");
    if (!c->has_fb || !c->tables_set) return fail(c, UVAD_E_STATE, "uvad_fbank: uvad_set_tables has not been called");
    const int64_t T = uvad_num_frames(c, S);
    if (T <= 0) return fail(c, UVAD_E_ARG, "uvad_fbank: input shorter than one frame");
    HIPCHK(c, hipSetDevice(c->device));
    if (B > 65535) return fail(c, UVAD_E_UNSUPPORTED, "uvad_fbank: B > 65535 (grid.y); split the batch");
    FbankArgs a{};
    a.pcm = d_pcm; a.pcm_is_i16 = is_i16; a.B = B; a.S = S; a.T = T;
    a.frame_len = c->fb.frame_len; a.frame_shift = c->fb.frame_shift; a.n_mels = c->fb.n_mels;
    a.preemph = c->fb.preemph; a.log_floor = c->fb.log_floor; a.remove_dc = c->fb.remove_dc; a.snip_edges = c->fb.snip_edges;
    a.feats = d_feats; a.plane_hi = plane_hi; a.plane_lo = plane_lo; a.plane_w = plane_w;
    a.tab.window = c->d_window; a.tab.mel_start = c->d_mel_start; a.tab.mel_len = c->d_mel_len;
    a.tab.mel_w = c->d_mel_w; a.tab.mel_wt = c->d_mel_wt; a.tab.mel_stride = c->mel_stride; a.tab.tw512 = c->d_tw512; a.tab.nyquist = c->mel_nyquist;
    HIPCHK(c, launch_fbank(a, (hipStream_t)stream));
    return UVAD_OK;
}

int uvad_fbank(uvad_ctx *c, const float *d_pcm, int B, int64_t S, float *d_feats, void *stream) {
    return fbank_impl(c, d_pcm, 0, B, S, d_feats, stream);
}
int uvad_fbank_i16(uvad_ctx *c, const int16_t *d_pcm, int B, int64_t S, float *d_feats, void *stream) {
    return fbank_impl(c, d_pcm, 1, B, S, d_feats, stream);
}

// The feed-forward layers one GEMM each (leaky_relu epilogue): workspace buffers Y[last] -> Z[0] -> Z[1] ...; the last one f32.
// modes 1 and 3 use the weight-stationary projection and the fused head for large launches; mode 2 keeps the tile-streaming / per-layer
// kernels everywhere (A/B, reference of the tests); mode 3 = mode 1 with three MFMA products per f32-equivalent product in those
// large-launch kernels and in the 16-sequence recurrence (weights rounded to 22 bits: GemmArgs::products)
static bool mode_is_ws(const uvad_ctx *c) { return c->gemm_mode == 1 || c->gemm_mode == 3; }
static bool mode_fuses_head(const uvad_ctx *c) { return mode_is_ws(c); }
static int mode_products(const uvad_ctx *c) { return c->gemm_mode == 3 ? 3 : 4; }
static int feed_forward_layers(uvad_ctx *c, const WsLayout &w, char *base, int B, int T, bool f16, hipStream_t s) {
    const uvad_model_cfg &m = c->mc;
    auto Yf = [&](int i) { return reinterpret_cast<float *>(base + w.off_Y[i]); };
    auto Zf = [&](int i) { return reinterpret_cast<float *>(base + w.off_Z[i]); };
    auto hi_of = [&](size_t off) { return reinterpret_cast<unsigned short *>(base + off); };
    auto lo_of = [&](size_t off, int width) { return reinterpret_cast<unsigned short *>(base + off) + plane_rows(w.M) * (size_t)width; };
    const int last = (m.num_layers - 1) & 1;
    const float *cur = Yf(last);
    int curw = w.Wd;
    for (int j = 0; j < m.lin_layers; ++j) {
        GemmArgs g{};
        g.W = c->lin_w[j]; g.ldw = gemm_padded_k(curw); g.Wsplit16 = c->lin_w_split16[j]; g.wscale = c->lin_w_scale[j]; g.bias = c->lin_b[j];
        g.M = (int)w.M; g.N = m.lin_hidden; g.B = B; g.T = T; g.act = 1; g.leaky_slope = m.leaky_slope;
        if (f16) {
            const size_t in_off = j == 0 ? w.off_Y[last] : w.off_Z[(j - 1) & 1];
            const int in_w = j == 0 ? w.Wd : w.Zw;
            g.Ah = hi_of(in_off); g.Al = lo_of(in_off, in_w); g.lda = in_w; g.K = in_w;
            if (j + 1 < m.lin_layers) { g.out_planes = 1; g.Ch = hi_of(w.off_Z[j & 1]); g.Cl = lo_of(w.off_Z[j & 1], w.Zw); g.ldc = w.Zw; }
            else { g.C = Zf(j & 1); g.ldc = m.lin_hidden; }
            HIPCHK(c, launch_gemm_f16p(g, s));
        } else {
            g.A = cur; g.lda = curw; g.a_mode = 0; g.K = curw; g.C = Zf(j & 1); g.ldc = m.lin_hidden;
            HIPCHK(c, launch_gemm(g, s));
        }
        cur = Zf(j & 1);
        curw = m.lin_hidden;
    }
    return UVAD_OK;
}

// true if a streaming step of T new frames runs the LSTM stack as one launch (lstm_stack.hip)
static bool stream_uses_stack(const uvad_ctx *c, int T) {
    const uvad_model_cfg &m = c->mc;
    if (!lstm_stack_supported(m.hidden, m.bidirectional ? 2 : 1, m.in_dim, T, m.num_layers)) return false;
    for (int k = 0; k < m.num_layers; ++k)
        if (!c->layers[k].w_ih_img) return false;
    return true;
}

// ... and its feed-forward layers + classifier inside that launch (every feed-forward layer 128 x 128)
static bool stream_head_in_stack(const uvad_ctx *c) {
    const uvad_model_cfg &m = c->mc;
    if (m.lin_layers > LSTM_STACK_MAX_LIN) return false;
    for (int j = 0; j < m.lin_layers; ++j)
        if (!c->lin_w_img[j]) return false;
    return true;
}

// ---- time-chunked layers ---------------------------------------------------------------------------------------------------
// A batch's recurrence (4-sequence form) occupies tiles x directions CUs and the projection in front of it needs the whole chip for a
// fraction of that time; layer l + 1 cannot start before layer l has finished (its first frame needs the backward pass's LAST
// step), so with one launch per stage the projections sit exposed between the recurrences (cfg 2 alone on the GPU: 1.55 of 7.1 ms).
// Inside ONE layer nothing forces that: the forward pass at frame t needs the gate rows up to t only, the backward pass those from t
// on.  The layer is therefore cut into time chunks: the projection of chunk i + 1 (weight-stationary GEMM over the row tiles that
// chunk needs, GemmArgs::ws_tiles) runs on the library's side stream, on the CUs the recurrence leaves idle, while the recurrence of
// chunk i (LstmArgs::steps, state carried in the workspace) runs on the caller's stream; only the first chunk's projection is exposed.
// Same kernels, same arithmetic per row: bit-identical outputs (tests/test_gpu_scale.py).
static int streams_overlap_probe(hipStream_t a, hipStream_t b) {   // 1: kernels on a and b run concurrently, 0: they serialise, < 0: HIP error
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ea); return -1; }
    int result = -1;
    do {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) break;
        // 3 ms of spinning on a, then an empty spin on b: if b's kernel retires while a's is still running the two
        // streams sit on different hardware queues; on one queue b waits behind a.
        if (launch_spin(300000ull, nullptr, a) != hipSuccess || hipEventRecord(ea, a) != hipSuccess) break;
        if (launch_spin(0ull, nullptr, b) != hipSuccess || hipEventRecord(eb, b) != hipSuccess) break;
        if (hipEventSynchronize(eb) != hipSuccess) break;
        const hipError_t q = hipEventQuery(ea);
        if (q != hipSuccess && q != hipErrorNotReady) break;
        result = q == hipErrorNotReady ? 1 : 0;
        if (hipEventSynchronize(ea) != hipSuccess) result = -1;
    } while (0);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return result;
}

// The side stream, PROVEN concurrent with the caller's stream s (HIP maps streams onto a few hardware queues and two streams on one
// queue serialise: the chunked schedule would then only add launches).  Probed once per (context, caller stream); never while s is
// being captured into a graph (the probe synchronises): a capture runs chunked only on a stream that has been used before.
static bool side_stream_for(uvad_ctx *c, hipStream_t s) {
    if (c->side && c->side_for == s && (s != nullptr || c->side_probed_for_null)) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    {
        auto f = c->side_failed.find(s);
        if (f != c->side_failed.end()) {
            if (--f->second > 0) return false;
            c->side_failed.erase(f);   // the verdict has expired: probe again
        }
    }
    // A stream that turns out to share s's hardware queue is kept alive until the search ends: destroyed at once, its queue would be
    // the least loaded one again and the next stream created would land on it too.
    std::vector<hipStream_t> same_queue;
    bool found = false;
    for (int attempt = 0; attempt < 6 && !found; ++attempt) {
        hipStream_t cand = c->side;   // first the one proven beside another caller stream, if any
        c->side = nullptr;
        if (!cand && hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
        const int r = streams_overlap_probe(s, cand);
        if (r == 1) {
            c->side = cand;
            c->side_for = s;
            c->side_probed_for_null = s == nullptr;
            found = true;
        } else {
            same_queue.push_back(cand);
            if (r < 0) break;
        }
    }
    for (hipStream_t q : same_queue) (void)hipStreamDestroy(q);
    if (found) return true;
    c->side_failed[s] = SIDE_RETRY_AFTER;
    return false;
}

static const ChunkPlan *chunk_plan(uvad_ctx *c, int tiles, int T, int D, int chunks, hipStream_t s) {
    const auto key = std::make_tuple(tiles, T, D, chunks);
    auto it = c->chunk_plans.find(key);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) return nullptr;
    if (it != c->chunk_plans.end()) {
        it->second.last_use = ++c->plan_clock;
        if (cs != hipStreamCaptureStatusNone) it->second.pinned = true;
        return &it->second;
    }
    if (cs != hipStreamCaptureStatusNone) return nullptr;   // (a new plan allocates and copies: not inside a capture)
    if (c->chunk_plans.size() >= MAX_CHUNK_PLANS) {   // a caller that walks through many shapes: drop the least recently used list
        auto old = c->chunk_plans.end();
        for (auto j = c->chunk_plans.begin(); j != c->chunk_plans.end(); ++j)
            if (!j->second.pinned && (old == c->chunk_plans.end() || j->second.last_use < old->second.last_use)) old = j;
        if (old != c->chunk_plans.end()) {
            // eager launches that read the list may still be in flight: drain the device before the memory goes back (this path
            // already costs an allocation and a blocking copy); lists a graph captured stay (pinned)
            if (hipDeviceSynchronize() != hipSuccess) return nullptr;
            (void)hipFree(old->second.d_list);
            c->chunk_plans.erase(old);
        }
    }
    ChunkPlan P;
    P.last_use = ++c->plan_clock;
    // Chunk lengths grow geometrically (x 1.3): only chunk 0's projection is exposed, so it is the short one, and projection i + 1 --
    // on the CUs the recurrence leaves free, about half the chip -- still finishes inside recurrence i (per frame a projection on
    // half the chip takes ~2/3 of the recurrence's time: the ratio of consecutive lengths has to stay below ~1.5; 1.0 / 1.15 / 1.3 /
    // 1.4 measured 6.85 / 6.76 / 6.70 / 6.77 ms per cfg-2 step with six chunks, 7.2 unchunked).
    {
        std::vector<double> wgt(chunks);
        double sum = 0.0;
        for (int i = 0; i < chunks; ++i) sum += (wgt[i] = std::pow(1.3, i));
        P.bound.assign(1, 0);
        double acc = 0.0;
        for (int i = 0; i < chunks; ++i) {
            acc += wgt[i];
            const int b = i + 1 == chunks ? T : std::min(T, std::max(P.bound.back() + 1, (int)std::lround(acc / sum * T)));
            if (b > P.bound.back()) P.bound.push_back(b);
        }
        P.bound.back() = T;
        P.chunks = (int)P.bound.size() - 1;
    }
    auto chunk_of = [&](int t) { return (int)(std::upper_bound(P.bound.begin(), P.bound.end(), t) - P.bound.begin()) - 1; };
    const long rows_per_tile = (long)SEQ_TILE * T, M = (long)tiles * rows_per_tile, mt = (M + 127) / 128;
    std::vector<std::vector<int>> lists[2];
    for (int d = 0; d < 2; ++d) lists[d].assign(P.chunks, {});
    for (long r = 0; r < mt; ++r) {
        const long first = 128 * r, last = std::min(128 * r + 127, M - 1);
        int tmin, tmax;
        if (first / rows_per_tile != last / rows_per_tile) { tmin = 0; tmax = T - 1; }   // the tile spans the end of one sequence tile and the start of the next
        else { tmin = (int)((first % rows_per_tile) / SEQ_TILE); tmax = (int)((last % rows_per_tile) / SEQ_TILE); }
        lists[0][chunk_of(tmin)].push_back((int)r);
        lists[1][chunk_of(T - 1 - tmax)].push_back((int)r);
    }
    std::vector<int> flat;
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < P.chunks; ++i) {
            P.off[d].push_back((int)flat.size());
            P.len[d].push_back((int)lists[d][i].size());
            flat.insert(flat.end(), lists[d][i].begin(), lists[d][i].end());
        }
    if (hipMalloc(reinterpret_cast<void **>(&P.d_list), std::max<size_t>(flat.size(), 1) * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemcpy(P.d_list, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(P.d_list); return nullptr; }
    return &(c->chunk_plans[key] = P);
}

// How many time chunks a layer of T frames is cut into when nothing is forced: none unless the recurrence leaves at least a quarter
// of the CUs to the projections; T / 96 chunks, at most 6 (each recurrence launch re-loads its weight image and costs ~10 us).
static int auto_time_chunks(int T, int tiles, int D, int n_cu) {
    if ((long)tiles * D * 4 > 3L * n_cu || T < 192) return 1;
    return std::min(6, T / 96);   // (lengths grow x 1.3 per chunk: six chunks of T = 1000 are 78 ... 290 frames)
}

// check_range: the features come from the caller (or from a front end with learnable scales) and may lie outside the f16
// range; the split-f16 layer-0 projection is then replaced by the exact-f32 one ON THE DEVICE (both are enqueued, a flag
// written by range_flag_kernel lets exactly one of them run), so the call stays asynchronous and capturable.
static int classify_impl(uvad_ctx *c, const float *d_feats, int B, int T, float *d_logits, float *d_probs,
                         void *ws, size_t ws_bytes, hipStream_t s, bool record_start, bool check_range,
                         const StreamState *ss, int ld_out, bool feats_in_planes, const FbankArgs *fused_fb) {
    const uvad_model_cfg &m = c->mc;
    const WsLayout w = carve(c, B, T);
    if (ws_bytes < w.total) return fail(c, UVAD_E_WORKSPACE, "workspace too small: need " + std::to_string(w.total) + " bytes");
    if (w.M > (size_t)0x7fffffff) return fail(c, UVAD_E_UNSUPPORTED, "B*T exceeds 2^31 rows; split the batch");
    char *base = reinterpret_cast<char *>(ws);
    float *G = reinterpret_cast<float *>(base + w.off_G);
    int *flag = reinterpret_cast<int *>(base + w.off_flag);
    const int H = m.hidden, D = w.D, N4 = 4 * H * D;
    // Activation buffer i as f32 rows, or as the (hi, lo) f16 planes of `width` columns
    auto Yf = [&](int i) { return reinterpret_cast<float *>(base + w.off_Y[i]); };
    auto Zf = [&](int i) { return reinterpret_cast<float *>(base + w.off_Z[i]); };
    auto hi_of = [&](size_t off) { return reinterpret_cast<unsigned short *>(base + off); };
    auto lo_of = [&](size_t off, int width) { return reinterpret_cast<unsigned short *>(base + off) + plane_rows(w.M) * (size_t)width; };
    // the split-f16 GEMM needs operands inside the f16 range: weights were checked by uvad_finalize (f16_ok)
    const bool f16 = c->gemm_mode >= 1 && c->f16_ok;
    // the last LSTM layer feeds the classifier kernel directly when there are no feed-forward layers: f32 then
    auto y_planes = [&](int k) { return f16 && (k + 1 < m.num_layers || m.lin_layers > 0); };
    if (c->timing && record_start) HIPCHK(c, hipEventRecord(c->ev[0], s));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev[1], s));
    // Streaming steps of a causal model: the whole stack in one launch (lstm_stack.hip; every layer of a sequence depends on that
    // sequence only, so a workgroup takes its 4 sequences through all layers).  Needs the f32 features (uvad_stream_step asks the
    // feature kernel for them when stream_uses_stack() says so).
    const bool use_stack = ss && !feats_in_planes && stream_uses_stack(c, T);
    if (use_stack) {
        LstmStackArgs q{};
        q.feats = d_feats; q.kin0 = m.in_dim; q.n_layers = m.num_layers;
        for (int k = 0; k < m.num_layers; ++k) { q.wih[k] = c->layers[k].w_ih_img; q.whh[k] = c->layers[k].w_hh; q.bias[k] = c->layers[k].bias; }
        q.h = ss->h; q.c = ss->c; q.layer_stride = ss->layer_stride;
        const int lastl = m.num_layers - 1;
        if (y_planes(lastl)) { q.Yh = hi_of(w.off_Y[lastl & 1]); q.Yl = lo_of(w.off_Y[lastl & 1], w.Wd); }
        else q.Y = Yf(lastl & 1);
        q.ldy = w.Wd; q.tiles = w.tiles; q.T = T; q.B = B;
        // the head in the same launch when every feed-forward layer is 128 -> 128 (the default head)
        const bool head_in = d_logits != nullptr && stream_head_in_stack(c);
        if (fused_fb && !head_in) return fail(c, UVAD_E_STATE, "internal: fused feature stage without the in-launch head");
        if (fused_fb) { q.fb = *fused_fb; q.fb_on = 1; }   // the feature stage in the same launch (uvad_stream_step decided)
        if (head_in) {
            for (int j = 0; j < m.lin_layers; ++j) { q.lin_w[j] = c->lin_w_img[j]; q.lin_b[j] = c->lin_b[j]; }
            q.n_lin = m.lin_layers; q.cls_w = c->cls_w; q.cls_b = c->cls_b; q.slope = m.leaky_slope;
            q.logits = d_logits; q.probs = d_probs; q.ld_out = ld_out > 0 ? ld_out : T;
        }
        HIPCHK(c, launch_lstm_stack(q, s));
        c->rec_tile_used = 4;
        if (head_in) {
            if (c->timing) {
                HIPCHK(c, hipEventRecord(c->layer_ev[2 * m.num_layers], s));
                HIPCHK(c, hipEventRecord(c->ev[2], s));
                HIPCHK(c, hipEventRecord(c->ev[3], s));
                c->ev_valid = true;
            }
            return UVAD_OK;
        }
    }
    // time chunks (see above): only for the 4-sequence recurrence on the weight-stationary split-f16 projections, never for streaming steps
    int NC = 1;
    const ChunkPlan *plan = nullptr;
    if (!ss && !use_stack && f16 && mode_is_ws(c) && c->chunk_mode != 1 && (H == 128 || H == 64) &&
        (c->rec_tile_mode ? c->rec_tile_mode : lstm_auto_tile(w.tiles, D, H, c->n_cu)) == 4) {
        int want = c->chunk_mode > 1 ? std::min(c->chunk_mode, T) : auto_time_chunks(T, w.tiles, D, c->n_cu);
        const long mt = (long)((w.M + 127) / 128);
        while (want > 1 && (mt / want) * (N4 / 128) < 2L * c->n_cu) --want;   // every chunk must still be a launch the weight-stationary kernel takes
        if (want > 1 && side_stream_for(c, s) && (plan = chunk_plan(c, w.tiles, T, D, want, s)) != nullptr) {
            NC = plan->chunks;
            if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            while ((int)c->ev_chunk.size() < NC) {
                hipEvent_t e = nullptr;
                HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                c->ev_chunk.push_back(e);
            }
        }
    }
    c->chunks_used = NC;
    for (int k = 0; k < (use_stack ? 0 : m.num_layers); ++k) {
        const LayerDev &L = c->layers[k];
        GemmArgs g{};
        g.W = L.w_ih; g.ldw = gemm_padded_k(L.in); g.Wsplit16 = L.w_ih_split16; g.wscale = L.w_ih_scale; g.bias = L.bias; g.C = G;
        g.M = (int)w.M; g.N = N4; g.ldc = N4; g.c_blocked = 1; g.B = B; g.T = T; g.act = 0; g.leaky_slope = 0.f;
        if (c->timing) HIPCHK(c, hipEventRecord(c->layer_ev[2 * k], s));
        if (f16) {
            if (k == 0) {
                if (!feats_in_planes)   // (uvad_forward: the feature kernel has written the planes itself)
                    HIPCHK(c, launch_split_features(d_feats, B, T, m.in_dim, w.Fp, w.tiles, hi_of(w.off_fplanes), lo_of(w.off_fplanes, w.Fp),
                                                    check_range ? flag : nullptr, s));
                g.Ah = hi_of(w.off_fplanes); g.Al = lo_of(w.off_fplanes, w.Fp); g.lda = w.Fp; g.K = w.Fp;
                if (check_range) { g.gate = flag; g.gate_run_if_set = 0; }
            } else {
                g.Ah = hi_of(w.off_Y[(k - 1) & 1]); g.Al = lo_of(w.off_Y[(k - 1) & 1], w.Wd); g.lda = w.Wd; g.K = w.Wd;
            }
            g.products = mode_products(c);
            if (NC > 1 && !(k == 0 && check_range) && gemm_f16p_ws_supported(g, c->n_cu)) {
                // ---- the layer in NC time chunks: every chunk's projection on the side stream (they follow each other there), the
                //      recurrence of chunk i on s as soon as projection i has finished, state carried through the workspace
                HIPCHK(c, hipEventRecord(c->ev_fork, s));               // layer k - 1 (or the features) complete
                HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
                unsigned *ctr = reinterpret_cast<unsigned *>(base + w.off_ctr);
                // The projection's persistent grid is sized to the CUs the recurrence leaves free: workgroups beyond that would wait in
                // the dispatcher and take the CUs of a finishing recurrence chunk before the next chunk's workgroups arrive (each holds
                // its CU for a whole projection chunk: measured, the recurrence then ran 20 % longer and the overlap gained nothing).
                const int cu_side = std::max(c->n_cu / 4, c->n_cu - w.tiles * D);
                for (int i = 0; i < NC; ++i) {
                    GemmArgs gi = g;
                    gi.ws_tiles = plan->d_list; gi.ws_dirs = D;
                    for (int d = 0; d < D; ++d) { gi.ws_off[d] = plan->off[d][i]; gi.ws_len[d] = plan->len[d][i]; }
                    HIPCHK(c, launch_gemm_f16p_ws(gi, ctr, cu_side, c->side));   // (chunk 0 too: on the whole chip it was 0.1 ms shorter and the layer's recurrences 0.3 ms longer)
                    HIPCHK(c, hipEventRecord(c->ev_chunk[i], c->side));
                }
                float *hst = reinterpret_cast<float *>(base + w.off_hc);
                float *cst = reinterpret_cast<float *>(base + w.off_hc + align_up((size_t)w.D * w.tiles * SEQ_TILE * m.hidden * sizeof(float)));
                for (int i = 0; i < NC; ++i) {
                    HIPCHK(c, hipStreamWaitEvent(s, c->ev_chunk[i], 0));
                    if (i == 0 && c->timing) HIPCHK(c, hipEventRecord(c->layer_ev[2 * k + 1], s));   // "projection" = what the recurrence had to wait for
                    LstmArgs r{};
                    r.G = G; r.ldg = N4; r.Whh_packed = L.w_hh; r.ldy = w.Wd;
                    if (y_planes(k)) { r.Yh = hi_of(w.off_Y[k & 1]); r.Yl = lo_of(w.off_Y[k & 1], w.Wd); }
                    else r.Y = Yf(k & 1);
                    r.tiles = w.tiles; r.T = T; r.H = H; r.dirs = D; r.tile_mode = 4; r.n_cu = c->n_cu; r.products = 4;
                    r.steps = plan->bound[i + 1] - plan->bound[i];
                    r.t_begin[0] = plan->bound[i];
                    r.t_begin[1] = T - plan->bound[i + 1];
                    r.h0 = i ? hst : nullptr; r.c0 = i ? cst : nullptr; r.hN = hst; r.cN = cst;
                    HIPCHK(c, launch_lstm(r, s, &c->rec_tile_used));
                }
                continue;
            }
            if (mode_is_ws(c) && gemm_f16p_ws_supported(g, c->n_cu))   // large launches: weights stay in registers, bit-identical gates
                HIPCHK(c, launch_gemm_f16p_ws(g, reinterpret_cast<unsigned *>(base + w.off_ctr), c->n_cu, s));
            else
                HIPCHK(c, launch_gemm_f16p(g, s));
            if (k == 0 && check_range) {   // the same projection by the exact kernel, run only if the flag is set
                g.A = d_feats; g.lda = m.in_dim; g.a_mode = 1; g.K = L.in; g.gate_run_if_set = 1;
                HIPCHK(c, launch_gemm(g, s));
            }
        } else {
            g.K = L.in;
            if (k == 0) { g.A = d_feats; g.lda = m.in_dim; g.a_mode = 1; }
            else { g.A = Yf((k - 1) & 1); g.lda = w.Wd; g.a_mode = 0; }
            HIPCHK(c, launch_gemm(g, s));
        }
        if (c->timing) HIPCHK(c, hipEventRecord(c->layer_ev[2 * k + 1], s));
        LstmArgs r{};
        r.G = G; r.ldg = N4; r.Whh_packed = L.w_hh; r.ldy = w.Wd;
        if (L.w_hh16_ok) {
            r.Whh16h_regs = L.w_hh16_regs; r.Whh16h_p2 = L.w_hh16_p2; r.whh16h_scale = L.w_hh16_scale;
            r.Whh16h_p2q = c->gemm_mode == 2 ? nullptr : L.w_hh16_p2q;   // mode 2 = the kernel set kept for comparisons: P2 from its f16 image
            r.p2q_scale = L.w_hh16_p2q_scale;
        }
        if (y_planes(k)) { r.Yh = hi_of(w.off_Y[k & 1]); r.Yl = lo_of(w.off_Y[k & 1], w.Wd); }
        else r.Y = Yf(k & 1);
        r.tiles = w.tiles; r.T = T; r.H = H; r.dirs = D; r.tile_mode = ss ? 4 : c->rec_tile_mode; r.n_cu = c->n_cu;
        r.products = f16 ? mode_products(c) : 4;
        if (ss) {   // carried (h, c) of this layer, updated in place
            r.h0 = r.hN = ss->h + (size_t)k * ss->layer_stride;
            r.c0 = r.cN = ss->c + (size_t)k * ss->layer_stride;
        }
        HIPCHK(c, launch_lstm(r, s, &c->rec_tile_used));
    }
    if (c->timing) HIPCHK(c, hipEventRecord(c->layer_ev[2 * m.num_layers], s));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev[2], s));
    const int last = (m.num_layers - 1) & 1;
    // The default head (two 128-unit feed-forward layers) in split-f16 mode, large launches: feed-forward layers, classifier and
    // sigmoid in one kernel (head_fused.hip); the LSTM output planes are read once and nothing but the logits is written.
    // (uvad_get_taps recomputes the feed-forward output from those planes when it is asked for.)
    if (f16 && mode_fuses_head(c) && head_fused_supported(w.Wd, m.lin_hidden, m.lin_layers, (long long)w.M, c->n_cu)) {
        HeadArgs h{};
        h.Yh = hi_of(w.off_Y[last]); h.Yl = lo_of(w.off_Y[last], w.Wd); h.M = (long long)w.M; h.K1 = w.Wd;
        h.W1 = c->lin_w_split16[0]; h.W2 = c->lin_w_split16[1]; h.w1scale = c->lin_w_scale[0]; h.w2scale = c->lin_w_scale[1];
        h.b1 = c->lin_b[0]; h.b2 = c->lin_b[1]; h.wc = c->cls_w; h.bc = c->cls_b; h.slope = m.leaky_slope;
        h.logits = d_logits; h.probs = d_probs; h.tiles = w.tiles; h.T = T; h.B = B; h.ld_out = ld_out > 0 ? ld_out : T;
        h.counter = reinterpret_cast<unsigned *>(base + w.off_ctr);
        h.products = mode_products(c);
        HIPCHK(c, launch_head_fused(h, c->n_cu, s));
        if (c->timing) {
            HIPCHK(c, hipEventRecord(c->ev[3], s));
            c->ev_valid = true;
        }
        return UVAD_OK;
    }
    int r_ff = feed_forward_layers(c, w, base, B, T, f16, s);
    if (r_ff) return r_ff;
    const float *cur = m.lin_layers > 0 ? Zf((m.lin_layers - 1) & 1) : Yf(last);
    const int curw = m.lin_layers > 0 ? m.lin_hidden : w.Wd;
    ClsArgs q{};
    q.Z = cur; q.ldz = curw; q.K = curw; q.w = c->cls_w; q.b = c->cls_b; q.logits = d_logits; q.probs = d_probs;
    q.tiles = w.tiles; q.T = T; q.B = B; q.ld_out = ld_out > 0 ? ld_out : T;
    HIPCHK(c, launch_classifier(q, s));
    if (c->timing) {
        HIPCHK(c, hipEventRecord(c->ev[3], s));
        c->ev_valid = true;
    }
    return UVAD_OK;
}

int uvad_classify(uvad_ctx *c, const float *d_feats, int B, int T, float *d_logits, float *d_probs,
                  void *ws, size_t ws_bytes, void *stream) {
    if (!c) return UVAD_E_ARG;
    if (!d_feats || B <= 0 || T <= 0 || !ws) return fail(c, UVAD_E_ARG, "uvad_classify: bad argument");
    if (!c->finalized) return fail(c, UVAD_E_STATE, "uvad_classify: uvad_finalize has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    return classify_impl(c, d_feats, B, T, d_logits, d_probs, ws, ws_bytes, (hipStream_t)stream, true, true, nullptr, 0, false, nullptr);
}

static int forward_impl(uvad_ctx *c, const void *d_pcm, int is_i16, int B, int64_t S, float *d_logits, float *d_probs,
                        void *ws, size_t ws_bytes, void *stream) {
    if (!c) return UVAD_E_ARG;
    if (!d_pcm || B <= 0 || S <= 0 || !ws) return fail(c, UVAD_E_ARG, "uvad_forward: bad argument");
    if (!c->finalized) return fail(c, UVAD_E_STATE, "uvad_forward: uvad_finalize has not been called");
    if (!c->has_fb || !c->tables_set) return fail(c, UVAD_E_STATE, "uvad_forward: uvad_set_tables has not been called");
    if (c->fb.n_mels != c->mc.in_dim) return fail(c, UVAD_E_ARG, "uvad_forward: n_mels != encoding_dim");
    const int64_t T = uvad_num_frames(c, S);
    if (T <= 0 || T > 0x7fffffff) return fail(c, UVAD_E_ARG, "uvad_forward: bad frame count");
    const WsLayout w = carve(c, B, T);
    if (ws_bytes < w.total) return fail(c, UVAD_E_WORKSPACE, "workspace too small: need " + std::to_string(w.total) + " bytes");
    float *feats = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + w.off_feats);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev[0], s));
    // split-f16 GEMM mode: the feature kernel writes the two f16 planes the first projection reads (K-blocked, tile-major rows) and
    // the f32 feature tensor never exists; exact-f32 mode: f32 features.  (log-mel values are within +-90: no range check.)
    const bool planes = c->gemm_mode >= 1 && c->f16_ok;
    unsigned short *ph = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(ws) + w.off_fplanes);
    int r = planes ? fbank_impl(c, d_pcm, is_i16, B, S, nullptr, stream, ph, ph + plane_rows(w.M) * (size_t)w.Fp, w.Fp)
                   : fbank_impl(c, d_pcm, is_i16, B, S, feats, stream);
    if (r) return r;
    return classify_impl(c, feats, B, (int)T, d_logits, d_probs, ws, ws_bytes, s, false, false, nullptr, 0, planes, nullptr);
}

int uvad_forward(uvad_ctx *c, const float *d_pcm, int B, int64_t S, float *d_logits, float *d_probs,
                 void *ws, size_t ws_bytes, void *stream) {
    return forward_impl(c, d_pcm, 0, B, S, d_logits, d_probs, ws, ws_bytes, stream);
}

int uvad_forward_i16(uvad_ctx *c, const int16_t *d_pcm, int B, int64_t S, float *d_logits, float *d_probs,
                     void *ws, size_t ws_bytes, void *stream) {
    return forward_impl(c, d_pcm, 1, B, S, d_logits, d_probs, ws, ws_bytes, stream);
}

int uvad_get_taps(uvad_ctx *c, int B, int T, float *d_lstm_out, float *d_lin_out, const void *ws, void *stream) {
    if (!c || !ws || B <= 0 || T <= 0) return UVAD_E_ARG;
    if (!c->finalized) return fail(c, UVAD_E_STATE, "not finalized");
    HIPCHK(c, hipSetDevice(c->device));
    const uvad_model_cfg &m = c->mc;
    const WsLayout w = carve(c, B, T);
    const char *base = reinterpret_cast<const char *>(ws);
    if (d_lstm_out) {
        const char *y = base + w.off_Y[(m.num_layers - 1) & 1];
        const bool planes = c->gemm_mode >= 1 && c->f16_ok && m.lin_layers > 0;   // what classify_impl made the last layer write
        HIPCHK(c, launch_untile(y, planes ? y + plane_rows(w.M) * (size_t)w.Wd * sizeof(unsigned short) : nullptr, w.Wd, w.Wd, d_lstm_out, w.tiles, T, B,
                                (hipStream_t)stream));
    }
    if (d_lin_out) {
        if (m.lin_layers <= 0) return fail(c, UVAD_E_ARG, "model has no feed-forward layers");
        const bool f16 = c->gemm_mode >= 1 && c->f16_ok;
        if (f16 && mode_fuses_head(c) && head_fused_supported(w.Wd, m.lin_hidden, m.lin_layers, (long long)w.M, c->n_cu)) {
            // the fused head keeps the feed-forward activations on chip: recompute them from the LSTM output planes of the last call
            // with the per-layer kernels -- four products, exact weights: the bits of the fused head in modes 1 / 2.  In mode 3 the
            // fused head ran THREE products on rounded weights and the per-layer kernels have no such form, so the tap that
            // produced the logits cannot be reproduced: refuse instead of returning a near-miss.
            if (mode_products(c) != 4)
                return fail(c, UVAD_E_UNSUPPORTED, "uvad_get_taps: the feed-forward tap is not available in GEMM mode 3 (fused head, three products); "
                                                   "pass d_lin_out = NULL or use mode 1");
            int r = feed_forward_layers(c, w, const_cast<char *>(base), B, T, true, (hipStream_t)stream);
            if (r) return r;
        }
        const float *z = reinterpret_cast<const float *>(base + w.off_Z[(m.lin_layers - 1) & 1]);
        HIPCHK(c, launch_untile(z, nullptr, m.lin_hidden, m.lin_hidden, d_lin_out, w.tiles, T, B, (hipStream_t)stream));
    }
    return UVAD_OK;
}

// ---- streaming ---------------------------------------------------------------------------------
// state layout (bytes, 256-aligned blocks): tail[2][B][frame_len] f32 (ping-pong) | per layer: h [Bpad][H], c [Bpad][H]
extern "C++" {
namespace {
struct StreamLayout {
    int tail = 0, Bpad = 0;
    size_t off_tail[2] = {0, 0}, off_h = 0, off_c = 0, layer_stride = 0, total = 0;
};
StreamLayout stream_layout(const uvad_ctx *c, int B) {
    StreamLayout L;
    L.tail = c->fb.frame_len;
    L.Bpad = (B + SEQ_TILE - 1) / SEQ_TILE * SEQ_TILE;
    size_t o = 0;
    for (int i = 0; i < 2; ++i) { L.off_tail[i] = o; o += align_up((size_t)B * L.tail * sizeof(float)); }
    L.layer_stride = align_up((size_t)L.Bpad * c->mc.hidden * sizeof(float));
    L.off_h = o; o += L.layer_stride * c->mc.num_layers;
    L.off_c = o; o += L.layer_stride * c->mc.num_layers;
    L.total = o;
    return L;
}
int stream_max_frames(const uvad_ctx *c, int chunk) { return chunk / c->fb.frame_shift + 1; }
// What the next step of a stream group does, from its host-side counters: which tail buffer it reads, whether it is the first
// chunk (left reflection), how many frames it completes and where the first of them starts inside a staging row.
struct StreamPlan { int parity = 0, first = 0, k = 0; int64_t offset = 0, n_after = 0; };
StreamPlan stream_plan(const uvad_ctx *c, const StreamCounters &sc, int chunk) {
    const int L = c->fb.frame_len, sh = c->fb.frame_shift, n_left = (L - sh) / 2;
    StreamPlan p;
    p.parity = (int)(sc.n_steps & 1);
    p.first = sc.n_samples == 0 ? 1 : 0;
    const int64_t n_prev = sc.n_samples, n = n_prev + chunk;
    // frame t spans [t*sh - n_left, t*sh - n_left + L): complete once n >= t*sh - n_left + L
    const int64_t f_hi = n + n_left - L >= 0 ? (n + n_left - L) / sh : -1;
    p.k = (int)(f_hi - (sc.n_frames - 1));
    if (p.k < 0) p.k = 0;
    p.n_after = n;
    p.offset = p.k > 0 ? sc.n_frames * sh - n_left - (n_prev - L) : 0;   // offset of the first new frame inside a staging row (tail = L samples)
    return p;
}
}  // namespace
}  // extern "C++"

size_t uvad_stream_state_bytes(const uvad_ctx *c, int B) {
    if (!c || !c->has_fb || !c->has_model || B <= 0) return 0;
    return stream_layout(c, B).total;
}

size_t uvad_stream_workspace_bytes(const uvad_ctx *c, int B, int chunk) {
    if (!c || !c->has_fb || !c->has_model || B <= 0 || chunk <= 0) return 0;
    const size_t staging = align_up((size_t)B * (c->fb.frame_len + chunk) * sizeof(float));
    return staging + carve(c, B, stream_max_frames(c, chunk)).total;
}

int uvad_stream_reset(uvad_ctx *c, void *d_state, int B, void *stream) {
    if (!c || !d_state || B <= 0) return UVAD_E_ARG;
    if (!c->has_fb || !c->has_model) return fail(c, UVAD_E_STATE, "streaming needs both a fbank and a model configuration");
    if (c->mc.bidirectional) return fail(c, UVAD_E_UNSUPPORTED, "streaming needs a causal model (lstm.bidirectional = False)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(d_state, 0, stream_layout(c, B).total, (hipStream_t)stream));
    c->streams[d_state] = StreamCounters();
    return UVAD_OK;
}

int uvad_stream_step(uvad_ctx *c, const float *d_pcm_chunk, int B, int chunk, void *d_state, float *d_logits, int ld_logits,
                     void *ws, size_t ws_bytes, void *stream) {
    if (!c) return UVAD_E_ARG;
    if (!d_pcm_chunk || !d_state || !d_logits || !ws || B <= 0 || chunk <= 0) return fail(c, UVAD_E_ARG, "uvad_stream_step: bad argument");
    if (!c->finalized || !c->has_fb || !c->tables_set) return fail(c, UVAD_E_STATE, "uvad_stream_step: context not ready (tables / weights)");
    if (c->mc.bidirectional) return fail(c, UVAD_E_UNSUPPORTED, "streaming needs a causal model (lstm.bidirectional = False)");
    if (c->fb.snip_edges) return fail(c, UVAD_E_UNSUPPORTED, "streaming implements the centred (snip_edges = 0) framing only");
    if (c->fb.n_mels != c->mc.in_dim) return fail(c, UVAD_E_ARG, "n_mels != encoding_dim");
    auto it = c->streams.find(d_state);
    if (it == c->streams.end()) return fail(c, UVAD_E_STATE, "uvad_stream_step: call uvad_stream_reset on this state first");
    StreamCounters &sc = it->second;
    const int L = c->fb.frame_len, sh = c->fb.frame_shift, n_left = (L - sh) / 2;
    if (sc.n_samples == 0 && chunk < n_left) return fail(c, UVAD_E_ARG, "first chunk must hold at least (frame_len - shift)/2 samples");
    if (ws_bytes < uvad_stream_workspace_bytes(c, B, chunk)) return fail(c, UVAD_E_WORKSPACE, "stream workspace too small");
    const StreamLayout S = stream_layout(c, B);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    char *st = reinterpret_cast<char *>(d_state);
    char *wsb = reinterpret_cast<char *>(ws);
    float *staging = reinterpret_cast<float *>(wsb);
    const size_t staging_bytes = align_up((size_t)B * (L + chunk) * sizeof(float));
    const StreamPlan plan = stream_plan(c, sc, chunk);
    const int par = plan.parity;
    const int k = plan.k;
    // A step that produces frames reads its rows straight from the chunk and the carried tail inside the feature kernel, which also
    // writes the next tail (FbankArgs::vs_*); only a step without frames (a short first chunk) runs the staging kernel for the tail.
    if (k <= 0)
        HIPCHK(c, launch_stream_stage(d_pcm_chunk, B, chunk, S.tail, n_left, plan.first,
                                      reinterpret_cast<const float *>(st + S.off_tail[par]),
                                      reinterpret_cast<float *>(st + S.off_tail[par ^ 1]), staging, s));
    sc.n_samples = plan.n_after;
    sc.n_steps += 1;
    if (k <= 0) return 0;
    if (k > ld_logits) return fail(c, UVAD_E_ARG, "ld_logits smaller than the number of new frames");
    const int64_t o = plan.offset;   // >= 0
    sc.n_frames += k;
    void *cws = wsb + staging_bytes;
    const WsLayout w = carve(c, B, k);
    float *feats = reinterpret_cast<float *>(reinterpret_cast<char *>(cws) + w.off_feats);
    FbankArgs fa{};
    fa.pcm = staging + o; fa.pcm_is_i16 = 0; fa.B = B; fa.S = (int64_t)(S.tail + chunk) - o; fa.T = k;
    fa.vs_chunk = d_pcm_chunk; fa.vs_tail_in = reinterpret_cast<const float *>(st + S.off_tail[par]);
    fa.vs_tail_out = reinterpret_cast<float *>(st + S.off_tail[par ^ 1]);
    fa.vs_tail = S.tail; fa.vs_chunk_len = chunk; fa.vs_first = plan.first; fa.vs_n_left = n_left; fa.vs_offset = (int)o;
    fa.row_stride = S.tail + chunk;
    fa.frame_len = L; fa.frame_shift = sh; fa.n_mels = c->fb.n_mels;
    fa.preemph = c->fb.preemph; fa.log_floor = c->fb.log_floor; fa.remove_dc = c->fb.remove_dc; fa.snip_edges = 1;
    fa.feats = feats;
    // as in uvad_forward: features straight into the first projection's operand planes -- unless the one-launch stack runs (f32 features)
    const bool planes = c->gemm_mode >= 1 && c->f16_ok && !stream_uses_stack(c, k);
    if (planes) {
        fa.plane_hi = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(cws) + w.off_fplanes);
        fa.plane_lo = fa.plane_hi + plane_rows(w.M) * (size_t)w.Fp;
        fa.plane_w = w.Fp;
    }
    fa.tab.window = c->d_window; fa.tab.mel_start = c->d_mel_start; fa.tab.mel_len = c->d_mel_len;
    fa.tab.mel_w = c->d_mel_w; fa.tab.mel_wt = c->d_mel_wt; fa.tab.mel_stride = c->mel_stride; fa.tab.tw512 = c->d_tw512; fa.tab.nyquist = c->mel_nyquist;
    // One launch for the whole step when the stack kernel also takes the head and the feature stage fits beside it (lstm_stack.hip)
    const bool fuse_fb = !planes && stream_uses_stack(c, k) && stream_head_in_stack(c) && lstm_stack_fb_lds_bytes(fa, k) > 0;
    if (!fuse_fb) HIPCHK(c, launch_fbank(fa, s));
    StreamState ss;
    ss.h = reinterpret_cast<float *>(st + S.off_h); ss.c = reinterpret_cast<float *>(st + S.off_c);
    ss.layer_stride = S.layer_stride / sizeof(float);
    const bool timing = c->timing;
    c->timing = false;
    const int r = classify_impl(c, feats, B, k, d_logits, nullptr, cws, ws_bytes - staging_bytes, s, false, false, &ss, ld_logits, planes, fuse_fb ? &fa : nullptr);
    c->timing = timing;
    return r < 0 ? r : k;
}

int uvad_stream_peek(const uvad_ctx *c, const void *d_state, int chunk, int *k, int64_t *offset, int *parity, int *first) {
    if (!c || !d_state || chunk <= 0 || !k || !offset || !parity || !first) return UVAD_E_ARG;
    auto it = c->streams.find(const_cast<void *>(d_state));
    if (it == c->streams.end() || !c->has_fb) return UVAD_E_STATE;
    const StreamPlan p = stream_plan(c, it->second, chunk);
    *k = p.k; *offset = p.offset; *parity = p.parity; *first = p.first;
    return UVAD_OK;
}

int uvad_stream_advance(uvad_ctx *c, void *d_state, int chunk) {
    if (!c || !d_state || chunk <= 0) return UVAD_E_ARG;
    auto it = c->streams.find(d_state);
    if (it == c->streams.end() || !c->has_fb) return fail(c, UVAD_E_STATE, "uvad_stream_advance: call uvad_stream_reset on this state first");
    StreamCounters &sc = it->second;
    const StreamPlan p = stream_plan(c, sc, chunk);
    sc.n_samples = p.n_after;
    sc.n_steps += 1;
    sc.n_frames += p.k;
    return p.k;
}

int uvad_median_filter(uvad_ctx *c, const float *d_probs, int B, int T, int kernel, uint8_t *d_labels, void *stream) {
    if (!c || !d_probs || !d_labels || B <= 0 || T <= 0) return UVAD_E_ARG;
    if (kernel < 1 || kernel % 2 == 0) return fail(c, UVAD_E_ARG, "median kernel must be odd");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_median(d_probs, B, T, kernel, d_labels, (hipStream_t)stream));
    return UVAD_OK;
}

int uvad_der_counts(uvad_ctx *c, const uint8_t *d_pred, const uint8_t *d_gt, int B, int T, uint32_t *d_counts, void *stream) {
    if (!c || !d_pred || !d_gt || !d_counts || B <= 0 || T <= 0) return UVAD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_der(d_pred, d_gt, B, T, d_counts, (hipStream_t)stream));
    return UVAD_OK;
}

int uvad_label_runs(uvad_ctx *c, const uint8_t *d_labels, int B, int T, int max_runs, int32_t *d_runs, int32_t *d_counts, void *stream) {
    if (!c || !d_labels || !d_runs || !d_counts || B <= 0 || T <= 0 || max_runs <= 0) return UVAD_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_runs(d_labels, B, T, max_runs, d_runs, d_counts, (hipStream_t)stream));
    return UVAD_OK;
}

int uvad_set_gemm_mode(uvad_ctx *c, int mode) {
    if (!c) return UVAD_E_ARG;
    if (mode < 0 || mode > 3)
        return fail(c, UVAD_E_ARG, "gemm mode must be 0 (exact f32 MFMA), 1 (split f16, 4 products), 2 (the same, tile-streaming kernels only) or 3 (split f16, 3 products)");
    c->gemm_mode = mode;
    return UVAD_OK;
}

int uvad_set_recurrent_tile(uvad_ctx *c, int sequences) {
    if (!c) return UVAD_E_ARG;
    if (sequences != 0 && sequences != 4 && sequences != 16) return fail(c, UVAD_E_ARG, "recurrent tile must be 0 (by batch size), 4 or 16 sequences per workgroup");
    if (sequences == 16 && c->has_model && c->mc.hidden != 128) return fail(c, UVAD_E_UNSUPPORTED, "the 16-sequence recurrent kernel exists for hidden_size 128 only");
    c->rec_tile_mode = sequences;
    return UVAD_OK;
}

int uvad_get_recurrent_tile(const uvad_ctx *c) { return c ? c->rec_tile_used : UVAD_E_ARG; }
int uvad_weights_shared_by(const uvad_ctx *c) {
    if (!c) return UVAD_E_ARG;
    return c->packed ? (int)c->packed.use_count() : 0;
}

int uvad_get_sincnet_form(const uvad_ctx *c) {
    if (!c) return UVAD_E_ARG;
    return c->sinc_f16_used ? 1 : 0;
}

int uvad_get_p2_on_fp8(const uvad_ctx *c) {
    if (!c) return UVAD_E_ARG;
    if (!c->has_model || !c->finalized || c->mc.hidden != 128 || c->gemm_mode == 2) return 0;
    for (const LayerDev &L : c->layers)
        if (!L.w_hh16_ok || !L.w_hh16_p2q) return 0;
    return 1;
}

int uvad_set_time_chunks(uvad_ctx *c, int chunks) {
    if (!c) return UVAD_E_ARG;
    if (chunks < 0 || chunks > 64) return fail(c, UVAD_E_ARG, "time chunks must be 0 (automatic), 1 (off) or 2 .. 64");
    c->chunk_mode = chunks;
    return UVAD_OK;
}

int uvad_get_time_chunks(const uvad_ctx *c) { return c ? c->chunks_used : UVAD_E_ARG; }

int uvad_recurrent_tile_for(const uvad_ctx *c, int B) {
    if (!c || !c->has_model || B <= 0) return UVAD_E_ARG;
    return lstm_auto_tile((B + SEQ_TILE - 1) / SEQ_TILE, c->mc.bidirectional ? 2 : 1, c->mc.hidden, c->n_cu);
}

int uvad_streams_overlap(uvad_ctx *c, void *stream_a, void *stream_b) {
    if (!c) return UVAD_E_ARG;
    if (stream_a == stream_b) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    const int result = streams_overlap_probe((hipStream_t)stream_a, (hipStream_t)stream_b);   // (the probe of the time-chunked layers, above)
    if (result < 0) return fail(c, UVAD_E_HIP, "uvad_streams_overlap: HIP error while probing");
    return result;
}

int uvad_set_timing(uvad_ctx *c, int enabled) {
    if (!c) return UVAD_E_ARG;
    c->timing = enabled != 0;
    c->ev_valid = false;
    return UVAD_OK;
}

int uvad_get_timing(uvad_ctx *c, float ms[5]) {
    if (!c || !ms) return UVAD_E_ARG;
    if (!c->ev_valid) return fail(c, UVAD_E_STATE, "no timed call recorded");
    HIPCHK(c, hipEventSynchronize(c->ev[3]));
    const int L = c->mc.num_layers;
    float fb = 0.f, proj = 0.f, rec = 0.f, head = 0.f, total = 0.f, t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&fb, c->ev[0], c->ev[1]));
    for (int k = 0; k < L; ++k) {
        HIPCHK(c, hipEventElapsedTime(&t, c->layer_ev[2 * k], c->layer_ev[2 * k + 1]));
        proj += t;
        HIPCHK(c, hipEventElapsedTime(&t, c->layer_ev[2 * k + 1], c->layer_ev[2 * k + 2]));
        rec += t;
    }
    HIPCHK(c, hipEventElapsedTime(&head, c->ev[2], c->ev[3]));
    HIPCHK(c, hipEventElapsedTime(&total, c->ev[0], c->ev[3]));
    ms[0] = fb; ms[1] = proj; ms[2] = rec; ms[3] = head; ms[4] = total;
    return UVAD_OK;
}

int uvad_get_layer_timing(uvad_ctx *c, float *ms, int n) {
    if (!c || !ms) return UVAD_E_ARG;
    if (!c->ev_valid) return fail(c, UVAD_E_STATE, "no timed call recorded");
    const int L = c->mc.num_layers;
    if (n < 2 * L) return fail(c, UVAD_E_ARG, "uvad_get_layer_timing: ms holds fewer than 2 * num_layers floats");
    HIPCHK(c, hipEventSynchronize(c->ev[3]));
    for (int k = 0; k < 2 * L; ++k) HIPCHK(c, hipEventElapsedTime(&ms[k], c->layer_ev[k], c->layer_ev[k + 1]));
    return 2 * L;
}

const char *uvad_last_error(const uvad_ctx *c) { return c ? c->err.c_str() : "null context"; }

void uvad_destroy(uvad_ctx *c) {
    if (!c) return;
    if (!c->allocs.empty() || !c->weight_allocs.empty() || c->packed || c->ev[0]) (void)hipSetDevice(c->device);
    free_weights(c);
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto &ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &kv : c->chunk_plans)
        if (kv.second.d_list) (void)hipFree(kv.second.d_list);
    for (auto &ev : c->ev_chunk) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

}  // extern "C"

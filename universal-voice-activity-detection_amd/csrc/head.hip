// head.hip -- per-frame tail of the classifier and the post-processing next to it.
//   classifier : logit = z . w + b ; prob = sigmoid(logit)
//                (reference: self.activation(self.classifier(outputs)),
//                 src/models/segmentation/PyanNet2.py:187)
//   untile     : tile-major activation rows -> canonical [B][T][W] (parity taps)
//   median     : threshold 0.5 + odd binary median, zero padded edges
//                (reference: median_filter, src/utils/helper.py:66-97, i.e. scipy.signal.medfilt)
//   runs       : 0/1 label rows -> (start, stop) frame pairs (reference: get_new_cuts, predict.py:472-490)
// All three are HBM-bound streaming kernels: 16-byte loads, wave-shuffle reductions, no LDS.
#include "uvad_internal.h"

namespace uvad {

namespace {

// 16 lanes per row: each lane owns K/16 (<= 16 per pass) elements, 4 rows per wave.
__global__ __launch_bounds__(256) void classifier_kernel(ClsArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 15, rsel = lane >> 4;
    const long long rows = (long long)a.tiles * a.T * SEQ_TILE;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 4 + rsel;
    float acc = 0.f;
    if (m < rows) {
        const float *z = a.Z + (size_t)m * a.ldz;
        for (int k = sub * 4; k < a.K; k += 64) {
            if (k + 3 < a.K) {
                const float4 v = *reinterpret_cast<const float4 *>(z + k);
                const float4 w = *reinterpret_cast<const float4 *>(a.w + k);
                acc = __builtin_fmaf(v.x, w.x, acc);
                acc = __builtin_fmaf(v.y, w.y, acc);
                acc = __builtin_fmaf(v.z, w.z, acc);
                acc = __builtin_fmaf(v.w, w.w, acc);
            } else {
                for (int e = k; e < a.K; ++e) acc = __builtin_fmaf(z[e], a.w[e], acc);
            }
        }
    }
    // reduce over the 16 lanes of the row (xor butterflies stay inside the 16-lane group)
    acc += __shfl_xor(acc, 8);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (sub == 0 && m < rows) {
        const long long per_tile = (long long)a.T * SEQ_TILE;
        const int tile = (int)(m / per_tile);
        const int rem = (int)(m - (long long)tile * per_tile);
        const int t = rem / SEQ_TILE, j = rem - t * SEQ_TILE;
        const int b = tile * SEQ_TILE + j;
        if (b < a.B) {
            const float logit = acc + a.b[0];
            const size_t o = (size_t)b * a.ld_out + t;
            if (a.logits) a.logits[o] = logit;
            if (a.probs) a.probs[o] = 1.0f / (1.0f + expf(-logit));
        }
    }
}

// src_lo == nullptr: src is f32; else src / src_lo are the two f16 planes (value = hi + lo * 2^-11)
__global__ __launch_bounds__(256) void untile_kernel(const void *src, const void *src_lo, int ld, int W, float *dst, int tiles, int T, int B) {
    const long long n4 = (long long)B * T * (W / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(i % (W / 4));
        const long long bt = i / (W / 4);
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int tile = b / SEQ_TILE, j = b - tile * SEQ_TILE;
        const size_t m = ((size_t)tile * T + t) * SEQ_TILE + j;
        if (!src_lo) {
            reinterpret_cast<float4 *>(dst)[i] = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(src) + m * ld + q * 4);
        } else {
            const size_t o = plane_index(m, q * 4, ld);   // K-blocked planes: 4 consecutive columns stay inside one 16-column block
            const _Float16 *h = reinterpret_cast<const _Float16 *>(src) + o, *l = reinterpret_cast<const _Float16 *>(src_lo) + o;
            reinterpret_cast<float4 *>(dst)[i] = make_float4(__builtin_fmaf((float)l[0], 0.00048828125f, (float)h[0]), __builtin_fmaf((float)l[1], 0.00048828125f, (float)h[1]),
                                                             __builtin_fmaf((float)l[2], 0.00048828125f, (float)h[2]), __builtin_fmaf((float)l[3], 0.00048828125f, (float)h[3]));
        }
    }
}

__global__ __launch_bounds__(256) void median_kernel(const float *probs, int B, int T, int half, uint8_t *labels) {
    const long long n = (long long)B * T;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i % T);
    const float *p = probs + (i - t);
    int ones = 0;
    for (int d = -half; d <= half; ++d) {
        const int u = t + d;
        if (u >= 0 && u < T) ones += !(p[u] < 0.5f);
    }
    labels[i] = ones > half ? 1 : 0;
}

// False-alarm / missed-detection frame counts per row (reference: get_false_alarm / get_missed_detection,
// src/scripts/predict.py:666-673, applied to 0/1 frame tensors): one wave per row, 16 bytes per lane per
// pass, wave-shuffle reduction.  counts[b] = {#(gt==0 & pred==1), #(gt==1 & pred==0)}.
__global__ __launch_bounds__(256) void der_kernel(const uint8_t *pred, const uint8_t *gt, int B, int T, uint32_t *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const uint8_t *p = pred + (size_t)b * T, *g = gt + (size_t)b * T;
    unsigned fa = 0, md = 0;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    int t = 0;
    if (vec) {
        for (t = lane * 16; t + 15 < T; t += 64 * 16) {
            const uint4 pv = *reinterpret_cast<const uint4 *>(p + t), gv = *reinterpret_cast<const uint4 *>(g + t);
            const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // labels are 0/1 bytes: bit 0 of every byte
                const unsigned pb = pw[i] & 0x01010101u, gb = gw[i] & 0x01010101u;
                fa += __builtin_popcount(pb & ~gb);
                md += __builtin_popcount(gb & ~pb);
            }
        }
        t = T / 16 * 16;
    }
    for (int u = t + lane; u < T; u += 64) {
        const unsigned pb = p[u] & 1u, gb = g[u] & 1u;
        fa += pb & ~gb & 1u;
        md += gb & ~pb & 1u;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { fa += __shfl_xor(fa, o); md += __shfl_xor(md, o); }
    if (lane == 0) { counts[2 * b] = fa; counts[2 * b + 1] = md; }
}

// Run-length extraction of 0/1 label rows (reference: the per-frame Python walk of get_new_cuts,
// src/scripts/predict.py:472-490): run i of row b starts at frame runs[b][i][0] (label 0 -> 1) and its first
// non-speech frame is runs[b][i][1] (T when the run is still open at the end).  One wave per row walks the row 64
// frames at a time; a ballot + popcount of the lanes below gives each edge its rank, so runs come out in order.
// counts[b] is the true number of runs even when it exceeds max_runs (the host checks for overflow).
__global__ __launch_bounds__(256) void runs_kernel(const uint8_t *labels, int B, int T, int max_runs, int *runs, int *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const uint8_t *p = labels + (size_t)b * T;
    int *r = runs + (size_t)b * max_runs * 2;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int n_start = 0, n_stop = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < T;
        const unsigned cur = in ? (p[t] & 1u) : 0u;
        const unsigned prev = (in && t > 0) ? (p[t - 1] & 1u) : 0u;
        const bool is_start = in && cur && !prev, is_stop = in && !cur && prev;
        const unsigned long long ms = __ballot(is_start), me = __ballot(is_stop);
        if (is_start) { const int i = n_start + __popcll(ms & below); if (i < max_runs) r[2 * i] = t; }
        if (is_stop) { const int i = n_stop + __popcll(me & below); if (i < max_runs) r[2 * i + 1] = t; }
        n_start += __popcll(ms);
        n_stop += __popcll(me);
    }
    if (lane == 0) {
        if (n_start > n_stop && n_stop < max_runs) r[2 * n_stop + 1] = T;   // run still open at the end of the row
        counts[b] = n_start;
    }
}

// busy-waits `ticks` of the constant 100 MHz counter (s_memrealtime), one wave: the stream-overlap probe's "long" kernel
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        t = __builtin_amdgcn_s_memrealtime();
    }
    if (sink && threadIdx.x == 0) *sink = t - t0;
}

}  // namespace

hipError_t launch_spin(unsigned long long ticks, unsigned long long *sink, hipStream_t s) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, sink);
    return hipGetLastError();
}

hipError_t launch_runs(const uint8_t *labels, int B, int T, int max_runs, int *runs, int *counts, hipStream_t s) {
    if (B <= 0 || T <= 0) return hipSuccess;
    hipLaunchKernelGGL(runs_kernel, dim3((B + 3) / 4), dim3(256), 0, s, labels, B, T, max_runs, runs, counts);
    return hipGetLastError();
}

hipError_t launch_der(const uint8_t *pred, const uint8_t *gt, int B, int T, uint32_t *counts, hipStream_t s) {
    if (B <= 0 || T <= 0) return hipSuccess;
    hipLaunchKernelGGL(der_kernel, dim3((B + 3) / 4), dim3(256), 0, s, pred, gt, B, T, counts);
    return hipGetLastError();
}

hipError_t launch_classifier(const ClsArgs &a, hipStream_t s) {
    const long long rows = (long long)a.tiles * a.T * SEQ_TILE;
    if (rows <= 0) return hipSuccess;
    const int grid = (int)((rows + 15) / 16);
    hipLaunchKernelGGL(classifier_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_untile(const void *src, const void *src_lo, int ld, int W, float *dst, int tiles, int T, int B, hipStream_t s) {
    const long long n4 = (long long)B * T * (W / 4);
    if (n4 <= 0) return hipSuccess;
    long long g = (n4 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(untile_kernel, dim3((int)g), dim3(256), 0, s, src, src_lo, ld, W, dst, tiles, T, B);
    return hipGetLastError();
}

hipError_t launch_median(const float *probs, int B, int T, int kernel, uint8_t *labels, hipStream_t s) {
    const long long n = (long long)B * T;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(median_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, probs, B, T, kernel / 2, labels);
    return hipGetLastError();
}

}  // namespace uvad

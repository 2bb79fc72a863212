// SSD training loss: reference ssd_loss.py:8-65 as ONE wavefront kernel per
// image -- Huber localisation loss over the positives, categorical cross-entropy on the
// renormalised / clipped probabilities, 3:1 hard-negative mining by descending loss rank, the
// per-image normalisation by the positive count -- and (optionally) the gradients of the
// batch-mean total loss w.r.t. the predicted deltas and the LOGITS feeding the softmax.
//
// No MFMA here: the path is HBM/latency bound (reads B*N*(2L+8)*4 bytes once).  The rank
// `argsort(argsort(masked_loss, DESCENDING)) < total_neg` (ssd_loss.py:54-57) is evaluated
// without sorting: a 4-pass 8-bit radix SELECT over the fp32 bit patterns (losses are >= 0, so
// the patterns are monotonic) finds the total_neg-th largest masked loss T; anchors above T are
// hard negatives, ties AT T are broken by ascending anchor index (tf.argsort DESCENDING orders
// equal keys by index: top_k semantics) through an ordered ballot/popcount prefix count.
#include "common.h"

namespace ssd {

constexpr int kLossThreads = 1024;
constexpr int kLossWaves = kLossThreads / 64;

struct LossParams {
    const float* yd;      // actual_deltas [B,N,4]           (nullable: skip the localisation term)
    const float* pd;      // pred_deltas   [B,N,4]
    const float* yl;      // actual_labels [B,N,L] (one-hot) (nullable: skip the confidence term)
    const float* pp;      // pred_labels   [B,N,L] probabilities
    int B, N, L;
    float neg_pos_ratio, loc_alpha;
    float* loc_loss;      // [B] (nullable)
    float* conf_loss;     // [B] (nullable)
    float* ce_out;        // [B,N] per-anchor cross-entropy (nullable)
    float* mask_out;      // [B,N] final_mask = pos + neg (nullable)
    float* grad_deltas;   // [B,N,4] (nullable)
    float* grad_logits;   // [B,N,L] (nullable)
    float grad_scale;     // d(total)/d(per-image loss): 1/batch for the Keras batch mean
    unsigned* keys;       // workspace [B,N]
    float* ce;            // workspace [B,N]
    unsigned char* flags; // workspace [B,N]: bit0 conf-positive, bit1 loc-positive
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// deterministic block sums: per-thread partials (index order), wave butterfly, waves in order
__device__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < kLossWaves; ++w) t += sh[w];
    return t;
}
__device__ int block_sum_i(int v, int* sh) {
    v = wave_sum_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < kLossWaves; ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(kLossThreads) void ssd_loss_kernel(const LossParams p) {
    __shared__ float shf[kLossWaves];
    __shared__ int shi[kLossWaves];
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_need;
    const int b = blockIdx.x, tid = threadIdx.x, N = p.N, L = p.L;
    const long base = (long)b * N;
    unsigned* keys = p.keys + base;
    float* ce = p.ce + base;
    unsigned char* flags = p.flags + base;

    // ---- phase A: per-anchor losses
    float loc_part = 0.f;
    int npos_conf = 0, npos_loc = 0;
    for (int n = tid; n < N; n += kLossThreads) {
        unsigned char fl = 0;
        if (p.yl) {
            const float* y = p.yl + (base + n) * L;
            const float* q = p.pp + (base + n) * L;
            float s = 0.f;
            for (int c = 0; c < L; ++c) s += q[c];
            float acc = 0.f;
            bool pos = false;
            for (int c = 0; c < L; ++c) {
                const float r = fminf(fmaxf(q[c] / s, 1e-7f), 1.0f - 1e-7f);
                acc += y[c] * logf(r);
                if (c > 0 && y[c] != 0.f) pos = true;
            }
            const float l = -acc;
            ce[n] = l;
            float m = l * y[0];
            if (m == 0.f) m = 0.f;                       // -0.0 ranks as 0.0
            keys[n] = __float_as_uint(m);
            if (pos) { fl |= 1; ++npos_conf; }
        }
        if (p.yd) {
            const float4 t = *reinterpret_cast<const float4*>(p.yd + (base + n) * 4);
            const float4 o = *reinterpret_cast<const float4*>(p.pd + (base + n) * 4);
            const float e[4] = {o.x - t.x, o.y - t.y, o.z - t.z, o.w - t.w};
            float h = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = fabsf(e[k]);
                const float qd = fminf(a, 1.0f);
                h += 0.5f * (qd * qd) + 1.0f * (a - qd);          // TF-2.0 huber_loss, delta = 1
            }
            if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) {
                fl |= 2;
                ++npos_loc;
                loc_part += h;
            }
        }
        flags[n] = fl;
    }
    const int pos_conf = block_sum_i(npos_conf, shi);
    const int pos_loc = block_sum_i(npos_loc, shi);
    const float loc_sum = block_sum(loc_part, shf);
    const float loc_den = pos_loc == 0 ? 1.0f : (float)pos_loc;
    if (p.yd && p.loc_loss && tid == 0) p.loc_loss[b] = loc_sum / loc_den * p.loc_alpha;

    float conf_den = 1.0f;
    if (p.yl) {
        // ---- phase B: radix select of the K-th largest masked loss
        const int K = (int)((float)pos_conf * p.neg_pos_ratio);       // tf.cast(total_pos * ratio, int32)
        unsigned prefix = 0, need = (unsigned)(K < 0 ? 0 : K);
        const bool all = K >= N, none = K <= 0;
        if (!all && !none) {
            for (int shift = 24; shift >= 0; shift -= 8) {
                for (int i = tid; i < 256; i += kLossThreads) hist[i] = 0;
                __syncthreads();
                const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
                for (int n = tid; n < N; n += kLossThreads) {
                    const unsigned k = keys[n];
                    if ((k & himask) == (prefix & himask)) atomicAdd(&hist[(k >> shift) & 255], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned cum = 0;
                    int bin = 255;
                    for (; bin > 0; --bin) {
                        if (cum + hist[bin] >= need) break;
                        cum += hist[bin];
                    }
                    sel_prefix = prefix | ((unsigned)bin << shift);
                    sel_need = need - cum;            // still to take among keys matching the new prefix
                }
                __syncthreads();
                prefix = sel_prefix;
                need = sel_need;
                __syncthreads();
            }
        }
        const unsigned T = prefix;      // K-th largest key; `need` of the keys == T are selected, lowest index first
        // ---- phase C: ordered selection + confidence sum
        float conf_part = 0.f;
        unsigned running = 0;           // keys == T seen in earlier chunks (same value in every thread)
        for (int n0 = 0; n0 < N; n0 += kLossThreads) {
            const int n = n0 + tid;
            const bool in = n < N;
            const unsigned k = in ? keys[n] : 0u;
            const bool eq = in && !all && !none && k == T;
            const unsigned long long bal = __ballot(eq);
            const int lane = tid & 63, wv = tid >> 6;
            __syncthreads();
            if (lane == 0) shi[wv] = __popcll(bal);
            __syncthreads();
            unsigned before = running;
            unsigned chunk_total = 0;
            for (int w = 0; w < kLossWaves; ++w) {
                if (w < wv) before += shi[w];
                chunk_total += shi[w];
            }
            before += __popcll(bal & ((1ull << lane) - 1ull));
            running += chunk_total;
            if (in) {
                const bool neg = all || (!none && (k > T || (eq && before < need)));
                const unsigned char fl = flags[n];
                const float fm = ((fl & 1) ? 1.0f : 0.0f) + (neg ? 1.0f : 0.0f);
                conf_part += fm * ce[n];
                if (p.mask_out) p.mask_out[base + n] = fm;
                if (p.ce_out) p.ce_out[base + n] = ce[n];
                keys[n] = __float_as_uint(fm);          // re-used by the gradient phase
            }
        }
        const float conf_sum = block_sum(conf_part, shf);
        conf_den = pos_conf == 0 ? 1.0f : (float)pos_conf;
        if (p.conf_loss && tid == 0) p.conf_loss[b] = conf_sum / conf_den;
    }

    // ---- phase D: gradients of grad_scale * (loc_loss[b] + conf_loss[b])
    if (p.grad_deltas && p.yd) {
        const float g = p.grad_scale * p.loc_alpha / loc_den;
        for (int n = tid; n < N; n += kLossThreads) {
            float4 r = {0.f, 0.f, 0.f, 0.f};
            if (flags[n] & 2) {
                const float4 t = *reinterpret_cast<const float4*>(p.yd + (base + n) * 4);
                const float4 o = *reinterpret_cast<const float4*>(p.pd + (base + n) * 4);
                r.x = g * fminf(fmaxf(o.x - t.x, -1.0f), 1.0f);
                r.y = g * fminf(fmaxf(o.y - t.y, -1.0f), 1.0f);
                r.z = g * fminf(fmaxf(o.z - t.z, -1.0f), 1.0f);
                r.w = g * fminf(fmaxf(o.w - t.w, -1.0f), 1.0f);
            }
            *reinterpret_cast<float4*>(p.grad_deltas + (base + n) * 4) = r;
        }
    }
    if (p.grad_logits && p.yl) {
        __syncthreads();
        for (int n = tid; n < N; n += kLossThreads) {
            const float gce = __uint_as_float(keys[n]) * p.grad_scale / conf_den;
            const float* y = p.yl + (base + n) * L;
            const float* q = p.pp + (base + n) * L;
            float* out = p.grad_logits + (base + n) * L;
            if (gce == 0.f) {
                for (int c = 0; c < L; ++c) out[c] = 0.f;
                continue;
            }
            float s = 0.f;
            for (int c = 0; c < L; ++c) s += q[c];
            // dL/dp_j = gce * (g_j / s - (sum_c g_c p_c) / s^2),  g_c = -y_c / r_c inside the clip range
            float t = 0.f;
            for (int c = 0; c < L; ++c) {
                const float qq = q[c] / s;
                if (y[c] != 0.f && qq >= 1e-7f && qq <= 1.0f - 1e-7f) t += (-y[c] / qq) * q[c];
            }
            float dot = 0.f;        // sum_j p_j dL/dp_j  (softmax backward)
            for (int c = 0; c < L; ++c) {
                const float qq = q[c] / s;
                const float gc = (y[c] != 0.f && qq >= 1e-7f && qq <= 1.0f - 1e-7f) ? -y[c] / qq : 0.f;
                dot += q[c] * (gce * (gc / s - t / (s * s)));
            }
            for (int c = 0; c < L; ++c) {
                const float qq = q[c] / s;
                const float gc = (y[c] != 0.f && qq >= 1e-7f && qq <= 1.0f - 1e-7f) ? -y[c] / qq : 0.f;
                const float dp = gce * (gc / s - t / (s * s));
                out[c] = q[c] * (dp - dot);
            }
        }
    }
}

}  // namespace ssd

using namespace ssd;

extern "C" {

size_t ssd_loss_workspace_bytes(int B, int N) {
    if (B < 0 || N < 0) return 0;
    return align_up((size_t)B * N * 4, 256) * 2 + align_up((size_t)B * N, 256);
}

int ssd_loss(const float* actual_deltas_dev, const float* pred_deltas_dev, const float* actual_labels_dev,
             const float* pred_labels_dev, int B, int N, int L, float neg_pos_ratio, float loc_loss_alpha,
             float* loc_loss_dev, float* conf_loss_dev, float* ce_out_dev, float* mask_out_dev,
             float* grad_deltas_dev, float* grad_logits_dev, float grad_scale, void* workspace_dev,
             size_t workspace_bytes, void* stream) {
    SSD_CHECK_ARG(B >= 0 && N >= 1 && L >= 1, "ssd_loss: bad sizes B=%d N=%d L=%d", B, N, L);
    if (B == 0) return SSD_OK;
    const bool loc = actual_deltas_dev != nullptr, conf = actual_labels_dev != nullptr;
    SSD_CHECK_ARG(loc || conf, "ssd_loss: neither a localisation nor a confidence target given");
    SSD_CHECK_ARG(!loc || pred_deltas_dev, "ssd_loss: pred_deltas is NULL");
    SSD_CHECK_ARG(!conf || pred_labels_dev, "ssd_loss: pred_labels is NULL");
    SSD_CHECK_ARG(!grad_deltas_dev || loc, "ssd_loss: grad_deltas needs the localisation target");
    SSD_CHECK_ARG(!grad_logits_dev || conf, "ssd_loss: grad_logits needs the confidence target");
    SSD_CHECK_ARG(workspace_dev && workspace_bytes >= ssd_loss_workspace_bytes(B, N),
                  "ssd_loss: workspace too small (%zu < %zu bytes)", workspace_bytes, ssd_loss_workspace_bytes(B, N));
    SSD_CHECK_ARG((((uintptr_t)actual_deltas_dev | (uintptr_t)pred_deltas_dev | (uintptr_t)grad_deltas_dev) & 15) == 0,
                  "ssd_loss: delta tensors must be 16-byte aligned");
    LossParams p{};
    p.yd = actual_deltas_dev; p.pd = pred_deltas_dev; p.yl = actual_labels_dev; p.pp = pred_labels_dev;
    p.B = B; p.N = N; p.L = L;
    p.neg_pos_ratio = neg_pos_ratio; p.loc_alpha = loc_loss_alpha;
    p.loc_loss = loc_loss_dev; p.conf_loss = conf_loss_dev; p.ce_out = ce_out_dev; p.mask_out = mask_out_dev;
    p.grad_deltas = grad_deltas_dev; p.grad_logits = grad_logits_dev; p.grad_scale = grad_scale;
    char* ws = (char*)workspace_dev;
    const size_t a = align_up((size_t)B * N * 4, 256);
    p.keys = (unsigned*)ws;
    p.ce = (float*)(ws + a);
    p.flags = (unsigned char*)(ws + 2 * a);
    hipLaunchKernelGGL(ssd_loss_kernel, dim3(B), dim3(kLossThreads), 0, (hipStream_t)stream, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // extern "C"

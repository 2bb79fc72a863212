// Box-math kernels for gfx950: prior boxes, decode, class-mask + threshold compaction,
// per-class greedy NMS (bitonic sort + ballot suppression), per-image top-K merge,
// pairwise IoU and target matching.  HBM/latency-bound integer + fp32 work: wavefront
// primitives (64-wide ballot / shuffle / popcount), LDS staging, no MFMA.
//
// Built with -ffp-contract=off: every product and sum is rounded separately, exactly as
// the reference's chain of elementwise TF ops does (utils/bbox_utils.py), so that anchor
// indices kept by NMS and match indices are bit-exact against the oracle.
#include "common.h"

namespace ssd {

// ------------------------------------------------------------------ shared device math
// utils/bbox_utils.py:61-85 (+ models/decoder.py:41 variance scaling when USE_VAR).
__device__ __forceinline__ float4 decode_box(const float4 p, float4 d, const float4 var,
                                             const bool use_var) {
    if (use_var) { d.x = d.x * var.x; d.y = d.y * var.y; d.z = d.z * var.z; d.w = d.w * var.w; }
    const float pw = p.w - p.y;
    const float ph = p.z - p.x;
    const float pcx = p.y + 0.5f * pw;
    const float pcy = p.x + 0.5f * ph;
    const float w = expf(d.w) * pw;
    const float h = expf(d.z) * ph;
    const float cx = (d.y * pw) + pcx;
    const float cy = (d.x * ph) + pcy;
    const float y1 = cy - (0.5f * h);
    const float x1 = cx - (0.5f * w);
    return make_float4(y1, x1, h + y1, w + x1);
}

// [3P] TF CombinedNonMaxSuppression IOU helper, boxes unclipped.
__device__ __forceinline__ float nms_iou(const float4 a, const float4 b) {
    const float ymin_i = fminf(a.x, a.z), ymax_i = fmaxf(a.x, a.z);
    const float xmin_i = fminf(a.y, a.w), xmax_i = fmaxf(a.y, a.w);
    const float ymin_j = fminf(b.x, b.z), ymax_j = fmaxf(b.x, b.z);
    const float xmin_j = fminf(b.y, b.w), xmax_j = fmaxf(b.y, b.w);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    const float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    const float inter = iy * ix;
    return ((inter) / (area_i + area_j - inter));
}

// utils/bbox_utils.py:44-59 for one (box, gt) pair.
__device__ __forceinline__ float pair_iou(const float4 p, const float4 g) {
    const float garea = (g.z - g.x) * (g.w - g.y);
    const float parea = (p.z - p.x) * (p.w - p.y);
    const float x_top = fmaxf(p.y, g.y), y_top = fmaxf(p.x, g.x);
    const float x_bot = fminf(p.w, g.w), y_bot = fminf(p.z, g.z);
    const float inter = fmaxf(x_bot - x_top, 0.0f) * fmaxf(y_bot - y_top, 0.0f);
    const float uni = parea + garea - inter;
    return ((inter) / (uni));
}

// Sort keys: ascending u64 order == (score desc, anchor asc[, class asc]).
__device__ __forceinline__ unsigned score_desc_key(float s) {
    const unsigned u = __float_as_uint(s);
    const unsigned asc = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float score_from_key(unsigned k) {
    const unsigned asc = ~k;
    const unsigned u = (asc & 0x80000000u) ? (asc ^ 0x80000000u) : ~asc;
    return __uint_as_float(u);
}

constexpr int kIdxBits = 22;    // N <= 4,194,304 anchors
constexpr int kClsBits = 10;    // L <= 1,024 classes

// ------------------------------------------------------------------ prior boxes (A1-A3)
constexpr int kMaxLevels = 16;
constexpr int kMaxArs = 15;
struct PriorCfg {
    int levels;
    int f[kMaxLevels];
    int a[kMaxLevels];          // anchors per cell = n_ars + 1
    int offset[kMaxLevels + 1];
    double cur[kMaxLevels];     // A1 scale s_k   (host float64, utils/bbox_utils.py:124)
    double nxt[kMaxLevels];     // A1 scale s_k+1
    float ar[kMaxLevels][kMaxArs];
};

__global__ void priors_kernel(const PriorCfg cfg, float4* __restrict__ out, int total) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int l = 0;
    while (l + 1 < cfg.levels && gid >= cfg.offset[l + 1]) ++l;
    const int local = gid - cfg.offset[l];
    const int A = cfg.a[l], f = cfg.f[l];
    const int a = local % A;
    const int cell = local / A;
    const int y = cell / f, x = cell % f;
    // utils/bbox_utils.py:139-146
    float h, w;
    if (a < A - 1) {
        const float s = sqrtf(cfg.ar[l][a]);
        h = (((float)cfg.cur[l]) / (s));
        w = (float)cfg.cur[l] * s;
    } else {
        h = w = sqrtf((float)(cfg.cur[l] * cfg.nxt[l]));
    }
    // utils/bbox_utils.py:164-165: int32 / int -> float64, + stride/2 in float64, cast.
    const double stride = 1.0 / (double)f;
    const float gy = (float)((double)y / (double)f + stride / 2.0);
    const float gx = (float)((double)x / (double)f + stride / 2.0);
    float4 r;
    r.x = (-h / 2.0f) + gy;
    r.y = (-w / 2.0f) + gx;
    r.z = (h / 2.0f) + gy;
    r.w = (w / 2.0f) + gx;
    // :176 clip_by_value(0, 1)
    r.x = fminf(fmaxf(r.x, 0.0f), 1.0f);
    r.y = fminf(fmaxf(r.y, 0.0f), 1.0f);
    r.z = fminf(fmaxf(r.z, 0.0f), 1.0f);
    r.w = fminf(fmaxf(r.w, 0.0f), 1.0f);
    out[gid] = r;
}

// ------------------------------------------------------------------ decode (D1)
__global__ void decode_kernel(const float4* __restrict__ priors, const float4* __restrict__ deltas,
                              const float4 var, const int use_var, const int N, const long total,
                              float4* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x)
        out[i] = decode_box(priors[i % N], deltas[i], var, use_var != 0);
}

// ------------------------------------------------- decode + class mask + compaction (D3)
// One block = 256 consecutive anchors of one image.  The [256, L] probability slab is
// contiguous in HBM: it is staged through LDS with coalesced loads, then each lane walks
// its own row (row stride L words: conflict-free for odd L such as 21).
// Candidates (score > thr, strict) are appended to the per-(image,class) list with one
// atomic each; the later sort makes the result independent of the append order.
// softmax of one row in place (models/header.py:66 `Activation("softmax")`): THE definition of the net's class
// probabilities -- softmax_kernel (ssd_net_forward's output) and the fused compact kernel of ssd_net_predict both call
// it from this translation unit, so the two paths produce the same bits
__device__ __forceinline__ void softmax_row(float* row, const int L) {
    float mx = row[0];
    for (int c = 1; c < L; ++c) mx = fmaxf(mx, row[c]);
    float s = 0.f;
    for (int c = 0; c < L; ++c) {
        const float ev = expf(row[c] - mx);
        row[c] = ev;
        s += ev;
    }
    for (int c = 0; c < L; ++c) row[c] = row[c] / s;
}

// 256 rows per block staged through LDS (coalesced in/out); one lane per row.
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ in, const long rows,
                                                      const int L, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const long r0 = (long)blockIdx.x * 256;
    const int nrows = (int)min((long)256, rows - r0);
    const int n = nrows * L;
    const float* src = in + r0 * L;
    for (int e = threadIdx.x; e < n; e += 256) tile[e] = src[e];
    __syncthreads();
    if ((int)threadIdx.x < nrows) softmax_row(tile + threadIdx.x * L, L);
    __syncthreads();
    float* dst = out + r0 * L;
    for (int e = threadIdx.x; e < n; e += 256) dst[e] = tile[e];
}

// LOGITS (ssd_net_predict): `probs` holds the head convs' LOGITS; the softmax runs on the LDS-staged slab right here
// (no separate softmax pass over the [B, N, L] buffer, no launch); needs use_lds.
template <bool DECODE, bool LOGITS>
__global__ __launch_bounds__(256) void compact_kernel(
    const float4* __restrict__ deltas, const float* __restrict__ probs,
    const float4* __restrict__ priors, const float4 var, const int N, const int L,
    const float score_thr, const int use_lds, float4* __restrict__ boxes_out,
    unsigned long long* __restrict__ cand_keys, int* __restrict__ cand_count, const int cap) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * 256;
    const int rows = min(256, N - i0);
    const float* src = probs + ((size_t)b * N + i0) * L;
    if (use_lds) {
        const int n = rows * L;
        for (int e = threadIdx.x; e < n; e += 256) tile[e] = src[e];
        __syncthreads();
    }
    const int t = threadIdx.x;
    if (t >= rows) return;
    const int i = i0 + t;
    if (LOGITS) softmax_row(tile + t * L, L);
    const float* row = use_lds ? (tile + t * L) : (src + (size_t)t * L);
    bool masked = false;
    if (DECODE) {
        // models/decoder.py:44-45: argmax (first max wins) == 0 -> zero the whole row.
        int am = 0;
        float best = row[0];
        for (int c = 1; c < L; ++c) {
            const float v = row[c];
            if (v > best) { best = v; am = c; }
        }
        masked = (am == 0);
    }
    bool any = false;
    for (int c = 0; c < L; ++c) {
        const float s = masked ? 0.0f : row[c];
        if (s > score_thr) {
            const int slot = atomicAdd(&cand_count[b * L + c], 1);
            if (slot < cap)
                cand_keys[((size_t)b * L + c) * cap + slot] =
                    ((unsigned long long)score_desc_key(s) << 32) | (unsigned)i;
            any = true;
        }
    }
    if (DECODE && any)
        boxes_out[(size_t)b * N + i] = decode_box(priors[i], deltas[(size_t)b * N + i], var, true);
}

// ------------------------------------------------------------------ block-wide sort
// "Normalized" bitonic network (every merge ascending; first sub-step mirrored), which
// sorts any n in place with no padding: a comparator whose upper index is >= n is a no-op.
// 256 threads; keys may live in LDS or in global memory (big-n fallback).
template <typename K>
__device__ void block_sort_asc(K* keys, const int n) {
    if (n < 2) { __syncthreads(); return; }
    int P = 2;
    while (P < n) P <<= 1;
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        const int hk = k >> 1;
        for (int t = threadIdx.x; t < half; t += blockDim.x) {
            const int blk = t / hk, pos = t - blk * hk;
            const int i = blk * k + pos;
            const int l = blk * k + (k - 1 - pos);
            if (l < n) {
                const K a = keys[i], c = keys[l];
                if (a > c) { keys[i] = c; keys[l] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                if (l < n) {
                    const K a = keys[i], c = keys[l];
                    if (a > c) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ per-class greedy NMS
// One 256-thread block per (image, class).  Sorted candidates are consumed in chunks of
// 256: (1) every thread tests its candidate against the boxes already kept (LDS list);
// (2) the survivors are compacted with ballot + popcount prefix; (3) wave 0 resolves the
// survivors in score order: the lowest set bit of the 64-wide alive ballot is kept, its
// box is broadcast by readlane and the remaining lanes clear their alive bit if IoU > thr.
constexpr int kSortLds = 4096;   // candidates sorted in LDS (32 KB); above: in place in HBM

__global__ __launch_bounds__(256) void nms_class_kernel(
    unsigned long long* __restrict__ cand_keys, int* __restrict__ cand_count,
    const float4* __restrict__ boxes, const int N, const int L, const int cap, const int maxk,
    const float iou_thr, unsigned long long* __restrict__ kept_keys, int* __restrict__ kept_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* kept_box = reinterpret_cast<float4*>(smem);                       // [maxk]
    float4* surv_box = kept_box + maxk;                                       // [256]
    unsigned long long* surv_key = reinterpret_cast<unsigned long long*>(surv_box + 256);   // [256]
    unsigned long long* skeys = surv_key + 256;                               // [kSortLds]
    int* s_misc = reinterpret_cast<int*>(skeys + kSortLds);                   // [8]

    const int bc = blockIdx.x;          // b * L + c
    const int b = bc / L, c = bc - b * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(cand_count[bc], cap);
    // every thread has read the count: its (image, class) owner resets it, so the NEXT call on this workspace starts from
    // zero without a memset launch (the net's predict path; a caller-owned workspace is still zeroed by the launcher)
    __syncthreads();
    if (tid == 0) cand_count[bc] = 0;
    if (n == 0) {
        if (tid == 0) kept_count[bc] = 0;
        return;
    }
    unsigned long long* gkeys = cand_keys + (size_t)bc * cap;
    unsigned long long* keys;
    if (n <= kSortLds) {
        for (int i = tid; i < n; i += 256) skeys[i] = gkeys[i];
        __syncthreads();
        keys = skeys;
    } else {
        keys = gkeys;
    }
    block_sort_asc(keys, n);

    const float4* img_boxes = boxes + (size_t)b * N;
    unsigned long long* out_keys = kept_keys + (size_t)bc * maxk;
    int kept = 0;
    for (int base = 0; base < n && kept < maxk; base += 256) {
        const int j = base + tid;
        bool alive = j < n;
        unsigned long long key = 0;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (alive) {
            key = keys[j];
            box = img_boxes[(unsigned)(key & 0xffffffffu)];
            for (int q = kept - 1; q >= 0; --q)
                if (nms_iou(box, kept_box[q]) > iou_thr) { alive = false; break; }
        }
        const unsigned long long m = __ballot(alive);
        if (lane == 0) s_misc[wave] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_misc[w];
        const int nsurv = s_misc[0] + s_misc[1] + s_misc[2] + s_misc[3];
        if (alive) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            surv_box[pos] = box;
            surv_key[pos] = key;
        }
        __syncthreads();
        if (wave == 0) {
            const int chunk_start = kept;
            for (int sb = 0; sb < nsurv && kept < maxk; sb += 64) {
                const int p = sb + lane;
                bool a = p < nsurv;
                float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned long long ky = 0;
                if (a) {
                    bx = surv_box[p];
                    ky = surv_key[p];
                    for (int q = kept - 1; q >= chunk_start; --q)
                        if (nms_iou(bx, kept_box[q]) > iou_thr) { a = false; break; }
                }
                unsigned long long mask = __ballot(a);
                while (mask != 0ull && kept < maxk) {
                    const int jl = __ffsll((long long)mask) - 1;
                    float4 kb;
                    kb.x = __shfl(bx.x, jl);
                    kb.y = __shfl(bx.y, jl);
                    kb.z = __shfl(bx.z, jl);
                    kb.w = __shfl(bx.w, jl);
                    if (lane == jl) {
                        kept_box[kept] = bx;
                        // merge key: [score desc : 32][anchor : 22][class : 10]
                        out_keys[kept] = (ky & 0xffffffff00000000ull) |
                                         ((ky & 0xffffffffull) << kClsBits) | (unsigned)c;
                    }
                    ++kept;
                    if (a && lane > jl && nms_iou(bx, kb) > iou_thr) a = false;
                    mask = __ballot(a) & ~((2ull << jl) - 1ull);
                }
            }
            if (lane == 0) s_misc[4] = kept;
        }
        __syncthreads();
        kept = s_misc[4];
        __syncthreads();
    }
    if (tid == 0) kept_count[bc] = kept;
}

// ------------------------------------------------------------------ per-image top-K merge
// One block per image: gather the kept (score, anchor, class) keys of all classes, sort,
// emit the first max_total rows (boxes clipped to [0,1] when clip), zero-pad the rest.
__global__ __launch_bounds__(256) void merge_topk_kernel(
    const unsigned long long* __restrict__ kept_keys, const int* __restrict__ kept_count,
    const float4* __restrict__ boxes, const int N, const int L, const int maxk,
    const int max_total, const int clip, const int use_lds,
    unsigned long long* __restrict__ merge_ws, float4* __restrict__ out_boxes,
    float* __restrict__ out_labels, float* __restrict__ out_scores, int* __restrict__ out_valid,
    int* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s_off = reinterpret_cast<int*>(smem);                                  // [L + 1]
    unsigned long long* lkeys =
        reinterpret_cast<unsigned long long*>(smem + ((size_t)(L + 1) * 4 + 15) / 16 * 16);
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        int acc = 0;
        for (int c = 0; c < L; ++c) { s_off[c] = acc; acc += kept_count[b * L + c]; }
        s_off[L] = acc;
    }
    __syncthreads();
    const int total = s_off[L];
    unsigned long long* keys = use_lds ? lkeys : (merge_ws + (size_t)b * L * maxk);
    for (int c = 0; c < L; ++c) {
        const int cnt = s_off[c + 1] - s_off[c];
        const unsigned long long* src = kept_keys + ((size_t)b * L + c) * maxk;
        for (int i = tid; i < cnt; i += 256) keys[s_off[c] + i] = src[i];
    }
    __syncthreads();
    block_sort_asc(keys, total);
    const int nv = min(total, max_total);
    if (tid == 0) out_valid[b] = nv;
    for (int r = tid; r < max_total; r += 256) {
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float lab = 0.f, sc = 0.f;
        int idx = -1;
        if (r < nv) {
            const unsigned long long k = keys[r];
            const unsigned lo = (unsigned)(k & 0xffffffffull);
            idx = (int)(lo >> kClsBits);
            lab = (float)(lo & ((1u << kClsBits) - 1u));
            sc = score_from_key((unsigned)(k >> 32));
            bx = boxes[(size_t)b * N + idx];
            if (clip) {
                bx.x = fminf(fmaxf(bx.x, 0.0f), 1.0f);
                bx.y = fminf(fmaxf(bx.y, 0.0f), 1.0f);
                bx.z = fminf(fmaxf(bx.z, 0.0f), 1.0f);
                bx.w = fminf(fmaxf(bx.w, 0.0f), 1.0f);
            }
        }
        out_boxes[(size_t)b * max_total + r] = bx;
        out_labels[(size_t)b * max_total + r] = lab;
        out_scores[(size_t)b * max_total + r] = sc;
        if (out_idx) out_idx[(size_t)b * max_total + r] = idx;
    }
}

// ------------------------------------------------------------------ IoU map (M1)
__global__ void iou_map_kernel(const float4* __restrict__ boxes, const int boxes_batched,
                               const float4* __restrict__ gt, const int N, const int G,
                               const long total, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % G);
        const long bi = e / G;
        const int i = (int)(bi % N);
        const long b = bi / N;
        const float4 p = boxes_batched ? boxes[b * N + i] : boxes[i];
        out[e] = pair_iou(p, gt[b * G + g]);
    }
}

// ------------------------------------------------------------------ encode (M3)
__device__ __forceinline__ float4 encode_box(const float4 p, const float4 g) {
    // utils/bbox_utils.py:96-113
    float bw = p.w - p.y, bh = p.z - p.x;
    const float bcx = p.y + 0.5f * bw, bcy = p.x + 0.5f * bh;
    const float gw = g.w - g.y, gh = g.z - g.x;
    const float gcx = g.y + 0.5f * gw, gcy = g.x + 0.5f * gh;
    if (bw == 0.0f) bw = 1e-3f;
    if (bh == 0.0f) bh = 1e-3f;
    const float dx = gw == 0.0f ? 0.0f : ((gcx - bcx) / (bw));
    const float dy = gh == 0.0f ? 0.0f : ((gcy - bcy) / (bh));
    const float dw = gw == 0.0f ? 0.0f : logf(((gw) / (bw)));
    const float dh = gh == 0.0f ? 0.0f : logf(((gh) / (bh)));
    return make_float4(dy, dx, dh, dw);
}

__global__ void encode_kernel(const float4* __restrict__ bboxes, const float4* __restrict__ gt,
                              const int N, const long total, float4* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x)
        out[i] = encode_box(bboxes[i % N], gt[i]);
}

// ------------------------------------------------------------------ match + encode (M2)
// One thread per (image, prior): loop over the (small) padded GT list staged in LDS.
__global__ __launch_bounds__(256) void match_encode_kernel(
    const float4* __restrict__ priors, const float4* __restrict__ gt, const int* __restrict__ gt_labels,
    const float4 var, const float iou_thr, const int N, const int G, const int L,
    float4* __restrict__ deltas, int* __restrict__ label_idx, int* __restrict__ match_idx,
    float* __restrict__ onehot) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* s_gt = reinterpret_cast<float4*>(smem);
    int* s_lab = reinterpret_cast<int*>(s_gt + G);
    const int b = blockIdx.y;
    for (int g = threadIdx.x; g < G; g += 256) {
        s_gt[g] = gt[(size_t)b * G + g];
        s_lab[g] = gt_labels[(size_t)b * G + g];
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 p = priors[i];
    // [3P] tf.argmax / reduce_max (Eigen reducers: accumulator starts at lowest(), replaced only
    // by a strictly greater value): first max wins and a NaN IoU (0/0: degenerate prior against a
    // padded ground-truth box) is never selected
    int am = 0;
    float best = G > 0 ? -3.402823466e38f : 0.0f;
    for (int g = 0; g < G; ++g) {
        const float v = pair_iou(p, s_gt[g]);
        if (v > best) { best = v; am = g; }
    }
    const bool pos = best > iou_thr;              // strict (utils/train_utils.py:117)
    float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
    int lab = 0;
    if (pos) { gb = s_gt[am]; lab = s_lab[am]; }
    float4 d = encode_box(p, gb);
    d.x = ((d.x) / (var.x));
    d.y = ((d.y) / (var.y));
    d.z = ((d.z) / (var.z));
    d.w = ((d.w) / (var.w));
    const size_t o = (size_t)b * N + i;
    deltas[o] = d;
    label_idx[o] = lab;
    match_idx[o] = am;
    if (onehot) {
        float* oh = onehot + o * L;
        for (int c = 0; c < L; ++c) oh[c] = (c == lab) ? 1.0f : 0.0f;
    }
}

// ------------------------------------------------------------------ host side
struct NmsWs {
    int* cand_count;
    int* kept_count;
    unsigned long long* cand_keys;
    unsigned long long* kept_keys;
    unsigned long long* merge_ws;
    float4* boxes;
    size_t bytes;
};

static NmsWs carve_ws(void* base, int B, int N, int L, int maxk) {
    NmsWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? (void*)((char*)base + off) : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    w.cand_count = (int*)take((size_t)B * L * 4);
    w.kept_count = (int*)take((size_t)B * L * 4);
    w.cand_keys = (unsigned long long*)take((size_t)B * L * N * 8);
    w.kept_keys = (unsigned long long*)take((size_t)B * L * maxk * 8);
    w.merge_ws = (unsigned long long*)take((size_t)B * L * maxk * 8);
    w.boxes = (float4*)take((size_t)B * N * 16);
    w.bytes = off;
    return w;
}

static int check_nms_args(int B, int N, int L, int max_per_class, int max_total) {
    SSD_CHECK_ARG(B >= 0 && N >= 0 && L >= 1, "decode_nms: bad sizes B=%d N=%d L=%d", B, N, L);
    SSD_CHECK_ARG(max_per_class >= 1 && max_total >= 1, "decode_nms: max_per_class=%d max_total=%d",
                  max_per_class, max_total);
    SSD_UNSUPPORTED_IF(N >= (1 << kIdxBits), "decode_nms: N=%d exceeds %d anchors", N, 1 << kIdxBits);
    SSD_UNSUPPORTED_IF(L > (1 << kClsBits), "decode_nms: L=%d exceeds %d classes", L, 1 << kClsBits);
    return SSD_OK;
}

// Shared driver: DECODE=true -> SSDDecoder path; false -> raw combined NMS.
// flags (the net's own predict path): kNmsLogits -- `scores` holds logits, softmax fused into the compaction;
// kNmsCountsClean -- the workspace's candidate counters are known to be zero (zeroed at allocation, re-zeroed by
// nms_class_kernel of the previous call): no memset launch
enum { kNmsLogits = 1, kNmsCountsClean = 2 };
static int run_nms(bool decode, const float* deltas, const float* scores, const float* priors_or_boxes,
                   const float* var, int B, int N, int L, int max_per_class, int max_total,
                   float iou_thr, float score_thr, int clip, float* boxes_out, float* labels_out,
                   float* scores_out, int* valid_out, int* kept_idx, void* ws, size_t ws_bytes,
                   hipStream_t st, int flags = 0, int ws_batch = 0) {
    int rc = check_nms_args(B, N, L, max_per_class, max_total);
    if (rc) return rc;
    if (B == 0) return SSD_OK;
    if (N == 0) {   // nothing to select: all-zero outputs
        SSD_HIP(hipMemsetAsync(boxes_out, 0, (size_t)B * max_total * 16, st));
        SSD_HIP(hipMemsetAsync(labels_out, 0, (size_t)B * max_total * 4, st));
        SSD_HIP(hipMemsetAsync(scores_out, 0, (size_t)B * max_total * 4, st));
        SSD_HIP(hipMemsetAsync(valid_out, 0, (size_t)B * 4, st));
        if (kept_idx) SSD_HIP(hipMemsetAsync(kept_idx, 0xff, (size_t)B * max_total * 4, st));
        return SSD_OK;
    }
    const int maxk = max_per_class < N ? max_per_class : N;
    const size_t nms_lds = (size_t)maxk * 16 + 256 * 16 + 256 * 8 + (size_t)kSortLds * 8 + 64;
    SSD_UNSUPPORTED_IF(nms_lds > 160 * 1024, "decode_nms: max_per_class=%d needs %zu B of LDS",
                       max_per_class, nms_lds);
    SSD_CHECK_ARG(ws != nullptr, "decode_nms: workspace is NULL");
    // the workspace is carved for ws_batch images when given (>= B): with kNmsCountsClean the counters have to sit at the
    // same place whatever batch a call runs (a layout carved for a smaller B would put its kept counts where a larger
    // batch's candidate counters live)
    SSD_CHECK_ARG(ws_batch == 0 || ws_batch >= B, "decode_nms: workspace carved for %d images, batch %d", ws_batch, B);
    NmsWs w = carve_ws(ws, ws_batch ? ws_batch : B, N, L, maxk);
    SSD_CHECK_ARG(ws_bytes >= w.bytes, "decode_nms: workspace %zu < required %zu", ws_bytes, w.bytes);
    SSD_CHECK_ARG(((uintptr_t)ws & 15) == 0, "decode_nms: workspace must be 16-byte aligned");

    if (!(flags & kNmsCountsClean)) SSD_HIP(hipMemsetAsync(w.cand_count, 0, (size_t)B * L * 4, st));
    const float4 v4 = var ? make_float4(var[0], var[1], var[2], var[3]) : make_float4(1, 1, 1, 1);
    const size_t tile_bytes = (size_t)256 * L * 4;
    const int use_lds = tile_bytes <= 96 * 1024;
    SSD_CHECK_ARG(!(flags & kNmsLogits) || (decode && use_lds), "decode_nms: fused softmax needs the LDS-staged decoder path");
    if (tile_bytes > 64 * 1024 && use_lds) {
        SSD_HIP(hipFuncSetAttribute((const void*)compact_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
        SSD_HIP(hipFuncSetAttribute((const void*)compact_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
        SSD_HIP(hipFuncSetAttribute((const void*)compact_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes));
    }
    dim3 grid(cdiv(N, 256), B);
    if (decode && (flags & kNmsLogits)) {
        hipLaunchKernelGGL((compact_kernel<true, true>), grid, dim3(256), tile_bytes, st,
                           (const float4*)deltas, scores, (const float4*)priors_or_boxes, v4, N, L,
                           score_thr, use_lds, w.boxes, w.cand_keys, w.cand_count, N);
    } else if (decode) {
        hipLaunchKernelGGL((compact_kernel<true, false>), grid, dim3(256), use_lds ? tile_bytes : 0, st,
                           (const float4*)deltas, scores, (const float4*)priors_or_boxes, v4, N, L,
                           score_thr, use_lds, w.boxes, w.cand_keys, w.cand_count, N);
    } else {
        hipLaunchKernelGGL((compact_kernel<false, false>), grid, dim3(256), use_lds ? tile_bytes : 0, st,
                           (const float4*)nullptr, scores, (const float4*)nullptr, v4, N, L,
                           score_thr, use_lds, (float4*)nullptr, w.cand_keys, w.cand_count, N);
    }
    SSD_LAUNCH_CHECK();
    const float4* nms_boxes = decode ? w.boxes : (const float4*)priors_or_boxes;
    if (nms_lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)nms_class_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)nms_lds));
    hipLaunchKernelGGL(nms_class_kernel, dim3(B * L), dim3(256), nms_lds, st, w.cand_keys,
                       w.cand_count, nms_boxes, N, L, N, maxk, iou_thr, w.kept_keys, w.kept_count);
    SSD_LAUNCH_CHECK();
    const size_t merge_keys = (size_t)L * maxk * 8;
    const size_t merge_hdr = align_up((size_t)(L + 1) * 4, 16);
    const int merge_lds_ok = merge_hdr + merge_keys <= 64 * 1024;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(B), dim3(256),
                       merge_hdr + (merge_lds_ok ? merge_keys : 0), st, w.kept_keys, w.kept_count,
                       nms_boxes, N, L, maxk, max_total, clip, merge_lds_ok, w.merge_ws,
                       (float4*)boxes_out, labels_out, scores_out, valid_out, kept_idx);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// The net's own predict path (csrc/ssd_net.hip): head LOGITS in, softmax fused into the compaction, no counter memset.
// `ws` is the net's workspace, zeroed at allocation.  Returns SSD_E_UNSUPPORTED when the fused form cannot run (L too
// large for the LDS-staged slab): the caller then runs the softmax layer + ssd_decode_nms.
bool decode_nms_fused_ok(int L) { return (size_t)256 * L * 4 <= 96 * 1024; }
int decode_nms_fused(const float* deltas, const float* logits, const float* priors, const float* var, int B, int N, int L,
                     int max_per_class, int max_total, float iou_thr, float score_thr, float* boxes, float* labels,
                     float* scores, int* valid, void* ws, size_t ws_bytes, int ws_batch, hipStream_t st) {
    return run_nms(true, deltas, logits, priors, var, B, N, L, max_per_class, max_total, iou_thr, score_thr, 1, boxes,
                   labels, scores, valid, nullptr, ws, ws_bytes, st, kNmsLogits | kNmsCountsClean, ws_batch);
}
int launch_softmax_lds(const float* in, long rows, int L, float* out, hipStream_t st) {
    hipLaunchKernelGGL(softmax_kernel, dim3(cdiv(rows, 256)), dim3(256), (size_t)256 * L * 4, st, in, rows, L, out);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

using namespace ssd;

extern "C" {

int ssd_priors_count(const int* fmaps, const int* n_ars, int levels) {
    if (!fmaps || !n_ars || levels < 0) return SSD_E_INVALID;
    long n = 0;
    for (int l = 0; l < levels; ++l) n += (long)fmaps[l] * fmaps[l] * (n_ars[l] + 1);
    return (int)n;
}

int ssd_priors(const int* fmaps, const float* const* ars, const int* n_ars, int levels,
               float* out_dev, void* stream) {
    SSD_CHECK_ARG(fmaps && ars && n_ars && out_dev, "ssd_priors: NULL argument");
    SSD_CHECK_ARG(levels >= 1 && levels <= kMaxLevels, "ssd_priors: levels=%d (1..%d)", levels, kMaxLevels);
    PriorCfg cfg;
    cfg.levels = levels;
    int off = 0;
    for (int l = 0; l < levels; ++l) {
        SSD_CHECK_ARG(fmaps[l] >= 1, "ssd_priors: feature map %d has size %d", l, fmaps[l]);
        SSD_CHECK_ARG(n_ars[l] >= 0 && n_ars[l] <= kMaxArs, "ssd_priors: level %d has %d ratios", l, n_ars[l]);
        cfg.f[l] = fmaps[l];
        cfg.a[l] = n_ars[l] + 1;
        cfg.offset[l] = off;
        off += fmaps[l] * fmaps[l] * cfg.a[l];
        // A1 (utils/bbox_utils.py:124), Python float64: 0.2 + (0.7/(m-1))*(k-1), k = l+1
        const double step = (0.9 - 0.2) / (double)(levels - 1);
        cfg.cur[l] = 0.2 + step * (double)((l + 1) - 1);
        cfg.nxt[l] = 0.2 + step * (double)((l + 2) - 1);
        for (int a = 0; a < n_ars[l]; ++a) cfg.ar[l][a] = ars[l][a];
    }
    cfg.offset[levels] = off;
    if (off == 0) return SSD_OK;
    hipLaunchKernelGGL(priors_kernel, dim3(cdiv(off, 256)), dim3(256), 0, (hipStream_t)stream, cfg,
                       (float4*)out_dev, off);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int ssd_decode_boxes(const float* priors_dev, const float* deltas_dev, const float* var, int B, int N,
                     float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && N >= 0, "ssd_decode_boxes: B=%d N=%d", B, N);
    const long total = (long)B * N;
    if (total == 0) return SSD_OK;
    SSD_CHECK_ARG(priors_dev && deltas_dev && out_dev, "ssd_decode_boxes: NULL pointer");
    const float4 v4 = var ? make_float4(var[0], var[1], var[2], var[3]) : make_float4(1, 1, 1, 1);
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)priors_dev, (const float4*)deltas_dev, v4, var ? 1 : 0, N, total,
                       (float4*)out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

size_t ssd_decode_nms_workspace_bytes(int B, int N, int L, int max_per_class) {
    if (B <= 0 || N <= 0 || L <= 0 || max_per_class <= 0) return 256;
    const int maxk = max_per_class < N ? max_per_class : N;
    return carve_ws(nullptr, B, N, L, maxk).bytes;
}

int ssd_decode_nms(const float* deltas_dev, const float* probs_dev, const float* priors_dev,
                   const float* var, int B, int N, int L, int max_per_class, int max_total,
                   float iou_thr, float score_thr, float* boxes_dev, float* labels_dev,
                   float* scores_dev, int* valid_dev, int* kept_idx_dev, void* workspace_dev,
                   size_t workspace_bytes, void* stream) {
    SSD_CHECK_ARG(var != nullptr, "ssd_decode_nms: variances pointer is NULL");
    SSD_CHECK_ARG(B == 0 || (boxes_dev && labels_dev && scores_dev && valid_dev),
                  "ssd_decode_nms: NULL output pointer");
    SSD_CHECK_ARG(B == 0 || N == 0 || (deltas_dev && probs_dev && priors_dev),
                  "ssd_decode_nms: NULL input pointer");
    return run_nms(true, deltas_dev, probs_dev, priors_dev, var, B, N, L, max_per_class, max_total,
                   iou_thr, score_thr, 1, boxes_dev, labels_dev, scores_dev, valid_dev, kept_idx_dev,
                   workspace_dev, workspace_bytes, (hipStream_t)stream);
}

int ssd_combined_nms(const float* boxes_dev, const float* scores_dev, int B, int N, int C,
                     int max_per_class, int max_total, float iou_thr, float score_thr, int clip_boxes,
                     float* boxes_out_dev, float* scores_out_dev, float* classes_out_dev,
                     int* valid_dev, int* kept_idx_dev, void* workspace_dev, size_t workspace_bytes,
                     void* stream) {
    SSD_CHECK_ARG(B == 0 || (boxes_out_dev && scores_out_dev && classes_out_dev && valid_dev),
                  "ssd_combined_nms: NULL output pointer");
    SSD_CHECK_ARG(B == 0 || N == 0 || (boxes_dev && scores_dev), "ssd_combined_nms: NULL input pointer");
    return run_nms(false, nullptr, scores_dev, boxes_dev, nullptr, B, N, C, max_per_class, max_total,
                   iou_thr, score_thr, clip_boxes, boxes_out_dev, classes_out_dev, scores_out_dev,
                   valid_dev, kept_idx_dev, workspace_dev, workspace_bytes, (hipStream_t)stream);
}

int ssd_iou_map(const float* boxes_dev, int boxes_batched, const float* gt_dev, int B, int N, int G,
                float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && N >= 0 && G >= 0, "ssd_iou_map: B=%d N=%d G=%d", B, N, G);
    const long total = (long)B * N * G;
    if (total == 0) return SSD_OK;
    SSD_CHECK_ARG(boxes_dev && gt_dev && out_dev, "ssd_iou_map: NULL pointer");
    const int blocks = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(iou_map_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)boxes_dev, boxes_batched, (const float4*)gt_dev, N, G, total, out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int ssd_encode_deltas(const float* bboxes_dev, const float* gt_dev, int B, int N, float* out_dev,
                      void* stream) {
    SSD_CHECK_ARG(B >= 0 && N >= 0, "ssd_encode_deltas: B=%d N=%d", B, N);
    const long total = (long)B * N;
    if (total == 0) return SSD_OK;
    SSD_CHECK_ARG(bboxes_dev && gt_dev && out_dev, "ssd_encode_deltas: NULL pointer");
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)bboxes_dev, (const float4*)gt_dev, N, total, (float4*)out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int ssd_match_encode(const float* priors_dev, const float* gt_boxes_dev, const int* gt_labels_dev,
                     const float* var, float iou_thr, int B, int N, int G, int L,
                     float* deltas_out_dev, int* label_idx_out_dev, int* match_idx_out_dev,
                     float* onehot_out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && N >= 0 && G >= 0 && L >= 1, "ssd_match_encode: B=%d N=%d G=%d L=%d", B, N, G, L);
    SSD_CHECK_ARG(var != nullptr, "ssd_match_encode: variances pointer is NULL");
    if ((long)B * N == 0) return SSD_OK;
    SSD_CHECK_ARG(priors_dev && deltas_out_dev && label_idx_out_dev && match_idx_out_dev,
                  "ssd_match_encode: NULL pointer");
    SSD_CHECK_ARG(G == 0 || (gt_boxes_dev && gt_labels_dev), "ssd_match_encode: NULL gt pointer");
    SSD_UNSUPPORTED_IF((size_t)G * 20 > 60 * 1024, "ssd_match_encode: G=%d exceeds the LDS-staged limit", G);
    const float4 v4 = make_float4(var[0], var[1], var[2], var[3]);
    hipLaunchKernelGGL(match_encode_kernel, dim3(cdiv(N, 256), B), dim3(256), (size_t)G * 20 + 16,
                       (hipStream_t)stream, (const float4*)priors_dev, (const float4*)gt_boxes_dev,
                       gt_labels_dev, v4, iou_thr, N, G, L, (float4*)deltas_out_dev, label_idx_out_dev,
                       match_idx_out_dev, onehot_out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // extern "C"

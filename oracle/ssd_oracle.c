/*
 * CPU oracle (TEST INFRASTRUCTURE, not product code) -- plain-C restatement of the
 * reference's decode / NMS / matching arithmetic, used where the NumPy/Python oracle
 * (oracle/bbox_oracle.py) is too slow: full-size parity checks and bench.py's
 * cpu_baseline leg.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may load this library.  The product path never links or calls it.
 *
 * PARITY STATUS: parity unpinned (see oracle/bbox_oracle.py header): the reference has
 * no golden vectors and TensorFlow is not installable in the build container.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utils/bbox_utils.py:61-85 with models/decoder.py:41 (deltas *= variances) folded in
 * front: every product and sum is rounded separately (-ffp-contract=off). */
static void decode_one(const float *p, const float *d, const float *var, float *o)
{
    float d0 = d[0] * var[0], d1 = d[1] * var[1], d2 = d[2] * var[2], d3 = d[3] * var[3];
    float pw = p[3] - p[1];
    float ph = p[2] - p[0];
    float pcx = p[1] + 0.5f * pw;
    float pcy = p[0] + 0.5f * ph;
    float w = expf(d3) * pw;
    float h = expf(d2) * ph;
    float cx = (d1 * pw) + pcx;
    float cy = (d0 * ph) + pcy;
    float y1 = cy - (0.5f * h);
    float x1 = cx - (0.5f * w);
    o[0] = y1;
    o[1] = x1;
    o[2] = h + y1;
    o[3] = w + x1;
}

void oracle_decode(const float *priors, const float *deltas, const float *var,
                   int B, int N, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            decode_one(priors + 4 * i, deltas + ((size_t)b * N + i) * 4, var,
                       out + ((size_t)b * N + i) * 4);
}

/* [3P] TF CombinedNonMaxSuppression IOU helper. */
static float nms_iou(const float *a, const float *b)
{
    float ymin_i = fminf(a[0], a[2]), ymax_i = fmaxf(a[0], a[2]);
    float xmin_i = fminf(a[1], a[3]), xmax_i = fmaxf(a[1], a[3]);
    float ymin_j = fminf(b[0], b[2]), ymax_j = fmaxf(b[0], b[2]);
    float xmin_j = fminf(b[1], b[3]), xmax_j = fmaxf(b[1], b[3]);
    float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    float iy = fmaxf(fminf(ymax_i, ymax_j) - fmaxf(ymin_i, ymin_j), 0.0f);
    float ix = fmaxf(fminf(xmax_i, xmax_j) - fmaxf(xmin_i, xmin_j), 0.0f);
    float inter = iy * ix;
    return inter / (area_i + area_j - inter);
}

typedef struct { float score; int idx; int cls; } cand_t;

static int cmp_cand(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    if (a->idx != b->idx) return a->idx < b->idx ? -1 : 1;
    return (a->cls > b->cls) - (a->cls < b->cls);
}

/* models/decoder.py:36-55 (SSDDecoder.call) followed by the [3P] combined NMS
 * (SURVEY.md Appendix B).  Outputs in the reference's return order + valid + the kept
 * anchor index per output row (-1 on padding rows). */
int oracle_decode_nms(const float *deltas, const float *probs, const float *priors,
                      const float *var, int B, int N, int L, int max_per_class,
                      int max_total, float iou_thr, float score_thr,
                      float *boxes, float *labels, float *scores, int *valid, int *kept_idx)
{
    float *dec = (float *)malloc((size_t)N * 4 * sizeof(float));
    unsigned char *masked = (unsigned char *)malloc((size_t)N);
    cand_t *cand = (cand_t *)malloc((size_t)N * sizeof(cand_t));
    cand_t *merged = (cand_t *)malloc((size_t)L * (max_per_class > N ? N : max_per_class) * sizeof(cand_t) + sizeof(cand_t));
    int *sel = (int *)malloc((size_t)(max_per_class > N ? N : max_per_class) * sizeof(int) + sizeof(int));
    if (!dec || !masked || !cand || !merged || !sel) return -1;
    int per_class = max_per_class > N ? N : max_per_class;
    for (int b = 0; b < B; ++b) {
        const float *pr = probs + (size_t)b * N * L;
        for (int i = 0; i < N; ++i) {
            decode_one(priors + 4 * i, deltas + ((size_t)b * N + i) * 4, var, dec + 4 * i);
            /* decoder.py:44-45: argmax (first max wins) == 0 -> the whole row is zeroed */
            int am = 0;
            float best = pr[(size_t)i * L];
            for (int c = 1; c < L; ++c)
                if (pr[(size_t)i * L + c] > best) { best = pr[(size_t)i * L + c]; am = c; }
            masked[i] = (am == 0);
        }
        int nm = 0;
        for (int c = 0; c < L; ++c) {
            int nc = 0;
            for (int i = 0; i < N; ++i) {
                float s = masked[i] ? 0.0f : pr[(size_t)i * L + c];
                if (s > score_thr) { cand[nc].score = s; cand[nc].idx = i; cand[nc].cls = c; ++nc; }
            }
            qsort(cand, nc, sizeof(cand_t), cmp_cand);
            int ns = 0;
            for (int q = 0; q < nc && ns < per_class; ++q) {
                int keep = 1;
                for (int j = ns - 1; j >= 0; --j)
                    if (nms_iou(dec + 4 * cand[q].idx, dec + 4 * sel[j]) > iou_thr) { keep = 0; break; }
                if (keep) { sel[ns++] = cand[q].idx; merged[nm++] = cand[q]; }
            }
        }
        qsort(merged, nm, sizeof(cand_t), cmp_cand);
        int n = nm < max_total ? nm : max_total;
        valid[b] = n;
        for (int r = 0; r < max_total; ++r) {
            float *ob = boxes + ((size_t)b * max_total + r) * 4;
            if (r < n) {
                const float *bx = dec + 4 * merged[r].idx;
                for (int k = 0; k < 4; ++k) ob[k] = fminf(fmaxf(bx[k], 0.0f), 1.0f);
                labels[(size_t)b * max_total + r] = (float)merged[r].cls;
                scores[(size_t)b * max_total + r] = merged[r].score;
                if (kept_idx) kept_idx[(size_t)b * max_total + r] = merged[r].idx;
            } else {
                ob[0] = ob[1] = ob[2] = ob[3] = 0.0f;
                labels[(size_t)b * max_total + r] = 0.0f;
                scores[(size_t)b * max_total + r] = 0.0f;
                if (kept_idx) kept_idx[(size_t)b * max_total + r] = -1;
            }
        }
    }
    free(dec); free(masked); free(cand); free(merged); free(sel);
    return 0;
}

/* utils/bbox_utils.py:27-59 for priors [N,4] x gt [B,G,4] -> [B,N,G]. */
static float pair_iou(const float *p, const float *g)
{
    float garea = (g[2] - g[0]) * (g[3] - g[1]);
    float parea = (p[2] - p[0]) * (p[3] - p[1]);
    float x_top = fmaxf(p[1], g[1]), y_top = fmaxf(p[0], g[0]);
    float x_bot = fminf(p[3], g[3]), y_bot = fminf(p[2], g[2]);
    float inter = fmaxf(x_bot - x_top, 0.0f) * fmaxf(y_bot - y_top, 0.0f);
    float uni = parea + garea - inter;
    return inter / uni;
}

void oracle_iou_map(const float *priors, const float *gt, int B, int N, int G, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int g = 0; g < G; ++g)
                out[((size_t)b * N + i) * G + g] = pair_iou(priors + 4 * i, gt + ((size_t)b * G + g) * 4);
}

/* utils/train_utils.py:90-127 + utils/bbox_utils.py:87-113.  label_idx/match_idx int32. */
void oracle_match_encode(const float *priors, const float *gt, const int *gt_labels,
                         const float *var, float iou_thr, int B, int N, int G,
                         float *deltas, int *label_idx, int *match_idx)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *p = priors + 4 * i;
            int am = 0;
            /* [3P] Eigen arg-max / max reducers: start at lowest(), strictly-greater replaces: first
             * max wins, a NaN IoU (0/0) is never selected */
            float best = G > 0 ? -3.402823466e38f : 0.0f;
            for (int g = 0; g < G; ++g) {
                float v = pair_iou(p, gt + ((size_t)b * G + g) * 4);
                if (v > best) { best = v; am = g; }
            }
            int pos = best > iou_thr;
            float gb[4] = {0, 0, 0, 0};
            int lab = 0;
            if (pos) { memcpy(gb, gt + ((size_t)b * G + am) * 4, 16); lab = gt_labels[(size_t)b * G + am]; }
            float bw = p[3] - p[1], bh = p[2] - p[0];
            float bcx = p[1] + 0.5f * bw, bcy = p[0] + 0.5f * bh;
            float gw = gb[3] - gb[1], gh = gb[2] - gb[0];
            float gcx = gb[1] + 0.5f * gw, gcy = gb[0] + 0.5f * gh;
            if (bw == 0.0f) bw = 1e-3f;
            if (bh == 0.0f) bh = 1e-3f;
            float dx = gw == 0.0f ? 0.0f : (gcx - bcx) / bw;
            float dy = gh == 0.0f ? 0.0f : (gcy - bcy) / bh;
            float dw = gw == 0.0f ? 0.0f : logf(gw / bw);
            float dh = gh == 0.0f ? 0.0f : logf(gh / bh);
            float *o = deltas + ((size_t)b * N + i) * 4;
            o[0] = dy / var[0]; o[1] = dx / var[1]; o[2] = dh / var[2]; o[3] = dw / var[3];
            label_idx[(size_t)b * N + i] = lab;
            match_idx[(size_t)b * N + i] = am;
        }
}

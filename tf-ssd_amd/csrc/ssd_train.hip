// Training step of the SSD graphs (reference trainer.py:50-76):
// training-mode forward (BatchNorm on batch statistics, moving averages updated with Keras'
// momentum 0.999), the HIP loss (csrc/ssd_loss.hip), backward of every layer and Adam.
//
//   conv backward-data   = the forward MFMA implicit-GEMM kernel (conv_mfma_kernel) on dY with the
//                          180-degree-rotated, in/out-transposed weights; stride-2 3x3 convs go
//                          through a zero-inserted dY (they are the four small SSD extras);
//   conv backward-weights= wgrad_mfma_kernel below: dW[K,N] = im2col(X)^T [K,M] * dY [M,N] on
//                          v_mfma_f32_16x16x4_f32, M split over the grid, deterministic two-stage sum;
//   depthwise backward   = gather kernels (HBM-bound);
//   BatchNorm            = two-pass batch statistics and the usual two-reduction backward, all
//                          reductions deterministic (fixed chunking, fixed order);
//   Adam                 = one fused kernel over the flat parameter / moment / gradient buffers.
// Gradients leave through a caller-owned flat buffer (parameter-table order), which is what the
// host all-reduces over RCCL between backward and the Adam step.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "ssd_bf16x3.h"
#include "ssd_net.h"

using namespace ssd;

extern "C" size_t ssd_loss_workspace_bytes(int B, int N);
extern "C" int ssd_loss(const float*, const float*, const float*, const float*, int, int, int, float, float, float*,
                        float*, float*, float*, float*, float*, float, void*, size_t, void*);

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kBnEps = 1e-3f;          // keras-applications MobileNetV2: BatchNormalization(epsilon=1e-3, momentum=0.999)
constexpr float kBnMomentum = 0.999f;

// ------------------------------------------------------------------ small elementwise kernels
__device__ __forceinline__ float bn_z(float y, float mean, float istd, float gamma, float beta) {
    return (y - mean) * istd * gamma + beta;
}
__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == SSD_ACT_RELU) return fmaxf(z, 0.f);
    if (act == SSD_ACT_RELU6) return fminf(fmaxf(z, 0.f), 6.f);
    return z;
}
// TF ReluGrad / Relu6Grad: gradient passes strictly inside the linear range
__device__ __forceinline__ float act_mask(float z, int act) {
    if (act == SSD_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == SSD_ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
    return 1.f;
}

// out = act(bn(pre)) (+ residual)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ pre, const long M, const int C,
                                                      const float* __restrict__ mean, const float* __restrict__ istd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const int act, const float* __restrict__ res,
                                                      float* __restrict__ out) {
    const long total = M * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        float v = act_fwd(bn_z(pre[e], mean[c], istd[c], gamma[c], beta[c]), act);
        if (res) v += res[e];
        out[e] = v;
    }
}

// dy = gamma * istd * (dz - dbeta / M - xhat * dgamma / M),  dz = dout * act'(z)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ pre,
                                                          const long M, const int C, const float* __restrict__ mean,
                                                          const float* __restrict__ istd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const int act,
                                                          const float* __restrict__ dgamma,
                                                          const float* __restrict__ dbeta, float* __restrict__ dy) {
    const long total = M * C;
    const float invM = 1.0f / (float)M;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float xh = (pre[e] - mean[c]) * istd[c];
        const float z = xh * gamma[c] + beta[c];
        const float dz = dout[e] * act_mask(z, act);
        dy[e] = gamma[c] * istd[c] * (dz - dbeta[c] * invM - xh * dgamma[c] * invM);
    }
}

// dz = dout * (out > 0)   (bias + ReLU layers)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                     const long total, const int act, float* __restrict__ dz) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
        dz[e] = dout[e] * act_mask(out[e], act);
}

// dst (+)= src
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                 const long total, const int accumulate) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
        dst[e] = accumulate ? dst[e] + src[e] : src[e];
}

// Keras moving average: var -= (var - value) * (1 - momentum)
// (the fused BatchNorm op hands Keras the Bessel-corrected batch variance, var * M / (M - 1), and Keras'
// BatchNormalization keeps it for the moving average -- `_bessels_correction_test_only` -- while the
// normalisation and the backward use the biased one)
__global__ void moving_update_kernel(float* mm, float* mv, const float* mean, const float* var, const int C,
                                     const float one_minus_momentum, const float bessel) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mm[c] -= (mm[c] - mean[c]) * one_minus_momentum;
        mv[c] -= (mv[c] - var[c] * bessel) * one_minus_momentum;
    }
}

// Wt[ky'][kx'][co_off + co][ci] = W[kh-1-ky'][kw-1-kx'][ci][co]   (HWIO -> rotated, transposed HWIO
// with I = CoPad, O = Ci; rows co >= Co stay zero: the caller memsets Wt)
__global__ __launch_bounds__(256) void rot_transpose_kernel(const float* __restrict__ w, const int kh, const int kw,
                                                           const int Ci, const int Co, const int CoPad,
                                                           const int co_off, float* __restrict__ wt) {
    const long total = (long)kh * kw * Ci * Co;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int co = (int)(e % Co);
        long r = e / Co;
        const int ci = (int)(r % Ci);
        r /= Ci;
        const int kx = (int)(r % kw), ky = (int)(r / kw);
        wt[((long)((kh - 1 - ky) * kw + (kw - 1 - kx)) * CoPad + co_off + co) * Ci + ci] = w[e];
    }
}

// Every weight re-pack of a training step as ONE launch (the weights change every step; 103 forward
// packs + 54 rotate/transposes + 50 backward packs were 0.75 ms of 5 us launches): blockIdx.y = job.
//   mode 0  forward:        dst[n][k] = W[k][n]                          ([Npad][Kpad], zero padded;
//           a head conv's label (w, Cout1 columns) and box (w2) kernels are stacked along n)
//   mode 1  backward-data:  dst[ci][(ky', kx', co')] = W[kh-1-ky'][kw-1-kx'][ci][co']   (rotated and
//           transposed: the forward conv kernel then computes dX from dY; co' < CoPad, zero padded)
struct PackJob {
    const float* w;
    const float* w2;
    float* dst;
    int mode, K, Cout, Cout1, Kpad, Npad, kh, kw, Ci, CoPad;
    int bf16;       // the net's precision: 0 -> the planes h, m, l (split-bf16 tiles) are written, 1 -> only the rounding r (bf16 tiles)
};
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJob* __restrict__ jobs) {
    const PackJob j = jobs[blockIdx.y];
    const int total = j.Npad * j.Kpad;            // < 2^31 (ssd_net_train_begin checks): 32-bit index arithmetic
    const int C2 = j.Cout - j.Cout1;
    // the jobs differ by three orders of magnitude (a 16 x 32 pointwise kernel .. head level 2's 160 x 11 520): every job gets
    // gridDim.x blocks, the small ones leave at once (16 blocks per job took 309 us of a 10.8 ms step)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int n = e / j.Kpad, k = e - n * j.Kpad;
        float v = 0.f;
        if (j.mode == 0) {
            if (n < j.Cout && k < j.K) v = n < j.Cout1 ? j.w[(long)k * j.Cout1 + n] : j.w2[(long)k * C2 + (n - j.Cout1)];
        } else if (n < j.Ci && k < j.kh * j.kw * j.CoPad) {
            const int co = k % j.CoPad, tap = k / j.CoPad;
            if (co < j.Cout) {
                const int ky = j.kh - 1 - tap / j.kw, kx = j.kw - 1 - tap % j.kw;
                const long row = (long)(ky * j.kw + kx) * j.Ci + n;
                v = co < j.Cout1 ? j.w[row * j.Cout1 + co] : j.w2[row * C2 + (co - j.Cout1)];
            }
        }
        j.dst[e] = v;
        // the same matrix as four bf16 planes behind the fp32 one: the exact split h, m, l (split-bf16 conv tiles,
        // ssd_conv3.hip) and the bf16 rounding r (bf16 tiles)
        // (only the planes this precision's tiles read: the re-pack runs every step)
        short* pl = reinterpret_cast<short*>(j.dst + total);
        const long d = plane_elem(n, k, j.Npad);          // planes: [Kpad / 32][Npad][32] (ssd_bf16x3.h)
        if (j.bf16) {
            pl[3 * (long)total + d] = rne1(v);
        } else {
            short h, m, l;
            split1(v, h, m, l);
            pl[d] = h;
            pl[(long)total + d] = m;
            pl[2 * (long)total + d] = l;
        }
    }
}

// dense[b][p][c] (row stride ld) = src[b * bs + off + p * ps + c]   (one head level out of the
// concatenated [B,N,K] gradient buffer); col0 = first dense column written
__global__ __launch_bounds__(256) void gather_head_kernel(const float* __restrict__ src, const long bs, const long off,
                                                         const long ps, const int B, const int P, const int C,
                                                         float* __restrict__ dense, const int ld, const int col0) {
    const long total = (long)B * P * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long r = e / C;
        const int p = (int)(r % P), b = (int)(r / P);
        dense[r * ld + col0 + c] = src[b * bs + off + (long)p * ps + c];
    }
}

// z[b][2*oy][2*ox][c] = y[b][oy][ox][c], zero elsewhere (the caller memsets z)
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ y, const int B, const int Ho, const int Wo,
                                                     const int C, const int Hz, const int Wz, float* __restrict__ z) {
    const long total = (long)B * Ho * Wo * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        long r = e / C;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        z[(((long)b * Hz + 2 * oy) * Wz + 2 * ox) * C + c] = y[e];
    }
}

// ------------------------------------------------------------------ column reductions over [M][C]
// block = 64 columns x 4 row lanes; grid = (ceil(C/64), chunks); partial[chunk][2][C].
enum { RED_SUM = 0, RED_SQDEV = 1, RED_BN_BWD = 2, RED_ACT_BWD = 3, RED_STATS = 4 };
struct RedParams {
    const float* a;       // SUM/SQDEV: x;  BN_BWD / ACT_BWD: dout
    const float* b;       // BN_BWD: pre;   ACT_BWD: out (nullable: no activation)
    const float *mean, *istd, *gamma, *beta;
    long M;
    int C, lda, act, cw;
    long rows_per_chunk;
    float* partial;
};
template <int OP>
__global__ __launch_bounds__(256) void col_reduce_kernel(const RedParams p) {
    // block = CW columns x (256 / CW) row lanes; CW in {16, 32, 64} chosen on the host so that
    // narrow tensors (C = 16 .. 32) keep every lane busy
    __shared__ float sh[2][256];
    const int CW = p.cw, RL = 256 / CW;
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    const long r0 = (long)blockIdx.y * p.rows_per_chunk;
    const long r1 = r0 + p.rows_per_chunk < p.M ? r0 + p.rows_per_chunk : p.M;
    float s1 = 0.f, s2 = 0.f;
    if (c < p.C) {
        float mean = 0.f, istd = 0.f, gamma = 0.f, beta = 0.f;
        if (OP == RED_SQDEV || OP == RED_BN_BWD) mean = p.mean[c];
        if (OP == RED_BN_BWD) { istd = p.istd[c]; gamma = p.gamma[c]; beta = p.beta[c]; }
        for (long r = r0 + rl; r < r1; r += RL) {
            const float a = p.a[r * p.lda + c];
            if (OP == RED_SUM) s1 += a;
            else if (OP == RED_SQDEV) { const float d = a - mean; s1 += d * d; }
            else if (OP == RED_BN_BWD) {
                const float xh = (p.b[r * p.C + c] - mean) * istd;
                const float dz = a * act_mask(xh * gamma + beta, p.act);
                s1 += dz;
                s2 += dz * xh;
            } else {
                s1 += p.b ? a * act_mask(p.b[r * p.C + c], p.act) : a;
            }
        }
    }
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0 && c < p.C) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < RL; ++k) {           // fixed order: deterministic
            t1 += sh[0][k * CW + cl];
            t2 += sh[1][k * CW + cl];
        }
        const long o = (long)blockIdx.y * 2 * p.C;
        p.partial[o + c] = t1;
        p.partial[o + p.C + c] = t2;
    }
}

// 16-byte form of col_reduce_kernel for C % 4 == 0 (every tensor of both graphs): thread = 4
// consecutive channels; block = CQ channel quads x (256 / CQ) row lanes.  RED_STATS: one pass for
// the batch statistics -- sums of (x - K) and (x - K)^2 with the per-column shift K = x[0][c] common to
// every chunk (a sample of the column, so (mean - K)^2 ~ var: no cancellation in s2/M - (s1/M)^2).
typedef float tf32x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(256) void col_reduce4_kernel(const RedParams p) {
    __shared__ tf32x4 sh[2][256];
    const int CQ = p.cw, RL = 256 / CQ;
    const int cl = threadIdx.x % CQ, rl = threadIdx.x / CQ;
    const int c = (blockIdx.x * CQ + cl) * 4;
    const long r0 = (long)blockIdx.y * p.rows_per_chunk;
    const long r1 = r0 + p.rows_per_chunk < p.M ? r0 + p.rows_per_chunk : p.M;
    tf32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        tf32x4 mean = s1, istd = s1, gamma = s1, beta = s1;
        if (OP == RED_SQDEV || OP == RED_BN_BWD) mean = *reinterpret_cast<const tf32x4*>(p.mean + c);
        if (OP == RED_STATS) mean = *reinterpret_cast<const tf32x4*>(p.a + c);          // the shift K
        if (OP == RED_BN_BWD) {
            istd = *reinterpret_cast<const tf32x4*>(p.istd + c);
            gamma = *reinterpret_cast<const tf32x4*>(p.gamma + c);
            beta = *reinterpret_cast<const tf32x4*>(p.beta + c);
        }
#pragma unroll 4
        for (long r = r0 + rl; r < r1; r += RL) {
            const tf32x4 a = *reinterpret_cast<const tf32x4*>(p.a + r * p.lda + c);
            if (OP == RED_SUM) s1 += a;
            else if (OP == RED_SQDEV) { const tf32x4 d = a - mean; s1 += d * d; }
            else if (OP == RED_STATS) { const tf32x4 d = a - mean; s1 += d; s2 += d * d; }
            else if (OP == RED_BN_BWD) {
                const tf32x4 xh = (*reinterpret_cast<const tf32x4*>(p.b + r * p.C + c) - mean) * istd;
                const tf32x4 z = xh * gamma + beta;
                tf32x4 dz;
#pragma unroll
                for (int j = 0; j < 4; ++j) dz[j] = a[j] * act_mask(z[j], p.act);
                s1 += dz;
                s2 += dz * xh;
            } else {
                if (p.b) {
                    const tf32x4 o = *reinterpret_cast<const tf32x4*>(p.b + r * p.C + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) s1[j] += a[j] * act_mask(o[j], p.act);
                } else {
                    s1 += a;
                }
            }
        }
    }
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    if (rl == 0 && c < p.C) {
        tf32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < RL; ++k) {           // fixed order: deterministic
            t1 += sh[0][k * CQ + cl];
            t2 += sh[1][k * CQ + cl];
        }
        const long o = (long)blockIdx.y * 2 * p.C;
        *reinterpret_cast<tf32x4*>(p.partial + o + c) = t1;
        *reinterpret_cast<tf32x4*>(p.partial + o + p.C + c) = t2;
    }
}
// out1[c] = scale * sum_chunks partial[.][0][c]; out2 likewise (nullable); mode 1: out2 = rsqrt(out1 + eps);
// mode 2 (batch statistics from RED_STATS partials, shift row `shift`): out1 = mean, out2 = biased variance,
// out3 = rsqrt(var + eps), and the Keras moving averages mm / mv move towards them
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ partial, const int chunks, const int C,
                                                          const float scale, float* out1, float* out2, const int mode,
                                                          const float eps, const float* shift = nullptr, float* out3 = nullptr,
                                                          float* mm = nullptr, float* mv = nullptr,
                                                          const float one_minus_momentum = 0.f, const float bessel = 1.f) {
    // block = 16 columns x 16 chunk lanes (lane k sums chunks k, k+16, ... in order; the 16 lane
    // sums are combined in a fixed order: deterministic).  The data is tiny; the kernel is latency
    // bound, hence many short dependent chains instead of few long ones.
    __shared__ float sh[2][16][16];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s1 = 0.f, s2 = 0.f;
    // eight chunks' loads in flight per lane before the (in-order) adds: the loop is a chain of L2 round trips otherwise
    // (256 chunks = 16 trips of ~0.4 us: 6.9 us per launch, 118 launches per MobileNetV2 step)
    if (c < C)
        for (int k0 = kl; k0 < chunks; k0 += 128) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 16 * u;
                const bool ok = k < chunks;
                const long o = (long)(ok ? k : 0) * 2 * C + c;
                a[u] = partial[o];
                b[u] = partial[o + C];
                if (!ok) a[u] = b[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += a[u];
                s2 += b[u];
            }
        }
    sh[0][kl][cl] = s1;
    sh[1][kl][cl] = s2;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    s1 = s2 = 0.f;
    for (int k = 0; k < 16; ++k) {
        s1 += sh[0][k][cl];
        s2 += sh[1][k][cl];
    }
    s1 *= scale;
    s2 *= scale;
    if (mode == 2) {
        const float mean = shift[c] + s1;
        const float var = fmaxf(s2 - s1 * s1, 0.0f);
        out1[c] = mean;
        out2[c] = var;
        out3[c] = 1.0f / sqrtf(var + eps);
        mm[c] -= (mm[c] - mean) * one_minus_momentum;       // Keras moving average: v -= (v - value) * (1 - momentum)
        mv[c] -= (mv[c] - var * bessel) * one_minus_momentum;   // Bessel-corrected value, see moving_update_kernel
        return;
    }
    if (out1) out1[c] = s1;
    if (mode == 1) out2[c] = 1.0f / sqrtf(s1 + eps);
    else if (out2) out2[c] = s2;
}
// out[i] = sum_chunks partial[chunk][i]: block = 64 elements x 4 chunk lanes (lane k sums chunks
// k, k+4, ... in order, then the 4 lane sums in a fixed order: deterministic)
__global__ __launch_bounds__(256) void chunk_sum_kernel(const float* __restrict__ partial, const int chunks,
                                                       const long n, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int el = threadIdx.x & 63, kl = threadIdx.x >> 6;
    for (long e0 = (long)blockIdx.x * 64; e0 < n; e0 += (long)gridDim.x * 64) {
        const long e = e0 + el;
        float s = 0.f;
        if (e < n)
            for (int k = kl; k < chunks; k += 4) s += partial[(long)k * n + e];
        __syncthreads();
        sh[kl][el] = s;
        __syncthreads();
        if (kl == 0 && e < n) out[e] = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
    }
}

// 16-byte form for n % 4 == 0: block = EQ element quads x (256 / EQ) chunk lanes (lane k sums chunks
// k, k + KL, ... in order, then the KL lane sums in a fixed order: deterministic).  Small outputs
// (a 96 x 24 pointwise kernel summed over 1024 M-chunks) get EQ = 4: 16 elements per block, 64
// chunk lanes -- the 64-element / 4-lane form left such layers with 36 blocks walking 256 chunks each.
template <int EQ>
__global__ __launch_bounds__(256) void chunk_sum4_kernel(const float* __restrict__ partial, const int chunks,
                                                        const long n, float* __restrict__ out, const int aligned_out) {
    constexpr int KL = 256 / EQ;
    __shared__ tf32x4 sh[KL][EQ];
    const int el = threadIdx.x % EQ, kl = threadIdx.x / EQ;
    const long e = ((long)blockIdx.x * EQ + el) * 4;
    tf32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e < n)
        for (int k0 = kl; k0 < chunks; k0 += 8 * KL) {          // eight loads in flight, added in chunk order
            tf32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * KL;
                v[u] = *reinterpret_cast<const tf32x4*>(partial + (long)(k < chunks ? k : 0) * n + e);
                if (k >= chunks) v[u] = tf32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    sh[kl][el] = acc;
    __syncthreads();
    if (kl != 0 || e >= n) return;
    tf32x4 t = sh[0][el];
    for (int k = 1; k < KL; ++k) t += sh[k][el];
    if (aligned_out) *reinterpret_cast<tf32x4*>(out + e) = t;
    else
        for (int j = 0; j < 4; ++j) out[e + j] = t[j];
}
static int chunk_sum(const float* partial, long chunks, long n, float* out, hipStream_t st) {
    if (n % 4 == 0 && ((uintptr_t)partial & 15) == 0) {
        const int aligned = ((uintptr_t)out & 15) == 0;
        if (n < 65536) hipLaunchKernelGGL(chunk_sum4_kernel<4>, dim3((unsigned)((n / 4 + 3) / 4)), dim3(256), 0, st, partial, (int)chunks, n, out, aligned);
        else hipLaunchKernelGGL(chunk_sum4_kernel<16>, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, st, partial, (int)chunks, n, out, aligned);
    } else {
        const long b = (n * 4 + 255) / 256;
        hipLaunchKernelGGL(chunk_sum_kernel, dim3((unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b))), dim3(256), 0, st, partial,
                           (int)chunks, n, out);
    }
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------ conv backward-weights (MFMA)
// dW[k = (tap, ci)][n] = sum_m X[pixel(m, tap)][ci] * G[m][n].  One workgroup = a 64 (ci) x 64 (n)
// tile of one tap over one M chunk; both operands are staged row-major ([m][channel], exactly as
// they lie in HBM: coalesced 16-byte loads) into LDS with an 80-float row stride, so that the
// MFMA fragments -- A[i][kk] = X[m0+kk][c0+i], B[kk][j] = G[m0+kk][n0+j], kk = lane/16 -- are
// conflict-free ds_read_b32 (bank = 16*kk + lane%16).
constexpr int WG_BM = 32;
struct WgradParams {
    const float* x;
    const float* g;
    float* partial;       // [chunks][K][N]
    int B, H, W, Cin, Ho, Wo, kh, kw, stride, dil, pad_t, pad_l;
    int N, ldg, K;
    long M, rows_per_chunk;
    int ctiles, ntiles;   // channel tiles per tap, n tiles
    int vec_x, vec_g;
    int im2col;           // 1 (Cin % 4 != 0: the RGB stems): a tile's rows are k = (tap, ci) jointly, one "tap"
};
// Tile shape = (WC x WN wave grid) x (TI x TJ 16x16 MFMA tiles per wave): TC = 16*TI*WC input channels by
// TN = 16*TJ*WN output channels.  MobileNetV2 has many 16-32 channel sides (a 16 -> 96 expand fills 19 %
// of a 64 x 64 tile), so the host picks per layer the shape with the least padded area.
template <int WC, int WN_, int TI, int TJ>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradParams p) {
    static_assert(WC * WN_ == 4, "4 waves");
    constexpr int TC = 16 * TI * WC, TN = 16 * TJ * WN_;
    // row strides = 16 (mod 64) floats: the four 16-lane groups of a fragment read (rows 4s + kk) hit
    // disjoint bank quarters
    constexpr int LDX = (TC + 63) / 64 * 64 + 16, LDG = (TN + 63) / 64 * 64 + 16;
    constexpr int XQ = TC / 4, GQ = TN / 4;                // float4 per row
    constexpr int XU = WG_BM * XQ, GU = WG_BM * GQ;        // float4 units per slab
    __shared__ __attribute__((aligned(16))) float Xs[WG_BM * LDX];
    __shared__ __attribute__((aligned(16))) float Gs[WG_BM * LDG];
    int t = blockIdx.x;
    const int nt = t % p.ntiles;
    t /= p.ntiles;
    const int ct = t % p.ctiles, tap = t / p.ctiles;           // im2col: tap == 0
    const int ky = tap / p.kw, kx = tap % p.kw;
    const int c0 = ct * TC, n0 = nt * TN;
    const int climit = p.im2col ? p.K : p.Cin;                  // valid rows of the tile
    const long mBeg = (long)blockIdx.y * p.rows_per_chunk;
    const long mEnd = mBeg + p.rows_per_chunk < p.M ? mBeg + p.rows_per_chunk : p.M;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wk = wv / WN_, wn = wv % WN_;
    f32x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long m0 = mBeg; m0 < mEnd; m0 += WG_BM) {
        __syncthreads();
#pragma unroll
        for (int u0 = 0; u0 < XU; u0 += 256) {
            const int u = u0 + tid;
            if (XU % 256 != 0 && u >= XU) break;
            const int row = u / XQ, q = (u - row * XQ) * 4;
            const long m = m0 + row;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (m < mEnd && p.im2col) {
                // K = kh*kw*Cin is small (27 for the RGB stems): the tile row index IS k = tap * Cin + ci
                const int ox = (int)(m % p.Wo);
                const long r = m / p.Wo;
                const int oy = (int)(r % p.Ho), b = (int)(r / p.Ho);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = c0 + q + j;
                    if (k >= p.K) continue;
                    const int tp = k / p.Cin, ci = k - tp * p.Cin;
                    const int iy = oy * p.stride - p.pad_t + (tp / p.kw) * p.dil, ix = ox * p.stride - p.pad_l + (tp % p.kw) * p.dil;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                        xv[j] = p.x[(((long)b * p.H + iy) * p.W + ix) * p.Cin + ci];
                }
            } else if (m < mEnd) {
                const int ox = (int)(m % p.Wo);
                const long r = m / p.Wo;
                const int oy = (int)(r % p.Ho), b = (int)(r / p.Ho);
                const int iy = oy * p.stride - p.pad_t + ky * p.dil, ix = ox * p.stride - p.pad_l + kx * p.dil;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const float* xp = p.x + (((long)b * p.H + iy) * p.W + ix) * p.Cin + c0 + q;
                    if (p.vec_x && c0 + q + 3 < p.Cin) xv = *reinterpret_cast<const f32x4*>(xp);
                    else
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (c0 + q + j < p.Cin) xv[j] = xp[j];
                }
            }
            *reinterpret_cast<f32x4*>(&Xs[row * LDX + q]) = xv;
        }
#pragma unroll
        for (int u0 = 0; u0 < GU; u0 += 256) {
            const int u = u0 + tid;
            if (GU % 256 != 0 && u >= GU) break;
            const int row = u / GQ, q = (u - row * GQ) * 4;
            const long m = m0 + row;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f};
            if (m < mEnd) {
                const float* gp = p.g + m * p.ldg + n0 + q;
                if (p.vec_g && n0 + q + 3 < p.N) gv = *reinterpret_cast<const f32x4*>(gp);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (n0 + q + j < p.N) gv[j] = gp[j];
            }
            *reinterpret_cast<f32x4*>(&Gs[row * LDG + q]) = gv;
        }
        __syncthreads();
        const int kk = lane >> 4, li = lane & 15;
#pragma unroll
        for (int s = 0; s < WG_BM / 4; ++s) {
            const int r = 4 * s + kk;
            float av[TI], bv[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) av[i] = Xs[r * LDX + (wk * TI + i) * 16 + li];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bv[j] = Gs[r * LDG + (wn * TJ + j) * 16 + li];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // lane owns D[i = 4*(lane/16) + r][j = lane%16] of each 16x16 tile
    float* out = p.partial + (long)blockIdx.y * p.K * p.N;
    const int li = lane & 15, lr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = n0 + (wn * TJ + j) * 16 + li;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = c0 + (wk * TI + i) * 16 + lr + r;
                if (ci < climit) out[((long)tap * p.Cin + ci) * p.N + n] = acc[i][j][r];
            }
        }
}
typedef void (*wgrad_kernel_t)(const WgradParams);
struct WgradCfg {
    int tc, tn;
    wgrad_kernel_t fn;
};
#define WGCFG(WC, WN_, TI, TJ) {16 * TI * WC, 16 * TJ * WN_, wgrad_mfma_kernel<WC, WN_, TI, TJ>}
static const WgradCfg kWgrad[] = {
    WGCFG(2, 2, 2, 2),      // 64 x 64
    WGCFG(1, 4, 1, 2),      // 16 x 128
    WGCFG(1, 4, 2, 2),      // 32 x 128
    WGCFG(1, 4, 1, 1),      // 16 x 64
    WGCFG(2, 2, 1, 2),      // 32 x 64
    WGCFG(4, 1, 2, 1),      // 128 x 16
    WGCFG(4, 1, 2, 2),      // 128 x 32
    WGCFG(2, 2, 2, 1),      // 64 x 32
    WGCFG(4, 1, 1, 1),      // 64 x 16
    WGCFG(2, 2, 4, 4),      // 128 x 128
    WGCFG(2, 2, 4, 2),      // 128 x 64
    WGCFG(2, 2, 2, 4),      // 64 x 128
};

// ------------------------------------------------------------------ depthwise 3x3 backward
struct DwBwdParams {
    const float* x;       // [B,H,W,C] forward input
    const float* g;       // [B,Ho,Wo,C] dY
    const float* w;       // [9][C]
    float* dx;            // [B,H,W,C]
    float* partial;       // [chunks][9][C]
    int B, H, W, C, Ho, Wo, stride, pad_t, pad_l, accumulate;
    long M, rows_per_chunk;
};
// dW[tap][c] partial sums: block = 16 channel quads (64 channels, 16-byte loads) x 16 row lanes,
// grid (ceil(C/64), chunks)
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const DwBwdParams p) {
    __shared__ __attribute__((aligned(16))) float sh[16][64];
    const int ql = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + ql * 4;
    const long r0 = (long)blockIdx.y * p.rows_per_chunk;
    const long r1 = r0 + p.rows_per_chunk < p.M ? r0 + p.rows_per_chunk : p.M;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < p.C) {
        // row lane rl walks a CONTIGUOUS run of output pixels with a sliding 3x3 window of x in registers:
        // along a row each step loads one new column (stride 1) or two (stride 2) instead of nine taps
        const long seg = (r1 - r0 + 15) / 16;
        const long mb = r0 + rl * seg, me = mb + seg < r1 ? mb + seg : r1;
        f32x4 xw[3][3];
        int pox = -2, poy = -1, pb = -1;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (long m = mb; m < me; ++m) {
            const int ox = (int)(m % p.Wo);
            const long r = m / p.Wo;
            const int oy = (int)(r % p.Ho), b = (int)(r / p.Ho);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.g + m * p.C + c);
            const bool slide = ox == pox + 1 && oy == poy && b == pb;
            const int first = slide ? 3 - p.stride : 0;            // first window column to (re)load
            if (slide) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (p.stride == 1) { xw[ky][0] = xw[ky][1]; xw[ky][1] = xw[ky][2]; }
                    else xw[ky][0] = xw[ky][2];
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * p.stride - p.pad_t + ky;
                const bool rowok = (unsigned)iy < (unsigned)p.H;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (kx < first) continue;
                    const int ix = ox * p.stride - p.pad_l + kx;
                    xw[ky][kx] = (rowok && (unsigned)ix < (unsigned)p.W)
                                     ? *reinterpret_cast<const f32x4*>(p.x + (((long)b * p.H + iy) * p.W + ix) * p.C + c)
                                     : zero;
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] += xw[ky][kx] * g;
            pox = ox; poy = oy; pb = b;
        }
    }
    float* out = p.partial + (long)blockIdx.y * 9 * p.C;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&sh[rl][ql * 4]) = acc[t];
        __syncthreads();
        if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < p.C) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += sh[k][threadIdx.x];       // fixed order: deterministic
            out[t * p.C + blockIdx.x * 64 + threadIdx.x] = v;
        }
    }
}
// dX[b][iy][ix][c] = sum_taps dY[b][(iy + pt - ky)/s][(ix + pl - kx)/s][c] * w[ky][kx][c]
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const DwBwdParams p) {
    const int C4 = p.C >> 2;
    const long total = (long)p.B * p.H * p.W * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4) * 4;
        long r = e / C4;
        const int ix = (int)(r % p.W);
        r /= p.W;
        const int iy = (int)(r % p.H), b = (int)(r / p.H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + p.pad_t - ky;
            if (ty < 0 || ty % p.stride) continue;
            const int oy = ty / p.stride;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + p.pad_l - kx;
                if (tx < 0 || tx % p.stride) continue;
                const int ox = tx / p.stride;
                if (ox >= p.Wo) continue;
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.g + (((long)b * p.Ho + oy) * p.Wo + ox) * p.C + c);
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (ky * 3 + kx) * p.C + c);
                acc += g * w;
            }
        }
        f32x4* d = reinterpret_cast<f32x4*>(p.dx + (((long)b * p.H + iy) * p.W + ix) * p.C + c);
        *d = p.accumulate ? *d + acc : acc;
    }
}

// stride-1 form (13 of the 17 depthwise layers): thread = 4 consecutive ix x 4 channels; the 3 x 6 window
// of dY serves all four outputs (4.5 loads per output instead of 9)
__global__ __launch_bounds__(256) void dw_dgrad4_kernel(const DwBwdParams p) {
    const int C4 = p.C >> 2, W4 = (p.W + 3) >> 2;
    const long total = (long)p.B * p.H * W4 * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4) * 4;
        long r = e / C4;
        const int ix0 = (int)(r % W4) * 4;
        r /= W4;
        const int iy = (int)(r % p.H), b = (int)(r / p.H);
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int oy = iy + p.pad_t - ky;
            if ((unsigned)oy >= (unsigned)p.Ho) continue;
            f32x4 g[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int ox = ix0 + p.pad_l - 2 + q;            // = (ix0 + j) + pad_l - kx  for  q = j + 2 - kx
                g[q] = (unsigned)ox < (unsigned)p.Wo ? *reinterpret_cast<const f32x4*>(p.g + (((long)b * p.Ho + oy) * p.Wo + ox) * p.C + c)
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(p.w + (ky * 3 + kx) * p.C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += g[j + 2 - kx] * w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ix0 + j >= p.W) break;
            f32x4* d = reinterpret_cast<f32x4*>(p.dx + (((long)b * p.H + iy) * p.W + ix0 + j) * p.C + c);
            *d = p.accumulate ? *d + acc[j] : acc[j];
        }
    }
}

// ------------------------------------------------------------------ max pool / L2 normalisation backward
// MaxPool2D backward, gather form (deterministic, also for the overlapping 3x3 stride-1 pool5): an
// input element receives dY of every window whose FIRST maximum (row-major scan over the valid
// cells -- TF ignores padded cells) it is.
struct PoolBwdParams {
    const float* x;      // [B,H,W,C] forward input
    const float* g;      // [B,Ho,Wo,C]
    float* dx;
    int B, H, W, C, Ho, Wo, k, stride, pad_t, pad_l, accumulate;
};
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const PoolBwdParams p) {
    const long total = (long)p.B * p.H * p.W * p.C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % p.C);
        long r = e / p.C;
        const int ix = (int)(r % p.W);
        r /= p.W;
        const int iy = (int)(r % p.H), b = (int)(r / p.H);
        const float* xb = p.x + (long)b * p.H * p.W * p.C + c;
        float acc = 0.f;
        // windows containing (iy, ix): oy*s - pt <= iy <= oy*s - pt + k - 1
        const int oy_lo = max(0, (iy + p.pad_t - p.k + 1 + p.stride - 1) / p.stride);
        const int ox_lo = max(0, (ix + p.pad_l - p.k + 1 + p.stride - 1) / p.stride);
        for (int oy = oy_lo; oy < p.Ho && oy * p.stride - p.pad_t <= iy; ++oy)
            for (int ox = ox_lo; ox < p.Wo && ox * p.stride - p.pad_l <= ix; ++ox) {
                float best = -INFINITY;
                int by = -1, bx = -1;
                for (int ky = 0; ky < p.k; ++ky) {
                    const int y = oy * p.stride - p.pad_t + ky;
                    if ((unsigned)y >= (unsigned)p.H) continue;
                    for (int kx = 0; kx < p.k; ++kx) {
                        const int xx = ox * p.stride - p.pad_l + kx;
                        if ((unsigned)xx >= (unsigned)p.W) continue;
                        const float v = xb[((long)y * p.W + xx) * p.C];
                        if (v > best || by < 0) { best = v; by = y; bx = xx; }
                    }
                }
                if (by == iy && bx == ix) acc += p.g[(((long)b * p.Ho + oy) * p.Wo + ox) * p.C + c];
            }
        p.dx[e] = p.accumulate ? p.dx[e] + acc : acc;
    }
}

// y = x * r * gamma, r = rsqrt(max(sum_c x^2, 1e-12)).  One wave per pixel:
//   dx = gamma*dy*r - x * r^3 * sum_c(gamma*dy*x)   (second term only while sum x^2 > 1e-12)
//   tmp = dy * x * r  (column sums of tmp = d gamma)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ gamma, const long pixels,
                                                        const int C, float* __restrict__ dx, const int accumulate,
                                                        float* __restrict__ tmp) {
    const int lane = threadIdx.x & 63;
    const long wave0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * 256) >> 6;
    for (long px = wave0; px < pixels; px += nwaves) {
        const float* xr = x + px * C;
        const float* gr = dy + px * C;
        float s = 0.f, t = 0.f;
        for (int c = lane; c < C; c += 64) {
            s += xr[c] * xr[c];
            t += gamma[c] * gr[c] * xr[c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); t += __shfl_xor(t, o); }
        const float r = 1.0f / sqrtf(fmaxf(s, 1e-12f));
        const float k3 = s > 1e-12f ? r * r * r * t : 0.f;
        for (int c = lane; c < C; c += 64) {
            const float v = gamma[c] * gr[c] * r - xr[c] * k3;
            dx[px * C + c] = accumulate ? dx[px * C + c] + v : v;
            tmp[px * C + c] = gr[c] * xr[c] * r;
        }
    }
}

// g += coef * w   (Keras l2 kernel regulariser: d(l2 * sum w^2)/dw = 2 * l2 * w)
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ g, const float* __restrict__ w, const long n,
                                                  const float coef) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) g[e] += coef * w[e];
}

// partial[blockIdx.x] = sum of w^2 over this block's grid-stride slice (fixed slices, fixed in-block tree:
// deterministic); reg_finalize sums the partials in order and applies the l2 factor
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ w, const long n, float* __restrict__ partial) {
    __shared__ float sh[256];
    float s = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) s += w[e] * w[e];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void reg_finalize_kernel(const float* __restrict__ partial, const int n, const float l2, float* out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)partial[i];
    *out = (float)(s * (double)l2);
}

// ------------------------------------------------------------------ Adam (TF training_ops.ApplyAdam form)
// m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); var -= alpha * m / (sqrt(v) + eps),
// alpha = lr * sqrt(1 - b2^t) / (1 - b1^t); g = grad * grad_scale (+ l2 * var where l2mask set)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ var, float* __restrict__ m, float* __restrict__ v,
                                                  const float* __restrict__ grad, const long n, const float alpha,
                                                  const float b1, const float b2, const float eps,
                                                  const float grad_scale) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const float g = grad[e] * grad_scale;
        const float mm = m[e] + (g - m[e]) * (1.0f - b1);
        const float vv = v[e] + (g * g - v[e]) * (1.0f - b2);
        m[e] = mm;
        v[e] = vv;
        var[e] -= alpha * mm / (sqrtf(vv) + eps);
    }
}

static inline int grid_for(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace ssd

// ====================================================================== training state
struct TrainLayer {
    float* pre = nullptr;        // BN layers: conv / depthwise output before BatchNorm [M][C]
    float* mean = nullptr;       // [C] batch statistics of the last forward
    float* var = nullptr;
    float* istd = nullptr;
    float* wfwd = nullptr;       // packed forward weights (dense convs)
    float* wbwd = nullptr;       // packed backward-data weights (rotated + transposed), nullptr: no dgrad
    int cpad = 0;                // dY channel count the backward-data conv sees (heads: padded to 32)
    long g_kernel = -1, g_bias = -1, g_gamma = -1, g_beta = -1, g_kernel2 = -1, g_bias2 = -1;
    bool active = false;
};

struct ssd_train_state {
    int batch = 0;
    size_t P = 0;
    int precision = 0;                  // the net's precision the weight re-pack jobs were planned for
    double step_flops[3] = {0, 0, 0};   // matrix-core FLOPs of the last forward_backward: fp32-MFMA convs, split-bf16 convs, weight gradients
    float *flat = nullptr, *m = nullptr, *v = nullptr;
    std::vector<long> poff;             // parameter -> offset in the flat trainable vector (-1: not trainable)
    std::vector<float*> act;            // training activations per tensor (own arena: no finalize needed)
    std::vector<float*> gact;           // gradient w.r.t. each activation tensor
    std::vector<char> gwritten;
    std::vector<TrainLayer> tl;
    std::vector<float*> owned;
    float *scratch_dy = nullptr, *scratch_dz = nullptr, *scratch_w = nullptr, *partial = nullptr;
    // weight gradients on a side stream beside the rest of the backward (dX chain, BatchNorm backward of the layers below):
    // dY alternates between two buffers (a buffer is rewritten only after the weight gradient that reads it is done), the
    // weight-gradient partial slab is its own
    float* dy_buf[2] = {nullptr, nullptr};
    float* partial_w = nullptr;
    size_t partial_w_floats = 0;
    hipStream_t wstream = nullptr;
    hipEvent_t ev_dy = nullptr, ev_wdone[2] = {nullptr, nullptr}, ev_wall = nullptr, ev_main_pos = nullptr;
    bool wpending[2] = {false, false};
    float* pack_jobs = nullptr;  // device array of PackJob (one launch re-packs every conv weight)
    int n_pack_jobs = 0;
    size_t partial_floats = 0;
    float *dgamma_tmp = nullptr, *dbeta_tmp = nullptr;
    float *deltas = nullptr, *probs = nullptr, *gdeltas = nullptr, *glogits = nullptr;
    void* loss_ws = nullptr;
    size_t loss_ws_bytes = 0;
    long step = 0;
    // gradient buckets for the data-parallel exchange (ssd_net_train_set_buckets): bucket k = flat offsets
    // [bucket_lo[k], bucket_lo[k + 1]) and is FINAL once the backward has passed every layer that owns a
    // parameter at or above bucket_lo[k]; bucket_ev[k] is recorded on the backward's stream right there
    std::vector<long> bucket_lo;
    std::vector<hipEvent_t> bucket_ev;
    std::vector<long> pending_hi;       // per layer i: end of the highest parameter of any ACTIVE layer j < i (0: none)
};

void ssd_train_state_free(ssd_train_state* s) {
    if (!s) return;
    for (auto e : s->bucket_ev)
        if (e) (void)hipEventDestroy(e);
    if (s->wstream) { (void)hipStreamSynchronize(s->wstream); (void)hipStreamDestroy(s->wstream); }
    for (hipEvent_t e : {s->ev_dy, s->ev_wdone[0], s->ev_wdone[1], s->ev_wall, s->ev_main_pos})
        if (e) (void)hipEventDestroy(e);
    for (float* p : s->owned)
        if (p) (void)hipFree(p);
    if (s->loss_ws) (void)hipFree(s->loss_ws);
    delete s;
}

namespace ssd {

static int talloc(ssd_train_state& s, size_t floats, float** out) {
    *out = nullptr;
    if (!floats) return SSD_OK;
    SSD_HIP(hipMalloc((void**)out, floats * sizeof(float)));
    s.owned.push_back(*out);
    return SSD_OK;
}

static bool trainable(const std::string& name) {
    const std::string var = name.substr(name.rfind('/') + 1);
    return var != "moving_mean" && var != "moving_variance";
}

// layers the training step executes: the un-fused layer list (fused kernels fold inference BatchNorm)
static bool train_runs(const Layer& l) { return l.kind == LK_CONV || l.kind == LK_DW || l.kind == LK_POOL || l.kind == LK_L2NORM; }

static long chunks_for(long M, long unit_blocks, long* rows_per_chunk, long min_rows, long max_chunks) {
    long chunks = (2048 + unit_blocks - 1) / unit_blocks;          // ~8 workgroups per CU in total
    const long maxc = (M + min_rows - 1) / min_rows;
    if (chunks > maxc) chunks = maxc;
    if (chunks > max_chunks) chunks = max_chunks;                    // second-stage sums walk the chunks serially
    if (chunks < 1) chunks = 1;
    long rpc = (M + chunks - 1) / chunks;
    rpc = (rpc + 31) / 32 * 32;
    *rows_per_chunk = rpc;
    return (M + rpc - 1) / rpc;
}

static int ensure_partial(ssd_train_state& s, size_t floats) {
    if (floats <= s.partial_floats) return SSD_OK;
    // grow-only; the old slab stays owned until the state is freed (sizes are planned once per batch)
    float* p = nullptr;
    int rc = talloc(s, floats, &p);
    if (rc) return rc;
    s.partial = p;
    s.partial_floats = floats;
    return SSD_OK;
}

template <int OP>
static int col_reduce(ssd_train_state& s, RedParams p, long* chunks_out, hipStream_t st) {
    const bool vec = p.C % 4 == 0 && p.lda % 4 == 0 && (((uintptr_t)p.a | (uintptr_t)p.b) & 15) == 0;
    if (vec) {
        // channel quads per block: a power of two <= 64 that wastes few lanes (C / 4 = 4 .. 320)
        const int q = p.C / 4;
        p.cw = q % 64 == 0 ? 64 : (q % 32 == 0 ? 32 : (q % 16 == 0 ? 16 : (q % 8 == 0 ? 8 : (q < 8 ? 4 : (q < 48 ? 16 : 64)))));
        const int ctiles = (q + p.cw - 1) / p.cw;
        long rpc = 0;
        const long chunks = chunks_for(p.M, ctiles, &rpc, 64, 256);
        int rc = ensure_partial(s, (size_t)chunks * 2 * p.C);
        if (rc) return rc;
        p.rows_per_chunk = rpc;
        p.partial = s.partial;
        hipLaunchKernelGGL(col_reduce4_kernel<OP>, dim3(ctiles, (unsigned)chunks), dim3(256), 0, st, p);
        SSD_LAUNCH_CHECK();
        *chunks_out = chunks;
        return SSD_OK;
    }
    SSD_UNSUPPORTED_IF(OP == RED_STATS, "train: batch statistics need C %% 4 == 0 (C = %d)", p.C);
    p.cw = p.C % 64 == 0 ? 64 : (p.C % 32 == 0 ? 32 : (p.C % 16 == 0 ? 16 : (p.C < 64 ? 32 : 64)));
    const int ctiles = (p.C + p.cw - 1) / p.cw;
    long rpc = 0;
    const long chunks = chunks_for(p.M, ctiles, &rpc, 64, 256);
    int rc = ensure_partial(s, (size_t)chunks * 2 * p.C);
    if (rc) return rc;
    p.rows_per_chunk = rpc;
    p.partial = s.partial;
    hipLaunchKernelGGL(col_reduce_kernel<OP>, dim3(ctiles, (unsigned)chunks), dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    *chunks_out = chunks;
    return SSD_OK;
}
static int col_finalize(ssd_train_state& s, long chunks, int C, float scale, float* out1, float* out2, int mode,
                        hipStream_t st) {
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, s.partial, (int)chunks, C, scale,
                       out1, out2, mode, kBnEps);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// batch statistics of pre [M][C] -> mean, var (biased), istd, and the moving-average update (unbiased variance): one
// pass over `pre` (shifted sums, see col_reduce4_kernel) + one finalize launch
static int bn_stats(ssd_train_state& s, TrainLayer& t, long M, int C, float* moving_mean, float* moving_var,
                    hipStream_t st) {
    RedParams p{};
    p.a = t.pre; p.M = M; p.C = C; p.lda = C;
    long chunks = 0;
    const float bessel = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.0f;
    if (C % 4 == 0 && ((uintptr_t)t.pre & 15) == 0) {
        int rc = col_reduce<RED_STATS>(s, p, &chunks, st);
        if (rc) return rc;
        hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, s.partial, (int)chunks, C,
                           1.0f / (float)M, t.mean, t.var, 2, kBnEps, t.pre, t.istd, moving_mean, moving_var,
                           1.0f - kBnMomentum, bessel);
        SSD_LAUNCH_CHECK();
        return SSD_OK;
    }
    int rc = col_reduce<RED_SUM>(s, p, &chunks, st);
    if (!rc) rc = col_finalize(s, chunks, C, 1.0f / (float)M, t.mean, nullptr, 0, st);
    if (rc) return rc;
    p.mean = t.mean;
    rc = col_reduce<RED_SQDEV>(s, p, &chunks, st);
    if (!rc) rc = col_finalize(s, chunks, C, 1.0f / (float)M, t.var, t.istd, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(moving_update_kernel, dim3((C + 255) / 256), dim3(256), 0, st, moving_mean, moving_var, t.mean, t.var, C,
                       1.0f - kBnMomentum, bessel);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

static ConvParams dense_conv_params(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int dil,
                                    int pt, int pl, int Ho, int Wo) {
    ConvParams p{};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
    p.kh = kh; p.kw = kw; p.stride = stride; p.dil = dil; p.pad_t = pt; p.pad_l = pl;
    p.K = kh * kw * Cin;
    p.Kpad = conv_kpad(p.K);
    p.Npad = conv_npad(Cout);
    p.M = (long)B * Ho * Wo;
    p.split_k = 1;
    p.out_pixel_stride = Cout;
    p.out_batch_stride = (long)Ho * Wo * Cout;
    return p;
}

// matrix-core FLOPs issued by the step being built, per instruction family (bench.py prices each at its own peak):
// [0] conv forward / backward-data on fp32-MFMA tiles, [1] on split-bf16 tiles, [2] weight gradients (fp32 MFMA)
static thread_local double g_step_flops[3] = {0, 0, 0};

// Tile choice of the training convs (forward and backward-data).  Default (SSD_HIP_TRAIN_AUTOTUNE unset or 1): the first
// time a conv shape is seen every valid tile of the net's precision is TIMED on the device (into a scratch output: a
// residual / accumulate-in-place epilogue must not run twice on the real tensor) and the fastest is kept for the process
// (measured: MobileNetV2 B=32 12.42 -> 11.15 ms per step, bf16 11.69 -> 10.70, VGG16 B=16 31.0 -> 28.8: the cost model
// favours large tiles that the short-K layers of a 32-image batch cannot fill).  Every step of a process runs the same
// tiles; ACROSS processes the choice -- and with it the last bits of a step -- may differ with the box (training has no
// bit-exactness contract; the inference path keeps its shipped tables).  SSD_HIP_TRAIN_AUTOTUNE=0: the cost model
// (conv_pick_config: deterministic); =2 also prints what was found.
struct TrainPickKey {
    long M, ops, obs; int K, Cout, kh, kw, stride, dil, H, W, Cin, flags, pad_t, pad_l;
    bool operator<(const TrainPickKey& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
static std::map<TrainPickKey, int>& train_picks() { static std::map<TrainPickKey, int> m; return m; }
static std::mutex& train_picks_mutex() { static std::mutex m; return m; }      // nets of several host threads share the memo

static int pick_measured(const ConvParams& p, hipStream_t st, int model_cfg, int verbose) {
    TrainPickKey k{};
    memset(&k, 0, sizeof(k));
    k.M = p.M; k.K = p.K; k.Cout = p.Cout; k.kh = p.kh; k.kw = p.kw; k.stride = p.stride; k.dil = p.dil; k.H = p.H; k.W = p.W;
    k.Cin = p.Cin; k.pad_t = p.pad_t; k.pad_l = p.pad_l; k.ops = p.out_pixel_stride; k.obs = p.out_batch_stride;
    k.flags = (p.residual ? 1 : 0) | (p.n_split ? 2 : 0) | (p.scale ? 4 : 0) | (p.shift ? 8 : 0) | (p.act << 4) | (p.bf16 << 8) |
              (p.vec_store << 9);
    {
        std::lock_guard<std::mutex> lock(train_picks_mutex());
        auto it = train_picks().find(k);
        // (a memoised pick is re-validated against THESE parameters: alignment of the pointers is not part of the key)
        if (it != train_picks().end()) return conv_config_valid(it->second, p) ? it->second : model_cfg;
    }
    float* scratch = nullptr;
    const size_t out_floats = (size_t)p.M * (size_t)(p.out_pixel_stride > p.Cout ? p.out_pixel_stride : p.Cout) + 4096;
    if (hipMalloc((void**)&scratch, out_floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return model_cfg; }
    ConvParams q = p;
    q.out = scratch;
    if (q.n_split) { q.out2 = scratch; q.n_split = 0; }          // (timing only: one destination)
    q.out_batch_stride = (long)p.Ho * p.Wo * q.out_pixel_stride;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int best = model_cfg;
    float best_ms = 1e30f, model_ms = 0.f;
    for (int c = 0; c < conv_num_configs(); ++c) {
        if (!conv_config_valid(c, q) || !conv_config_allowed(c, q.bf16)) continue;
        const char* cn = conv_config_name(c);
        if (!strncmp(cn, "wino_", 5) || !strncmp(cn, "skinny_", 7) || (!strncmp(cn, "direct", 6) && c != model_cfg)) continue;
        if (conv_launch(q, c, st)) { (void)hipGetLastError(); continue; }
        float ms = 1e30f;
        for (int trial = 0; trial < 3; ++trial) {
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 3; ++r) (void)conv_launch(q, c, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms = t < ms ? t : ms;
            if (trial == 0 && ms > 2.0f * best_ms) break;
        }
        if (c == model_cfg) model_ms = ms;
        if (ms < best_ms) { best_ms = ms; best = c; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(scratch);
    if (verbose)
        fprintf(stderr, "[ssd train tune] M=%ld K=%d N=%d k%dx%d s%d: model %s %.1f us -> %s %.1f us\n", p.M, p.K, p.Cout, p.kh, p.kw,
                p.stride, conv_config_name(model_cfg), model_ms * 1000.f / 3, conv_config_name(best), best_ms * 1000.f / 3);
    {
        std::lock_guard<std::mutex> lock(train_picks_mutex());
        train_picks()[k] = best;
    }
    return best;
}

static int launch_conv(ConvParams& p, hipStream_t st) {
    p.vec_store = (((uintptr_t)p.out & 15) == 0) && (p.out_pixel_stride % 4 == 0) && (p.out_batch_stride % 4 == 0);
    int cfg = conv_pick_config(p);
    SSD_UNSUPPORTED_IF(cfg < 0, "train: no conv kernel for Cin=%d Cout=%d k=%dx%d", p.Cin, p.Cout, p.kh, p.kw);
    static const int tune = getenv("SSD_HIP_TRAIN_AUTOTUNE") ? atoi(getenv("SSD_HIP_TRAIN_AUTOTUNE")) : 1;
    if (tune) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs == hipStreamCaptureStatusNone) cfg = pick_measured(p, st, cfg, tune > 1);
    }
    const char* cn = conv_config_name(cfg);
    g_step_flops[(strncmp(cn, "mfma3_", 6) == 0 || strncmp(cn, "bf16_", 5) == 0) ? 1 : 0] += 2.0 * (double)p.M * p.K * p.Cout;
    return conv_launch(p, cfg, st);
}

// dW [K][N] (row-major, = Keras HWIO) = im2col(X)^T * G with tile shape `cfg`
static int wgrad_with(ssd_train_state& s, const Layer& l, int B, const float* x, const float* g, int ldg, int N, float* dW,
                      hipStream_t st, const WgradCfg* cfg, bool count) {
    WgradParams p{};
    p.x = x; p.g = g;
    p.B = B; p.H = l.H; p.W = l.W; p.Cin = l.Cin; p.Ho = l.Ho; p.Wo = l.Wo;
    p.kh = l.kh; p.kw = l.kw; p.stride = l.stride; p.dil = l.dil; p.pad_t = l.pt; p.pad_l = l.pl;
    p.N = N; p.ldg = ldg; p.K = l.kh * l.kw * l.Cin;
    p.M = (long)B * l.Ho * l.Wo;
    p.im2col = l.Cin % 4 != 0;            // RGB stems (Cin = 3): one tile spans all K = 27 rows instead of 3 of 16 per tap
    const int rows = p.im2col ? p.K : l.Cin;
    p.ctiles = (rows + cfg->tc - 1) / cfg->tc;
    p.ntiles = (N + cfg->tn - 1) / cfg->tn;
    p.vec_x = (l.Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    p.vec_g = (ldg % 4 == 0) && (((uintptr_t)g & 15) == 0);
    const long tiles = (p.im2col ? 1L : (long)l.kh * l.kw) * p.ctiles * p.ntiles;
    long rpc = 0;
    long chunks = chunks_for(p.M, tiles, &rpc, 128, 1024);
    // bound the slab: chunks * K * N floats
    const size_t kn = (size_t)p.K * N;
    while (chunks > 1 && (size_t)chunks * kn > ((size_t)96 << 20)) {
        rpc *= 2;
        chunks = (p.M + rpc - 1) / rpc;
    }
    if ((size_t)chunks * kn > s.partial_w_floats) {      // grow-only, own slab (weight gradients may run beside the reductions of the main chain)
        float* np = nullptr;
        int rca = talloc(s, (size_t)chunks * kn, &np);
        if (rca) return rca;
        s.partial_w = np;
        s.partial_w_floats = (size_t)chunks * kn;
    }
    p.rows_per_chunk = rpc;
    p.partial = s.partial_w;
    hipLaunchKernelGGL(cfg->fn, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    if (count) g_step_flops[2] += 2.0 * (double)p.M * p.K * N;
    return chunk_sum(s.partial_w, chunks, (long)kn, dW, st);
}

// Tile shape of a weight gradient.  Heuristic: least padded tile area first (MFMA work); among equals the shape that
// stages the fewest floats per M row (every tile re-reads its 32-row slabs of X and G: ctiles * ntiles * (tc + tn)).
// Measured: by staged floats alone 128 x 128 wins everywhere, +4 % on VGG16 (512-channel layers, no padding) but -3 % on
// MobileNetV2 (96 -> 576 expands padded to 128 x 640); the lexicographic rule keeps both.  With SSD_HIP_TRAIN_AUTOTUNE
// (default on, see launch_conv) every shape of the table is timed the first time a layer is seen (dW is written, not
// accumulated: running it repeatedly is harmless) and the fastest kept for the process.
static int wgrad(ssd_train_state& s, const Layer& l, int B, const float* x, const float* g, int ldg, int N, float* dW,
                 hipStream_t st) {
    const bool im2col = l.Cin % 4 != 0;
    const int rows = im2col ? l.kh * l.kw * l.Cin : l.Cin;
    const WgradCfg* cfg = &kWgrad[0];
    {
        long best_area = -1, best_staged = -1;
        for (const auto& c : kWgrad) {
            const long ct = (rows + c.tc - 1) / c.tc, nt = (N + c.tn - 1) / c.tn;
            const long area = ct * c.tc * nt * c.tn, staged = ct * nt * (c.tc + c.tn);
            if (best_area < 0 || area < best_area || (area == best_area && staged < best_staged)) {
                best_area = area; best_staged = staged; cfg = &c;
            }
        }
    }
    static const int tune = getenv("SSD_HIP_TRAIN_AUTOTUNE") ? atoi(getenv("SSD_HIP_TRAIN_AUTOTUNE")) : 1;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (tune && cs == hipStreamCaptureStatusNone) {
        TrainPickKey k{};
        memset(&k, 0, sizeof(k));
        k.M = (long)B * l.Ho * l.Wo; k.K = l.kh * l.kw * l.Cin; k.Cout = N; k.kh = l.kh; k.kw = l.kw; k.stride = l.stride;
        k.dil = l.dil; k.H = l.H; k.W = l.W; k.Cin = l.Cin; k.flags = (1 << 20) | ldg;      // bit 20: a weight gradient
        auto it = train_picks().find(k);
        if (it != train_picks().end()) {
            cfg = &kWgrad[it->second];
        } else {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            int best = (int)(cfg - kWgrad);
            float best_ms = 1e30f, model_ms = 0.f;
            for (int c = 0; c < (int)(sizeof(kWgrad) / sizeof(kWgrad[0])); ++c) {
                if (wgrad_with(s, l, B, x, g, ldg, N, dW, st, &kWgrad[c], false)) { (void)hipGetLastError(); continue; }
                float ms = 1e30f;
                for (int trial = 0; trial < 3; ++trial) {
                    (void)hipEventRecord(e0, st);
                    for (int r = 0; r < 2; ++r) (void)wgrad_with(s, l, B, x, g, ldg, N, dW, st, &kWgrad[c], false);
                    (void)hipEventRecord(e1, st);
                    (void)hipEventSynchronize(e1);
                    float t = 0.f;
                    (void)hipEventElapsedTime(&t, e0, e1);
                    ms = t < ms ? t : ms;
                    if (trial == 0 && ms > 2.0f * best_ms) break;
                }
                if (&kWgrad[c] == cfg) model_ms = ms;
                if (ms < best_ms) { best_ms = ms; best = c; }
            }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            if (tune > 1)
                fprintf(stderr, "[ssd train tune] wgrad M=%ld K=%d N=%d: heuristic %dx%d %.1f us -> %dx%d %.1f us\n", k.M, k.K, N, cfg->tc,
                        cfg->tn, model_ms * 500.f, kWgrad[best].tc, kWgrad[best].tn, best_ms * 500.f);
            train_picks()[k] = best;
            cfg = &kWgrad[best];
        }
    }
    return wgrad_with(s, l, B, x, g, ldg, N, dW, st, cfg, true);
}

}  // namespace ssd

extern "C" {

size_t ssd_net_trainable_floats(const ssd_net* net) {
    if (!net) return 0;
    size_t n = 0;
    for (const auto& p : net->params)
        if (trainable(p.name)) n += p.count;
    return n;
}

long ssd_net_trainable_offset(const ssd_net* net, const char* name) {
    if (!net || !name) return -1;
    long off = 0;
    for (const auto& p : net->params) {
        if (!trainable(p.name)) continue;
        if (p.name == name) return off;
        off += (long)p.count;
    }
    return -1;
}

int ssd_net_train_begin(ssd_net* net, int batch) {
    SSD_CHECK_ARG(net && batch >= 1, "ssd_net_train_begin: bad arguments");
    for (const auto& p : net->params)
        if (!p.set) {
            set_error("ssd_net_train_begin: parameter '%s' was never set", p.name.c_str());
            return SSD_E_STATE;
        }
    if (net->train && net->train->batch >= batch) return SSD_OK;
    // keep optimiser state across a re-plan for a larger batch
    ssd_train_state* old = net->train;
    auto* s = new ssd_train_state();
    s->batch = batch;
    int rc = SSD_OK;
    // ---- flat parameter vector (trainable parameters in table order); params point into it
    s->P = ssd_net_trainable_floats(net);
    rc = talloc(*s, s->P, &s->flat);
    if (!rc) rc = talloc(*s, s->P, &s->m);
    if (!rc) rc = talloc(*s, s->P, &s->v);
    if (rc) { ssd_train_state_free(s); return rc; }
    s->poff.assign(net->params.size(), -1);
    {
        long off = 0;
        for (size_t i = 0; i < net->params.size(); ++i) {
            if (!trainable(net->params[i].name)) continue;
            s->poff[i] = off;
            off += (long)net->params[i].count;
        }
    }
    // ---- activations, activation gradients
    s->act.assign(net->tensors.size(), nullptr);
    s->gact.assign(net->tensors.size(), nullptr);
    s->gwritten.assign(net->tensors.size(), 0);
    for (size_t i = 1; i < net->tensors.size() && !rc; ++i) {
        rc = talloc(*s, net->tensors[i].per_image * batch, &s->act[i]);
        if (!rc) rc = talloc(*s, net->tensors[i].per_image * batch, &s->gact[i]);
    }
    // ---- per-layer buffers
    s->tl.assign(net->layers.size(), TrainLayer());
    size_t max_out = 0, max_dz = 0, max_w = 0;
    int maxC = 0;
    for (size_t i = 0; i < net->layers.size() && !rc; ++i) {
        const Layer& l = net->layers[i];
        if (!train_runs(l)) continue;
        TrainLayer& t = s->tl[i];
        t.active = true;
        const size_t mc = (size_t)batch * l.Ho * l.Wo * l.Cout;
        maxC = std::max(maxC, l.Cout);
        if (l.kind == LK_POOL) continue;
        if (l.kind == LK_L2NORM) {
            t.g_gamma = s->poff[l.p_gamma];
            max_out = std::max(max_out, mc);
            continue;
        }
        t.g_kernel = s->poff[l.p_kernel];
        if (l.p_bias >= 0) t.g_bias = s->poff[l.p_bias];
        if (l.p_kernel2 >= 0) { t.g_kernel2 = s->poff[l.p_kernel2]; t.g_bias2 = s->poff[l.p_bias2]; }
        if (l.p_bn >= 0) {
            t.g_gamma = s->poff[l.p_bn];
            t.g_beta = s->poff[l.p_bn + 1];
            rc = talloc(*s, mc, &t.pre);
            if (!rc) rc = talloc(*s, l.Cout, &t.mean);
            if (!rc) rc = talloc(*s, l.Cout, &t.var);
            if (!rc) rc = talloc(*s, l.Cout, &t.istd);
        }
        if (l.kind == LK_CONV) {
            const int K = l.kh * l.kw * l.Cin;
            if (!rc) rc = talloc(*s, conv_packed_floats(K, l.Cout), &t.wfwd);
            t.cpad = l.head_kind ? round_up(l.Cout, 32) : l.Cout;
            if (l.in != 0) {        // the image needs no gradient
                const int Kb = l.kh * l.kw * t.cpad;
                if (!rc) rc = talloc(*s, conv_packed_floats(Kb, l.Cin), &t.wbwd);
                max_w = std::max(max_w, (size_t)l.kh * l.kw * t.cpad * l.Cin);
            }
            size_t dy = (size_t)batch * l.Ho * l.Wo * t.cpad;
            max_out = std::max(max_out, dy);
            if (l.stride == 2 && l.in != 0) {
                const int Hz = (l.Ho - 1) * 2 + 1, Wz = (l.Wo - 1) * 2 + 1;
                max_dz = std::max(max_dz, (size_t)batch * Hz * Wz * l.Cout);
            }
            SSD_UNSUPPORTED_IF(l.stride > 2 || (l.stride == 2 && l.dil != 1), "train: unsupported conv geometry in %s",
                               l.name.c_str());
        } else {
            max_out = std::max(max_out, mc);
        }
    }
    // one PackJob per forward / backward-data weight matrix (the parameters will live at flat + poff)
    std::vector<PackJob> jobs;
    for (size_t i = 0; i < net->layers.size() && !rc; ++i) {
        const Layer& l = net->layers[i];
        const TrainLayer& t = s->tl[i];
        if (!t.active || l.kind != LK_CONV) continue;
        PackJob j{};
        j.bf16 = net->precision;
        j.w = s->flat + s->poff[l.p_kernel];
        j.w2 = l.p_kernel2 >= 0 ? s->flat + s->poff[l.p_kernel2] : nullptr;
        j.K = l.kh * l.kw * l.Cin;
        j.Cout = l.Cout;
        j.Cout1 = l.p_kernel2 >= 0 ? l.Cout1 : l.Cout;
        j.kh = l.kh; j.kw = l.kw; j.Ci = l.Cin; j.CoPad = t.cpad;
        j.mode = 0; j.dst = t.wfwd; j.Kpad = conv_kpad(j.K); j.Npad = conv_npad(l.Cout);
        jobs.push_back(j);
        if (t.wbwd) {
            const int Kb = l.kh * l.kw * t.cpad;
            j.mode = 1; j.dst = t.wbwd; j.Kpad = conv_kpad(Kb); j.Npad = conv_npad(l.Cin);
            jobs.push_back(j);
        }
    }
    for (const PackJob& j : jobs)
        SSD_UNSUPPORTED_IF((long)j.Npad * j.Kpad > 0x7fffffffL, "train: a packed weight matrix of %d x %d exceeds 32-bit indexing", j.Npad, j.Kpad);
    s->n_pack_jobs = (int)jobs.size();
    s->precision = net->precision;
    if (!rc && !jobs.empty()) rc = talloc(*s, (jobs.size() * sizeof(PackJob) + 3) / 4, &s->pack_jobs);
    if (!rc) rc = talloc(*s, max_out, &s->scratch_dy);
    if (!rc) rc = talloc(*s, max_out, &s->dy_buf[1]);
    s->dy_buf[0] = s->scratch_dy;
    if (!rc && getenv("SSD_HIP_TRAIN_WGRAD_STREAM") && atoi(getenv("SSD_HIP_TRAIN_WGRAD_STREAM")) == 0) {
        s->wstream = nullptr;                       // diagnostics: weight gradients in line on the caller's stream
    } else if (!rc) {
        if (hipStreamCreateWithFlags(&s->wstream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_dy, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_wdone[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_wdone[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_wall, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_main_pos, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            set_error("ssd_net_train_begin: side stream / events for the weight gradients could not be created");
            rc = SSD_E_HIP;
        }
    }
    if (!rc) rc = talloc(*s, max_dz, &s->scratch_dz);
    if (!rc) rc = talloc(*s, max_w, &s->scratch_w);
    if (!rc) rc = talloc(*s, maxC, &s->dgamma_tmp);
    if (!rc) rc = talloc(*s, maxC, &s->dbeta_tmp);
    const size_t N = net->num_priors;
    if (!rc) rc = talloc(*s, (size_t)batch * N * 4, &s->deltas);
    if (!rc) rc = talloc(*s, (size_t)batch * N * net->L, &s->probs);
    if (!rc) rc = talloc(*s, (size_t)batch * N * 4, &s->gdeltas);
    if (!rc) rc = talloc(*s, (size_t)batch * N * net->L, &s->glogits);
    if (!rc) {
        s->loss_ws_bytes = ssd_loss_workspace_bytes(batch, (int)N);
        if (hipMalloc(&s->loss_ws, s->loss_ws_bytes) != hipSuccess) {
            set_error("ssd_net_train_begin: loss workspace allocation failed");
            rc = SSD_E_HIP;
        }
    }
    if (rc) {                           // nothing of the net was touched yet: the previous state (if any) stays valid
        ssd_train_state_free(s);
        return rc;
    }
    // ---- every allocation succeeded: adopt the optimiser state and move the parameters
    bool copy_failed = false;
    if (old) {
        copy_failed |= hipMemcpy(s->m, old->m, s->P * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess;
        copy_failed |= hipMemcpy(s->v, old->v, s->P * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess;
        s->step = old->step;
    } else {
        copy_failed |= hipMemset(s->m, 0, s->P * sizeof(float)) != hipSuccess;
        copy_failed |= hipMemset(s->v, 0, s->P * sizeof(float)) != hipSuccess;
    }
    {
        long off = 0;
        for (size_t i = 0; i < net->params.size() && !copy_failed; ++i) {
            if (s->poff[i] < 0) continue;
            copy_failed |= hipMemcpy(s->flat + off, net->params[i].dev, net->params[i].count * sizeof(float),
                                     hipMemcpyDeviceToDevice) != hipSuccess;
            off += (long)net->params[i].count;
        }
    }
    if (s->n_pack_jobs)
        copy_failed |= hipMemcpy(s->pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice) != hipSuccess;
    if (copy_failed) {
        set_error("ssd_net_train_begin: device copy failed");
        ssd_train_state_free(s);
        return SSD_E_HIP;
    }
    for (size_t i = 0; i < net->params.size(); ++i) {
        Param& p = net->params[i];
        if (s->poff[i] < 0) continue;
        if (!p.in_flat) (void)hipFree(p.dev);
        p.dev = s->flat + s->poff[i];
        p.in_flat = true;
    }
    if (old) ssd_train_state_free(old);
    net->train = s;
    net->finalized = false;          // derived inference weights must be rebuilt from the (moved) parameters
    net->drop_graphs();
    return SSD_OK;
}

// Training-mode forward + loss + backward.  grads_flat_dev [ssd_net_trainable_floats] receives
// d(mean_b (loc_b + conf_b)) / d(parameter) in parameter-table order (Keras layouts);
// loc_loss_dev / conf_loss_dev [B] receive the per-image loss terms.
int ssd_net_train_forward_backward(ssd_net* net, const float* image_dev, int B, const float* actual_deltas_dev,
                                   const float* actual_labels_dev, float neg_pos_ratio, float loc_loss_alpha,
                                   float* grads_flat_dev, float* loc_loss_dev, float* conf_loss_dev, void* stream) {
    SSD_CHECK_ARG(net && image_dev && actual_deltas_dev && actual_labels_dev && grads_flat_dev,
                  "ssd_net_train_forward_backward: NULL argument");
    if (!net->train) { set_error("ssd_net_train_forward_backward: call ssd_net_train_begin() first"); return SSD_E_STATE; }
    ssd_train_state& s = *net->train;
    SSD_CHECK_ARG(B >= 1 && B <= s.batch, "ssd_net_train_forward_backward: batch %d exceeds the planned %d", B, s.batch);
    if (s.precision != net->precision) {
        set_error("ssd_net_train_forward_backward: the precision option changed since ssd_net_train_begin (the weight re-pack was planned for %d): call ssd_net_train_begin again", s.precision);
        return SSD_E_STATE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int N = net->num_priors, L = net->L;
    s.act[0] = const_cast<float*>(image_dev);
    int rc = SSD_OK;
    g_step_flops[0] = g_step_flops[1] = g_step_flops[2] = 0;
    struct FlopsOut {           // whatever path returns: the state holds what this call issued
        ssd_train_state& s;
        ~FlopsOut() { for (int i = 0; i < 3; ++i) s.step_flops[i] = g_step_flops[i]; }
    } flops_out{s};

    // re-pack the (just updated) weights of every conv: forward and backward-data forms, one launch
    if (s.n_pack_jobs) {
        hipLaunchKernelGGL(pack_jobs_kernel, dim3(128, (unsigned)s.n_pack_jobs), dim3(256), 0, st,
                           reinterpret_cast<const PackJob*>(s.pack_jobs));
        SSD_LAUNCH_CHECK();
    }
    // ------------------------------------------------------------ forward (training mode)
    for (size_t i = 0; i < net->layers.size(); ++i) {
        const Layer& l = net->layers[i];
        TrainLayer& t = s.tl[i];
        if (!t.active) continue;
        const long M = (long)B * l.Ho * l.Wo;
        const float* x = s.act[l.in];
        if (l.kind == LK_POOL) {
            rc = launch_maxpool(x, B, l.H, l.W, l.Cin, l.kh, l.stride, l.pt, l.pl, l.Ho, l.Wo, s.act[l.out], st);
            if (rc) return rc;
            continue;
        }
        if (l.kind == LK_L2NORM) {
            rc = launch_l2norm(x, M, l.Cin, net->params[l.p_gamma].dev, s.act[l.out], st);
            if (rc) return rc;
            continue;
        }
        if (l.kind == LK_CONV) {
            ConvParams p = dense_conv_params(B, l.H, l.W, l.Cin, l.Cout, l.kh, l.kw, l.stride, l.dil, l.pt, l.pl, l.Ho, l.Wo);
            p.in = x;
            p.w = t.wfwd;
            p.w3 = conv_split_planes(t.wfwd, l.kh * l.kw * l.Cin, l.Cout);
            p.bf16 = net->precision;          // precision 1: the cost model takes the bf16 (one-product) tiles
            if (l.p_bn >= 0) {
                p.out = t.pre;
                p.act = SSD_ACT_NONE;
            } else {
                p.shift = l.p_bias2 >= 0 ? nullptr : net->params[l.p_bias].dev;
                p.act = l.act;
                if (l.head_kind == 0) {
                    p.out = s.act[l.out];
                } else {
                    // fused label + box head conv: bias vector = [label bias | box bias]
                    SSD_HIP(hipMemcpyAsync(s.dgamma_tmp, net->params[l.p_bias].dev, (size_t)l.Cout1 * sizeof(float),
                                           hipMemcpyDeviceToDevice, st));
                    SSD_HIP(hipMemcpyAsync(s.dgamma_tmp + l.Cout1, net->params[l.p_bias2].dev,
                                           (size_t)(l.Cout - l.Cout1) * sizeof(float), hipMemcpyDeviceToDevice, st));
                    p.shift = s.dgamma_tmp;
                    p.out = s.probs + l.head_off;
                    p.out_pixel_stride = l.head_ps;
                    p.out_batch_stride = l.head_bs;
                    p.n_split = l.Cout1;
                    p.out2 = s.deltas + l.head2_off;
                    p.out2_pixel_stride = l.head2_ps;
                    p.out2_batch_stride = l.head2_bs;
                    p.vec_store2 = (p.out2_pixel_stride % 4 == 0) && (p.out2_batch_stride % 4 == 0);
                }
            }
            rc = launch_conv(p, st);
            if (rc) return rc;
        } else {    // depthwise
            rc = launch_dwconv3x3(x, B, l.H, l.W, l.Cin, l.stride, l.pt, l.pl, l.Ho, l.Wo, net->params[l.p_kernel].dev,
                                  nullptr, nullptr, SSD_ACT_NONE, t.pre, st);
            if (rc) return rc;
        }
        if (l.p_bn >= 0) {
            rc = bn_stats(s, t, M, l.Cout, net->params[l.p_bn + 2].dev, net->params[l.p_bn + 3].dev, st);
            if (rc) return rc;
            hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * l.Cout)), dim3(256), 0, st, t.pre, M, l.Cout, t.mean,
                               t.istd, net->params[l.p_bn].dev, net->params[l.p_bn + 1].dev, l.act,
                               l.res >= 0 ? s.act[l.res] : nullptr, s.act[l.out]);
            SSD_LAUNCH_CHECK();
        }
    }
    rc = launch_softmax(s.probs, (long)B * N, L, s.probs, st);
    if (rc) return rc;
    rc = ssd_loss(actual_deltas_dev, s.deltas, actual_labels_dev, s.probs, B, N, L, neg_pos_ratio, loc_loss_alpha,
                  loc_loss_dev, conf_loss_dev, nullptr, nullptr, s.gdeltas, s.glogits, 1.0f / (float)B, s.loss_ws,
                  s.loss_ws_bytes, stream);
    if (rc) return rc;

    // ------------------------------------------------------------ backward
    std::fill(s.gwritten.begin(), s.gwritten.end(), 0);
    // gradient buckets: everything at or above `threshold` in the flat vector is final -> publish those buckets
    size_t next_bucket = s.bucket_lo.size();
    // weight gradients run on the side stream (s.wstream) beside the rest of the backward; dY alternates between two buffers
    int dy_cur = 0;
    for (int k = 0; k < 2; ++k) s.wpending[k] = false;
    auto acquire_dy = [&]() -> float* {      // next dY buffer: the main stream first waits for the weight gradient still reading it
        dy_cur ^= 1;
        if (s.wstream && s.wpending[dy_cur]) {
            (void)hipStreamWaitEvent(st, s.ev_wdone[dy_cur], 0);
            s.wpending[dy_cur] = false;
        }
        return s.dy_buf[dy_cur];
    };
    auto join_wgrads = [&]() -> int {        // everything issued on the side stream so far is ordered before what follows on st
        if (!s.wstream) return SSD_OK;
        SSD_HIP(hipEventRecord(s.ev_wall, s.wstream));
        SSD_HIP(hipStreamWaitEvent(st, s.ev_wall, 0));
        s.wpending[0] = s.wpending[1] = false;
        return SSD_OK;
    };
    // A bucket is final when BOTH streams have passed this point: the main stream (BatchNorm / depthwise / bias gradients,
    // the data-gradient chain) and the side stream (the dense convs' weight gradients).  The bucket's event is recorded on
    // the SIDE stream behind a wait for the main stream's position -- the main stream itself never waits (round 4 joined the
    // side stream INTO the main stream here and therefore ran the weight gradients in line under buckets: +0.9 ms per step).
    static const bool wside_buckets = !getenv("SSD_HIP_WGRAD_SIDE_BUCKETS") || atoi(getenv("SSD_HIP_WGRAD_SIDE_BUCKETS")) != 0;
    auto mark_ready = [&](long threshold) -> int {
        bool joined = false;
        while (next_bucket > 0 && s.bucket_lo[next_bucket - 1] >= threshold) {
            if (getenv("SSD_HIP_DEBUG_BUCKETS")) fprintf(stderr, "[ssd] bucket %zu (lo %ld) final at threshold %ld\n", next_bucket - 1, s.bucket_lo[next_bucket - 1], threshold);
            if (s.wstream && wside_buckets) {
                if (!joined) {
                    SSD_HIP(hipEventRecord(s.ev_main_pos, st));
                    SSD_HIP(hipStreamWaitEvent(s.wstream, s.ev_main_pos, 0));
                    joined = true;
                }
                SSD_HIP(hipEventRecord(s.bucket_ev[next_bucket - 1], s.wstream));
            } else {
                if (!joined) {
                    const int rj = join_wgrads();
                    if (rj) return rj;
                    joined = true;
                }
                SSD_HIP(hipEventRecord(s.bucket_ev[next_bucket - 1], st));
            }
            --next_bucket;
        }
        return SSD_OK;
    };
    for (int i = (int)net->layers.size() - 1; i >= 0; --i) {
        const Layer& l = net->layers[i];
        TrainLayer& t = s.tl[i];
        if (!t.active) continue;
        if (!s.pending_hi.empty()) {        // layers > i are done: what no layer <= i owns is final
            rc = mark_ready(s.pending_hi[i]);
            if (rc) return rc;
        }
        const long M = (long)B * l.Ho * l.Wo;
        const float* x = s.act[l.in];
        float* const dyb = (l.kind == LK_POOL) ? nullptr : acquire_dy();      // this layer's dY / scratch buffer
        if (l.kind == LK_POOL || l.kind == LK_L2NORM) {
            if (!s.gwritten[l.out]) {
                set_error("train: no gradient reached tensor '%s'", net->tensors[l.out].name.c_str());
                return SSD_E_STATE;
            }
            if (l.kind == LK_POOL) {
                PoolBwdParams pp{};
                pp.x = x; pp.g = s.gact[l.out]; pp.dx = s.gact[l.in];
                pp.B = B; pp.H = l.H; pp.W = l.W; pp.C = l.Cin; pp.Ho = l.Ho; pp.Wo = l.Wo;
                pp.k = l.kh; pp.stride = l.stride; pp.pad_t = l.pt; pp.pad_l = l.pl;
                pp.accumulate = s.gwritten[l.in];
                hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * l.H * l.W * l.Cin)), dim3(256), 0, st, pp);
                SSD_LAUNCH_CHECK();
            } else {
                const long blocks = (M + 3) / 4;
                hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, x,
                                   s.gact[l.out], net->params[l.p_gamma].dev, M, l.Cin, s.gact[l.in], (int)s.gwritten[l.in],
                                   dyb);
                SSD_LAUNCH_CHECK();
                RedParams rp{};
                rp.a = dyb; rp.M = M; rp.C = l.Cin; rp.lda = l.Cin;
                long chunks = 0;
                rc = col_reduce<RED_SUM>(s, rp, &chunks, st);
                if (!rc) rc = col_finalize(s, chunks, l.Cin, 1.0f, grads_flat_dev + t.g_gamma, nullptr, 0, st);
                if (rc) return rc;
            }
            s.gwritten[l.in] = 1;
            continue;
        }
        const float* dY = nullptr;       // gradient w.r.t. the conv / depthwise output, dense [M][ldy]
        int ldy = l.Cout;
        if (l.head_kind) {
            ldy = t.cpad;
            if (t.cpad != l.Cout) SSD_HIP(hipMemsetAsync(dyb, 0, (size_t)M * ldy * sizeof(float), st));
            hipLaunchKernelGGL(gather_head_kernel, dim3(grid_for(M * l.Cout1)), dim3(256), 0, st, s.glogits, l.head_bs,
                               l.head_off, l.head_ps, B, l.Ho * l.Wo, l.Cout1, dyb, ldy, 0);
            hipLaunchKernelGGL(gather_head_kernel, dim3(grid_for(M * (l.Cout - l.Cout1))), dim3(256), 0, st, s.gdeltas,
                               l.head2_bs, l.head2_off, l.head2_ps, B, l.Ho * l.Wo, l.Cout - l.Cout1, dyb, ldy,
                               l.Cout1);
            SSD_LAUNCH_CHECK();
            dY = dyb;
            // bias gradients = column sums
            RedParams rp{};
            rp.a = dY; rp.M = M; rp.C = l.Cout; rp.lda = ldy;
            long chunks = 0;
            rc = col_reduce<RED_SUM>(s, rp, &chunks, st);
            if (!rc) rc = col_finalize(s, chunks, l.Cout, 1.0f, s.dbeta_tmp, nullptr, 0, st);
            if (rc) return rc;
            SSD_HIP(hipMemcpyAsync(grads_flat_dev + t.g_bias, s.dbeta_tmp, (size_t)l.Cout1 * sizeof(float),
                                   hipMemcpyDeviceToDevice, st));
            SSD_HIP(hipMemcpyAsync(grads_flat_dev + t.g_bias2, s.dbeta_tmp + l.Cout1,
                                   (size_t)(l.Cout - l.Cout1) * sizeof(float), hipMemcpyDeviceToDevice, st));
        } else {
            const float* dOut = s.gact[l.out];
            if (!s.gwritten[l.out]) {
                set_error("train: no gradient reached tensor '%s'", net->tensors[l.out].name.c_str());
                return SSD_E_STATE;
            }
            if (l.res >= 0) {       // out = bn(conv) + res: the residual branch receives dOut as is
                hipLaunchKernelGGL(add_kernel, dim3(grid_for(M * l.Cout)), dim3(256), 0, st, s.gact[l.res], dOut, M * l.Cout,
                                   (int)s.gwritten[l.res]);
                SSD_LAUNCH_CHECK();
                s.gwritten[l.res] = 1;
            }
            if (l.p_bn >= 0) {
                RedParams rp{};
                rp.a = dOut; rp.b = t.pre; rp.mean = t.mean; rp.istd = t.istd;
                rp.gamma = net->params[l.p_bn].dev; rp.beta = net->params[l.p_bn + 1].dev;
                rp.M = M; rp.C = l.Cout; rp.lda = l.Cout; rp.act = l.act;
                long chunks = 0;
                rc = col_reduce<RED_BN_BWD>(s, rp, &chunks, st);
                // s1 = dbeta, s2 = dgamma
                if (!rc) rc = col_finalize(s, chunks, l.Cout, 1.0f, grads_flat_dev + t.g_beta, grads_flat_dev + t.g_gamma, 0, st);
                if (rc) return rc;
                hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(M * l.Cout)), dim3(256), 0, st, dOut, t.pre, M, l.Cout,
                                   t.mean, t.istd, net->params[l.p_bn].dev, net->params[l.p_bn + 1].dev, l.act,
                                   grads_flat_dev + t.g_gamma, grads_flat_dev + t.g_beta, dyb);
                SSD_LAUNCH_CHECK();
                dY = dyb;
            } else {
                RedParams rp{};
                rp.a = dOut; rp.b = l.act ? s.act[l.out] : nullptr;
                rp.M = M; rp.C = l.Cout; rp.lda = l.Cout; rp.act = l.act;
                long chunks = 0;
                rc = col_reduce<RED_ACT_BWD>(s, rp, &chunks, st);
                if (!rc) rc = col_finalize(s, chunks, l.Cout, 1.0f, grads_flat_dev + t.g_bias, nullptr, 0, st);
                if (rc) return rc;
                if (l.act) {
                    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(M * l.Cout)), dim3(256), 0, st, dOut, s.act[l.out],
                                       M * l.Cout, l.act, dyb);
                    SSD_LAUNCH_CHECK();
                    dY = dyb;
                } else {
                    dY = dOut;
                }
            }
        }
        if (l.kind == LK_DW) {
            DwBwdParams dp{};
            dp.x = x; dp.g = dY; dp.w = net->params[l.p_kernel].dev; dp.dx = s.gact[l.in];
            dp.B = B; dp.H = l.H; dp.W = l.W; dp.C = l.Cin; dp.Ho = l.Ho; dp.Wo = l.Wo;
            dp.stride = l.stride; dp.pad_t = l.pt; dp.pad_l = l.pl;
            dp.M = M;
            dp.accumulate = s.gwritten[l.in];
            const int ctiles = (l.Cin + 63) / 64;
            long rpc = 0;
            const long chunks = chunks_for(M, ctiles, &rpc, 64, 2048);
            rc = ensure_partial(s, (size_t)chunks * 9 * l.Cin);
            if (rc) return rc;
            dp.rows_per_chunk = rpc;
            dp.partial = s.partial;
            hipLaunchKernelGGL(dw_wgrad_kernel, dim3(ctiles, (unsigned)chunks), dim3(256), 0, st, dp);
            rc = chunk_sum(s.partial, chunks, 9L * l.Cin, grads_flat_dev + t.g_kernel, st);
            if (rc) return rc;
            if (l.stride == 1)
                hipLaunchKernelGGL(dw_dgrad4_kernel, dim3(grid_for((long)B * l.H * ((l.W + 3) / 4) * (l.Cin / 4))), dim3(256), 0, st, dp);
            else
                hipLaunchKernelGGL(dw_dgrad_kernel, dim3(grid_for((long)B * l.H * l.W * (l.Cin / 4))), dim3(256), 0, st, dp);
            SSD_LAUNCH_CHECK();
            s.gwritten[l.in] = 1;
            continue;
        }
        // ---- dense conv: weight gradient(s), on the side stream behind dY (they are needed only at the end of the step /
        // at their gradient bucket: the data-gradient chain below does not wait for them)
        // (not under the bucketed data-parallel exchange: there the communication stream already runs beside the backward,
        // and a third stream measured slower at world size 1 -- 12.6 against 11.5 ms -- than the weight gradients in line)
        const bool side = s.wstream && (s.bucket_lo.empty() || wside_buckets);
        hipStream_t wst = side ? s.wstream : st;
        if (side) {
            SSD_HIP(hipEventRecord(s.ev_dy, st));
            SSD_HIP(hipStreamWaitEvent(s.wstream, s.ev_dy, 0));
        }
        if (l.p_kernel2 < 0) {
            rc = wgrad(s, l, B, x, dY, ldy, l.Cout, grads_flat_dev + t.g_kernel, wst);
        } else {
            rc = wgrad(s, l, B, x, dY, ldy, l.Cout1, grads_flat_dev + t.g_kernel, wst);
            if (!rc) rc = wgrad(s, l, B, x, dY + l.Cout1, ldy, l.Cout - l.Cout1, grads_flat_dev + t.g_kernel2, wst);
        }
        if (rc) return rc;
        // VGG16: kernel_regularizer=l2(5e-4) on every backbone / extra conv (models/ssd_vgg16.py:44-45; the head
        // convs of models/header.py have none): d(5e-4 * sum w^2)/dw = 1e-3 * w, added right behind the layer's
        // weight gradient so that the gradient bucket it lies in is final when the backward has passed the layer
        if (net->backbone == SSD_VGG16 && !l.head_kind) {
            const Param& w = net->params[l.p_kernel];
            hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((long)w.count)), dim3(256), 0, wst, grads_flat_dev + t.g_kernel,
                               w.dev, (long)w.count, 2.0f * 5e-4f);
            SSD_LAUNCH_CHECK();
        }
        if (side && dY == dyb) {             // the buffer may be rewritten only after this weight gradient
            SSD_HIP(hipEventRecord(s.ev_wdone[dy_cur], s.wstream));
            s.wpending[dy_cur] = true;
        }
        // ---- data gradient: conv of dY with the rotated / transposed weights
        if (!t.wbwd) continue;
        const int d = l.dil, kh = l.kh, kw = l.kw;
        const float* gin = dY;
        int Hg = l.Ho, Wg = l.Wo;
        if (l.stride == 2) {
            Hg = (l.Ho - 1) * 2 + 1;
            Wg = (l.Wo - 1) * 2 + 1;
            SSD_HIP(hipMemsetAsync(s.scratch_dz, 0, (size_t)B * Hg * Wg * ldy * sizeof(float), st));
            hipLaunchKernelGGL(dilate2_kernel, dim3(grid_for(M * ldy)), dim3(256), 0, st, dY, B, l.Ho, l.Wo, ldy, Hg, Wg,
                               s.scratch_dz);
            SSD_LAUNCH_CHECK();
            gin = s.scratch_dz;
        }
        const int pt = (kh - 1) * d - l.pt, pl = (kw - 1) * d - l.pl;
        SSD_UNSUPPORTED_IF(pt < 0 || pl < 0, "train: backward-data padding of %s is negative", l.name.c_str());
        ConvParams p = dense_conv_params(B, Hg, Wg, ldy, l.Cin, kh, kw, 1, d, pt, pl, l.H, l.W);
        p.in = gin;
        p.w = t.wbwd;
        p.w3 = conv_split_planes(t.wbwd, p.K, p.Cout);
        p.bf16 = net->precision;
        p.out = s.gact[l.in];
        p.act = SSD_ACT_NONE;
        p.residual = s.gwritten[l.in] ? s.gact[l.in] : nullptr;      // accumulate in the epilogue
        rc = launch_conv(p, st);
        if (rc) return rc;
        s.gwritten[l.in] = 1;
    }
    rc = join_wgrads();             // the caller's stream sees the complete gradient vector
    if (rc) return rc;
    rc = mark_ready(0);
    if (rc) return rc;
    return SSD_OK;
}

// Gradient buckets for the data-parallel exchange ("prefer ... overlapped with backward"):
// bucket k covers the flat offsets [lo[k], lo[k + 1]) (lo ascending, lo[0] == 0, the last bucket ends at
// ssd_net_trainable_floats).  The backward walks the layers last to first, i.e. it finishes the gradient vector
// from its END; ssd_net_train_forward_backward records an event per bucket at the moment every layer owning a
// parameter at or above lo[k] has been passed, and ssd_net_train_wait_bucket makes another stream (the one the
// RCCL all-reduce of that bucket is issued on) wait for exactly that point -- the exchange of the head / extras
// gradients then runs beside the backward of the backbone.  n == 0 clears the plan.
int ssd_net_train_set_buckets(ssd_net* net, int n, const long* lo) {
    SSD_CHECK_ARG(net && n >= 0 && (n == 0 || lo), "ssd_net_train_set_buckets: bad arguments");
    if (!net->train) { set_error("ssd_net_train_set_buckets: call ssd_net_train_begin() first"); return SSD_E_STATE; }
    ssd_train_state& s = *net->train;
    for (int k = 0; k < n; ++k)
        SSD_CHECK_ARG((k == 0 ? lo[0] == 0 : lo[k] > lo[k - 1]) && lo[k] < (long)s.P, "ssd_net_train_set_buckets: starts must ascend from 0 below %zu", s.P);
    for (auto e : s.bucket_ev)
        if (e) (void)hipEventDestroy(e);
    s.bucket_ev.clear();
    s.bucket_lo.assign(lo, lo + n);
    s.pending_hi.clear();
    if (n == 0) return SSD_OK;
    s.bucket_ev.resize(n, nullptr);
    for (int k = 0; k < n; ++k) SSD_HIP(hipEventCreateWithFlags(&s.bucket_ev[k], hipEventDisableTiming));
    // pending_hi[i] = end of the highest trainable parameter owned by an active layer j <= i
    s.pending_hi.assign(net->layers.size(), 0);
    long run = 0;
    for (size_t i = 0; i < net->layers.size(); ++i) {
        const Layer& l = net->layers[i];
        if (s.tl[i].active)
            for (int q : {l.p_kernel, l.p_bias, l.p_kernel2, l.p_bias2, l.p_bn, l.p_bn >= 0 ? l.p_bn + 1 : -1, l.p_gamma})
                if (q >= 0 && q < (int)s.poff.size() && s.poff[q] >= 0) run = std::max(run, s.poff[q] + (long)net->params[q].count);
        s.pending_hi[i] = run;
    }
    return SSD_OK;
}

int ssd_net_train_wait_bucket(ssd_net* net, int k, void* stream) {
    SSD_CHECK_ARG(net && net->train && k >= 0 && k < (int)net->train->bucket_ev.size(), "ssd_net_train_wait_bucket: bad bucket %d", k);
    SSD_HIP(hipStreamWaitEvent((hipStream_t)stream, net->train->bucket_ev[k], 0));
    return SSD_OK;
}

// Sum of the layers' regularisation losses, the term Keras adds to `loss` / `val_loss` beside the compiled
// losses: VGG16's kernel_regularizer=l2(5e-4) on every backbone / extra conv (models/ssd_vgg16.py:44-45),
// nothing for MobileNetV2 (keras-applications builds it without regularisers; models/header.py:60-61 has none).
int ssd_net_regularization_loss(ssd_net* net, float* host_out) {
    SSD_CHECK_ARG(net && host_out, "ssd_net_regularization_loss: NULL argument");
    *host_out = 0.f;
    if (net->backbone != SSD_VGG16) return SSD_OK;
    std::vector<const Param*> ws;
    for (const auto& l : net->layers)
        if (l.kind == LK_CONV && !l.head_kind && l.p_kernel >= 0 && net->params[l.p_kernel].dev) ws.push_back(&net->params[l.p_kernel]);
    if (ws.empty()) return SSD_OK;
    constexpr int kBlocks = 64;
    float* scratch = nullptr;
    SSD_HIP(hipMalloc((void**)&scratch, (ws.size() * kBlocks + 1) * sizeof(float)));
    for (size_t i = 0; i < ws.size(); ++i)
        hipLaunchKernelGGL(sumsq_kernel, dim3(kBlocks), dim3(256), 0, nullptr, ws[i]->dev, (long)ws[i]->count, scratch + i * kBlocks);
    hipLaunchKernelGGL(reg_finalize_kernel, dim3(1), dim3(1), 0, nullptr, scratch, (int)(ws.size() * kBlocks), 5e-4f,
                       scratch + ws.size() * kBlocks);
    const hipError_t e = hipMemcpy(host_out, scratch + ws.size() * kBlocks, sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(scratch);
    SSD_HIP(e);
    return SSD_OK;
}

// One Adam update of every trainable parameter from grads_flat_dev (e.g. after the host's RCCL
// all-reduce; grad_scale = 1 / world_size turns the reduced SUM into the global-batch mean).
int ssd_net_adam_step(ssd_net* net, const float* grads_flat_dev, float lr, float beta1, float beta2, float eps,
                      float grad_scale, void* stream) {
    SSD_CHECK_ARG(net && grads_flat_dev, "ssd_net_adam_step: NULL argument");
    if (!net->train) { set_error("ssd_net_adam_step: call ssd_net_train_begin() first"); return SSD_E_STATE; }
    ssd_train_state& s = *net->train;
    s.step += 1;
    const double t = (double)s.step;
    const float alpha = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((long)s.P)), dim3(256), 0, (hipStream_t)stream, s.flat, s.m, s.v,
                       grads_flat_dev, (long)s.P, alpha, beta1, beta2, eps, grad_scale);
    SSD_LAUNCH_CHECK();
    net->finalized = false;
    net->drop_graphs();
    return SSD_OK;
}

long ssd_net_train_steps(const ssd_net* net) { return (net && net->train) ? net->train->step : 0; }

int ssd_net_train_matrix_flops(const ssd_net* net, double* out3) {
    SSD_CHECK_ARG(net && out3, "ssd_net_train_matrix_flops: NULL argument");
    if (!net->train) { set_error("ssd_net_train_matrix_flops: call ssd_net_train_begin() first"); return SSD_E_STATE; }
    for (int i = 0; i < 3; ++i) out3[i] = net->train->step_flops[i];
    return SSD_OK;
}

// Debug / parity hook: copy a buffer of the LAST ssd_net_train_forward_backward (at batch B) to the
// host.  what = "probs" | "deltas" | "grad_logits" | "grad_deltas" | "<tensor name>" (training
// activation) | "grad:<tensor name>" | "pre:<layer name>" (conv output before BatchNorm) |
// "mean:<layer>" | "var:<layer>".  Returns the element count (host_out may be NULL to query).
long ssd_net_train_fetch(ssd_net* net, const char* what, int B, float* host_out, size_t cap) {
    if (!net || !what || !net->train || B < 1 || B > net->train->batch) {
        set_error("ssd_net_train_fetch: bad arguments / no training state");
        return SSD_E_INVALID;
    }
    ssd_train_state& s = *net->train;
    const std::string w(what);
    const float* src = nullptr;
    size_t n = 0;
    const size_t N = net->num_priors;
    if (w == "probs") { src = s.probs; n = (size_t)B * N * net->L; }
    else if (w == "deltas") { src = s.deltas; n = (size_t)B * N * 4; }
    else if (w == "grad_logits") { src = s.glogits; n = (size_t)B * N * net->L; }
    else if (w == "grad_deltas") { src = s.gdeltas; n = (size_t)B * N * 4; }
    else if (w.rfind("pre:", 0) == 0 || w.rfind("mean:", 0) == 0 || w.rfind("var:", 0) == 0) {
        const std::string name = w.substr(w.find(':') + 1);
        for (size_t i = 0; i < net->layers.size(); ++i)
            if (net->layers[i].name == name && s.tl[i].active && s.tl[i].pre) {
                const Layer& l = net->layers[i];
                if (w[0] == 'p') { src = s.tl[i].pre; n = (size_t)B * l.Ho * l.Wo * l.Cout; }
                else { src = w[0] == 'm' ? s.tl[i].mean : s.tl[i].var; n = l.Cout; }
            }
    } else {
        const bool grad = w.rfind("grad:", 0) == 0;
        auto it = net->tensor_index.find(grad ? w.substr(5) : w);
        if (it != net->tensor_index.end() && it->second > 0) {
            src = grad ? s.gact[it->second] : s.act[it->second];
            n = net->tensors[it->second].per_image * (size_t)B;
        }
    }
    if (!src) {
        set_error("ssd_net_train_fetch: unknown buffer '%s'", what);
        return SSD_E_INVALID;
    }
    if (!host_out) return (long)n;
    if (cap < n) { set_error("ssd_net_train_fetch: buffer too small"); return SSD_E_INVALID; }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("ssd_net_train_fetch: copy failed");
        return SSD_E_HIP;
    }
    return (long)n;
}

}  // extern "C"

// HBM-bound NHWC ops for gfx950: depthwise 3x3 (+BN +ReLU6), MaxPool (TF SAME), channel
// L2 normalisation, row softmax, BatchNorm folding.  All use 16-byte channel-vector
// accesses (4 fp32 channels per lane, lanes consecutive along C => coalesced).
#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act1(float v, int act) {
    if (act == SSD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == SSD_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

// ------------------------------------------------------------------ depthwise 3x3
// One thread = 4 channels x TX consecutive output columns of one output row.  The 3 input
// rows are walked with a sliding register window so each input vector is loaded once per
// thread; the 9 weight vectors + scale/shift live in registers.  Lanes are consecutive in
// C/4, then in x-strips, so a wave reads whole 256..1024-byte runs of a pixel's channels.
template <int STRIDE, int TX>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(
    const float* __restrict__ in, const int B, const int H, const int W, const int C, const int pad_t,
    const int pad_l, const int Ho, const int Wo, const float* __restrict__ w,
    const float* __restrict__ scale, const float* __restrict__ shift, const int act,
    float* __restrict__ out) {
    const int C4 = C >> 2;
    const int strips = (Wo + TX - 1) / TX;
    const long total = (long)B * Ho * strips * C4;
    constexpr int NIN = (TX - 1) * STRIDE + 3;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long r = e / C4;
        const int sx = (int)(r % strips);
        r /= strips;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int c = c4 * 4;
        const int ox0 = sx * TX;
        f32x4 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w + t * C + c);
        f32x4 acc[TX];
#pragma unroll
        for (int t = 0; t < TX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ix0 = ox0 * STRIDE - pad_l;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * STRIDE - pad_t + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            const float* rowp = in + (((long)b * H + iy) * W) * C + c;
            f32x4 xin[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int ix = ix0 + j;
                xin[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ix < (unsigned)W) xin[j] = *reinterpret_cast<const f32x4*>(rowp + (long)ix * C);
            }
#pragma unroll
            for (int t = 0; t < TX; ++t)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 x = xin[t * STRIDE + kx];
                    const f32x4 ww = wv[ky * 3 + kx];
                    acc[t] += x * ww;          // vector form: contracts to v_pk_fma_f32
                }
        }
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + c);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + c);
#pragma unroll
        for (int t = 0; t < TX; ++t) {
            const int ox = ox0 + t;
            if (ox >= Wo) break;
            f32x4 v = acc[t] * sc + sh;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act1(v[j], act);
            *reinterpret_cast<f32x4*>(out + (((long)b * Ho + oy) * Wo + ox) * C + c) = v;
        }
    }
}

int launch_dwconv3x3(const float* in, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                     int Ho, int Wo, const float* w, const float* scale, const float* shift, int act,
                     float* out, hipStream_t st) {
    constexpr int TX = 4;
    const long total = (long)B * Ho * ((Wo + TX - 1) / TX) * (C / 4);
    if (total == 0) return SSD_OK;
    const int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    if (stride == 1)
        hipLaunchKernelGGL((dwconv3x3_kernel<1, TX>), dim3(blocks), dim3(256), 0, st, in, B, H, W, C, pad_t,
                           pad_l, Ho, Wo, w, scale, shift, act, out);
    else
        hipLaunchKernelGGL((dwconv3x3_kernel<2, TX>), dim3(blocks), dim3(256), 0, st, in, B, H, W, C, pad_t,
                           pad_l, Ho, Wo, w, scale, shift, act, out);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------ max pool (TF SAME: pads ignored)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, const int B, const int H,
                                                      const int W, const int C, const int k, const int stride,
                                                      const int pad_t, const int pad_l, const int Ho,
                                                      const int Wo, float* __restrict__ out, short* __restrict__ planes,
                                                      const long plane, const int np) {
    const int C4 = C >> 2;
    const long total = (long)B * Ho * Wo * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C4) * 4;
        long r = e / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad_t + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad_l + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((long)b * H + iy) * W + ix) * C + c);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]);
                m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(out + e * 4) = m;
        if (planes) store_planes4(planes, plane, np, e / C4, c, (long)B * Ho * Wo, m);      // the bf16 planes the LDS-DMA conv tiles read (ssd_convdma.hip)
    }
}

int launch_maxpool(const float* in, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                   int Ho, int Wo, float* out, hipStream_t st, short* planes, long plane, int np) {
    const long total = (long)B * Ho * Wo * (C / 4);
    if (total == 0) return SSD_OK;
    const int blocks = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, st, in, B, H, W, C, k, stride, pad_t, pad_l,
                       Ho, Wo, out, planes, plane, np);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------ L2 normalisation over C
// One wave per pixel: lanes stride over C in float4, 64-wide butterfly reduction.
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ in, const long pixels,
                                                     const int C, const float* __restrict__ gamma,
                                                     float* __restrict__ out, short* __restrict__ planes, const long plane,
                                                     const int np) {
    const int lane = threadIdx.x & 63;
    const long wave0 = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * 256) >> 6;
    for (long px = wave0; px < pixels; px += nwaves) {
        const float* x = in + px * C;
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float inv = 1.0f / sqrtf(fmaxf(s, 1e-12f));
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 y = v * inv * g;
            *reinterpret_cast<f32x4*>(out + px * C + c) = y;
            if (planes) store_planes4(planes, plane, np, px, c, pixels, y);
        }
    }
}

int launch_l2norm(const float* in, long pixels, int C, const float* gamma, float* out, hipStream_t st, short* planes,
                  long plane, int np) {
    if (pixels == 0) return SSD_OK;
    const int blocks = (int)(cdiv(pixels, 4) < 8192 ? cdiv(pixels, 4) : 8192);
    hipLaunchKernelGGL(l2norm_kernel, dim3(blocks), dim3(256), 0, st, in, pixels, C, gamma, out, planes, plane, np);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------ softmax over the last dim
// (the LDS-staged kernel lives in ssd_bbox.hip beside the fused decoder, which shares its row function: same bits)
int launch_softmax_lds(const float* in, long rows, int L, float* out, hipStream_t st);

__global__ void softmax_direct_kernel(const float* __restrict__ in, const long rows, const int L,
                                      float* __restrict__ out) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const float* x = in + r * L;
        float mx = x[0];
        for (int c = 1; c < L; ++c) mx = fmaxf(mx, x[c]);
        float s = 0.f;
        for (int c = 0; c < L; ++c) s += expf(x[c] - mx);
        for (int c = 0; c < L; ++c) out[r * L + c] = expf(x[c] - mx) / s;
    }
}

int launch_softmax(const float* in, long rows, int L, float* out, hipStream_t st) {
    if (rows == 0) return SSD_OK;
    const size_t lds = (size_t)256 * L * 4;
    if (lds <= 64 * 1024) {
        return launch_softmax_lds(in, rows, L, out, st);
    } else {
        const int blocks = (int)(cdiv(rows, 256) < 4096 ? cdiv(rows, 256) : 4096);
        hipLaunchKernelGGL(softmax_direct_kernel, dim3(blocks), dim3(256), 0, st, in, rows, L, out);
    }
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------ BatchNorm folding
// tf.nn.batch_normalization inference form: inv = gamma * rsqrt(var + eps);
// y = x * inv + (beta - mean * inv)   =>   scale = inv, shift = beta - mean * inv.
__global__ void fold_bn_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float eps, const int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = inv;
    shift[c] = beta[c] - mean[c] * inv;
}

__global__ void scale_weights_kernel(const float* in, const float* scale, const int rows, const int nvalid,
                                     const int cols, const int by_col, float* out) {
    const long total = (long)rows * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / cols), c = (int)(e - (long)r * cols);
        const float s = by_col ? scale[c] : (r < nvalid ? scale[r] : 1.0f);
        out[e] = in[e] * s;
    }
}

int launch_scale_rows(const float* in, const float* scale, int rows, int nvalid, int cols, float* out, hipStream_t st) {
    const long total = (long)rows * cols;
    if (total == 0) return SSD_OK;
    hipLaunchKernelGGL(scale_weights_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0, st,
                       in, scale, rows, nvalid, cols, 0, out);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_scale_cols(const float* in, const float* scale, int rows, int cols, float* out, hipStream_t st) {
    const long total = (long)rows * cols;
    if (total == 0) return SSD_OK;
    hipLaunchKernelGGL(scale_weights_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0, st,
                       in, scale, rows, rows, cols, 1, out);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                   float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, mean, var, eps, C,
                       scale, shift);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

using namespace ssd;

extern "C" {

int ssd_dwconv3x3(const float* in_dev, int B, int H, int W, int C, int stride, int pad_t, int pad_l, int pad_b,
                  int pad_r, const float* w_dev, const float* scale_dev, const float* shift_dev, int act,
                  float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 1, "dwconv3x3: bad sizes");
    SSD_CHECK_ARG(stride == 1 || stride == 2, "dwconv3x3: stride %d (1 or 2)", stride);
    SSD_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && pad_b >= 0 && pad_r >= 0, "dwconv3x3: negative padding");
    SSD_UNSUPPORTED_IF(C % 4 != 0, "dwconv3x3: C=%d must be a multiple of 4", C);
    const int Ho = ssd_conv_out_size(H, 3, stride, 1, pad_t, pad_b);
    const int Wo = ssd_conv_out_size(W, 3, stride, 1, pad_l, pad_r);
    SSD_CHECK_ARG(Ho >= 1 && Wo >= 1, "dwconv3x3: empty output");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && w_dev && out_dev, "dwconv3x3: NULL pointer");
    return launch_dwconv3x3(in_dev, B, H, W, C, stride, pad_t, pad_l, Ho, Wo, w_dev, scale_dev, shift_dev, act,
                            out_dev, (hipStream_t)stream);
}

int ssd_maxpool2d(const float* in_dev, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                  int pad_b, int pad_r, float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 1 && k >= 1 && stride >= 1, "maxpool2d: bad sizes");
    SSD_UNSUPPORTED_IF(C % 4 != 0, "maxpool2d: C=%d must be a multiple of 4", C);
    const int Ho = ssd_conv_out_size(H, k, stride, 1, pad_t, pad_b);
    const int Wo = ssd_conv_out_size(W, k, stride, 1, pad_l, pad_r);
    SSD_CHECK_ARG(Ho >= 1 && Wo >= 1, "maxpool2d: empty output");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && out_dev, "maxpool2d: NULL pointer");
    return launch_maxpool(in_dev, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, out_dev, (hipStream_t)stream);
}

int ssd_l2norm(const float* in_dev, long pixels, int C, const float* gamma_dev, float* out_dev, void* stream) {
    SSD_CHECK_ARG(pixels >= 0 && C >= 1, "l2norm: bad sizes");
    SSD_UNSUPPORTED_IF(C % 4 != 0, "l2norm: C=%d must be a multiple of 4", C);
    if (pixels == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && gamma_dev && out_dev, "l2norm: NULL pointer");
    return launch_l2norm(in_dev, pixels, C, gamma_dev, out_dev, (hipStream_t)stream);
}

int ssd_softmax(const float* in_dev, long rows, int L, float* out_dev, void* stream) {
    SSD_CHECK_ARG(rows >= 0 && L >= 1, "softmax: bad sizes");
    if (rows == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && out_dev, "softmax: NULL pointer");
    return launch_softmax(in_dev, rows, L, out_dev, (hipStream_t)stream);
}

}  // extern "C"

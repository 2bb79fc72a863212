// GPU input pipeline: reference utils/data_utils.py:22-23
//     img = tf.image.convert_image_dtype(img, tf.float32)      // uint8 -> float32 * (1/255)
//     img = tf.image.resize(img, (final_height, final_width))  // bilinear, half-pixel centres
// as one HBM-bound kernel: a thread produces one output pixel (3 channels) from its 4 source
// pixels; no intermediate float image.  [3P] TF 2.x ResizeBilinear CPU kernel semantics
// (half_pixel_centers = true, antialias = false): scale = in / out (fp32);
// src = (dst + 0.5) * scale - 0.5; lower = max(floor(src), 0); upper = min(ceil(src), in - 1);
// lerp = src - floor(src); out = top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp.
// Compiled with -ffp-contract=off: every multiply/add rounds separately like the reference's ops.
#include "common.h"

namespace ssd {

__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ img, const int B,
                                                        const int H, const int W, const int C, const int Ho,
                                                        const int Wo, float* __restrict__ out) {
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const float inv255 = (float)(1.0 / 255.0);
    const long total = (long)B * Ho * Wo;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wo);
        const long r = e / Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fyf = floorf(fy), fxf = floorf(fx);
        const int y0 = max((int)fyf, 0), y1 = min((int)ceilf(fy), H - 1);
        const int x0 = max((int)fxf, 0), x1 = min((int)ceilf(fx), W - 1);
        const float ly = fy - fyf, lx = fx - fxf;
        const unsigned char* base = img + (long)b * H * W * C;
        const unsigned char *ptl = base + ((long)y0 * W + x0) * C, *ptr_ = base + ((long)y0 * W + x1) * C;
        const unsigned char *pbl = base + ((long)y1 * W + x0) * C, *pbr = base + ((long)y1 * W + x1) * C;
        float* o = out + e * C;
        for (int c = 0; c < C; ++c) {
            const float tl = (float)ptl[c] * inv255, tr = (float)ptr_[c] * inv255;
            const float bl = (float)pbl[c] * inv255, br = (float)pbr[c] * inv255;
            const float top = tl + (tr - tl) * lx;
            const float bot = bl + (br - bl) * lx;
            o[c] = top + (bot - top) * ly;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Augmentation (reference augmentation.py, used at trainer.py:42), the deterministic pieces with the random draws as
// inputs: per-channel image mean (expand's fill colour, contrast's pivot), the geometric chain expand -> crop -> bilinear
// resize -> horizontal flip as ONE gather kernel over a virtual canvas, and the photometric chain brightness -> contrast
// -> hue -> saturation -> clip as one elementwise kernel.  Same per-op fp32 rounding as the oracle (oracle/augment_oracle.py;
// this file is compiled with -ffp-contract=off).

// mean[b][c] over H*W of (img + add[b]): one workgroup per image, float64 partial sums in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void image_mean_kernel(const float* __restrict__ img, const int HW, const int C,
                                                         const float* __restrict__ add, float* __restrict__ mean) {
    __shared__ double red[1024];
    const int b = blockIdx.x;
    const float* x = img + (long)b * HW * C;
    const float a = add ? add[b] : 0.0f;
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int i = threadIdx.x; i < HW; i += 1024) s += (double)(x[(long)i * C + c] + a);
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) mean[b * C + c] = (float)(red[0] / (double)HW);
        __syncthreads();
    }
}

// params[b] = {canvas_h, canvas_w, pad_top, pad_left, crop_y, crop_x, crop_h, crop_w, flip, use_crop}: the image sits at
// (pad_top, pad_left) of a canvas_h x canvas_w canvas filled with fill[b][c]; the window (crop_y, crop_x, crop_h, crop_w)
// of that canvas is resized (bilinear, half-pixel centres) to H x W, then flipped left-right.  use_crop 0: flip only.
__global__ __launch_bounds__(256) void augment_geometry_kernel(const float* __restrict__ img, const int B, const int H,
                                                              const int W, const int C, const int Ho, const int Wo,
                                                              const int* __restrict__ params, const float* __restrict__ fill,
                                                              float* __restrict__ out) {
    const long total = (long)B * Ho * Wo;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wo);
        const long r = e / Wo;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        const int* q = params + b * 10;
        const int pad_top = q[2], pad_left = q[3], cy = q[4], cx = q[5], ch = q[6], cw = q[7];
        const int sx = q[8] ? Wo - 1 - ox : ox;
        const float* xb = img + (long)b * H * W * C;
        float* o = out + e * C;
        if (!q[9]) {        // flip only (Ho == H, Wo == W: host check)
            for (int c = 0; c < C; ++c) o[c] = xb[((long)oy * W + sx) * C + c];
            continue;
        }
        const float sy = (float)ch / (float)Ho, sxs = (float)cw / (float)Wo;
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)sx + 0.5f) * sxs - 0.5f;
        const float fyf = floorf(fy), fxf = floorf(fx);
        const int y0 = max((int)fyf, 0), y1 = min((int)ceilf(fy), ch - 1);
        const int x0 = max((int)fxf, 0), x1 = min((int)ceilf(fx), cw - 1);
        const float ly = fy - fyf, lx = fx - fxf;
        // canvas -> image coordinates; outside the image the canvas holds the fill colour
        const int iy0 = cy + y0 - pad_top, iy1 = cy + y1 - pad_top, ix0 = cx + x0 - pad_left, ix1 = cx + x1 - pad_left;
        const bool vy0 = (unsigned)iy0 < (unsigned)H, vy1 = (unsigned)iy1 < (unsigned)H;
        const bool vx0 = (unsigned)ix0 < (unsigned)W, vx1 = (unsigned)ix1 < (unsigned)W;
        for (int c = 0; c < C; ++c) {
            const float f = fill[b * C + c];
            const float tl = (vy0 && vx0) ? xb[((long)iy0 * W + ix0) * C + c] : f;
            const float tr = (vy0 && vx1) ? xb[((long)iy0 * W + ix1) * C + c] : f;
            const float bl = (vy1 && vx0) ? xb[((long)iy1 * W + ix0) * C + c] : f;
            const float br = (vy1 && vx1) ? xb[((long)iy1 * W + ix1) * C + c] : f;
            const float top = tl + (tr - tl) * lx;
            const float bot = bl + (br - bl) * lx;
            o[c] = top + (bot - top) * ly;
        }
    }
}

__device__ __forceinline__ void rgb_to_hsv(const float r, const float g, const float b, float& h, float& s, float& v) {
    v = fmaxf(fmaxf(r, g), b);
    const float mn = fminf(fminf(r, g), b);
    const float range = v - mn;
    s = v > 0.0f ? range / v : 0.0f;
    const float norm = range > 0.0f ? 1.0f / (6.0f * range) : 0.0f;
    const float two6 = (float)(2.0 / 6.0), four6 = (float)(4.0 / 6.0);
    h = r == v ? norm * (g - b) : (g == v ? norm * (b - r) + two6 : norm * (r - g) + four6);
    h = range > 0.0f ? h : 0.0f;
    h = h < 0.0f ? h + 1.0f : h;
}
__device__ __forceinline__ void hsv_to_rgb(const float h, const float s, const float v, float& r, float& g, float& b) {
    const float c = s * v;
    const float m = v - c;
    const float dh = h * 6.0f;
    float f = dh;
    while (f >= 2.0f) f -= 2.0f;
    const float x = c * (1.0f - fabsf(f - 1.0f));
    const int hc = (int)floorf(dh);
    float rr = c, gg = 0.0f, bb = x;             // category 5 (and beyond)
    if (hc == 0) { rr = c; gg = x; bb = 0.0f; }
    else if (hc == 1) { rr = x; gg = c; bb = 0.0f; }
    else if (hc == 2) { rr = 0.0f; gg = c; bb = x; }
    else if (hc == 3) { rr = 0.0f; gg = x; bb = c; }
    else if (hc == 4) { rr = x; gg = 0.0f; bb = c; }
    r = rr + m; g = gg + m; b = bb + m;
}

// in place on RGB float images; params[b] = {brightness delta, contrast factor, hue delta, saturation factor}, flags[b] bits
// 0..3 say which of the four run (reference order), mean[b][3] = per-channel mean of the image contrast pivots on (the
// brightness-adjusted image: image_mean_kernel with add = delta); always ends with clip to [0,1]
__global__ __launch_bounds__(256) void augment_color_kernel(float* __restrict__ img, const int B, const long HW,
                                                           const float* __restrict__ params, const int* __restrict__ flags,
                                                           const float* __restrict__ mean) {
    const long total = (long)B * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int b = (int)(e / HW);
        const float* q = params + b * 4;
        const int fl = flags[b];
        float* p = img + e * 3;
        float r = p[0], g = p[1], bl = p[2];
        if (fl & 1) { r = r + q[0]; g = g + q[0]; bl = bl + q[0]; }
        if (fl & 2) {
            const float* m = mean + b * 3;
            r = (r - m[0]) * q[1] + m[0];
            g = (g - m[1]) * q[1] + m[1];
            bl = (bl - m[2]) * q[1] + m[2];
        }
        if (fl & 4) {
            float h, s, v;
            rgb_to_hsv(r, g, bl, h, s, v);
            h = h + q[2];
            h = h < 0.0f ? h + 1.0f : h;
            h = h >= 1.0f ? h - 1.0f : h;
            hsv_to_rgb(h, s, v, r, g, bl);
        }
        if (fl & 8) {
            float h, s, v;
            rgb_to_hsv(r, g, bl, h, s, v);
            s = fminf(fmaxf(s * q[3], 0.0f), 1.0f);
            hsv_to_rgb(h, s, v, r, g, bl);
        }
        p[0] = fminf(fmaxf(r, 0.0f), 1.0f);
        p[1] = fminf(fmaxf(g, 0.0f), 1.0f);
        p[2] = fminf(fmaxf(bl, 0.0f), 1.0f);
    }
}

}  // namespace ssd

using namespace ssd;

extern "C" int ssd_image_mean(const float* img_dev, int B, int H, int W, int C, const float* add_dev, float* mean_out_dev,
                              void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 1 && C <= 16, "ssd_image_mean: bad sizes");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(img_dev && mean_out_dev, "ssd_image_mean: NULL pointer");
    hipLaunchKernelGGL(image_mean_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, img_dev, H * W, C, add_dev, mean_out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

extern "C" int ssd_augment_geometry(const float* img_dev, int B, int H, int W, int C, int out_h, int out_w,
                                    const int* params_dev, const float* fill_dev, float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 1 && out_h >= 1 && out_w >= 1, "ssd_augment_geometry: bad sizes");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(img_dev && params_dev && fill_dev && out_dev && img_dev != out_dev, "ssd_augment_geometry: NULL pointer / in-place call");
    const long total = (long)B * out_h * out_w;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(augment_geometry_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                       (hipStream_t)stream, img_dev, B, H, W, C, out_h, out_w, params_dev, fill_dev, out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

extern "C" int ssd_augment_color(float* img_dev, int B, int H, int W, const float* params_dev, const int* flags_dev,
                                 const float* mean_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1, "ssd_augment_color: bad sizes");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(img_dev && params_dev && flags_dev && mean_dev, "ssd_augment_color: NULL pointer");
    const long total = (long)B * H * W;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(augment_color_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                       (hipStream_t)stream, img_dev, B, (long)H * W, params_dev, flags_dev, mean_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

extern "C" int ssd_preprocess(const unsigned char* image_u8_dev, int B, int H, int W, int C, int out_h, int out_w,
                              float* out_dev, void* stream) {
    SSD_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && C >= 1 && out_h >= 1 && out_w >= 1, "ssd_preprocess: bad sizes");
    if (B == 0) return SSD_OK;
    SSD_CHECK_ARG(image_u8_dev && out_dev, "ssd_preprocess: NULL pointer");
    const long total = (long)B * out_h * out_w;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                       (hipStream_t)stream, image_u8_dev, B, H, W, C, out_h, out_w, out_dev);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

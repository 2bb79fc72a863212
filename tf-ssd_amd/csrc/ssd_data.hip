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

}  // namespace ssd

using namespace ssd;

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

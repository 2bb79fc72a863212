// Dense convolution for gfx950 as an implicit GEMM on the fp32 matrix cores.
//
//   out[m, n] = act( sum_k X[m, k] * W[k, n] * scale[n] + shift[n] ) (+ residual[m, n])
//   m = (b, oy, ox) output pixel, n = output channel, k = (ky, kx, ci)
//
// Mapping to CDNA4 (MI355X):
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 256 FLOP/clk/CU): the WEIGHTS are the MFMA "A"
//     operand and the PIXELS the "B" operand, so each lane ends up holding 4 consecutive
//     output channels of one pixel -> one 16-byte store per accumulator tile, and the
//     per-channel scale/shift (folded BatchNorm or bias) is a 16-byte load.
//   * Both operands are K-contiguous (NHWC activations; weights pre-packed [N][K]), so a
//     lane fetches 4 consecutive k with one ds_read_b128 and feeds 4 MFMA k-steps from it
//     (the MFMA k index is a summation index: any bijection shared by A and B is valid).
//   * 256-thread blocks = 4 waves (WM x WN), wave tile (16*MT) x (16*NT); LDS tiles are
//     UNPADDED [rows][BK] with the 16-byte column index XOR-swizzled by the row (BK >= 32;
//     conflict-free fragment reads and tile writes under gfx950's b128 lane grouping) or
//     [rows][BK+4] for BK = 16; two LDS stages, global->register prefetch of the next K tile in
//     flight while the current one is multiplied.
//   * im2col is never materialised: for k x k convs the per-row (b, iy0, ix0) is decoded
//     once per block and each K tile (BK | Cin) lies inside one filter tap; TF's asymmetric
//     SAME / ZeroPadding2D pads are explicit (pad_t, pad_l) and out-of-image taps load zeros.
//   * The epilogue writes through (batch stride, pixel stride) so the SSD head convs store
//     straight into the concatenated [B, N, K] buffers (reference models/header.py:34-41).
#include "ssd_conv.h"
#include "ssd_conv_mfma.h"

namespace ssd {

// conv_mfma_kernel: ssd_conv_mfma.h

// Split-K reduction + epilogue (deterministic order: s = 0, 1, ...).  A thread sums 4
// consecutive flat [M*Cout] elements with float4 loads (slabs are 16-byte aligned when
// M*Cout % 4 == 0); the epilogue / scatter to the destination(s) is per element.
template <int V>
__global__ void splitk_reduce_kernel(const ConvParams p) {
    const long total = p.M * p.Cout;
    const long groups = total / V;
    const int HoWo = p.Ho * p.Wo;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x) {
        float t[V];
#pragma unroll
        for (int j = 0; j < V; ++j) t[j] = 0.f;
        for (int s = 0; s < p.split_k; ++s) {
            const float* src = p.partial + (long)s * total + g * V;
            if (V == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int j = 0; j < V; ++j) t[j] += v[j];
            } else {
                t[0] += src[0];
            }
        }
        long m = (g * V) / p.Cout;
        int n = (int)(g * V - m * p.Cout);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float r = t[j];
            if (p.scale) r = r * p.scale[n];
            if (p.shift) r = r + p.shift[n];
            r = apply_act(r, p.act);
            if (p.residual) r += p.residual[g * V + j];
            const int b = (int)(m / HoWo);
            const long pix = m - (long)b * HoWo;
            if (p.n_split && n >= p.n_split)
                p.out2[(long)b * p.out2_batch_stride + pix * p.out2_pixel_stride + (n - p.n_split)] = r;
            else
                p.out[(long)b * p.out_batch_stride + pix * p.out_pixel_stride + n] = r;
            if (++n == p.Cout) { n = 0; ++m; }
            t[j] = r;
        }
        // (dense outputs with Cout % 4 == 0 only: the four elements lie in one row)
        if (V == 4 && p.op) store_planes4_lin(p.op, p.op_plane, p.op_np, g * V, p.Cout, p.M, f32x4{t[0], t[1], t[2], t[3]});
    }
}

// VALU direct convolution for the layers the MFMA path cannot take (Cin not a multiple of 4:
// the RGB stems Conv1 / conv1_1, K = 27).  Weights [K][Cout] staged in LDS; one thread =
// one pixel x 4 output channels; lanes sharing a pixel broadcast-load the same inputs.
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [K][Cout4]
    const int Cout4 = (p.Cout + 3) & ~3;
    for (int e = threadIdx.x; e < p.K * Cout4; e += 256) {
        const int k = e / Cout4, n = e - k * Cout4;
        wl[e] = n < p.Cout ? p.w[(long)n * p.Kpad + k] : 0.f;
    }
    __syncthreads();
    const int cq_n = Cout4 / 4;
    const int HoWo = p.Ho * p.Wo;
    const long total = p.M * cq_n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long m = e / cq_n;
        const int n = (int)(e - m * cq_n) * 4;
        const int b = (int)(m / HoWo);
        const long pix = m - (long)b * HoWo;
        const int oy = (int)(pix / p.Wo), ox = (int)(pix - (long)oy * p.Wo);
        const float* xb = p.in + (long)b * p.H * p.W * p.Cin;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < p.kh; ++ky) {
            const int iy = oy * p.stride - p.pad_t + ky * p.dil;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int kx = 0; kx < p.kw; ++kx) {
                const int ix = ox * p.stride - p.pad_l + kx * p.dil;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const float* xp = xb + ((long)iy * p.W + ix) * p.Cin;
                const float* wp = wl + (ky * p.kw + kx) * p.Cin * Cout4 + n;
                for (int ci = 0; ci < p.Cin; ++ci) {
                    const float x = xp[ci];
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + ci * Cout4);
                    acc += x * w4;
                }
            }
        }
        float* orow = p.out + (long)b * p.out_batch_stride + pix * p.out_pixel_stride;
        for (int j = 0; j < 4; ++j) {
            if (n + j >= p.Cout) break;
            float t = acc[j];
            if (p.scale) t = t * p.scale[n + j];
            if (p.shift) t = t + p.shift[n + j];
            t = apply_act(t, p.act);
            if (p.residual) t += p.residual[m * p.Cout + n + j];
            orow[n + j] = t;
        }
    }
}

// RGB stem (Cin = 3, 3x3: Conv1 of MobileNetV2, conv1_1 of VGG16): K = 27 is too short
// for the MFMA tiles, and the layer is bound by its output write.  One thread = PX output
// pixels x 32 output channels; the 27 x 32 weight slab sits in LDS and is read with
// wave-uniform (broadcast) 16-byte reads; 32 accumulators per pixel stay in registers and
// leave as eight 16-byte stores (each lane writes its pixel's 128 contiguous bytes).
template <int PX>
__global__ __launch_bounds__(256) void conv_stem_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 32 + 256 * 36];
    float* so = wl + 27 * 32;                          // [256][36] output staging
    const int cg = blockIdx.y * 32;                    // output-channel group
    for (int e = threadIdx.x; e < 27 * 32; e += 256) {
        const int k = e >> 5, n = cg + (e & 31);
        wl[e] = n < p.Cout ? p.w[(long)n * p.Kpad + k] : 0.f;
    }
    __syncthreads();
    const int HoWo = p.Ho * p.Wo;
    for (long base = (long)blockIdx.x * 256 * PX; base < p.M; base += (long)gridDim.x * 256 * PX) {
        f32x4 acc[PX][8];
        long mm[PX];
        const float* xb[PX];
        int iy0[PX], ix0[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            mm[q] = base + q * 256 + threadIdx.x;
            const long m = mm[q] < p.M ? mm[q] : 0;
            const int b = (int)(m / HoWo);
            const int pix = (int)(m - (long)b * HoWo);
            const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
            iy0[q] = oy * p.stride - p.pad_t;
            ix0[q] = ox * p.stride - p.pad_l;
            xb[q] = p.in + (long)b * p.H * p.W * 3;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[q][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            {
                const int ky = tap / 3, kx = tap - ky * 3;
                float x[PX][3];
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    const int iy = iy0[q] + ky * p.dil, ix = ix0[q] + kx * p.dil;
                    x[q][0] = x[q][1] = x[q][2] = 0.f;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                        const float* xp = xb[q] + ((long)iy * p.W + ix) * 3;
                        x[q][0] = xp[0]; x[q][1] = xp[1]; x[q][2] = xp[2];
                    }
                }
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float* wk = wl + (tap * 3 + ci) * 32;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wk + c * 4);
#pragma unroll
                        for (int q = 0; q < PX; ++q) {
                            acc[q][c] += x[q][ci] * w4;      // v_pk_fma_f32
                        }
                    }
                }
            }
        }
        // stores staged through LDS: a lane's 32 channels are 128 contiguous bytes, but
        // lanes are pixels, so direct stores would be 16-byte pieces at a 128-byte stride
        // (measured 3.1x HBM write amplification); the transposed copy writes whole lines.
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x4 v = acc[q][c];
                if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + cg + c * 4);
                if (p.shift) v = v + *reinterpret_cast<const f32x4*>(p.shift + cg + c * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
                *reinterpret_cast<f32x4*>(so + threadIdx.x * 36 + c * 4) = v;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 256 * 8; e += 256) {
                const int pl = e >> 3, c4 = e & 7;
                const long m = base + q * 256 + pl;
                if (m >= p.M) continue;
                const int b = (int)(m / HoWo);
                const long pix = m - (long)b * HoWo;
                const f32x4 v = *reinterpret_cast<const f32x4*>(so + pl * 36 + c4 * 4);
                *reinterpret_cast<f32x4*>(p.out + (long)b * p.out_batch_stride + pix * p.out_pixel_stride + cg + c4 * 4) = v;
                if (p.op) store_planes4(p.op, p.op_plane, p.op_np, m, cg + c4 * 4, p.M, v);      // (dense outputs only)
            }
        }
    }
}

static bool stem_ok(const ConvParams& p) {
    return p.Cin == 3 && p.kh == 3 && p.kw == 3 && (p.Cout % 32) == 0 && !p.residual && !p.n_split && p.vec_store &&
           (!p.scale || (((uintptr_t)p.scale & 15) == 0)) && (!p.shift || (((uintptr_t)p.shift & 15) == 0));
}

// HWIO [K][Cout] -> packed [Npad][Kpad], zero padded.
__global__ void pack_weights_kernel(const float* __restrict__ hwio, int K, int Cout, int Kpad, int Npad,
                                    float* __restrict__ packed) {
    const long total = (long)Npad * Kpad;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / Kpad), k = (int)(e - (long)n * Kpad);
        packed[e] = (n < Cout && k < K) ? hwio[(long)k * Cout + n] : 0.f;
    }
}

// ------------------------------------------------------------------ config table
typedef void (*conv_kernel_t)(const ConvParams);
struct ConvCfg {
    const char* name;
    int BM, BN, BK;
    conv_kernel_t gemm, general;
};
#define CFG(MT, NT, WM, WN, BK)                                                            \
    {"mfma_" #MT "x" #NT "_" #WM "x" #WN "_k" #BK, 16 * MT * WM, 16 * NT * WN, BK,          \
     conv_mfma_kernel<MT, NT, WM, WN, BK, true>, conv_mfma_kernel<MT, NT, WM, WN, BK, false>}
static const ConvCfg kCfgs[] = {
    CFG(2, 1, 4, 1, 32),   // 0: 128 x 16
    CFG(2, 2, 4, 1, 32),   // 1: 128 x 32
    CFG(2, 4, 4, 1, 32),   // 2: 128 x 64
    CFG(2, 6, 4, 1, 32),   // 3: 128 x 96
    CFG(2, 4, 2, 2, 32),   // 4: 64 x 128
    CFG(1, 2, 2, 2, 32),   // 5: 32 x 64
    CFG(4, 4, 2, 2, 32),   // 6: 128 x 128
    CFG(1, 1, 4, 1, 32),   // 7: 64 x 16
    CFG(2, 9, 4, 1, 32),   // 8: 128 x 144
    CFG(2, 12, 4, 1, 32),  // 9: 128 x 192
    CFG(1, 4, 1, 4, 32),   // 10: 16 x 256
    CFG(2, 2, 2, 2, 32),   // 11: 64 x 64
    CFG(2, 6, 4, 1, 16),   // 12: 128 x 96, BK 16 (K = 16 expand layers)
    CFG(1, 2, 1, 4, 32),   // 13: 16 x 128
    CFG(4, 2, 4, 1, 32),   // 14: 256 x 32
    CFG(2, 8, 2, 2, 32),   // 15: 64 x 256
    CFG(2, 7, 4, 1, 32),   // 16: 128 x 112 (fused heads, A*(L+4) = 100)
    CFG(2, 5, 2, 2, 32),   // 17: 64 x 160  (fused heads, 150)
    CFG(1, 7, 4, 1, 32),   // 18: 64 x 112
    CFG(1, 5, 2, 2, 32),   // 19: 32 x 160
    CFG(4, 3, 2, 2, 32),   // 20: 128 x 96 (64 x 48 wave tiles)
    CFG(4, 6, 2, 2, 32),   // 21: 128 x 192
    CFG(1, 7, 4, 1, 64),   // 22..: BK = 64 variants for the long-K 3x3 convs (heads, extras)
    CFG(1, 5, 2, 2, 64),
    CFG(2, 7, 4, 1, 64),
    CFG(2, 5, 2, 2, 64),
    CFG(1, 2, 2, 2, 64),
    CFG(2, 4, 2, 2, 64),
    CFG(1, 1, 4, 1, 64),
    CFG(2, 2, 2, 2, 64),
    // 30..: large per-wave tiles (more MFMAs per fragment read / staged byte; 1 workgroup per CU)
    CFG(4, 5, 2, 2, 32),   // 128 x 160
    CFG(4, 7, 4, 1, 32),   // 256 x 112
    CFG(4, 4, 4, 1, 32),   // 256 x 64
    CFG(4, 6, 4, 1, 32),   // 256 x 96
    CFG(4, 3, 4, 1, 32),   // 256 x 48
    CFG(4, 5, 4, 1, 32),   // 256 x 80
    CFG(2, 10, 2, 2, 32),  // 64 x 320
};
constexpr int kNumMfmaCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
// config ids: [0, kNumMfmaCfgs) implicit-GEMM tiles, then the Winograd F(2x2,3x3) tiles
// (csrc/ssd_wino.hip), last = VALU direct kernel
// then the skinny (in-workgroup K split) tiles (csrc/ssd_skinny.hip)
// then the split-bf16 implicit-GEMM tiles (csrc/ssd_conv3.hip)
static int skinny_cfg0() { return kNumMfmaCfgs + wino_num_configs(); }
static int mfma3_cfg0() { return kNumMfmaCfgs + wino_num_configs() + skinny_num_configs(); }
// then the same tiles as bf16 (one-product) kernels: only chosen when asked for by name / by the net's precision-1 mode
static int bf16_cfg0() { return kNumMfmaCfgs + wino_num_configs() + skinny_num_configs() + mfma3_num_configs(); }
// then the LDS-DMA tiles over pre-split activation planes (csrc/ssd_convdma.hip): "dma3_*" (fp32 nets), "dmab_*" (bf16 mode)
static int dma3_cfg0() { return kNumMfmaCfgs + wino_num_configs() + skinny_num_configs() + 2 * mfma3_num_configs(); }
static int dmab_cfg0() { return dma3_cfg0() + dma_num_configs(); }
static int direct_cfg() { return dma3_cfg0() + 2 * dma_num_configs(); }
#define kSkinnyCfg0 skinny_cfg0()
#define kMfma3Cfg0 mfma3_cfg0()
#define kBf16Cfg0 bf16_cfg0()
#define kDma3Cfg0 dma3_cfg0()
#define kDmabCfg0 dmab_cfg0()
#define kDirectCfg direct_cfg()
bool conv_config_is_dma(int cfg) { return cfg >= kDma3Cfg0 && cfg < kDirectCfg; }
bool conv_config_writes_planes(int cfg, const ConvParams& p) {
    if (cfg == kDirectCfg) return stem_ok(p);              // the RGB stem kernel does, the generic VALU kernel does not
    return (cfg >= 0 && cfg < kNumMfmaCfgs) || (cfg >= kMfma3Cfg0 && cfg < kDirectCfg);      // fp32-MFMA, split-bf16, bf16 and LDS-DMA tiles
}

int conv_num_mfma_configs() { return kNumMfmaCfgs; }
int conv_num_configs() { return kDirectCfg + 1; }
const char* conv_config_name(int cfg) {
    if (cfg == kDirectCfg) return "direct_valu";
    if (cfg >= kDmabCfg0 && cfg < kDirectCfg) return dma_config_name(cfg - kDmabCfg0, 1);
    if (cfg >= kDma3Cfg0 && cfg < kDmabCfg0) return dma_config_name(cfg - kDma3Cfg0, 3);
    if (cfg >= kBf16Cfg0 && cfg < kDma3Cfg0) return bf16_config_name(cfg - kBf16Cfg0);
    if (cfg >= kMfma3Cfg0 && cfg < kBf16Cfg0) return mfma3_config_name(cfg - kMfma3Cfg0);
    if (cfg >= kSkinnyCfg0 && cfg < kMfma3Cfg0) return skinny_config_name(cfg - kSkinnyCfg0);
    if (cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0) return wino_config_name(cfg - kNumMfmaCfgs);
    if (cfg < 0 || cfg >= kNumMfmaCfgs) return "?";
    return kCfgs[cfg].name;
}

bool conv_config_allowed(int cfg, int precision) {
    const bool bf16 = (cfg >= kBf16Cfg0 && cfg < kDma3Cfg0) || (cfg >= kDmabCfg0 && cfg < kDirectCfg);
    if (!precision) return !bf16;
    const bool split = (cfg >= kMfma3Cfg0 && cfg < kBf16Cfg0) || (cfg >= kDma3Cfg0 && cfg < kDmabCfg0);
    const bool wino = cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0;
    return !split && !wino;
}

static bool is_gemm1x1(const ConvParams& p) {
    return p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
}

bool conv_config_valid(int cfg, const ConvParams& p) {
    if (cfg == kDirectCfg) return (size_t)p.K * ((p.Cout + 3) & ~3) * 4 <= 64 * 1024;
    if (cfg >= kDmabCfg0 && cfg < kDirectCfg) return dma_config_valid(cfg - kDmabCfg0, 1, p);
    if (cfg >= kDma3Cfg0 && cfg < kDmabCfg0) return dma_config_valid(cfg - kDma3Cfg0, 3, p);
    if (cfg >= kBf16Cfg0 && cfg < kDma3Cfg0) return mfma3_config_valid(cfg - kBf16Cfg0, p);
    if (cfg >= kMfma3Cfg0 && cfg < kBf16Cfg0) return mfma3_config_valid(cfg - kMfma3Cfg0, p);
    if (cfg >= kSkinnyCfg0 && cfg < kMfma3Cfg0) return skinny_config_valid(cfg - kSkinnyCfg0, p);
    if (cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0) return wino_config_valid(cfg - kNumMfmaCfgs, p);
    if (cfg < 0 || cfg >= kNumMfmaCfgs) return false;
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w & 15)) return false;
    if (p.Cin % 4) return false;
    if (p.M > 0x7fffffffL - 1024) return false;   // 32-bit pixel indices in the kernel
    if (is_gemm1x1(p)) return true;
    if (p.kh * p.kw > 32) return false;          // per-row tap validity is a 32-bit mask
    return p.Cin % kCfgs[cfg].BK == 0;
}

// Cost model used when no autotuned choice is supplied: MFMA time of the padded tiles on
// 256 CUs vs the L2->LDS operand traffic; the smaller estimate wins.
int conv_pick_config(const ConvParams& p) {
    int best = -1;
    double best_cost = 1e300;
    for (int c = 0; c < kNumMfmaCfgs; ++c) {
        if (!conv_config_valid(c, p)) continue;
        const ConvCfg& g = kCfgs[c];
        const double mb = (double)((p.M + g.BM - 1) / g.BM), nb = (double)((p.Cout + g.BN - 1) / g.BN);
        const double kk = (double)round_up(p.K, g.BK);
        const double blocks = mb * nb;
        const double waves = ceil(blocks / 256.0);
        const double t_mfma = waves * (double)g.BM * g.BN * kk * 2.0 / (157.3e12 / 256.0);
        const double bytes = (mb * g.BM * kk * nb + nb * g.BN * kk * mb) * 4.0 + (double)p.M * p.Cout * 4.0;
        const double t_mem = bytes / 6.0e12;
        const double cost = (t_mfma > t_mem ? t_mfma : t_mem) + 0.25 * (t_mfma < t_mem ? t_mfma : t_mem);
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    // split-bf16 tiles (need the weights' planes): matrix rate derated to what the kernels measured (the staging of
    // the split operands shares the loop), 6 bytes per weight element; not for the smallest problems
    // (bf16 mode: the same tiles with one product instead of six -- matrix rate x 4 in the model, 2 bytes per weight element)
    if (p.w3 && p.M >= 2048 && p.K >= 64) {
        for (int c = 0; c < mfma3_num_configs(); ++c) {
            if (!mfma3_config_valid(c, p)) continue;
            int BM, BN;
            mfma3_tile(c, &BM, &BN);
            const double mb = (double)((p.M + BM - 1) / BM), nb = (double)((p.Cout + BN - 1) / BN);
            const double kk = (double)round_up(p.K, 32);
            const double waves = ceil(mb * nb / 256.0);
            const double t_mfma = waves * (double)BM * BN * kk * 2.0 / ((p.bf16 ? 1000e12 : 260e12) / 256.0);
            const double bytes = mb * BM * kk * nb * 4.0 + nb * BN * kk * mb * (p.bf16 ? 2.0 : 6.0) + (double)p.M * p.Cout * 4.0;
            const double t_mem = bytes / 6.0e12;
            const double cost = (t_mfma > t_mem ? t_mfma : t_mem) + 0.25 * (t_mfma < t_mem ? t_mfma : t_mem);
            if (cost < best_cost) { best_cost = cost; best = (p.bf16 ? kBf16Cfg0 : kMfma3Cfg0) + c; }
        }
    }
    if (best < 0 && conv_config_valid(kDirectCfg, p)) best = kDirectCfg;
    return best;
}

long conv_grid_blocks(int cfg, const ConvParams& p) {
    if (cfg >= kDmabCfg0 && cfg < kDirectCfg) return dma_grid_blocks(cfg - kDmabCfg0, p);
    if (cfg >= kDma3Cfg0 && cfg < kDmabCfg0) return dma_grid_blocks(cfg - kDma3Cfg0, p);
    if (cfg >= kBf16Cfg0 && cfg < kDma3Cfg0) return mfma3_grid_blocks(cfg - kBf16Cfg0, p);
    if (cfg >= kMfma3Cfg0 && cfg < kBf16Cfg0) return mfma3_grid_blocks(cfg - kMfma3Cfg0, p);
    if (cfg >= kSkinnyCfg0 && cfg < kMfma3Cfg0) return skinny_grid_blocks(cfg - kSkinnyCfg0, p);
    if (cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0) return wino_grid_blocks(cfg - kNumMfmaCfgs, p);
    if (cfg < 0 || cfg >= kNumMfmaCfgs) return 0;
    const ConvCfg& g = kCfgs[cfg];
    return ((p.M + g.BM - 1) / g.BM) * ((p.Cout + g.BN - 1) / g.BN);
}
int conv_k_tiles(int cfg, const ConvParams& p) {
    if (cfg >= kDma3Cfg0 && cfg < kDirectCfg) return dma_k_tiles(cfg >= kDmabCfg0 ? 1 : 3, p);
    if (cfg >= kMfma3Cfg0 && cfg < kDma3Cfg0) return mfma3_k_tiles(p);      // (split-bf16 and bf16 tiles)
    if (cfg >= kSkinnyCfg0 && cfg < kMfma3Cfg0) return p.K / 16;
    if (cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0) return wino_k_tiles(p);
    if (cfg < 0 || cfg >= kNumMfmaCfgs) return 0;
    return (p.K + kCfgs[cfg].BK - 1) / kCfgs[cfg].BK;
}

size_t conv_splitk_workspace_floats(const ConvParams& p, int) {
    return p.split_k > 1 ? (size_t)p.split_k * p.M * p.Cout : 0;
}

int conv_launch(const ConvParams& p, int cfg, hipStream_t st) {
    if (p.M == 0 || p.Cout == 0) return SSD_OK;
    if (!conv_config_valid(cfg, p)) {
        set_error("conv2d: config %d (%s) cannot run Cin=%d k=%dx%d stride=%d", cfg, conv_config_name(cfg),
                  p.Cin, p.kh, p.kw, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (cfg >= kDma3Cfg0 && cfg < kDirectCfg) {
        const bool b1 = cfg >= kDmabCfg0;
        const int rc = dma_launch(p, cfg - (b1 ? kDmabCfg0 : kDma3Cfg0), b1 ? 1 : 3, st);
        if (rc || p.split_k <= 1) return rc;
        return launch_splitk_reduce(p, st);
    }
    if (cfg >= kMfma3Cfg0 && cfg < kDma3Cfg0) {
        const bool bf16 = cfg >= kBf16Cfg0;
        const int rc = mfma3_launch(p, cfg - (bf16 ? kBf16Cfg0 : kMfma3Cfg0), st, bf16);
        if (rc || p.split_k <= 1) return rc;
        return launch_splitk_reduce(p, st);
    }
    if (cfg >= kSkinnyCfg0 && cfg < kMfma3Cfg0) return skinny_launch(p, cfg - kSkinnyCfg0, st);
    if (cfg >= kNumMfmaCfgs && cfg < kSkinnyCfg0) {
        const int rc = wino_launch(p, cfg - kNumMfmaCfgs, st);
        if (rc || p.split_k <= 1) return rc;
        return launch_splitk_reduce(p, st);
    }
    if (cfg == kDirectCfg && stem_ok(p)) {
        constexpr int PX = 2;
        const long blocks = cdiv(p.M, 256 * PX);
        dim3 grid((unsigned)(blocks < 65535 ? blocks : 65535), p.Cout / 32);
        hipLaunchKernelGGL(conv_stem_kernel<PX>, grid, dim3(256), 0, st, p);
        SSD_LAUNCH_CHECK();
        return SSD_OK;
    }
    if (cfg == kDirectCfg) {
        const long total = p.M * ((p.Cout + 3) / 4);
        const int blocks = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
        const size_t lds = (size_t)p.K * ((p.Cout + 3) & ~3) * 4;
        hipLaunchKernelGGL(conv_direct_kernel, dim3(blocks), dim3(256), lds, st, p);
        SSD_LAUNCH_CHECK();
        return SSD_OK;
    }
    const ConvCfg& g = kCfgs[cfg];
    const long mb = (p.M + g.BM - 1) / g.BM;
    const int nb = (p.Cout + g.BN - 1) / g.BN;
    SSD_UNSUPPORTED_IF(mb * nb > 0x7fffffffL, "conv2d: grid too large");
    dim3 grid((unsigned)(mb * nb), p.split_k > 1 ? p.split_k : 1);
    conv_kernel_t k = is_gemm1x1(p) ? g.gemm : g.general;
    hipLaunchKernelGGL(k, grid, dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    if (p.split_k > 1) return launch_splitk_reduce(p, st);
    return SSD_OK;
}

int launch_splitk_reduce(const ConvParams& p, hipStream_t st) {
    const long total = p.M * p.Cout;
    const bool vec = (total % 4 == 0) && (((uintptr_t)p.partial & 15) == 0);
    const long groups = vec ? total / 4 : total;
    const int blocks = (int)(cdiv(groups, 256) < 8192 ? cdiv(groups, 256) : 8192);
    if (vec) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    // only the 4-wide form writes the output's bf16 planes: on the scalar path (an unaligned caller workspace) they come
    // from a pass of their own, never silently stale
    if (!vec && p.op && p.out) return launch_split_planes(p.out, p.M * p.Cout, p.Cout, p.op_np, p.op, p.op_plane, st);
    return SSD_OK;
}

int launch_pack_weights(const float* hwio, int K, int Cout, int Kpad, int Npad, float* packed, hipStream_t st) {
    const long total = (long)Npad * Kpad;
    if (total == 0) return SSD_OK;
    const int blocks = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, hwio, K, Cout, Kpad, Npad, packed);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int fill_conv_params(const ssd_conv_desc* d, ConvParams* p) {
    SSD_CHECK_ARG(d != nullptr, "conv2d: descriptor is NULL");
    SSD_CHECK_ARG(d->B >= 0 && d->H >= 1 && d->W >= 1 && d->Cin >= 1 && d->Cout >= 1,
                  "conv2d: bad tensor sizes B=%d H=%d W=%d Cin=%d Cout=%d", d->B, d->H, d->W, d->Cin, d->Cout);
    SSD_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->dilation >= 1,
                  "conv2d: bad kernel %dx%d stride %d dilation %d", d->kh, d->kw, d->stride, d->dilation);
    SSD_CHECK_ARG(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_b >= 0 && d->pad_r >= 0, "conv2d: negative padding");
    SSD_CHECK_ARG(d->act >= 0 && d->act <= 2, "conv2d: unknown activation %d", d->act);
    const int Ho = ssd_conv_out_size(d->H, d->kh, d->stride, d->dilation, d->pad_t, d->pad_b);
    const int Wo = ssd_conv_out_size(d->W, d->kw, d->stride, d->dilation, d->pad_l, d->pad_r);
    SSD_CHECK_ARG(Ho >= 1 && Wo >= 1, "conv2d: empty output (%d x %d)", Ho, Wo);
    *p = ConvParams{};
    p->B = d->B; p->H = d->H; p->W = d->W; p->Cin = d->Cin; p->Ho = Ho; p->Wo = Wo; p->Cout = d->Cout;
    p->kh = d->kh; p->kw = d->kw; p->stride = d->stride; p->dil = d->dilation;
    p->pad_t = d->pad_t; p->pad_l = d->pad_l;
    p->K = d->kh * d->kw * d->Cin;
    p->Kpad = conv_kpad(p->K);
    p->Npad = conv_npad(d->Cout);
    p->M = (long)d->B * Ho * Wo;
    p->act = d->act;
    p->split_k = 1;
    return SSD_OK;
}

}  // namespace ssd

using namespace ssd;

extern "C" {

int ssd_same_pads(int in, int k, int stride, int dilation, int* before, int* after) {
    if (in < 1 || k < 1 || stride < 1 || dilation < 1) return SSD_E_INVALID;
    const int out = (in + stride - 1) / stride;
    const int keff = (k - 1) * dilation + 1;
    int p = (out - 1) * stride + keff - in;
    if (p < 0) p = 0;
    if (before) *before = p / 2;
    if (after) *after = p - p / 2;
    return out;
}

int ssd_conv_out_size(int in, int k, int stride, int dilation, int pad_before, int pad_after) {
    if (in < 1 || k < 1 || stride < 1 || dilation < 1) return 0;
    const int keff = (k - 1) * dilation + 1;
    const int span = in + pad_before + pad_after - keff;
    if (span < 0) return 0;
    return span / stride + 1;
}

size_t ssd_conv_packed_weight_floats(int kh, int kw, int Cin, int Cout) {
    if (kh < 1 || kw < 1 || Cin < 1 || Cout < 1) return 0;
    return conv_packed_floats(kh * kw * Cin, Cout);
}

int ssd_conv_pack_weights(const float* hwio_dev, int kh, int kw, int Cin, int Cout, float* packed_dev,
                          void* stream) {
    SSD_CHECK_ARG(hwio_dev && packed_dev, "ssd_conv_pack_weights: NULL pointer");
    SSD_CHECK_ARG(kh >= 1 && kw >= 1 && Cin >= 1 && Cout >= 1, "ssd_conv_pack_weights: bad shape");
    const int K = kh * kw * Cin;
    int rc = launch_pack_weights(hwio_dev, K, Cout, conv_kpad(K), conv_npad(Cout), packed_dev, (hipStream_t)stream);
    if (!rc) rc = launch_pack_split(packed_dev, K, Cout, (hipStream_t)stream);
    return rc;
}

int ssd_conv_num_configs(void) { return conv_num_configs(); }
const char* ssd_conv_config_name(int cfg) { return conv_config_name(cfg); }

int ssd_conv2d_ex(const ssd_conv_desc* d, const float* in_dev, const float* packed_w_dev,
                  const float* scale_dev, const float* shift_dev, const float* residual_dev, float* out_dev,
                  long out_batch_stride, long out_pixel_stride, int config, int split_k,
                  float* splitk_ws_dev, void* stream) {
    ConvParams p;
    int rc = fill_conv_params(d, &p);
    if (rc) return rc;
    if (p.M == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && packed_w_dev && out_dev, "conv2d: NULL pointer");
    SSD_CHECK_ARG(!d->has_residual || residual_dev, "conv2d: has_residual set but residual is NULL");
    p.in = in_dev; p.w = packed_w_dev; p.scale = scale_dev; p.shift = shift_dev;
    p.w3 = conv_split_planes(packed_w_dev, p.K, p.Cout);
    p.residual = d->has_residual ? residual_dev : nullptr;
    p.out = out_dev;
    p.out_pixel_stride = out_pixel_stride > 0 ? out_pixel_stride : p.Cout;
    p.out_batch_stride = out_batch_stride > 0 ? out_batch_stride : (long)p.Ho * p.Wo * p.out_pixel_stride;
    p.vec_store = (((uintptr_t)out_dev & 15) == 0) && (p.out_pixel_stride % 4 == 0) && (p.out_batch_stride % 4 == 0);
    if (split_k > 1) {
        SSD_CHECK_ARG(splitk_ws_dev != nullptr, "conv2d: split_k > 1 needs a workspace");
        p.split_k = split_k;
        p.partial = splitk_ws_dev;
    }
    const int cfg = config >= 0 ? config : conv_pick_config(p);
    SSD_UNSUPPORTED_IF(cfg < 0, "conv2d: no kernel for Cin=%d Cout=%d k=%dx%d", p.Cin, p.Cout, p.kh, p.kw);
    if (cfg == conv_num_configs() - 1) p.split_k = 1;
    return conv_launch(p, cfg, (hipStream_t)stream);
}

int ssd_split_planes(const float* x_dev, long n, int channels, int planes, void* planes_dev, long plane_stride, void* stream) {
    SSD_CHECK_ARG(x_dev && planes_dev && n >= 0, "ssd_split_planes: bad arguments");
    SSD_CHECK_ARG(plane_stride >= n && plane_stride % 8 == 0, "ssd_split_planes: plane_stride %ld must be >= n and a multiple of 8", plane_stride);
    SSD_CHECK_ARG((((uintptr_t)x_dev | (uintptr_t)planes_dev) & 15) == 0, "ssd_split_planes: pointers must be 16-byte aligned");
    return launch_split_planes(x_dev, n, channels, planes, static_cast<short*>(planes_dev), plane_stride, (hipStream_t)stream);
}

int ssd_join_planes(const void* planes_dev, long n, int channels, int planes, long plane_stride, float* x_dev, void* stream) {
    SSD_CHECK_ARG(x_dev && planes_dev && n >= 0 && (planes == 1 || planes == 3) && plane_stride >= n, "ssd_join_planes: bad arguments");
    return launch_join_planes(static_cast<const short*>(planes_dev), n, channels, planes, plane_stride, x_dev, (hipStream_t)stream);
}

int ssd_conv2d_planes(const ssd_conv_desc* d, const void* in_planes_dev, int planes, long in_plane_stride,
                      const float* packed_w_dev, const float* scale_dev, const float* shift_dev, const float* residual_dev,
                      float* out_dev, long out_batch_stride, long out_pixel_stride, void* out_planes_dev,
                      long out_plane_stride, int config, int split_k, float* splitk_ws_dev, void* stream) {
    ConvParams p;
    int rc = fill_conv_params(d, &p);
    if (rc) return rc;
    if (p.M == 0) return SSD_OK;
    SSD_CHECK_ARG(in_planes_dev && packed_w_dev && out_dev, "conv2d_planes: NULL pointer");
    SSD_CHECK_ARG(planes == 1 || planes == 3, "conv2d_planes: planes must be 1 (bf16 mode) or 3 (exact split)");
    SSD_CHECK_ARG(in_plane_stride >= (long)p.B * p.H * p.W * p.Cin, "conv2d_planes: in_plane_stride shorter than the tensor");
    SSD_CHECK_ARG(!d->has_residual || residual_dev, "conv2d_planes: has_residual set but residual is NULL");
    SSD_CHECK_ARG(config >= 0 && config < conv_num_configs() && conv_config_is_dma(config), "conv2d_planes: config %d is not an LDS-DMA tile", config);
    p.xp = static_cast<const short*>(in_planes_dev); p.xp_plane = in_plane_stride; p.xp_np = planes;
    p.bf16 = planes == 1;
    p.w = packed_w_dev; p.scale = scale_dev; p.shift = shift_dev;
    p.w3 = conv_split_planes(packed_w_dev, p.K, p.Cout);
    p.residual = d->has_residual ? residual_dev : nullptr;
    p.out = out_dev;
    p.out_pixel_stride = out_pixel_stride > 0 ? out_pixel_stride : p.Cout;
    p.out_batch_stride = out_batch_stride > 0 ? out_batch_stride : (long)p.Ho * p.Wo * p.out_pixel_stride;
    p.vec_store = (((uintptr_t)out_dev & 15) == 0) && (p.out_pixel_stride % 4 == 0) && (p.out_batch_stride % 4 == 0);
    if (out_planes_dev) {
        SSD_CHECK_ARG(p.out_pixel_stride == p.Cout && p.out_batch_stride == (long)p.Ho * p.Wo * p.Cout && p.Cout % 32 == 0,
                      "conv2d_planes: plane output needs a dense [B,Ho,Wo,Cout] destination with Cout %% 32 == 0 (whole channel slices)");
        SSD_CHECK_ARG(out_plane_stride >= p.M * p.Cout && out_plane_stride % 8 == 0, "conv2d_planes: bad out_plane_stride");
        p.op = static_cast<short*>(out_planes_dev); p.op_plane = out_plane_stride; p.op_np = planes;
    }
    if (split_k > 1) {
        SSD_CHECK_ARG(splitk_ws_dev != nullptr, "conv2d_planes: split_k > 1 needs a workspace");
        p.split_k = split_k;
        p.partial = splitk_ws_dev;
    }
    return conv_launch(p, config, (hipStream_t)stream);
}

size_t ssd_conv_wino_weight_floats(int Cin, int Cout) {
    if (Cin < 1 || Cout < 1) return 0;
    return wino_weight_floats(Cin, Cout);
}

int ssd_conv_wino_pack_weights(const float* hwio_dev, int Cin, int Cout, float* wino_w_dev, void* stream) {
    SSD_CHECK_ARG(hwio_dev && wino_w_dev && Cin >= 1 && Cout >= 1, "ssd_conv_wino_pack_weights: bad arguments");
    SSD_HIP(hipMemsetAsync(wino_w_dev, 0, wino_weight_floats(Cin, Cout) * sizeof(float), (hipStream_t)stream));
    return launch_wino_pack(hwio_dev, Cin, Cout, conv_npad(Cout), 0, wino_w_dev, (hipStream_t)stream);
}

int ssd_conv_wino_num_configs(void) { return wino_num_configs(); }

int ssd_conv2d_wino(const ssd_conv_desc* d, const float* in_dev, const float* wino_w_dev, const float* scale_dev,
                    const float* shift_dev, float* out_dev, long out_batch_stride, long out_pixel_stride,
                    int wino_config, int split_k, float* splitk_ws_dev, void* stream) {
    ConvParams p;
    int rc = fill_conv_params(d, &p);
    if (rc) return rc;
    if (p.M == 0) return SSD_OK;
    SSD_CHECK_ARG(in_dev && wino_w_dev && out_dev, "conv2d_wino: NULL pointer");
    SSD_CHECK_ARG(!d->has_residual, "conv2d_wino: residual is not supported");
    p.in = in_dev; p.wino_w = wino_w_dev; p.scale = scale_dev; p.shift = shift_dev;
    p.out = out_dev;
    p.out_pixel_stride = out_pixel_stride > 0 ? out_pixel_stride : p.Cout;
    p.out_batch_stride = out_batch_stride > 0 ? out_batch_stride : (long)p.Ho * p.Wo * p.out_pixel_stride;
    p.vec_store = (((uintptr_t)out_dev & 15) == 0) && (p.out_pixel_stride % 4 == 0) && (p.out_batch_stride % 4 == 0);
    if (split_k > 1) {
        SSD_CHECK_ARG(splitk_ws_dev != nullptr, "conv2d_wino: split_k > 1 needs a workspace");
        SSD_CHECK_ARG(split_k <= p.Cin / 16, "conv2d_wino: split_k %d exceeds the %d channel slabs", split_k, p.Cin / 16);
        p.split_k = split_k;
        p.partial = splitk_ws_dev;
    }
    SSD_UNSUPPORTED_IF(!wino_applicable(p), "conv2d_wino: needs a 3x3 stride-1 dilation-1 conv with Cin %% 16 == 0");
    const int cfg = conv_num_mfma_configs() + (wino_config >= 0 ? wino_config : 0);
    return conv_launch(p, cfg, (hipStream_t)stream);
}

int ssd_conv2d(const ssd_conv_desc* d, const float* in_dev, const float* packed_w_dev, const float* scale_dev,
               const float* shift_dev, const float* residual_dev, float* out_dev, long out_batch_stride,
               long out_pixel_stride, void* stream) {
    return ssd_conv2d_ex(d, in_dev, packed_w_dev, scale_dev, shift_dev, residual_dev, out_dev,
                         out_batch_stride, out_pixel_stride, -1, 1, nullptr, stream);
}

}  // extern "C"

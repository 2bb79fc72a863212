// "Skinny" dense convolution for the SSD tail (extras 2-4, head levels 3-6 -- reference
// models/ssd_mobilenet_v2.py:25-32, models/header.py:60-61 -- and every other conv whose GEMM has few
// rows M = B*Ho*Wo and a long K = kh*kw*Cin): out[M, N] = im2col(X)[M, K] * W[K, N] with K = 512 ... 4608
// and M = 64 ... 1600 at B = 64.  The tiled implicit-GEMM kernel (ssd_conv.hip) fills the 256 CUs for
// these shapes only by splitting K over the GRID, i.e. a slab of partial sums + a second launch
// (splitk_reduce_kernel): 20 launches of 5-25 us for the ten tail layers, run one behind the other.
// Here the K split happens INSIDE the workgroup:
//
//   workgroup = 8 waves, output tile = (MT*16 pixels) x (NT*16 channels); wave w walks the 16-wide k
//   groups w, w + 8, ... of the WHOLE K range with MFMA fragments loaded STRAIGHT from global memory
//   (weights packed [Npad][Kpad] K-contiguous = A operand, NHWC pixels = B operand; both 16-byte loads,
//   L2 resident; no LDS staging, no barrier in the main loop; two k groups in flight per wave); the
//   eight partial accumulator sets meet in LDS and are summed in wave order (deterministic), followed by
//   the full epilogue (folded BN / bias, activation, residual, the SSD heads' dual strided destination).
//
// One launch per layer, no slab, no reduce kernel.  fp32 v_mfma_f32_16x16x4_f32 (exact).
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kSkWaves = 8;

__device__ __forceinline__ float sk_act(float v, int act) {
    if (act == SSD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == SSD_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

template <int MT, int NT>
__global__ __launch_bounds__(kSkWaves * 64) void conv_skinny_kernel(const ConvParams p) {
    constexpr int TILES = MT * NT;
    __shared__ __attribute__((aligned(16))) float red[kSkWaves * TILES * 64 * 4];

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = (p.Cout + NT * 16 - 1) / (NT * 16);
    const int mb = blockIdx.x / nbn, nbk = blockIdx.x - mb * nbn;
    const int m0 = mb * MT * 16, n0 = nbk * NT * 16;
    const int HoWo = p.Ho * p.Wo, Cin = p.Cin;
    const int ntaps = p.kh * p.kw;

    // ---- this lane's pixels (B operand rows): base offset of the window origin, per-tap validity mask
    long base[MT];
    unsigned mask[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = m0 + mi * 16 + l15;
        const bool mv = m < (int)p.M;
        const int mm = mv ? m : 0;
        const int b = mm / HoWo, pix = mm - b * HoWo;
        const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
        base[mi] = (((long)b * p.H + iy0) * p.W + ix0) * Cin + g4 * 4;
        unsigned mk = 0;
        for (int t = 0; t < ntaps; ++t) {
            const int ky = t / p.kw, kx = t - ky * p.kw;
            const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
            if (mv && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mk |= 1u << t;
        }
        mask[mi] = mk;
    }
    // ---- this lane's weight rows (A operand): rows >= Cout of the packed matrix are zero up to Npad
    const float* wrow[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int row = n0 + ni * 16 + l15;
        wrow[ni] = p.w + (long)(row < p.Npad ? row : p.Npad - 1) * p.Kpad + g4 * 4;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- k groups g = wave, wave + 8, ...: (tap, channel offset) tracked incrementally (scalar)
    const int KG = p.K / 16;
    int g = wave;
    int tap = (g * 16) / Cin, ci0 = g * 16 - tap * Cin;
    int ky = tap / p.kw, kx = tap - ky * p.kw;
    struct Frag {
        f32x4 a[NT], b[MT];
    };
    auto load = [&](Frag& f, int gg, int t_, int ky_, int kx_, int ci_) {
        const long toff = ((long)ky_ * p.dil * p.W + (long)kx_ * p.dil) * Cin + ci_;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) f.a[ni] = *reinterpret_cast<const f32x4*>(wrow[ni] + (long)gg * 16);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const bool v = (mask[mi] >> t_) & 1u;
            // out-of-range taps read the tensor's first bytes (always mapped) and are replaced by zero
            const f32x4 x = *reinterpret_cast<const f32x4*>(p.in + (v ? base[mi] + toff : (long)(g4 * 4)));
            f.b[mi] = v ? x : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto advance = [&]() {          // to this wave's next k group
        g += kSkWaves;
        ci0 += kSkWaves * 16;
        while (ci0 >= Cin) {
            ci0 -= Cin;
            ++tap;
            if (++kx == p.kw) { kx = 0; ++ky; }
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ni][s], f.b[mi][s], acc[mi][ni], 0, 0, 0);
    };
    Frag f0, f1;
    if (g < KG) load(f0, g, tap, ky, kx, ci0);
    while (g < KG) {
        advance();
        if (g < KG) load(f1, g, tap, ky, kx, ci0);
        mma(f0);
        if (g >= KG) break;
        advance();
        if (g < KG) load(f0, g, tap, ky, kx, ci0);
        mma(f1);
    }

    // ---- the 8 partial sums meet in LDS; wave w < TILES owns output tile w and adds them in wave order
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
            *reinterpret_cast<f32x4*>(red + ((wave * TILES + mi * NT + ni) * 64 + lane) * 4) = acc[mi][ni];
    __syncthreads();
    for (int tile = wave; tile < TILES; tile += kSkWaves) {
        const int mi = tile / NT, ni = tile - mi * NT;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kSkWaves; ++w) v = v + *reinterpret_cast<const f32x4*>(red + ((w * TILES + tile) * 64 + lane) * 4);
        // epilogue: lane holds out[m][n .. n + 3]
        const int m = m0 + mi * 16 + l15;
        const int n = n0 + ni * 16 + g4 * 4;
        if (m >= (int)p.M || n >= p.Cout) continue;
        const int b = m / HoWo, pix = m - b * HoWo;
        float* orow = p.out + (long)b * p.out_batch_stride + (long)pix * p.out_pixel_stride;
        float* orow2 = p.n_split ? p.out2 + (long)b * p.out2_batch_stride + (long)pix * p.out2_pixel_stride - p.n_split : nullptr;
        const bool straddle = p.n_split && n < p.n_split && n + 3 >= p.n_split;
        if (n + 3 < p.Cout && !straddle) {
            if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) v = v + *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = sk_act(v[j], p.act);
            if (p.residual) {
                const float* rr = p.residual + (long)m * p.Cout + n;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rr[j];
            }
            const bool side2 = p.n_split && n >= p.n_split;
            float* dst = (side2 ? orow2 : orow) + n;
            if ((side2 ? p.vec_store2 : p.vec_store) && ((((uintptr_t)dst) & 15) == 0)) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            }
        } else {
            for (int j = 0; j < 4; ++j) {
                if (n + j >= p.Cout) break;
                float t = v[j];
                if (p.scale) t = t * p.scale[n + j];
                if (p.shift) t = t + p.shift[n + j];
                t = sk_act(t, p.act);
                if (p.residual) t += p.residual[(long)m * p.Cout + n + j];
                float* drow = (p.n_split && n + j >= p.n_split) ? orow2 : orow;
                drow[n + j] = t;
            }
        }
    }
}

typedef void (*skinny_kernel_t)(const ConvParams);
struct SkinnyCfg {
    const char* name;
    int tm, tn;
    skinny_kernel_t fn;
};
#define SCFG(MT, NT) {"skinny_" #MT "x" #NT, MT * 16, NT * 16, conv_skinny_kernel<MT, NT>}
const SkinnyCfg kSkinny[] = {
    SCFG(2, 2),    // 32 pixels x 32 channels
    SCFG(1, 2),    // 16 x 32
    SCFG(1, 4),    // 16 x 64
    SCFG(2, 4),    // 32 x 64
    SCFG(4, 2),    // 64 x 32
};
constexpr int kNumSkinny = sizeof(kSkinny) / sizeof(kSkinny[0]);

}  // namespace

int skinny_num_configs() { return kNumSkinny; }
const char* skinny_config_name(int i) { return (i >= 0 && i < kNumSkinny) ? kSkinny[i].name : "?"; }

// Small-M, long-K dense convs only (the candidates are timed by the finalize-time autotune: keep the list
// to the shapes the kernel is for); K in whole 16-wide groups inside one tap.
bool skinny_config_valid(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumSkinny) return false;
    if (p.split_k > 1 || p.Cin % 16 != 0 || p.kh * p.kw > 32 || p.K != p.kh * p.kw * p.Cin) return false;
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w & 15) || (p.Kpad & 3)) return false;
    if (p.M > 8192 || p.K < 256) return false;
    if ((long)p.B * p.H * p.W * p.Cin >= 0x7fffffffL) return false;
    return true;
}
long skinny_grid_blocks(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumSkinny) return 0;
    return ((p.M + kSkinny[i].tm - 1) / kSkinny[i].tm) * ((p.Cout + kSkinny[i].tn - 1) / kSkinny[i].tn);
}
int skinny_launch(const ConvParams& p, int i, hipStream_t st) {
    if (!skinny_config_valid(i, p)) {
        set_error("conv2d: skinny config %d cannot run this convolution", i);
        return SSD_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kSkinny[i].fn, dim3((unsigned)skinny_grid_blocks(i, p)), dim3(kSkWaves * 64), 0, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

// Row-band inverted-residual kernel (blocks 3-6, see ssd_bandblock.hip) with the two 1x1 convolutions on the BF16
// matrix cores at FP32 accuracy (exact three-way bf16 split of both operands, six MFMAs per product: ssd_bf16x3.h).
// The network's results stay within the 1e-4 contract; they are not bit-identical to the fp32-MFMA kernels.
//
// Organisation = ssd_bandblock.hip (full-width bands, X fragments in registers, E double-buffered in swizzled 64-byte
// LDS rows, one barrier per 16-channel chunk, depthwise per lane as the project's B fragment) with these differences:
//   * X is split once per band into three bf16 planes (lane = pixel x 8 channels g4*8.., K padded to 32)
//   * expand: 16 expanded channels x K = 32 = ONE k-step: 6 MFMAs per pixel tile and chunk; weights pre-split at
//     finalize (split3_we_kernel), A fragments straight from L2
//   * project: K = 32 = TWO consecutive 16-channel chunks: the depthwise output of the even chunk waits in registers,
//     after the odd chunk the lane's 8 values are split and 6 x NT MFMAs run (k-slot (g4, j): j < 4 -> even chunk channel
//     g4*4 + j, j >= 4 -> odd chunk; the pre-split project weights are packed in that order, split3_wp_kernel)
#include <cstdlib>

#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kB3Threads = 512;
constexpr int kB3C = 16;

__device__ __forceinline__ void b3_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int b3_ne(int T) { return 8 + T * 8 * 16; }

// out[plane][row][32]: We (BatchNorm scale folded, packed [Ce][kpad]) split, K = Cin padded to 32
__global__ __launch_bounds__(256) void split3_we_kernel(const float* __restrict__ we, int Ce, int Cin, int kpad, short* __restrict__ out) {
    const long total = (long)Ce * 32;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int row = (int)(e >> 5), k = (int)(e & 31);
        const float v = k < Cin ? we[(long)row * kpad + k] : 0.f;
        short h, m, l;
        split1(v, h, m, l);
        out[e] = h;
        out[total + e] = m;
        out[2 * total + e] = l;
        out[3 * total + e] = rne1(v);        // plane 3: the bf16 rounding (precision-1 form of the kernel)
    }
}
// out[plane][row][pair][g4][8]: Wp (packed [npad][kpad], K = Ce) split, k-slots in chunk-pair order
__global__ __launch_bounds__(256) void split3_wp_kernel(const float* __restrict__ wp, int npad, int Ce, int kpad, int npairs,
                                                       short* __restrict__ out) {
    const long total = (long)npad * npairs * 32;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int j = (int)(e & 7), g4 = (int)((e >> 3) & 3);
        const long rp = e >> 5;
        const int pair = (int)(rp % npairs), row = (int)(rp / npairs);
        const int ch = (2 * pair + (j >> 2)) * 16 + g4 * 4 + (j & 3);
        const float v = ch < Ce ? wp[(long)row * kpad + ch] : 0.f;
        short h, m, l;
        split1(v, h, m, l);
        out[e] = h;
        out[total + e] = m;
        out[2 * total + e] = l;
        out[3 * total + e] = rne1(v);
    }
}

// NP = 3: fp32 results through the exact three-way split (six MFMAs per product); NP = 1: the net's bf16 mode (operands
// rounded once, one MFMA per product, plane 3 of the packed weights) -- a third of the plane registers, so blocks 1-2
// (T = 9 / 8 tiles per wave) fit as well
// WDMA (round 6, the whole-image kernel's recipe): the weight fragments reach LDS by LDS-DMA (two We stages, one Wp stage; 1 KB
// blocks of 16 rows x 32 bf16, quad-swizzled through the per-lane source offset) instead of waiting in registers one chunk ahead:
// 36 - 60 registers less, no per-chunk 64-bit addresses
typedef __attribute__((address_space(3))) void* b3_lds_dst_t;
template <int CIN, int NT, int T, int TO, int S, int P, int NP, bool WDMA>
__device__ __forceinline__ void band3_body(const FusedBlockParams& p, char* __restrict__ smem) {
    constexpr int WPL = NP == 1 ? 3 : 0;
    static_assert(P % 8 == 0 && CIN <= 32 && CIN % 8 == 0, "");
    constexpr int NE = b3_ne(T);
    constexpr int EBUF = NE * kB3C * 4;

    char* Es = smem;
    float* Ps = reinterpret_cast<float*>(smem + 2 * EBUF);       // [11][Ce]

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.bands;
    const int items = p.B * nb;
    const int bid = (items & 7) == 0 ? (int)(blockIdx.x & 7) * (items >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // XCD-aware
    const int img = bid / nb, band = bid - img * nb;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, Ce = p.Ce;
    const int ro0 = band * Ho / nb, R = (band + 1) * Ho / nb - ro0;
    const int ri0 = S * ro0 - p.pad_t;
    const int HB = S * (R - 1) + 3, QB = HB * P;
    const int npt = (QB + 15) >> 4;
    const int Po = Wo + 1, npo = (R * Po + 15) >> 4;
    const int nchunk = Ce / kB3C, npairs = (nchunk + 1) >> 1;
    const int nti = npt > wave ? (npt - wave + 7) >> 3 : 0;
    const int nto = npo > wave ? (npo - wave + 7) >> 3 : 0;
    const long plane_e = (long)Ce * 32, plane_p = (long)p.npad_p * npairs * 32;

    for (int u = tid; u < 11 * (Ce / 4); u += kB3Threads) {
        const int row = u / (Ce / 4), c4 = (u - row * (Ce / 4)) * 4;
        const float* src = row == 0 ? p.eh : row == 10 ? p.dh : p.wd + (long)(row - 1) * Ce;
        *reinterpret_cast<f32x4*>(Ps + row * Ce + c4) = *reinterpret_cast<const f32x4*>(src + c4);
    }
    if (tid < 64) *reinterpret_cast<f32x4*>(Es + (tid >> 5) * EBUF + (tid & 31) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the wave's input tiles: X split once into three bf16 planes (lane = pixel x channels g4*8 .. g4*8 + 7)
    BP<NP> xs[T];
    unsigned realm = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = t * 8 + wave;
        const int q = tile * 16 + l15;
        const int rb = q / P, c = q - rb * P;
        const int ri = ri0 + rb;
        const bool real = tile < npt && rb < HB && c < W && (unsigned)ri < (unsigned)H;
        realm |= real ? (1u << t) : 0u;
        const bool have = real && g4 * 8 < CIN;
        const float* xp = p.x + (((long)img * H + (real ? ri : 0)) * W + (real ? c : 0)) * CIN + (g4 * 8 < CIN ? g4 * 8 : 0);
        const f32x4 a = have ? *reinterpret_cast<const f32x4*>(xp) : f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 b = have ? *reinterpret_cast<const f32x4*>(xp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        xs[t] = splitN<NP>(a, b);
    }
    const int ew = (8 + wave * 16 + l15) * 64 + ((g4 ^ ((l15 >> 1) & 3)) << 4);

    int ea[TO][3];
    int opix[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const int tile = t * 8 + wave;
        const int qo = tile * 16 + l15;
        const int rol = qo / Po, co = qo - rol * Po;
        const bool realo = tile < npo && rol < R && co < Wo;
        opix[t] = realo ? (ro0 + rol) * Wo + co : -1;
        const int qor = realo ? (S * rol) * P + S * co - p.pad_l : 0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int e = 8 + qor + dx;
            ea[t][dx] = e * 64 + ((g4 ^ ((e >> 1) & 3)) << 4);
        }
    }

    short* Wes = reinterpret_cast<short*>(Ps + 11 * Ce);         // WDMA: [2][NP] blocks of 512 bf16
    short* Wps = Wes + 2 * NP * 512;                             //       [NP][NT] blocks
    const int fslot = l15 * 32 + ((g4 ^ ((l15 >> 1) & 3)) * 8);  // the lane's fragment slot inside a block
    const int dr = lane >> 2, dq8 = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.we3 + WPL * plane_e), 0, (int)(NP * plane_e * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.wp3 + WPL * plane_p), 0, (int)(NP * plane_p * 2), 0x00020000);
    const int voff_e = (dr * 32 + dq8) * 2, voff_p = (dr * npairs * 32 + dq8) * 2;
    auto dma_we = [&](int j, int stage) {
        for (int b = wave; b < NP; b += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_e, (b3_lds_dst_t)(Wes + (stage * NP + b) * 512), 16, voff_e,
                                                     (int)((b * plane_e + (long)j * kB3C * 32) * 2), 0, 0);
    };
    auto dma_wp = [&](int pair) {
        for (int b = wave; b < NP * NT; b += 8) {
            const int pl = b / NT, ni = b - pl * NT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (b3_lds_dst_t)(Wps + b * 512), 16, voff_p,
                                                     (int)((pl * plane_p + ((long)ni * 16 * npairs + pair) * 32) * 2), 0, 0);
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto load_we = [&](int j) {
        BP<NP> w;
        if constexpr (WDMA) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) w.p[pl] = *reinterpret_cast<const bf16x8*>(Wes + ((j & 1) * NP + pl) * 512 + fslot);
        } else {
            const short* base = p.we3 + WPL * plane_e + ((long)(j * kB3C + l15) * 32 + g4 * 8);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) w.p[pl] = *reinterpret_cast<const bf16x8*>(base + pl * plane_e);
        }
        return w;
    };
    auto load_wp = [&](BP<NP> (&w)[NT], int pair) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if constexpr (WDMA) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) w[ni].p[pl] = *reinterpret_cast<const bf16x8*>(Wps + (pl * NT + ni) * 512 + fslot);
            } else {
                const short* base = p.wp3 + WPL * plane_p + ((((long)(ni * 16 + l15) * npairs + pair) * 4 + g4) * 8);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) w[ni].p[pl] = *reinterpret_cast<const bf16x8*>(base + pl * plane_p);
            }
        }
    };
    if constexpr (WDMA) {
        dma_we(0, 0);
        if (nchunk > 1) dma_we(1, 1);
        dma_wp(0);
        dma_wait();
    }
    __syncthreads();

    auto expand = [&](int j, const BP<NP>& wa) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + j * kB3C + g4 * 4);
        char* eb = Es + (j & 1) * EBUF + ew;
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += 2) {
            if (t0 >= nti) break;
            const int t1 = t0 + 1 < T ? t0 + 1 : t0;
            f32x4 a0 = mmaN<NP>(wa, xs[t0], sh), a1 = sh;
            if (t0 + 1 < T) a1 = mmaN<NP>(wa, xs[t1], sh);
            const float hi0 = (realm >> t0) & 1u ? 6.0f : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) a0[e] = __builtin_amdgcn_fmed3f(a0[e], 0.0f, hi0);
            *reinterpret_cast<f32x4*>(eb + t0 * 8192) = a0;
            if (t0 + 1 < T && t0 + 1 < nti) {
                const float hi1 = (realm >> t1) & 1u ? 6.0f : 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) a1[e] = __builtin_amdgcn_fmed3f(a1[e], 0.0f, hi1);
                *reinterpret_cast<f32x4*>(eb + t1 * 8192) = a1;
            }
        }
    };

    f32x4 acc[TO][NT];
    f32x4 dprev[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        dprev[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[t][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if constexpr (WDMA) {
        expand(0, load_we(0));
        for (int i = 0; i < nchunk; ++i) {
            dma_wait();                 // the copies of an iteration ago have landed ...
            b3_lds_barrier();           // ... and are visible; E(i) is complete; everyone is done reading E(i - 1)
            const bool odd = i & 1, last = i + 1 == nchunk;
            const bool flush = odd || last;
            if (i + 2 < nchunk) dma_we(i + 2, i & 1);           // the stage expand(i) read before this barrier
            if (!odd && i > 0) dma_wp(i >> 1);                  // everyone projected the pair before at iteration i - 1
            if (!odd && last && i > 0) {                        // a lone last chunk projects in the iteration its weights are issued in
                dma_wait();
                b3_lds_barrier();
            }
            const char* eb = Es + (i & 1) * EBUF;
            f32x4 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(Ps + (1 + k) * Ce + i * kB3C + g4 * 4);
            const f32x4 dh = *reinterpret_cast<const f32x4*>(Ps + 10 * Ce + i * kB3C + g4 * 4);
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                if (t >= nto) break;
                f32x4 d = dh;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        d += *reinterpret_cast<const f32x4*>(eb + ea[t][dx] + dy * P * 64) * w[dy * 3 + dx];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_fmed3f(d[e], 0.0f, 6.0f);
                if (!flush) {
                    dprev[t] = d;
                } else {
                    const BP<NP> ds = odd ? splitN<NP>(dprev[t], d) : splitN<NP>(d, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        BP<NP> wpf;
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) wpf.p[pl] = *reinterpret_cast<const bf16x8*>(Wps + (pl * NT + ni) * 512 + fslot);
                        acc[t][ni] = mmaN<NP>(wpf, ds, acc[t][ni]);
                    }
                }
            }
            if (i + 1 < nchunk) expand(i + 1, load_we(i + 1));
        }
    } else {
    BP<NP> wa = load_we(0);
    expand(0, wa);
    if (nchunk > 1) wa = load_we(1);

    for (int i = 0; i < nchunk; ++i) {
        b3_lds_barrier();           // E(i) is complete; everyone is done reading E(i - 1)
        const bool flush = (i & 1) || i + 1 == nchunk;      // the project runs after every second chunk (and after a last odd one)
        BP<NP> wp[NT];
        if (flush) load_wp(wp, i >> 1);                     // in flight across the depthwise
        const char* eb = Es + (i & 1) * EBUF;
        f32x4 w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(Ps + (1 + k) * Ce + i * kB3C + g4 * 4);
        const f32x4 dh = *reinterpret_cast<const f32x4*>(Ps + 10 * Ce + i * kB3C + g4 * 4);
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (t >= nto) break;
            f32x4 d = dh;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    d += *reinterpret_cast<const f32x4*>(eb + ea[t][dx] + dy * P * 64) * w[dy * 3 + dx];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_fmed3f(d[e], 0.0f, 6.0f);
            if (!flush) {
                dprev[t] = d;
            } else {
                const BP<NP> ds = (i & 1) ? splitN<NP>(dprev[t], d) : splitN<NP>(d, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[t][ni] = mmaN<NP>(wp[ni], ds, acc[t][ni]);
            }
        }
        if (i + 1 < nchunk) {
            expand(i + 1, wa);
            if (i + 2 < nchunk) wa = load_we(i + 2);        // in flight across the barrier and the next depthwise
        }
    }
    }

    const long img_o = (long)img * Ho * Wo;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (opix[t] < 0) continue;
        float* yp = p.y + (img_o + opix[t]) * p.Cout + g4 * 4;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if (ni * 16 + g4 * 4 >= p.Cout) continue;
            f32x4 v = acc[t][ni] + *reinterpret_cast<const f32x4*>(p.ph + ni * 16 + g4 * 4);
            if (p.residual) v = v + *reinterpret_cast<const f32x4*>(p.x + (img_o + opix[t]) * p.Cout + ni * 16 + g4 * 4);
            *reinterpret_cast<f32x4*>(yp + ni * 16) = v;
        }
    }
}

template <int CIN, int NT, int T, int TO, int S, int P, int NP, bool WDMA = false>
__global__ __launch_bounds__(kB3Threads) void mbv2_band3_block_kernel(const FusedBlockParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_b3[];
    band3_body<CIN, NT, T, TO, S, P, NP, WDMA>(p, smem_b3);
}

typedef void (*band3_kernel_t)(const FusedBlockParams);
struct Band3Cfg {
    int cin, nt, t, to, stride, pitch;
    band3_kernel_t fn;          // split-bf16 (fp32 results); nullptr: the shape only exists in the bf16 form
    band3_kernel_t fn1;         // bf16 (precision 1)
    band3_kernel_t fn_d, fn1_d; // the same with the weights staged by LDS-DMA (FusedBlockParams.form2)
};
#define B3CFG(CIN, NT, T, TO, S, P) {CIN, NT, T, TO, S, P, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 3>, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 1>, \
                                     mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 3, true>, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 1, true>}
#define B3DCFG(CIN, NT, T, TO, S, P) {CIN, NT, T, TO, S, P, nullptr, nullptr, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 3, true>, nullptr}
#define B1CFG(CIN, NT, T, TO, S, P) {CIN, NT, T, TO, S, P, nullptr, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 1>, nullptr, mbv2_band3_block_kernel<CIN, NT, T, TO, S, P, 1, true>}
const Band3Cfg kBand3[] = {
    B1CFG(16, 2, 9, 2, 2, 152),   // bf16 form only: block 1 (16 -> 96 -> 24, 150x150 -> 75x75) and block 2 (75x75), the
    B1CFG(24, 2, 8, 6, 1, 80),    // fp32 band kernel's shapes
    B1CFG(24, 2, 8, 6, 1, 136),   // block 2 of the 512x512 graph
    // blocks 1-2 (Cin 16 / 24, T = 9 / 6 tiles per wave) stay on the fp32 band kernel: the three split X planes cost
    // 12 registers per tile, the kernel spilled (9 / 46 registers) and measured 149-158 / 162-172 us against 143 / 105
    // round 6: with the weight fragments in LDS the split form of blocks 1-2 fits (second form only; block 1 with 8 tile slots per
    // wave = 2 output rows per band, so that E + the weight stages stay below 160 KB)
    B3DCFG(16, 2, 8, 2, 2, 152),
    B3DCFG(24, 2, 7, 5, 1, 80),
    B3DCFG(24, 2, 6, 4, 1, 80),
    B3CFG(24, 2, 7, 2, 2, 80),    // block 3
    B3CFG(32, 2, 4, 4, 1, 40),    // blocks 4-5
    B3CFG(32, 4, 4, 1, 2, 40),    // block 6
    B3CFG(24, 2, 7, 2, 2, 136),   // the 512x512 graph: block 3 (128 -> 64), blocks 4-5 (64x64), block 6 (64 -> 32)
    B3CFG(32, 2, 4, 4, 1, 72),
    B3CFG(32, 4, 4, 1, 2, 72),
};

int band3_max_rows(const Band3Cfg& c, const FusedBlockParams& p) {
    const int hb = c.t * 8 * 16 / c.pitch;
    int r = c.stride == 1 ? hb - 2 : (hb - 1) / 2;
    const int po = p.Wo + 1;
    while (r > 0 && (r * po + 15) / 16 > c.to * 8) --r;
    return r;
}

size_t band3_lds_bytes(const Band3Cfg& c, const FusedBlockParams& p);
const Band3Cfg* pick_band3(const FusedBlockParams& p, bool bf16) {
    static const int split12 = getenv("SSD_BAND3_SPLIT12") ? atoi(getenv("SSD_BAND3_SPLIT12")) : 0;   // blocks 1-2 on the split form (experiment; 2: block 2's smaller band)
    if (p.Ce % kB3C != 0 || p.Cout % 8 != 0 || p.Cin > 32) return nullptr;
    if (p.stride == 1 && (p.H != p.Ho || p.W != p.Wo || p.pad_t != 1 || p.pad_l != 1)) return nullptr;
    if (p.stride == 2 && (p.residual || p.Ho != (p.H + 1) / 2 || p.Wo != (p.W + 1) / 2 || p.pad_t > 1 || p.pad_l > 1 ||
                          p.pad_t < 0 || p.pad_l < 0))
        return nullptr;
    if (p.residual && p.Cin != p.Cout) return nullptr;
    if (p.e_out) return nullptr;
    for (const auto& c : kBand3) {
        if (c.cin != p.Cin || c.stride != p.stride || (p.Cout + 15) / 16 != c.nt || p.npad_p < c.nt * 16) continue;
        if (bf16 ? !c.fn1 : !(c.fn || (split12 && p.form2 && c.fn_d))) continue;
        if (!bf16 && !c.fn && split12 == 2 && c.cin == 24 && c.t == 7) continue;
        if (!bf16 && !c.fn && band3_lds_bytes(c, p) + (size_t)3 * (2 + c.nt) * 1024 > 160 * 1024) continue;
        if (p.W + 1 > c.pitch || p.W + 8 < c.pitch) continue;
        if (p.stride == 2 && 2 * (p.Wo - 1) - p.pad_l + 2 >= c.pitch) continue;
        if (band3_max_rows(c, p) < 1) continue;
        return &c;
    }
    return nullptr;
}

size_t band3_lds_bytes(const Band3Cfg& c, const FusedBlockParams& p) {
    return (size_t)2 * b3_ne(c.t) * kB3C * 4 + (size_t)11 * p.Ce * 4;
}

}  // namespace

bool band3_block_supported(const FusedBlockParams& p) {
    const Band3Cfg* c = pick_band3(p, p.bf16 != 0);
    return c && band3_lds_bytes(*c, p) <= 160 * 1024;
}

// bf16 planes of the two 1x1 weight matrices (shorts): sizes and the packing launches (run at finalize)
size_t band3_we_shorts(int Ce) { return (size_t)4 * Ce * 32; }
size_t band3_wp_shorts(int npad_p, int Ce) { return (size_t)4 * npad_p * (((Ce / kB3C) + 1) / 2) * 32; }
int launch_band3_pack(const float* we, int Ce, int Cin, int kpad_e, short* we3, const float* wp, int npad_p, int kpad_p,
                      short* wp3, hipStream_t st) {
    const int npairs = ((Ce / kB3C) + 1) / 2;
    hipLaunchKernelGGL(split3_we_kernel, dim3((Ce * 32 + 255) / 256), dim3(256), 0, st, we, Ce, Cin, kpad_e, we3);
    SSD_LAUNCH_CHECK();
    hipLaunchKernelGGL(split3_wp_kernel, dim3((npad_p * npairs * 32 + 255) / 256), dim3(256), 0, st, wp, npad_p, Ce, kpad_p, npairs, wp3);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_band3_block(FusedBlockParams p, hipStream_t st) {
    const Band3Cfg* c = pick_band3(p, p.bf16 != 0);
    if (!c || !p.we3 || !p.wp3) {
        set_error("band3 block: unsupported shape Cin=%d Ce=%d Cout=%d %dx%d stride=%d (or weights not split)", p.Cin, p.Ce, p.Cout,
                  p.H, p.W, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    const int rmax = band3_max_rows(*c, p);
    p.bands = (p.Ho + rmax - 1) / rmax;
    size_t lds = band3_lds_bytes(*c, p);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "band3 block: needs %zu B of LDS", lds);
    band3_kernel_t fn = p.bf16 ? c->fn1 : c->fn;
    // second form: weight fragments through LDS (two We stages + one Wp stage of 1 KB blocks) where they fit
    const size_t lds_d = lds + (size_t)(p.bf16 ? 1 : 3) * (2 + c->nt) * 1024;
    if (p.form2 && lds_d <= 160 * 1024 && (p.bf16 ? c->fn1_d : c->fn_d)) {
        fn = p.bf16 ? c->fn1_d : c->fn_d;
        lds = lds_d;
    }
    if (lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)((long)p.B * p.bands)), dim3(kB3Threads), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

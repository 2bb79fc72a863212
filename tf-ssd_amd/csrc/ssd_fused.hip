// Fused MobileNetV2 inverted-residual block for gfx950:
//
//     y = project_BN( relu6(dw_BN( dw3x3( relu6(expand_BN( x * We )) ) )) * Wp ) [+ x]
//
// The reference graph ([3P] keras-applications MobileNetV2 blocks) runs
// this as 3 convs + 3 BatchNorms + 2 ReLU6 (+ add): the 6x-expanded tensor is written and read
// twice (44 MB/img of the 102 MB/img activation traffic).  Here one workgroup owns a TH x TW
// tile of output pixels of one image and walks the expanded channels in chunks of 48 (48
// divides every 6*Cin):
//
//   A  expand   E[halo px][48] = relu6(BN(X[halo px][Cin] * We[Cin][48]))   fp32 MFMA, X tile in LDS;
//               halo pixels outside the image are forced to 0 (the depthwise pads E, not X)
//   B  depthw.  D[out px][48]  = relu6(BN(sum_taps E[...] * Wd))            VALU from LDS
//   C  project  acc[out px][Cout] += D[out px][48] * Wp[48][Cout]           fp32 MFMA, acc in registers
//
// and finally y = acc * scale + shift (+ residual taken from the X tile).  The expanded and the
// depthwise tensors never leave the CU; HBM traffic is x in + y out (+ weights from L2).
// Halo pixels are expanded redundantly (100/64 for an 8x8 stride-1 tile).  Weight chunks are
// prefetched global->registers one chunk ahead.
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

constexpr int kCK = 48;          // expanded channels per chunk
constexpr int kLDE = kCK + 4;    // LDS row stride of E / D / Wp chunk tiles

// Expand NP (1 or 2) halo-pixel tiles of 16 pixels against the 48-channel weight chunk:
// 3*NP independent accumulator chains keep the fp32 MFMA pipe issuing back to back.
template <int NP, int CINP, int IW, int IPX>
__device__ __forceinline__ void expand_px_tiles(const float* Xs, const float* Wes, float* Es, const float* Ps,
                                                const int Ce, const int ce0, const int pt0, const int pt1,
                                                const int lane, const int iy0, const int ix0, const int H,
                                                const int W) {
    constexpr int LDX = CINP + 4;
    const int frow = lane & 15, fk = (lane >> 4) * 4;
    const int pts[2] = {pt0, pt1};
    f32x4 ea[NP][3];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) ea[q][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < CINP / 16; ++kc) {
        f32x4 xb4[NP], wa[3];
#pragma unroll
        for (int q = 0; q < NP; ++q)
            xb4[q] = *reinterpret_cast<const f32x4*>(Xs + (pts[q] * 16 + frow) * LDX + kc * 16 + fk);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
            wa[ct] = *reinterpret_cast<const f32x4*>(Wes + (ct * 16 + frow) * LDX + kc * 16 + fk);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    ea[q][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ct][s], xb4[q][s], ea[q][ct], 0, 0, 0);
    }
    // lane holds E[px = pt*16 + (lane & 15)][ce = ct*16 + (lane >> 4)*4 + 0..3]
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int hp = pts[q] * 16 + (lane & 15);
        const int r = hp / IW, c = hp - r * IW;
        const bool inimg = hp < IPX && (unsigned)(iy0 + r) < (unsigned)H && (unsigned)(ix0 + c) < (unsigned)W;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int cl = ct * 16 + (lane >> 4) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inimg) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(Ps + ce0 + cl);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + Ce + ce0 + cl);
                v = ea[q][ct] * sc + sh;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu6f(v[j]);
            }
            *reinterpret_cast<f32x4*>(Es + hp * kLDE + cl) = v;
        }
    }
}

// Persistent workgroups: each loops over output tiles (tile = blockIdx.x, += gridDim.x) and
// prefetches the NEXT tile's input halo and first weight chunk into registers while the last
// channel chunk of the current tile is computed, so global latency is exposed once per
// workgroup instead of once per tile.
template <int CINP, int NTC, int S, int TH, int TW, bool RES>
__global__ __launch_bounds__(256, 2) void mbv2_block_kernel(const FusedBlockParams p) {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int IPX = IH * IW, IPXP = (IPX + 15) / 16 * 16, NPT = IPXP / 16;
    constexpr int OPX = TH * TW;                    // 64 or 32
    constexpr int WPX = OPX / 16;                   // waves along pixels in phase C
    constexpr int WN = 4 / WPX;                     // waves along output channels
    constexpr int NTW = NTC / WN;                   // n-tiles per wave
    constexpr int LDX = CINP + 4;
    static_assert(OPX % 16 == 0 && 4 % WPX == 0 && NTC % WN == 0, "tile split");
    constexpr int WE_U = kCK * CINP / 4, WP_U = NTC * 16 * kCK / 4;       // float4 units per chunk
    constexpr int WE_R = (WE_U + 255) / 256, WP_R = (WP_U + 255) / 256;
    constexpr int X_U = IPXP * (CINP / 4), X_R = (X_U + 255) / 256;
    // phase B: one thread = 4 channels x SL consecutive output columns (sliding window)
    constexpr int SL = S == 1 ? 4 : 2;
    constexpr int NSTRIP = TH * (TW / SL);
    constexpr int NIN = (SL - 1) * S + 3;
    static_assert(NSTRIP * (kCK / 4) <= 256 && TW % SL == 0, "phase B mapping");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [IPXP][LDX]
    float* Es = Xs + IPXP * LDX;            // [IPXP][kLDE]
    float* Ds = Es + IPXP * kLDE;           // [OPX][kLDE]
    float* Wes = Ds + OPX * kLDE;           // [kCK][LDX]
    float* Wps = Wes + kCK * LDX;           // [NTC*16][kLDE]
    float* Ps = Wps + NTC * 16 * kLDE;      // [13][Ce]: es, eh, wd[9], ds, dh

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = p.dbg ? clock64() : 0;
#define TICK(i) do { if (p.dbg) { const long long t1 = clock64(); tacc[i] += t1 - t0; t0 = t1; } } while (0)
    const int Ce = p.Ce;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const long total_tiles = (long)p.B * tiles_per_img;

    // ---- weight chunk prefetch (global -> registers -> LDS)
    f32x4 wer[WE_R], wpr[WP_R], xr[X_R];
    auto load_w = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = tid + i * 256;
            const int row = u / (CINP / 4), k4 = (u - row * (CINP / 4)) * 4;
            wer[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < WE_U && k4 < p.kpad_e)
                wer[i] = *reinterpret_cast<const f32x4*>(p.we + (long)(chunk * kCK + row) * p.kpad_e + k4);
        }
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = tid + i * 256;
            const int row = u / (kCK / 4), k4 = (u - row * (kCK / 4)) * 4;
            wpr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < WP_U && row < p.npad_p)
                wpr[i] = *reinterpret_cast<const f32x4*>(p.wp + (long)row * p.kpad_p + chunk * kCK + k4);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = tid + i * 256;
            const int row = u / (CINP / 4), k4 = (u - row * (CINP / 4)) * 4;
            if (u < WE_U) *reinterpret_cast<f32x4*>(Wes + row * LDX + k4) = wer[i];
        }
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = tid + i * 256;
            const int row = u / (kCK / 4), k4 = (u - row * (kCK / 4)) * 4;
            if (u < WP_U) *reinterpret_cast<f32x4*>(Wps + row * kLDE + k4) = wpr[i];
        }
    };
    // X halo tile of tile index t -> registers (zeros outside the image / beyond Cin)
    auto load_x = [&](long t) {
        const int b = (int)(t / tiles_per_img);
        const int rem = (int)(t - (long)b * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int iy0 = ty * TH * S - p.pad_t, ix0 = tx * TW * S - p.pad_l;
        const float* xb = p.x + (long)b * p.H * p.W * p.Cin;
#pragma unroll
        for (int i = 0; i < X_R; ++i) {
            const int u = tid + i * 256;
            const int hp = u / (CINP / 4), k4 = (u - hp * (CINP / 4)) * 4;
            const int r = hp / IW, c = hp - r * IW;
            const int iy = iy0 + r, ix = ix0 + c;
            xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < X_U && hp < IPX && k4 < p.Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                xr[i] = *reinterpret_cast<const f32x4*>(xb + ((long)iy * p.W + ix) * p.Cin + k4);
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < X_R; ++i) {
            const int u = tid + i * 256;
            const int hp = u / (CINP / 4), k4 = (u - hp * (CINP / 4)) * 4;
            if (u < X_U) *reinterpret_cast<f32x4*>(Xs + hp * LDX + k4) = xr[i];
        }
    };

    long tile = blockIdx.x;
    if (tile >= total_tiles) return;
    load_w(0);
    load_x(tile);
    // per-channel parameters of the whole block -> LDS, once per workgroup
    for (int u = tid; u < 13 * (Ce / 4); u += 256) {
        const int row = u / (Ce / 4), c4 = (u - row * (Ce / 4)) * 4;
        const float* src = row == 0 ? p.es : row == 1 ? p.eh : row == 11 ? p.ds : row == 12 ? p.dh
                                                                   : p.wd + (long)(row - 2) * Ce;
        *reinterpret_cast<f32x4*>(Ps + row * Ce + c4) = *reinterpret_cast<const f32x4*>(src + c4);
    }
    store_x();
    store_w();
    __syncthreads();
    TICK(0);

    const int frow = lane & 15, fk = (lane >> 4) * 4;
    const int wpx = wave % WPX, wn = wave / WPX;     // phase C wave coordinates
    // phase B coordinates of this thread
    const int bc4 = (tid % (kCK / 4)) * 4;
    const int bstrip = tid / (kCK / 4);
    const int boy = bstrip / (TW / SL), box0 = (bstrip - boy * (TW / SL)) * SL;
    const int nchunk = Ce / kCK;

    for (; tile < total_tiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_img);
        const int rem = (int)(tile - (long)b * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
        const long next = tile + gridDim.x;
        f32x4 acc[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < nchunk; ++ch) {
            const int ce0 = ch * kCK;
            if (ch + 1 < nchunk) {
                load_w(ch + 1);
            } else if (next < total_tiles) {       // last chunk: prefetch the next tile
                load_w(0);
                load_x(next);
            }

            // ---- phase A: expand (MFMA): wave handles halo pixel tiles wave, wave+4, ...
            {
                int pt = wave;
                for (; pt + 4 < NPT; pt += 8)
                    expand_px_tiles<2, CINP, IW, IPX>(Xs, Wes, Es, Ps, Ce, ce0, pt, pt + 4, lane, iy0, ix0, p.H, p.W);
                if (pt < NPT)
                    expand_px_tiles<1, CINP, IW, IPX>(Xs, Wes, Es, Ps, Ce, ce0, pt, pt, lane, iy0, ix0, p.H, p.W);
            }
            __syncthreads();
            TICK(1);

            // ---- phase B: depthwise 3x3 + BN + ReLU6 (VALU, LDS -> LDS), sliding register window
            if (tid < NSTRIP * (kCK / 4)) {
                f32x4 a[SL];
#pragma unroll
                for (int t = 0; t < SL; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    f32x4 e[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j)
                        e[j] = *reinterpret_cast<const f32x4*>(Es + ((boy * S + ky) * IW + box0 * S + j) * kLDE + bc4);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(Ps + (2 + ky * 3 + kx) * Ce + ce0 + bc4);
#pragma unroll
                        for (int t = 0; t < SL; ++t) {
                            const f32x4 x = e[t * S + kx];
                            a[t][0] = fmaf(x[0], w[0], a[t][0]);
                            a[t][1] = fmaf(x[1], w[1], a[t][1]);
                            a[t][2] = fmaf(x[2], w[2], a[t][2]);
                            a[t][3] = fmaf(x[3], w[3], a[t][3]);
                        }
                    }
                }
                const f32x4 sc = *reinterpret_cast<const f32x4*>(Ps + 11 * Ce + ce0 + bc4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + 12 * Ce + ce0 + bc4);
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    f32x4 v = a[t] * sc + sh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = relu6f(v[j]);
                    *reinterpret_cast<f32x4*>(Ds + (boy * TW + box0 + t) * kLDE + bc4) = v;
                }
            }
            __syncthreads();
            TICK(2);

            // ---- phase C: project (MFMA), accumulators stay in registers across chunks
#pragma unroll
            for (int kc = 0; kc < kCK / 16; ++kc) {
                const f32x4 db = *reinterpret_cast<const f32x4*>(Ds + (wpx * 16 + frow) * kLDE + kc * 16 + fk);
                f32x4 wa[NTW];
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni)
                    wa[ni] = *reinterpret_cast<const f32x4*>(Wps + ((wn * NTW + ni) * 16 + frow) * kLDE + kc * 16 + fk);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni)
                        acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ni][s], db[s], acc[ni], 0, 0, 0);
            }
            __syncthreads();
            TICK(3);
            if (ch + 1 < nchunk) {
                store_w();
                __syncthreads();
            }
            TICK(4);
        }

        // ---- epilogue: project BN (+ residual from the X tile), 16-byte stores
        const int po = wpx * 16 + (lane & 15);
        const int oy = po / TW, ox = po - oy * TW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy < p.Ho && gx < p.Wo) {
            float* yrow = p.y + (((long)b * p.Ho + gy) * p.Wo + gx) * p.Cout;
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) {
                const int n = (wn * NTW + ni) * 16 + (lane >> 4) * 4;
                if (n >= p.Cout) continue;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(p.ps + n);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(p.ph + n);
                f32x4 v = acc[ni] * sc + sh;
                if (RES) {
                    // stride 1: the output pixel's input is halo pixel (oy + 1, ox + 1)
                    const f32x4 xres = *reinterpret_cast<const f32x4*>(Xs + ((oy + 1) * IW + ox + 1) * LDX + n);
                    v = v + xres;
                }
                *reinterpret_cast<f32x4*>(yrow + n) = v;
            }
        }
        if (next < total_tiles) {
            __syncthreads();          // every wave is done with Xs (residual) before it is replaced
            store_x();
            store_w();
            __syncthreads();
        }
        TICK(5);
    }
    if (p.dbg && (tid & 63) == 0)
        for (int i = 0; i < 6; ++i) p.dbg[((long)blockIdx.x * 4 + wave) * 6 + i] = tacc[i];
}

template <int CINP, int NTC, int S, int TH, int TW>
constexpr size_t fused_static_floats() {
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int IPXP = (IH * IW + 15) / 16 * 16;
    return (size_t)IPXP * (CINP + 4) + (size_t)IPXP * kLDE + (size_t)TH * TW * kLDE + (size_t)kCK * (CINP + 4) +
           (size_t)NTC * 16 * kLDE;
}

typedef void (*fused_kernel_t)(const FusedBlockParams);

struct FusedCfg {
    int cinp, ntc, stride, th, tw, res;
    size_t static_floats;
    fused_kernel_t fn;
};
#define FCFG(CINP, NTC, S, TH, TW, RES)                                             \
    {CINP, NTC, S, TH, TW, RES, fused_static_floats<CINP, NTC, S, TH, TW>(),        \
     mbv2_block_kernel<CINP, NTC, S, TH, TW, RES != 0>}
static const FusedCfg kFused[] = {
    FCFG(16, 2, 2, 4, 8, 0),   // block_1: 16 -> 96 -> 24, stride 2
    FCFG(32, 2, 1, 8, 8, 1),   // block_2 / 4 / 5: residual, Cout <= 32
    FCFG(32, 2, 1, 8, 8, 0),
    FCFG(32, 2, 2, 4, 8, 0),   // block_3: 24 -> 144 -> 32, stride 2
    FCFG(32, 4, 2, 4, 8, 0),   // block_6: 32 -> 192 -> 64, stride 2
    FCFG(32, 4, 1, 8, 8, 0),
    FCFG(32, 4, 1, 8, 8, 1),
};

static const FusedCfg* pick_fused(const FusedBlockParams& p) {
    if (p.Ce % kCK != 0 || p.Cin % 4 != 0 || p.Cout % 4 != 0) return nullptr;
    const int cinp = p.Cin <= 16 ? 16 : (p.Cin <= 32 ? 32 : 0);
    const int ntc = p.Cout <= 32 ? 2 : (p.Cout <= 64 ? 4 : 0);
    if (!cinp || !ntc) return nullptr;
    for (const auto& c : kFused)
        if (c.cinp == cinp && c.ntc == ntc && c.stride == p.stride && c.res == (p.residual ? 1 : 0)) return &c;
    return nullptr;
}

bool fused_block_supported(const FusedBlockParams& p) { return pick_fused(p) != nullptr; }

int launch_fused_block(FusedBlockParams p, hipStream_t st) {
    const FusedCfg* c = pick_fused(p);
    if (!c) {
        set_error("fused block: unsupported shape Cin=%d Ce=%d Cout=%d stride=%d", p.Cin, p.Ce, p.Cout, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    p.tiles_y = (p.Ho + c->th - 1) / c->th;
    p.tiles_x = (p.Wo + c->tw - 1) / c->tw;
    const long tiles = (long)p.B * p.tiles_y * p.tiles_x;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0)
            num_cu = 256;
    }
    const long blocks = tiles < 2L * num_cu ? tiles : 2L * num_cu;      // persistent: 2 workgroups per CU
    const size_t lds = (c->static_floats + (size_t)13 * p.Ce) * sizeof(float);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "fused block: needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)c->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(c->fn, dim3((unsigned)blocks), dim3(256), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

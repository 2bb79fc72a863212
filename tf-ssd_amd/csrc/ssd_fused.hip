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
#include <cstdlib>

#include "ssd_conv.h"
#include "ssd_bf16x3.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup fence
// over ALL address spaces, and since vmcnt also counts stores on gfx950 the compiler then
// drains every outstanding global load (s_waitcnt vmcnt(0)) at the first LDS access after
// the barrier -- which would stall on the weight / next-tile prefetch that is deliberately
// kept in flight across the phases of this kernel.
__device__ __forceinline__ void lds_barrier() {
    // (an address-space-restricted __builtin_amdgcn_fence(..., "local") still drained vmcnt on
    // ROCm 7.2, hence the explicit LDS-counter wait + raw barrier; "memory" keeps the compiler
    // from moving LDS accesses across it)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Prefetch loads the compiler must not wait for: a 16-byte global load issued through inline
// asm is invisible to hipcc's s_waitcnt bookkeeping, so it stays in flight across the phase
// barriers; wait_prefetch() is the matching hand-placed wait (every destination is passed
// through an empty "+v" statement so no consumer can be scheduled above the wait).
__device__ __forceinline__ f32x4 gload16_async(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_prefetch(f32x4 (&r)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}

constexpr int kCK = 48;          // expanded channels per chunk
// LDS row strides.  A b128 access of 16 rows x 4 column groups (MFMA fragments, accumulator-layout
// tile writes) is conflict-free under gfx950's lane grouping when the stride is 2 (mod 4) quads:
// 56 floats for the E / D / Wp chunk tiles (52 was 2-way conflicted), 24 for Cin 16 / 24 X tiles.
// Cin = 32 keeps 36 (9 quads, conflicted): 40 would push blocks 4/5 past 80 KB, i.e. to one
// workgroup per CU.
constexpr int kLDE = kCK + 8;
constexpr int ldx_for(int cinp) { return cinp <= 24 ? 24 : cinp + 4; }

// Expand NP (1 or 2) halo-pixel tiles of 16 pixels against the 48-channel weight chunk:
// 3*NP independent accumulator chains keep the fp32 MFMA pipe issuing back to back.
template <int NP, int CINP, int IW, int IPX>
__device__ __forceinline__ void expand_px_tiles(const float* Xs, const float* Wes, float* Es, const float* Ps,
                                                const int Ce, const int ce0, const int pt0, const int pt1,
                                                const int lane, const int iy0, const int ix0, const int H,
                                                const int W, const int ablate = 0) {
    constexpr int LDX = ldx_for(CINP);
    const int frow = lane & 15, fk = (lane >> 4) * 4;
    const int pts[2] = {pt0, pt1};
    // accumulators start at the folded BatchNorm shift (the scale is folded into the weights)
    f32x4 ea[NP][3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + Ce + ce0 + ct * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int q = 0; q < NP; ++q) ea[q][ct] = sh;
    }
    if (!(ablate & 1))
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
    if (CINP % 16 == 8 && !(ablate & 1)) {
        // K tail of 8 (Cin = 24): lane group g reads k = 16*(CINP/16) + 2g + {0, 1} with one
        // ds_read_b64, i.e. 2 MFMA k-steps instead of the 4 a zero-padded 16-wide unit would cost
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int K0 = CINP / 16 * 16;
        const int fk2 = (lane >> 4) * 2;
        f32x2 xb2[NP], wa2[3];
#pragma unroll
        for (int q = 0; q < NP; ++q)
            xb2[q] = *reinterpret_cast<const f32x2*>(Xs + (pts[q] * 16 + frow) * LDX + K0 + fk2);
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
            wa2[ct] = *reinterpret_cast<const f32x2*>(Wes + (ct * 16 + frow) * LDX + K0 + fk2);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    ea[q][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa2[ct][s], xb2[q][s], ea[q][ct], 0, 0, 0);
    }
    // lane holds E[px = pt*16 + (lane & 15)][ce = ct*16 + (lane >> 4)*4 + 0..3]
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int hp = pts[q] * 16 + (lane & 15);
        const int r = hp / IW, c = hp - r * IW;
        const bool inimg = hp < IPX && (unsigned)(iy0 + r) < (unsigned)H && (unsigned)(ix0 + c) < (unsigned)W;
        const float hi = inimg ? 6.0f : 0.0f;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int cl = ct * 16 + (lane >> 4) * 4;
            f32x4 v = ea[q][ct];
            // relu6 inside the image, 0 outside it (the depthwise pads E): one v_med3 with a
            // per-lane upper bound of 6 or 0 instead of med3 + select
            if (!(ablate & 8))
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], 0.0f, hi);
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
    constexpr int LDX = ldx_for(CINP);
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
    float* Ps = Wps + NTC * 16 * kLDE;      // [13][Ce]: es, eh, wd[9], ds, dh; then [2][NTC*16]: ps, ph

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = p.dbg ? clock64() : 0;
#define TICK(i) do { if (p.dbg) { const long long t1 = clock64(); tacc[i] += t1 - t0; t0 = t1; } } while (0)
    const bool getenv_dbg_split = p.dbg != nullptr;
    const int Ce = p.Ce;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const long total_tiles = (long)p.B * tiles_per_img;

    // ---- weight chunk prefetch (global -> registers -> LDS)
    f32x4 wer[WE_R], wpr[WP_R], xr[X_R];
    // (indices are clamped instead of predicated: every lane always loads from a valid
    //  address, the stores below drop the lanes that are out of range)
    auto load_w = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = min(tid + i * 256, WE_U - 1);
            const int row = u / (CINP / 4), k4 = (u - row * (CINP / 4)) * 4;
            wer[i] = gload16_async(p.we + (long)(chunk * kCK + row) * p.kpad_e + k4);
        }
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = min(tid + i * 256, WP_U - 1);
            const int row = min(u / (kCK / 4), p.npad_p - 1), k4 = (u % (kCK / 4)) * 4;
            wpr[i] = gload16_async(p.wp + (long)row * p.kpad_p + chunk * kCK + k4);
        }
    };
    auto store_w = [&]() {
        wait_prefetch(wer);
        wait_prefetch(wpr);
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
            const int iy = min(max(iy0 + r, 0), p.H - 1), ix = min(max(ix0 + c, 0), p.W - 1);
            xr[i] = gload16_async(xb + ((long)iy * p.W + ix) * p.Cin + min(k4, p.Cin - 4));
        }
    };
    // the zero padding (outside the image / beyond Cin / beyond the halo) is applied here
    auto store_x = [&](long t) {
        const int b = (int)(t / tiles_per_img);
        const int rem = (int)(t - (long)b * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int iy0 = ty * TH * S - p.pad_t, ix0 = tx * TW * S - p.pad_l;
        wait_prefetch(xr);
#pragma unroll
        for (int i = 0; i < X_R; ++i) {
            const int u = tid + i * 256;
            const int hp = u / (CINP / 4), k4 = (u - hp * (CINP / 4)) * 4;
            const int r = hp / IW, c = hp - r * IW;
            const bool ok = hp < IPX && k4 < p.Cin && (unsigned)(iy0 + r) < (unsigned)p.H &&
                            (unsigned)(ix0 + c) < (unsigned)p.W;
            if (u < X_U) *reinterpret_cast<f32x4*>(Xs + hp * LDX + k4) = ok ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
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
    for (int u = tid; u < 2 * NTC * 16; u += 256) {
        const int row = u / (NTC * 16), n = u - row * (NTC * 16);
        Ps[13 * Ce + u] = n < p.Cout ? (row == 0 ? p.ps[n] : p.ph[n]) : 0.f;
    }
    store_x(tile);
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
        for (int i = 0; i < NTW; ++i)
            acc[i] = *reinterpret_cast<const f32x4*>(Ps + 13 * Ce + NTC * 16 + (wn * NTW + i) * 16 + (lane >> 4) * 4);

        for (int ch = 0; ch < nchunk; ++ch) {
            const int ce0 = ch * kCK;
            // Prefetch (asm loads, see gload16_async): the next chunk's weights -- chunk 0 of the
            // next tile during the last chunk -- and on the last chunk the next tile's input halo.
            // A prefetch is issued ONLY if it will be consumed: the compiler treats the
            // destination VGPRs of an unconsumed asm load as dead and would reuse them while
            // the load is still in flight.
            {
                const bool last = ch + 1 == nchunk, has_next = next < total_tiles;
                if (!last || has_next) load_w(last ? 0 : ch + 1);
                if (last && has_next) load_x(next);
            }

            // ---- phase A: expand (MFMA): wave handles halo pixel tiles wave, wave+4, ...
            {
                int pt = wave;
                for (; pt + 4 < NPT; pt += 8)
                    expand_px_tiles<2, CINP, IW, IPX>(Xs, Wes, Es, Ps, Ce, ce0, pt, pt + 4, lane, iy0, ix0, p.H, p.W, p.ablate);
                if (pt < NPT)
                    expand_px_tiles<1, CINP, IW, IPX>(Xs, Wes, Es, Ps, Ce, ce0, pt, pt, lane, iy0, ix0, p.H, p.W, p.ablate);
            }
            if (getenv_dbg_split) TICK(4);      // diagnostics: slot 4 = expand compute (+ weight staging), slot 1 = its barrier wait
            if (!(p.ablate & 16)) lds_barrier();
            TICK(1);

            // ---- phase B: depthwise 3x3 + BN + ReLU6 (VALU, LDS -> LDS), sliding register window
            if (tid < NSTRIP * (kCK / 4) && !(p.ablate & 2)) {
                f32x4 a[SL];
                {
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + 12 * Ce + ce0 + bc4);
#pragma unroll
                    for (int t = 0; t < SL; ++t) a[t] = sh;
                }
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
                            a[t] += e[t * S + kx] * w;        // vector form: contracts to v_pk_fma_f32
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    f32x4 v = a[t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = relu6f(v[j]);
                    *reinterpret_cast<f32x4*>(Ds + (boy * TW + box0 + t) * kLDE + bc4) = v;
                }
            }
            if (!(p.ablate & 16)) lds_barrier();
            TICK(2);

            // ---- phase C: project (MFMA), accumulators stay in registers across chunks
            if (!(p.ablate & 4))
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
            if (!(p.ablate & 16)) lds_barrier();
            TICK(3);
            if (ch + 1 < nchunk) {
                store_w();
                if (!(p.ablate & 16)) lds_barrier();
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
                f32x4 v = acc[ni];
                if (RES) {
                    // stride 1: the output pixel's input is halo pixel (oy + 1, ox + 1)
                    const f32x4 xres = *reinterpret_cast<const f32x4*>(Xs + ((oy + 1) * IW + ox + 1) * LDX + n);
                    v = v + xres;
                }
                *reinterpret_cast<f32x4*>(yrow + n) = v;
            }
        }
        if (next < total_tiles) {
            if (!(p.ablate & 16)) lds_barrier();          // every wave is done with Xs (residual) before it is replaced
            store_x(next);
            store_w();
            if (!(p.ablate & 16)) lds_barrier();
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
    return (size_t)IPXP * ldx_for(CINP) + (size_t)IPXP * kLDE + (size_t)TH * TW * kLDE + (size_t)kCK * ldx_for(CINP) +
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
    FCFG(24, 2, 1, 8, 8, 1),   // block_2: 24 -> 144 -> 24 (K = 24: 6 MFMA k-steps, not 8)
    FCFG(24, 2, 2, 4, 8, 0),   // block_3: 24 -> 144 -> 32, stride 2
    FCFG(32, 2, 1, 8, 8, 1),   // block_4 / 5: residual, Cout <= 32
    FCFG(32, 2, 1, 8, 8, 0),
    FCFG(32, 2, 2, 4, 8, 0),   // block_3: 24 -> 144 -> 32, stride 2
    FCFG(32, 4, 2, 4, 8, 0),   // block_6: 32 -> 192 -> 64, stride 2
    FCFG(32, 4, 1, 8, 8, 0),
    FCFG(32, 4, 1, 8, 8, 1),
};

static const FusedCfg* pick_fused(const FusedBlockParams& p) {
    if (p.Ce % kCK != 0 || p.Cin % 4 != 0 || p.Cout % 4 != 0 || p.e_out) return nullptr;   // (the tile kernel never writes E out)
    const int cinp = p.Cin <= 16 ? 16 : (p.Cin == 24 ? 24 : (p.Cin <= 32 ? 32 : 0));
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
    static int per_cu = 0;
    if (!per_cu) {
        const char* e = getenv("SSD_FUSED_BLOCKS_PER_CU");      // diagnostics knob
        per_cu = e ? atoi(e) : 2;
        if (per_cu < 1) per_cu = 2;
    }
    const long blocks = tiles < (long)per_cu * num_cu ? tiles : (long)per_cu * num_cu;   // persistent workgroups
    const size_t lds = (c->static_floats + (size_t)13 * p.Ce + (size_t)2 * c->ntc * 16) * sizeof(float);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "fused block: needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)c->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(c->fn, dim3((unsigned)blocks), dim3(256), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

// ------------------------------------------------------------------------------------------
// Fused MobileNetV2 stem: Conv1 (3x3 stride 2, 3 -> 32, BN, ReLU6) -> expanded_conv_depthwise
// (3x3, BN, ReLU6) -> expanded_conv_project (1x1, 32 -> 16, BN).  Unfused these three layers
// move 830 MB per 64-image batch (two 184 MB 32-channel maps written and re-read); fused, the
// 300x300x3 image goes in and the 150x150x16 map comes out.  One workgroup = 8 x 16 output
// pixels: image patch -> LDS, Conv1 on the 10 x 18 halo (VALU, weights broadcast from LDS,
// zero outside the feature map because the depthwise pads Conv1's OUTPUT), depthwise from LDS
// (sliding window), project on the fp32 MFMA (K = 32), 16-byte stores.
// NP selects the matrix instruction of Conv1 and the project: 0 = v_mfma_f32_16x16x4_f32 (exact fp32 products), 3 = the
// exact three-way bf16 split on v_mfma_f32_16x16x32_bf16 (fp32 results: K = 27 / 32 is ONE k-step, six instructions of
// 16 cycles instead of eight of 32 per 16 x 16 tile), 1 = operands rounded once to bf16 (the net's bf16 mode).
constexpr int kSTH = 8, kSTW = 16;
constexpr int kSIH = kSTH + 2, kSIW = kSTW + 2;           // Conv1 halo tile 10 x 18
constexpr int kSPH = 2 * (kSIH - 1) + 3, kSPW = 2 * (kSIW - 1) + 3;   // image patch 21 x 37
constexpr int kSLD = 40;                                  // LDS row stride of the 32-channel tiles (10 quads: conflict-free b128 fragment reads)
// DMA form of the patch (round 5): rows start at a 16-byte aligned float (up to 3 floats left of the patch's first), 29 quads =
// 116 floats per row, TWO stages: tile t + 1's rows are copied global -> LDS by `buffer_load_dwordx4 ... lds` while tile t is
// computed (the register-staged patch load cost 20 of the kernel's 112 us: tests/micro/stem_ablate.py, SSD_STEM_ABLATE = 8 / 16)
constexpr int kSPQ = (kSPW * 3 + 3 + 3) / 4, kSPR = kSPQ * 4;
constexpr int kSPS = (kSPH * kSPQ + 63) / 64 * 256;       // floats per patch stage: whole 64-lane instructions (640 units)

template <bool DMA>
constexpr int stem_patch_floats() { return DMA ? 2 * kSPS : kSPH * kSPW * 3 + 5; }
template <bool DMA>
constexpr int stem_lds_floats() {
    return stem_patch_floats<DMA>() + kSIH * kSIW * kSLD + kSTH * kSTW * kSLD + 27 * 32 + 9 * 32 + 16 * kSLD + 32 * 2 + 32 * 2 + 16 * 2;
}
// (a __device__ body: a __global__ function that declares buffer resources loses its host stub)
template <int NP, bool DMA>
__device__ __forceinline__ void stem_body(const StemParams& p, float* __restrict__ sm) {
    constexpr int RP = DMA ? kSPR : kSPW * 3;                  // floats between two patch rows in LDS
    float* patch = sm;                                         // [21*37*3] image patch (+5 pad); DMA: [2][640 units of 4]
    float* C1 = patch + stem_patch_floats<DMA>();              // [180][36] Conv1 output (halo), 16-byte aligned
    float* D = C1 + kSIH * kSIW * kSLD;                        // [128][36] depthwise output
    float* W1 = D + kSTH * kSTW * kSLD;                        // [27][32] Conv1 weights * BN scale
    float* Wd = W1 + 27 * 32;                                  // [9][32]  depthwise weights * BN scale
    float* Wp = Wd + 9 * 32;                                   // [16][36] project weights * BN scale
    float* H1 = Wp + 16 * kSLD;                                // [32] Conv1 BN shift
    float* Hd = H1 + 32;                                       // [32] depthwise BN shift
    float* Hp = Hd + 32;                                       // [16] project BN shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // weights (BatchNorm scale folded in) are staged once per persistent workgroup
    // Conv1 on the MFMA: K = 27 taps x channels padded to 32 = 2 k-blocks of 16; lane (ch = l15, g4) holds
    // the A fragments W[ch][k = kb*16 + g4*4 + s] * BN scale of both 16-channel tiles for the whole
    // persistent loop (16 VGPRs; the VALU form held 27 x 4 weights per thread and ran at 36 % of the
    // packed-FMA rate: 63 of the kernel's 131 us), and the patch offsets of ITS four k per k-block
    f32x4 w1a[2][2];
    int koff[2][4];
    int kcol[2][4];            // kx * 3 + ci of the lane's k values (DMA form: the right-edge mask)
    // NP > 0: one 32-wide k-step; lane (ch = l15, g4) holds k = g4*8 .. +7 of both channel tiles as bf16 planes, and the
    // project's A fragment (row n = l15) the same way
    BP<NP ? NP : 1> w1b[2], wpb;
    if (NP == 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = kb * 16 + (lane >> 4) * 4 + s4;
                koff[kb][s4] = k < 27 ? (k / 9) * RP + (k % 9) : 0;      // (ky, kx*3 + ci) inside the patch
                kcol[kb][s4] = k < 27 ? k % 9 : 0;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int ch = ct * 16 + (lane & 15);
                    w1a[ct][kb][s4] = k < 27 ? p.w1[(long)ch * p.kpad1 + k] * p.s1[ch] : 0.f;
                }
            }
    } else {
        f32x4 wl[2][2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = (lane >> 4) * 8 + j;
            koff[j >> 2][j & 3] = k < 27 ? (k / 9) * RP + (k % 9) : 0;
            kcol[j >> 2][j & 3] = k < 27 ? k % 9 : 0;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int ch = ct * 16 + (lane & 15);
                wl[ct][j >> 2][j & 3] = k < 27 ? p.w1[(long)ch * p.kpad1 + k] * p.s1[ch] : 0.f;
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) w1b[ct] = splitN<NP ? NP : 1>(wl[ct][0], wl[ct][1]);
        const int n = lane & 15;
        const float* wr = p.wp + (long)n * p.kpadp + (lane >> 4) * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(wr), hi = *reinterpret_cast<const f32x4*>(wr + 4);
        wpb = splitN<NP ? NP : 1>(lo * p.sp[n], hi * p.sp[n]);
    }
    for (int e = tid; e < 9 * 32; e += 256) Wd[e] = p.wd[e] * p.sd[e & 31];
    for (int e = tid; e < 16 * 32; e += 256) {
        const int n = e >> 5, k = e & 31;
        Wp[n * kSLD + k] = p.wp[(long)n * p.kpadp + k] * p.sp[n];
    }
    if (tid < 32) { H1[tid] = p.h1[tid]; Hd[tid] = p.hd[tid]; }
    if (tid < 16) Hp[tid] = p.hp[tid];

    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const long total_tiles = (long)p.B * tiles_per_img;
    // DMA form: the patch rows as 16-byte units u = row * 29 + quad, 609 of them = 10 wave-instructions, wave w issues
    // instructions w, w + 4, w + 8; a unit's LDS slot is u * 16 bytes (the rows are 29 quads apart).  Source rows start at
    // float fs = floor(ix0 * 3 / 4) * 4 of image row iy0 + row; rows outside the image and offsets before the image's first byte
    // are out of the resource's range: the hardware writes zeros (the padding).  Floats beyond the row's end hold the next row's
    // first pixels: the right-edge tiles mask them in the gather.
    int du_row[3], du_q[3];
    if (DMA) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int u = (wave + 4 * j) * 64 + lane;
            du_row[j] = u / kSPQ;
            du_q[j] = u - du_row[j] * kSPQ;
        }
    }
    auto tile_origin = [&](long tile, int& b, int& oy0, int& ox0, int& iy0, int& ix0) {
        b = (int)(tile / tiles_per_img);
        const int rem = (int)(tile - (long)b * tiles_per_img);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        oy0 = ty * kSTH; ox0 = tx * kSTW;                          // output tile origin (150 x 150 grid)
        iy0 = (oy0 - 1) * 2 - p.pad_t; ix0 = (ox0 - 1) * 2 - p.pad_l;   // image patch origin (Conv1-output halo origin: dw SAME pad 1)
    };
    auto issue_patch = [&](long tile, int stage) {
        if constexpr (DMA) {
            int b, oy0, ox0, iy0, ix0;
            tile_origin(tile, b, oy0, ox0, iy0, ix0);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (long)b * p.H * p.W * 3), 0,
                                                                               p.H * p.W * 12, 0x00020000);
            const int fs = ((ix0 * 3) >> 2) * 4;                   // (arithmetic shift: floor for negative origins)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ii = __builtin_amdgcn_readfirstlane(wave) + 4 * j;       // (wave-uniform: the LDS base goes to M0)
                if (ii * 64 >= kSPH * kSPQ) continue;
                const int iy = iy0 + du_row[j];
                const bool ok = du_row[j] < kSPH && (unsigned)iy < (unsigned)p.H;
                const int off = ok ? (iy * p.W * 3 + fs + du_q[j] * 4) * 4 : (int)0x80000000;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + stage * kSPS + ii * 256), 16, off, 0, 0, 0);
            }
        }
    };
    if (DMA && blockIdx.x < total_tiles) {
        issue_patch(blockIdx.x, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first tile's patch: landed before the loop's first barrier
    }
    int it = 0;
    for (long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x, ++it) {
        int b, oy0, ox0, iy0, ix0;
        tile_origin(tile, b, oy0, ox0, iy0, ix0);
        const int cy0 = oy0 - 1, cx0 = ox0 - 1;
        const float* img = p.x + (long)b * p.H * p.W * 3;
        const float* pbuf = patch;                                  // this tile's patch; psh: floats between a row's first float and ix0
        int psh = 0;

        __syncthreads();        // previous tile fully consumed (and the weights are visible); DMA: this tile's patch has landed (vmcnt)
        if constexpr (DMA) {
            pbuf = patch + (it & 1) * kSPS;
            psh = ix0 * 3 - ((ix0 * 3) >> 2) * 4;
            if (tile + gridDim.x < total_tiles) issue_patch(tile + gridDim.x, (it + 1) & 1);      // that stage was read by the PREVIOUS tile's Conv1
        } else {
        // image patch: 21 rows x 111 contiguous floats
        {
            constexpr int NLD = (kSPH * kSPW * 3 + 255) / 256;
            float tmp[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {                         // all loads in flight together
                const int e = tid + i * 256;
                const int r = e / (kSPW * 3), j = e - r * (kSPW * 3);
                const int iy = iy0 + r, ixc = ix0 * 3 + j;         // ixc = ix * 3 + channel
                tmp[i] = 0.f;
                if (!(p.ablate & 8) && e < kSPH * kSPW * 3 && (unsigned)iy < (unsigned)p.H && ixc >= 0 && ixc < p.W * 3)
                    tmp[i] = img[(long)iy * p.W * 3 + ixc];
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + i * 256;
                if (e < kSPH * kSPW * 3) patch[e] = tmp[i];
            }
        }
        __syncthreads();
        }
        // DMA form, tiles whose patch crosses the image's right edge: floats at or beyond (W - ix0) * 3 of a row are not padding
        // but the next row's pixels -> read as zero
        const int xlim = (p.W - ix0) * 3;
        const bool xedge = DMA && xlim < kSPW * 3;

        // ---- Conv1 on the halo (MFMA): 12 pixel tiles of 16 halo pixels, 3 per wave; B fragment = the
        //      lane's pixel x its 4 k of the k-block, gathered from the patch with 4 ds_read_b32
        if (!(p.ablate & 1)) {
            constexpr int NHP = kSIH * kSIW;                       // 180 halo pixels
            int hp[3], rr[3], cc[3];
            const float* pp[3];
            f32x4 a0[3], a1[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                hp[q] = (wave * 3 + q) * 16 + (lane & 15);
                const int hpc = hp[q] < NHP ? hp[q] : NHP - 1;      // the last tile's tail reads a valid pixel, never stored
                rr[q] = hpc / kSIW;
                cc[q] = hpc - rr[q] * kSIW;
                pp[q] = pbuf + (2 * rr[q]) * RP + 2 * cc[q] * 3 + psh;
                a0[q] = *reinterpret_cast<const f32x4*>(H1 + (lane >> 4) * 4);
                a1[q] = *reinterpret_cast<const f32x4*>(H1 + 16 + (lane >> 4) * 4);
            }
            // six independent accumulator chains (3 pixel tiles x 2 channel tiles) keep the matrix pipe issuing
            if (NP == 0) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        float bq[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) bq[q] = pp[q][koff[kb][s4]];
                        if (xedge)
#pragma unroll
                            for (int q = 0; q < 3; ++q)
                                if (cc[q] * 6 + kcol[kb][s4] >= xlim) bq[q] = 0.f;
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            a0[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[0][kb][s4], bq[q], a0[q], 0, 0, 0);
                            a1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[1][kb][s4], bq[q], a1[q], 0, 0, 0);
                        }
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        lo[j] = pp[q][koff[0][j]];
                        hi[j] = pp[q][koff[1][j]];
                    }
                    if (xedge) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (cc[q] * 6 + kcol[0][j] >= xlim) lo[j] = 0.f;
                            if (cc[q] * 6 + kcol[1][j] >= xlim) hi[j] = 0.f;
                        }
                    }
                    const BP<NP ? NP : 1> b = splitN<NP ? NP : 1>(lo, hi);
                    a0[q] = mmaN<NP ? NP : 1>(w1b[0], b, a0[q]);
                    a1[q] = mmaN<NP ? NP : 1>(w1b[1], b, a1[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                // ReLU6 inside the feature map, 0 outside (the depthwise pads Conv1's OUTPUT)
                const bool in = (unsigned)(cy0 + rr[q]) < (unsigned)p.H1 && (unsigned)(cx0 + cc[q]) < (unsigned)p.W1;
                const float hi = in ? 6.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[q][j] = __builtin_amdgcn_fmed3f(a0[q][j], 0.0f, hi);
                    a1[q][j] = __builtin_amdgcn_fmed3f(a1[q][j], 0.0f, hi);
                }
                if (hp[q] < NHP) {
                    *reinterpret_cast<f32x4*>(C1 + hp[q] * kSLD + (lane >> 4) * 4) = a0[q];
                    *reinterpret_cast<f32x4*>(C1 + hp[q] * kSLD + 16 + (lane >> 4) * 4) = a1[q];
                }
            }
        }
        if constexpr (DMA) lds_barrier();      // LDS only: the next tile's patch copy stays in flight until the top of the loop
        else __syncthreads();

        // ---- depthwise: thread = 4 channels x 4 consecutive columns (8 rows x 4 strips x 8 groups = 256)
        if (!(p.ablate & 2)) {
            const int c4 = (tid & 7) * 4, strip = tid >> 3;
            const int oy = strip >> 2, ox = (strip & 3) * 4;
            f32x4 a[4];
            const f32x4 sh = *reinterpret_cast<const f32x4*>(Hd + c4);
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = sh;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                f32x4 e[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) e[j] = *reinterpret_cast<const f32x4*>(C1 + ((oy + ky) * kSIW + ox + j) * kSLD + c4);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Wd + (ky * 3 + kx) * 32 + c4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        a[t] += e[t + kx] * w;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 v = a[t];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = relu6f(v[j]);
                *reinterpret_cast<f32x4*>(D + (oy * kSTW + ox + t) * kSLD + c4) = v;
            }
        }
        if constexpr (DMA) lds_barrier();      // LDS only: the next tile's patch copy stays in flight until the top of the loop
        else __syncthreads();

        // ---- project 32 -> 16 on the MFMA: wave handles pixel tiles wave, wave + 4 (8 tiles of 16 px)
        const int frow = lane & 15, fk = (lane >> 4) * 4;
        f32x4 acc[2];
        acc[0] = acc[1] = *reinterpret_cast<const f32x4*>(Hp + (lane >> 4) * 4);
        if (NP == 0) {
#pragma unroll
            for (int kc = 0; kc < ((p.ablate & 4) ? 0 : 2); ++kc) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(Wp + frow * kSLD + kc * 16 + fk);
                f32x4 db[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) db[q] = *reinterpret_cast<const f32x4*>(D + ((wave + 4 * q) * 16 + frow) * kSLD + kc * 16 + fk);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], db[q][s], acc[q], 0, 0, 0);
            }
        } else if (!(p.ablate & 4)) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* dr = D + ((wave + 4 * q) * 16 + frow) * kSLD + (lane >> 4) * 8;
                const BP<NP ? NP : 1> b = splitN<NP ? NP : 1>(*reinterpret_cast<const f32x4*>(dr), *reinterpret_cast<const f32x4*>(dr + 4));
                acc[q] = mmaN<NP ? NP : 1>(wpb, b, acc[q]);
            }
        }
        // DMA form: the next tile's patch (issued at the top of this tile, a whole tile of work ago) must have LANDED before this wave
        // reaches the barrier that opens the next tile -- waited for here, in front of this tile's stores, so that the wait does not
        // include them (the LDS-only barriers above do not drain vmcnt; hipcc does not order `buffer_load ... lds` against a later
        // __syncthreads() by itself: checked in the ISA)
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int po = (wave + 4 * q) * 16 + (lane & 15);
            const int oy = oy0 + po / kSTW, ox = ox0 + po % kSTW;
            if (oy < p.H1 && ox < p.W1)
                *reinterpret_cast<f32x4*>(p.y + (((long)b * p.H1 + oy) * p.W1 + ox) * 16 + (lane >> 4) * 4) = acc[q];
        }
    }
}

template <int NP, bool DMA>
__global__ __launch_bounds__(256) void mbv2_stem_kernel(const StemParams p) {
    __shared__ __attribute__((aligned(1024))) float sm[stem_lds_floats<DMA>()];
    stem_body<NP, DMA>(p, sm);
}

bool stem_supported(const StemParams& p) { return p.H1 >= 1 && p.W1 >= 1; }

// Which form of the stem kernel runs for a net of this precision: 1 = bf16 operands (precision 1), 3 = the split-bf16
// form (fp32 nets; 115 -> 105 us at B = 64), 0 = the fp32-MFMA form (SSD_STEM_FORM=0: diagnostics)
int stem_form(int precision) {
    static const int form = getenv("SSD_STEM_FORM") ? atoi(getenv("SSD_STEM_FORM")) : 3;
    return precision == 1 ? 1 : (form == 3 ? 3 : 0);
}

int launch_stem(StemParams p, hipStream_t st) {
    if (p.B == 0) return SSD_OK;
    p.tiles_y = (p.H1 + kSTH - 1) / kSTH;
    p.tiles_x = (p.W1 + kSTW - 1) / kSTW;
    const long tiles = (long)p.B * p.tiles_y * p.tiles_x;
    const long blocks = tiles < 512 ? tiles : 512;           // persistent: 2 workgroups per CU
    static const int ablate = getenv("SSD_STEM_ABLATE") ? atoi(getenv("SSD_STEM_ABLATE")) : 0;
    p.ablate = ablate;
    const int form = stem_form(p.bf16);
    // the DMA form's source rows must start 16-byte aligned: W % 4 == 0 and a 16-byte aligned batch (SSD_STEM_DMA=0: diagnostics)
    static const bool want_dma = !(getenv("SSD_STEM_DMA") && atoi(getenv("SSD_STEM_DMA")) == 0);
    const bool dma = want_dma && p.W % 4 == 0 && (((uintptr_t)p.x) & 15) == 0 && (long)p.H * p.W * 12 < 0x7fffffffL;
    const auto fn = dma ? (form == 1 ? mbv2_stem_kernel<1, true> : form == 3 ? mbv2_stem_kernel<3, true> : mbv2_stem_kernel<0, true>)
                        : (form == 1 ? mbv2_stem_kernel<1, false> : form == 3 ? mbv2_stem_kernel<3, false> : mbv2_stem_kernel<0, false>);
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

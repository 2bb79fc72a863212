// Row-band fused MobileNetV2 inverted-residual block for the HIGH-resolution stages (blocks 1-6 of
// SSD300: 150x150 / 75x75 / 38x38 maps, Cin 16 / 24 / 32):
//
//     y = project_BN( relu6(dw_BN( dw3x3( relu6(expand_BN( x * We )) ) )) * Wp ) [+ x]
//
// ([3P] keras-applications MobileNetV2 block_k_expand .. block_k_add, SURVEY.md Appendix A.)
// The whole-image kernel (ssd_imgblock.hip) showed what works on gfx950 -- no horizontal halo, the
// depthwise computed by each lane AS the project MFMA's B fragment (D never returns to LDS), one LDS
// barrier per 16-channel chunk, conflict-free E layout -- but needs the whole map in one workgroup.
// Here a workgroup owns a full-width BAND of R output rows of one image:
//
//   pixel space of the band's input rows: q = rb * P + c, pitch P >= W + 1 (the columns c >= W of a row are
//     the zero padding to the right of row rb AND to the left of row rb + 1), P a multiple of 8; 16
//     consecutive q = one MFMA pixel tile; tiles are dealt round-robin to the 8 waves (T per wave)
//   X   the wave's input tiles as MFMA B fragments, loaded ONCE from global into registers
//   per 16-channel chunk of the Ce expanded channels (E double-buffered in LDS, ONE barrier per chunk):
//     A  expand   E[q][16] = relu6(X[q][Cin] * We[Cin][16] + shift), 0 at pad positions / rows outside the
//                 image -> LDS row e = q + 8, 64-byte rows, 16-byte quad index XOR (e >> 1) & 3:
//                 conflict-free b128 tile writes and stride-1 fragment reads at any alignment
//     B  depthw.  each lane computes ITS output pixel x 4 channels from 9 ds_read_b128 (3 address
//                 registers per tile: the dy offsets are immediates because P % 8 == 0 keeps the swizzle)
//                 = exactly the B fragment of phase C
//     C  project  acc[qo][Cout] += D[qo][16] * Wp[16][Cout], accumulators in registers
//   the A fragments of We / Wp (a few KB per chunk, identical for every workgroup: L1 / L2 resident) are
//   loaded straight from global one chunk ahead -- no weight staging through LDS
//   epilogue    y = acc + shift (+ x), 16-byte stores
//
// Only the two (stride 1) or one (stride 2) halo rows between bands are expanded twice (1.2-1.3x of the
// expand, nothing of the depthwise / project) where the 8x8 tiles of ssd_fused.hip recompute 1.56x.
#include <cstdlib>

#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kBThreads = 512;
constexpr int kBC = 16;            // expanded channels per chunk

__device__ __forceinline__ void band_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS rows of the E chunk for T input tiles per wave: 8 leading zero rows (q = -1 is read by the left
// tap of column 0 in band row 0) + every tile slot
constexpr int band_ne(int T) { return 8 + T * 8 * 16; }

// WDMA (round 6, FusedBlockParams.form2): the fp32 weight fragments reach LDS by LDS-DMA (two stages each of the chunk's We rows
// and Wp rows; 1 KB blocks of 16 rows x 16 floats, quad-swizzled through the per-lane source offset) instead of waiting in
// registers one chunk ahead
typedef __attribute__((address_space(3))) void* band_lds_dst_t;
template <int CIN, int NT, int T, int TO, int S, int P, bool DBG, bool WDMA>
__device__ __forceinline__ void band_body(const FusedBlockParams& p, char* __restrict__ smem) {
    static_assert(P % 8 == 0, "the dy tap offsets must keep the quad swizzle");
    constexpr int KC = CIN / 16;                  // 16-wide k blocks of the expand
    constexpr bool TAIL = (CIN % 16) == 8;        // + one 8-wide k tail (Cin = 24): 2 MFMA k-steps
    constexpr int NE = band_ne(T);
    constexpr int EBUF = NE * kBC * 4;            // bytes per E buffer

    char* Es = smem;                                             // [2][NE][16] floats, swizzled
    float* Ps = reinterpret_cast<float*>(smem + 2 * EBUF);       // [11][Ce]: expand shift, taps [9], depthwise shift

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the tile-count tests below are s_cbranch, not exec masks
    const int nb = p.bands;
    // XCD-aware item order: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (own L2 each); the
    // bands of one image -- which share their halo rows of x -- get ids that land on the same XCD
    const int items = p.B * nb;
    const int bid = (items & 7) == 0 ? (int)(blockIdx.x & 7) * (items >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int img = bid / nb, band = bid - img * nb;
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, Ce = p.Ce;
    const int ro0 = band * Ho / nb, R = (band + 1) * Ho / nb - ro0;
    const int ri0 = S * ro0 - p.pad_t;            // first input row of the band (may be -1)
    const int HB = S * (R - 1) + 3, QB = HB * P;
    const int npt = (QB + 15) >> 4;               // input pixel tiles
    const int Po = Wo + 1, npo = (R * Po + 15) >> 4;
    const int nchunk = Ce / kBC;
    // tiles are dealt round-robin: tile = t * 8 + wave; this wave's counts
    const int nti = npt > wave ? (npt - wave + 7) >> 3 : 0;
    const int nto = npo > wave ? (npo - wave + 7) >> 3 : 0;

    // diagnostics (ssd_net_profile_fused): per-wave cycles of 0 prologue, 1 barrier wait, 2 depthwise,
    // 3 project, 4 expand, 5 epilogue
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long tk0 = DBG ? clock64() : 0;
#define BTICK(i) do { if (DBG) { const long long t1_ = clock64(); tacc[i] += t1_ - tk0; tk0 = t1_; } } while (0)

    // ---- per-channel parameters -> LDS; leading zero rows of both E buffers
    for (int u = tid; u < 11 * (Ce / 4); u += kBThreads) {
        const int row = u / (Ce / 4), c4 = (u - row * (Ce / 4)) * 4;
        const float* src = row == 0 ? p.eh : row == 10 ? p.dh : p.wd + (long)(row - 1) * Ce;
        *reinterpret_cast<f32x4*>(Ps + row * Ce + c4) = *reinterpret_cast<const f32x4*>(src + c4);
    }
    if (tid < 64) *reinterpret_cast<f32x4*>(Es + (tid >> 5) * EBUF + (tid & 31) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the wave's input tiles: B fragments of X in registers, loaded once
    f32x4 xb[T][KC];
    f32x2 xt[T];
    unsigned realm = 0;                           // bit t: this lane's pixel of tile t is a real image pixel
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = t * 8 + wave;
        const int q = tile * 16 + l15;
        const int rb = q / P, c = q - rb * P;
        const int ri = ri0 + rb;
        const bool real = tile < npt && rb < HB && c < W && (unsigned)ri < (unsigned)H;
        realm |= real ? (1u << t) : 0u;
        const float* xp = p.x + (((long)img * H + (real ? ri : 0)) * W + (real ? c : 0)) * CIN;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            xb[t][kc] = real ? *reinterpret_cast<const f32x4*>(xp + kc * 16 + g4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (TAIL) xt[t] = real ? *reinterpret_cast<const f32x2*>(xp + KC * 16 + g4 * 2) : f32x2{0.f, 0.f};
    }
    // E write address of this lane inside a tile slot (tile t adds t * 8 tiles * 1 KiB): row 8 + l15, quad swizzled
    const int ew = (8 + wave * 16 + l15) * 64 + ((g4 ^ ((l15 >> 1) & 3)) << 4);

    // ---- the wave's output tiles (own pixel space qo = rol * Po + co): window origins in E, 3 addresses per tile
    int ea[TO][3];
    int opix[TO];                                 // (ro0 + rol) * Wo + co of a real output pixel, else -1
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const int tile = t * 8 + wave;
        const int qo = tile * 16 + l15;
        const int rol = qo / Po, co = qo - rol * Po;
        const bool realo = tile < npo && rol < R && co < Wo;
        opix[t] = realo ? (ro0 + rol) * Wo + co : -1;
        const int qor = realo ? (S * rol) * P + S * co - p.pad_l : 0;      // window origin (tap dy = dx = 0); -1 is the zero row
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int e = 8 + qor + dx;
            ea[t][dx] = e * 64 + ((g4 ^ ((e >> 1) & 3)) << 4);
        }
    }

    // ---- A fragments of the weights: straight from global (L1 / L2 hits) one chunk ahead, or (WDMA) through LDS
    constexpr int NBE = KC + (TAIL ? 1 : 0);      // 1 KB blocks of a We chunk (16 rows x 16 floats each)
    float* Wes = Ps + 11 * p.Ce;                  // WDMA: [2][NBE] blocks of 256 floats
    float* Wps = Wes + 2 * NBE * 256;             //       [2][NT] blocks
    const int fslot = l15 * 16 + ((g4 ^ ((l15 >> 1) & 3)) * 4);                  // the lane's 16-byte slot inside a block
    const int tslot = l15 * 16 + (((g4 >> 1) ^ ((l15 >> 1) & 3)) * 4) + (g4 & 1) * 2;   // ... its 8 bytes of the k tail
    const int dr = lane >> 2, dq4 = ((lane & 3) ^ ((lane >> 3) & 3)) * 4;
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.we), 0, (int)((long)p.Ce * p.kpad_e * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wp), 0, (int)((long)p.npad_p * p.kpad_p * 4), 0x00020000);
    const int voff_e = (dr * p.kpad_e + dq4) * 4, voff_p = (dr * p.kpad_p + dq4) * 4;
    auto dma_we = [&](int j, int stage) {
        for (int b = wave; b < NBE; b += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_e, (band_lds_dst_t)(Wes + (stage * NBE + b) * 256), 16, voff_e,
                                                     (int)(((long)j * kBC * p.kpad_e + b * 16) * 4), 0, 0);
    };
    auto dma_wp = [&](int j, int stage) {
        for (int b = wave; b < NT; b += 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (band_lds_dst_t)(Wps + (stage * NT + b) * 256), 16, voff_p,
                                                     (int)(((long)b * 16 * p.kpad_p + j * kBC) * 4), 0, 0);
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    f32x4 wa[KC], wan[KC], wp[NT], wpn[NT];
    f32x2 wat, watn;
    auto load_we = [&](f32x4 (&a)[KC], f32x2& at, int j) {
        if constexpr (WDMA) {
            const float* wb = Wes + (j & 1) * NBE * 256;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) a[kc] = *reinterpret_cast<const f32x4*>(wb + kc * 256 + fslot);
            if (TAIL) at = *reinterpret_cast<const f32x2*>(wb + KC * 256 + tslot);
        } else {
            const float* wr = p.we + (long)(j * kBC + l15) * p.kpad_e;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) a[kc] = *reinterpret_cast<const f32x4*>(wr + kc * 16 + g4 * 4);
            if (TAIL) at = *reinterpret_cast<const f32x2*>(wr + KC * 16 + g4 * 2);
        }
    };
    auto load_wp = [&](f32x4 (&a)[NT], int j) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if constexpr (WDMA) a[ni] = *reinterpret_cast<const f32x4*>(Wps + ((j & 1) * NT + ni) * 256 + fslot);
            else a[ni] = *reinterpret_cast<const f32x4*>(p.wp + (long)(ni * 16 + l15) * p.kpad_p + j * kBC + g4 * 4);
        }
    };
    if constexpr (WDMA) {
        dma_we(0, 0);
        if (nchunk > 1) dma_we(1, 1);
        dma_wp(0, 0);
        dma_wait();
    } else {
        load_we(wa, wat, 0);
        load_wp(wp, 0);
    }
    __syncthreads();                              // Ps, zero rows

    auto expand = [&](int j, const f32x4 (&a)[KC], const f32x2 at) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + j * kBC + g4 * 4);
        char* eb = Es + (j & 1) * EBUF + ew;
        // two tiles at a time: independent accumulator chains (dependent fp32 MFMAs issue every 40 cycles, not 32)
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += 2) {
            if (t0 >= nti) break;                 // scalar
            f32x4 acc0 = sh, acc1 = sh;
            const bool two = t0 + 1 < T;
            if (!(DBG && (p.ablate & 1))) {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][s], xb[t0][kc][s], acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][s], xb[t0 + 1 < T ? t0 + 1 : t0][kc][s], acc1, 0, 0, 0);
                    }
                if (TAIL) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[s], xt[t0][s], acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[s], xt[t0 + 1 < T ? t0 + 1 : t0][s], acc1, 0, 0, 0);
                    }
                }
            }
            const float hi0 = (realm >> t0) & 1u ? 6.0f : 0.0f;           // relu6 at real pixels, 0 at pad positions
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[e] = __builtin_amdgcn_fmed3f(acc0[e], 0.0f, hi0);
            *reinterpret_cast<f32x4*>(eb + t0 * 8192) = acc0;
            if (two && t0 + 1 < nti) {
                const float hi1 = (realm >> (t0 + 1)) & 1u ? 6.0f : 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[e] = __builtin_amdgcn_fmed3f(acc1[e], 0.0f, hi1);
                *reinterpret_cast<f32x4*>(eb + (t0 + 1) * 8192) = acc1;
            }
        }
    };

    f32x4 acc[TO][NT];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[t][ni] = f32x4{0.f, 0.f, 0.f, 0.f};


    // depthwise (this lane's pixel x 4 channels = the B fragment) + project MFMAs of chunk i, tile by tile.
    // Measured and not kept (each within the +-3 % run-to-run noise, at a cost in registers): requesting tile
    // t + 1's nine E vectors before tile t's arithmetic (LDS latency is not what the time goes to), scalar
    // v_fma_f32 instead of v_pk_fma_f32.
    auto dwproject = [&](int i, const f32x4 (&wpc)[NT]) {
        const char* eb = Es + (i & 1) * EBUF;
        f32x4 w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(Ps + (1 + k) * Ce + i * kBC + g4 * 4);
        const f32x4 dh = *reinterpret_cast<const f32x4*>(Ps + 10 * Ce + i * kBC + g4 * 4);
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (t >= nto) break;                  // scalar
            f32x4 d = dh;
            if (!(DBG && (p.ablate & 2))) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        d += *reinterpret_cast<const f32x4*>(eb + ea[t][dx] + dy * P * 64) * w[dy * 3 + dx];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_fmed3f(d[e], 0.0f, 6.0f);
            BTICK(2);
            if (!(DBG && (p.ablate & 4)))
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[t][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpc[ni][s4], d[s4], acc[t][ni], 0, 0, 0);
            BTICK(3);
        }
    };

    // (Running { expand (i + 1), depthwise + project (i) } in the opposite order on waves 4-7, so that SIMD
    // partners are never in the same phase, was measured: +-3 %, no gain -- the kernel is issue bound, not
    // latency bound.)
    if constexpr (WDMA) {
        load_we(wa, wat, 0);
        expand(0, wa, wat);
        BTICK(0);
        for (int i = 0; i < nchunk; ++i) {
            dma_wait();                 // the copies of an iteration ago have landed ...
            band_lds_barrier();         // ... and are visible; E(i) is complete; everyone is done reading E(i - 1)
            BTICK(1);
            if (i + 2 < nchunk) dma_we(i + 2, i & 1);           // the stage expand(i) read before this barrier
            if (i + 1 < nchunk) dma_wp(i + 1, (i + 1) & 1);     // the stage dwproject(i - 1) read before this barrier
            load_wp(wp, i);
            dwproject(i, wp);
            if (i + 1 < nchunk) {
                load_we(wa, wat, i + 1);
                expand(i + 1, wa, wat);
            }
            BTICK(4);
        }
    } else {
    expand(0, wa, wat);
    if (nchunk > 1) load_we(wan, watn, 1);
    BTICK(0);
    for (int i = 0; i < nchunk; ++i) {
        band_lds_barrier();         // E(i) is complete; everyone is done reading E(i - 1)
        BTICK(1);
        if (i + 1 < nchunk) load_wp(wpn, i + 1);      // in flight across the depthwise / project
        dwproject(i, wp);
        if (i + 1 < nchunk) {
            expand(i + 1, wan, watn);
            if (i + 2 < nchunk) load_we(wan, watn, i + 2);   // in flight across the barrier and the next depthwise / project
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) wp[ni] = wpn[ni];
        }
        BTICK(4);
    }

    }

    // ---- epilogue: y = acc + shift (+ x); lane = 4 consecutive output channels of its pixel
    const long img_o = (long)img * Ho * Wo;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (opix[t] < 0) continue;
        float* yp = p.y + (img_o + opix[t]) * p.Cout + g4 * 4;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if (ni * 16 + g4 * 4 >= p.Cout) continue;          // Cout = 24: the second channel tile is half empty
            f32x4 v = acc[t][ni] + *reinterpret_cast<const f32x4*>(p.ph + ni * 16 + g4 * 4);
            if (p.residual)                                     // stride 1, Cin == Cout: same layout as y
                v = v + *reinterpret_cast<const f32x4*>(p.x + (img_o + opix[t]) * p.Cout + ni * 16 + g4 * 4);
            *reinterpret_cast<f32x4*>(yp + ni * 16) = v;
        }
    }
    BTICK(5);
    if (DBG && p.dbg && lane == 0)
        for (int i_ = 0; i_ < 6; ++i_) p.dbg[((long)blockIdx.x * 8 + wave) * 6 + i_] = tacc[i_];
#undef BTICK
}

template <int CIN, int NT, int T, int TO, int S, int P, bool DBG, bool WDMA = false>
__global__ __launch_bounds__(kBThreads) void mbv2_band_block_kernel(const FusedBlockParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem_band[];
    band_body<CIN, NT, T, TO, S, P, DBG, WDMA>(p, smem_band);
}

typedef void (*band_kernel_t)(const FusedBlockParams);
struct BandCfg {
    int cin, nt, t, to, stride, pitch;
    band_kernel_t fn, fn_dbg;       // fn_dbg: cycle counters + phase ablation (ssd_net_profile_fused)
    band_kernel_t fn_d;             // weights staged by LDS-DMA (FusedBlockParams.form2)
};
#define BCFG(CIN, NT, T, TO, S, P) {CIN, NT, T, TO, S, P, mbv2_band_block_kernel<CIN, NT, T, TO, S, P, false>, mbv2_band_block_kernel<CIN, NT, T, TO, S, P, true>, \
                                    mbv2_band_block_kernel<CIN, NT, T, TO, S, P, false, true>}
const BandCfg kBand[] = {
    BCFG(16, 2, 9, 2, 2, 152),   // block 1: 16 -> 96 -> 24, 150x150 -> 75x75
    BCFG(24, 2, 8, 6, 1, 80),    // block 2: 24 -> 144 -> 24 (+x) at 75x75
    BCFG(24, 2, 7, 2, 2, 80),    // block 3: 24 -> 144 -> 32, 75x75 -> 38x38
    BCFG(32, 2, 4, 4, 1, 40),    // blocks 4-5: 32 -> 192 -> 32 (+x) at 38x38
    BCFG(32, 4, 4, 1, 2, 40),    // block 6: 32 -> 192 -> 64, 38x38 -> 19x19
    // the 512x512 graph (BASELINE configs[4]): maps 128 / 64 wide.  (Block 1 at 256x256 stays on the 8x8-tile kernel: a full-width
    // band holds ONE output row -- 3 input rows of pitch 264 fill the 9 tile slots -- and measured 127 us against 116.)
    BCFG(24, 2, 8, 6, 1, 136),   // block 2 at 128x128
    BCFG(24, 2, 7, 2, 2, 136),   // block 3: 128x128 -> 64x64
    BCFG(32, 2, 4, 4, 1, 72),    // blocks 4-5 at 64x64
    BCFG(32, 4, 4, 1, 2, 72),    // block 6: 64x64 -> 32x32
};

// largest band (output rows) a configuration can hold: input tiles and output tiles both have to fit
int band_max_rows(const BandCfg& c, const FusedBlockParams& p) {
    const int hb = c.t * 8 * 16 / c.pitch;                       // band input rows that fit the tile slots
    int r = c.stride == 1 ? hb - 2 : (hb - 1) / 2;
    const int po = p.Wo + 1;
    while (r > 0 && (r * po + 15) / 16 > c.to * 8) --r;
    return r;
}

const BandCfg* pick_band(const FusedBlockParams& p) {
    if (p.Ce % kBC != 0 || p.kpad_e % 4 != 0 || p.kpad_p % 4 != 0 || p.Cout % 8 != 0) return nullptr;
    if (p.stride == 1 && (p.H != p.Ho || p.W != p.Wo || p.pad_t != 1 || p.pad_l != 1)) return nullptr;
    if (p.stride == 2 && (p.residual || p.Ho != (p.H + 1) / 2 || p.Wo != (p.W + 1) / 2 || p.pad_t > 1 || p.pad_l > 1 ||
                          p.pad_t < 0 || p.pad_l < 0))
        return nullptr;
    if (p.residual && p.Cin != p.Cout) return nullptr;
    if (p.e_out) return nullptr;
    for (const auto& c : kBand) {
        if (c.cin != p.Cin || c.stride != p.stride || (p.Cout + 15) / 16 != c.nt || p.npad_p < c.nt * 16) continue;
        if (p.W + 1 > c.pitch || p.W + 8 < c.pitch) continue;    // the configuration's pitch is for this width
        // stride 2: the right-most tap column 2 (Wo - 1) - pad_l + 2 must be a pad column (or inside the map)
        if (p.stride == 2 && 2 * (p.Wo - 1) - p.pad_l + 2 >= c.pitch) continue;
        if (band_max_rows(c, p) < 1) continue;
        return &c;
    }
    return nullptr;
}

size_t band_lds_bytes(const BandCfg& c, const FusedBlockParams& p) {
    return (size_t)2 * band_ne(c.t) * kBC * 4 + (size_t)11 * p.Ce * 4;
}

}  // namespace

bool band_block_supported(const FusedBlockParams& p) {
    const BandCfg* c = pick_band(p);
    return c && band_lds_bytes(*c, p) <= 160 * 1024;
}

int launch_band_block(FusedBlockParams p, hipStream_t st) {
    const BandCfg* c = pick_band(p);
    if (!c) {
        set_error("band block: unsupported shape Cin=%d Ce=%d Cout=%d %dx%d stride=%d", p.Cin, p.Ce, p.Cout, p.H, p.W, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    static const int ablate = getenv("SSD_BAND_ABLATE") ? atoi(getenv("SSD_BAND_ABLATE")) : 0;
    if (!p.ablate) p.ablate = ablate;
    const int rmax = band_max_rows(*c, p);
    p.bands = (p.Ho + rmax - 1) / rmax;          // band b = output rows [b * Ho / bands, (b + 1) * Ho / bands)
    const size_t lds = band_lds_bytes(*c, p);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "band block: needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)c->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    band_kernel_t fn = (p.dbg || p.ablate) ? c->fn_dbg : c->fn;
    size_t lds_use = lds;
    const size_t lds_d = lds + (size_t)2 * (c->cin / 16 + ((c->cin % 16) == 8 ? 1 : 0) + c->nt) * 1024;     // + two stages of We and Wp blocks
    if (p.form2 && fn == c->fn && lds_d <= 160 * 1024) {
        fn = c->fn_d;
        lds_use = lds_d;
    }
    if (lds_use > 64 * 1024 && fn != c->fn)
        SSD_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use));
    hipLaunchKernelGGL(fn, dim3((unsigned)((long)p.B * p.bands)), dim3(kBThreads), lds_use, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

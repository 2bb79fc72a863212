// conv_mfma_kernel (implicit GEMM on the matrix cores, see ssd_conv.hip for the mapping) as a device body shared by
// the fp32-MFMA kernels (ssd_conv.hip) and the split-bf16 kernels (ssd_conv3.hip, SPLIT3: both operands are split
// exactly into three bf16 planes on their way into LDS and every 16 x 16 x 32 block is six v_mfma_f32_16x16x32_bf16,
// ssd_bf16x3.h).
#pragma once
#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == SSD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == SSD_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

// Epilogue of the implicit-GEMM tiles (shared by conv_mfma_body and the LDS-DMA tiles of ssd_convdma.hip): the lane holds
// out[m = m0 + (wm * MT + mi) * 16 + (lane & 15)][n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4 + 0..3].
template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[MT][NT], const long m0, const int n0,
                                              const int wm, const int wn, const int lane, const int HoWo) {
    // ---- epilogue: lane holds out[m = .. + (lane & 15)][n = .. + (lane >> 4) * 4 + 0..3]
    // (M < 2^31 is checked on the host: 32-bit index arithmetic)
    if (p.split_k > 1) {           // partial sums only; scale/shift/act/residual happen in the reduce
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int m = (int)m0 + (wm * MT + mi) * 16 + (lane & 15);
            if (m >= (int)p.M) continue;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
                if (n >= p.Cout) continue;
                float* prow = p.partial + ((long)blockIdx.y * p.M + m) * p.Cout + n;
                if (n + 3 < p.Cout && (p.Cout & 3) == 0) {
                    *reinterpret_cast<f32x4*>(prow) = acc[mi][ni];
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (n + j < p.Cout) prow[j] = acc[mi][ni][j];
                }
            }
        }
        return;
    }
    // All loads of the epilogue are issued first (scale/shift per column group, residual per
    // tile), the stores follow back to back: a load between two stores costs a full store
    // round trip on gfx950 (vmcnt counts stores and the waits are not selective).
    f32x4 sc[NT], sh[NT];
    bool vecn[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
        const bool straddle = p.n_split && n < p.n_split && n + 3 >= p.n_split;
        vecn[ni] = (n + 3 < p.Cout) && !straddle;
        sc[ni] = f32x4{1.f, 1.f, 1.f, 1.f};
        sh[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (vecn[ni]) {
            if (p.scale) sc[ni] = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) sh[ni] = *reinterpret_cast<const f32x4*>(p.shift + n);
        }
    }
    const bool res_vec = p.residual && (p.Cout & 3) == 0;
    f32x4 rs[MT][NT];
    if (res_vec) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int m = (int)m0 + (wm * MT + mi) * 16 + (lane & 15);
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
                rs[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < (int)p.M && vecn[ni]) rs[mi][ni] = *reinterpret_cast<const f32x4*>(p.residual + (long)m * p.Cout + n);
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = (int)m0 + (wm * MT + mi) * 16 + (lane & 15);
        if (m >= (int)p.M) continue;
        const int b = m / HoWo;
        const int pix = m - b * HoWo;
        float* orow = p.out + (long)b * p.out_batch_stride + (long)pix * p.out_pixel_stride;
        float* orow2 = p.n_split ? p.out2 + (long)b * p.out2_batch_stride + (long)pix * p.out2_pixel_stride - p.n_split : nullptr;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
            if (n >= p.Cout) continue;
            f32x4 v = acc[mi][ni];
            if (vecn[ni]) {
                v = v * sc[ni] + sh[ni];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
                if (res_vec) {
                    v = v + rs[mi][ni];
                } else if (p.residual) {
                    const float* rr = p.residual + (long)m * p.Cout + n;
                    for (int j = 0; j < 4; ++j) v[j] += rr[j];
                }
                const bool side2 = p.n_split && n >= p.n_split;
                float* dst = (side2 ? orow2 : orow) + n;
                if ((side2 ? p.vec_store2 : p.vec_store) && ((((uintptr_t)dst) & 15) == 0)) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
                }
                if (p.op) store_planes4(p.op, p.op_plane, p.op_np, m, n, p.M, v);
            } else {
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= p.Cout) break;
                    float t = v[j];
                    if (p.scale) t = t * p.scale[n + j];
                    if (p.shift) t = t + p.shift[n + j];
                    t = apply_act(t, p.act);
                    if (p.residual) t += p.residual[(long)m * p.Cout + n + j];
                    float* drow = (p.n_split && n + j >= p.n_split) ? orow2 : orow;
                    drow[n + j] = t;
                }
            }
        }
    }
}

// Diagnostic builds only (tests/micro/conv_ablate.py): -DSSD_CONV_ABLATE=bits removes one phase
// of the main loop -- 1 global loads, 2 LDS stores, 4 MFMAs (+ fragment reads), 8 barriers,
// 16 MFMAs only (fragment reads kept),
// 64 LDS-only raw barrier.  0 = the production kernel.
#ifndef SSD_CONV_ABLATE
#define SSD_CONV_ABLATE 0
#endif

template <int MT, int NT, int WM, int WN, int BK, bool GEMM1X1, int NP>
__device__ __forceinline__ void conv_mfma_body(const ConvParams& p, float* __restrict__ smem) {
    // NP: 0 = fp32 MFMA tiles; 3 = split-bf16 tiles (exact three-way split, six bf16 MFMAs per product, fp32 results);
    // 1 = bf16 tiles (the net's "precision 1" mode: operands rounded once to bf16, ONE MFMA per product, fp32 accumulation)
    constexpr bool SPLIT3 = NP != 0;
    constexpr int WPL = NP == 1 ? 3 : 0;        // first weight plane read: [h, m, l, r] -- r = the bf16 rounding
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN;
    static_assert(NP == 0 || BK == 32, "split-bf16 / bf16 tiles: one K = 32 MFMA step per tile");
    // LDS tile rows: BK >= 32 uses UNPADDED rows with an XOR swizzle of the 16-byte column index,
    // col ^ f(row) with f = (row >> 1) & 7 (BK 32) / row & 15 (BK 64).  Under gfx950's actual
    // ds_read_b128 / ds_write_b128 lane grouping ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) this
    // is conflict-free for the fragment reads AND the tile writes, whereas rows padded to BK + 4
    // are 2-way conflicted on both (SQ_LDS_BANK_CONFLICT was exactly 1/3 of SQ_LDS_IDX_ACTIVE
    // for every config) -- and it takes 11 % less LDS.  BK = 16 keeps the padded rows.
    constexpr bool SWZ = BK >= 32;
    constexpr int LDK = SWZ ? BK : BK + 4;
    constexpr int UPR = BK / 4;                 // float4 units per tile row
    constexpr int XU = BM * UPR, WU = BN * UPR;
    constexpr int NTHR = 64 * WM * WN;          // 256; the split-bf16 tiles also come with 8 waves
    constexpr int XP = (XU + NTHR - 1) / NTHR, WP = (WU + NTHR - 1) / NTHR;
    static_assert(NTHR == 256 || (SPLIT3 && (NTHR == 512 || NTHR == 768 || NTHR == 1024)),
                  "4 waves per block (8 / 12 / 16: split-bf16 and bf16 tiles only)");
    // two LDS stages: tile kt+1 is written while tile kt is multiplied -> one barrier per K tile
    // (fp32: 2 * (BM + BN) * LDK floats; split-bf16: 2 stages x 3 planes x (BM + BN) rows of 32 bf16 = 64 bytes)
    constexpr int STAGE_FLOATS = SPLIT3 ? (BM + BN) * 16 * NP : (BM + BN) * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nb_n = (p.Cout + BN - 1) / BN;
    const int mblk = blockIdx.x / nb_n, nblk = blockIdx.x - mblk * nb_n;
    const long m0 = (long)mblk * BM;
    const int n0 = nblk * BN;
    const int HoWo = p.Ho * p.Wo;

    // ---- per-thread load bookkeeping (rows are the same for every K tile)
    // Addresses are (block base pointer, uniform) + (per-row byte offset, VGPR) + (per-tile byte
    // offset, SGPR): a load costs an add and a select in the loop.  Lanes whose element is
    // padding / out of range read offset 0 of the block base (always mapped) and the value is
    // replaced by zero when the tile is written to LDS -- the loop body has no divergent branch.
    const int b_first = (int)m0 / HoWo;
    const char* xbase = reinterpret_cast<const char*>(p.in + (GEMM1X1 ? m0 * p.Cin : (long)b_first * p.H * p.W * p.Cin));
    const char* wbase = reinterpret_cast<const char*>(p.w + (long)n0 * p.Kpad);
    int xoff[XP];              // bytes, relative to xbase (may be negative on padded taps: those are invalid)
    unsigned xvalid[XP];       // general: bit t = tap t of this row lies inside the image; 1x1: row valid
    int woff[WP];              // bytes, relative to wbase
    // split-bf16: the weights come pre-split (three bf16 planes [3][Npad][Kpad] behind the packed fp32 weights,
    // launch_pack_split): 16-byte units (plane, row, quad of 8 k) straight into their swizzled LDS slots
    constexpr int WU3 = BN * 4 * (SPLIT3 ? NP : 1), WP3 = SPLIT3 ? (WU3 + NTHR - 1) / NTHR : 1;
    int w3off[WP3], w3dst[WP3];
    const short* w3base = nullptr;
    if constexpr (SPLIT3) {
        w3base = p.w3 + (long)n0 * 32;              // planes [Kpad / 32][Npad][32]: row n of k-slice s at (s * Npad + n) * 32
#pragma unroll
        for (int ps = 0; ps < WP3; ++ps) {
            const int u = min(tid + ps * NTHR, WU3 - 1);
            const int pl = u / (BN * 4), rem = u - pl * (BN * 4);
            const int row = rem >> 2, q = rem & 3;
            const int r = min(row, min(BN, p.Npad - n0) - 1);     // clamped: rows past the tile / Npad are never used
            w3off[ps] = (pl + WPL) * p.Npad * p.Kpad + r * 32 + q * 8;
            w3dst[ps] = (pl * BN + row) * 64 + ((q ^ ((row >> 1) & 3)) << 4);
        }
    }
    const float* xrow1[XP];    // 1x1 path: plain row pointers + predicated loads measured faster there
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int u = tid + ps * NTHR;
        const int row = u / UPR;
        const long m = m0 + row;
        const bool ok = (u < XU) && (m < p.M);
        xvalid[ps] = 0;
        xoff[ps] = 0;
        if (GEMM1X1) {
            xvalid[ps] = ok ? 1u : 0u;
            xrow1[ps] = p.in + (ok ? m : 0) * p.Cin + (tid % UPR) * 4;
        } else if (ok) {
            const int b = (int)m / HoWo;              // M < 2^31 (host check)
            const int pix = (int)m - b * HoWo;
            const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
            xoff[ps] = ((((b - b_first) * p.H + iy0) * p.W + ix0) * p.Cin + (tid % UPR) * 4) * 4;
            for (int ky = 0; ky < p.kh; ++ky)
                for (int kx = 0; kx < p.kw; ++kx) {
                    const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) xvalid[ps] |= 1u << (ky * p.kw + kx);
                }
        }
    }
#pragma unroll
    for (int ps = 0; ps < WP; ++ps) {
        const int u = tid + ps * NTHR;
        const int r = min(u / UPR, min(BN, p.Npad - n0) - 1);     // clamped: rows past the tile / Npad are never used
        woff[ps] = (r * p.Kpad + (tid % UPR) * 4) * 4;
    }
    const int kq4 = (tid % UPR) * 4;           // identical for every pass (256 % UPR == 0)

    const int nkt_total = (p.K + BK - 1) / BK;
    int kt_begin = 0, kt_end = nkt_total;
    if (p.split_k > 1) {
        const int per = (nkt_total + p.split_k - 1) / p.split_k;
        kt_begin = blockIdx.y * per;
        kt_end = min(nkt_total, kt_begin + per);
    }

    // uniform state of the tile being loaded: k0, its tap (general path) and byte offsets
    int l_k0 = 0, l_tap = 0, l_ci = 0, l_ky = 0, l_kx = 0, l_xtile = 0;
    // General path: the K tiles are walked channel-slice-major, taps innermost -- tile t covers
    // tap t % (kh*kw) of input channels [(t / (kh*kw)) * BK, +BK).  The kh*kw taps of one channel
    // slice re-read the same 128-byte pixel lines, so they hit in L1/L2 back to back; with the
    // tap-major order (whole Cin per tap) every tap re-fetched the block's input region from
    // the fabric (FETCH_SIZE of the 3x3 head conv: 6.7x its algorithmic bytes).
    const int ntaps = p.kh * p.kw;
    auto tile_setup = [&](int kt) {            // once; afterwards tile_advance()
        if (GEMM1X1) {
            l_k0 = kt * BK;
            l_xtile = l_k0 * 4;
        } else {
            const int cs = kt / ntaps;
            l_tap = kt - cs * ntaps;
            l_ci = cs * BK;
            l_ky = l_tap / p.kw;
            l_kx = l_tap - l_ky * p.kw;
            l_k0 = l_tap * p.Cin + l_ci;
            l_xtile = ((l_ky * p.dil * p.W + l_kx * p.dil) * p.Cin + l_ci) * 4;
        }
    };
    auto tile_advance = [&]() {                // kt -> kt + 1 (Cin % BK == 0 on the general path)
        if (GEMM1X1) {
            l_k0 += BK;
            l_xtile += BK * 4;
        } else {
            ++l_tap;
            if (++l_kx == p.kw) { l_kx = 0; ++l_ky; }
            if (l_tap == ntaps) { l_tap = 0; l_ky = 0; l_kx = 0; l_ci += BK; }
            l_k0 = l_tap * p.Cin + l_ci;
            l_xtile = ((l_ky * p.dil * p.W + l_kx * p.dil) * p.Cin + l_ci) * 4;
        }
    };
    auto x_is_valid = [&](int ps) -> bool {
        if (GEMM1X1) return xvalid[ps] && (l_k0 + kq4 < p.K);
        return (xvalid[ps] >> l_tap) & 1u;
    };

    // general path: the pixels come through a raw BUFFER load whose range ends with the input tensor -- a padded /
    // out-of-range tap reads offset 2^31 (beyond the range) and the hardware returns zeros: no per-element select when the
    // tile is written (16 v_cndmask per thread and K tile; the loop's time is its MFMAs PLUS its other instructions:
    // split-bf16 tiles -5 .. -10 %, profiles/HISTORY.md round 4)
    constexpr bool BUFZ = !GEMM1X1;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xbase), 0,
        // (a tile's own offsets are within a few images of xbase; the range is capped below 2^31 so that the zero-fill offset
        // stays out of range for tensors of any size)
        (int)min((long)0x7fffffffL, (long)(reinterpret_cast<const char*>(p.in + (long)p.B * p.H * p.W * p.Cin) - xbase)),
        0x00020000);
    f32x4 xr[XP], wr[WP];
    auto load_tile = [&]() {                   // the tile described by the l_* state
        if (SSD_CONV_ABLATE & 1) return;
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            if (GEMM1X1) {
                xr[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (x_is_valid(ps)) xr[ps] = *reinterpret_cast<const f32x4*>(xrow1[ps] + l_k0);
            } else {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, x_is_valid(ps) ? xoff[ps] + l_xtile : (int)0x80000000, 0, 0);
                xr[ps] = __builtin_bit_cast(f32x4, r);      // valid offsets are >= 0: SGPR base + u32 offset
            }
        }
#pragma unroll
        for (int ps = 0; ps < WP; ++ps) {
            // BK = 64 over a Kpad that is only a multiple of 32: the k >= Kpad half (kq4 >= 32 of
            // the last tile) re-reads the first half, 128 bytes back (finite values; the matching
            // X columns are zero) instead of running past the row / the buffer
            const int koff = (BK <= 32 || l_k0 + kq4 < p.Kpad) ? l_k0 * 4 : -128;
            if (GEMM1X1 && (SSD_CONV_ABLATE & 256)) {
                const int u = tid + ps * NTHR;
                const int n = n0 + u / UPR;
                wr[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u < WU && n < p.Npad && l_k0 + kq4 < p.Kpad)
                    wr[ps] = *reinterpret_cast<const f32x4*>(p.w + (long)n * p.Kpad + l_k0 + kq4);
            } else
            wr[ps] = *reinterpret_cast<const f32x4*>(wbase + (unsigned)(woff[ps] + koff));
        }
    };
    // swizzled float column of this thread's 16-byte unit; the swizzle key of a row is the same in
    // every 256-thread pass (a pass advances the row by 256 / UPR, a multiple of the key's period)
    const int st_row0 = tid / UPR;
    const int st_col = SWZ ? ((((kq4 >> 2) ^ (BK == 32 ? (st_row0 >> 1) & 7 : st_row0 & 15)) << 2)) : kq4;
    auto store_tile = [&](int stage) {
        if (SSD_CONV_ABLATE & 2) return;
        float* Xs = smem + stage * STAGE_FLOATS;
        float* Ws = Xs + BM * LDK;
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            const int u = tid + ps * NTHR;
            if (XU % NTHR == 0 || u < XU)
                *reinterpret_cast<f32x4*>(Xs + (u / UPR) * LDK + st_col) =
                    (GEMM1X1 || BUFZ || x_is_valid(ps)) ? xr[ps] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ps = 0; ps < WP; ++ps) {
            const int u = tid + ps * NTHR;
            if (WU % NTHR == 0 || u < WU) *reinterpret_cast<f32x4*>(Ws + (u / UPR) * LDK + st_col) = wr[ps];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 4;
    // fragment column (floats) per 16-wide k unit: tile rows are multiples of 16, so the swizzle
    // key only depends on frow
    int fcol[BK / 16];
#pragma unroll
    for (int kc = 0; kc < BK / 16; ++kc)
        fcol[kc] = SWZ ? (((kc * 4 + (lane >> 4)) ^ (BK == 32 ? (frow >> 1) & 7 : frow)) << 2) : kc * 16 + fk;
#ifdef SSD_C3_PROF
    long long c3tp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long c3t0 = clock64();
#endif
#ifndef SSD_C3_VARIANT
#define SSD_C3_VARIANT 0      // experiments (tools/gpu/build_c3var.sh + tests/micro/conv3_variants.py); 0 = the production kernel
#endif
#ifdef SSD_C3_PROF            // diagnostics (tests/micro/conv3_prof.py): per-wave cycles of the loop's phases, dumped by one workgroup
#define C3T(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long t_ = clock64(); c3tp[i] += t_ - c3t0; c3t0 = t_; } while (0)
#else
#define C3T(i) do {} while (0)
#endif
#ifndef SSD_C3_ABLATE
#define SSD_C3_ABLATE 0       // diagnostics (tests/micro/conv3_ablate.py): 1 no MFMA, 2 no fragment reads, 4 no split, 8 no global loads, 16 no LDS stores
#endif
    if constexpr (SPLIT3) {
        // ---- split-bf16 main loop (same two-stage pipeline; a deeper global prefetch measured slower: the loop is
        // bound by the L2 -> CU bytes per tile, hence the 8-wave 256-row tiles, not by load latency)
        f32x4 xs0[XP];
        bf16x8 ws0[WP3];
        unsigned vm0 = 0;
        auto load3 = [&](f32x4 (&X)[XP], bf16x8 (&W)[WP3], unsigned& vm) {      // the tile described by the l_* state
            vm = 0;
            if (SSD_C3_ABLATE & 8) return;
#pragma unroll
            for (int ps = 0; ps < XP; ++ps) {
                const bool ok = x_is_valid(ps);
                if (GEMM1X1) {
                    vm |= (ok ? 1u : 0u) << ps;
                    X[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok) X[ps] = *reinterpret_cast<const f32x4*>(xrow1[ps] + l_k0);
                } else {
                    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? xoff[ps] + l_xtile : (int)0x80000000, 0, 0);
                    X[ps] = __builtin_bit_cast(f32x4, r);
                }
            }
#pragma unroll
            for (int ps = 0; ps < WP3; ++ps) W[ps] = *reinterpret_cast<const bf16x8*>(w3base + w3off[ps] + (long)l_k0 * p.Npad);
        };
        // three bf16 planes per operand, rows of 32 bf16 = 64 bytes: a thread's 4 k-values become 8 bytes per plane at
        // 16-byte quad (kq >> 1) ^ ((row >> 1) & 3), half kq & 1 (the fragment reads' swizzle); weights arrive split
        const int kq = tid % UPR;
        const int xcol = ((((kq >> 1) ^ (((tid / UPR) >> 1) & 3)) << 4)) + (kq & 1) * 8;
        auto store3 = [&](int stage, const f32x4 (&X)[XP], const bf16x8 (&W)[WP3], const unsigned vm) {
            char* Xs = reinterpret_cast<char*>(smem + stage * STAGE_FLOATS);
            char* Ws = Xs + NP * BM * 64;
#ifdef SSD_C3_PROF
            if constexpr (NP == 3 && XU % NTHR == 0 && WU3 % NTHR == 0) {      // phases timed apart: load wait / split / store issue / drain
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                C3T(5);
                uint2 hh[XP], mm[XP], ll[XP];
#pragma unroll
                for (int ps = 0; ps < XP; ++ps) {
                    const f32x4 v = (GEMM1X1 || BUFZ || ((vm >> ps) & 1u)) ? X[ps] : f32x4{0.f, 0.f, 0.f, 0.f};
                    split4(v, hh[ps], mm[ps], ll[ps]);
                    asm volatile("" : "+v"(hh[ps]), "+v"(mm[ps]), "+v"(ll[ps]));
                }
                C3T(6);
#pragma unroll
                for (int ps = 0; ps < XP; ++ps) {
                    char* d = Xs + ((tid + ps * NTHR) / UPR) * 64 + xcol;
                    *reinterpret_cast<uint2*>(d) = hh[ps];
                    *reinterpret_cast<uint2*>(d + BM * 64) = mm[ps];
                    *reinterpret_cast<uint2*>(d + 2 * BM * 64) = ll[ps];
                }
#pragma unroll
                for (int ps = 0; ps < WP3; ++ps) *reinterpret_cast<bf16x8*>(Ws + w3dst[ps]) = W[ps];
                asm volatile("" ::: "memory");
                { const long long t_ = clock64(); c3tp[7] += t_ - c3t0; c3t0 = t_; }      // issue only (no lgkmcnt wait)
                C3T(8);                                                                  // drain
                return;
            }
#endif
#pragma unroll
            for (int ps = 0; ps < XP; ++ps) {
                const int u = tid + ps * NTHR;
                if (XU % NTHR == 0 || u < XU) {
                    const f32x4 v = (GEMM1X1 || BUFZ || ((vm >> ps) & 1u)) ? X[ps] : f32x4{0.f, 0.f, 0.f, 0.f};
                    char* d = Xs + (u / UPR) * 64 + xcol;
                    if constexpr (NP == 1) {
                        *reinterpret_cast<uint2*>(d) = rne4(v);
                        continue;
                    }
                    uint2 h, m, l;
                    if (SSD_C3_ABLATE & 4) {
                        h = make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1]));
                        m = make_uint2(__float_as_uint(v[2]), __float_as_uint(v[3]));
                        l = h;
                    } else
                    split4(v, h, m, l);
                    if (SSD_C3_ABLATE & 16) { asm volatile("" ::"v"(h), "v"(m), "v"(l)); continue; }
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + BM * 64) = m;
                    *reinterpret_cast<uint2*>(d + 2 * BM * 64) = l;
                }
            }
#pragma unroll
            for (int ps = 0; ps < WP3; ++ps) {
                if (SSD_C3_ABLATE & 16) { asm volatile("" ::"v"(W[ps])); continue; }
                if (WU3 % NTHR == 0 || tid + ps * NTHR < WU3) *reinterpret_cast<bf16x8*>(Ws + w3dst[ps]) = W[ps];
            }
        };
        const int fq = ((lane >> 4) ^ ((frow >> 1) & 3)) << 4;
        auto mma_tile = [&](int stage) {
            if (SSD_C3_ABLATE & 2) return;
            const char* Xb = reinterpret_cast<const char*>(smem + stage * STAGE_FLOATS);
            const char* Wb = Xb + NP * BM * 64;
            BP<NP> b[MT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const char* r = Xb + ((wm * MT + mi) * 16 + frow) * 64 + fq;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) b[mi].p[pl] = *reinterpret_cast<const bf16x8*>(r + pl * BM * 64);
            }
#if SSD_C3_VARIANT & 1
            // variant 1: two weight fragments at a time -> 2 * MT independent accumulator chains per product term (the
            // per-accumulator order of the six terms is unchanged: same bits)
            constexpr int NT2 = (NP == 3) ? (NT & ~1) : 0;
#pragma unroll
            for (int ni = 0; ni < NT2; ni += 2) {
                const char* r0 = Wb + ((wn * NT + ni) * 16 + frow) * 64 + fq;
                BP<NP> a0, a1;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    a0.p[pl] = *reinterpret_cast<const bf16x8*>(r0 + pl * BN * 64);
                    a1.p[pl] = *reinterpret_cast<const bf16x8*>(r0 + 16 * 64 + pl * BN * 64);
                }
                constexpr int TW[6] = {1, 0, 2, 0, 1, 0}, TX[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.p[TW[t]], b[mi].p[TX[t]], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.p[TW[t]], b[mi].p[TX[t]], acc[mi][ni + 1], 0, 0, 0);
                    }
            }
#else
            constexpr int NT2 = 0;
#endif
#pragma unroll
            for (int ni = NT2; ni < NT; ++ni) {
                const char* r = Wb + ((wn * NT + ni) * 16 + frow) * 64 + fq;
                BP<NP> a;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) a.p[pl] = *reinterpret_cast<const bf16x8*>(r + pl * BN * 64);
                if (SSD_C3_ABLATE & 1) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) asm volatile("" ::"v"(a.p[pl]), "v"(b[mi].p[pl]));
                    continue;
                }
                // (issuing the six products term by term across the MT accumulators measured no faster than the chains)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = mmaN<NP>(a, b[mi], acc[mi][ni]);
            }
        };
        // 8 waves: the second wave group runs its half of the staging BEFORE its MFMAs ("ping-pong"): on every SIMD
        // one wave multiplies while the other splits / stores the next tile, with one barrier per tile as before
        // (a stage is written one iteration after its last read and read one iteration after its last write by
        // either group).  Both groups hold tile kt + 1 in registers when iteration kt starts.
        // 12 waves (three per SIMD): a third group runs its MFMAs between its stores and its loads -- the three waves of a SIMD
        // are in three different phases, and a wave's serial chain (MFMAs + staging; profiles/HISTORY.md, cycle timeline) is a
        // third shorter per unit of tile
        const int wgrp = NTHR > 256 ? __builtin_amdgcn_readfirstlane(tid >> 6) >> 2 : 0;
        const bool late = NTHR > 256 && wgrp == (NTHR == 512 ? 1 : 2);
        const bool mid = NTHR >= 768 && wgrp == 1;                       // (16 waves: the fourth group runs in the first one's order)
        if (kt_begin < kt_end) {
            tile_setup(kt_begin);
            load3(xs0, ws0, vm0);
            store3(0, xs0, ws0, vm0);
            if (kt_begin + 1 < kt_end) {
                tile_advance();
                load3(xs0, ws0, vm0);
            }
        }
#if SSD_C3_VARIANT & 8
        // variant 8: the PIXELS (streamed from HBM / the Infinity Cache: the long latency) are fetched TWO K tiles ahead into
        // two register sets, the weight planes (L2-resident) stay one tile ahead.  Set A holds tile kt + 1 and set B tile
        // kt + 2 when an even iteration starts; the loop is unrolled by two so that both sets are named statically.
        {
            f32x4 xsB[XP];
            unsigned vmB = 0;
            auto loadX = [&](f32x4 (&X)[XP], unsigned& vm) {
                vm = 0;
#pragma unroll
                for (int ps = 0; ps < XP; ++ps) {
                    const bool ok = x_is_valid(ps);
                    vm |= (ok ? 1u : 0u) << ps;
                    if (GEMM1X1) {
                        X[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (ok) X[ps] = *reinterpret_cast<const f32x4*>(xrow1[ps] + l_k0);
                    } else {
                        X[ps] = *reinterpret_cast<const f32x4*>(xbase + (unsigned)(ok ? xoff[ps] + l_xtile : 0));
                    }
                }
            };
            auto loadW = [&](bf16x8 (&W)[WP3], int k0) {
#pragma unroll
                for (int ps = 0; ps < WP3; ++ps) W[ps] = *reinterpret_cast<const bf16x8*>(w3base + w3off[ps] + (long)k0 * p.Npad);
            };
            // state here: tile kt_begin stored in stage 0; (xs0, ws0) hold tile kt_begin + 1 (if any), l_* describe it
            int k0_next = l_k0;                          // k offset of the tile whose WEIGHTS are loaded next (tile kt + 2)
            if (kt_begin + 2 < kt_end) {
                tile_advance();
                k0_next = l_k0;
                loadX(xsB, vmB);                         // pixels of tile kt_begin + 2
            }
            __syncthreads();
            auto step = [&](int kt, f32x4 (&XA)[XP], unsigned& vmA) {      // XA holds tile kt + 1; refilled with tile kt + 3
                const int stage = (kt - kt_begin) & 1;
                if (!late) mma_tile(stage);
                if (kt + 1 < kt_end) store3(stage ^ 1, XA, ws0, vmA);
                if (kt + 2 < kt_end) loadW(ws0, k0_next);                  // weights of tile kt + 2
                if (kt + 3 < kt_end) {
                    tile_advance();
                    k0_next = l_k0;
                    loadX(XA, vmA);                                        // pixels of tile kt + 3
                }
                if (late) mma_tile(stage);
                __syncthreads();
            };
            for (int kt = kt_begin; kt < kt_end; kt += 2) {
                step(kt, xs0, vm0);
                if (kt + 1 < kt_end) step(kt + 1, xsB, vmB);
            }
        }
        if (false)
#else
        __syncthreads();
#endif
#if SSD_C3_VARIANT & 2
        // variant 2 (4-wave tiles): ONE LDS stage, two barriers per K tile -- half the LDS, so two or three workgroups
        // share a CU and fill each other's barrier / staging phases (instead of the in-workgroup double buffer)
        if constexpr (NTHR == 256) {
            for (int kt = kt_begin; kt < kt_end; ++kt) {
                mma_tile(0);
                __syncthreads();
                if (kt + 1 < kt_end) store3(0, xs0, ws0, vm0);
                if (kt + 2 < kt_end) {
                    tile_advance();
                    load3(xs0, ws0, vm0);
                }
                __syncthreads();
            }
        } else
#endif
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int stage = (kt - kt_begin) & 1;
            if (!late && !mid) { mma_tile(stage); C3T(0); }
            if (kt + 1 < kt_end) store3(stage ^ 1, xs0, ws0, vm0);
            C3T(1);
            if (NTHR >= 768 && mid) mma_tile(stage);
            if (kt + 2 < kt_end) {
                tile_advance();
                load3(xs0, ws0, vm0);          // in flight across the barrier and the next tile's MFMAs
            }
            C3T(2);
            if (late) { mma_tile(stage); C3T(3); }
            __syncthreads();
            C3T(4);
        }
    } else {
    if (kt_begin < kt_end) {
        tile_setup(kt_begin);
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int stage = (kt - kt_begin) & 1;
        const float* Xs = smem + stage * STAGE_FLOATS;
        const float* Ws = Xs + BM * LDK;
        const bool more = kt + 1 < kt_end;
        if (more) {
            tile_advance();
            load_tile();
        }
        {
#pragma unroll
        for (int kc = 0; kc < ((SSD_CONV_ABLATE & 4) ? 0 : BK / 16); ++kc) {
            f32x4 a[NT], b[MT];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                a[ni] = *reinterpret_cast<const f32x4*>(Ws + ((wn * NT + ni) * 16 + frow) * LDK + fcol[kc]);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                b[mi] = *reinterpret_cast<const f32x4*>(Xs + ((wm * MT + mi) * 16 + frow) * LDK + fcol[kc]);
            if (SSD_CONV_ABLATE & 16) {
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) asm volatile("" ::"v"(a[ni]));
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) asm volatile("" ::"v"(b[mi]));
                continue;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ni][s], b[mi][s], acc[mi][ni], 0, 0, 0);
        }
        }
        if (more) store_tile(stage ^ 1);
        if (SSD_CONV_ABLATE & 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier
        else if (!(SSD_CONV_ABLATE & 8)) __syncthreads();
    }
    }

    conv_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, lane, HoWo);
#ifdef SSD_C3_PROF
    if constexpr (SPLIT3) {
        __syncthreads();
        if (blockIdx.x == gridDim.x / 3 && lane == 0 && p.split_k <= 1) {
            float* d = p.out + m0 * p.out_pixel_stride + n0 + wave * 12;
            for (int i = 0; i < 9; ++i) d[i] = (float)c3tp[i];
            d[9] = (float)(kt_end - kt_begin);
        }
    }
#endif
}


template <int MT, int NT, int WM, int WN, int BK, bool GEMM1X1>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr bool SWZ = BK >= 32;
    constexpr int LDK = SWZ ? BK : BK + 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * 16 * (MT * WM + NT * WN) * LDK];
    conv_mfma_body<MT, NT, WM, WN, BK, GEMM1X1, 0>(p, smem);
}

// split-bf16 variant: dynamic LDS, 2 * 3 * 16 * (MT*WM + NT*WN) * 64 bytes
#if SSD_C3_VARIANT & 2
#define SSD_C3_BOUNDS(WM, WN) __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? ((SSD_C3_VARIANT & 4) ? 3 : 2) : 1)
#else
#define SSD_C3_BOUNDS(WM, WN) __launch_bounds__(64 * WM * WN)
#endif
template <int MT, int NT, int WM, int WN, bool GEMM1X1>
__global__ SSD_C3_BOUNDS(WM, WN) void conv_mfma3_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem3[];
    conv_mfma_body<MT, NT, WM, WN, 32, GEMM1X1, 3>(p, smem3);
}
// bf16 variant (the net's precision-1 mode): one plane per operand, 2 * 16 * (MT*WM + NT*WN) * 64 bytes of LDS
template <int MT, int NT, int WM, int WN, bool GEMM1X1>
__global__ __launch_bounds__(64 * WM * WN) void conv_bf16_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem1[];
    conv_mfma_body<MT, NT, WM, WN, 32, GEMM1X1, 1>(p, smem1);
}

}  // namespace ssd

// Implicit-GEMM convolution whose operands reach LDS WITHOUT passing through registers (round 5).
//
// conv_mfma3_kernel (ssd_conv3.hip) loads fp32 pixels into VGPRs, splits them into three bf16 planes with ~100 VALU
// instructions per thread and K tile and writes 72 KB per K tile through ds_write (~80 B/clk/CU): its time is its matrix
// time PLUS that staging (profiles/HISTORY.md, round 4: MFMA-only skeleton 0.78 of the split-bf16 peak, the kernel 0.44 -
// 0.58).  Here the activation arrives PRE-SPLIT -- bf16 planes [np][Cin / 32][B*H*W][32] (slice-major: ssd_bf16x3.h) written by its producer's epilogue
// (store_planes4) or by split_planes_kernel -- next to the weights' planes (pack_split_kernel), and both operands are
// copied global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA: 1 KB = 16 tile rows x 64 bytes per wave instruction, no
// VGPR, no ds_write, no split):
//   * LDS image per (k-step, plane): rows of 32 bf16 = 64 bytes, the 16-byte quad q of row r at slot q ^ ((r >> 1) & 3)
//     (the fragment reads' conflict-free swizzle of ssd_conv_mfma.h).  The DMA writes lane i at base + 16 i, i.e. row
//     i >> 2, slot i & 3 -- so lane i FETCHES quad (i & 3) ^ ((i >> 3) & 3) of its row: the swizzle lives in the source
//     address, the destination stays lane-linear (cdna_hip_programming.md rule 21);
//   * padded taps / rows beyond M: the lane's buffer offset is 2^31, beyond the resource's range, and the hardware
//     deposits ZEROS in its LDS slot (tests/micro/lds_dma_probe.hip) -- no branch, no select on data;
//   * a wave owns whole 16-row blocks of the tile (all planes and k-steps of a block share one address register and one
//     tap-validity mask); two LDS stages, the next tile's DMA is issued before the current tile's MFMAs and waited for
//     (vmcnt(0)) at the one barrier per tile;
//   * np = 3 (fp32 nets): six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block on the exact three-way split, fp32 results
//     (ssd_bf16x3.h); np = 1 (the bf16 mode): bf16 STORAGE of the activation, one MFMA per block, K = 64 per barrier.
// Epilogue = conv_epilogue (ssd_conv_mfma.h): BN / bias, activation, residual, head routing, optional plane output.
// config ids: behind the bf16 tiles (ssd_conv.hip): "dma3_*" (np = 3) and "dmab_*" (np = 1).
#include "ssd_conv_mfma.h"

namespace ssd {

typedef __attribute__((address_space(3))) void* lds_dst_t;

// (the body is a __device__ function: a __global__ body that declares __amdgpu_buffer_rsrc_t objects loses its host stub)
template <int MT, int NT, int WM, int WN, int NP, bool GEMM1X1>
__device__ __forceinline__ void conv_dma_body(const ConvParams& p, char* __restrict__ dsm) {
    constexpr int KS = NP == 1 ? 2 : 1;          // 32-wide k-steps per LDS stage
    constexpr int WPL = NP == 1 ? 3 : 0;         // first weight plane: [h, m, l, r]
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN, NW = WM * WN;
    constexpr int RBX = BM / 16, RBW = BN / 16;                          // 16-row blocks of the two operands
    constexpr int RPX = (RBX + NW - 1) / NW, RPW = (RBW + NW - 1) / NW;   // ... per wave
    constexpr int XBYTES = KS * NP * BM * 64, STAGE = KS * NP * (BM + BN) * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nb_n = (p.Cout + BN - 1) / BN;
    const int mblk = blockIdx.x / nb_n, nblk = blockIdx.x - mblk * nb_n;
    const long m0 = (long)mblk * BM;
    const int n0 = nblk * BN;
    const int HoWo = p.Ho * p.Wo;
    const int ntaps = p.kh * p.kw;

    // ---- per-lane DMA sources: lane i of a 16-row block fetches row i >> 2, quad (i & 3) ^ ((i >> 3) & 3)
    const int r16 = lane >> 2;
    const int q8 = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;
    const int b_first = (int)(m0 / HoWo);
    // slice-major planes: pixel g, channel c at ((c / 32) * P + g) * 32 + c % 32, P = B H W.  The resource starts at the tile's
    // first pixel (1x1) / first image of slice 0; a K tile's slice goes into the scalar offset (P * 64 bytes per slice)
    const long npix = (long)p.B * p.H * p.W;
    const long xbase_e = (GEMM1X1 ? m0 : (long)b_first * p.H * p.W) * 32;        // elements into a plane
    const long xrange = min((long)0x7fffffffL, (npix * p.Cin - xbase_e) * 2);
    const int slice_b = (int)(npix * 64);                                          // bytes between two 32-channel slices
    __amdgpu_buffer_rsrc_t xrs[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        xrs[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.xp + pl * p.xp_plane + xbase_e), 0, (int)xrange, 0x00020000);
    const long wplane_b = (long)p.Npad * p.Kpad * 2;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.w3), 0, (int)(4 * wplane_b), 0x00020000);

    int xoff[RPX];
    unsigned xvalid[RPX];
#pragma unroll
    for (int j = 0; j < RPX; ++j) {
        const int rb = wave + j * NW;
        const long m = m0 + rb * 16 + r16;
        xoff[j] = 0;
        xvalid[j] = 0;
        if (rb < RBX && m < p.M) {
            if (GEMM1X1) {
                xoff[j] = ((rb * 16 + r16) * 32 + q8) * 2;
                xvalid[j] = 1u;
            } else {
                const int b = (int)(m / HoWo);
                const int pix = (int)(m - (long)b * HoWo);
                const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
                xoff[j] = ((((b - b_first) * p.H + iy0) * p.W + ix0) * 32 + q8) * 2;
                for (int ky = 0; ky < p.kh; ++ky)
                    for (int kx = 0; kx < p.kw; ++kx) {
                        const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
                        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) xvalid[j] |= 1u << (ky * p.kw + kx);
                    }
            }
        }
    }
    int woff[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int rb = wave + j * NW;
        const int r = min(rb * 16 + r16, min(BN, p.Npad - n0) - 1);          // clamped: rows past the tile / Npad are never used
        woff[j] = ((n0 + r) * 32 + q8) * 2;                                  // weights: [Kpad / 32][Npad][32] per plane
    }

    // ---- K walk: tiles of KS k-steps; general path channel-slice-major with the taps innermost (ssd_conv_mfma.h)
    const int nks = p.K / 32;                                       // K % 32 == 0 (host check)
    const int ntiles = GEMM1X1 ? (nks + KS - 1) / KS : ntaps * (p.Cin / (32 * KS));
    int kt_begin = 0, kt_end = ntiles;
    if (p.split_k > 1) {
        const int per = (ntiles + p.split_k - 1) / p.split_k;
        kt_begin = blockIdx.y * per;
        kt_end = min(ntiles, kt_begin + per);
    }
    int l_k0 = 0, l_tap = 0, l_ci = 0, l_ky = 0, l_kx = 0, l_xtile = 0, l_xs = 0;
    auto tile_setup = [&](int kt) {
        if (GEMM1X1) {
            l_k0 = kt * 32 * KS;
            l_xtile = 0;
            l_xs = kt * KS * slice_b;
        } else {
            const int cs = kt / ntaps;
            l_tap = kt - cs * ntaps;
            l_ci = cs * 32 * KS;
            l_ky = l_tap / p.kw;
            l_kx = l_tap - l_ky * p.kw;
            l_k0 = l_tap * p.Cin + l_ci;
            l_xtile = (l_ky * p.dil * p.W + l_kx * p.dil) * 64;
            l_xs = (l_ci >> 5) * slice_b;
        }
    };
    auto tile_advance = [&]() {
        if (GEMM1X1) {
            l_k0 += 32 * KS;
            l_xs += KS * slice_b;
        } else {
            ++l_tap;
            if (++l_kx == p.kw) { l_kx = 0; ++l_ky; }
            if (l_tap == ntaps) { l_tap = 0; l_ky = 0; l_kx = 0; l_ci += 32 * KS; }
            l_k0 = l_tap * p.Cin + l_ci;
            l_xtile = (l_ky * p.dil * p.W + l_kx * p.dil) * 64;
            l_xs = (l_ci >> 5) * slice_b;
        }
    };
    // k-steps of the tile at l_k0 that exist (1x1 path with KS = 2: the last tile may hold one)
    auto steps_here = [&]() -> int { return KS == 1 ? 1 : min(KS, nks - l_k0 / 32); };

    auto issue = [&](int stage) {              // DMA of the tile described by the l_* state into `stage`
        char* sb = dsm + stage * STAGE;
        const int ns = steps_here();
#pragma unroll
        for (int j = 0; j < RPX; ++j) {
            const int rb = wave + j * NW;
            if (RBX % NW != 0 && rb >= RBX) continue;
            const bool ok = GEMM1X1 ? (xvalid[j] != 0) : (((xvalid[j] >> l_tap) & 1u) != 0);
            const int vo = ok ? xoff[j] + l_xtile : (int)0x80000000;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (KS > 1 && s >= ns) break;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs[pl], (lds_dst_t)(sb + ((s * NP + pl) * BM + rb * 16) * 64), 16, vo, l_xs + s * slice_b, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int rb = wave + j * NW;
            if (RBW % NW != 0 && rb >= RBW) continue;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (KS > 1 && s >= ns) break;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_dst_t)(sb + XBYTES + ((s * NP + pl) * BN + rb * 16) * 64), 16, woff[j],
                                                             (int)((pl + WPL) * wplane_b) + ((l_k0 >> 5) + s) * p.Npad * 64, 0, 0);
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fq = ((lane >> 4) ^ ((frow >> 1) & 3)) << 4;
    auto mma_tile = [&](int stage, int ns) {
        const char* sb = dsm + stage * STAGE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (KS > 1 && s >= ns) break;
            const char* Xb = sb + s * NP * BM * 64;
            const char* Wb = sb + XBYTES + s * NP * BN * 64;
            BP<NP> b[MT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const char* r = Xb + ((wm * MT + mi) * 16 + frow) * 64 + fq;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) b[mi].p[pl] = *reinterpret_cast<const bf16x8*>(r + pl * BM * 64);
            }
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const char* r = Wb + ((wn * NT + ni) * 16 + frow) * 64 + fq;
                BP<NP> a;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) a.p[pl] = *reinterpret_cast<const bf16x8*>(r + pl * BN * 64);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = mmaN<NP>(a, b[mi], acc[mi][ni]);
            }
        }
    };

    int ns_cur = 1;
    if (kt_begin < kt_end) {
        tile_setup(kt_begin);
        ns_cur = steps_here();
        issue(0);
    }
    // the first tile's LDS-DMA has to have LANDED before any wave reads it: the copies count on vmcnt, and nothing in the
    // workgroup-barrier contract makes hipcc drain vmcnt for them (ROCm 7.2 happens to) -- wait explicitly, as the stem does
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int stage = (kt - kt_begin) & 1;
        int ns_next = 1;
        if (kt + 1 < kt_end) {
            tile_advance();
            ns_next = steps_here();
            issue(stage ^ 1);        // lands while this tile is multiplied; the stage was last read before the previous barrier
        }
        mma_tile(stage, ns_cur);
        ns_cur = ns_next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's copies, before the barrier that publishes them
        __syncthreads();
    }
    conv_epilogue<MT, NT>(p, acc, m0, n0, wm, wn, lane, HoWo);
}

template <int MT, int NT, int WM, int WN, int NP, bool GEMM1X1>
__global__ __launch_bounds__(64 * WM * WN) void conv_dma_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(1024))) char dsm_dma[];
    conv_dma_body<MT, NT, WM, WN, NP, GEMM1X1>(p, dsm_dma);
}

namespace {

typedef void (*convd_kernel_t)(const ConvParams);
struct ConvDCfg {
    const char* name3;
    const char* name1;
    int BM, BN, threads;
    convd_kernel_t gemm3, general3, gemm1, general1;
};
#define DCFG(MT, NT, WM, WN)                                                                                  \
    {"dma3_" #MT "x" #NT "_" #WM "x" #WN, "dmab_" #MT "x" #NT "_" #WM "x" #WN, 16 * MT * WM, 16 * NT * WN, 64 * WM * WN, \
     conv_dma_kernel<MT, NT, WM, WN, 3, true>, conv_dma_kernel<MT, NT, WM, WN, 3, false>,                       \
     conv_dma_kernel<MT, NT, WM, WN, 1, true>, conv_dma_kernel<MT, NT, WM, WN, 1, false>}
const ConvDCfg kCfgD[] = {
    DCFG(4, 4, 4, 2),    // 256 x 128, 8 waves
    DCFG(2, 4, 4, 2),    // 128 x 128, 8 waves
    DCFG(4, 4, 2, 4),    // 128 x 256, 8 waves
    DCFG(4, 2, 4, 2),    // 256 x 64, 8 waves
    DCFG(2, 7, 8, 1),    // 256 x 112, 8 waves (fused heads, A (L + 4) = 100)
    DCFG(4, 5, 4, 2),    // 256 x 160, 8 waves (fused heads, 150)
    DCFG(4, 4, 2, 2),    // 128 x 128, 4 waves
    DCFG(2, 7, 4, 1),    // 128 x 112, 4 waves
    DCFG(2, 5, 2, 2),    // 64 x 160, 4 waves
    DCFG(2, 4, 6, 2),    // 192 x 128, 12 waves
    DCFG(3, 4, 4, 3),    // 192 x 192, 12 waves
    DCFG(2, 4, 8, 2),    // 256 x 128, 16 waves
    DCFG(2, 2, 2, 2),    // 64 x 64, 4 waves
    DCFG(2, 5, 6, 2),    // 192 x 160, 12 waves (round 6: the 150-column heads on one tile column)
};
constexpr int kNumCfgD = sizeof(kCfgD) / sizeof(kCfgD[0]);

bool cd_gemm1x1(const ConvParams& p) {
    return p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
}

// fp32 activation -> bf16 planes (np = 3: exact split; np = 1: bf16 rounding); the fallback producer of a tensor whose
// own producing kernel has no plane epilogue
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, const long n4, const int C, const int np,
                                                           short* __restrict__ planes, const long plane) {
    const long P = n4 * 4 / C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256)
        store_planes4_lin(planes, plane, np, e * 4, C, P, *reinterpret_cast<const f32x4*>(x + e * 4));
}

// bf16 planes -> fp32 (h + m + l is exact; np = 1: the rounded value): debug fetches / tests
__global__ __launch_bounds__(256) void join_planes_kernel(const short* __restrict__ planes, const long n, const int C, const int np,
                                                          const long plane, float* __restrict__ x) {
    const long P = n / C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long pix = e / C;
        const long s = plane_elem(pix, (int)(e - pix * C), P);
        float v = __uint_as_float((unsigned)(unsigned short)planes[s] << 16);
        if (np == 3) {
            v += __uint_as_float((unsigned)(unsigned short)planes[plane + s] << 16);
            v += __uint_as_float((unsigned)(unsigned short)planes[2 * plane + s] << 16);
        }
        x[e] = v;
    }
}

}  // namespace

int dma_num_configs() { return kNumCfgD; }
const char* dma_config_name(int i, int np) { return (i >= 0 && i < kNumCfgD) ? (np == 1 ? kCfgD[i].name1 : kCfgD[i].name3) : "?"; }
bool dma_config_valid(int i, int np, const ConvParams& p) {
    if (i < 0 || i >= kNumCfgD) return false;
    if (!p.xp || !p.w3 || p.xp_np != np) return false;
    if (((uintptr_t)p.xp & 15) || ((uintptr_t)p.w3 & 15) || (p.xp_plane & 7)) return false;
    if (p.M > 0x7fffffffL - 1024) return false;
    if (4 * (long)p.Npad * p.Kpad * 2 > 0x7fffffffL) return false;
    // slice-major planes: a K tile's channel slice is a scalar offset of up to the whole plane (31 bits)
    if ((long)p.B * p.H * p.W * p.Cin * 2 > 0x7fffffffL) return false;
    if (cd_gemm1x1(p)) return p.Cin % 32 == 0;
    if (p.kh * p.kw > 32) return false;
    // a tile's rows span at most BM / (Ho Wo) + 2 images: their per-lane offsets (64 bytes per pixel) must fit 31 bits
    if (((long)kCfgD[i].BM / (p.Ho * p.Wo) + 3) * p.H * p.W * 64 > 0x3fffffffL) return false;
    return p.Cin % (np == 1 ? 64 : 32) == 0;
}
long dma_grid_blocks(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumCfgD) return 0;
    return ((p.M + kCfgD[i].BM - 1) / kCfgD[i].BM) * ((p.Cout + kCfgD[i].BN - 1) / kCfgD[i].BN);
}
int dma_k_tiles(int np, const ConvParams& p) { return (p.K + 31) / 32 / (np == 1 ? 2 : 1); }
void dma_tile(int i, int* BM, int* BN) {
    *BM = (i >= 0 && i < kNumCfgD) ? kCfgD[i].BM : 0;
    *BN = (i >= 0 && i < kNumCfgD) ? kCfgD[i].BN : 0;
}

int dma_launch(const ConvParams& p, int i, int np, hipStream_t st) {
    if (!dma_config_valid(i, np, p)) {
        set_error("conv2d: LDS-DMA config %d cannot run Cin=%d k=%dx%d (input planes %s)", i, p.Cin, p.kh, p.kw, p.xp ? "present" : "missing");
        return SSD_E_UNSUPPORTED;
    }
    const long blocks = dma_grid_blocks(i, p);
    SSD_UNSUPPORTED_IF(blocks > 0x7fffffffL, "conv2d: grid too large");
    const ConvDCfg& g = kCfgD[i];
    const convd_kernel_t fn = np == 1 ? (cd_gemm1x1(p) ? g.gemm1 : g.general1) : (cd_gemm1x1(p) ? g.gemm3 : g.general3);
    const int lds = 2 * (np == 1 ? 2 : 3) * (g.BM + g.BN) * 64;
    if (lds > 64 * 1024) SSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((unsigned)blocks, p.split_k > 1 ? p.split_k : 1);
    hipLaunchKernelGGL(fn, grid, dim3(g.threads), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_split_planes(const float* x, long n, int C, int np, short* planes, long plane, hipStream_t st) {
    SSD_CHECK_ARG(n % 4 == 0 && (np == 1 || np == 3), "split_planes: element count %ld must be a multiple of 4, planes 1 or 3", n);
    SSD_CHECK_ARG(C > 0 && C % 32 == 0 && n % C == 0, "split_planes: %d channels (slice-major planes need a multiple of 32 that divides the %ld elements)", C, n);
    if (n == 0) return SSD_OK;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, st, x, n4, C, np, planes, plane);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_join_planes(const short* planes, long n, int C, int np, long plane, float* x, hipStream_t st) {
    if (n == 0) return SSD_OK;
    SSD_CHECK_ARG(C > 0 && C % 32 == 0 && n % C == 0, "join_planes: %d channels do not divide the %ld elements into whole 32-channel slices", C, n);
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(join_planes_kernel, dim3(blocks), dim3(256), 0, st, planes, n, C, np, plane, x);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

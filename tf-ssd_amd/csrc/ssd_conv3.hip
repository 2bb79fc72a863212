// Implicit-GEMM convolution on the BF16 matrix cores at FP32 accuracy: the tiles of conv_mfma_kernel
// (ssd_conv_mfma.h: same loaders, K walk, epilogue) with both operands split exactly into three bf16 planes on their
// way into LDS and six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block (ssd_bf16x3.h) -- 2.67x the throughput of
// v_mfma_f32_16x16x4_f32 and, unlike it, overlapping with the vector ALU.  The wave tiles are 2x..4x wide in BOTH
// dimensions: a split fragment is 48 bytes per lane for 96 matrix cycles, so it has to be reused from registers
// (3 (1/MT + 1/NT) LDS reads per six-pack; at MT = 1 -- or under Winograd's 8 accumulator banks -- LDS binds instead).
// config ids: behind the skinny tiles (ssd_conv.hip).
#include "ssd_conv_mfma.h"

namespace ssd {

namespace {

typedef void (*conv3_kernel_t)(const ConvParams);
struct Conv3Cfg {
    const char* name;
    int BM, BN, threads;
    conv3_kernel_t gemm, general;
    const char* name1;                  // the same tile as a bf16 (one-product) kernel: the net's precision-1 mode
    conv3_kernel_t gemm1, general1;
};
#define C3CFG(MT, NT, WM, WN)                                                           \
    {"mfma3_" #MT "x" #NT "_" #WM "x" #WN, 16 * MT * WM, 16 * NT * WN, 64 * WM * WN,      \
     conv_mfma3_kernel<MT, NT, WM, WN, true>, conv_mfma3_kernel<MT, NT, WM, WN, false>,  \
     "bf16_" #MT "x" #NT "_" #WM "x" #WN, conv_bf16_kernel<MT, NT, WM, WN, true>, conv_bf16_kernel<MT, NT, WM, WN, false>}
const Conv3Cfg kCfg3[] = {
    C3CFG(4, 4, 2, 2),    // 128 x 128
    C3CFG(2, 4, 2, 2),    // 64 x 128
    C3CFG(4, 2, 2, 2),    // 128 x 64
    C3CFG(2, 2, 2, 2),    // 64 x 64
    C3CFG(2, 7, 4, 1),    // 128 x 112 (fused heads, A*(L+4) = 100)
    C3CFG(2, 5, 2, 2),    // 64 x 160  (fused heads, 150)
    C3CFG(4, 5, 2, 2),    // 128 x 160
    C3CFG(2, 4, 4, 1),    // 128 x 64
    C3CFG(1, 4, 2, 2),    // 32 x 128
    C3CFG(2, 2, 4, 1),    // 128 x 32
    C3CFG(4, 3, 2, 2),    // 128 x 96
    C3CFG(2, 3, 2, 2),    // 64 x 96
    // 8 waves: twice the rows per staged weight tile (the loop is bound by the L2 -> CU bytes per tile)
    C3CFG(4, 4, 4, 2),    // 256 x 128
    C3CFG(4, 4, 2, 4),    // 128 x 256
    C3CFG(2, 4, 4, 2),    // 128 x 128
    C3CFG(4, 2, 4, 2),    // 256 x 64
    C3CFG(2, 7, 8, 1),    // 256 x 112
    C3CFG(4, 5, 4, 2),    // 256 x 160
    // 12 / 16 waves (three / four per SIMD; the middle group runs its MFMAs between its stores and its loads): no faster per
    // K tile than the 8-wave tiles (profiles/HISTORY.md, round 4), but 192-row tiles fit some grids in fewer rounds of workgroups
    // (head level 1 at B = 64: 121 tiles x split-K 2 = 242 workgroups, 184 -> 151 us; VGG fc7 158 -> 137 us)
    C3CFG(2, 4, 6, 2),    // 192 x 128, 12 waves
    C3CFG(3, 4, 4, 3),    // 192 x 192, 12 waves
    C3CFG(2, 4, 8, 2),    // 256 x 128, 16 waves (118 registers)
    // round 6: the 150-column heads (A (L + 4) = 6 x 25) on ONE 160-wide tile column instead of two 128-wide ones (41 % of
    // whose matrix work is padding), at 192 rows: head level 1 at B = 64 = 121 tiles x split-K 2 = 242 workgroups
    C3CFG(2, 5, 6, 2),    // 192 x 160, 12 waves
};
constexpr int kNumCfg3 = sizeof(kCfg3) / sizeof(kCfg3[0]);
int c3_lds_bytes(const Conv3Cfg& g, int planes) {
#if SSD_C3_VARIANT & 2
    if (g.threads == 256) return planes * (g.BM + g.BN) * 64;       // experiment: one LDS stage for the 4-wave tiles
#endif
    return 2 * planes * (g.BM + g.BN) * 64;
}

// four bf16 planes behind the packed fp32 weights: h, m, l (the exact split, x = h + m + l) and r = the bf16 rounding of
// x (round to nearest even) for the one-product bf16 kernels
// (plane layout [Kpad / 32][Npad][32], ssd_bf16x3.h: a tile's 16 rows x 32 k are 1 KB of contiguous memory)
// (row_major: plain [Npad][Kpad] planes -- what the whole-image block kernels stage themselves, ssd_imgblock.hip)
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, const long total, const int Kpad, const int Npad,
                                                         const int row_major, short* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long n = e / Kpad;
        const long d = row_major ? e : plane_elem(n, (int)(e - n * Kpad), Npad);
        short h, m, l;
        split1(w[e], h, m, l);
        out[d] = h;
        out[total + d] = m;
        out[2 * total + d] = l;
        out[3 * total + d] = rne1(w[e]);
    }
}

bool c3_gemm1x1(const ConvParams& p) {
    return p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.Ho == p.H && p.Wo == p.W;
}

}  // namespace

int mfma3_num_configs() { return kNumCfg3; }
const char* mfma3_config_name(int i) { return (i >= 0 && i < kNumCfg3) ? kCfg3[i].name : "?"; }
bool mfma3_config_valid(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumCfg3) return false;
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w3 & 15) || !p.w3) return false;
    if (p.Cin % 4) return false;
    if (p.M > 0x7fffffffL - 1024) return false;
    if (c3_gemm1x1(p)) return true;
    if (p.kh * p.kw > 32) return false;
    return p.Cin % 32 == 0;
}
long mfma3_grid_blocks(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumCfg3) return 0;
    return ((p.M + kCfg3[i].BM - 1) / kCfg3[i].BM) * ((p.Cout + kCfg3[i].BN - 1) / kCfg3[i].BN);
}
int mfma3_k_tiles(const ConvParams& p) { return (p.K + 31) / 32; }
void mfma3_tile(int i, int* BM, int* BN) {
    *BM = (i >= 0 && i < kNumCfg3) ? kCfg3[i].BM : 0;
    *BN = (i >= 0 && i < kNumCfg3) ? kCfg3[i].BN : 0;
}

const char* bf16_config_name(int i) { return (i >= 0 && i < kNumCfg3) ? kCfg3[i].name1 : "?"; }

// the kernel only (the caller adds the split-K reduction); bf16: the one-product form of the same tile
int mfma3_launch(const ConvParams& p, int i, hipStream_t st, bool bf16) {
    if (!mfma3_config_valid(i, p)) {
        set_error("conv2d: split-bf16 / bf16 config %d cannot run Cin=%d k=%dx%d", i, p.Cin, p.kh, p.kw);
        return SSD_E_UNSUPPORTED;
    }
    const long blocks = mfma3_grid_blocks(i, p);
    SSD_UNSUPPORTED_IF(blocks > 0x7fffffffL, "conv2d: grid too large");
    const conv3_kernel_t fn = bf16 ? (c3_gemm1x1(p) ? kCfg3[i].gemm1 : kCfg3[i].general1)
                                   : (c3_gemm1x1(p) ? kCfg3[i].gemm : kCfg3[i].general);
    const int lds = c3_lds_bytes(kCfg3[i], bf16 ? 1 : 3);
    if (lds > 64 * 1024) SSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((unsigned)blocks, p.split_k > 1 ? p.split_k : 1);
    hipLaunchKernelGGL(fn, grid, dim3(kCfg3[i].threads), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

int launch_pack_split(float* packed, int K, int Cout, hipStream_t st, bool row_major) {
    const long total = (long)conv_kpad(K) * conv_npad(Cout);
    if (total == 0) return SSD_OK;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_split_kernel, dim3(blocks), dim3(256), 0, st, packed, total, conv_kpad(K), conv_npad(Cout), row_major ? 1 : 0,
                       const_cast<short*>(conv_split_planes(packed, K, Cout)));
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

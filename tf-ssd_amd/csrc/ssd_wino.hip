// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (gfx950): the 3x3 stride-1 dense
// convs of the SSD graphs -- the twelve head convs (reference models/header.py:60-61; 38 % of the
// MobileNetV2-SSD FLOPs) and VGG16's backbone (models/ssd_vgg16.py:52-72; 98 % of its FLOPs) --
// with 16 multiplications per 2x2 output tile and channel pair instead of 36 (2.25x fewer MFMAs):
//
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, arXiv:1509.09308)
//
//   d = 4x4 input patch of a tile (pad resolved by zero fill), g = 3x3 filter, (.) elementwise.
//   U[a][b][co][ci] = (G g G^T)[a][b] is precomputed once (wino_pack_kernel); per (a,b) the sum
//   over ci is a GEMM  M_ab[tile][co] = V_ab[tile][ci] * U_ab[ci][co]  on v_mfma_f32_16x16x4_f32.
//
// Fully fused, nothing but x, U and y touches HBM.  Loop nest chosen for registers and traffic:
//   for a in 0..3                         (row of the transformed tile)
//     for ci-slab of 16 channels          (K loop; optional split over blockIdx.y)
//        stage V_{a,0..3}: each (tile, channel quad) item loads the TWO patch rows B^T needs for
//           this a (8 x 16-byte loads), combines them (B^T d)[a][.] and applies the column
//           transform in registers -> 4 LDS rows; stage U_{a,0..3} slabs
//        4 x NT MFMA tiles x 4 k-steps into acc[b][.]
//     fold: r0 = M_a0 + M_a1 + M_a2, r1 = M_a1 - M_a2 - M_a3;  Y[dy][.] += A^T[dy][a] * r
// so 4 accumulator banks + 4 output banks are live (not 16), the input is re-read 4x (from L2)
// instead of 16x, and a slab costs 8 + ~8 global 16-byte loads per thread for 16*NT MFMAs per
// wave -- the same load/MFMA ratio as the direct kernel.  Weights are the MFMA A operand and tiles
// the B operand, so a lane owns 4 consecutive output channels of one tile: the epilogue applies
// scale/shift/activation and stores the tile's 4 pixels with 16-byte stores through the same
// (batch stride, pixel stride, n_split) routing as conv_mfma_kernel.
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wino_act(float v, int act) {
    if (act == SSD_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == SSD_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}

// U[a][b][row_off + n][ci] = sum_ij G[a][i] G[b][j] g[i][j][ci][n] for n < Cout (the caller zeroes U
// first: rows up to Npad stay zero; row_off places a second kernel behind the first -- the fused
// label + box head conv).  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ hwio, const int Cin, const int Cout,
                                                       const int Npad, const int row_off, float* __restrict__ U) {
    const long total = (long)Cout * Cin;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ci = (int)(e % Cin), n = (int)(e / Cin);
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = hwio[((long)(i * 3 + j) * Cin + ci) * Cout + n];
        float t[4][3];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) t[a][j] = G[a][0] * g[0][j] + G[a][1] * g[1][j] + G[a][2] * g[2][j];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                U[((long)(a * 4 + b) * Npad + row_off + n) * Cin + ci] = t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2];
    }
}

template <int NT, int WT, int WN>
__global__ __launch_bounds__(256, (NT <= 4 ? 2 : 1)) void conv_wino_kernel(const ConvParams p) {
    // LDS rows are UNPADDED 16-float slabs with the 16-byte quad index XOR-ed by (row >> 1) & 3: under
    // gfx950's b128 lane grouping this is conflict-free for the fragment reads AND the tile writes
    // (brute-forced over the four lane groups; padded 24-float rows were 2-way conflicted on the writes:
    // SQ_LDS_BANK_CONFLICT 5 % of the wave cycles) and takes a third less LDS
    constexpr int TT = 16 * WT, TN = 16 * NT * WN, LDK = 16;
    constexpr int XI = TT * 4, XP = (XI + 255) / 256;             // input items (tile, channel quad)
    constexpr int WU = 4 * TN * 4, WP = (WU + 255) / 256;         // U units: 4 b x TN rows x 4 quads
    static_assert(WT * WN == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float smem[4 * (TT + TN) * LDK];
    float* Vs = smem;                      // [4 b][TT][LDK]
    float* Us = smem + 4 * TT * LDK;       // [4 b][TN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WN, wn = wave % WN;
    const int TY = (p.Ho + 1) >> 1, TX = (p.Wo + 1) >> 1;
    const long Ttot = (long)p.B * TY * TX;
    const int nb_n = (p.Cout + TN - 1) / TN;
    const int tblk = blockIdx.x / nb_n, nblk = blockIdx.x - tblk * nb_n;
    const long t0 = (long)tblk * TT;
    const int n0 = nblk * TN;
    const float* U = p.wino_w;
    const long ustride = (long)p.Npad * p.Cin;        // one (a, b) matrix

    // ---- per-thread bookkeeping: input items
    long xbase[XP];            // float offset of patch (0,0), channel quad of this item (may be "negative": masked)
    unsigned xmask[XP];        // bit r*4+c: patch element inside the image
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int it = tid + ps * 256;
        const long t = t0 + (it >> 2);
        xmask[ps] = 0;
        xbase[ps] = 0;
        if (it < XI && t < Ttot) {
            const int tx = (int)(t % TX);
            const long r = t / TX;
            const int ty = (int)(r % TY), b = (int)(r / TY);
            const int iy0 = 2 * ty - p.pad_t, ix0 = 2 * tx - p.pad_l;
            xbase[ps] = (((long)b * p.H + iy0) * p.W + ix0) * p.Cin + (it & 3) * 4;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if ((unsigned)(iy0 + rr) < (unsigned)p.H && (unsigned)(ix0 + cc) < (unsigned)p.W) xmask[ps] |= 1u << (rr * 4 + cc);
        }
    }
    long woff[WP];
#pragma unroll
    for (int ps = 0; ps < WP; ++ps) {
        const int u = min(tid + ps * 256, WU - 1);
        const int bsel = u / (TN * 4), rem = u - bsel * (TN * 4);
        const int row = min(n0 + (rem >> 2), p.Npad - 1);          // clamped: rows past Npad are never used
        woff[ps] = (long)bsel * ustride + (long)row * p.Cin + (rem & 3) * 4;
    }

    const int nslab = p.Cin / 16;
    int s_begin = 0, s_end = nslab;
    if (p.split_k > 1) {
        const int per = (nslab + p.split_k - 1) / p.split_k;
        s_begin = blockIdx.y * per;
        s_end = min(nslab, s_begin + per);
    }

    f32x4 Y[2][2][NT];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) Y[dy][dx][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = ((lane >> 4) ^ ((frow >> 1) & 3)) << 2;      // swizzled fragment quad (tile rows are multiples of 16)
    const long rowstep = (long)p.W * p.Cin;

#pragma unroll 1
    for (int a = 0; a < 4; ++a) {
        // B^T row a of d: a0: d0 - d2, a1: d1 + d2, a2: d2 - d1, a3: d1 - d3
        const int r1 = a == 0 ? 0 : (a == 2 ? 2 : 1);
        const int r2 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
        const float sgn = a == 1 ? 1.0f : -1.0f;
        f32x4 acc[4][NT];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[b][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        f32x4 x1[XP][4], x2[XP][4], wr[WP];       // raw patch rows / U units of the slab in flight (combined at store time)
        auto load_slab = [&](int s) {
            const int ci0 = s * 16;
#pragma unroll
            for (int ps = 0; ps < XP; ++ps) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
                    if ((xmask[ps] >> (r1 * 4 + cc)) & 1u)
                        v1 = *reinterpret_cast<const f32x4*>(p.in + xbase[ps] + r1 * rowstep + (long)cc * p.Cin + ci0);
                    if ((xmask[ps] >> (r2 * 4 + cc)) & 1u)
                        v2 = *reinterpret_cast<const f32x4*>(p.in + xbase[ps] + r2 * rowstep + (long)cc * p.Cin + ci0);
                    x1[ps][cc] = v1;
                    x2[ps][cc] = v2;
                }
            }
#pragma unroll
            for (int ps = 0; ps < WP; ++ps)
                wr[ps] = *reinterpret_cast<const f32x4*>(U + (long)a * 4 * ustride + woff[ps] + ci0);
        };
        auto store_slab = [&]() {
#pragma unroll
            for (int ps = 0; ps < XP; ++ps) {
                const int it = tid + ps * 256;
                if (XI % 256 != 0 && it >= XI) continue;
                float* dst = Vs + (it >> 2) * LDK + (((it & 3) ^ ((it >> 3) & 3)) << 2);      // row = it >> 2: key (row >> 1) & 3
                f32x4 c[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) c[cc] = x1[ps][cc] + sgn * x2[ps][cc];
                // column transform (d B): b0: c0 - c2, b1: c1 + c2, b2: c2 - c1, b3: c1 - c3
                *reinterpret_cast<f32x4*>(dst + 0 * TT * LDK) = c[0] - c[2];
                *reinterpret_cast<f32x4*>(dst + 1 * TT * LDK) = c[1] + c[2];
                *reinterpret_cast<f32x4*>(dst + 2 * TT * LDK) = c[2] - c[1];
                *reinterpret_cast<f32x4*>(dst + 3 * TT * LDK) = c[1] - c[3];
            }
#pragma unroll
            for (int ps = 0; ps < WP; ++ps) {
                const int u = tid + ps * 256;
                if (WU % 256 != 0 && u >= WU) continue;
                const int bsel = u / (TN * 4), rem = u - bsel * (TN * 4);
                *reinterpret_cast<f32x4*>(Us + (bsel * TN + (rem >> 2)) * LDK + (((rem & 3) ^ ((rem >> 3) & 3)) << 2)) = wr[ps];
            }
        };

        if (s_begin < s_end) load_slab(s_begin);
        for (int s = s_begin; s < s_end; ++s) {
            __syncthreads();               // previous slab fully consumed
            store_slab();
            __syncthreads();
            if (s + 1 < s_end) load_slab(s + 1);       // in flight during the MFMAs
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 vb = *reinterpret_cast<const f32x4*>(Vs + (b * TT + wt * 16 + frow) * LDK + fk);
                f32x4 ua[NT];
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
                    ua[ni] = *reinterpret_cast<const f32x4*>(Us + (b * TN + (wn * NT + ni) * 16 + frow) * LDK + fk);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni)
                        acc[b][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[ni][k], vb[k], acc[b][ni], 0, 0, 0);
            }
        }
        // fold M_a. into Y (A^T M A): A^T = [[1,1,1,0],[0,1,-1,-1]]
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const f32x4 q0 = acc[0][ni] + acc[1][ni] + acc[2][ni];
            const f32x4 q1 = acc[1][ni] - acc[2][ni] - acc[3][ni];
            if (a <= 2) { Y[0][0][ni] += q0; Y[0][1][ni] += q1; }
            if (a == 1) { Y[1][0][ni] += q0; Y[1][1][ni] += q1; }
            if (a >= 2) { Y[1][0][ni] -= q0; Y[1][1][ni] -= q1; }
        }
    }

    // ---- epilogue: lane holds tile t = t0 + wt*16 + (lane & 15), channels n .. n+3
    const long t = t0 + wt * 16 + (lane & 15);
    if (t >= Ttot) return;
    const int tx = (int)(t % TX);
    const long rr = t / TX;
    const int ty = (int)(rr % TY), bb = (int)(rr / TY);
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int oy = 2 * ty + dy;
        if (oy >= p.Ho) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ox = 2 * tx + dx;
            if (ox >= p.Wo) continue;
            const int pix = oy * p.Wo + ox;
            if (p.split_k > 1) {            // raw partial sums; the epilogue runs in splitk_reduce_kernel
                float* prow = p.partial + ((long)blockIdx.y * p.M + (long)bb * HoWo + pix) * p.Cout;
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
                    if (n >= p.Cout) continue;
                    if (n + 3 < p.Cout && (p.Cout & 3) == 0) {
                        *reinterpret_cast<f32x4*>(prow + n) = Y[dy][dx][ni];
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (n + j < p.Cout) prow[n + j] = Y[dy][dx][ni][j];
                    }
                }
                continue;
            }
            float* orow = p.out + (long)bb * p.out_batch_stride + (long)pix * p.out_pixel_stride;
            float* orow2 = p.n_split ? p.out2 + (long)bb * p.out2_batch_stride + (long)pix * p.out2_pixel_stride - p.n_split : nullptr;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int n = n0 + (wn * NT + ni) * 16 + (lane >> 4) * 4;
                if (n >= p.Cout) continue;
                f32x4 v = Y[dy][dx][ni];
                const bool straddle = p.n_split && n < p.n_split && n + 3 >= p.n_split;
                if (n + 3 < p.Cout && !straddle) {
                    if (p.scale) v = v * *reinterpret_cast<const f32x4*>(p.scale + n);
                    if (p.shift) v = v + *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = wino_act(v[j], p.act);
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
                        float tv = v[j];
                        if (p.scale) tv = tv * p.scale[n + j];
                        if (p.shift) tv = tv + p.shift[n + j];
                        tv = wino_act(tv, p.act);
                        float* drow = (p.n_split && n + j >= p.n_split) ? orow2 : orow;
                        drow[n + j] = tv;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ config table
typedef void (*wino_kernel_t)(const ConvParams);
struct WinoCfg {
    const char* name;
    int TT, TN;
    wino_kernel_t fn;
};
#define WCFG(NT, WT, WN) {"wino_" #NT "x" #WT "x" #WN, 16 * WT, 16 * NT * WN, conv_wino_kernel<NT, WT, WN>}
static const WinoCfg kWino[] = {
    WCFG(7, 4, 1),    // 64 tiles x 112 channels (fused heads, A*(L+4) = 100)
    WCFG(5, 2, 2),    // 32 x 160 (fused heads, 150)
    WCFG(4, 2, 2),    // 32 x 128
    WCFG(4, 4, 1),    // 64 x 64
    WCFG(2, 2, 2),    // 32 x 64
    WCFG(8, 4, 1),    // 64 x 128
    WCFG(5, 4, 1),    // 64 x 80
    WCFG(3, 4, 1),    // 64 x 48
    WCFG(4, 1, 4),    // 16 x 256
    WCFG(2, 4, 1),    // 64 x 32
};
constexpr int kNumWino = sizeof(kWino) / sizeof(kWino[0]);

int wino_num_configs() { return kNumWino; }
const char* wino_config_name(int i) { return (i >= 0 && i < kNumWino) ? kWino[i].name : "?"; }

bool wino_applicable(const ConvParams& p) {
    return p.kh == 3 && p.kw == 3 && p.stride == 1 && p.dil == 1 && p.Cin % 16 == 0 && !p.residual &&
           (((uintptr_t)p.in & 15) == 0) && p.Ho >= 1 && p.Wo >= 1 && p.pad_t >= 0 && p.pad_t <= 2 && p.pad_l >= 0 &&
           p.pad_l <= 2 && p.M < 0x7fffffffL - 1024;
}
bool wino_config_valid(int i, const ConvParams& p) {
    return i >= 0 && i < kNumWino && p.wino_w != nullptr && (((uintptr_t)p.wino_w & 15) == 0) && wino_applicable(p);
}
long wino_grid_blocks(int i, const ConvParams& p) {
    if (i < 0 || i >= kNumWino) return 0;
    const long T = (long)p.B * ((p.Ho + 1) / 2) * ((p.Wo + 1) / 2);
    return ((T + kWino[i].TT - 1) / kWino[i].TT) * ((p.Cout + kWino[i].TN - 1) / kWino[i].TN);
}
int wino_k_tiles(const ConvParams& p) { return p.Cin / 16; }

int wino_launch(const ConvParams& p, int i, hipStream_t st) {
    if (!wino_config_valid(i, p)) {
        set_error("conv2d: Winograd config %d cannot run this convolution", i);
        return SSD_E_UNSUPPORTED;
    }
    const long blocks = wino_grid_blocks(i, p);
    SSD_UNSUPPORTED_IF(blocks > 0x7fffffffL, "conv2d: grid too large");
    dim3 grid((unsigned)blocks, p.split_k > 1 ? p.split_k : 1);
    hipLaunchKernelGGL(kWino[i].fn, grid, dim3(256), 0, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

size_t wino_weight_floats(int Cin, int Cout) { return (size_t)16 * conv_npad(Cout) * Cin; }

int launch_wino_pack(const float* hwio, int Cin, int Cout, int Npad, int row_off, float* U, hipStream_t st) {
    const long total = (long)Cout * Cin;
    if (total == 0) return SSD_OK;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, hwio, Cin, Cout, Npad, row_off, U);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

// Depthwise 3x3 + BN + ReLU6 -> project 1x1 + BN (+ residual) in one kernel, for the
// MobileNetV2 blocks whose expand conv runs as its own GEMM (blocks 7-16: Cin 64-160, the
// expanded map of block 13 is an SSD feature map and has to reach HBM anyway):
//
//     y = project_BN( relu6(dw_BN( dw3x3(E) )) * Wp ) [+ x]
//
// ([3P] keras-applications MobileNetV2 block_k_depthwise / _depthwise_BN / _depthwise_relu /
// _project / _project_BN / _add, SURVEY.md Appendix A.)  Run as two kernels the depthwise
// output D (as large as E) is written and re-read, and both kernels are latency-bound at the
// 19x19 / 10x10 resolutions (23 104 / 6 400 pixels per 64-image batch: ~90 / 25 pixels per CU).
// Here one workgroup owns TH x TW output pixels of one image (full-width row bands: halo only
// above and below) and walks the expanded channels in chunks of 48:
//
//   A  stage     E halo tile [IH*IW px][48] -> LDS (zeros outside the image: the depthwise pads E),
//                project weight chunk [NTB*16][48] and depthwise taps/shift [10][48] -> LDS
//   B  depthw.   D[out px][48] = relu6(sum_taps E * Wd + shift)   VALU from LDS, a thread owns a
//                strip of SL consecutive output pixels x 4 channels (sliding window)
//   C  project   acc[out px][Cout] += D[out px][48] * Wp[48][Cout]  fp32 MFMA, accumulators in registers
//
// BN scales are folded into Wd / Wp on the host side of the library (ssd_net.hip), so the
// kernel only adds the shifts.  With few pixels per image the grid can additionally split the
// output channels (gridDim.y): each half recomputes the (cheap, VALU) depthwise.
#include <cstdlib>

#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCK = 48;           // expanded channels per chunk (divides 384, 576, 960)
constexpr int kCQ = kCK / 4;      // channel quads per chunk
// LDS row stride (floats) of the E / D / Wp tiles: 14 quads make the b128 MFMA fragment reads
// conflict-free under gfx950's lane grouping (13 quads are 2-way conflicted); the stride-2
// shape keeps 13 so that its double-buffered tiles still fit the 160 KB of LDS.
constexpr int ld_for(int stride) { return stride == 2 ? kCK + 4 : kCK + 8; }

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.0f), 6.0f); }

// LDS-only workgroup barrier and compiler-invisible prefetch loads: the same idiom as in
// ssd_fused.hip (see the comments there) -- the next chunk's E tile / weights stay in flight
// across the depthwise and MFMA phases; hipcc would drain them at the first barrier otherwise.
// Every issued load IS consumed (an unconsumed asm load's destination is dead to the compiler).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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

template <int S, int TH, int TW, int SL, int WM, int WN, int NTW>
struct DwProjShape {
    static constexpr int PT = TH * TW;                              // output pixels per tile
    static constexpr int PG = ((PT + 15) / 16 + WM - 1) / WM * WM;  // 16-pixel MFMA groups (multiple of WM)
    static constexpr int MTW = PG / WM;                             // pixel groups per wave
    static constexpr int NTB = NTW * WN;                            // 16-channel output tiles per workgroup
    static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    static constexpr int HP = IH * IW;                              // halo pixels
    static constexpr int NSX = (TW + SL - 1) / SL;                  // strips per output row
    static constexpr int STRIPS = TH * NSX;
    static constexpr int WIN = (SL - 1) * S + 3;                    // input columns a strip needs
    static constexpr int LD = ld_for(S);
    static constexpr size_t lds_floats = (size_t)HP * LD + (size_t)PG * 16 * LD + (size_t)NTB * 16 * LD + 10 * kCK;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(STRIPS * kCQ <= 256, "depthwise mapping needs STRIPS * 12 <= 256 threads");
};

template <int S, int TH, int TW, int SL, int WM, int WN, int NTW>
__global__ __launch_bounds__(256) void dwproj_kernel(const DwProjParams p) {
    using Sh = DwProjShape<S, TH, TW, SL, WM, WN, NTW>;
    constexpr int PT = Sh::PT, PG = Sh::PG, MTW = Sh::MTW, NTB = Sh::NTB, IW = Sh::IW, HP = Sh::HP, kLD = Sh::LD;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Es = sm;                                  // [HP][kLD]
    float* Ds = Es + HP * kLD;                       // [PG*16][kLD]
    float* Ws = Ds + PG * 16 * kLD;                  // [NTB*16][kLD]
    float* Wd = Ws + NTB * 16 * kLD;                 // [10][kCK]: 9 taps + shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int b = blockIdx.x / tiles_per_img;
    const int rem = blockIdx.x - b * tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
    const int nt0 = blockIdx.y * NTB;                // first 16-channel output tile of this workgroup
    const float* eb = p.e + (long)b * p.H * p.W * p.Ce;

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // depthwise mapping: thread = (strip of SL output pixels in one row, channel quad)
    const int cq = tid % kCQ, strip = tid / kCQ;
    const bool dw_on = strip < Sh::STRIPS;
    const int sr = strip / Sh::NSX, sx0 = (strip - sr * Sh::NSX) * SL;
    const int frow = lane & 15, fk = (lane >> 4) * 4;

    // ---- staging: chunk c+1 travels global -> registers while chunk c is computed.  Indices are
    // clamped, not predicated: every lane loads from a valid address; the LDS writes drop the
    // lanes that are out of range and zero the halo pixels outside the image.
    constexpr int E_U = HP * kCQ, W_U = NTB * 16 * kCQ;
    constexpr int E_R = (E_U + 255) / 256, W_R = (W_U + 255) / 256;
    f32x4 er[E_R], wr[W_R], dr[1];
    const float* eptr[E_R];     // address of this lane's E elements at channel 0 of the chunk
    bool ein[E_R];
#pragma unroll
    for (int i = 0; i < E_R; ++i) {
        const int u = min(tid + i * 256, E_U - 1);
        const int hp = u / kCQ, k4 = (u - hp * kCQ) * 4;
        const int r = hp / IW, cc = hp - r * IW;
        const int iy = iy0 + r, ix = ix0 + cc;
        ein[i] = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        eptr[i] = eb + ((long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix, 0), p.W - 1)) * p.Ce + k4;
    }
    const float* wptr[W_R];
#pragma unroll
    for (int i = 0; i < W_R; ++i) {
        const int u = min(tid + i * 256, W_U - 1);
        const int row = u / kCQ, k4 = (u - row * kCQ) * 4;
        wptr[i] = p.wp + (long)min(nt0 * 16 + row, p.npad_p - 1) * p.kpad_p + k4;
    }
    const int dt = min(tid, 10 * kCQ - 1) / kCQ, dk4 = (min(tid, 10 * kCQ - 1) % kCQ) * 4;
    const float* dptr = (dt < 9 ? p.wd + (long)dt * p.Ce : p.dh) + dk4;
    auto load_chunk = [&](int ch0) {
#pragma unroll
        for (int i = 0; i < E_R; ++i) er[i] = gload16_async(eptr[i] + ch0);
#pragma unroll
        for (int i = 0; i < W_R; ++i) wr[i] = gload16_async(wptr[i] + ch0);
        dr[0] = gload16_async(dptr + ch0);
    };
    auto store_chunk = [&]() {
        wait_prefetch(er);
        wait_prefetch(wr);
        wait_prefetch(dr);
#pragma unroll
        for (int i = 0; i < E_R; ++i) {
            const int u = tid + i * 256;
            const int hp = u / kCQ, k4 = (u - hp * kCQ) * 4;
            if (u < E_U) *reinterpret_cast<f32x4*>(Es + hp * kLD + k4) = ein[i] ? er[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < W_R; ++i) {
            const int u = tid + i * 256;
            const int row = u / kCQ, k4 = (u - row * kCQ) * 4;
            if (u < W_U) *reinterpret_cast<f32x4*>(Ws + row * kLD + k4) = wr[i];
        }
        if (tid < 10 * kCQ) *reinterpret_cast<f32x4*>(Wd + dt * kCK + dk4) = dr[0];
    };

    const int nchunks = p.Ce / kCK;
    load_chunk(0);
    store_chunk();
    lds_barrier();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more && !(p.ablate & 4)) load_chunk((c + 1) * kCK);
        // ---- B: depthwise 3x3 + shift + ReLU6 -> Ds
        if (dw_on && !(p.ablate & 1)) {
            f32x4 o[SL];
            const f32x4 sh = *reinterpret_cast<const f32x4*>(Wd + 9 * kCK + cq * 4);
#pragma unroll
            for (int j = 0; j < SL; ++j) o[j] = sh;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f32x4 win[Sh::WIN];
                const float* erow = Es + ((sr * S + dy) * IW) * kLD + cq * 4;
#pragma unroll
                for (int i = 0; i < Sh::WIN; ++i)
                    win[i] = *reinterpret_cast<const f32x4*>(erow + min(sx0 * S + i, IW - 1) * kLD);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Wd + (dy * 3 + dx) * kCK + cq * 4);
#pragma unroll
                    for (int j = 0; j < SL; ++j) o[j] += win[j * S + dx] * w;
                }
            }
#pragma unroll
            for (int j = 0; j < SL; ++j) {
                if (sx0 + j < TW) {
                    f32x4 v = o[j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = relu6f(v[q]);
                    *reinterpret_cast<f32x4*>(Ds + (sr * TW + sx0 + j) * kLD + cq * 4) = v;
                }
            }
        }
        lds_barrier();
        // ---- C: project on the MFMA (weights = A operand, pixels = B operand)
#pragma unroll
        for (int kc = 0; kc < ((p.ablate & 2) ? 0 : kCK / 16); ++kc) {
            f32x4 a[NTW], bb[MTW];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni)
                a[ni] = *reinterpret_cast<const f32x4*>(Ws + ((wn * NTW + ni) * 16 + frow) * kLD + kc * 16 + fk);
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi)
                bb[mi] = *reinterpret_cast<const f32x4*>(Ds + ((wm * MTW + mi) * 16 + frow) * kLD + kc * 16 + fk);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ni][s], bb[mi][s], acc[mi][ni], 0, 0, 0);
        }
        lds_barrier();
        if (more && !(p.ablate & 4)) {
            store_chunk();
            lds_barrier();
        }
    }

    // ---- epilogue: lane holds y[pixel = group*16 + (lane & 15)][n = tile*16 + (lane >> 4)*4 + 0..3]
    f32x4 shv[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        const int n = (nt0 + wn * NTW + ni) * 16 + (lane >> 4) * 4;
        shv[ni] = n < p.Cout ? *reinterpret_cast<const f32x4*>(p.ph + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) {
        const int px = (wm * MTW + mi) * 16 + (lane & 15);
        const int r = px / TW, cc = px - r * TW;
        const int oy = oy0 + r, ox = ox0 + cc;
        if (px >= PT || oy >= p.Ho || ox >= p.Wo) continue;
        const long o = (((long)b * p.Ho + oy) * p.Wo + ox) * p.Cout;
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) {
            const int n = (nt0 + wn * NTW + ni) * 16 + (lane >> 4) * 4;
            if (n >= p.Cout) continue;
            f32x4 v = acc[mi][ni] + shv[ni];
            if (p.res) v = v + *reinterpret_cast<const f32x4*>(p.res + o + n);
            *reinterpret_cast<f32x4*>(p.y + o + n) = v;
        }
    }
}

// 8-wave variant: waves 0-3 ("producers") stage chunks and run the depthwise, waves 4-7
// ("consumers") run the project MFMAs of the previous chunk at the same time; E / D / Wp / Wd
// tiles are double-buffered in LDS and one workgroup barrier per chunk hands them over.
// At the start of iteration i:   Ds[i&1] = D(i),  Ws[i&1] = Wp(i),  Es/Wd[(i+1)&1] = E/wd(i+1),
// producer registers hold the in-flight loads of E/wd(i+2) and Wp(i+1).
//   producers, iteration i:  depthwise(i+1): Es[(i+1)&1] -> Ds[(i+1)&1];
//                            registers -> Es/Wd[i&1] (E/wd(i+2)), Ws[(i+1)&1] (Wp(i+1));
//                            issue loads of E/wd(i+3), Wp(i+2)
//   consumers, iteration i:  acc += Ds[i&1] x Ws[i&1]
// (a phase ablation of the 4-wave kernel showed its depthwise, MFMA and staging times simply add
// up: 6.4 + 8.5 + 4.3 us of a 33 us block_7 launch, 10.9 + 20.4 + 9.7 of 55 us for block_11.)
// BF16 (the net's precision-1 mode): the project runs on the bf16 matrix cores -- the consumer waves round their D and
// Wp fragments (fp32 in LDS, as the producers wrote them) to bf16 on the fly, 8 k-values per lane, and issue TWO
// v_mfma_f32_16x16x32_bf16 per (pixel group, output tile) and 48-channel chunk (k 0-31, k 32-47 + zeros) instead of
// twelve v_mfma_f32_16x16x4_f32; depthwise, BatchNorm shifts, ReLU6 and the residual add stay fp32.
template <int S, int TH, int TW, int SL, int WM, int WN, int NTW, int NOPS, bool BF16 = false>
__global__ __launch_bounds__(512) void dwproj8_kernel(const DwProjParams p) {
    using Sh = DwProjShape<S, TH, TW, SL, WM, WN, NTW>;
    constexpr int PT = Sh::PT, PG = Sh::PG, MTW = Sh::MTW, NTB = Sh::NTB, IW = Sh::IW, HP = Sh::HP, kLD = Sh::LD;
    constexpr int ES = HP * kLD, DS = PG * 16 * kLD, WS = NTB * 16 * kLD, WD = 10 * kCK;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Es = sm;                  // [2][HP][kLD]
    float* Ds = Es + 2 * ES;         // [2][PG*16][kLD]
    float* Ws = Ds + 2 * DS;         // [2][NTB*16][kLD]
    float* Wd = Ws + 2 * WS;         // [2][10][kCK]

    const int tid = threadIdx.x & 255, lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const bool producer = threadIdx.x < 256;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_per_img = p.tiles_y * p.tiles_x;
    const int b = blockIdx.x / tiles_per_img;
    const int rem = blockIdx.x - b * tiles_per_img;
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
    const int nt0 = blockIdx.y * NTB;
    const float* eb = p.e + (long)b * p.H * p.W * p.Ce;
    const int nchunks = p.Ce / kCK;

    if (producer) {
        const int cq = tid % kCQ, strip = tid / kCQ;
        const bool dw_on = strip < Sh::STRIPS;
        const int sr = strip / Sh::NSX, sx0 = (strip - sr * Sh::NSX) * SL;
        constexpr int E_U = HP * kCQ, W_U = NTB * 16 * kCQ;
        constexpr int E_R = (E_U + 255) / 256, W_R = (W_U + 255) / 256;
        f32x4 er[E_R], wr[W_R], dr[1];
        const float* eptr[E_R];
        bool ein[E_R];
#pragma unroll
        for (int i = 0; i < E_R; ++i) {
            const int u = min(tid + i * 256, E_U - 1);
            const int hp = u / kCQ, k4 = (u - hp * kCQ) * 4;
            const int r = hp / IW, cc = hp - r * IW;
            const int iy = iy0 + r, ix = ix0 + cc;
            ein[i] = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            eptr[i] = eb + ((long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix, 0), p.W - 1)) * p.Ce + k4;
        }
        const float* wptr[W_R];
#pragma unroll
        for (int i = 0; i < W_R; ++i) {
            const int u = min(tid + i * 256, W_U - 1);
            const int row = u / kCQ, k4 = (u - row * kCQ) * 4;
            wptr[i] = p.wp + (long)min(nt0 * 16 + row, p.npad_p - 1) * p.kpad_p + k4;
        }
        const int dt = min(tid, 10 * kCQ - 1) / kCQ, dk4 = (min(tid, 10 * kCQ - 1) % kCQ) * 4;
        const float* dptr = (dt < 9 ? p.wd + (long)dt * p.Ce : p.dh) + dk4;
        // (every issued asm load is consumed by the matching store_* below: same chunk conditions)
        auto load_e = [&](int k) {
#pragma unroll
            for (int i = 0; i < E_R; ++i) er[i] = gload16_async(eptr[i] + k * kCK);
            dr[0] = gload16_async(dptr + k * kCK);
        };
        auto load_w = [&](int k) {
#pragma unroll
            for (int i = 0; i < W_R; ++i) wr[i] = gload16_async(wptr[i] + k * kCK);
        };
        auto store_e = [&](int buf) {
            wait_prefetch(er);
            wait_prefetch(dr);
#pragma unroll
            for (int i = 0; i < E_R; ++i) {
                const int u = tid + i * 256;
                const int hp = u / kCQ, k4 = (u - hp * kCQ) * 4;
                if (u < E_U)
                    *reinterpret_cast<f32x4*>(Es + buf * ES + hp * kLD + k4) = ein[i] ? er[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (tid < 10 * kCQ) *reinterpret_cast<f32x4*>(Wd + buf * WD + dt * kCK + dk4) = dr[0];
        };
        auto store_w = [&](int buf) {
            wait_prefetch(wr);
#pragma unroll
            for (int i = 0; i < W_R; ++i) {
                const int u = tid + i * 256;
                const int row = u / kCQ, k4 = (u - row * kCQ) * 4;
                if (u < W_U) *reinterpret_cast<f32x4*>(Ws + buf * WS + row * kLD + k4) = wr[i];
            }
        };
        auto depthwise = [&](int buf) {          // Es/Wd[buf] -> Ds[buf]
            if (!dw_on || (p.ablate & 1)) return;
            const float* E = Es + buf * ES;
            const float* Wt = Wd + buf * WD;
            float* D = Ds + buf * DS;
            f32x4 o[SL];
            const f32x4 sh = *reinterpret_cast<const f32x4*>(Wt + 9 * kCK + cq * 4);
#pragma unroll
            for (int j = 0; j < SL; ++j) o[j] = sh;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                f32x4 win[Sh::WIN];
                const float* erow = E + ((sr * S + dy) * IW) * kLD + cq * 4;
#pragma unroll
                for (int i = 0; i < Sh::WIN; ++i)
                    win[i] = *reinterpret_cast<const f32x4*>(erow + min(sx0 * S + i, IW - 1) * kLD);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Wt + (dy * 3 + dx) * kCK + cq * 4);
#pragma unroll
                    for (int j = 0; j < SL; ++j) o[j] += win[j * S + dx] * w;
                }
            }
#pragma unroll
            for (int j = 0; j < SL; ++j) {
                if (sx0 + j < TW) {
                    f32x4 v = o[j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = relu6f(v[q]);
                    *reinterpret_cast<f32x4*>(D + (sr * TW + sx0 + j) * kLD + cq * 4) = v;
                }
            }
        };
        // prologue 1: chunk 0 -> Es/Wd[0], Ws[0]
        load_e(0);
        load_w(0);
        store_e(0);
        store_w(0);
        if (nchunks > 1) load_e(1);
        lds_barrier();
        // prologue 2: D(0); E/wd(1) -> Es/Wd[1]; loads of E/wd(2), Wp(1) in flight
        depthwise(0);
        if (nchunks > 1) store_e(1);
        if (!(p.ablate & 4)) {
            if (nchunks > 2) load_e(2);
            if (nchunks > 1) load_w(1);
        }
        lds_barrier();
        for (int i = 0; i < nchunks; ++i) {
            if (i + 1 < nchunks) depthwise((i + 1) & 1);
            if (!(p.ablate & 4)) {
                if (i + 2 < nchunks) store_e(i & 1);
                if (i + 1 < nchunks) store_w((i + 1) & 1);
                if (i + 3 < nchunks) load_e(i + 3);
                if (i + 2 < nchunks) load_w(i + 2);
            }
            lds_barrier();
        }
        return;
    }

    // ---- consumers: project MFMAs + epilogue
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = (lane >> 4) * 4;
    f32x4 shv[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        const int n = (nt0 + wn * NTW + ni) * 16 + (lane >> 4) * 4;
        shv[ni] = n < p.Cout ? *reinterpret_cast<const f32x4*>(p.ph + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();
    lds_barrier();
    for (int i = 0; i < nchunks; ++i) {
        const float* D = Ds + (i & 1) * DS;
        const float* W = Ws + (i & 1) * WS;
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k0 = ks * 32 + (lane >> 4) * 8;
                const bool ok = k0 < kCK;                  // second step: k 32..47 live in lanes g4 < 2, the rest multiply zeros
                const int kk = ok ? k0 : 0;
                BP<1> a[NTW], bb[MTW];
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) {
                    const float* r = W + ((wn * NTW + ni) * 16 + frow) * kLD + kk;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(r), hi = *reinterpret_cast<const f32x4*>(r + 4);
                    a[ni] = splitN<1>(ok ? lo : z, ok ? hi : z);
                }
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi) {
                    const float* r = D + ((wm * MTW + mi) * 16 + frow) * kLD + kk;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(r), hi = *reinterpret_cast<const f32x4*>(r + 4);
                    bb[mi] = splitN<1>(ok ? lo : z, ok ? hi : z);
                }
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = mmaN<1>(a[ni], bb[mi], acc[mi][ni]);
            }
            lds_barrier();
            continue;
        }
#pragma unroll
        for (int kc = 0; kc < ((p.ablate & 2) ? 0 : kCK / 16); ++kc) {
            f32x4 a[NTW], bb[MTW];
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni)
                a[ni] = *reinterpret_cast<const f32x4*>(W + ((wn * NTW + ni) * 16 + frow) * kLD + kc * 16 + fk);
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi)
                bb[mi] = *reinterpret_cast<const f32x4*>(D + ((wm * MTW + mi) * 16 + frow) * kLD + kc * 16 + fk);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NTW; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ni][s], bb[mi][s], acc[mi][ni], 0, 0, 0);
                        // A wave whose NEXT MFMA waits in the issue stage for the matrix pipe blocks every
                        // other wave of its SIMD (tests/micro/mfma_valu_overlap.hip): idle through the
                        // 8-pass shadow instead, so that the producer wave sharing the SIMD can issue.
                        if (NOPS > 0) asm volatile("s_nop %0" ::"n"(NOPS));
                    }
        }
        lds_barrier();
    }
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) {
        const int px = (wm * MTW + mi) * 16 + (lane & 15);
        const int r = px / TW, cc = px - r * TW;
        const int oy = oy0 + r, ox = ox0 + cc;
        if (px >= PT || oy >= p.Ho || ox >= p.Wo) continue;
        const long o = (((long)b * p.Ho + oy) * p.Wo + ox) * p.Cout;
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) {
            const int n = (nt0 + wn * NTW + ni) * 16 + (lane >> 4) * 4;
            if (n >= p.Cout) continue;
            f32x4 v = acc[mi][ni] + shv[ni];
            if (p.res) v = v + *reinterpret_cast<const f32x4*>(p.res + o + n);
            *reinterpret_cast<f32x4*>(p.y + o + n) = v;
        }
    }
}

typedef void (*dwproj_fn)(const DwProjParams);
struct DwProjCfg {
    int stride, cout_min, cout_max, th, tw, ntb, n_split;
    size_t lds;
    dwproj_fn fn;
    size_t lds8;
    dwproj_fn fn8[4];       // s_nop 0 (none) / 3 / 5 / 6 after each consumer MFMA
    dwproj_fn fn8b;         // the project on the bf16 matrix cores (precision 1)
};
#define DCFG(S, TH, TW, SL, WM, WN, NTW, CMIN, CMAX, NSPLIT)                                               \
    {S, CMIN, CMAX, TH, TW, NTW * WN, NSPLIT, DwProjShape<S, TH, TW, SL, WM, WN, NTW>::lds_floats * 4,       \
     dwproj_kernel<S, TH, TW, SL, WM, WN, NTW>, DwProjShape<S, TH, TW, SL, WM, WN, NTW>::lds_floats * 8,      \
     {dwproj8_kernel<S, TH, TW, SL, WM, WN, NTW, 0>, dwproj8_kernel<S, TH, TW, SL, WM, WN, NTW, 3>,              \
      dwproj8_kernel<S, TH, TW, SL, WM, WN, NTW, 5>, dwproj8_kernel<S, TH, TW, SL, WM, WN, NTW, 6>},              \
     dwproj8_kernel<S, TH, TW, SL, WM, WN, NTW, 0, true>}
const DwProjCfg kDwProj[] = {
    // stride 1, 19-wide row bands (blocks 7-12 of SSD300): 6 pixel groups x Cout/16 tiles on 2x2 waves
    DCFG(1, 5, 19, 5, 2, 2, 2, 1, 64, 1),
    DCFG(1, 5, 19, 5, 2, 2, 3, 65, 96, 1),
    // 10-wide bands (blocks 13-16): 4 pixel groups, output channels split over gridDim.y
    DCFG(2, 5, 10, 3, 4, 1, 5, 97, 160, 2),
    DCFG(1, 5, 10, 3, 4, 1, 5, 97, 160, 2),
    DCFG(1, 5, 10, 3, 4, 1, 10, 161, 320, 2),
};

const DwProjCfg* pick(const DwProjParams& p) {
    if (p.Ce % kCK != 0 || p.Cout % 4 != 0) return nullptr;
    for (const auto& c : kDwProj)
        if (c.stride == p.stride && p.Cout >= c.cout_min && p.Cout <= c.cout_max &&
            (p.Cout + 15) / 16 <= c.ntb * c.n_split)
            return &c;
    return nullptr;
}

}  // namespace

bool dwproj_supported(const DwProjParams& p) { return pick(p) != nullptr; }

int launch_dwproj(DwProjParams p, hipStream_t st) {
    const DwProjCfg* c = pick(p);
    if (!c) {
        set_error("dw+project: unsupported shape Ce=%d Cout=%d stride=%d", p.Ce, p.Cout, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    p.tiles_y = (p.Ho + c->th - 1) / c->th;
    p.tiles_x = (p.Wo + c->tw - 1) / c->tw;
    const long tiles = (long)p.B * p.tiles_y * p.tiles_x;
    SSD_UNSUPPORTED_IF(tiles > 0x7fffffffL, "dw+project: grid too large");
    static const int ablate = getenv("SSD_DWPROJ_ABLATE") ? atoi(getenv("SSD_DWPROJ_ABLATE")) : 0;
    p.ablate = ablate;
    static const int waves = getenv("SSD_DWPROJ_WAVES") ? atoi(getenv("SSD_DWPROJ_WAVES")) : 8;     // diagnostics knob
    static const int nopsel = getenv("SSD_DWPROJ_NOP") ? atoi(getenv("SSD_DWPROJ_NOP")) & 3 : 0;     // diagnostics knob
    if (waves == 8 && c->lds8 <= 160 * 1024) {
        dwproj_fn fn8 = p.bf16 ? c->fn8b : c->fn8[nopsel];
        SSD_HIP(hipFuncSetAttribute((const void*)fn8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds8));
        hipLaunchKernelGGL(fn8, dim3((unsigned)tiles, c->n_split), dim3(512), c->lds8, st, p);
        SSD_LAUNCH_CHECK();
        return SSD_OK;
    }
    if (c->lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)c->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds));
    hipLaunchKernelGGL(c->fn, dim3((unsigned)tiles, c->n_split), dim3(256), c->lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;
}

}  // namespace ssd

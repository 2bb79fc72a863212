// Whole-image fused MobileNetV2 inverted-residual block for the low-resolution stages
// (blocks 7-12 at 19x19 and 14-16 at 10x10 of SSD300; Cin 64 / 96 / 160):
//
//     y = project_BN( relu6(dw_BN( dw3x3( relu6(expand_BN( x * We )) ) )) * Wp ) [+ x]
//
// ([3P] keras-applications MobileNetV2 block_k_expand .. block_k_add, SURVEY.md Appendix A.)
// The tile kernel of ssd_fused.hip recomputes a 1.6-1.9x expand halo around 8x8 tiles and needs
// Cin <= 32; run as expand GEMM + depthwise/project kernel these blocks write the 6x expanded map
// E to HBM and read it back (41-62 MB per layer at B=64).  At 19x19 / 10x10 a WHOLE image fits one
// workgroup, so there is no halo at all: the image border is the zero padding.
//
//   workgroup = (image, group g of the G expanded-channel groups), 512 threads = 8 waves
//   pixel space: q = r * P + c with pitch P = W + 1 -- the one pad column per row is the right
//                padding of row r and the left padding of row r + 1; 16 consecutive q = one MFMA
//                pixel tile, a wave owns T tiles for the whole kernel
//   X  the wave's pixel tiles as MFMA B fragments, loaded ONCE from global into registers
//   per 16-channel chunk of the group's Ce / G expanded channels (one LDS barrier per chunk,
//   E and the weight chunks double-buffered):
//     A  expand   E[q][16] = relu6(X[q][Cin] * We[Cin][16] + shift), 0 at pad positions -> LDS
//     B  depthw.  D[q][16] = relu6(sum_taps E[q + dy*P + dx] * Wd + shift): each lane computes ITS
//                 pixel x 4 channels, i.e. the MFMA B fragment of phase C, from 9 conflict-free
//                 ds_read_b128 -- D never goes back to LDS
//     C  project  acc[q][Cout] += D[q][16] * Wp[16][Cout]   accumulators in registers
//   G > 1: each group holds a partial sum over its channels and stores it as an fp32 slab; y = project
//   BN shift + the G slabs added in group order (deterministic) + residual.  Default: a second,
//   chip-wide launch (image_combine_kernel) does that sum.  Option "image_ticket": the groups of an
//   image meet through an arrival ticket inside the launch (write-through slab stores, agent-scope
//   acquire by the last arriver; cdna_hip_programming.md "in-launch split-K reduction") -- correct for
//   any placement, but the one combining CU per image reads its 370-550 KB at ~30 GB/s (14-21 us).
//   G = 1 (B >= #CUs): direct epilogue, no slabs.
#include <cstdlib>

#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kIC = 16;           // expanded channels per chunk
constexpr int kILD = 24;          // LDS row stride (floats) of the E / Wp chunk tiles: 6 quads, conflict-free b128 fragments
constexpr int kIThreads = 512;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// compiler-invisible prefetch loads + the matching hand-placed wait (idiom of ssd_fused.hip:
// every issued load IS consumed)
__device__ __forceinline__ f32x4 gload16_async(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// write-through (sc1) 16-byte store: the slab lines do not stay dirty in this XCD's L2, so publishing
// them needs no L2 write-back fence (cdna_hip_programming.md, in-launch split-K reduction)
__device__ __forceinline__ void gstore16_sc1(float* ptr, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_prefetch(f32x4 (&r)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}

// S = 2 (block 13): the depthwise / project side works on the Ho x Wo output map (its own pixel space
// qo = ro * (Wo + 1) + co, TO tiles per wave), every lane reads its 9 taps at (2 ro - pad + dy,
// 2 co - pad + dx) of the E chunk; p.e_out != nullptr additionally writes E (after BN + ReLU6) to HBM
// once -- block_13_expand_relu is SSD feature map 1 -- instead of writing it and reading it back.
template <int CIN, int NT, int T, int S = 1>
__global__ __launch_bounds__(kIThreads) void mbv2_image_block_kernel(const FusedBlockParams p) {
    constexpr int TO = S == 1 ? T : 1;            // output pixel tiles per wave
    constexpr int KC = CIN / 16;                  // 16-wide k blocks of the expand
    constexpr int LDW = CIN + 8;                  // We chunk row stride: 18 / 26 / 42 quads (2 mod 4)
    constexpr int NCH = T == 1 ? 2 : 1;           // independent expand accumulator chains per tile
    constexpr int WE_U = kIC * CIN / 4, WE_R = (WE_U + kIThreads - 1) / kIThreads;
    constexpr int WP_U = NT * 16 * kIC / 4, WP_R = (WP_U + kIThreads - 1) / kIThreads;

    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int G = p.groups, B = p.B;
    // the G groups of an image are B block ids apart: the same XCD whenever B % 8 == 0 (speed only)
    const int grp = blockIdx.x / B, img = blockIdx.x - grp * B;
    const int H = p.H, W = p.W, P = W + 1, Q = H * P;
    const int npt = (Q + 15) >> 4;
    const int NE = npt * 16 + 2 * P + 2;          // E rows: index q + P + 1, zero rows above / below
    const int CeG = p.Ce / G, cbeg = grp * CeG, nchunk = CeG / kIC;

    float* Es = sm;                               // [2][NE][kILD]
    float* Wes = Es + 2 * NE * kILD;              // [2][16][LDW]
    float* Wps = Wes + 2 * kIC * LDW;             // [2][NT*16][kILD]
    float* Ps = Wps + 2 * NT * 16 * kILD;         // [11][CeG]: expand shift, depthwise taps [9], depthwise shift
    int* flag = reinterpret_cast<int*>(Ps + 11 * CeG);

    // diagnostics (ssd_net_profile_fused): per-wave cycles of 0 prologue, 1 barrier wait, 2 depthwise,
    // 3 project, 4 expand + weight staging, 5 epilogue / combine
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = p.dbg ? clock64() : 0;
#define ITICK(i) do { if (p.dbg) { const long long t1 = clock64(); tacc[i] += t1 - t0; t0 = t1; } } while (0)
#define IDUMP() do { if (p.dbg && lane == 0) for (int i_ = 0; i_ < 6; ++i_) p.dbg[((long)blockIdx.x * 8 + wave) * 6 + i_] = tacc[i_]; } while (0)

    // ---- weight chunk prefetch (global -> registers -> LDS), one "set" = { Wp(j), We(j + 1) }
    f32x4 wer[WE_R], wpr[WP_R];
    auto load_we_to = [&](f32x4 (&r)[WE_R], int j) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = min(tid + i * kIThreads, WE_U - 1);
            const int row = u / (CIN / 4), k4 = (u - row * (CIN / 4)) * 4;
            r[i] = gload16_async(p.we + (long)(cbeg + j * kIC + row) * p.kpad_e + k4);
        }
    };
    auto store_we_from = [&](const f32x4 (&r)[WE_R], int buf) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = tid + i * kIThreads;
            const int row = u / (CIN / 4), k4 = (u - row * (CIN / 4)) * 4;
            if (u < WE_U) *reinterpret_cast<f32x4*>(Wes + (buf * kIC + row) * LDW + k4) = r[i];
        }
    };
    auto load_we = [&](int j) { load_we_to(wer, j); };
    auto store_we = [&](int buf) { store_we_from(wer, buf); };
    auto load_wp = [&](int j) {
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = min(tid + i * kIThreads, WP_U - 1);
            const int row = u >> 2, k4 = (u & 3) * 4;
            wpr[i] = gload16_async(p.wp + (long)row * p.kpad_p + cbeg + j * kIC + k4);
        }
    };
    auto store_wp = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = tid + i * kIThreads;
            const int row = u >> 2, k4 = (u & 3) * 4;
            if (u < WP_U) *reinterpret_cast<f32x4*>(Wps + (buf * NT * 16 + row) * kILD + k4) = wpr[i];
        }
    };

    // ---- prologue: every global load of the start-up is in flight at once (weight chunks We(0),
    //      We(1), Wp(0), the group's per-channel parameters, the wave's X fragments)
    f32x4 wer0[WE_R];
    load_we_to(wer0, 0);
    if (nchunk > 1) load_we(1);
    load_wp(0);
    // E rows no expand ever writes (above the map, below its last pixel tile) are zero for good; the
    // pad column and the tail of the last tile are written as zeros by every expand
    for (int u = tid; u < 2 * (2 * P + 2) * (kILD / 4); u += kIThreads) {
        const int buf = u / ((2 * P + 2) * (kILD / 4)), v = u - buf * ((2 * P + 2) * (kILD / 4));
        const int row = v / (kILD / 4), c4 = (v - row * (kILD / 4)) * 4;
        const int e = row < P + 1 ? row : npt * 16 + row;          // rows [0, P] and [npt*16 + P + 1, NE)
        *reinterpret_cast<f32x4*>(Es + (buf * NE + e) * kILD + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 xb[T][KC];
    int qs[T];             // pixel index of this lane in tile t (0 for tiles beyond the map)
    bool tvalid[T], real[T];
    int opix[T];           // r * W + c of a real pixel
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = wave * T + t;
        tvalid[t] = tile < npt;
        const int q = tile * 16 + l15;
        const int r = q / P, c = q - r * P;
        real[t] = tvalid[t] && q < Q && c < W;
        qs[t] = tvalid[t] ? q : 0;
        opix[t] = real[t] ? r * W + c : 0;
        const float* xp = p.x + ((long)img * H * W + opix[t]) * CIN + g4 * 4;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            xb[t][kc] = real[t] ? *reinterpret_cast<const f32x4*>(xp + kc * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // output side: the same tiles for stride 1; the Ho x Wo map's own tiles for stride 2
    const int Ho = p.Ho, Wo = p.Wo;
    int qo[TO], opo[TO];   // E index of the window centre (stride 1) / origin (stride 2); ro * Wo + co
    bool realo[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (S == 1) {
            qo[t] = qs[t];
            opo[t] = opix[t];
            realo[t] = real[t];
        } else {
            const int Po = Wo + 1, tile = wave * TO + t;
            const int q = tile * 16 + l15;
            const int ro = q / Po, co = q - ro * Po;
            realo[t] = tile * 16 < Ho * Po && q < Ho * Po && co < Wo;
            // centre of the 3x3 window in E coordinates (TF SAME pads: pad_t / pad_l before)
            qo[t] = (tile * 16 < Ho * Po && q < Ho * Po) ? (2 * ro - p.pad_t + 1) * P + (2 * co - p.pad_l + 1) : 0;
            opo[t] = realo[t] ? ro * Wo + co : 0;
        }
    }
    for (int u = tid; u < 11 * (CeG / 4); u += kIThreads) {
        const int row = u / (CeG / 4), c4 = (u - row * (CeG / 4)) * 4;
        const float* src = row == 0 ? p.eh : row == 10 ? p.dh : p.wd + (long)(row - 1) * p.Ce;
        *reinterpret_cast<f32x4*>(Ps + row * CeG + c4) = *reinterpret_cast<const f32x4*>(src + cbeg + c4);
    }
    wait_prefetch(wer0);
    store_we_from(wer0, 0);
    wait_prefetch(wpr);
    store_wp(0);
    if (nchunk > 1) { wait_prefetch(wer); store_we(1); }
    __syncthreads();

    auto expand = [&](int j) {
        f32x4 ea[T][NCH];
        const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + j * kIC + g4 * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            ea[t][0] = sh;
            if (NCH == 2) ea[t][NCH - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* wes = Wes + ((j & 1) * kIC + l15) * LDW + g4 * 4;
        if (!(p.ablate & 1))
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(wes + kc * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    ea[t][kc % NCH] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xb[t][kc][s], ea[t][kc % NCH], 0, 0, 0);
        }
        float* es = Es + ((j & 1) * NE + P + 1) * kILD + g4 * 4;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 v = ea[t][0];
            if (NCH == 2) v = v + ea[t][NCH - 1];
            const float hi = real[t] ? 6.0f : 0.0f;       // relu6 at real pixels, 0 at pad positions
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, hi);
            if (tvalid[t]) *reinterpret_cast<f32x4*>(es + qs[t] * kILD) = v;
            if (p.e_out && real[t]) {
                const long eo = ((long)img * H * W + opix[t]) * p.Ce + cbeg + j * kIC + g4 * 4;
                *reinterpret_cast<f32x4*>(p.e_out + eo) = v;
                if (p.e_planes) store_planes4(p.e_planes, p.e_plane, p.planes_np, (long)img * H * W + opix[t], cbeg + j * kIC + g4 * 4, (long)B * H * W, v);
            }
        }
    };

    f32x4 acc[TO][NT];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[t][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    expand(0);
    ITICK(0);
    if (nchunk > 1) {           // set 1 = { Wp(1), We(2) } -> registers
        load_wp(1);
        if (nchunk > 2) load_we(2);
    }

    for (int i = 0; i < nchunk; ++i) {
        lds_barrier();          // E(i) and weight set i are visible; everyone is done with interval i - 1
        ITICK(1);
        if (i + 1 < nchunk) {
            wait_prefetch(wpr);
            store_wp((i + 1) & 1);
            if (i + 2 < nchunk) {
                wait_prefetch(wer);
                store_we(i & 1);
                load_wp(i + 2);
                if (i + 3 < nchunk) load_we(i + 3);
            }
        }
        ITICK(4);
        // ---- depthwise in MFMA-fragment layout
        f32x4 a[TO];
        if (p.ablate & 2) {
#pragma unroll
            for (int t = 0; t < TO; ++t) a[t] = f32x4{1.f, 1.f, 1.f, 1.f};
        } else {
            const f32x4 dh = *reinterpret_cast<const f32x4*>(Ps + 10 * CeG + i * kIC + g4 * 4);
#pragma unroll
            for (int t = 0; t < TO; ++t) a[t] = dh;
            const float* es = Es + ((i & 1) * NE + P + 1) * kILD + g4 * 4;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Ps + (1 + (dy + 1) * 3 + dx + 1) * CeG + i * kIC + g4 * 4);
#pragma unroll
                    for (int t = 0; t < TO; ++t) {
                        const f32x4 e = *reinterpret_cast<const f32x4*>(es + (qo[t] + dy * P + dx) * kILD);
                        a[t] += e * w;
                    }
                }
#pragma unroll
            for (int t = 0; t < TO; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[t][e] = __builtin_amdgcn_fmed3f(a[t][e], 0.0f, 6.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
        ITICK(2);
        // ---- project
        if (!(p.ablate & 4)) {
            const float* wps = Wps + ((i & 1) * NT * 16 + l15) * kILD + g4 * 4;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(wps + ni * 16 * kILD);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < TO; ++t)
                        acc[t][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], a[t][s], acc[t][ni], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ITICK(3);
        if (i + 1 < nchunk) expand(i + 1);
        ITICK(4);
    }

    // ---- epilogue
    const long img_off = (long)img * Ho * Wo * p.Cout;
    if (G == 1) {
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (!realo[t]) continue;
            float* yp = p.y + img_off + (long)opo[t] * p.Cout + g4 * 4;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                f32x4 v = acc[t][ni] + *reinterpret_cast<const f32x4*>(p.ph + ni * 16 + g4 * 4);
                if (p.residual)                                     // Cin == Cout: same layout as y
                    v = v + *reinterpret_cast<const f32x4*>(p.x + img_off + (long)opo[t] * p.Cout + ni * 16 + g4 * 4);
                *reinterpret_cast<f32x4*>(yp + ni * 16) = v;
                if (p.y_planes) store_planes4(p.y_planes, p.y_plane, p.planes_np, (long)img * Ho * Wo + opo[t], g4 * 4 + ni * 16, (long)B * Ho * Wo, v);
            }
        }
        ITICK(5);
        IDUMP();
        return;
    }
    if (p.ablate & 16) return;
    const long slab_stride = (long)B * Ho * Wo * p.Cout;              // between groups
    {
        float* sp = p.slabs + (long)grp * slab_stride + img_off + g4 * 4;
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (!realo[t]) continue;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                if (p.tickets) gstore16_sc1(sp + (long)opo[t] * p.Cout + ni * 16, acc[t][ni]);
                else *reinterpret_cast<f32x4*>(sp + (long)opo[t] * p.Cout + ni * 16) = acc[t][ni];
        }
    }
    if (!p.tickets) {        // combine by the follow-up kernel (image_combine_kernel): the launch boundary publishes the slabs
        ITICK(5);
        IDUMP();
        return;
    }
    // publish the slab, draw a ticket; the last arriver of the image combines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.tickets + img, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(G - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(p.tickets + img, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        flag[0] = last;
    }
    __syncthreads();
    if (!flag[0] || (p.ablate & 8)) {
        ITICK(5);
        IDUMP();
        return;
    }
    // last arriver: y = shift + sum of the G partial sums in GROUP order (deterministic whoever
    // arrives last) + residual, as one coalesced pass over the image.  (Measured: this pass runs at
    // ~30 GB/s -- one CU's outstanding-miss budget at the loaded ~2-3 us latency -- whichever way
    // the loads are batched; fetching only the other groups' values in accumulator layout, or four
    // groups per round trip, changed nothing.)
    {
        const int c4n = p.Cout / 4;
        const int nvec = Ho * Wo * c4n;
        const float* s0 = p.slabs + img_off;
        const float* xr = p.x + img_off;                             // residual: Cin == Cout, same layout
        float* yo = p.y + img_off;
#pragma unroll 2
        for (int e = tid; e < nvec; e += kIThreads) {
            const int n4 = (e % c4n) * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(p.ph + n4);
            f32x4 s[8];
            for (int g0 = 0; g0 < G; g0 += 8) {
#pragma unroll
                for (int gg = 0; gg < 8; ++gg)
                    if (g0 + gg < G) s[gg] = *reinterpret_cast<const f32x4*>(s0 + (long)(g0 + gg) * slab_stride + (long)e * 4);
#pragma unroll
                for (int gg = 0; gg < 8; ++gg)
                    if (g0 + gg < G) v = v + s[gg];
            }
            if (p.residual) v = v + *reinterpret_cast<const f32x4*>(xr + (long)e * 4);
            *reinterpret_cast<f32x4*>(yo + (long)e * 4) = v;
        }
    }
    ITICK(5);
    IDUMP();
}


// ---------------------------------------------------------------------------------------------------------------
// The same block on the BF16 matrix cores (the net's precision-1 "bf16" mode: NP = 1, every operand of the two 1x1
// convolutions rounded once to bf16, ONE v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation) and, NP = 3, its
// split-bf16 form for the fp32 net: the exact three-way split of both operands, six matrix instructions per product, fp32
// results (FusedBlockParams.bf16 == 3; it enters the finalize-time race as img_choice 2 and wins blocks 7-10 and 13-15 at
// B = 64: 42-65 -> 35-53 us; the 96-channel blocks 11-12 spill 91 registers in this form and stay on the kernel above).
// Differences to the kernel above:
//   * X is converted once into bf16 B fragments (lane = pixel x 8 channels per 32-wide k-step): CIN / 8 registers per
//     pixel tile instead of CIN / 4
//   * the A fragments come from the bf16 planes of the scale-folded We ([Ce][kpad_e]) / Wp ([npad_p][kpad_p]) through LDS
//   * the project runs every SECOND 16-channel chunk on K = 32 = the two chunks' depthwise outputs (k-slot (g4, j):
//     j < 4 -> even chunk channel g4*4 + j, j >= 4 -> odd chunk), converted in registers
//   * E, the depthwise taps, the BatchNorm shifts, ReLU6, the residual add and the partial-sum slabs stay fp32
//   * weights through LDS: the chunk's expand rows (Wes, two stages, row stride KS*64 + 32 B) and the chunk pair's project
//     rows (Wps, one stage, 64-byte rows with the 16-byte slot XOR-swizzled by (row >> 2) & 3) are copied global ->
//     registers -> LDS one iteration ahead of their use, so no matrix instruction waits on an L2 round trip and a
//     fragment is one ds_read_b128 (reading them straight from L2, the first form of this kernel, measured 72.0 k against
//     82.0 k images/s in the bf16 mode and spilled 44-730 registers in the NP = 3 form)
typedef short bf16x4 __attribute__((ext_vector_type(4)));
#ifdef SSD_IMAGE16_PROF
#define I16TICK(i) ITICK(i)
#define I16DUMP() IDUMP()
#else
#define I16TICK(i) do {} while (0)
#define I16DUMP() do {} while (0)
#endif
template <int CIN, int NT, int T, int S, int NP>
__global__ __launch_bounds__(kIThreads) void mbv2_image16_block_kernel(const FusedBlockParams p) {
    static_assert(CIN % 32 == 0, "whole 32-channel k-steps");
    constexpr int WPL = NP == 1 ? 3 : 0;          // first plane read of [h, m, l, r]
    constexpr int KS = CIN / 32;                  // k-steps of the expand
    constexpr int TO = S == 1 ? T : 1;            // output pixel tiles per wave
    constexpr int NCH = T == 1 ? 2 : 1;           // independent expand accumulator chains per tile
    constexpr int SEH = KS * 32 + 16;             // Wes row stride in bf16 (KS*4 + 2 quads: 2 mod 4)
    constexpr int WE_U = NP * 16 * KS * 4, WE_R = (WE_U + kIThreads - 1) / kIThreads;       // 16-byte units of a We chunk
    constexpr int WP_U = NP * NT * 16 * 8, WP_R = (WP_U + kIThreads - 1) / kIThreads;       // 8-byte units of a Wp chunk pair

    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int G = p.groups, B = p.B;
    const int grp = blockIdx.x / B, img = blockIdx.x - grp * B;
    const int H = p.H, W = p.W, P = W + 1, Q = H * P;
    const int npt = (Q + 15) >> 4;
    const int NE = npt * 16 + 2 * P + 2;          // E rows: index q + P + 1, zero rows above / below
    const int CeG = p.Ce / G, cbeg = grp * CeG, nchunk = CeG / kIC;

#ifdef SSD_IMAGE16_PROF                          // diagnostics build only (phases as in the kernel above): the counters cost registers
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = p.dbg ? clock64() : 0;
#endif
    float* Es = sm;                               // [2][NE][kILD]
    float* Ps = Es + 2 * NE * kILD;               // [11][CeG]: expand shift, depthwise taps [9], depthwise shift
    short* Wes = reinterpret_cast<short*>(Ps + 11 * CeG);      // [2][NP][16][SEH]
    short* Wps = Wes + 2 * NP * 16 * SEH;                      // [NP][NT*16][32]
    const long plane_e = (long)p.Ce * p.kpad_e, plane_p = (long)p.npad_p * p.kpad_p;
    const short* we16 = p.we3 + WPL * plane_e;
    const short* wp16 = p.wp3 + WPL * plane_p;

    // ---- weight chunks global -> registers -> LDS
    constexpr int WE_N = WE_R, WP_N = WP_R;
    bf16x8 wer[WE_N];
    bf16x4 wpr[WP_N];
    auto stage_we_load_to = [&](bf16x8 (&wer)[WE_N], int j) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = min(tid + i * kIThreads, WE_U - 1);
            const int quad = u % (KS * 4), row = (u / (KS * 4)) & 15, pl = u / (KS * 64);
            wer[i] = *reinterpret_cast<const bf16x8*>(we16 + pl * plane_e + (long)(cbeg + j * kIC + row) * p.kpad_e + quad * 8);
        }
    };
    auto stage_we_store_from = [&](const bf16x8 (&wer)[WE_N], int buf) {
#pragma unroll
        for (int i = 0; i < WE_R; ++i) {
            const int u = tid + i * kIThreads;
            const int quad = u % (KS * 4), row = (u / (KS * 4)) & 15, pl = u / (KS * 64);
            if (u < WE_U) *reinterpret_cast<bf16x8*>(Wes + ((buf * NP + pl) * 16 + row) * SEH + quad * 8) = wer[i];
        }
    };
    auto stage_we_load = [&](int j) { stage_we_load_to(wer, j); };
    auto stage_we_store = [&](int buf) { stage_we_store_from(wer, buf); };
    // pair = chunks (2 pair, 2 pair + 1); the half of a lone last chunk's partner is zero
    auto stage_wp_load = [&](int pair) {
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = min(tid + i * kIThreads, WP_U - 1);
            const int q4 = u & 3, half = (u >> 2) & 1, row = (u >> 3) % (NT * 16), pl = u / (NT * 128);
            const int ch = 2 * pair + half;
            wpr[i] = ch < nchunk ? *reinterpret_cast<const bf16x4*>(wp16 + pl * plane_p + (long)row * p.kpad_p + cbeg + ch * kIC + q4 * 4)
                                 : bf16x4{0, 0, 0, 0};
        }
    };
    auto stage_wp_store = [&]() {
#pragma unroll
        for (int i = 0; i < WP_R; ++i) {
            const int u = tid + i * kIThreads;
            const int q4 = u & 3, half = (u >> 2) & 1, row = (u >> 3) % (NT * 16), pl = u / (NT * 128);
            if (u < WP_U)
                *reinterpret_cast<bf16x4*>(Wps + (pl * NT * 16 + row) * 32 + ((q4 ^ ((row >> 2) & 3)) * 8) + half * 4) = wpr[i];
        }
    };

    for (int u = tid; u < 2 * (2 * P + 2) * (kILD / 4); u += kIThreads) {
        const int buf = u / ((2 * P + 2) * (kILD / 4)), v = u - buf * ((2 * P + 2) * (kILD / 4));
        const int row = v / (kILD / 4), c4 = (v - row * (kILD / 4)) * 4;
        const int e = row < P + 1 ? row : npt * 16 + row;          // rows [0, P] and [npt*16 + P + 1, NE)
        *reinterpret_cast<f32x4*>(Es + (buf * NE + e) * kILD + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    BP<NP> xs[T][KS];
    int qs[T];
    bool tvalid[T], real[T];
    int opix[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = wave * T + t;
        tvalid[t] = tile < npt;
        const int q = tile * 16 + l15;
        const int r = q / P, c = q - r * P;
        real[t] = tvalid[t] && q < Q && c < W;
        qs[t] = tvalid[t] ? q : 0;
        opix[t] = real[t] ? r * W + c : 0;
        const float* xp = p.x + ((long)img * H * W + opix[t]) * CIN + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 lo = real[t] ? *reinterpret_cast<const f32x4*>(xp + ks * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 hi = real[t] ? *reinterpret_cast<const f32x4*>(xp + ks * 32 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            xs[t][ks] = splitN<NP>(lo, hi);
        }
    }
    const int Ho = p.Ho, Wo = p.Wo;
    int qo[TO], opo[TO];
    bool realo[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (S == 1) {
            qo[t] = qs[t];
            opo[t] = opix[t];
            realo[t] = real[t];
        } else {
            const int Po = Wo + 1, tile = wave * TO + t;
            const int q = tile * 16 + l15;
            const int ro = q / Po, co = q - ro * Po;
            realo[t] = tile * 16 < Ho * Po && q < Ho * Po && co < Wo;
            qo[t] = (tile * 16 < Ho * Po && q < Ho * Po) ? (2 * ro - p.pad_t + 1) * P + (2 * co - p.pad_l + 1) : 0;
            opo[t] = realo[t] ? ro * Wo + co : 0;
        }
    }
    for (int u = tid; u < 11 * (CeG / 4); u += kIThreads) {
        const int row = u / (CeG / 4), c4 = (u - row * (CeG / 4)) * 4;
        const float* src = row == 0 ? p.eh : row == 10 ? p.dh : p.wd + (long)(row - 1) * p.Ce;
        *reinterpret_cast<f32x4*>(Ps + row * CeG + c4) = *reinterpret_cast<const f32x4*>(src + cbeg + c4);
    }
    {               // start-up: We(0), We(1), Wp(pair 0) in flight together, then We(2) and Wp(pair 1) into the loop's registers
        bf16x8 w0[WE_N], w1[WE_N];
        stage_we_load_to(w0, 0);
        stage_we_load_to(w1, nchunk > 1 ? 1 : 0);
        stage_wp_load(0);
        stage_we_store_from(w0, 0);
        stage_we_store_from(w1, 1);
        stage_wp_store();
        if (nchunk > 2) stage_we_load(2);
        if (nchunk > 2) stage_wp_load(1);
    }
    __syncthreads();

    auto expand = [&](int j) {
        f32x4 ea[T][NCH];
        const f32x4 sh = *reinterpret_cast<const f32x4*>(Ps + j * kIC + g4 * 4);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            ea[t][0] = sh;
            if (NCH == 2) ea[t][NCH - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const short* wl = Wes + ((j & 1) * NP * 16 + l15) * SEH + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            BP<NP> wea;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) wea.p[pl] = *reinterpret_cast<const bf16x8*>(wl + pl * 16 * SEH + ks * 32);
#pragma unroll
            for (int t = 0; t < T; ++t) ea[t][ks % NCH] = mmaN<NP>(wea, xs[t][ks], ea[t][ks % NCH]);
        }
        float* es = Es + ((j & 1) * NE + P + 1) * kILD + g4 * 4;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 v = ea[t][0];
            if (NCH == 2) v = v + ea[t][NCH - 1];
            const float hi = real[t] ? 6.0f : 0.0f;       // relu6 at real pixels, 0 at pad positions
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, hi);
            if (tvalid[t]) *reinterpret_cast<f32x4*>(es + qs[t] * kILD) = v;
            if (p.e_out && real[t]) {
                const long eo = ((long)img * H * W + opix[t]) * p.Ce + cbeg + j * kIC + g4 * 4;
                *reinterpret_cast<f32x4*>(p.e_out + eo) = v;
                if (p.e_planes) store_planes4(p.e_planes, p.e_plane, p.planes_np, (long)img * H * W + opix[t], cbeg + j * kIC + g4 * 4, (long)B * H * W, v);
            }
        }
    };

    f32x4 acc[TO][NT];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[t][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    I16TICK(0);
    expand(0);
    f32x4 dprev[TO];
    I16TICK(4);

    for (int i = 0; i < nchunk; ++i) {
        lds_barrier();          // E(i) is visible; everyone is done with E(i - 1)
        I16TICK(1);
        const bool odd = i & 1;
        const bool flush = odd || i + 1 == nchunk;
        {
            // We(i + 2) (loaded an iteration ago) -> the stage expand(i) read before this barrier; We(i + 3) on its way
            if (i + 2 < nchunk) stage_we_store(i & 1);
            if (i + 3 < nchunk) stage_we_load(i + 3);
            // Wp of the pair (i, i + 1): everyone projected the pair before at iteration i - 1; then the next pair's
            // loads go out, two iterations ahead of their store
            if (!odd && i > 0 && i + 1 < nchunk) {
                stage_wp_store();
                if (i + 2 < nchunk) stage_wp_load((i >> 1) + 1);
            }
        }
        I16TICK(4);
        // ---- depthwise in MFMA-fragment layout
        f32x4 a[TO];
        {
            const f32x4 dh = *reinterpret_cast<const f32x4*>(Ps + 10 * CeG + i * kIC + g4 * 4);
#pragma unroll
            for (int t = 0; t < TO; ++t) a[t] = dh;
            const float* es = Es + ((i & 1) * NE + P + 1) * kILD + g4 * 4;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(Ps + (1 + (dy + 1) * 3 + dx + 1) * CeG + i * kIC + g4 * 4);
#pragma unroll
                    for (int t = 0; t < TO; ++t) {
                        const f32x4 e = *reinterpret_cast<const f32x4*>(es + (qo[t] + (S == 1 ? dy * P + dx : dy * P + dx)) * kILD);
                        a[t] += e * w;
                    }
                }
#pragma unroll
            for (int t = 0; t < TO; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[t][e] = __builtin_amdgcn_fmed3f(a[t][e], 0.0f, 6.0f);
        }
        I16TICK(2);
        // ---- project (every second chunk: K = 32)
        if (!flush) {
#pragma unroll
            for (int t = 0; t < TO; ++t) dprev[t] = a[t];
        } else {
            BP<NP> d[TO];
#pragma unroll
            for (int t = 0; t < TO; ++t) d[t] = odd ? splitN<NP>(dprev[t], a[t]) : splitN<NP>(a[t], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                BP<NP> wa;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const int row = ni * 16 + l15;
                    wa.p[pl] = *reinterpret_cast<const bf16x8*>(Wps + (pl * NT * 16 + row) * 32 + ((g4 ^ ((row >> 2) & 3)) * 8));
                }
#pragma unroll
                for (int t = 0; t < TO; ++t) acc[t][ni] = mmaN<NP>(wa, d[t], acc[t][ni]);
            }
        }
        I16TICK(3);
        if (odd && i + 2 == nchunk) {     // the next chunk is a lone last one: its project reads Wps in its own iteration
            lds_barrier();
            stage_wp_store();
        }
        if (i + 1 < nchunk) {
            expand(i + 1);
        }
        I16TICK(4);
    }

    // ---- epilogue (fp32): G = 1 direct; G > 1 partial-sum slab, combined by image_combine_kernel
    const long img_off = (long)img * Ho * Wo * p.Cout;
    if (G == 1) {
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (!realo[t]) continue;
            float* yp = p.y + img_off + (long)opo[t] * p.Cout + g4 * 4;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                f32x4 v = acc[t][ni] + *reinterpret_cast<const f32x4*>(p.ph + ni * 16 + g4 * 4);
                if (p.residual)
                    v = v + *reinterpret_cast<const f32x4*>(p.x + img_off + (long)opo[t] * p.Cout + ni * 16 + g4 * 4);
                *reinterpret_cast<f32x4*>(yp + ni * 16) = v;
                if (p.y_planes) store_planes4(p.y_planes, p.y_plane, p.planes_np, (long)img * Ho * Wo + opo[t], g4 * 4 + ni * 16, (long)B * Ho * Wo, v);
            }
        }
        I16TICK(5);
        I16DUMP();
        return;
    }
    const long slab_stride = (long)B * Ho * Wo * p.Cout;
    float* sp = p.slabs + (long)grp * slab_stride + img_off + g4 * 4;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (!realo[t]) continue;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) *reinterpret_cast<f32x4*>(sp + (long)opo[t] * p.Cout + ni * 16) = acc[t][ni];
    }
    I16TICK(5);
    I16DUMP();
}
#undef ITICK
#undef IDUMP
#undef I16TICK
#undef I16DUMP

// y = shift + sum of the G slabs in group order (+ residual): the combine as its own launch
__global__ __launch_bounds__(256) void image_combine_kernel(const float* __restrict__ slabs, const float* __restrict__ ph,
                                                            const float* __restrict__ xres, float* __restrict__ y,
                                                            long nvec, long slab_vec, int G, int c4n, short* __restrict__ yp,
                                                            long y_plane, int np) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ph + (e % c4n) * 4);
        f32x4 s[4];
        for (int g0 = 0; g0 < G; g0 += 4) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (g0 + gg < G) s[gg] = *reinterpret_cast<const f32x4*>(slabs + ((long)(g0 + gg) * slab_vec + e) * 4);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (g0 + gg < G) v = v + s[gg];
        }
        if (xres) v = v + *reinterpret_cast<const f32x4*>(xres + e * 4);
        *reinterpret_cast<f32x4*>(y + e * 4) = v;
        if (yp) store_planes4(yp, y_plane, np, e / c4n, (int)(e % c4n) * 4, nvec / c4n, v);
    }
}

typedef void (*image_kernel_t)(const FusedBlockParams);
struct ImageCfg {
    int cin, nt, t, stride;
    image_kernel_t fn;
    image_kernel_t fn16;        // the bf16 form (precision 1)
    image_kernel_t fn3;         // the split-bf16 form: fp32 results on the bf16 matrix cores (FusedBlockParams.bf16 == 3)
};
#define ICFG(CIN, NT, T) {CIN, NT, T, 1, mbv2_image_block_kernel<CIN, NT, T, 1>, mbv2_image16_block_kernel<CIN, NT, T, 1, 1>, \
                             mbv2_image16_block_kernel<CIN, NT, T, 1, 3>}
#define ICFG2(CIN, NT, T) {CIN, NT, T, 2, mbv2_image_block_kernel<CIN, NT, T, 2>, mbv2_image16_block_kernel<CIN, NT, T, 2, 1>, \
                              mbv2_image16_block_kernel<CIN, NT, T, 2, 3>}
const ImageCfg kImage[] = {
    ICFG(64, 4, 3),     // blocks 7-9:   64 -> 384 -> 64 at 19x19
    ICFG(64, 6, 3),     // block 10:     64 -> 384 -> 96
    ICFG(96, 6, 3),     // blocks 11-12: 96 -> 576 -> 96
    ICFG(160, 10, 1),   // blocks 14-15: 160 -> 960 -> 160 at 10x10
    ICFG(160, 20, 1),   // block 16:     160 -> 960 -> 320
    ICFG2(96, 10, 3),   // block 13:     96 -> 576 -> 160, depthwise stride 2 (19x19 -> 10x10), E written out
};

size_t image_lds_bytes(const ImageCfg& c, const FusedBlockParams& p, int G) {
    const int P = p.W + 1, npt = (p.H * P + 15) / 16, NE = npt * 16 + 2 * P + 2;
    if (p.bf16) {
        const int np = p.bf16 == 3 ? 3 : 1, ks = c.cin / 32;
        const size_t w = (size_t)(2 * np * 16 * (ks * 32 + 16) + np * c.nt * 16 * 32) * sizeof(short);      // Wes + Wps
        return ((size_t)2 * NE * kILD + (size_t)11 * (p.Ce / G) + 4) * sizeof(float) + w;
    }
    const size_t fl = (size_t)2 * NE * kILD + (size_t)2 * kIC * (c.cin + 8) + (size_t)2 * c.nt * 16 * kILD +
                      (size_t)11 * (p.Ce / G) + 4;
    return fl * sizeof(float);
}

const ImageCfg* pick_image(const FusedBlockParams& p) {
    if (p.Ce % kIC != 0 || p.Cout % 16 != 0 || p.kpad_e % 4 != 0 || p.kpad_p % 4 != 0) return nullptr;
    if (p.stride == 1 && (p.H != p.Ho || p.W != p.Wo)) return nullptr;
    if (p.stride == 2 && (p.residual || p.Ho != (p.H + 1) / 2 || p.Wo != (p.W + 1) / 2 || p.pad_t > 1 || p.pad_l > 1 ||
                          (p.Ho * (p.Wo + 1) + 15) / 16 > 8))
        return nullptr;
    if (p.stride != 1 && p.stride != 2) return nullptr;
    if (p.residual && p.Cin != p.Cout) return nullptr;
    const int npt = (p.H * (p.W + 1) + 15) / 16;
    FusedBlockParams q = p;
    if (q.bf16 == 3) q.bf16 = 0;        // the split form falls back to the fp32 form where its weight tiles do not fit
    for (const auto& c : kImage)
        if (c.cin == p.Cin && c.nt * 16 == p.Cout && c.stride == p.stride && npt <= 8 * c.t && npt > 8 * (c.t == 3 ? 1 : 0) &&
            image_lds_bytes(c, q, 1) <= 160 * 1024)
            return &c;
    return nullptr;
}

}  // namespace

bool image_block_supported(const FusedBlockParams& p) { return pick_image(p) != nullptr; }
bool image_block_split_fits(FusedBlockParams p) {
    const ImageCfg* c = pick_image(p);
    p.bf16 = 3;
    return c && p.we3 && p.wp3 && image_lds_bytes(*c, p, p.groups < 1 ? 1 : p.groups) <= 160 * 1024;
}

// Number of expanded-channel groups for batch B: the smallest divisor of Ce / 16 that fills the
// CUs (one workgroup per CU: 78-120 KB of LDS), at most 12.
int image_block_groups(const FusedBlockParams& p, int B) {
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cu <= 0)
            num_cu = 256;
    }
    const int units = p.Ce / kIC;
    int best = 1;
    static const int cap = getenv("SSD_IMAGE_GROUPS_CAP") ? atoi(getenv("SSD_IMAGE_GROUPS_CAP")) : 12;   // diagnostics
    for (int g = 1; g <= 12 && g <= units && g <= cap; ++g) {
        if (units % g) continue;
        best = g;
        if ((long)B * g * 8 >= (long)num_cu * 7) break;
    }
    return best;
}

size_t image_block_slab_floats(const FusedBlockParams& p, int B) {
    const int G = image_block_groups(p, B);
    return G > 1 ? (size_t)G * B * p.Ho * p.Wo * p.Cout : 0;
}

int launch_image_block(FusedBlockParams p, hipStream_t st) {
    const ImageCfg* c = pick_image(p);
    if (!c) {
        set_error("image block: unsupported shape Cin=%d Ce=%d Cout=%d %dx%d stride=%d", p.Cin, p.Ce, p.Cout, p.H, p.W, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    if (p.groups < 1) p.groups = 1;
    static const int ablate = getenv("SSD_IMAGE_ABLATE") ? atoi(getenv("SSD_IMAGE_ABLATE")) : 0;
    if (!p.ablate) p.ablate = ablate;
    SSD_CHECK_ARG((p.Ce / kIC) % p.groups == 0, "image block: %d groups do not divide Ce/16 = %d", p.groups, p.Ce / kIC);
    SSD_CHECK_ARG(p.groups == 1 || p.slabs, "image block: %d groups need the slab workspace", p.groups);
    if (p.bf16 == 3 && image_lds_bytes(*c, p, p.groups) > 160 * 1024) p.bf16 = 0;      // split form's weight tiles do not fit: fp32 form
    const size_t lds = image_lds_bytes(*c, p, p.groups);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "image block: needs %zu B of LDS", lds);
    SSD_CHECK_ARG(!p.bf16 || (p.we3 && p.wp3), "image block: the bf16 form needs the weights' bf16 planes");
    SSD_CHECK_ARG(!p.bf16 || !p.tickets, "image block: the bf16 form combines by the second launch only");
    const image_kernel_t fn = p.bf16 == 3 ? c->fn3 : p.bf16 ? c->fn16 : c->fn;
    const bool second = p.form2 && p.bf16 && !p.tickets && !p.ablate && image_block2_supported(p);
    if (second) {
        const int rc = launch_image_block2(p, st);
        if (rc) return rc;
    }
    if (lds > 64 * 1024 && !second)
        SSD_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // groups > 1: p.tickets == nullptr (default) combines the group slabs in a second launch -- the
    // launch boundary publishes them and every CU takes part (B=64: 43 us per block 7 instead of 46);
    // with tickets the last arriving group of each image combines inside the launch.
    if (!second) {
        hipLaunchKernelGGL(fn, dim3((unsigned)((long)p.B * p.groups)), dim3(kIThreads), lds, st, p);
        SSD_LAUNCH_CHECK();
    }
    if (p.groups > 1 && !p.tickets && !(p.ablate & 24)) {
        const long nvec = (long)p.B * p.Ho * p.Wo * p.Cout / 4;
        const int blocks = (int)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096);
        hipLaunchKernelGGL(image_combine_kernel, dim3(blocks), dim3(256), 0, st, p.slabs, p.ph, p.residual ? p.x : nullptr, p.y,
                           nvec, nvec, p.groups, p.Cout / 4, p.y_planes, p.y_plane, p.planes_np);
        SSD_LAUNCH_CHECK();
    }
    return SSD_OK;
}

}  // namespace ssd

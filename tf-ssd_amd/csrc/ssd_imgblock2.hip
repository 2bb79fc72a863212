// Whole-image fused MobileNetV2 inverted-residual block, second form (round 6): the split-bf16 / bf16 kernel of
// ssd_imgblock.hip (mbv2_image16_block_kernel) rebuilt around what its ISA accounting showed -- per 16-channel chunk and wave
// ~430 VALU + 37 ds_read_b128 + 72 MFMAs taking turns, a third of the VALU being LDS / global address arithmetic the compiler
// could not hoist at 252 of 256 registers:
//
//   * the GEOMETRY IS COMPILE TIME (H, W template parameters: 19 x 19 and 10 x 10 are the only maps these blocks see):
//     every LDS address of the loop is ONE register + an instruction immediate
//   * a lane owns T ADJACENT pixels q0 .. q0 + T - 1 of the padded pixel line (q = r * (W + 1) + c) instead of one pixel
//     in each of T tiles 16 apart: MFMA pixel tile t = the t-th pixel of the wave's 16 lanes (a tile is any 16 pixels), so
//     the 3 x 3 windows of a lane's T pixels overlap: 3 (T + 2) ds_read_b128 per chunk instead of 9 T (15 against 27 at
//     T = 3) and each E vector feeds up to three FMAs from registers
//   * E rows of ROW = 24 (T odd) / 20 (T even) floats: lanes T pixels apart then sit T * ROW = 8 (mod 16) words apart --
//     conflict-free b128 reads under gfx950's lane groups
//   * the weight chunks reach LDS by LDS-DMA (buffer_load ... lds: no register round trip, no ds_write, no per-thread 64-bit
//     addresses; the fragment swizzle lives in the per-lane SOURCE offset as in ssd_convdma.hip).  For the project's rows to
//     be plain contiguous 64-byte pieces of the weight plane, the two chunks of a pair take INTERLEAVED channels: the pair's
//     32 channels c0 .. c0 + 31, "even" chunk row r = channel c0 + (r >> 2) * 8 + (r & 3), "odd" chunk + 4 -- lane group g4
//     then holds channels c0 + g4 * 8 .. + 7 over the pair = k-slots g4 * 8 .. + 7 of the project MFMA
//   * the per-channel parameters sit chunk-major in LDS ([chunk][11][16], same interleave)
//
// The arithmetic (split, six-pack order, FMA order of the depthwise, chunk order of the project, slabs) is the first form's;
// only the channel -> k-slot assignment inside a project MFMA differs (another order of the same 32-term fp32 sums).
// Stride 1, and stride 2 for block 13 (one output pixel per lane, its expanded map -- SSD feature map 1 -- written to HBM from
// the expand's epilogue); a group takes whole chunk PAIRS (the Ce / 32 pairs dealt as evenly as they go over the groups: no lone last
// chunk); everything else keeps the first form.
#include <cstdlib>

#include "ssd_bf16x3.h"
#include "ssd_conv.h"

namespace ssd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_dst2_t;

namespace {

constexpr int kC = 16;            // expanded channels per chunk

__device__ __forceinline__ void lds_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (a __device__ body: a __global__ function that declares __amdgpu_buffer_rsrc_t objects loses its host stub)
// LOOP 0: one chunk per iteration (runtime buffer parity, project every second chunk); 1: unrolled by chunk pairs.
// Measured and NOT kept (profiles/HISTORY.md, round 6): 12 waves x 2 pixels and 4 waves x 6 pixels (one wave per SIMD, 512
// registers: 500+ v_accvgpr moves per pair), the pair loop with fenced phases + sched_group_barrier interleave, the two waves
// of a SIMD walking an iteration in opposite order -- all equal or slower; the loop's time stays the SUM of its matrix, vector
// and LDS time (phase ablation: tests/micro/imgblock2_prof.py on a -DSSD_IMAGE2_ABLATE build)
template <int CIN, int NT, int NW, int T, int H, int W, int NP, int LOOP, int ABL, int S = 1>
__device__ __forceinline__ void image16v2_body(const FusedBlockParams& p, float* __restrict__ sm2) {
    static_assert(CIN % 32 == 0, "whole 32-channel k-steps");
    constexpr int NTH = NW * 64;
    constexpr int P = W + 1, Q = H * P;
    static_assert(NW * 16 * T >= Q, "the waves' pixel slots cover the padded map");
    constexpr int NPIX = NW * 16 * T;
    constexpr int NE = NPIX + 2 * P + 2;           // E rows: index q + P + 1, zero rows above / below
    // floats per E row (16 used).  Stride 1: the window reads of lanes T pixels apart; stride 2 (block 13): the window reads of
    // lanes TWO pixels apart (one output pixel per lane) -- both want 8 (mod 16) words between neighbouring lanes
    constexpr int ROW = S == 2 ? 20 : (T & 1) ? 24 : 20;
    static_assert(((S == 2 ? 2 : T) * ROW) % 16 == 8, "neighbouring lanes must sit 8 (mod 16) words apart");
    constexpr int TO = S == 1 ? T : 1;             // output pixels per lane
    constexpr int Ho = S == 1 ? H : (H + 1) / 2, Wo = S == 1 ? W : (W + 1) / 2, Po = Wo + 1;
    static_assert(S == 1 || NW * 16 >= Ho * Po, "stride 2: one output pixel per lane covers the output map");
    constexpr int WPL = NP == 1 ? 3 : 0;           // first plane read of [h, m, l, r]
    constexpr int KS = CIN / 32;                   // k-steps of the expand
    constexpr int EBUF = NE * ROW;                 // floats per E buffer
    constexpr int PCH = 11 * kC;                   // floats of one chunk's parameters: expand shift, taps [9], depthwise shift
    constexpr int NBE = NP * KS, NBP = NP * NT;    // 1 KB blocks (16 rows x 32 bf16) of a We chunk / a Wp chunk pair

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int G = p.groups, B = p.B;
    const int grp = blockIdx.x / B, img = blockIdx.x - grp * B;
    // the group's chunk PAIRS: the Ce / 32 pairs dealt as evenly as they go (groups need not hold the same number: 60 chunks
    // over 4 groups = 16, 16, 14, 14 -- always an even number, never a lone last chunk)
    const int pairs_total = p.Ce / (2 * kC), pq = pairs_total / G, pr = pairs_total - pq * G;
    const int npairs = pq + (grp < pr ? 1 : 0), pair0 = grp * pq + min(grp, pr);
    const int cbeg = pair0 * 2 * kC, nchunk = 2 * npairs, CeG = nchunk * kC;
    const int CeGmax = (pq + (pr ? 1 : 0)) * 2 * kC;              // (the LDS layout is the same in every group)

    // diagnostics (ssd_net_profile_fused, p.dbg): shader-clock stamps at the phase boundaries, scalar registers only
    const long long tk0 = p.dbg ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long tk1 = 0, tk2 = 0, tk3 = 0;

    float* Es = sm2;                               // [2][NE][ROW]
    float* Ps = Es + 2 * EBUF;                     // [nchunk][11][16]
    short* Wes = reinterpret_cast<short*>(Ps + 11 * CeGmax);   // [2][NP][KS] blocks of 512 bf16, quad-swizzled
    short* Wps = Wes + 2 * NBE * 512;                          // [NP][NT] blocks
    const long plane_e = (long)p.Ce * p.kpad_e, plane_p = (long)p.npad_p * p.kpad_p;

    // ---- weight chunks global -> LDS by LDS-DMA.  One wave instruction copies a 1 KB block = 16 rows x 64 bytes (one
    //      32-wide k-step of 16 weight rows of one plane); lane i fetches row i >> 2, quad (i & 3) ^ ((i >> 3) & 3), the block
    //      lands lane-linear: slot s of row r holds quad s ^ ((r >> 1) & 3) -- conflict-free b128 fragment reads
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.we3 + WPL * plane_e), 0, (int)(NP * plane_e * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(p.wp3 + WPL * plane_p), 0, (int)(NP * plane_p * 2), 0x00020000);
    const int dr = lane >> 2, dq8 = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;
    const int voff_e = ((((dr >> 2) * 8 + (dr & 3)) * p.kpad_e) + dq8) * 2;       // expand rows: the pair's interleaved channels
    const int voff_p = (dr * p.kpad_p + dq8) * 2;                                  // project rows: 32 contiguous channels
    auto dma_we = [&](const int j, const int stage) {            // chunk j = 2 pair + half
        const int ch0 = cbeg + (j >> 1) * 32 + (j & 1) * 4;
        for (int b = wave_s; b < NBE; b += NW) {
            const int pl = b / KS, ks = b - pl * KS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_e, (lds_dst2_t)(Wes + (stage * NBE + b) * 512), 16, voff_e,
                                                     (int)((pl * plane_e + (long)ch0 * p.kpad_e + ks * 32) * 2), 0, 0);
        }
    };
    auto dma_wp = [&](const int pair) {
        for (int b = wave_s; b < NBP; b += NW) {
            const int pl = b / NT, nb = b - pl * NT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_p, (lds_dst2_t)(Wps + b * 512), 16, voff_p,
                                                     (int)((pl * plane_p + (long)nb * 16 * p.kpad_p + cbeg + pair * 32) * 2), 0, 0);
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    dma_we(0, 0);           // start-up: We(0), We(1), Wp(pair 0) on their way beside the X loads
    dma_we(1, 1);
    dma_wp(0);

    // E rows no expand ever writes (above the map: rows [0, P]; below its last pixel slot: [NPIX + P + 1, NE)) are zero for good
    for (int u = tid; u < 2 * (2 * P + 2) * (ROW / 4); u += NTH) {
        const int buf = u / ((2 * P + 2) * (ROW / 4)), v = u - buf * ((2 * P + 2) * (ROW / 4));
        const int row = v / (ROW / 4), c4 = (v - row * (ROW / 4)) * 4;
        const int e = row < P + 1 ? row : NPIX + row;
        *reinterpret_cast<f32x4*>(Es + buf * EBUF + e * ROW + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- the lane's T adjacent pixels: X as bf16 B fragments (NP planes), loaded and split once
    const int q0 = (wave * 16 + l15) * T;
    BP<NP> xs[T][KS];
    bool real[T];
    int opix[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int q = q0 + t;
        const int r = q / P, c = q - r * P;
        real[t] = q < Q && c < W;
        opix[t] = real[t] ? r * W + c : 0;
        const float* xp = p.x + ((long)img * (H * W) + opix[t]) * CIN + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 lo = real[t] ? *reinterpret_cast<const f32x4*>(xp + ks * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 hi = real[t] ? *reinterpret_cast<const f32x4*>(xp + ks * 32 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            xs[t][ks] = splitN<NP>(lo, hi);
        }
    }
    // per-channel parameters, chunk-major: Ps[(j * 11 + row) * 16 + r], r = the chunk's MFMA row; channel of (j, r) =
    // cbeg + (j >> 1) * 32 + (r >> 2) * 8 + (j & 1) * 4 + (r & 3): the four channels of one source quad stay together
    for (int u = tid; u < 11 * (CeG / 4); u += NTH) {
        const int row = u / (CeG / 4), c4 = (u - row * (CeG / 4)) * 4;          // c4: channel offset inside the group
        const float* src = row == 0 ? p.eh : row == 10 ? p.dh : p.wd + (long)(row - 1) * p.Ce;
        const int w = c4 & 31, j = 2 * (c4 >> 5) + ((w >> 2) & 1), r = (w >> 3) * 4;
        *reinterpret_cast<f32x4*>(Ps + (j * 11 + row) * kC + r) = *reinterpret_cast<const f32x4*>(src + cbeg + c4);
    }
    dma_wait();
    if (p.dbg) tk3 = (long long)__builtin_amdgcn_s_memtime();        // own loads + copies landed
    __syncthreads();
    if (p.dbg) tk1 = (long long)__builtin_amdgcn_s_memtime();

    // one address register each: the lane's E row of pixel q0 (+ the buffer / tap / pixel offsets as immediates), its
    // parameter vector, its fragment slot inside a 1 KB weight block
    const float* e_rd = Es + (q0 + P + 1) * ROW + g4 * 4;       // (stride 2: re-pointed at the lane's output pixel below)
    float* e_wr = Es + (q0 + P + 1) * ROW + g4 * 4;
    const float* ps_l = Ps + g4 * 4;
    const int fslot = l15 * 32 + ((g4 ^ ((l15 >> 1) & 3)) * 8);
    const short* we_l = Wes + fslot;
    const short* wp_l = Wps + fslot;
    float relu_hi[T];
#pragma unroll
    for (int t = 0; t < T; ++t) relu_hi[t] = real[t] ? 6.0f : 0.0f;       // relu6 at real pixels, 0 at pad positions
    // output side: the same pixels for stride 1; stride 2: ONE pixel of the Ho x Wo map per lane (own pixel line qo = ro *
    // (Wo + 1) + co), its 3 x 3 window centred at (2 ro - pad_t + 1, 2 co - pad_l + 1) of the E map (TF SAME pads)
    bool realo[TO];
    int opo[TO];
    if constexpr (S == 1) {
#pragma unroll
        for (int t = 0; t < T; ++t) { realo[t] = real[t]; opo[t] = opix[t]; }
    } else {
        const int qo = wave * 16 + l15, ro = qo / Po, co = qo - ro * Po;
        const bool in = qo < Ho * Po;
        realo[0] = in && co < Wo;
        opo[0] = realo[0] ? ro * Wo + co : 0;
        e_rd = Es + ((in ? (2 * ro - p.pad_t + 1) * P + (2 * co - p.pad_l + 1) : 0) + P + 1) * ROW + g4 * 4;
    }

    // expand chunk j into E buffer pb: E = relu6(We x X + shift)
    auto expand = [&](const int j, const int pb) {
        f32x4 ea[T];
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ps_l + j * PCH);
#pragma unroll
        for (int t = 0; t < T; ++t) ea[t] = sh;
        const short* wl = we_l + pb * NBE * 512;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            BP<NP> wea;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) wea.p[pl] = *reinterpret_cast<const bf16x8*>(wl + (pl * KS + ks) * 512);
#pragma unroll
            for (int t = 0; t < T; ++t) if (!(ABL & 1)) ea[t] = mmaN<NP>(wea, xs[t][ks], ea[t]);
        }
        float* ew = e_wr + pb * EBUF;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 v = ea[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.0f, relu_hi[t]);
            *reinterpret_cast<f32x4*>(ew + t * ROW) = v;
            if (S == 2 && p.e_out && real[t]) {       // block 13: the expanded map is SSD feature map 1 -- written once, from here
                const int ch = cbeg + (j >> 1) * 32 + g4 * 8 + (j & 1) * 4;       // the lane's four channels of chunk j
                *reinterpret_cast<f32x4*>(p.e_out + ((long)img * (H * W) + opix[t]) * p.Ce + ch) = v;
                if (p.e_planes) store_planes4(p.e_planes, p.e_plane, p.planes_np, (long)img * (H * W) + opix[t], ch, (long)B * (H * W), v);
            }
        }
    };
    // depthwise of chunk i from E buffer pb: the lane's TO output pixels x 4 channels = the project MFMA's B fragment
    auto depthwise = [&](const int i, const int pb, f32x4 (&a)[TO]) {
        const float* pc = ps_l + i * PCH;
        const f32x4 dh = *reinterpret_cast<const f32x4*>(pc + 10 * kC);
#pragma unroll
        for (int t = 0; t < TO; ++t) a[t] = dh;
        const float* es = e_rd + pb * EBUF;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            f32x4 w[3], e[TO + 2];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) w[dx] = *reinterpret_cast<const f32x4*>(pc + (1 + (dy + 1) * 3 + dx) * kC);
#pragma unroll
            for (int j = 0; j < TO + 2; ++j)
                e[j] = (ABL & 16) ? w[j % 3] : *reinterpret_cast<const f32x4*>(es + (dy * P + j - 1) * ROW);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int t = 0; t < TO; ++t) {
                    if (ABL & 2) a[t] += e[t + dx];
                    else a[t] += e[t + dx] * w[dx];
                }
        }
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[t][e] = __builtin_amdgcn_fmed3f(a[t][e], 0.0f, 6.0f);
    };
    f32x4 acc[TO][NT];
#pragma unroll
    for (int t = 0; t < TO; ++t)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[t][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    // project of a chunk pair: K = 32 = the lane's 4 + 4 channels of the two chunks (k-slots g4 * 8 ..)
    auto project = [&](const f32x4 (&a0)[TO], const f32x4 (&a1)[TO]) {
        BP<NP> d[TO];
#pragma unroll
        for (int t = 0; t < TO; ++t) d[t] = (ABL & 8) ? xs[t][0] : splitN<NP>(a0[t], a1[t]);
        if (ABL & 8) {
#pragma unroll
            for (int t = 0; t < TO; ++t) asm volatile("" :: "v"(a0[t]), "v"(a1[t]));
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            BP<NP> wa;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) wa.p[pl] = *reinterpret_cast<const bf16x8*>(wp_l + (pl * NT + ni) * 512);
#pragma unroll
            for (int t = 0; t < TO; ++t) {
                if (!(ABL & 4)) acc[t][ni] = mmaN<NP>(wa, d[t], acc[t][ni]);
                else asm volatile("" :: "v"(d[t].p[0]), "v"(wa.p[0]));
            }
        }
    };

    expand(0, 0);

    if constexpr (LOOP == 0) {
        f32x4 dprev[TO];
        for (int c = 0; c < nchunk; ++c) {
            dma_wait();              // the copies issued an iteration ago have landed ...
            lds_barrier2();          // ... and are visible; E(c) is visible; everyone is done with E(c - 1)
            const int pb = c & 1;
            if (c + 2 < nchunk) dma_we(c + 2, pb);                  // the stage expand(c) read before this barrier
            if (!pb && c > 0) dma_wp(c >> 1);                       // everyone projected the pair before at iteration c - 1
            f32x4 a[TO];
            depthwise(c, pb, a);
            if (!pb) {
#pragma unroll
                for (int t = 0; t < TO; ++t) dprev[t] = a[t];
            } else {
                project(dprev, a);
            }
            if (c + 1 < nchunk) expand(c + 1, pb ^ 1);
        }
    } else {
        for (int i = 0; i < nchunk; i += 2) {
            f32x4 a0[TO], a1[TO];
            dma_wait();
            lds_barrier2();          // E(i), We(i + 1) visible; everyone is done with E(i - 1) and with the pair before
            if (i + 2 < nchunk) dma_we(i + 2, 0);
            if (i > 0) dma_wp(i >> 1);
            depthwise(i, 0, a0);
            expand(i + 1, 1);
            dma_wait();
            lds_barrier2();          // E(i + 1), We(i + 2), Wp(pair) visible
            if (i + 3 < nchunk) dma_we(i + 3, 1);
            depthwise(i + 1, 1, a1);
            project(a0, a1);
            if (i + 2 < nchunk) expand(i + 2, 0);
        }
    }

    if (p.dbg) tk2 = (long long)__builtin_amdgcn_s_memtime();
    auto dump = [&]() {          // 0 prologue (own part), 1 prologue barrier wait, 2 first expand + chunk loop, 3 epilogue
        if (p.dbg && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long tk4 = (long long)__builtin_amdgcn_s_memtime();
            long long* o = p.dbg + ((long)blockIdx.x * NW + wave) * 6;
            o[0] = tk3 - tk0; o[1] = tk1 - tk3; o[2] = tk2 - tk1; o[3] = tk4 - tk2; o[4] = 0; o[5] = 1;
        }
    };
    // ---- epilogue (fp32): G = 1 direct; G > 1 partial-sum slab, combined by image_combine_kernel
    const long img_off = (long)img * (Ho * Wo) * p.Cout;
    if (G == 1) {
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            if (!realo[t]) continue;
            float* yp = p.y + img_off + (long)opo[t] * p.Cout + g4 * 4;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                f32x4 v = acc[t][ni] + *reinterpret_cast<const f32x4*>(p.ph + ni * 16 + g4 * 4);
                if (S == 1 && p.residual)
                    v = v + *reinterpret_cast<const f32x4*>(p.x + img_off + (long)opo[t] * p.Cout + ni * 16 + g4 * 4);
                *reinterpret_cast<f32x4*>(yp + ni * 16) = v;
                if (p.y_planes) store_planes4(p.y_planes, p.y_plane, p.planes_np, (long)img * (Ho * Wo) + opo[t], g4 * 4 + ni * 16, (long)B * (Ho * Wo), v);
            }
        }
        dump();
        return;
    }
    const long slab_stride = (long)B * (Ho * Wo) * p.Cout;
    float* sp = p.slabs + (long)grp * slab_stride + img_off + g4 * 4;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        if (!realo[t]) continue;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) *reinterpret_cast<f32x4*>(sp + (long)opo[t] * p.Cout + ni * 16) = acc[t][ni];
    }
    dump();
}

template <int CIN, int NT, int NW, int T, int H, int W, int NP, int LOOP, int ABL = 0, int S = 1>
__global__ __launch_bounds__(NW * 64) void mbv2_image16v2_kernel(const FusedBlockParams p) {
    extern __shared__ __attribute__((aligned(1024))) float sm2[];
    image16v2_body<CIN, NT, NW, T, H, W, NP, LOOP, ABL, S>(p, sm2);
}

typedef void (*image2_kernel_t)(const FusedBlockParams);
struct Image2Cfg {
    int cin, nt, nw, t, h, w, stride;
    image2_kernel_t fn1, fn3;       // bf16 mode (NP = 1), split-bf16 form (NP = 3)
};
#define I2CFG(CIN, NT, NW, T, H, W, LOOP) {CIN, NT, NW, T, H, W, 1, mbv2_image16v2_kernel<CIN, NT, NW, T, H, W, 1, LOOP>, mbv2_image16v2_kernel<CIN, NT, NW, T, H, W, 3, LOOP>}
#define I2ABL(A) {64, 4, 8, 3, 19, 19, 1, mbv2_image16v2_kernel<64, 4, 8, 3, 19, 19, 1, 0, A>, mbv2_image16v2_kernel<64, 4, 8, 3, 19, 19, 3, 0, A>}
#define I2CFG2(CIN, NT, NW, T, H, W, LOOP) {CIN, NT, NW, T, H, W, 2, mbv2_image16v2_kernel<CIN, NT, NW, T, H, W, 1, LOOP, 0, 2>, mbv2_image16v2_kernel<CIN, NT, NW, T, H, W, 3, LOOP, 0, 2>}
const Image2Cfg kImage2[] = {       // (the first configuration of a shape is the default; SSD_IMAGE2_VARIANT=n picks the n-th: A/B runs)
    I2CFG(64, 4, 8, 3, 19, 19, 1),     // blocks 7-9:   64 -> 384 -> 64 at 19x19
    I2CFG(64, 4, 8, 3, 19, 19, 0),
#ifdef SSD_IMAGE2_ABLATE               // diagnostics build: variants 2 .. 7 = the single-chunk loop with phases removed (wrong results)
    I2ABL(1), I2ABL(2), I2ABL(4), I2ABL(8), I2ABL(16), I2ABL(31),
#endif
    I2CFG(64, 6, 8, 3, 19, 19, 1),     // block 10:     64 -> 384 -> 96
    I2CFG(64, 6, 8, 3, 19, 19, 0),
    I2CFG(96, 6, 8, 3, 19, 19, 0),     // blocks 11-12: 96 -> 576 -> 96 (the pair loop spills 84 registers here: 86 us against 52)
    I2CFG2(96, 10, 8, 3, 19, 19, 0),   // block 13:     96 -> 576 -> 160, depthwise stride 2 (19x19 -> 10x10), E written out
    I2CFG2(96, 10, 8, 3, 19, 19, 1),
    I2CFG(160, 10, 8, 1, 10, 10, 1),   // blocks 14-15: 160 -> 960 -> 160 at 10x10
    I2CFG(160, 10, 8, 1, 10, 10, 0),
    I2CFG(160, 20, 8, 1, 10, 10, 1),   // block 16:     160 -> 960 -> 320
    I2CFG(160, 20, 8, 1, 10, 10, 0),
};


size_t image2_lds_bytes(const Image2Cfg& c, const FusedBlockParams& p, int G) {
    const int P = c.w + 1, npix = c.nw * 16 * c.t, NE = npix + 2 * P + 2, row = c.stride == 2 ? 20 : (c.t & 1) ? 24 : 20;
    const int np = p.bf16 == 3 ? 3 : 1, ks = c.cin / 32;
    const size_t w = (size_t)(2 * np * ks + np * c.nt) * 1024;      // Wes (two stages) + Wps
    const int pairs = p.Ce / (2 * kC), cmax = ((pairs + G - 1) / G) * 2 * kC;      // channels of the largest group
    return ((size_t)2 * NE * row + (size_t)11 * cmax) * sizeof(float) + w;
}

const Image2Cfg* pick_image2(const FusedBlockParams& p, int variant) {
    if (!p.bf16 || p.Ce % (2 * kC) != 0 || p.kpad_e % 32 != 0 || p.kpad_p % 32 != 0) return nullptr;
    if (p.stride == 1 && (p.e_out || p.H != p.Ho || p.W != p.Wo)) return nullptr;
    if (p.stride == 2 && (p.residual || p.Ho != (p.H + 1) / 2 || p.Wo != (p.W + 1) / 2 || p.pad_t < 0 || p.pad_t > 1 || p.pad_l < 0 || p.pad_l > 1)) return nullptr;
    if (p.stride != 1 && p.stride != 2) return nullptr;
    if (p.residual && p.Cin != p.Cout) return nullptr;
    int seen = 0;
    for (const auto& c : kImage2)
        if (c.cin == p.Cin && c.nt * 16 == p.Cout && c.h == p.H && c.w == p.W && c.stride == p.stride && p.kpad_e == c.cin && p.npad_p >= c.nt * 16 && seen++ == variant) return &c;
    return nullptr;
}

}  // namespace

bool image_block2_supported(const FusedBlockParams& p) {
    const Image2Cfg* c = pick_image2(p, 0);
    const int G = p.groups < 1 ? 1 : p.groups;
    return c && p.we3 && p.wp3 && G <= p.Ce / (2 * kC) && image2_lds_bytes(*c, p, G) <= 160 * 1024;
}

int launch_image_block2(FusedBlockParams p, hipStream_t st) {
    static const int variant = getenv("SSD_IMAGE2_VARIANT") ? atoi(getenv("SSD_IMAGE2_VARIANT")) : 0;
    const Image2Cfg* c = pick_image2(p, variant);
    if (!c) c = pick_image2(p, 0);
    if (!c) {
        set_error("image block (second form): unsupported shape Cin=%d Ce=%d Cout=%d %dx%d stride=%d", p.Cin, p.Ce, p.Cout, p.H, p.W, p.stride);
        return SSD_E_UNSUPPORTED;
    }
    if (p.B == 0) return SSD_OK;
    if (p.groups < 1) p.groups = 1;
    SSD_CHECK_ARG(p.groups <= p.Ce / (2 * kC), "image block (second form): %d groups of %d chunk pairs", p.groups, p.Ce / (2 * kC));
    SSD_CHECK_ARG(p.groups == 1 || p.slabs, "image block: %d groups need the slab workspace", p.groups);
    SSD_CHECK_ARG(p.we3 && p.wp3, "image block: the bf16 forms need the weights' bf16 planes");
    const size_t lds = image2_lds_bytes(*c, p, p.groups);
    SSD_UNSUPPORTED_IF(lds > 160 * 1024, "image block (second form): needs %zu B of LDS", lds);
    const image2_kernel_t fn = p.bf16 == 3 ? c->fn3 : c->fn1;
    if (lds > 64 * 1024)
        SSD_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)((long)p.B * p.groups)), dim3(c->nw * 64), lds, st, p);
    SSD_LAUNCH_CHECK();
    return SSD_OK;         // (groups > 1: the caller, launch_image_block, adds image_combine_kernel)
}

}  // namespace ssd

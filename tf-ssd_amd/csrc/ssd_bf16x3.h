// FP32 products on the BF16 matrix cores (gfx950): every fp32 operand is split EXACTLY into three bf16 pieces,
//
//     x = h + m + l,   h = x & 0xffff0000,  m = (x - h) & 0xffff0000,  l = x - h - m      (8 + 8 + 8 significand bits)
//     x * y = hh + hm + mh + hl + lh + mm  (+ ml + lm + ll, dropped: <= ~2^-23 |x y|, the size of ONE fp32 rounding)
//
// i.e. six v_mfma_f32_16x16x32_bf16 with fp32 accumulation replace eight v_mfma_f32_16x16x4_f32 per 16 x 16 x 32 block.
// Measured (tests/micro/bf16x3_mfma.hip): error vs float64 2.3e-6 against 3.2e-6 for the fp32 MFMA chain on |D| <= 22
// (as accurate as the fp32 instruction), 404 TFLOP/s fp32-equivalent against 151 (2.67x), and -- the point on gfx950,
// where the f32-input MFMA runs at the VECTOR rate and shares the VALU's issue -- the bf16 MFMAs overlap with a
// kernel's VALU / LDS work instead of adding to it.  Results are not bit-identical to the fp32-MFMA kernels (another,
// equally valid, rounding of the same sums).
#pragma once
#include <hip/hip_runtime.h>

namespace ssd {

typedef float b3_f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct B3 {
    bf16x8 h, m, l;
};

__device__ __forceinline__ void split1(float x, short& h, short& m, short& l) {
    const unsigned hb = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h = (short)(hb >> 16);
    m = (short)(mb >> 16);
    l = (short)(__float_as_uint(r2) >> 16);
}
// four k-values -> 4 bf16 (8 bytes) per plane; pair packing with v_perm_b32 (upper halves of two dwords)
__device__ __forceinline__ void split4(const b3_f32x4 v, uint2& h, uint2& m, uint2& l) {
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hb[j] = __float_as_uint(v[j]) & 0xffff0000u;
        const float r1 = v[j] - __uint_as_float(hb[j]);
        mb[j] = __float_as_uint(r1) & 0xffff0000u;
        lb[j] = __float_as_uint(r1 - __uint_as_float(mb[j]));
    }
    h = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
    m = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
    l = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
}
// lane's 8 k-values (a = k 0..3, b = k 4..7) -> the three bf16 fragments
__device__ __forceinline__ B3 split3(const b3_f32x4 a, const b3_f32x4 b) {
    B3 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        short h, m, l;
        split1(a[j], h, m, l);
        r.h[j] = h; r.m[j] = m; r.l[j] = l;
        split1(b[j], h, m, l);
        r.h[4 + j] = h; r.m[4 + j] = m; r.l[4 + j] = l;
    }
    return r;
}
// acc += W * X over one K = 32 step, six products, small terms first
__device__ __forceinline__ b3_f32x4 mma6(const B3& w, const B3& x, b3_f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.m, x.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, x.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.l, x.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, x.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.m, x.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, x.h, acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------------------------------------------------------
// Precision-generic forms: NP = 3 the exact three-way split above (fp32 results), NP = 1 the "bf16" mode of the net
// (option "precision" 1; BASELINE.json configs[3] / [4]): every matrix operand is rounded ONCE to bf16 (round to
// nearest even, v_cvt_pk_bf16_f32) and every product is ONE v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- the
// standard mixed-precision contraction, 1/6 of the split form's matrix work.  Epilogues (BatchNorm shift, ReLU6,
// residual adds), the depthwise taps, softmax and the box math stay fp32 in either mode.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float b3_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned rne2(float a, float b) {          // two fp32 -> packed bf16 pair (a in the low half)
    const bf16x2_t r = __builtin_convertvector(b3_f32x2{a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ short rne1(float a) { return (short)(rne2(a, 0.f) & 0xffffu); }
__device__ __forceinline__ uint2 rne4(const b3_f32x4 v) { return make_uint2(rne2(v[0], v[1]), rne2(v[2], v[3])); }

template <int NP>
struct BP {
    bf16x8 p[NP];          // NP = 3: h, m, l;  NP = 1: the bf16 rounding
};
template <int NP>
__device__ __forceinline__ BP<NP> splitN(const b3_f32x4 a, const b3_f32x4 b) {
    BP<NP> r;
    if constexpr (NP == 3) {
        const B3 s = split3(a, b);
        r.p[0] = s.h; r.p[1] = s.m; r.p[2] = s.l;
    } else {
        const uint4 u = make_uint4(rne2(a[0], a[1]), rne2(a[2], a[3]), rne2(b[0], b[1]), rne2(b[2], b[3]));
        r.p[0] = __builtin_bit_cast(bf16x8, u);
    }
    return r;
}
template <int NP>
__device__ __forceinline__ b3_f32x4 mmaN(const BP<NP>& w, const BP<NP>& x, b3_f32x4 acc) {
    if constexpr (NP == 3) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[1], x.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x.p[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[2], x.p[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x.p[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[1], x.p[0], acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x.p[0], acc, 0, 0, 0);
}
// weight planes at pack time: NP = 3 the exact split, NP = 1 plane 0 = the bf16 rounding (planes 1, 2 unused: zero)
__device__ __forceinline__ void pack_planes(float x, int bf16_mode, short& h, short& m, short& l) {
    if (bf16_mode) { h = rne1(x); m = 0; l = 0; }
    else split1(x, h, m, l);
}

// The activation as bf16 PLANES for the LDS-DMA tiles of its consumers (ssd_convdma.hip): np = 3 the exact split
// x = h + m + l (planes h, m, l), np = 1 the bf16 rounding; `plane` elements between planes.  Inside a plane the layout is
// SLICE-MAJOR, [C / 32][P pixels][32 channels]: the 16 tile rows x 64 bytes one LDS-DMA wave-instruction copies are 1 KB of
// CONTIGUOUS memory for consecutive pixels (a [pixel][C] plane makes them 16 half-used 128-byte lines C * 2 bytes apart:
// tests/micro/lds_dma_rate.hip measures 35 - 48 GB/s per workgroup for that against 50 - 67 contiguous, and the conv tiles
// gained 12 - 19 %).  The packed weights' planes use the same scheme, [Kpad / 32][Npad][32] (plane_elem(n, k, Npad)).
// Four consecutive channels of one pixel = one 8-byte store per plane.
__device__ __host__ __forceinline__ long plane_elem(const long pix, const int c, const long P) {
    return (((long)(c >> 5) * P + pix) << 5) + (c & 31);
}
__device__ __forceinline__ void store_planes4(short* __restrict__ op, const long plane, const int np, const long pix, const int c,
                                              const long P, const b3_f32x4 v) {
    const long e = plane_elem(pix, c, P);
    if (np == 1) {
        *reinterpret_cast<uint2*>(op + e) = rne4(v);
    } else {
        uint2 h, m, l;
        split4(v, h, m, l);
        *reinterpret_cast<uint2*>(op + e) = h;
        *reinterpret_cast<uint2*>(op + plane + e) = m;
        *reinterpret_cast<uint2*>(op + 2 * plane + e) = l;
    }
}
// ... from the element index e = pix * C + c of the fp32 tensor (the elementwise producers walk that)
__device__ __forceinline__ void store_planes4_lin(short* __restrict__ op, const long plane, const int np, const long e, const int C,
                                                  const long P, const b3_f32x4 v) {
    const long pix = e / C;
    store_planes4(op, plane, np, pix, (int)(e - pix * C), P, v);
}

}  // namespace ssd

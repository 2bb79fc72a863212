// Micro-test (MI355X): fp32 products on the bf16 matrix cores through a 3-way split.
//   x = h + m + l exactly (h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = x - h - m: 8 + 8 + 8 significand bits),
//   x * y ~= hh + hm + mh + hl + lh + mm  (dropped: ml + lm + ll <= ~2^-23 |xy|)  -> 6 v_mfma_f32_16x16x32_bf16
// against 8 v_mfma_f32_16x16x4_f32 for the same 16 x 16 x 32 block.
// (1) accuracy of D = A(16x32) * B(32x16) vs float64, both paths; (2) cycles per block, alone and beside a VALU loop.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned xb = __float_as_uint(x[j]);
        const unsigned hb = xb & 0xffff0000u;
        const float r1 = x[j] - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(mb);
        h[j] = (short)(hb >> 16);
        m[j] = (short)(mb >> 16);
        l[j] = (short)(__float_as_uint(r2) >> 16);
    }
}

// A [16][32] row-major, B [16 cols][32 k] (k contiguous per column), D [16 rows m][16 cols n]
__global__ void check_kernel(const float* A, const float* B, float* D32, float* D16) {
    const int lane = threadIdx.x, l15 = lane & 15, g4 = lane >> 4;
    // fp32 path: 8 k-steps of 4; lane holds A[m = l15][k = s*4 + g4]... use the repo's convention: element s of the f32x4 at k = kb*16 + g4*4 + s
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < 2; ++kb)
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[l15 * 32 + kb * 16 + g4 * 4 + s], B[l15 * 32 + kb * 16 + g4 * 4 + s], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D32[(g4 * 4 + j) * 16 + l15] = acc[j];
    // bf16 x 3 path: lane holds k = g4*8 .. g4*8+7 of row / column l15 for BOTH operands
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = A[l15 * 32 + g4 * 8 + j]; b[j] = B[l15 * 32 + g4 * 8 + j]; }
    bf16x8 ah, am, al, bh, bm, bl;
    split3(a, ah, am, al);
    split3(b, bh, bm, bl);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D16[(g4 * 4 + j) * 16 + l15] = c[j];
}

// throughput: waves [0, nm) run MFMA blocks (mode 0: 8 fp32 MFMAs per block, 1: 6 bf16 MFMAs per block), waves [nm, 8) a v_fma loop
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, int nm, int valu_iters) {
    const int wave = threadIdx.x >> 6;
    if (wave < nm) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (short)(0x3f80 + threadIdx.x % 7); fb[j] = (short)(0x3f00 + j); }
        const float a32 = 1.0f + threadIdx.x * 1e-3f, b32 = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {           // 4 independent blocks per iteration
                if (MODE == 0) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int s = 0; s < 6; ++s) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
        for (int it = 0; it < valu_iters; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
        float sacc = 0.f;
        for (int j = 0; j < 8; ++j) sacc += v[j];
        out[blockIdx.x * 512 + threadIdx.x] = sacc;
    }
}

int main() {
    std::vector<float> A(512), B(512);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.7f;
    float *dA, *dB, *d32, *d16;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&d32, 1024); hipMalloc(&d16, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16);
    std::vector<float> r32(256), r16(256);
    hipMemcpy(r32.data(), d32, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(r16.data(), d16, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, e16 = 0, mx = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)A[m * 32 + k] * (double)B[n * 32 + k];
            e32 = fmax(e32, fabs(r32[m * 16 + n] - ref));
            e16 = fmax(e16, fabs(r16[m * 16 + n] - ref));
            mx = fmax(mx, fabs(ref));
        }
    printf("max |D| %.3f   max err fp32 MFMA %.3e   bf16x3 (6 MFMA) %.3e   (fp32 eps * max|D| = %.3e)\n", mx, e32, e16, mx * 1.19e-7);
    float* out;
    hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int cfg = 0; cfg < 3; ++cfg) {       // 0: 8 MFMA waves; 1: 4 MFMA + 4 VALU waves; 2: 4 MFMA waves + 4 idle
            const int nm = cfg == 0 ? 8 : 4, vi = cfg == 1 ? iters * 24 : 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(512), 0, 0, out, iters, nm, vi);
                else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(512), 0, 0, out, iters, nm, vi);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double blocks = 1024.0 * nm * iters * 4;            // 16x16x32 blocks
            printf("%s  %s: %.3f ms, %.1f TFLOP/s fp32-equivalent (%.2f ns per 16x16x32 block per CU-wave-slot)\n",
                   mode ? "bf16x3 (6 x 16x16x32_bf16)" : "fp32  (8 x 16x16x4_f32) ", cfg == 0 ? "8 MFMA waves          " : cfg == 1 ? "4 MFMA + 4 v_fma waves" : "4 MFMA waves alone    ",
                   ms, blocks * 16 * 16 * 32 * 2 / ms / 1e9, ms * 1e6 / (blocks / 256.0));
        }
    return 0;
}

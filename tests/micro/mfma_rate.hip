// Microbenchmark (diagnostics): issue rate of the fp32 MFMA shapes on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nacc, double flop_per_mfma, int blocks, int threads) {
        kern<<<blocks, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<blocks, threads>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double mf = (double)iters * nacc;
        printf("%-28s blocks %4d x %3d thr: %.1f cycles/MFMA/wave, %.1f TFLOP/s\n", name, blocks, threads,
               (double)h / mf, mf * flop_per_mfma * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
    };
    run("16x16x4 f32, 1 acc", k16<1>, 1, 2048, 256, 256);
    run("16x16x4 f32, 2 acc", k16<2>, 2, 2048, 256, 256);
    run("16x16x4 f32, 4 acc", k16<4>, 4, 2048, 256, 256);
    run("16x16x4 f32, 8 acc", k16<8>, 8, 2048, 256, 256);
    run("16x16x4 f32, 8 acc, 2/SIMD", k16<8>, 8, 2048, 512, 256);
    run("16x16x4 f32, 8 acc, 4/SIMD", k16<8>, 8, 2048, 1024, 256);
    run("32x32x2 f32, 1 acc", k32<1>, 1, 4096, 256, 256);
    run("32x32x2 f32, 2 acc", k32<2>, 2, 4096, 256, 256);
    run("32x32x2 f32, 4 acc", k32<4>, 4, 4096, 256, 256);
    run("32x32x2 f32, 4 acc, 2/SIMD", k32<4>, 4, 4096, 512, 256);
    return 0;
}

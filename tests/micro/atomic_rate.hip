// Diagnostics (round 6): what does it cost to sum TWO partial results per output element with fp32 L2 atomics
// (global_atomic_add_f32, no return) instead of two slab stores + a combine launch?  Shape of MobileNetV2 block 7 at B = 64:
// 64 images x 361 pixels x 64 channels = 1 478 656 floats (5.9 MB); 128 workgroups (2 groups x 64 images) of 512 threads,
// each lane holding 12 float4 (3 pixel tiles x 4 channel tiles) like the image kernel's epilogue.
//   hipcc --offload-arch=gfx950 -O3 tests/micro/atomic_rate.hip -o tests/micro/bin/atomic_rate && tests/micro/bin/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int B = 64, HW = 361, C = 64, G = 2;

__device__ __forceinline__ long elem(int img, int wave, int lane, int t, int ni) {
    const int l15 = lane & 15, g4 = lane >> 4;
    int q = (wave * 16 + l15) * 3 + t;            // adjacent pixels per lane (second form)
    if (q >= HW) q = HW - 1;
    return ((long)img * HW + q) * C + ni * 16 + g4 * 4;
}
__global__ __launch_bounds__(512) void slab_store(float* slabs, float v) {
    const int grp = blockIdx.x / B, img = blockIdx.x % B, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s = slabs + (long)grp * B * HW * C;
    for (int t = 0; t < 3; ++t)
        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f4*>(s + elem(img, wave, lane, t, ni)) = f4{v, v + 1, v + 2, v + 3};
}
__global__ __launch_bounds__(256) void combine(const float* slabs, const float* res, float* y, long nvec) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) {
        f4 v = *reinterpret_cast<const f4*>(slabs + e * 4) + *reinterpret_cast<const f4*>(slabs + (nvec + e) * 4) + *reinterpret_cast<const f4*>(res + e * 4);
        *reinterpret_cast<f4*>(y + e * 4) = v;
    }
}
__global__ __launch_bounds__(512) void atomic_sum(float* y, const float* res, float v) {
    const int grp = blockIdx.x / B, img = blockIdx.x % B, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l15 = lane & 15;
    const bool real = (wave * 16 + l15) * 3 + 2 < HW;
    for (int t = 0; t < 3; ++t)
        for (int ni = 0; ni < 4; ++ni) {
            const long e = elem(img, wave, lane, t, ni);
            f4 a = f4{v, v + 1, v + 2, v + 3};
            if (grp == 0) a = a + *reinterpret_cast<const f4*>(res + e);
            if (!real) continue;
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(y + e + k, a[k]);
        }
}
__global__ void zero(float* y, long nvec) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nvec; e += (long)gridDim.x * 256) *reinterpret_cast<f4*>(y + e * 4) = f4{0, 0, 0, 0};
}
int main() {
    const long n = (long)B * HW * C, nvec = n / 4;
    float *slabs, *y, *res;
    hipMalloc(&slabs, G * n * 4); hipMalloc(&y, n * 4); hipMalloc(&res, n * 4);
    hipMemset(res, 0, n * 4); hipMemset(y, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 5; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %7.2f us per pass\n", name, ms * 1000 / 200);
    };
    const int cb = (int)((nvec + 255) / 256);
    timeit("slab store (2 groups)", [&] { hipLaunchKernelGGL(slab_store, dim3(G * B), dim3(512), 0, 0, slabs, 1.0f); });
    timeit("combine launch", [&] { hipLaunchKernelGGL(combine, dim3(cb), dim3(256), 0, 0, slabs, res, y, nvec); });
    timeit("slab store + combine", [&] { hipLaunchKernelGGL(slab_store, dim3(G * B), dim3(512), 0, 0, slabs, 1.0f);
                                         hipLaunchKernelGGL(combine, dim3(cb), dim3(256), 0, 0, slabs, res, y, nvec); });
    timeit("zero launch", [&] { hipLaunchKernelGGL(zero, dim3(cb), dim3(256), 0, 0, y, nvec); });
    timeit("atomic sum (2 groups)", [&] { hipLaunchKernelGGL(atomic_sum, dim3(G * B), dim3(512), 0, 0, y, res, 1.0f); });
    timeit("zero + atomic sum", [&] { hipLaunchKernelGGL(zero, dim3(cb), dim3(256), 0, 0, y, nvec);
                                      hipLaunchKernelGGL(atomic_sum, dim3(G * B), dim3(512), 0, 0, y, res, 1.0f); });
    // determinism: two runs of zero + atomic sum give the same bits
    std::vector<float> a(n), b(n);
    hipLaunchKernelGGL(zero, dim3(cb), dim3(256), 0, 0, y, nvec); hipLaunchKernelGGL(atomic_sum, dim3(G * B), dim3(512), 0, 0, y, res, 0.3f);
    hipMemcpy(a.data(), y, n * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(zero, dim3(cb), dim3(256), 0, 0, y, nvec); hipLaunchKernelGGL(atomic_sum, dim3(G * B), dim3(512), 0, 0, y, res, 0.3f);
    hipMemcpy(b.data(), y, n * 4, hipMemcpyDeviceToHost);
    long diff = 0; for (long i = 0; i < n; ++i) diff += a[i] != b[i];
    printf("two runs differ in %ld of %ld elements\n", diff, n);
    return 0;
}

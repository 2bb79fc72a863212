// Microbenchmark (diagnostics): do fp32 MFMA work and fp32 VALU / LDS work of DIFFERENT waves on
// the same SIMD overlap on gfx950?  One 512-thread workgroup per CU: waves 0-3 run a pure
// v_mfma_f32_16x16x4_f32 loop, waves 4-7 run (a) v_fma_f32, (b) v_pk_fma_f32, (c) ds_read_b128
// loops.  Each role is timed alone and together (whole-kernel hipEvent time).
//   hipcc --offload-arch=gfx950 -O3 tests/micro/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bits: 1 = MFMA waves active, 2 = second role active; role: 0 v_fma, 1 v_pk_fma, 2 ds_read_b128
template <int ROLE>
__global__ __launch_bounds__(512) void k(float* out, int iters_mfma, int iters_b, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 0.5f;
    __syncthreads();
    const bool mf = threadIdx.x < 256;
    float s = 0;
    if (mf) {
        if (mode & 1) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
            float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
            for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    // mode & 8: idle the MFMA wave while its MFMA executes, so that its NEXT MFMA does
                    // not sit in the issue stage (does a waiting MFMA block the SIMD's other waves?)
                    if (mode & 8) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                }
            }
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (mode & 2) {
        if (mode & 4) __builtin_amdgcn_s_setprio(3);     // role waves above the MFMA waves
        if (ROLE == 0) {
            float v[16];
            for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
            const float m = 1.0001f, c = 0.5f;
            for (int it = 0; it < iters_b; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
            }
            for (int i = 0; i < 16; ++i) s += v[i];
        } else if (ROLE == 1) {
            f32x2 v[16];
            for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x, (float)i};
            const f32x2 m = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
            for (int it = 0; it < iters_b; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
            }
            for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
        } else {
            f32x4 t = {0, 0, 0, 0};
            const float* base = lds + (threadIdx.x & 255) * 4;
            for (int it = 0; it < iters_b; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x4 r = *reinterpret_cast<const volatile f32x4*>(base + i * 1024);
                    asm volatile("" ::"v"(r));          // consumed without VALU work
                }
            }
            s = t[0] + t[1] + t[2] + t[3];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](auto kern, int im, int ib, int mode) {
        kern<<<256, 512>>>(out, im, ib, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<256, 512>>>(out, im, ib, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    const int im = 4000;   // 32000 MFMAs per wave ~ 1.02 M cycles
    struct { const char* name; int iters; } roles[3] = {{"v_fma_f32 (16 chains)", 16000}, {"v_pk_fma_f32 (16 chains)", 16000}, {"ds_read_b128 x8", 16000}};
    for (int r = 0; r < 3; ++r) {
        float a, b, c;
        if (r == 0) { a = run(k<0>, im, roles[r].iters, 1); b = run(k<0>, im, roles[r].iters, 2); c = run(k<0>, im, roles[r].iters, 3); }
        else if (r == 1) { a = run(k<1>, im, roles[r].iters, 1); b = run(k<1>, im, roles[r].iters, 2); c = run(k<1>, im, roles[r].iters, 3); }
        else { a = run(k<2>, im, roles[r].iters, 1); b = run(k<2>, im, roles[r].iters, 2); c = run(k<2>, im, roles[r].iters, 3); }
        float d = r == 0 ? run(k<0>, im, roles[r].iters, 7) : r == 1 ? run(k<1>, im, roles[r].iters, 7) : run(k<2>, im, roles[r].iters, 7);
        float e = r == 0 ? run(k<0>, im, roles[r].iters, 9) : r == 1 ? run(k<1>, im, roles[r].iters, 9) : run(k<2>, im, roles[r].iters, 9);
        float f = r == 0 ? run(k<0>, im, roles[r].iters, 11) : r == 1 ? run(k<1>, im, roles[r].iters, 11) : run(k<2>, im, roles[r].iters, 11);
        printf("%-28s MFMA alone %.3f ms | role alone %.3f ms | together %.3f ms | together, role waves s_setprio(3) %.3f ms  (max %.3f, sum %.3f)\n"
               "%-28s MFMA + 24 idle cycles after each: alone %.3f ms | together %.3f ms\n",
               roles[r].name, a, b, c, d, a > b ? a : b, a + b, "", e, f);
    }
    return 0;
}

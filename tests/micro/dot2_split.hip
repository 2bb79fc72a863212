// Diagnostic: is x - float(h) through v_dot2c_f32_bf16 exact?  (h = truncated bf16 of x, packed as a pair)
//   hipcc --offload-arch=gfx950 -O3 tests/micro/dot2_split.hip -o tests/micro/bin/dot2_split && tests/micro/bin/dot2_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* r_ref, float* r_dot, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const float a = x[i], b = x[i + 1];
    const unsigned ha = __float_as_uint(a) & 0xffff0000u, hb = __float_as_uint(b) & 0xffff0000u;
    r_ref[i] = a - __uint_as_float(ha);
    r_ref[i + 1] = b - __uint_as_float(hb);
    const unsigned pair = __builtin_amdgcn_perm(hb, ha, 0x07060302u);       // (ha.hi16, hb.hi16): a in the low half
    const bf2 h = __builtin_bit_cast(bf2, pair);
    const bf2 sel_lo = __builtin_bit_cast(bf2, 0x0000bf80u);                // (-1, 0)
    const bf2 sel_hi = __builtin_bit_cast(bf2, 0xbf800000u);                // (0, -1)
    r_dot[i] = __builtin_amdgcn_fdot2_f32_bf16(h, sel_lo, a, false);
    r_dot[i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, sel_hi, b, false);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
        if (i % 7 == 0) u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 60) << 23);     // moderate exponents
        if (i % 1001 == 0) u &= 0x807fffffu;                                                // denormals
        float f; memcpy(&f, &u, 4);
        if (f != f || f - f != 0.f) f = 1.0f + i * 1e-7f;                                   // no NaN / inf
        x[i] = f;
    }
    float *dx, *dr, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, dr, dd, n);
    std::vector<float> r(n), d(n);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_norm = 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(&r[i], &d[i], 4)) {
            ++bad;
            unsigned u; memcpy(&u, &x[i], 4);
            const int e = (u >> 23) & 255;
            if (e > 24 && e < 250) { if (bad_norm++ < 5) printf("x=%a ref=%a dot=%a\n", x[i], r[i], d[i]); }
        }
    printf("mismatches %ld of %d (normal-range inputs: %ld)\n", bad, n, bad_norm);
    return 0;
}

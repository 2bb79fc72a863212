// Which CUs / XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) run on?  Every workgroup records the XCC id and the
// hardware CU id it ran on; masks tried: the first 64 bits, bits 64..127, every 4th group of 16, "XCD-interleaved" guesses.
// hipcc --offload-arch=gfx950 cu_mask_probe.hip -o cumask && ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hwid;
    }
    // stay resident a little so the grid spreads over every allowed CU
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 1024;
    unsigned* d; hipMalloc(&d, nb * 8); hipMemset(d, 0xff, nb * 8);
    hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per;      // xcc -> (se, cu) -> count
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per[h[2 * b]][(se << 8) | (sh << 4) | cu]++;
    }
    printf("%-28s:", name);
    int total = 0;
    for (auto& x : per) { printf(" xcc%u:%zu", x.first, x.second.size()); total += (int)x.second.size(); }
    printf("  -> %d distinct CUs\n", total);
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("CUs %d\n", pr.multiProcessorCount);
    std::vector<uint32_t> m(8, 0);
    auto setr = [&](int lo, int hi) { std::fill(m.begin(), m.end(), 0u); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); };
    setr(0, 256); run("all 256", m);
    setr(0, 64); run("bits 0..63", m);
    setr(64, 128); run("bits 64..127", m);
    setr(0, 32); run("bits 0..31", m);
    setr(0, 8); run("bits 0..7", m);
    std::fill(m.begin(), m.end(), 0u); for (int i = 0; i < 256; ++i) if (i % 8 < 2) m[i / 32] |= 1u << (i % 32); run("i % 8 < 2", m);
    std::fill(m.begin(), m.end(), 0u); for (int i = 0; i < 256; ++i) if (i % 4 == 0) m[i / 32] |= 1u << (i % 32); run("i % 4 == 0", m);
    std::fill(m.begin(), m.end(), 0u); for (int i = 0; i < 256; ++i) if ((i / 8) % 4 == 0) m[i / 32] |= 1u << (i % 32); run("(i / 8) % 4 == 0", m);
    return 0;
}

// How fast does a workgroup's LDS-DMA stream (`buffer_load_dwordx4 ... lds`, the operand path of ssd_convdma.hip) run, and what
// does it depend on?  Every workgroup repeats { all waves issue their share of a BURST of 1 KB wave-instructions; s_waitcnt
// vmcnt(0); s_barrier } -- the double-buffered tile loop without its matrix instructions -- over an L2-resident source.
//   pitch  bytes between the 16 rows of one wave-instruction (64: the 1 KB is contiguous; 128.. : 64 useful bytes per row, the
//          planes' [pixel][Cin] layout with Cin = pitch / 2 channels)
//   burst  KB per barrier interval and workgroup;  waves per workgroup;  workgroups per CU (by LDS size)
// hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o ldsdma_rate && ./ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_dst_t;

__global__ __launch_bounds__(1024) void stream_kernel(const char* src, long region, int pitch, int burst_kb, int iters, int lds_bytes) {
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* base = src + (long)blockIdx.x % 64 * region;          // 64 regions: L2-resident, shared like weights / neighbouring tiles
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
    const int per_wave = burst_kb / nw;                               // 1 KB instructions per wave and burst
    const int lane_off = (lane >> 2) * pitch + (lane & 3) * 16;      // row lane >> 2, 16-byte quad lane & 3
    const int step = 16 * pitch;                                      // bytes of source per instruction
    int pos = wave * per_wave * step;
    for (int it = 0; it < iters; ++it) {
        char* dst = sm + ((it & 1) * (lds_bytes / 2)) + wave * per_wave * 1024;
        for (int j = 0; j < per_wave; ++j) {
            int off = pos + j * step;
            if (off + step > (int)region) off -= (int)region / step * step;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_dst_t)(dst + j * 1024), 16, off + lane_off, 0, 0, 0);
        }
        pos += nw * per_wave * step;
        if (pos + per_wave * step > (int)region) pos = wave * per_wave * step;
        __syncthreads();
    }
}

int main() {
    const long region = 2 << 20;                   // 2 MB per region x 64 regions = 128 MB: Infinity-Cache resident; the L2 holds what is shared
    char* src;
    hipMalloc(&src, 64 * region);
    hipMemset(src, 1, 64 * region);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    printf("%-8s %-6s %-6s %-7s %-10s | %-12s %-12s %-10s\n", "pitch", "waves", "burst", "WG/CU", "regionKB", "us/burst", "GB/s per WG", "TB/s chip");
    for (long reg : {256L << 10, 2L << 20})
        for (int pitch : {64, 128, 256, 1024})
            for (int waves : {8, 16})
                for (int burst : {32, 64})
                    for (int wgcu : {1, 2}) {
                        const int lds = wgcu == 1 ? 2 * burst * 1024 + 8192 : 2 * burst * 1024;
                        if (wgcu == 2 && (lds > 80 * 1024 || waves * 64 * 2 > 2048)) continue;
                        if (wgcu == 1 && lds < 81 * 1024) {}        // one per CU is enforced through the grid size below
                        hipFuncSetAttribute((const void*)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        const int use_lds = wgcu == 1 ? 96 * 1024 > lds ? 96 * 1024 : lds : lds;    // > 80 KB: one workgroup per CU
                        const int grid = 256 * wgcu;
                        hipLaunchKernelGGL(stream_kernel, grid, waves * 64, use_lds, 0, src, reg, pitch, burst, 20, lds);
                        hipEventRecord(e0);
                        hipLaunchKernelGGL(stream_kernel, grid, waves * 64, use_lds, 0, src, reg, pitch, burst, iters, lds);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float ms = 0;
                        hipEventElapsedTime(&ms, e0, e1);
                        const double us = ms * 1e3 / iters;
                        printf("%-8d %-6d %-6d %-7d %-10ld | %-12.3f %-12.1f %-10.2f\n", pitch, waves, burst, wgcu, reg >> 10, us, burst * 1024 / us / 1e3,
                               burst * 1024.0 * grid / us / 1e6);
                    }
    return 0;
}

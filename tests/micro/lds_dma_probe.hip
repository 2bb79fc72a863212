// What does `buffer_load_dwordx4 ... offen lds` (LDS-DMA, gfx950) write for an OUT-OF-RANGE lane, and is the SGPR offset part
// of the range check?  (csrc/ssd_convdma.hip relies on: out-of-range lanes deposit ZEROS in their LDS slot.)
// hipcc --offload-arch=gfx950 lds_dma_probe.hip -o ldsdma && ./ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* g, int nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned smem[3 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 256; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(g), 0, nbytes, 0x00020000);
    // case A: lanes 5 and 17 read offset 2^31 (out of range)
    int off = lane * 16;
    if (lane == 5 || lane == 17) off = (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
    // case B: voffset in range, soffset pushes lanes >= 32 beyond num_records (nbytes = 1024 + 512)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 256), 16, lane * 16, 1024, 0, 0);
    // case C: voffset beyond num_records for lanes >= 32 on its own
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 512), 16, lane * 16 + 1024, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 3 * 256; i += 64) out[i] = smem[i];
}
int main() {
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
    unsigned *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 3 * 1024); hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, g, 1024 + 512, o);
    unsigned r[768]; hipMemcpy(r, o, 3072, hipMemcpyDeviceToHost);
    printf("A: in-range lane 4 -> %x (expect %x); OOB lane 5 -> %x %x %x %x, lane 17 -> %x (0 = zero-filled, deadbeef = untouched)\n",
           r[16], 0x1000 + 16, r[20], r[21], r[22], r[23], r[68]);
    printf("B (soffset 1024, num_records 1536): lane 31 -> %x (expect %x), lane 32 -> %x, lane 63 -> %x\n", r[256 + 124], 0x1000 + 256 + 124, r[256 + 128], r[256 + 252]);
    printf("C (voffset + 1024 itself):          lane 31 -> %x (expect %x), lane 32 -> %x, lane 63 -> %x\n", r[512 + 124], 0x1000 + 256 + 124, r[512 + 128], r[512 + 252]);
    return 0;
}

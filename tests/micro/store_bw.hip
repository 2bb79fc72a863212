// Microbenchmark (diagnostics): HBM write bandwidth vs store pattern on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// pattern: each wave store instruction writes `rows` rows x (64/rows lanes x 16 B) contiguous bytes,
// rows are `stride` floats apart (like the GEMM epilogue: 16 pixels x 64 B, stride = N channels)
template <int ROWS>
__global__ void wr(float* out, long rows_total, int stride, int seg_per_row) {
    // logical matrix [rows_total][stride]; a block handles 64 rows x all segments? keep simple:
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    constexpr int LPR = 64 / ROWS;                 // lanes per row
    const int r = lane / LPR, c = (lane % LPR) * 4;
    const long chunks_per_rowgroup = stride / (LPR * 4);
    const long total = (rows_total / ROWS) * chunks_per_rowgroup;
    for (long t = wave; t < total; t += nwaves) {
        const long rg = t / chunks_per_rowgroup, ch = t % chunks_per_rowgroup;
        float* p = out + (rg * ROWS + r) * stride + ch * (LPR * 4) + c;
        *reinterpret_cast<f32x4*>(p) = f32x4{1.f, 2.f, 3.f, (float)t};
    }
}
int main() {
    const long rows = 23104 * 4; const int stride = 576;       // 213 MB
    float* out; hipMalloc(&out, rows * stride * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        kern<<<2048, 256>>>(out, rows, stride, 0); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) kern<<<2048, 256>>>(out, rows, stride, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.2f TB/s\n", name, 10.0 * rows * stride * 4 / (ms * 1e-3) / 1e12);
    };
    run("16 rows x 64 B per store instr", wr<16>);
    run(" 8 rows x 128 B", wr<8>);
    run(" 4 rows x 256 B", wr<4>);
    run(" 2 rows x 512 B", wr<2>);
    run(" 1 row  x 1 KB (fully contiguous)", wr<1>);
    return 0;
}

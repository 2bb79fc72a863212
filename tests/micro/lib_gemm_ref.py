"""Reference point only (not a product path): fp32 GEMM rate of the ROCm BLAS library that ships
with torch on the 1x1-conv / head shapes of SSD300-MobileNetV2 at B=64, to put the hand-written
conv_mfma_kernel's TFLOP/s in context.  python tests/micro/lib_gemm_ref.py"""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [  # (name, M, K, N)
    ("block_7_expand", 23104, 64, 384), ("block_7_project", 23104, 384, 64),
    ("block_11_expand", 23104, 96, 576), ("block_11_project", 23104, 576, 96),
    ("block_14_expand", 6400, 160, 960), ("block_14_project", 6400, 960, 160),
    ("block_16_project", 6400, 960, 320), ("Conv_1", 6400, 320, 1280),
    ("extra1_1", 6400, 1280, 256), ("head1 (im2col K)", 23104, 5184, 100), ("head2 (im2col K)", 6400, 11520, 150),
]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(K, N, device="cuda")
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(10):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("%-20s M=%6d K=%6d N=%5d  %8.4f ms  %6.1f TF/s" % (name, M, K, N, best, 2.0 * M * K * N / best / 1e9))

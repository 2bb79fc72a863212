"""What does an RCCL all-reduce cost at WORLD SIZE 1 (the forced single-rank exchange of bench.py --train --force-dist)?
Times dist.all_reduce on the flat gradient's sizes (8.5 MB buckets, the whole 34 MB vector) against a plain device copy.
usage: python tests/micro/rccl_single_rank.py"""
import os
import socket
import time

import torch
import torch.distributed as dist

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for n in (2_123_420, 8_493_678):          # one of four buckets / the whole MobileNetV2-SSD gradient (fp32 elements)
    g = torch.randn(n, device="cuda")
    h = torch.empty_like(g)
    dist.all_reduce(g)
    torch.cuda.synchronize()
    for name, fn in (("all_reduce (RCCL, 1 rank)", lambda: dist.all_reduce(g)), ("device copy", lambda: h.copy_(g))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        host = (time.perf_counter() - t0) / 20
        dev = e0.elapsed_time(e1) / 20
        print("%-28s %5.1f MB: %.3f ms on the device per call (%.0f GB/s), %.3f ms host" % (name, n * 4e-6, dev, n * 4e-9 / (dev * 1e-3), host * 1e3))
dist.destroy_process_group()

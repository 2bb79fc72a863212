"""Multi-GPU plumbing for the inference path: one process per GPU, contiguous batch shards,
no collective in the data path.  ``torch.distributed`` (backend "nccl" == RCCL
on ROCm; "gloo" on CPU in the tests) is used only to gather the fixed-size results and for
timing barriers."""
import os

import torch
import torch.distributed as dist


def force_collectives():
    """``SSD_HIP_FORCE_DIST=1``: take the multi-rank code paths (communicator, collectives, bucketed exchange on the
    communication stream) whenever a process group is initialised, also at world size 1 -- how the N > 1 paths are
    exercised on a one-GPU box (``bench.py --force-dist``, ``tests/test_train.py``)."""
    return os.environ.get("SSD_HIP_FORCE_DIST", "0") == "1"


def _single():
    """True when no collective is needed: no process group, or one rank and collectives not forced."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and not force_collectives()


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world 1)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        if world == 1:           # forced single-rank group: torchrun did not provide the rendezvous
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
    return rank, local_rank, world


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for this rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_detections(boxes, labels, scores, n_total=None):
    """All-gather per-rank detections ([b,T,4], [b,T], [b,T]) in rank order -> full batch on
    every rank.  Shards may differ in size by one image (padded for the collective)."""
    if _single():
        return boxes, labels, scores
    world = dist.get_world_size()
    T = boxes.shape[1]
    packed = torch.cat([boxes.reshape(boxes.shape[0], T * 4), labels, scores], dim=1)      # [b, 6T]
    counts = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([packed.shape[0]], dtype=torch.int64, device=packed.device))
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = torch.zeros((mx, 6 * T), dtype=packed.dtype, device=packed.device)
    pad[:packed.shape[0]] = packed
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    full = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    if n_total is not None:
        assert full.shape[0] == n_total
    return full[:, :4 * T].reshape(-1, T, 4), full[:, 4 * T:5 * T], full[:, 5 * T:]


def max_over_ranks(value):
    """Max of a host float over all ranks (bench timing)."""
    if _single():
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- training data-parallel: batch-DP, one exchange step per optimiser
# step -- the SUM all-reduce of the flat gradient vector (8.53 M fp32 = 34.1 MB for
# MobileNetV2-SSD).  The native backward hands over ONE flat buffer in parameter-table order, so
# the natural collective is a single large all-reduce (xGMI is a point-to-point mesh: few large
# messages beat many small ones); `bucket_floats` splits it into contiguous buckets issued
# asynchronously back to back for callers that want to bound the message size.
def gradient_buckets(n, bucket_floats):
    """[(lo, hi)] contiguous, covering [0, n), each at most bucket_floats long (None: one bucket)."""
    if not bucket_floats or bucket_floats >= n:
        return [(0, n)] if n else []
    return [(lo, min(lo + int(bucket_floats), n)) for lo in range(0, n, int(bucket_floats))]


def allreduce_gradients(flat, bucket_floats=None):
    """In-place SUM all-reduce of the flat gradient tensor over the default process group.
    Returns the world size (the caller scales by 1/world in the optimiser: Keras' batch mean over
    the global batch for equal shards).  No-op (returns 1) without an initialised group."""
    if _single():
        return 1
    works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
             for lo, hi in gradient_buckets(flat.numel(), bucket_floats)]
    for w in works:
        w.wait()
    return dist.get_world_size()


def bucket_starts(n, n_buckets):
    """Starts of ``n_buckets`` equal contiguous buckets of a flat vector of ``n`` floats (multiples of 4 floats:
    16-byte aligned slices); fewer if ``n`` is small."""
    n_buckets = max(1, min(int(n_buckets), max(1, n // 4)))
    step = -(-n // n_buckets)
    step += (-step) % 4
    return [lo for lo in range(0, n, step)]


def allreduce_gradients_as_ready(flat, starts, wait_bucket=None, comm_stream=None):
    """SUM all-reduce of the flat gradient bucket by bucket IN THE ORDER THE BACKWARD FINISHES THEM: the native
    backward (``ssd_net_train_forward_backward``) walks the layers last to first and therefore completes the
    gradient vector from its END (heads and extras first, stem last); bucket k = ``flat[starts[k]:starts[k+1]]``
    is exchanged as soon as ``wait_bucket(k, comm_stream)`` has ordered ``comm_stream`` behind the bucket's
    completion event (``ssd_net_train_wait_bucket``), while the backward of the earlier layers is still running
    on the caller's stream.  xGMI is a point-to-point mesh: a handful of multi-megabyte buckets keeps every
    message large.  Returns the world size; the caller's stream waits for every bucket before returning."""
    if _single():
        return 1
    n = flat.numel()
    bounds = list(starts) + [n]
    works = []
    for k in reversed(range(len(starts))):
        piece = flat[bounds[k]:bounds[k + 1]]
        if wait_bucket is not None:
            wait_bucket(k, comm_stream)
        if comm_stream is not None:
            with torch.cuda.stream(comm_stream):      # RCCL's stream orders itself behind the CURRENT stream
                works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
        else:
            works.append(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return dist.get_world_size()


def mean_over_ranks(value):
    """Mean of a host float over all ranks (validation loss of a data-parallel fit)."""
    if _single():
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()

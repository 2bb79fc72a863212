"""ctypes binding of libssd_hip.so (include/ssd_hip.h) -- the only door from the Python
host code to the HIP kernels.  PyTorch-ROCm is used for device memory and streams only.

There is deliberately NO CPU fallback: if the shared library is missing, or no gfx950
device is visible, every compute entry point raises.
"""
import ctypes
import os
import sys


def configure_serving(lanes=3):
    """OPT IN to the measured serving setup: ``lanes`` in-order lanes, each on its own hardware queue,
    i.e. the HIP runtime limited to that many queues.  The runtime reads GPU_MAX_HW_QUEUES when it starts, so this has to
    run before the first device call of the process (``predictor.py`` and ``bench.py`` do it first thing); it changes
    nothing when the process already chose a value, and answers False when the runtime is already up (too late: one
    lane is used, ``models.decoder.default_lanes()`` answers 1).  Process-wide consequences, which is why importing the
    package no longer does this on its own (round 6): EVERY HIP stream of the process then shares ``lanes`` queues (a
    training step's main / weight-gradient / RCCL streams included), hipGraph replay of the forked one-at-a-time step is off
    below four queues, and ``get_decoder_model`` builds ``lanes`` replicas of the net (arena + bf16 planes each) once a
    ``predict`` call is long enough to use them."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return os.environ["GPU_MAX_HW_QUEUES"] == str(lanes)
    t = sys.modules.get("torch")
    try:
        if t is not None and t.cuda.is_initialized():
            return False
    except Exception:
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(lanes))
    return True


# the same opt-in through the environment: SSD_HIP_HW_QUEUES=<n> (nothing is touched when it is unset)
if os.environ.get("SSD_HIP_HW_QUEUES", "").isdigit() and int(os.environ["SSD_HIP_HW_QUEUES"]) > 0:
    configure_serving(int(os.environ["SSD_HIP_HW_QUEUES"]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSD_HIP_LIBRARY") or os.path.join(_HERE, "libssd_hip.so")   # override: diagnostic builds

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """struct ssd_conv_desc (include/ssd_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "B", "H", "W", "Cin", "Cout", "kh", "kw", "stride", "dilation",
        "pad_t", "pad_l", "pad_b", "pad_r", "act", "has_residual")]


ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
MOBILENET_V2, VGG16 = 0, 1

# name -> (restype, argtypes); every symbol include/ssd_hip.h declares.
_SIGNATURES = {
    "ssd_version": (ctypes.c_char_p, []),
    "ssd_last_error": (ctypes.c_char_p, []),
    "ssd_init": (ctypes.c_int, [ctypes.c_int]),
    "ssd_priors_count": (ctypes.c_int, [c_int_p, c_int_p, ctypes.c_int]),
    "ssd_priors": (ctypes.c_int, [c_int_p, ctypes.POINTER(c_float_p), c_int_p, ctypes.c_int, vp, vp]),
    "ssd_decode_boxes": (ctypes.c_int, [vp, vp, c_float_p, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_decode_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "ssd_decode_nms": (ctypes.c_int, [vp, vp, vp, c_float_p] + [ctypes.c_int] * 5 +
                       [ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "ssd_combined_nms": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 +
                         [ctypes.c_float, ctypes.c_float, ctypes.c_int, vp, vp, vp, vp, vp, vp,
                          ctypes.c_size_t, vp]),
    "ssd_iou_map": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_encode_deltas": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_match_encode": (ctypes.c_int, [vp, vp, vp, c_float_p, ctypes.c_float] + [ctypes.c_int] * 4 +
                         [vp, vp, vp, vp, vp]),
    "ssd_preprocess": (ctypes.c_int, [vp] + [ctypes.c_int] * 6 + [vp, vp]),
    "ssd_image_mean": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp, vp]),
    "ssd_augment_geometry": (ctypes.c_int, [vp] + [ctypes.c_int] * 6 + [vp, vp, vp, vp]),
    "ssd_augment_color": (ctypes.c_int, [vp] + [ctypes.c_int] * 3 + [vp, vp, vp, vp]),
    "ssd_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "ssd_loss": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                ctypes.c_float, vp, vp, vp, vp, vp, vp, ctypes.c_float, vp, ctypes.c_size_t, vp]),
    "ssd_same_pads": (ctypes.c_int, [ctypes.c_int] * 4 + [c_int_p, c_int_p]),
    "ssd_conv_out_size": (ctypes.c_int, [ctypes.c_int] * 6),
    "ssd_conv_packed_weight_floats": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "ssd_conv_pack_weights": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp]),
    "ssd_conv2d": (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp,
                                  ctypes.c_long, ctypes.c_long, vp]),
    "ssd_conv_num_configs": (ctypes.c_int, []),
    "ssd_conv_config_name": (ctypes.c_char_p, [ctypes.c_int]),
    "ssd_conv2d_ex": (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp,
                                     ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_split_planes": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, ctypes.c_long, vp]),
    "ssd_join_planes": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_long, vp, vp]),
    "ssd_conv2d_planes": (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, ctypes.c_int, ctypes.c_long, vp, vp, vp, vp, vp,
                                         ctypes.c_long, ctypes.c_long, vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_conv_wino_weight_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "ssd_conv_wino_pack_weights": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_conv_wino_num_configs": (ctypes.c_int, []),
    "ssd_conv2d_wino": (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, ctypes.c_long, ctypes.c_long,
                                       ctypes.c_int, ctypes.c_int, vp, vp]),
    "ssd_dwconv3x3": (ctypes.c_int, [vp] + [ctypes.c_int] * 9 + [vp, vp, vp, ctypes.c_int, vp, vp]),
    "ssd_maxpool2d": (ctypes.c_int, [vp] + [ctypes.c_int] * 10 + [vp, vp]),
    "ssd_l2norm": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, vp, vp, vp]),
    "ssd_softmax": (ctypes.c_int, [vp, ctypes.c_long, ctypes.c_int, vp, vp]),
    "ssd_net_create": (vp, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_int_p, ctypes.c_int]),
    "ssd_net_destroy": (None, [vp]),
    "ssd_net_num_params": (ctypes.c_int, [vp]),
    "ssd_net_param_name": (ctypes.c_char_p, [vp, ctypes.c_int]),
    "ssd_net_param_rank": (ctypes.c_int, [vp, ctypes.c_int]),
    "ssd_net_param_shape": (c_int_p, [vp, ctypes.c_int]),
    "ssd_net_set_param": (ctypes.c_int, [vp, ctypes.c_char_p, c_float_p, ctypes.c_size_t]),
    "ssd_net_get_param": (ctypes.c_int, [vp, ctypes.c_char_p, c_float_p, ctypes.c_size_t]),
    "ssd_net_finalize": (ctypes.c_int, [vp, ctypes.c_int]),
    "ssd_net_get_tuning": (ctypes.c_long, [vp, ctypes.c_char_p, ctypes.c_size_t]),
    "ssd_net_set_tuning": (ctypes.c_int, [vp, ctypes.c_char_p]),
    "ssd_net_tuning_stats": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "ssd_net_memory_bytes": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_size_t)]),
    "ssd_build_id": (ctypes.c_char_p, []),
    "ssd_stream_create": (vp, [ctypes.c_int]),
    "ssd_stream_create_masked": (vp, [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]),
    "ssd_stream_destroy": (ctypes.c_int, [vp]),
    "ssd_net_regularization_loss": (ctypes.c_int, [vp, c_float_p]),
    "ssd_net_train_set_buckets": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_long)]),
    "ssd_net_train_wait_bucket": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "ssd_net_num_priors": (ctypes.c_int, [vp]),
    "ssd_net_feature_map_size": (ctypes.c_int, [vp, ctypes.c_int]),
    "ssd_net_forward": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp, vp]),
    "ssd_net_predict": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, c_float_p, ctypes.c_int,
                                       ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp]),
    "ssd_net_fetch_activation": (ctypes.c_long, [vp, ctypes.c_char_p, c_float_p, ctypes.c_size_t]),
    "ssd_net_fetch_planes": (ctypes.c_long, [vp, ctypes.c_char_p, c_float_p, ctypes.c_size_t, c_int_p]),
    "ssd_net_num_layers": (ctypes.c_int, [vp]),
    "ssd_net_layer_name": (ctypes.c_char_p, [vp, ctypes.c_int]),
    "ssd_net_layer_kind": (ctypes.c_char_p, [vp, ctypes.c_int]),
    "ssd_net_layer_config": (ctypes.c_char_p, [vp, ctypes.c_int]),
    "ssd_net_layer_flops": (ctypes.c_double, [vp, ctypes.c_int, ctypes.c_int]),
    "ssd_net_layer_bytes": (ctypes.c_double, [vp, ctypes.c_int, ctypes.c_int]),
    "ssd_net_layer_executed_flops": (ctypes.c_double, [vp, ctypes.c_int, ctypes.c_int]),
    "ssd_net_set_option": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_int]),
    "ssd_net_profile_fused": (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "ssd_net_set_timing": (ctypes.c_int, [vp, ctypes.c_int]),
    "ssd_net_read_timing": (ctypes.c_int, [vp, c_float_p, c_int_p]),
    "ssd_net_profile_layers": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, c_float_p, vp]),
    "ssd_net_train_begin": (ctypes.c_int, [vp, ctypes.c_int]),
    "ssd_net_trainable_floats": (ctypes.c_size_t, [vp]),
    "ssd_net_trainable_offset": (ctypes.c_long, [vp, ctypes.c_char_p]),
    "ssd_net_train_forward_backward": (ctypes.c_int, [vp, vp, ctypes.c_int, vp, vp, ctypes.c_float, ctypes.c_float,
                                                      vp, vp, vp, vp]),
    "ssd_net_adam_step": (ctypes.c_int, [vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, vp]),
    "ssd_net_train_steps": (ctypes.c_long, [vp]),
    "ssd_net_train_matrix_flops": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double)]),
    "ssd_net_train_fetch": (ctypes.c_long, [vp, ctypes.c_char_p, ctypes.c_int, c_float_p, ctypes.c_size_t]),
}

_lib = None
_inited = False
_workspaces = {}


class SsdHipError(RuntimeError):
    pass


def lib():
    """Load libssd_hip.so (no GPU needed for loading).  Raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsdHipError(
                "libssd_hip.so is missing (%s). Build it with tf-ssd_amd/csrc/build.sh or "
                "__graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)    # AttributeError here == the .so is stale: rebuild it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ssd_last_error().decode()
        if rc == -1:
            raise ValueError("%s: %s" % (what, msg))
        raise SsdHipError("%s failed (%d): %s" % (what, rc, msg))


def device():
    """The torch device of this process (one process per GPU: LOCAL_RANK selects it)."""
    global _inited
    if not torch.cuda.is_available():
        raise SsdHipError("no HIP device visible: the SSD kernels need an MI355X (gfx950); "
                          "there is no CPU fallback")
    idx = torch.cuda.current_device()
    if not _inited:
        check(lib().ssd_init(idx), "ssd_init")
        _inited = True
    return torch.device("cuda", idx)


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)


_stream_pool = {False: [], True: []}


def new_stream(high_priority=False):
    """A non-blocking native stream (``ssd_stream_create``) as a torch stream object: not ordered against
    the legacy NULL stream, unlike ``torch.cuda.Stream()`` on this stack (DecoderModel lanes).  Streams come from a
    process-wide pool and go back to it with ``free_stream``: they are never destroyed while the process lives --
    torch's caching allocator keeps references to every stream a tensor was ``record_stream``-ed on and touches them
    when it recycles the block, so destroying a lane's stream crashes a later allocation (or the interpreter's exit).
    Reuse bounds the number of native streams by the largest number ever in use at once."""
    device()
    hp = bool(high_priority)
    if _stream_pool[hp]:
        return _stream_pool[hp].pop()
    h = lib().ssd_stream_create(1 if hp else 0)
    if not h:
        raise SsdHipError("ssd_stream_create: %s" % lib().ssd_last_error().decode())
    st = torch.cuda.ExternalStream(h)
    st._ssd_high_priority = hp
    return st


def new_masked_stream(cu_bits):
    """A native stream restricted to the compute units in ``cu_bits`` (iterable of CU bit indices of
    ``hipExtStreamCreateWithCUMask``).  Not pooled (the mask is part of the stream); never destroyed, like the others."""
    device()
    words = [0] * 8
    for b in cu_bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint * 8)(*words)
    hnd = lib().ssd_stream_create_masked(arr, 8)
    if not hnd:
        raise SsdHipError("ssd_stream_create_masked: %s" % lib().ssd_last_error().decode())
    st = torch.cuda.ExternalStream(hnd)
    st._ssd_high_priority = None
    return st


def free_stream(st):
    """Return a stream made by ``new_stream`` to the pool (idempotent per stream object)."""
    hp = getattr(st, "_ssd_high_priority", None)
    if hp is None or any(st is q for q in _stream_pool[hp]):
        return
    _stream_pool[hp].append(st)


def to_dev(x, dtype=torch.float32):
    """numpy / list / torch (any device) -> contiguous tensor on the GPU."""
    dev = device()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x)), dtype=dtype).to(dev).contiguous()


def pinned_empty(shape, dtype=torch.float32):
    """A page-locked host tensor (``t.numpy()`` is a zero-copy NumPy view to fill): batches handed to
    ``DecoderModel.submit`` / ``predict`` in such buffers are copied to the GPU by the lane's own stream (asynchronous DMA
    beside the other lanes' kernels) instead of through the staged, blocking copy pageable memory takes."""
    device()
    return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)


def ptr(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


def host4(v):
    a = (ctypes.c_float * 4)(*[float(x) for x in v])
    return ctypes.cast(a, c_float_p), a


def workspace(nbytes):
    """Grow-only scratch buffer per device (caller-owned memory, as the ABI requires)."""
    dev = device()
    key = dev.index
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws

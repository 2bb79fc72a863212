"""Shared model object behind ``get_model`` of both backbones: a thin Python handle on the
native graph runner (``ssd_net_*`` in include/ssd_hip.h).  PyTorch is the device-memory /
stream provider only; weights live in the native net, keyed by the Keras variable names
(``<layer>/<variable>``) in Keras layouts."""
import ctypes
import os

import numpy as np
import torch

import ssd_hip as _h

# options that change which kernel configurations finalize may choose (the memo is per option set)
_TABLE_OPTIONS = ("use_wino", "image_split", "conv_dma")
# options that make the native net drop `finalized` (ssd_net_set_option): the Python handle must re-finalize too
_REFINALIZE_OPTIONS = ("precision", "image_split", "use_wino", "conv_dma")
_STALE_WARNED = set()


class SSDModel(object):
    """Callable like the Keras model the reference builds: ``model(images) ->
    (pred_deltas [B,N,4], pred_labels [B,N,L])``; also ``predict``, ``load_weights``,
    ``save_weights``, ``get_weights``, ``set_weights``."""

    def __init__(self, backbone, hyper_params, max_batch=None, precision="fp32"):
        """``precision``: "fp32" (default: the reference's arithmetic, fp32 results everywhere) or "bf16" (this
        build's extension for BASELINE.json configs[3] / [4]: every matrix operand of the dense / 1x1 convolutions is
        rounded once to bf16, one bf16 MFMA per product, fp32 accumulation; BatchNorm shifts, activations, residual
        adds, depthwise taps, softmax and the box math stay fp32)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
        self.precision = precision
        self.backbone = backbone
        self.hyper_params = hyper_params
        self.img_size = int(hyper_params["img_size"])
        self.total_labels = int(hyper_params["total_labels"])
        n_ars = [len(a) for a in hyper_params["aspect_ratios"]]
        arr = (ctypes.c_int * len(n_ars))(*n_ars)
        lib = _h.lib()
        self._net = lib.ssd_net_create(_h.MOBILENET_V2 if backbone == "mobilenet_v2" else _h.VGG16,
                                       self.img_size, len(n_ars), arr, self.total_labels)
        if not self._net:
            raise ValueError("ssd_net_create: %s" % lib.ssd_last_error().decode())
        self.param_specs = []
        for i in range(lib.ssd_net_num_params(self._net)):
            name = lib.ssd_net_param_name(self._net, i).decode()
            rank = lib.ssd_net_param_rank(self._net, i)
            shp = lib.ssd_net_param_shape(self._net, i)
            self.param_specs.append((name, tuple(shp[j] for j in range(rank))))
        self.num_priors = lib.ssd_net_num_priors(self._net)
        fm = [lib.ssd_net_feature_map_size(self._net, l) for l in range(len(n_ars))]
        if list(fm) != [int(f) for f in hyper_params["feature_map_shapes"]]:
            raise ValueError("feature_map_shapes %s do not match the %s graph at img_size %d (%s)" % (
                hyper_params["feature_map_shapes"], backbone, self.img_size, fm))
        self._max_batch = max_batch
        self._finalized_for = 0
        self._weights_set = False
        self._weights_version = 0          # bumped by every weight change (lane replicas follow it)
        self._options = {}
        if precision == "bf16":
            self.set_option("precision", 1)

    def __del__(self):
        try:
            if getattr(self, "_comm_stream", None) is not None:
                _h.free_stream(self._comm_stream)
                self._comm_stream = None
            if getattr(self, "_net", None):
                _h.lib().ssd_net_destroy(self._net)
                self._net = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights):
        """weights: dict name -> array in the Keras layout (missing names keep their value)."""
        lib = _h.lib()
        _h.device()
        shapes = dict(self.param_specs)
        for name, value in weights.items():
            if name not in shapes:
                raise ValueError("unknown parameter %r" % name)
            a = np.ascontiguousarray(np.asarray(value, dtype=np.float32))
            if tuple(a.shape) != shapes[name]:
                raise ValueError("parameter %r has shape %s, expected %s" % (name, a.shape, shapes[name]))
            _h.check(lib.ssd_net_set_param(self._net, name.encode(), a.ctypes.data_as(_h.c_float_p), a.size),
                     "set_weights")
        self._weights_set = True
        self._finalized_for = 0
        self._weights_version += 1

    def clone(self):
        """An independent replica (own native net: arena, scratch, streams) with the same weights, options
        and tile tuning -- a second LANE for ``DecoderModel`` to keep two batches in flight."""
        m = SSDModel(self.backbone, self.hyper_params, self._max_batch, precision=self.precision)
        m.set_weights(self.get_weights())
        for k, v in self._options.items():
            m.set_option(k, v)
        if self._finalized_for:
            m._tuning_explicit = self.get_tuning()     # the replica runs exactly the parent's kernels
        elif getattr(self, "_tuning_explicit", None) is not None:
            m._tuning_explicit = self._tuning_explicit
        return m

    def get_weights(self):
        lib = _h.lib()
        out = {}
        for name, shape in self.param_specs:
            a = np.empty(shape, np.float32)
            _h.check(lib.ssd_net_get_param(self._net, name.encode(), a.ctypes.data_as(_h.c_float_p), a.size),
                     "get_weights")
            out[name] = a
        return out

    def layer_order(self):
        """Weighted layers in the order of Keras' ``model.layers`` -- what ``load_weights(by_name=False)``
        (the reference's call, trainer.py:48 / predictor.py:46) pairs position by position with the
        file's ``layer_names``.  [3P, restated from the Keras functional-API source, not executable here]
        ``Network._map_graph_network`` sorts layers by DEPTH (longest path to an output, descending) and
        breaks ties by the pre-order of a depth-first walk from ``outputs = [pred_deltas, pred_labels]``
        (models/ssd_*.py: ``Model(inputs, [pred_deltas, pred_labels])``; models/header.py:60-66):
        backbone + extras are one chain (depth strictly decreasing); ``{i}_conv_label_output`` sits at depth
        2 (-> labels_head -> conf), ``{i}_conv_boxes_output`` at depth 1 (-> loc), so ALL label convs come
        before ALL box convs; VGG16's ``l2_normalization`` (depth 3, reached first in the walk through
        1_conv_boxes_output) ties with ``conv11_2`` and precedes it.  Loading ``by_name=True`` on the
        Keras side does not depend on any of this."""
        table = []
        for name, _ in self.param_specs:
            l = name.rsplit("/", 1)[0]
            if l not in table:
                table.append(l)
        chain = [l for l in table if not l[0].isdigit() and l != "l2_normalization"]
        if "l2_normalization" in table:
            chain.insert(chain.index("conv11_2"), "l2_normalization")
        labels = sorted((l for l in table if l.endswith("_conv_label_output")), key=lambda n: int(n.split("_")[0]))
        boxes = sorted((l for l in table if l.endswith("_conv_boxes_output")), key=lambda n: int(n.split("_")[0]))
        return chain + labels + boxes

    def save_weights(self, path):
        """Keras ``Model.save_weights`` (reference trainer.py:65 via ModelCheckpoint,
        utils/io_utils.py:17-29).  ``*.h5`` / ``*.hdf5`` / ``*.keras``: a real HDF5 file in the
        Keras layout (utils/h5_writer.py; opens in h5py / Keras ``load_weights``); ``*.npz``: NumPy
        archive keyed by ``<layer>/<variable>``."""
        w = self.get_weights()
        if str(path).endswith(".npz"):
            with open(path, "wb") as f:
                np.savez(f, **w)
            return
        from utils import h5_writer
        h5_writer.save_keras_weights(path, w, layer_order=self.layer_order())

    def load_weights(self, path, by_name=False):
        """Keras ``Model.load_weights`` (reference predictor.py:46, trainer.py:48): the container
        is recognised by its magic -- HDF5 (a Keras checkpoint, e.g. one trained by the
        reference; pure-Python reader utils/h5_reader.py) or NumPy ``.npz``.  Without ``by_name``
        the file must hold exactly this model's variables (Keras raises on a topology mismatch);
        with ``by_name`` only the layers present in both are loaded."""
        with open(path, "rb") as f:
            magic = f.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            from utils import h5_reader
            w = h5_reader.load_keras_weights(path)
        elif magic[:2] == b"PK":
            with np.load(path) as z:
                w = {k: z[k] for k in z.files}
        else:
            raise ValueError("%s is neither an HDF5 (Keras) nor an .npz weights file" % path)
        shapes = dict(self.param_specs)
        if by_name:
            w = {k: v for k, v in w.items() if k in shapes}
        else:
            missing = [k for k in shapes if k not in w]
            extra = [k for k in w if k not in shapes]
            if missing or extra:
                raise ValueError("weights file %s does not match the %s graph: %d missing (e.g. %s), %d unexpected "
                                 "(e.g. %s); use by_name=True for a partial load" % (
                                     path, self.backbone, len(missing), missing[:2], len(extra), extra[:2]))
        self.set_weights(w)

    # ------------------------------------------------------------------ execution
    def _ensure(self, B):
        if not self._weights_set:
            raise RuntimeError("model has no weights: call set_weights()/load_weights() first")
        want = max(B, self._max_batch or 0)
        if self._finalized_for < want:
            self._finalize(want)

    def _finalize(self, want):
        """``ssd_net_finalize`` with the kernel-choice table resolved as tuning.py describes (explicit
        table > SSD_HIP_TUNE_CACHE > shipped table > process memo > on-device autotune on a miss)."""
        import tuning
        lib = _h.lib()
        key = tuning.table_key(self.backbone, self.img_size, self.total_labels, self.hyper_params["aspect_ratios"], want)
        if self.precision != "fp32":
            key += "_" + self.precision           # the bf16 mode has its own kernel families and its own tables
        opts = tuning.options_key({k: v for k, v in self._options.items() if k in _TABLE_OPTIONS})
        source, text, path = "autotune", None, None
        explicit = getattr(self, "_tuning_explicit", None)
        cache = os.environ.get("SSD_HIP_TUNE_CACHE")
        if cache:
            # only valid for this library build and this device (the marketing name is not stable -- it
            # reads "" under rocprofv3 -- the ISA name is)
            props = torch.cuda.get_device_properties(torch.cuda.current_device())
            dev = "%s_cu%d" % (str(getattr(props, "gcnArchName", "gpu")).split(":")[0], props.multi_processor_count)
            path = os.path.join(cache, "%s_%s_%s%s.tune" % (key, dev, lib.ssd_build_id().decode(), ("_" + opts) if opts else ""))
        if explicit is not None:
            source, text = "explicit", explicit
        elif path and os.path.exists(path):
            with open(path) as f:
                source, text = "cache", f.read()
        elif not opts and os.environ.get("SSD_HIP_IGNORE_SHIPPED", "0") != "1" and tuning.load_shipped(key) is not None:
            source, text = "shipped", tuning.load_shipped(key)
        elif tuning.memo_get(key, opts) is not None:
            source, text = "memo", tuning.memo_get(key, opts)
        _h.check(lib.ssd_net_set_tuning(self._net, tuning.body(text or "").encode()), "ssd_net_set_tuning")
        if text is None and os.environ.get("SSD_HIP_AUTOTUNE", "1") == "0":
            raise RuntimeError("no tuning table for %s and SSD_HIP_AUTOTUNE=0 (tables: %s)" % (key, tuning.SHIPPED_DIR))
        _h.check(lib.ssd_net_finalize(self._net, want), "ssd_net_finalize")
        self._finalized_for = want
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        _h.check(lib.ssd_net_tuning_stats(self._net, ctypes.byref(a), ctypes.byref(b)), "ssd_net_tuning_stats")
        table = self.get_tuning()
        hdr = tuning.header(text or "")
        this_build = lib.ssd_build_id().decode()
        # a table measured on another build of the kernels still pins VALID choices (same bits in every process), but
        # they may no longer be the fastest: say so instead of reporting it as current
        stale = bool(text) and hdr.get("build") not in (None, this_build)
        self.tuning_info = {"key": key, "source": source, "table_sha16": tuning.sha16(table), "layers_from_table": a.value,
                            "choices_timed_on_device": b.value, "reproducible": b.value == 0,
                            "table_build": hdr.get("build"), "this_build": this_build, "stale": stale}
        if stale and os.environ.get("SSD_HIP_WARN_STALE_TABLE", "1") != "0" and key not in _STALE_WARNED:
            _STALE_WARNED.add(key)
            import warnings
            warnings.warn("kernel table %s (%s) was measured on build %s, this library is %s: choices stay valid and "
                          "reproducible but may be slower than a re-measured table (tools/make_tuning_tables.py)" % (
                              key, source, hdr.get("build"), this_build))
        if explicit is None:
            tuning.memo_put(key, opts, table)
        if path and not os.path.exists(path):
            # one process per GPU may race on the same file: write privately, publish atomically
            os.makedirs(cache, exist_ok=True)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "w") as f:
                f.write(tuning.with_header(table, key=key, build=lib.ssd_build_id().decode()))
            os.replace(tmp, path)

    def memory_summary(self):
        """Device bytes the finalized net holds (``ssd_net_memory_bytes``): activation arena, the bf16 planes its chosen
        LDS-DMA tiles read (none where the table names no such tile), whole-image slabs, split-K slabs.  Every lane
        replica of ``get_decoder_model(..., lanes=n)`` holds the same again."""
        out = (ctypes.c_size_t * 4)()
        _h.check(_h.lib().ssd_net_memory_bytes(self._net, out), "ssd_net_memory_bytes")
        return {"arena": int(out[0]), "planes": int(out[1]), "image_slabs": int(out[2]), "splitk_slabs": int(out[3])}

    def set_tuning(self, text):
        """Pin the kernel choices (a table from ``get_tuning()`` of another instance, or a file): the next
        finalize uses it instead of the shipped table / autotune.  ``None`` returns to the default."""
        self._tuning_explicit = text
        self._finalized_for = 0

    def __call__(self, images):
        x = _h.to_dev(images)
        if x.dim() != 4 or x.shape[1] != self.img_size or x.shape[2] != self.img_size or x.shape[3] != 3:
            raise ValueError("expected images [B,%d,%d,3], got %s" % (self.img_size, self.img_size, tuple(x.shape)))
        B = x.shape[0]
        self._ensure(max(B, 1))
        dev = x.device
        deltas = torch.empty((B, self.num_priors, 4), dtype=torch.float32, device=dev)
        probs = torch.empty((B, self.num_priors, self.total_labels), dtype=torch.float32, device=dev)
        _h.check(_h.lib().ssd_net_forward(self._net, _h.ptr(x), B, _h.ptr(deltas), _h.ptr(probs), _h.stream()),
                 "ssd_net_forward")
        self._last_input = x   # keep the aliased input alive until the next call
        return deltas, probs

    def predict(self, x, batch_size=32, steps=None, verbose=0):
        outs = ([], [])
        n = x.shape[0]
        for i in range(0, n, batch_size):
            d, p = self(x[i:i + batch_size])
            outs[0].append(d.cpu().numpy())
            outs[1].append(p.cpu().numpy())
        return np.concatenate(outs[0], 0), np.concatenate(outs[1], 0)

    def get_tuning(self):
        lib = _h.lib()
        n = lib.ssd_net_get_tuning(self._net, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        lib.ssd_net_get_tuning(self._net, buf, n + 1)
        return buf.value.decode()

    def fetch_activation(self, name):
        """Activation of a named layer of the last forward (debug / parity tests)."""
        lib = _h.lib()
        n = lib.ssd_net_fetch_activation(self._net, name.encode(), None, 0)
        if n < 0:
            raise ValueError(lib.ssd_last_error().decode())
        a = np.empty((n,), np.float32)
        got = lib.ssd_net_fetch_activation(self._net, name.encode(), a.ctypes.data_as(_h.c_float_p), a.size)
        if got < 0:
            raise RuntimeError(lib.ssd_last_error().decode())
        return a

    def fetch_planes(self, name):
        """The bf16 planes of a named activation of the last forward (what the LDS-DMA conv tiles of its consumers read),
        joined back to fp32: ``(array, planes)`` with planes 3 (exact split) or 1 (bf16 rounding); ``(None, planes)`` when
        no running layer asked for them."""
        lib = _h.lib()
        npl = ctypes.c_int(0)
        n = lib.ssd_net_fetch_planes(self._net, name.encode(), None, 0, ctypes.byref(npl))
        if n < 0:
            raise ValueError(lib.ssd_last_error().decode())
        if n == 0:
            return None, npl.value
        a = np.empty((n,), np.float32)
        got = lib.ssd_net_fetch_planes(self._net, name.encode(), a.ctypes.data_as(_h.c_float_p), a.size, ctypes.byref(npl))
        if got < 0:
            raise RuntimeError(lib.ssd_last_error().decode())
        return a, npl.value

    def layers(self, B):
        lib = _h.lib()
        out = []
        for i in range(lib.ssd_net_num_layers(self._net)):
            out.append({"name": lib.ssd_net_layer_name(self._net, i).decode(),
                        "kind": lib.ssd_net_layer_kind(self._net, i).decode(),
                        "config": lib.ssd_net_layer_config(self._net, i).decode(),
                        "flops": lib.ssd_net_layer_flops(self._net, i, B),
                        "executed_flops": lib.ssd_net_layer_executed_flops(self._net, i, B),
                        "bytes": lib.ssd_net_layer_bytes(self._net, i, B)})
        return out

    def predict_on_device(self, images, priors, variances, max_total=200, iou_threshold=0.5,
                          score_threshold=0.5):
        """Forward + SSDDecoder in ONE native call (``ssd_net_predict``): what
        ``get_decoder_model(...).predict`` runs per batch.  Returns device tensors
        (boxes, labels, scores, valid)."""
        x = _h.to_dev(images)
        B = x.shape[0]
        self._ensure(max(B, 1))
        pri = _h.to_dev(priors)
        dev = x.device
        boxes = torch.empty((B, max_total, 4), dtype=torch.float32, device=dev)
        labels = torch.empty((B, max_total), dtype=torch.float32, device=dev)
        scores = torch.empty((B, max_total), dtype=torch.float32, device=dev)
        valid = torch.empty((B,), dtype=torch.int32, device=dev)
        var_p, _keep = _h.host4(variances)
        _h.check(_h.lib().ssd_net_predict(self._net, _h.ptr(x), B, _h.ptr(pri), var_p, int(max_total),
                                          float(iou_threshold), float(score_threshold), _h.ptr(boxes),
                                          _h.ptr(labels), _h.ptr(scores), _h.ptr(valid), _h.stream()),
                 "ssd_net_predict")
        self._last_input = x
        return boxes, labels, scores, valid

    # ------------------------------------------------------------------ training
    def compile(self, optimizer=None, loss=None, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7,
                neg_pos_ratio=None, loc_loss_alpha=None):
        """Keras ``Model.compile(optimizer=Adam(learning_rate=1e-3), loss=[loc, conf])`` (reference
        trainer.py:52-53).  The optimiser is always Adam (the reference's choice) with the Keras
        defaults; ``loss`` may be the two bound methods of a ``CustomLoss`` (its ratio / alpha are
        then used)."""
        self._adam = dict(lr=float(learning_rate), b1=float(beta_1), b2=float(beta_2), eps=float(epsilon))
        owner = getattr(loss[0], "__self__", None) if loss else None
        self._neg_pos_ratio = float(neg_pos_ratio if neg_pos_ratio is not None else
                                    getattr(owner, "neg_pos_ratio", self.hyper_params.get("neg_pos_ratio", 3)))
        self._loc_alpha = float(loc_loss_alpha if loc_loss_alpha is not None else
                                getattr(owner, "loc_loss_alpha", self.hyper_params.get("loc_loss_alpha", 1)))

    def trainable_offsets(self):
        """name -> (offset, shape) inside the flat gradient / parameter vector."""
        lib = _h.lib()
        out = {}
        for name, shape in self.param_specs:
            off = lib.ssd_net_trainable_offset(self._net, name.encode())
            if off >= 0:
                out[name] = (off, shape)
        return out

    def forward_backward(self, images, actual_deltas, actual_labels):
        """Training-mode forward + loss + backward.  Returns (loc_loss [B], conf_loss [B],
        grads_flat) as device tensors; ``grads_flat`` is what data-parallel ranks all-reduce."""
        if not hasattr(self, "_adam"):
            self.compile()
        if not self._weights_set:
            raise RuntimeError("model has no weights: call set_weights()/load_weights() first")
        lib = _h.lib()
        x = _h.to_dev(images)
        yd, yl = _h.to_dev(actual_deltas), _h.to_dev(actual_labels)
        B = x.shape[0]
        if yd.shape != (B, self.num_priors, 4) or yl.shape != (B, self.num_priors, self.total_labels):
            raise ValueError("bad target shapes %s / %s" % (tuple(yd.shape), tuple(yl.shape)))
        if getattr(self, "_train_batch", 0) < B:
            _h.check(lib.ssd_net_train_begin(self._net, B), "ssd_net_train_begin")
            self._train_batch = B
            self._finalized_for = 0
            self._bucket_starts = None         # a new training state has no bucket events: the plan is re-made
        P = lib.ssd_net_trainable_floats(self._net)
        if getattr(self, "_grads", None) is None or self._grads.numel() != P:
            self._grads = torch.empty((P,), dtype=torch.float32, device=x.device)
        loc = torch.empty((B,), dtype=torch.float32, device=x.device)
        conf = torch.empty((B,), dtype=torch.float32, device=x.device)
        _h.check(lib.ssd_net_train_forward_backward(self._net, _h.ptr(x), B, _h.ptr(yd), _h.ptr(yl),
                                                    self._neg_pos_ratio, self._loc_alpha, _h.ptr(self._grads),
                                                    _h.ptr(loc), _h.ptr(conf), _h.stream()), "train_forward_backward")
        self._last_input = x
        return loc, conf, self._grads

    def apply_gradients(self, grads_flat, learning_rate=None, grad_scale=1.0):
        a = self._adam
        lr = a["lr"] if learning_rate is None else float(learning_rate)
        _h.check(_h.lib().ssd_net_adam_step(self._net, _h.ptr(grads_flat), lr, a["b1"], a["b2"], a["eps"],
                                            float(grad_scale), _h.stream()), "ssd_net_adam_step")
        self._finalized_for = 0          # inference weights (folded BN, packed) are stale now
        self._weights_version += 1

    def train_on_batch(self, images, targets, learning_rate=None):
        """Keras ``Model.train_on_batch``: one optimisation step; with torch.distributed
        initialised, gradients are summed over the ranks (RCCL all-reduce over xGMI) and
        averaged (batch data-parallel).  Returns (loss, loc_loss, conf_loss) host
        floats of THIS rank's batch (Keras logs the batch means; ``loss`` includes the regularisation term,
        the two components do not)."""
        self.plan_gradient_exchange(images.shape[0])
        # the regularisation term at the weights this batch is evaluated with, like Keras.  Taken BEFORE the step is
        # issued (the weights do not change until apply_gradients): its device round trip (VGG16 only) then sits
        # between two steps instead of between the backward and the first gradient bucket's all-reduce
        reg = self.regularization_loss()
        loc, conf, g = self.forward_backward(images, targets[0], targets[1])
        world = self.exchange_gradients(g)
        self.apply_gradients(g, learning_rate, 1.0 / world)
        lm, cm = float(loc.mean().item()), float(conf.mean().item())
        return lm + cm + reg, lm, cm

    def plan_gradient_exchange(self, batch):
        """Before ``forward_backward`` of a data-parallel step: plan the gradient buckets (one completion event each
        in the native backward) when more than one rank takes part (or SSD_HIP_FORCE_DIST=1)."""
        import parallel
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or parallel.force_collectives())
        self._n_buckets = int(os.environ.get("SSD_HIP_GRAD_BUCKETS", "4")) if multi else 0
        if self._n_buckets > 1:
            self._plan_gradient_buckets(batch, self._n_buckets)

    def exchange_gradients(self, g):
        """After ``forward_backward``: SUM all-reduce of the flat gradient over the ranks; returns the world size.
        With planned buckets the exchange of a bucket starts when the backward has finished it (heads / extras
        first) and runs on its own stream beside the backward of the backbone."""
        import parallel
        if getattr(self, "_n_buckets", 0) > 1 and getattr(self, "_bucket_starts", None) is not None:
            lib = _h.lib()
            return parallel.allreduce_gradients_as_ready(
                g, self._bucket_starts,
                wait_bucket=lambda k, st: _h.check(lib.ssd_net_train_wait_bucket(self._net, k, _h.vp(st.cuda_stream)), "wait_bucket"),
                comm_stream=self._comm_stream)
        return parallel.allreduce_gradients(g)

    def _plan_gradient_buckets(self, batch, n_buckets):
        """``ssd_net_train_set_buckets``: equal contiguous buckets of the flat gradient, one completion event each."""
        import parallel
        lib = _h.lib()
        if getattr(self, "_train_batch", 0) < batch:
            _h.check(lib.ssd_net_train_begin(self._net, int(batch)), "ssd_net_train_begin")
            self._train_batch = int(batch)
            self._finalized_for = 0
            self._bucket_starts = None
        P = lib.ssd_net_trainable_floats(self._net)
        starts = parallel.bucket_starts(P, n_buckets)          # may hold fewer than n_buckets starts (small P)
        if getattr(self, "_bucket_starts", None) is None or list(self._bucket_starts) != list(starts):
            arr = (ctypes.c_long * len(starts))(*starts)
            _h.check(lib.ssd_net_train_set_buckets(self._net, len(starts), arr), "ssd_net_train_set_buckets")
            self._bucket_starts = starts
            if getattr(self, "_comm_stream", None) is None:
                self._comm_stream = _h.new_stream()
        return self._bucket_starts

    def regularization_loss(self):
        """Keras' ``sum(model.losses)``: l2(5e-4) over VGG16's regularised kernels, 0 for MobileNetV2 -- part
        of the ``loss`` / ``val_loss`` Keras logs and checkpoints on (reference models/ssd_vgg16.py:44-45)."""
        if self.backbone != "vgg16":
            return 0.0
        out = ctypes.c_float(0.0)
        _h.check(_h.lib().ssd_net_regularization_loss(self._net, ctypes.byref(out)), "regularization_loss")
        return float(out.value)

    def train_fetch(self, what, B):
        """Buffer of the last training forward/backward (debug / parity tests)."""
        lib = _h.lib()
        n = lib.ssd_net_train_fetch(self._net, what.encode(), B, None, 0)
        if n < 0:
            raise ValueError(lib.ssd_last_error().decode())
        a = np.empty((n,), np.float32)
        if lib.ssd_net_train_fetch(self._net, what.encode(), B, a.ctypes.data_as(_h.c_float_p), a.size) < 0:
            raise RuntimeError(lib.ssd_last_error().decode())
        return a

    def evaluate_on_batch(self, images, targets):
        """Validation loss of one batch (inference-mode forward + the HIP loss)."""
        from ssd_loss import CustomLoss
        d, p = self(images)
        cl = CustomLoss(getattr(self, "_neg_pos_ratio", 3.0), getattr(self, "_loc_alpha", 1.0))
        lm = float(cl.loc_loss_fn(targets[0], d).mean().item())
        cm = float(cl.conf_loss_fn(targets[1], p).mean().item())
        return lm + cm + self.regularization_loss(), lm, cm

    def set_option(self, name, value):
        value = int(value)
        if name == "precision" and value not in (0, 1):
            raise ValueError("option precision must be 0 (fp32) or 1 (bf16), got %r" % (value,))
        _h.check(_h.lib().ssd_net_set_option(self._net, name.encode(), value), "set_option")
        self._options[name] = value
        if name == "precision":
            # the table key, the tune-cache file and clone() all follow self.precision: a net switched after
            # construction must not memoise its bf16 table under the fp32 key
            self.precision = "bf16" if value else "fp32"
            self._train_batch = 0
        if name in _REFINALIZE_OPTIONS:
            self._finalized_for = 0

    def set_timing(self, enabled):
        _h.check(_h.lib().ssd_net_set_timing(self._net, int(enabled)), "set_timing")

    def read_timing(self, B):
        """Per-layer mean ms over the forwards since the last read (+ 'decode_nms' row)."""
        lib = _h.lib()
        n = lib.ssd_net_num_layers(self._net)
        ms = (ctypes.c_float * (n + 1))()
        cnt = ctypes.c_int(0)
        _h.check(lib.ssd_net_read_timing(self._net, ms, ctypes.byref(cnt)), "read_timing")
        info = self.layers(B)
        info.append({"name": "decode_nms", "kind": "nms", "config": "", "flops": 0.0, "executed_flops": 0.0, "bytes": 0.0})
        k = max(cnt.value, 1)
        for i, rec in enumerate(info):
            rec["ms"] = float(ms[i]) / k
        return info, cnt.value

    def profile_layers(self, images, reps=5):
        x = _h.to_dev(images)
        B = x.shape[0]
        self._ensure(B)
        n = _h.lib().ssd_net_num_layers(self._net)
        ms = (ctypes.c_float * n)()
        _h.check(_h.lib().ssd_net_profile_layers(self._net, _h.ptr(x), B, reps, ms, _h.stream()), "profile_layers")
        info = self.layers(B)
        for i, rec in enumerate(info):
            rec["ms"] = float(ms[i])
        return info


def keras_default_init(param_specs, backbone, seed=0):
    """Host-side initial weights: what ``get_model`` leaves in a freshly built Keras model.
    Conv kernels: glorot_uniform (Keras default; extras + heads, models/header.py:60-61) or
    glorot_normal (VGG16, models/ssd_vgg16.py:44); biases 0; BatchNorm gamma 1, beta 0,
    mean 0, variance 1; L2Normalization scale 20.  The MobileNetV2 backbone would load
    ImageNet weights (models/ssd_mobilenet_v2.py:16) -- no network here, so it gets He-normal
    kernels instead."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in param_specs:
        var = name.rsplit("/", 1)[1]
        if var in ("kernel", "depthwise_kernel"):
            kh, kw, cin, cout = shape
            fan_in, fan_out = kh * kw * cin, kh * kw * cout
            if var == "depthwise_kernel":
                fan_in, fan_out = kh * kw, kh * kw
            is_backbone = backbone == "mobilenet_v2" and not (name.startswith("extra") or name[0].isdigit())
            if is_backbone:
                a = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
            elif backbone == "vgg16" and not name[0].isdigit():
                a = rng.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out))
            else:
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                a = rng.uniform(-lim, lim, shape)
            w[name] = a.astype(np.float32)
        elif var in ("gamma", "moving_variance"):
            w[name] = np.ones(shape, np.float32)
        elif var == "scale":
            w[name] = np.full(shape, 20.0, np.float32)
        else:
            w[name] = np.zeros(shape, np.float32)
    return w

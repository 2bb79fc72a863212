"""Drop-in for the reference's ``models/decoder.py``: ``SSDDecoder`` and
``get_decoder_model``.  The whole of ``SSDDecoder.call`` -- variance scaling, box decode,
argmax class mask, per-class greedy NMS, top-K merge, clipping, zero padding -- runs as
three HIP kernels behind ``ssd_decode_nms`` (include/ssd_hip.h)."""
import numpy as np
import torch

import ssd_hip as _h


class SSDDecoder(object):
    """reference models/decoder.py:6-55.

    inputs:  [pred_deltas (B,N,4), pred_label_probs (B,N,L)]
    outputs: pred_bboxes (B,top_n,4), pred_labels (B,top_n), pred_scores (B,top_n)
    """

    def __init__(self, prior_boxes, variances, max_total_size=200, score_threshold=0.5, **kwargs):
        self.name = kwargs.pop("name", "ssd_decoder")
        self.prior_boxes = prior_boxes
        self.variances = variances
        self.max_total_size = max_total_size
        self.score_threshold = score_threshold
        self.iou_threshold = 0.5           # TF default of combined_non_max_suppression
        self._priors_dev = None
        self.last_valid_detections = None  # TF's 4th output, discarded by the reference (:49)
        self.last_kept_indices = None

    def get_config(self):
        """reference models/decoder.py:26-34."""
        pb = self.prior_boxes
        pb = pb.detach().cpu().numpy() if isinstance(pb, torch.Tensor) else np.asarray(pb)
        return {
            "name": self.name,
            "prior_boxes": pb,
            "variances": self.variances,
            "max_total_size": self.max_total_size,
            "score_threshold": self.score_threshold,
        }

    def call(self, inputs, return_indices=False):
        """reference models/decoder.py:36-55."""
        pred_deltas = _h.to_dev(inputs[0])
        pred_label_probs = _h.to_dev(inputs[1])
        if self._priors_dev is None:
            self._priors_dev = _h.to_dev(self.prior_boxes)
        pri = self._priors_dev
        if pred_deltas.dim() != 3 or pred_label_probs.dim() != 3 or pred_deltas.shape[2] != 4 \
                or pred_deltas.shape[:2] != pred_label_probs.shape[:2] or pri.shape[0] != pred_deltas.shape[1]:
            raise ValueError("bad shapes %s / %s / %s" % (tuple(pred_deltas.shape),
                                                          tuple(pred_label_probs.shape), tuple(pri.shape)))
        B, N, L = pred_label_probs.shape
        T = int(self.max_total_size)
        dev = pred_deltas.device
        boxes = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        labels = torch.empty((B, T), dtype=torch.float32, device=dev)
        scores = torch.empty((B, T), dtype=torch.float32, device=dev)
        valid = torch.empty((B,), dtype=torch.int32, device=dev)
        kept = torch.empty((B, T), dtype=torch.int32, device=dev) if return_indices else None
        var_p, _keep = _h.host4(self.variances)
        lib = _h.lib()
        ws = _h.workspace(lib.ssd_decode_nms_workspace_bytes(B, N, L, T))
        _h.check(lib.ssd_decode_nms(_h.ptr(pred_deltas), _h.ptr(pred_label_probs), _h.ptr(pri), var_p,
                                    B, N, L, T, T, float(self.iou_threshold), float(self.score_threshold),
                                    _h.ptr(boxes), _h.ptr(labels), _h.ptr(scores), _h.ptr(valid),
                                    _h.ptr(kept), _h.ptr(ws), ws.numel(), _h.stream()), "SSDDecoder.call")
        self.last_valid_detections = valid
        self.last_kept_indices = kept
        return boxes, labels, scores

    __call__ = call


class DecoderModel(object):
    """What ``get_decoder_model`` returns: ``Model(inputs=base.input, outputs=[bboxes,
    classes, scores])`` (reference models/decoder.py:68-69), used through ``predict``."""

    def __init__(self, base_model, decoder):
        self.base_model = base_model
        self.decoder = decoder

    def __call__(self, images):
        d = self.decoder
        if hasattr(self.base_model, "predict_on_device"):
            b, l, s, v = self.base_model.predict_on_device(
                images, d.prior_boxes, d.variances, max_total=d.max_total_size,
                iou_threshold=d.iou_threshold, score_threshold=d.score_threshold)
            d.last_valid_detections = v
            return b, l, s
        deltas, probs = self.base_model(images)
        return d([deltas, probs])

    def predict_on_batch(self, images):
        return tuple(t.cpu().numpy() for t in self(images))

    def predict(self, x, steps=None, verbose=0, batch_size=32):
        """Keras ``Model.predict``: x is an array [n,S,S,3] (split into ``batch_size``
        chunks) or an iterable of batches (an image array, or a tuple whose first element
        is the image batch, like the reference's padded-batch dataset).  Returns three
        NumPy arrays concatenated over the batches (reference predictor.py:52)."""
        if isinstance(x, (np.ndarray, torch.Tensor)):
            n = x.shape[0]
            batches = (x[i:i + batch_size] for i in range(0, n, batch_size))
        else:
            batches = iter(x)
        outs = ([], [], [])
        done = 0
        for batch in batches:
            if steps is not None and done >= steps:
                break
            imgs = batch[0] if isinstance(batch, (tuple, list)) else batch
            res = self.predict_on_batch(imgs)
            for acc, r in zip(outs, res):
                acc.append(r)
            done += 1
            if verbose:
                print("\r%d/%s" % (done, steps if steps is not None else "?"), end="", flush=True)
        if verbose:
            print()
        if done == 0:
            T = int(self.decoder.max_total_size)
            return (np.zeros((0, T, 4), np.float32), np.zeros((0, T), np.float32), np.zeros((0, T), np.float32))
        return tuple(np.concatenate(a, 0) for a in outs)


def get_decoder_model(base_model, prior_boxes, hyper_params):
    """reference models/decoder.py:57-69."""
    decoder = SSDDecoder(prior_boxes, hyper_params["variances"])
    return DecoderModel(base_model, decoder)

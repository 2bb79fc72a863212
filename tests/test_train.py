"""N1 backward + Adam + data-parallel step: the native training step
(csrc/ssd_train.hip through the C ABI) against torch-CPU autograd of the same graph
(oracle/train_oracle.py).  Bars: network outputs 1e-4 abs; per-image losses 1e-4 rel; every
parameter gradient within 1e-3 of the tensor's max |gradient| (fp32, ~50 layers, batch statistics);
BatchNorm moving averages 1e-5; Adam update vs the NumPy ApplyAdam on the same gradients 1e-6."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gradient_buckets_and_adam_oracle():
    sys.path.insert(0, os.path.join(REPO, "tf-ssd_amd"))
    import parallel
    from oracle import train_oracle as to
    assert parallel.gradient_buckets(10, None) == [(0, 10)]
    assert parallel.gradient_buckets(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.gradient_buckets(0, 4) == []
    import torch
    assert parallel.allreduce_gradients(torch.ones(5)) == 1          # no process group: no-op
    # Adam first step moves every weight by ~lr against the gradient sign
    g = np.array([0.5, -2.0, 1e-3], np.float32)
    var, m, v = to.adam_step(np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32), g, 1)
    np.testing.assert_allclose(var, -1e-3 * np.sign(g), rtol=5e-3)
    np.testing.assert_allclose(m, 0.1 * g, rtol=1e-6)
    np.testing.assert_allclose(v, 0.001 * g * g, rtol=2e-5)


_DP_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "tf-ssd_amd")]
import parallel
rank, _, world = parallel.init_distributed("gloo")
rng = np.random.default_rng(100 + rank)
g = torch.from_numpy(rng.standard_normal(1000).astype(np.float32))
ref = sum(np.random.default_rng(100 + r).standard_normal(1000).astype(np.float32) for r in range(world))
w = parallel.allreduce_gradients(g, bucket_floats=int(sys.argv[2]) or None)
assert w == world
np.testing.assert_allclose(g.numpy(), ref, rtol=1e-6, atol=1e-6)
# averaged update is identical on every rank (replicas stay in sync)
upd = g / w
gathered = [torch.empty_like(upd) for _ in range(world)]
dist.all_gather(gathered, upd)
for t in gathered:
    assert torch.equal(t, gathered[0])
dist.destroy_process_group()
print("DP_OK", rank)
'''


@pytest.mark.parametrize("bucket", [0, 300])
def test_two_process_gloo_gradient_allreduce(tmp_path, bucket):
    """World-size-2 gloo run of the gradient exchange (single flat all-reduce and bucketed)."""
    script = tmp_path / "w.py"
    script.write_text(_DP_WORKER)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, str(bucket)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "DP_OK %d" % r in o, o[-2000:]


def _targets(hp, B, seed=3):
    from oracle import bbox_oracle as bo
    priors = bo.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    gt, gl = helpers.gt_inputs(B, G=8, L=hp["total_labels"], seed=seed)
    yd, yl = bo.calculate_actual_outputs(priors, gt, gl, hp)
    return yd, yl


@pytest.mark.gpu
def test_train_step_matches_autograd_oracle():
    import torch
    from models.ssd_mobilenet_v2 import get_model
    from oracle import train_oracle as to
    from ssd_loss import CustomLoss
    hp = helpers.hyper_params("mobilenet_v2")
    w = {k: v.copy() for k, v in helpers.synthetic_weights("mobilenet_v2", hp).items()}
    B = 4
    x = helpers.images(B, 300, seed=21)
    yd, yl = _targets(hp, B)
    assert (yl[..., 1:].sum(-1) > 0).sum(1).min() > 0
    m = get_model(hp)
    m.set_weights(w)
    cl = CustomLoss(hp["neg_pos_ratio"], hp["loc_loss_alpha"])
    m.compile(loss=[cl.loc_loss_fn, cl.conf_loss_fn])
    loc, conf, g = m.forward_backward(x, yd, yl)
    loc, conf, g = loc.cpu().numpy(), conf.cpu().numpy(), g.cpu().numpy().copy()
    probs = m.train_fetch("probs", B).reshape(B, -1, hp["total_labels"])
    deltas = m.train_fetch("deltas", B).reshape(B, -1, 4)
    # the hard-negative selection the device made on ITS probabilities (bit-exactness of the
    # selection rule itself is tests/test_loss.py); the oracle differentiates with the same mask
    cl.conf_loss_fn(yl, probs)
    fm = cl.last_final_mask.cpu().numpy()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = to.train_step("mobilenet_v2", hp, w, x, yd, yl, 3.0, 1.0, final_mask=fm)
    assert np.abs(probs - ref["probs"]).max() <= 1e-4
    assert np.abs(deltas - ref["deltas"]).max() <= 1e-4 * max(1.0, float(np.abs(ref["deltas"]).max()))
    np.testing.assert_allclose(loc, ref["loc"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(conf, ref["conf"], rtol=1e-4, atol=1e-6)
    # selection agrees with the oracle's own (near-ties aside)
    own = to.lo.conf_loss_fn(yl, ref["probs"], 3.0, return_parts=True)[2]
    assert (own != fm).mean() < 1e-3
    offs = m.trainable_offsets()
    assert set(offs) == set(ref["grads"])
    worst = (0.0, None)
    for name, (off, shape) in offs.items():
        got = g[off:off + int(np.prod(shape))].reshape(shape)
        rg = ref["grads"][name]
        scale = max(float(np.abs(rg).max()), 1e-6)
        err = float(np.abs(got - rg).max()) / scale
        if err > worst[0]:
            worst = (err, name)
        assert err <= 1e-3, "%s: gradient off by %.3e of its max (%.3e)" % (name, err, scale)
    print("worst relative gradient error %.2e (%s)" % worst)
    # BatchNorm moving averages after one training forward
    after = m.get_weights()
    for name, val in ref["moving"].items():
        np.testing.assert_allclose(after[name], val, rtol=1e-5, atol=1e-6, err_msg=name)
    # Adam: two steps vs the NumPy ApplyAdam on the device's own gradients
    before = {k: v.copy() for k, v in after.items()}
    m.apply_gradients(m._grads, learning_rate=1e-3)
    step1 = m.get_weights()
    st = {}
    for name, (off, shape) in offs.items():
        gg = g[off:off + int(np.prod(shape))].reshape(shape)
        var, mm, vv = to.adam_step(before[name], np.zeros(shape, np.float32), np.zeros(shape, np.float32), gg, 1)
        st[name] = (mm, vv)
        np.testing.assert_allclose(step1[name], var, rtol=1e-6, atol=1e-7, err_msg=name)
    loc2, conf2, g2 = m.forward_backward(x, yd, yl)
    g2 = g2.cpu().numpy().copy()
    m.apply_gradients(m._grads, learning_rate=1e-3, grad_scale=0.5)
    step2 = m.get_weights()
    for name in ("Conv1/kernel", "block_5_depthwise/depthwise_kernel", "bn_Conv1/gamma", "extra2_2/bias",
                 "1_conv_label_output/kernel", "6_conv_boxes_output/bias"):
        off, shape = offs[name]
        gg = g2[off:off + int(np.prod(shape))].reshape(shape) * np.float32(0.5)
        var, _, _ = to.adam_step(step1[name], st[name][0], st[name][1], gg, 2)
        np.testing.assert_allclose(step2[name], var, rtol=1e-6, atol=1e-7, err_msg=name)
    # the step reduces the loss on the same batch (sanity of sign / scale)
    assert float((loc2 + conf2).mean()) < float((loc + conf).mean())
    # inference after training re-finalises from the updated parameters
    d, p = m(x[:1])
    assert np.isfinite(p.cpu().numpy()).all()


@pytest.mark.gpu
def test_trainer_entry_point_fit(tmp_path, monkeypatch, capsys):
    """trainer.py: fit loop with LR schedule, validation, best-val_loss checkpoint (reference
    trainer.py:57-76) on a few synthetic steps; the checkpoint is a Keras-layout HDF5 file the
    predictor-side loader reads back."""
    import importlib
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("SSD_TRAINER_EPOCHS", "2")
    monkeypatch.setenv("SSD_TRAINER_STEPS", "3")
    monkeypatch.setenv("SSD_TRAINER_BATCH", "4")
    trainer = importlib.import_module("trainer")
    hist = trainer.main(["--backbone", "mobilenet_v2"])
    out = capsys.readouterr().out
    assert "Epoch 1/2" in out and "Epoch 2/2" in out and "val_loss" in out
    assert len(hist["loss"]) == 2 and np.isfinite(hist["loss"]).all() and np.isfinite(hist["val_loss"]).all()
    assert hist["loss"][1] < hist["loss"][0]
    path = os.path.join("trained", "ssd_mobilenet_v2_model_weights.h5")
    assert os.path.exists(path)
    from utils import h5_reader
    w = h5_reader.load_keras_weights(path)
    assert "bn_Conv1/moving_mean" in w and w["Conv1/kernel"].shape == (3, 3, 3, 32)

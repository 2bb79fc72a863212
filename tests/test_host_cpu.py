"""CPU tests of the host side: the C-ABI library loads and exports every symbol
include/ssd_hip.h declares (no compute calls without a GPU), host logic of the drop-in
modules, and the N>1 sharding/gather plumbing with world-size-2 gloo processes."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "ssd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ssd_hip
    assert os.path.exists(ssd_hip.LIB_PATH), "build libssd_hip.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(ssd_hip.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    # and the ctypes table covers all of them
    assert set(names) <= set(ssd_hip._SIGNATURES), set(names) - set(ssd_hip._SIGNATURES)
    l = ssd_hip.lib()
    assert b"gfx950" in l.ssd_version()


def test_pure_host_entry_points():
    import ssd_hip
    l = ssd_hip.lib()
    b, a = ctypes.c_int(), ctypes.c_int()
    assert l.ssd_same_pads(10, 3, 2, 1, ctypes.byref(b), ctypes.byref(a)) == 5 and (b.value, a.value) == (0, 1)
    assert l.ssd_same_pads(19, 3, 1, 6, ctypes.byref(b), ctypes.byref(a)) == 19 and (b.value, a.value) == (6, 6)
    assert l.ssd_conv_out_size(5, 3, 1, 1, 0, 0) == 3 and l.ssd_conv_out_size(5, 3, 0, 1, 0, 0) == 0
    assert l.ssd_conv_packed_weight_floats(3, 3, 576, 84) == 5184 * 96 * 3      # fp32 [Npad][Kpad] + its four bf16 planes (h, m, l, r)
    fm = (ctypes.c_int * 6)(19, 10, 5, 3, 2, 1)
    na = (ctypes.c_int * 6)(3, 5, 5, 5, 3, 3)
    assert l.ssd_priors_count(fm, na, 6) == 2268
    assert l.ssd_conv_num_configs() >= 10 and l.ssd_conv_config_name(0).startswith(b"mfma_")
    # graph construction needs no device
    net = l.ssd_net_create(ssd_hip.MOBILENET_V2, 300, 6, na, 21)
    assert net and l.ssd_net_num_priors(net) == 2268 and l.ssd_net_num_params(net) == 300
    assert l.ssd_net_forward(net, None, 1, None, None, None) == -4          # not finalized
    assert b"finalize" in l.ssd_last_error()
    assert l.ssd_net_set_param(net, b"nope", (ctypes.c_float * 1)(), 1) == -1
    l.ssd_net_destroy(net)
    assert not l.ssd_net_create(7, 300, 6, na, 21) and b"bad arguments" in l.ssd_last_error()


def test_no_cpu_fallback():
    """Without a GPU every compute entry point of the Python surface must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ssd_hip
    from utils import bbox_utils
    with pytest.raises(ssd_hip.SsdHipError):
        bbox_utils.generate_prior_boxes([19, 10, 5, 3, 2, 1], [[1., 2., .5]] * 6)
    with pytest.raises(ssd_hip.SsdHipError):
        bbox_utils.get_bboxes_from_deltas(np.zeros((4, 4), np.float32), np.zeros((1, 4, 4), np.float32))


def test_hyper_params_semantics():
    from utils import train_utils, io_utils
    hp = train_utils.get_hyper_params("mobilenet_v2")
    assert hp is train_utils.SSD["mobilenet_v2"]                     # mutated global, like the reference
    assert hp["feature_map_shapes"] == [19, 10, 5, 3, 2, 1] and hp["variances"] == [0.1, 0.1, 0.2, 0.2]
    assert train_utils.get_hyper_params("vgg16", img_size=512, nope=3, iou_threshold=0)["img_size"] == 512
    assert train_utils.SSD["vgg16"]["iou_threshold"] == 0.5 and "nope" not in train_utils.SSD["vgg16"]
    train_utils.SSD["vgg16"]["img_size"] = 300
    assert [train_utils.scheduler(e) for e in (0, 99, 100, 124, 125)] == [1e-3, 1e-3, 1e-4, 1e-4, 1e-5]
    assert train_utils.get_step_size(4952, 32) == 155
    args = io_utils.handle_args([])
    assert args.backbone == "mobilenet_v2" and args.handle_gpu is False
    assert io_utils.handle_args(["-handle-gpu", "--backbone", "vgg16"]).backbone == "vgg16"
    with pytest.raises(AssertionError):
        io_utils.is_valid_backbone("resnet")
    assert io_utils.get_log_path("vgg16").startswith("logs/vgg16/")


def test_host_logic_vs_executed_reference(tmp_path, monkeypatch):
    """H1 / E1 host logic against outputs of the REFERENCE's own functions (tests/golden/host_logic.json,
    written by make_host_golden.py, which executes the reference's TF-free definitions via ast)."""
    import copy, json
    from utils import train_utils, io_utils
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic.json")))
    pristine = copy.deepcopy(train_utils.SSD)
    try:
        for case in g["get_hyper_params"]:
            train_utils.SSD.clear()
            train_utils.SSD.update(copy.deepcopy(pristine))
            out = train_utils.get_hyper_params(case["backbone"], **case["kwargs"])
            assert out == case["out"], case
            assert list(out) == list(case["out"]) or sorted(out) == sorted(case["out"])
        train_utils.SSD.clear()
        train_utils.SSD.update(copy.deepcopy(pristine))
        train_utils.get_hyper_params("mobilenet_v2", img_size=512)
        assert train_utils.get_hyper_params("mobilenet_v2") == g["sticky_img_size_after_override"]
    finally:
        train_utils.SSD.clear()
        train_utils.SSD.update(pristine)
    for e, lr in g["scheduler"]:
        assert train_utils.scheduler(e) == lr, e
    for t, b, n in g["get_step_size"]:
        assert train_utils.get_step_size(t, b) == n
    monkeypatch.chdir(tmp_path)
    for m, path in g["get_model_path"].items():
        assert io_utils.get_model_path(m) == path
    assert os.path.isdir(tmp_path / "trained") == g["get_model_path_creates_dir"]
    for b, ok in g["is_valid_backbone"].items():
        if ok:
            io_utils.is_valid_backbone(b)
        else:
            with pytest.raises(AssertionError):
                io_utils.is_valid_backbone(b)


def test_tf_free_helpers_vs_executed_reference(tmp_path, monkeypatch):
    """The remaining TF-free reference functions on / beside the hot path (A1 ``get_scale_for_nth_feature_map``,
    ``get_log_path``, ``handle_args``, ``get_total_item_size``, ``get_labels``, ``get_custom_imgs``) against outputs
    of the reference's OWN definitions executed by tests/golden/make_host_golden.py."""
    import json, time, types
    from utils import bbox_utils, io_utils, data_utils
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_logic.json")))
    for k, kw, want in g["get_scale_for_nth_feature_map"]:
        assert bbox_utils.get_scale_for_nth_feature_map(k, **kw) == want, (k, kw)       # the same float64 expression: exact
    assert g["get_scale_m1"] == "ZeroDivisionError"
    with pytest.raises(ZeroDivisionError):
        bbox_utils.get_scale_for_nth_feature_map(1, m=1)
    frozen = time.struct_time((2020, 1, 2, 3, 4, 5, 3, 2, 0))
    monkeypatch.setattr(io_utils.time, "localtime", lambda *a: frozen)
    for m, pf, want in g["get_log_path_at_2020_01_02_03_04_05"]:
        assert io_utils.get_log_path(m, pf) == want
    for argv, want in g["handle_args"]:
        assert vars(io_utils.handle_args(argv)) == want, argv
    for argv, want in g["handle_args_bad"]:
        assert want == "SystemExit 2"
        with pytest.raises(SystemExit) as e:
            io_utils.handle_args(argv)
        assert e.value.code == 2
    names = g["get_labels"]
    info = types.SimpleNamespace(
        splits={"train": types.SimpleNamespace(num_examples=2501), "validation": types.SimpleNamespace(num_examples=2510),
                "test": types.SimpleNamespace(num_examples=4952)},
        features={"labels": types.SimpleNamespace(names=names)})
    for sp, want in g["get_total_item_size"]:
        assert data_utils.get_total_item_size(info, sp) == want
        assert data_utils.get_total_item_size({"splits": {"train": 2501, "validation": 2510, "test": 4952}}, sp) == want
    with pytest.raises(AssertionError):
        data_utils.get_total_item_size(info, "all")
    assert data_utils.get_labels(info) == names
    for f in ("b.jpg", "a.png", "c.txt"):
        (tmp_path / f).write_bytes(b"")
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "nested.jpg").write_bytes(b"")
    got = [os.path.relpath(q, tmp_path) for q in data_utils.get_custom_imgs(str(tmp_path))]
    # the reference lists in os.walk order (unspecified); this build sorts: the same SET, one fixed order
    assert sorted(got) == g["get_custom_imgs_sorted_relative"] and got == sorted(got)
    assert data_utils.get_custom_imgs(str(tmp_path / "does_not_exist")) == g["get_custom_imgs_missing_dir"]


def test_shard_range():
    import parallel
    for n in (0, 1, 7, 64, 4952):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(repo)r, "tf-ssd_amd"))
import torch, parallel
rank, local, world = parallel.init_distributed("gloo")
assert world == 2
n, T = 5, 3
lo, hi = parallel.shard_range(n, rank, world)
idx = torch.arange(lo, hi, dtype=torch.float32)
boxes = idx.view(-1, 1, 1).expand(-1, T, 4).contiguous()
labels = (idx.view(-1, 1) + 100).expand(-1, T).contiguous()
scores = (idx.view(-1, 1) + 200).expand(-1, T).contiguous()
b, l, s = parallel.gather_detections(boxes, labels, scores, n_total=n)
assert b.shape == (n, T, 4) and torch.equal(b[:, 0, 0], torch.arange(n, dtype=torch.float32))
assert torch.equal(l[:, 0], torch.arange(n, dtype=torch.float32) + 100)
assert torch.equal(s[:, 2], torch.arange(n, dtype=torch.float32) + 200)
assert parallel.max_over_ranks(1.0 + rank) == 2.0
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.stdout.write("rank " + str(rank) + " ok\n"); sys.stdout.flush()
'''


def test_two_process_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"repo": REPO})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:                 # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


# ---------------------------------------------------------------------------------------------
# Graph builder of the native library (host-only code: no GPU needed) against the oracle's
# parameter table and against PUBLISHED known answers for the two backbones -- the only
# external anchors available offline (the reference has no tests / fixtures):
#   * keras.applications.MobileNetV2(include_top=False, alpha=1.0).summary():
#       Total params 2,257,984 / trainable 2,223,872 / non-trainable 34,112
#   * keras.applications.VGG16(include_top=False).summary(): Total params 14,714,688
#   * SSD300 (Liu et al. 2016) VGG16 head: 8732 default boxes; the reference's MobileNetV2
#     feature maps [19,10,5,3,2,1] with the same aspect ratios give 2268
#     (utils/train_utils.py:12-33, utils/bbox_utils.py:139-158).
@pytest.mark.parametrize("backbone,n_priors", [("mobilenet_v2", 2268), ("vgg16", 8732)])
def test_native_graph_matches_oracle_and_published_counts(backbone, n_priors):
    import numpy as np
    import helpers
    from oracle import net_oracle as no
    from models._net import SSDModel
    hp = helpers.hyper_params(backbone)
    m = SSDModel(backbone, hp)                      # ssd_net_create only: host code
    assert m.param_specs == no.param_specs(backbone, hp)
    assert m.num_priors == n_priors
    names = [n for n, _ in m.param_specs]
    count = lambda specs: sum(int(np.prod(s)) for _, s in specs)
    if backbone == "mobilenet_v2":
        back = m.param_specs[:names.index("Conv_1_bn/moving_variance") + 1]
        assert count(back) == 2257984
        assert count([(n, s) for n, s in back if "moving_" in n]) == 34112
        assert count(back) - 34112 == 2223872
    else:
        back = m.param_specs[:names.index("conv5_3/bias") + 1]
        assert count(back) == 14714688
    # a hyper-parameter set that does not describe this graph is refused
    bad = dict(hp)
    bad["feature_map_shapes"] = list(hp["feature_map_shapes"][:-1]) + [7]
    with pytest.raises(ValueError):
        SSDModel(backbone, bad)
    # checkpoint layer order = Keras' model.layers (depth-sorted): every weighted layer once, the
    # chain first, ALL label heads before ALL box heads, l2_normalization tied with (and before) conv11_2
    order = m.layer_order()
    table = []
    for n in names:
        if n.rsplit("/", 1)[0] not in table:
            table.append(n.rsplit("/", 1)[0])
    assert sorted(order) == sorted(table) and len(set(order)) == len(order)
    assert order[-12:] == ["%d_conv_label_output" % i for i in range(1, 7)] + ["%d_conv_boxes_output" % i for i in range(1, 7)]
    chain = [l for l in table if not l[0].isdigit() and l != "l2_normalization"]
    assert [l for l in order[:-12] if l != "l2_normalization"] == chain
    if backbone == "vgg16":
        assert order.index("l2_normalization") == order.index("conv11_2") - 1 == order.index("conv11_1") + 1


def test_u1_box_helpers_product_vs_oracle():
    """normalize / denormalize / renormalize of the PRODUCT's utils.bbox_utils
    (reference utils/bbox_utils.py:178-222) against the oracle and hand-computed answers,
    incl. round-half-to-even and clipping."""
    from utils import bbox_utils as bu
    from oracle import bbox_oracle as bo
    rng = np.random.default_rng(7)
    b = np.array([[[10.5, 20.5, 30.5, 41.5]]], np.float32)
    n = bu.normalize_bboxes(b, 100, 200).numpy()
    np.testing.assert_allclose(n, [[[0.105, 0.1025, 0.305, 0.2075]]], rtol=1e-6)
    d = bu.denormalize_bboxes(np.array([[[0.105, 0.1025, 0.305, 0.2075]]], np.float32), 100, 200).numpy()
    np.testing.assert_array_equal(d, [[[10, 20, 30, 42]]])        # 10.5 -> 10, 20.5 -> 20: half to even
    r = bu.renormalize_bboxes_with_min_max(np.array([[0.2, 0.2, 0.6, 1.2]], np.float32),
                                           np.array([0.1, 0.1, 0.9, 0.9], np.float32)).numpy()
    np.testing.assert_allclose(r, [[0.125, 0.125, 0.625, 1.0]], rtol=1e-6)
    for shape in ((7, 4), (3, 5, 4), (2, 1, 9, 4)):
        x = rng.uniform(-0.2, 1.2, shape).astype(np.float32)
        for h, w in ((300, 300), (375, 500), (1, 7)):
            np.testing.assert_array_equal(bu.denormalize_bboxes(x, h, w).numpy(), bo.denormalize_bboxes(x, h, w))
            px = (x * np.float32(max(h, w))).astype(np.float32)
            np.testing.assert_array_equal(bu.normalize_bboxes(px, h, w).numpy(), bo.normalize_bboxes(px, h, w))
        mm = np.array([0.1, 0.2, 0.7, 0.9], np.float32)
        np.testing.assert_array_equal(bu.renormalize_bboxes_with_min_max(x, mm).numpy(),
                                      bo.renormalize_bboxes_with_min_max(x, mm))
    # exact .5 products round to even in both directions
    half = np.array([[0.5, 1.5, 2.5, 3.5]], np.float32)
    np.testing.assert_array_equal(bu.denormalize_bboxes(half, 1, 1).numpy(), [[0, 2, 2, 4]])


def test_new_entry_points_validate_without_a_device():
    """Argument / state checks of the round-2 entry points run before any device access: they can
    be exercised on a CPU-only host (status codes + ssd_last_error, nothing thrown across the ABI)."""
    import ssd_hip
    l = ssd_hip.lib()
    vp = ctypes.c_void_p
    one = vp(16)            # a non-NULL, 16-byte aligned dummy (never dereferenced: the checks fail first)
    # ssd_loss: sizes, missing pairs, workspace
    assert l.ssd_loss(one, one, one, one, 1, 0, 21, 3.0, 1.0, None, None, None, None, None, None, 1.0, one, 1 << 20, None) == -1
    assert b"bad sizes" in l.ssd_last_error()
    assert l.ssd_loss(None, None, None, None, 1, 8, 21, 3.0, 1.0, None, None, None, None, None, None, 1.0, one, 1 << 20, None) == -1
    assert l.ssd_loss(one, None, None, None, 1, 8, 21, 3.0, 1.0, None, None, None, None, None, None, 1.0, one, 1 << 20, None) == -1
    assert l.ssd_loss(one, one, None, None, 1, 8, 21, 3.0, 1.0, None, None, None, None, one, one, 1.0, one, 1 << 20, None) == -1
    assert b"grad_logits" in l.ssd_last_error()
    assert l.ssd_loss(one, one, one, one, 4, 2268, 21, 3.0, 1.0, None, None, None, None, None, None, 1.0, one, 16, None) == -1
    assert b"workspace" in l.ssd_last_error()
    assert l.ssd_loss_workspace_bytes(32, 2268) >= 32 * 2268 * 9 and l.ssd_loss_workspace_bytes(-1, 5) == 0
    assert l.ssd_loss(one, one, one, one, 0, 8, 21, 3.0, 1.0, None, None, None, None, None, None, 1.0, None, 0, None) == 0   # empty batch
    # ssd_preprocess
    assert l.ssd_preprocess(one, 1, 0, 5, 3, 300, 300, one, None) == -1
    assert l.ssd_preprocess(None, 1, 5, 5, 3, 300, 300, one, None) == -1
    assert l.ssd_preprocess(None, 0, 5, 5, 3, 300, 300, None, None) == 0
    # Winograd op: weight size query, not-applicable geometry
    assert l.ssd_conv_wino_weight_floats(576, 100) == 16 * 112 * 576 and l.ssd_conv_wino_num_configs() >= 4
    d = ssd_hip.ConvDesc(1, 8, 8, 16, 16, 3, 3, 2, 1, 1, 1, 1, 1, 0, 0)
    assert l.ssd_conv2d_wino(ctypes.byref(d), one, one, None, None, one, 0, 0, 0, 1, None, None) == -3
    d = ssd_hip.ConvDesc(1, 8, 8, 16, 16, 3, 3, 1, 1, 1, 1, 1, 1, 0, 1)
    assert l.ssd_conv2d_wino(ctypes.byref(d), one, one, None, None, one, 0, 0, 0, 1, None, None) == -1       # residual
    # training entry points: state machine
    na = (ctypes.c_int * 6)(3, 5, 5, 5, 3, 3)
    net = l.ssd_net_create(ssd_hip.MOBILENET_V2, 300, 6, na, 21)
    assert l.ssd_net_trainable_floats(net) == 8493678                  # 8.53 M parameters minus the moving statistics
    assert l.ssd_net_trainable_offset(net, b"Conv1/kernel") == 0
    assert l.ssd_net_trainable_offset(net, b"bn_Conv1/moving_mean") == -1
    assert l.ssd_net_trainable_offset(net, b"bn_Conv1/gamma") == 864
    assert l.ssd_net_train_forward_backward(net, one, 1, one, one, 3.0, 1.0, one, None, None, None) == -4
    assert b"train_begin" in l.ssd_last_error()
    assert l.ssd_net_adam_step(net, one, 1e-3, 0.9, 0.999, 1e-7, 1.0, None) == -4
    assert l.ssd_net_train_begin(net, 0) == -1
    assert l.ssd_net_train_begin(net, 4) == -4 and b"never set" in l.ssd_last_error()
    assert l.ssd_net_train_steps(net) == 0 and l.ssd_net_train_fetch(net, b"probs", 1, None, 0) < 0
    l.ssd_net_destroy(net)


def test_data_utils_padded_batch_and_custom_image_listing(tmp_path):
    """Host logic of the reference's dataset plumbing (utils/data_utils.py:47-59, 80-91, 117-122; predictor.py:43):
    padded batches (ground truth padded with 0 / -1 to the longest of the batch, ragged last batch), the listing
    of custom images (files directly inside the directory, no recursion), the item-size helper."""
    import numpy as np
    from utils import data_utils
    items = []
    rng = np.random.default_rng(0)
    for g in (3, 0, 5, 1, 2):
        items.append((rng.random((8, 8, 3), dtype=np.float32), rng.random((g, 4), dtype=np.float32),
                      rng.integers(1, 21, g).astype(np.int32)))
    batches = list(data_utils.padded_batch(iter(items), 2))
    assert [b[0].shape[0] for b in batches] == [2, 2, 1]
    assert batches[0][1].shape == (2, 3, 4) and batches[1][1].shape == (2, 5, 4) and batches[2][1].shape == (1, 2, 4)
    np.testing.assert_array_equal(batches[0][2][1], [-1, -1, -1])          # the empty item: all padding
    np.testing.assert_array_equal(batches[0][1][1], np.zeros((3, 4), np.float32))
    np.testing.assert_array_equal(batches[1][2][1], [items[3][2][0], -1, -1, -1, -1])
    np.testing.assert_array_equal(batches[1][1][0], items[2][1])
    np.testing.assert_array_equal(batches[1][0][1], items[3][0])
    (tmp_path / "sub").mkdir()
    for n in ("b.png", "a.jpg", "sub/c.png"):
        (tmp_path / n).write_bytes(b"x")
    assert data_utils.get_custom_imgs(str(tmp_path)) == [str(tmp_path / "a.jpg"), str(tmp_path / "b.png")]
    assert data_utils.get_total_item_size({"splits": {"train": 5, "validation": 7, "test": 9}}, "train+validation") == 12
    assert data_utils.get_total_item_size({"splits": {"train": 5, "validation": 7, "test": 9}}, "test") == 9
    its = list(data_utils.synthetic_voc_items(3, 21, seed=1))
    assert all(it["image"].dtype == np.uint8 and it["image"].ndim == 3 for it in its)
    assert all(it["objects"]["label"].max() <= 19 and len(it["objects"]["bbox"]) == len(it["objects"]["is_difficult"]) for it in its)


def test_make_tf_golden_tool_stays_runnable():
    """tools/make_tf_golden.py is the committed route from "parity unpinned" to "pinned" (it needs TensorFlow and a
    checkout of the reference, neither exists here): keep it runnable -- it parses, `--help` works without
    TensorFlow, it takes the reference ONLY from `--reference` (no hard-wired path), and every file it writes is
    one tests/test_tf_golden.py consumes (and vice versa)."""
    import ast, re, subprocess
    tool = os.path.join(REPO, "tools", "make_tf_golden.py")
    src = open(tool).read()
    ast.parse(src)
    r = subprocess.run([sys.executable, tool, "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--reference" in r.stdout
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--reference" in r.stderr            # required argument, argparse's own error
    assert "/root/reference" not in src
    written = set(re.findall(r'"(tf_[a-z_%]+\.npz)"', src))
    consumer = open(os.path.join(REPO, "tests", "test_tf_golden.py")).read()
    read = set(re.findall(r'_tf\("(tf_[a-z_%]+\.npz)"', consumer))
    assert written == read and len(written) >= 6, (written, read)


def test_shipped_kernel_tables_were_measured_at_this_build():
    """Every table under tf-ssd_amd/tables/ records the ``ssd_build_id()`` it was measured on; the id is a hash of the
    kernel sources, the public header and the compile flags (csrc/build.sh, bench.build_id_from_sources).  A kernel
    change without re-measured tables would pin choices that are still valid but possibly slower, and the bench line
    would report ``kernel_table.stale``: caught here, without a GPU."""
    import glob
    import bench
    want = bench.build_id_from_sources()
    tables = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf-ssd_amd", "tables", "*.tune")))
    assert len(tables) >= 20
    stale = []
    for t in tables:
        head = [l for l in open(t).read().splitlines() if l.startswith("#build=")]
        if not head or head[0].split("=", 1)[1] != want:
            stale.append(os.path.basename(t))
    assert not stale, "tables measured at another build than %s (re-run tools/make_tuning_tables.py): %s" % (want, stale[:4])


def test_serving_setup_is_an_explicit_opt_in_placed_before_the_runtime_starts():
    """importing the package must not change the process environment.  The measured serving setup (three
    in-order lanes on three hardware queues) is an explicit opt-in -- ``ssd_hip.configure_serving()`` (what ``predictor.py``
    and ``bench.py`` call first thing) or SSD_HIP_HW_QUEUES=3 -- placed before torch / the HIP runtime start, and never
    overrides a value the process already chose; ``get_decoder_model`` defaults to that many lanes in auto mode."""
    pkg = os.path.join(REPO, "tf-ssd_amd")
    code = ("import os, sys; sys.path.insert(0, %r); import ssd_hip; %s; from models import decoder; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), decoder.default_lanes())")
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SSD_HIP_HW_QUEUES", "SSD_HIP_LANES")}
    run = lambda env, call="pass": subprocess.check_output([sys.executable, "-c", code % (pkg, call)], env=env, text=True).split()
    assert run(base) == ["None", "1"]                                       # a plain import leaves the environment alone
    assert run(base, "assert ssd_hip.configure_serving()") == ["3", "3"]
    assert run(base, "assert ssd_hip.configure_serving(2)") == ["2", "2"]
    assert run(dict(base, SSD_HIP_HW_QUEUES="3")) == ["3", "3"]             # the same opt-in through the environment
    assert run(dict(base, GPU_MAX_HW_QUEUES="8"), "assert not ssd_hip.configure_serving()") == ["8", "1"]   # the process chose: left alone
    assert run(dict(base, GPU_MAX_HW_QUEUES="2")) == ["2", "2"]
    src = open(os.path.join(pkg, "predictor.py")).read()
    assert src.index("ssd_hip.configure_serving()") < src.index("from utils import")


_RANKS_WORKER = '''
import os, sys
sys.path[:0] = [%(repo)r]
import torch.distributed as dist
dist.init_process_group("gloo")
import bench
rep = bench.ranks_report(dist, 0.010 * (1 + dist.get_rank()), 10)
assert rep["ranks_seen"] == 2 and rep["world_size"] == 2 and rep["backend"] == "gloo", rep
assert abs(rep["ms_per_step_rank_min"] - 1.0) < 1e-9 and abs(rep["ms_per_step_rank_max"] - 2.0) < 1e-9, rep
print("rank %%d ok" %% dist.get_rank(), flush=True)
dist.destroy_process_group()
'''


def test_two_process_gloo_ranks_report(tmp_path):
    """N > 1 bench lines state what the collective saw: `ranks_seen` (an all-reduce of ones) and the per-rank spread of the
    step time -- with world size 2 over gloo here, over RCCL on a multi-GPU node."""
    script = tmp_path / "ranks_worker.py"
    script.write_text(_RANKS_WORKER % {"repo": REPO})
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_augmentation_host_logic_vs_oracle():
    """The host side of ``tf-ssd_amd/augmentation.py`` (draws -> integer geometry, box arithmetic, the sampler's acceptance
    rule) against oracle/augment_oracle.py, without a device: bit-exact float32 box math for random draws, the sampler's
    windows valid and accepted by the ORACLE's rule."""
    import augmentation as aug
    from oracle import augment_oracle as ao
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = int(rng.integers(8, 400)), int(rng.integers(8, 400))
        draws = (rng.uniform(1.0, 4.0), rng.random(), rng.random())
        assert aug.expand_geometry(h, w, *draws) == ao.expand_geometry(h, w, *draws)
        n = int(rng.integers(1, 6))
        c, s = rng.uniform(0.1, 0.9, (n, 2)), rng.uniform(0.02, 0.4, (n, 2))
        g = np.clip(np.concatenate([c - s, c + s], 1), 0, 1).astype(np.float32)
        fh, fw, pt, pl = aug.expand_geometry(h, w, *draws)
        _, og, _ = ao.expand_image(np.zeros((h, w, 3), np.float32), g, *draws)
        np.testing.assert_array_equal(aug.expand_boxes(g, h, w, fh, fw, pt, pl), og)
        np.testing.assert_array_equal(aug.flip_boxes(g), ao.flip_horizontally(np.zeros((2, 2, 3), np.float32), g)[1])
        mm = np.sort(rng.random(4).astype(np.float32).reshape(2, 2), 0).reshape(4)
        if mm[2] > mm[0] and mm[3] > mm[1]:
            np.testing.assert_array_equal(aug.renormalize(g, mm), ao.renormalize_bboxes_with_min_max(g, mm))
    aug.seed(3)
    g = np.array([[0.2, 0.3, 0.6, 0.7], [0.5, 0.1, 0.9, 0.4]], np.float32)
    for mo in (0.1, 0.3, 0.5, 0.7, 0.9):
        for _ in range(30):
            y, x, hh, ww = aug.sample_distorted_bounding_box(200, 300, g, mo)
            assert 0 <= y and 0 <= x and 0 < hh and 0 < ww and y + hh <= 200 and x + ww <= 300
            assert y + hh < 200 or hh == 200, "Uniform(n) is exclusive: the last offset is never drawn"
            assert (y, x, hh, ww) == (0, 0, 200, 300) or ao.satisfies_overlap((y, x, y + hh, x + ww), g, mo, 200, 300)
    with pytest.raises(ValueError):
        aug.sample_distorted_bounding_box(200, 300, np.zeros((0, 4), np.float32), 0.5)       # TF raises on an empty list
    # zero-area padding rows never satisfy; a window without a pixel never does
    assert not ao.satisfies_overlap((0, 0, 200, 300), np.zeros((3, 4), np.float32), 0.1, 200, 300)
    assert not aug.window_satisfies((0, 0, 200, 300), aug.pixel_rectangles(np.zeros((3, 4), np.float32), 200, 300), 0.1)
    assert aug.sample_distorted_bounding_box(200, 300, np.zeros((3, 4), np.float32), 0.1) == (0, 0, 200, 300)
    # integer pixel rectangles: a box of 0.9 pixels (truncates to zero area) is skipped, one of exactly one pixel counts
    tiny = np.array([[0.5, 0.5, 0.5 + 0.9 / 200, 0.5 + 0.9 / 300]], np.float32)
    one = np.array([[0.5, 0.5, 0.5 + 1.01 / 200, 0.5 + 1.01 / 300]], np.float32)
    for box, want in ((tiny, False), (one, True)):
        assert ao.satisfies_overlap((90, 140, 110, 160), box, 0.5, 200, 300) is want
        assert aug.window_satisfies((90, 140, 110, 160), aug.pixel_rectangles(box, 200, 300), 0.5) is want
    aug.seed(1)
    a = [aug.get_random_bool() for _ in range(400)]
    assert 120 < sum(a) < 280
    assert {float(aug.get_random_min_overlap()) for _ in range(200)} == {float(np.float32(v)) for v in (0.1, 0.3, 0.5, 0.7, 0.9)}

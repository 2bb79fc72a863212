"""N3: the pure-Python HDF5 reader / writer behind ``load_weights`` /
``save_weights`` (reference utils/io_utils.py:17-29, predictor.py:46, trainer.py:65).

The reader is pinned against fixtures written by the REAL HDF5 library (h5py 3.3 / libhdf5
1.10.6, tests/golden/make_keras_h5.py) in the layout Keras' ``save_weights`` produces; the
writer's output is re-read by our reader and, when an h5py-capable interpreter exists in the
container (/opt/conda/bin/python3.9), by h5py itself."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CONDA_PY = "/opt/conda/bin/python3.9"


def _expected():
    with np.load(os.path.join(GOLD, "keras_tiny_expected.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("fixture,subset", [
    ("keras_tiny_weights.h5", None),            # h5py defaults: superblock v0, symbol-table groups, fixed strings
    ("keras_tiny_model.h5", None),              # model.save style: /model_weights + vlen string attributes
    ("keras_tiny_chunked_names.h5", None),      # layer_names0/1/... (Keras above 64 KB)
    ("keras_tiny_latest.h5", 9),                # libver='latest': superblock v3, OHDR, link messages
])
def test_reader_on_h5py_written_fixtures(fixture, subset):
    from utils import h5_reader
    exp = _expected()
    w = h5_reader.load_keras_weights(os.path.join(GOLD, fixture))
    assert len(w) == (subset or len(exp))
    for k, v in w.items():
        assert v.dtype == np.float32 and v.shape == exp[k].shape, k
        np.testing.assert_array_equal(v, exp[k])
    assert "l2_normalization/scale" in w or subset       # reference's unnamed tf.Variable -> "scale"


def test_reader_navigation_and_errors(tmp_path):
    from utils import h5_reader
    f = h5_reader.H5File(os.path.join(GOLD, "keras_tiny_model.h5"))
    assert set(f.root.keys()) == {"model_weights"}
    assert f.root.attrs["backend"] == b"tensorflow"
    assert b"toy" in f.root.attrs["model_config"]                    # variable-length string (global heap)
    g = f.root["model_weights"]
    names = [n.decode() for n in g.attrs["layer_names"]]
    assert names[:3] == ["input_1", "Conv1_pad", "Conv1"] and len(names) == 10
    d = g["bn_Conv1/bn_Conv1/gamma:0"]
    assert d.shape == (32,) and d.dtype == np.dtype("<f4")
    assert g["scalar_layer/scalar_layer/step:0"].read().shape == ()
    assert g["input_1"].keys() == [] and len(g["input_1"].attrs["weight_names"]) == 0
    with pytest.raises(KeyError):
        g["nope"]
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"PK\x03\x04 definitely not hdf5")
    assert not h5_reader.is_hdf5(str(bad))
    with pytest.raises(h5_reader.H5Error):
        h5_reader.H5File(str(bad))
    # valid HDF5, but a group without layer_names is not a Keras weights file
    from utils import h5_writer
    nk = str(tmp_path / "plain.h5")
    h5_writer.save_keras_weights(nk, {"a/kernel": np.zeros((2, 2), np.float32)})
    raw = bytearray(open(nk, "rb").read())
    raw = raw.replace(b"layer_names", b"layer_nomes")
    open(nk, "wb").write(bytes(raw))
    with pytest.raises(h5_reader.H5Error):
        h5_reader.load_keras_weights(nk)


def _full_table(backbone):
    from oracle import net_oracle as no
    import helpers
    hp = helpers.hyper_params(backbone)
    rng = np.random.default_rng(9)
    return {n: rng.standard_normal(s).astype(np.float32) for n, s in no.param_specs(backbone, hp)}


@pytest.mark.parametrize("backbone", ["mobilenet_v2", "vgg16"])
def test_writer_roundtrip_full_parameter_tables(tmp_path, backbone):
    """Every variable of both graphs (300 / 71 arrays, up to 1024x1024x... VGG conv6 = 18.9 MB)
    through writer -> reader, bit-exact, Keras names and layouts; and through the real HDF5
    library when available."""
    from utils import h5_reader, h5_writer
    w = _full_table(backbone)
    order = []
    for k in w:
        l = k.rsplit("/", 1)[0]
        if l not in order:
            order.append(l)
    path = str(tmp_path / ("ssd_%s_model_weights.h5" % backbone))
    h5_writer.save_keras_weights(path, w, layer_order=["input_1"] + order)
    assert h5_reader.is_hdf5(path)
    got = h5_reader.load_keras_weights(path)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])
    f = h5_reader.H5File(path)
    assert [n.decode() for n in f.root.attrs["layer_names"]] == ["input_1"] + order
    assert f.root.attrs["backend"] == b"tensorflow"
    if backbone == "vgg16":
        assert [n.decode() for n in f.root["l2_normalization"].attrs["weight_names"]] == ["l2_normalization/Variable:0"]
    if not os.path.exists(CONDA_PY):
        pytest.skip("no h5py-capable interpreter in this container: real-library cross-check skipped")
    np.savez(str(tmp_path / "exp.npz"), **w)
    script = (
        "import h5py, numpy as np, sys\n"
        "f = h5py.File(sys.argv[1], 'r'); exp = np.load(sys.argv[2]); n = 0\n"
        "names = [x.decode() for x in f.attrs['layer_names']]\n"
        "for l in names:\n"
        "    g = f[l]\n"
        "    for wn in g.attrs['weight_names']:\n"
        "        wn = wn.decode(); var = wn.split('/')[-1].split(':')[0]\n"
        "        var = 'scale' if var == 'Variable' else var\n"
        "        assert g[wn].dtype == np.float32 and np.array_equal(g[wn][()], exp[l + '/' + var]), wn\n"
        "        n += 1\n"
        "print('H5PY_OK', n)\n")
    out = subprocess.run([CONDA_PY, "-c", script, path, str(tmp_path / "exp.npz")], capture_output=True, text=True,
                         env={"PATH": "/usr/bin:/bin"})
    if out.returncode != 0 and "No module named 'h5py'" in out.stderr:
        pytest.skip("interpreter without h5py")
    assert out.returncode == 0, out.stderr[-2000:]
    assert "H5PY_OK %d" % len(w) in out.stdout

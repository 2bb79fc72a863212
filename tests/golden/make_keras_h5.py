"""Generates the Keras-layout HDF5 fixtures for the pure-Python HDF5 reader
(tf-ssd_amd/utils/h5_reader.py) with the REAL HDF5 library: run with an interpreter that has
h5py (in the build container: /opt/conda/bin/python3.9, h5py 3.3.0 / libhdf5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py

The files mimic what TF-2.0 Keras `save_weights_to_hdf5_group` writes (layer_names /
weight_names fixed-length string attributes, `<layer>/<weight name>` contiguous float32
datasets, backend / keras_version attributes) -- TensorFlow itself is not installable here, so
the layout follows tensorflow/python/keras/saving/hdf5_format.py from source knowledge ([3P]).
Outputs (small, committed): keras_tiny_weights.h5 (h5py defaults = what Keras produces),
keras_tiny_model.h5 (`model.save` style: weights under /model_weights, a variable-length string
attribute), keras_tiny_latest.h5 (libver='latest': superblock v3, v2 object headers, link
messages), keras_tiny_chunked_names.h5 (layer_names split over layer_names0/1 like Keras does
above 64 KB), keras_tiny_expected.npz (the same arrays keyed '<layer>/<variable>')."""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2024)

# (layer, [(keras weight name, shape)]) -- names/shapes of the MobileNetV2-SSD / VGG16-SSD graphs
LAYERS = [
    ("input_1", []),
    ("Conv1_pad", []),
    ("Conv1", [("Conv1/kernel:0", (3, 3, 3, 32))]),
    ("bn_Conv1", [("bn_Conv1/gamma:0", (32,)), ("bn_Conv1/beta:0", (32,)),
                  ("bn_Conv1/moving_mean:0", (32,)), ("bn_Conv1/moving_variance:0", (32,))]),
    ("Conv1_relu", []),
    ("expanded_conv_depthwise", [("expanded_conv_depthwise/depthwise_kernel:0", (3, 3, 32, 1))]),
    ("expanded_conv_project", [("expanded_conv_project/kernel:0", (1, 1, 32, 16))]),
    ("toy_dense", [("toy_dense/kernel:0", (1, 1, 16, 8)), ("toy_dense/bias:0", (8,))]),     # not a layer of the SSD graphs
    ("l2_normalization", [("l2_normalization/Variable:0", (512,))]),   # reference models/ssd_vgg16.py:25-28
    ("scalar_layer", [("scalar_layer/step:0", ())]),
]


def values():
    out = {}
    for layer, ws in LAYERS:
        for name, shape in ws:
            out[name] = rng.standard_normal(shape).astype(np.float32)
    return out


def save_attributes(group, name, data, chunk=None, vlen=False):
    """Keras passes a list of bytes; h5py 2.10 (the reference's environment.yml) stores that as a
    FIXED-length string array (np.array(list) -> dtype 'S<n>'), h5py >= 3 as variable-length
    strings.  Both occur in the wild: the default fixture uses the fixed form, the `model` one vlen."""
    conv = (lambda d: d) if vlen else (lambda d: np.array(d) if len(d) else np.zeros((0,), "S1"))
    if chunk is None:
        group.attrs[name] = conv(data)
    else:
        for i in range(0, len(data), chunk):
            group.attrs["%s%d" % (name, i // chunk)] = conv(data[i:i + chunk])


def write_weights(g, vals, layers, chunk=None, vlen=False):
    save_attributes(g, "layer_names", [l.encode("utf8") for l, _ in layers], chunk, vlen)
    g.attrs["backend"] = "tensorflow".encode("utf8")
    g.attrs["keras_version"] = "2.2.4-tf".encode("utf8")
    for layer, ws in layers:
        lg = g.create_group(layer)
        save_attributes(lg, "weight_names", [n.encode("utf8") for n, _ in ws], None, vlen)
        for n, shape in ws:
            val = vals[n]
            d = lg.create_dataset(n, val.shape, dtype=val.dtype)
            if not val.shape:
                d[()] = val
            else:
                d[:] = val


def main():
    vals = values()
    with h5py.File(os.path.join(HERE, "keras_tiny_weights.h5"), "w") as f:
        write_weights(f, vals, LAYERS)
    with h5py.File(os.path.join(HERE, "keras_tiny_model.h5"), "w") as f:
        f.attrs["keras_version"] = "2.2.4-tf".encode("utf8")
        f.attrs["backend"] = "tensorflow".encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": "toy"}})   # str -> vlen
        write_weights(f.create_group("model_weights"), vals, LAYERS, vlen=True)
    with h5py.File(os.path.join(HERE, "keras_tiny_latest.h5"), "w", libver="latest") as f:
        write_weights(f, vals, LAYERS[2:8])
    with h5py.File(os.path.join(HERE, "keras_tiny_chunked_names.h5"), "w") as f:
        write_weights(f, vals, LAYERS, chunk=4)
    exp = {}
    for layer, ws in LAYERS:
        for n, _ in ws:
            var = n.split("/")[-1].split(":")[0]
            exp["%s/%s" % (layer, "scale" if var.startswith("Variable") else var)] = vals[n]
    np.savez(os.path.join(HERE, "keras_tiny_expected.npz"), **exp)
    for fn in sorted(os.listdir(HERE)):
        if fn.startswith("keras_tiny"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()

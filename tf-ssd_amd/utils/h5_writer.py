"""Minimal pure-Python HDF5 WRITER for Keras-layout weight files (no h5py in this image).

Counterpart of h5_reader.py: writes the classic (libver 'earliest') subset of the HDF5 file
format -- superblock v0, v1 object headers, groups as symbol tables (one level-0 B-tree node,
SNOD leaves, local heap), fixed-length string / scalar attributes, contiguous little-endian
datasets -- so that checkpoints written by ``SSDModel.save_weights('*.h5')`` open in h5py /
Keras ``load_weights`` (reference trainer.py:65, predictor.py:46, utils/io_utils.py:17-29).
tests/test_host_cpu.py re-reads the output with the real HDF5 library when an h5py-capable
interpreter is present in the container.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 16          # SNOD holds up to 2*LEAF_K symbols
INTERNAL_K = 16      # a B-tree node holds up to 2*INTERNAL_K children
OBJECT_HEADER_LIMIT = 64512      # Keras' HDF5_OBJECT_HEADER_LIMIT: longer name lists are chunked


def _pad8(b):
    return b + bytes((-len(b)) % 8)


class _File(object):
    def __init__(self):
        self.buf = bytearray(96)     # superblock v0 (56 bytes) + root symbol table entry (40)

    def alloc(self, data):
        self.buf += bytes((-len(self.buf)) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", d) for d in shape)


_F32 = bytes.fromhex("11201f000400000000002000170800177f000000")       # IEEE f32 little-endian


def _datatype(dt):
    dt = np.dtype(dt)
    if dt == np.dtype("<f4"):
        return _F32
    if dt.kind == "S":
        # class 3 (string) v1, null-padded ASCII
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)
    if dt.kind in "iu" and dt.byteorder in "<=|":
        bits = 0x08 if dt.kind == "i" else 0
        return struct.pack("<BBBBIHH", 0x10, bits, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    raise ValueError("h5_writer: unsupported dtype %s" % dt)


def _attribute(name, value):
    nb = name.encode("utf-8") + b"\0"
    if isinstance(value, (bytes, str)):
        v = value.encode("utf-8") if isinstance(value, str) else value
        arr = np.array(v, dtype="S%d" % max(len(v), 1))
    else:
        arr = np.asarray(value)
        if arr.dtype.kind == "U" or arr.dtype == object:
            arr = np.array([x.encode("utf-8") if isinstance(x, str) else x for x in arr.reshape(-1)]).reshape(arr.shape)
        if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
            arr = arr.astype("S1")
        if arr.dtype.kind == "f":
            arr = arr.astype("<f4")
    dt, ds = _datatype(arr.dtype), _dataspace(arr.shape)
    head = struct.pack("<BBHHH", 1, 0, len(nb), len(dt), len(ds))
    return _msg(0x000C, head + _pad8(nb) + _pad8(dt) + _pad8(ds) + arr.tobytes())


def _chunk_names(names):
    """Keras save_attributes_to_hdf5_group: split so that each chunk stays under the limit."""
    data = [n.encode("utf-8") if isinstance(n, str) else n for n in names]
    arr = np.array(data) if data else np.zeros((0,), "S1")
    if arr.nbytes <= OBJECT_HEADER_LIMIT:
        return [arr]
    k = 1
    parts = np.array_split(arr, k)
    while any(p.nbytes > OBJECT_HEADER_LIMIT for p in parts):
        k += 1
        parts = np.array_split(arr, k)
    return parts


def _write_dataset(f, arr):
    arr = np.array(arr, dtype="<f4", order="C")        # (keeps 0-d shapes)
    data_addr = f.alloc(arr.tobytes()) if arr.size else UNDEF
    msgs = [
        _msg(0x0001, _dataspace(arr.shape)),
        _msg(0x0003, _datatype(arr.dtype), flags=1),
        _msg(0x0005, bytes.fromhex("0202020100000000"), flags=1),       # fill value v2: late alloc, ifset, size 0
        _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)),  # layout v3, contiguous
    ]
    return f.alloc(_object_header(msgs))


def _write_group(f, children, attrs):
    """children: {name: (object header address, btree address or None, heap address or None)}.
    Returns (header address, btree address, heap address)."""
    names = sorted(children, key=lambda s: s.encode("utf-8"))
    if len(names) > 4 * LEAF_K * INTERNAL_K:
        raise ValueError("h5_writer: too many members in one group (%d)" % len(names))
    # local heap data segment: "" at offset 0, then the names (8-byte aligned), then one free block
    seg = bytearray(8)
    offs = {}
    for n in names:
        offs[n] = len(seg)
        seg += _pad8(n.encode("utf-8") + b"\0")
    free_off = len(seg)
    seg += struct.pack("<QQ", 1, 16)            # free block: next = H5HL_FREE_NULL (1), size 16
    seg_addr = f.alloc(bytes(seg))
    heap_addr = f.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_off, seg_addr))
    # SNOD leaves
    per = 2 * LEAF_K
    leaves = []
    for i in range(0, max(len(names), 1), per):
        part = names[i:i + per]
        body = b"SNOD" + struct.pack("<BBH", 1, 0, len(part))
        for n in part:
            hdr, bt, hp = children[n]
            if bt is not None:
                body += struct.pack("<QQII", offs[n], hdr, 1, 0) + struct.pack("<QQ", bt, hp)
            else:
                body += struct.pack("<QQII", offs[n], hdr, 0, 0) + bytes(16)
        body += bytes(8 + per * 40 - len(body))         # nodes are allocated at their full size
        leaves.append((f.alloc(body), offs[part[-1]] if part else 0))
    if not names:
        leaves = []
    # one level-0 B-tree node over the leaves
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(leaves), UNDEF, UNDEF)
    node += struct.pack("<Q", 0)                         # key 0: the empty string
    for addr, last_off in leaves:
        node += struct.pack("<QQ", addr, last_off)       # child i, key i+1 = greatest name in child i
    node += bytes(24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(node))
    bt_addr = f.alloc(node)
    msgs = [_msg(0x0011, struct.pack("<QQ", bt_addr, heap_addr))]
    for k, v in attrs:
        msgs.append(_attribute(k, v))
    return f.alloc(_object_header(msgs)), bt_addr, heap_addr


def save_keras_weights(path, weights, layer_order=None, weight_order=None):
    """Write ``{"<layer>/<variable>": array}`` as a Keras ``save_weights`` HDF5 file:
    root attrs ``layer_names`` / ``backend`` / ``keras_version``; one group per layer with a
    ``weight_names`` attr (``<layer>/<variable>:0``) and datasets at ``/<layer>/<layer>/<variable>:0``.
    ``layer_order``: layer names in model order (default: first appearance in ``weights``).
    The reference's L2Normalization scale is written under its TF name ``Variable:0``."""
    layers = {}
    for key in weights:
        layer, var = key.rsplit("/", 1)
        layers.setdefault(layer, []).append(var)
    order = list(layer_order) if layer_order is not None else list(layers)
    for l in layers:
        if l not in order:
            order.append(l)
    f = _File()
    top = {}
    for layer in order:
        vars_ = layers.get(layer, [])
        wnames = []
        inner = {}
        for var in vars_:
            kvar = "Variable" if (layer.startswith("l2_normalization") and var == "scale") else var
            wnames.append("%s/%s:0" % (layer, kvar))
            hdr = _write_dataset(f, weights["%s/%s" % (layer, var)])
            inner["%s:0" % kvar] = (hdr, None, None)
        attrs = [("weight_names%s" % ("" if len(_chunk_names(wnames)) == 1 else i), c)
                 for i, c in enumerate(_chunk_names(wnames))]
        if vars_:
            sub = _write_group(f, inner, [])
            top[layer] = _write_group(f, {layer: sub}, attrs)
        else:
            top[layer] = _write_group(f, {}, attrs)
    chunks = _chunk_names(order)
    rattrs = [("layer_names%s" % ("" if len(chunks) == 1 else i), c) for i, c in enumerate(chunks)]
    rattrs += [("backend", b"tensorflow"), ("keras_version", b"2.2.4-tf")]
    root_hdr, root_bt, root_hp = _write_group(f, top, rattrs)
    eof = len(f.buf) + ((-len(f.buf)) % 8)
    f.buf += bytes(eof - len(f.buf))
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_bt, root_hp)
    assert len(sb) == 96
    f.buf[0:96] = sb
    with open(path, "wb") as fh:
        fh.write(bytes(f.buf))

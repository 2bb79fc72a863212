"""Where a net's kernel choices come from, in order:

1. ``SSD_HIP_TUNE_CACHE=<dir>`` (explicit, read/write): a table saved there for this build + device;
2. the tables SHIPPED in ``tf-ssd_amd/tables/`` (measured on an MI355X for the BASELINE.json shapes and
   the shapes the GPU tests use, ``tools/make_tuning_tables.py``): the default -- no timing at all, so the
   same kernels and the same bits in every process;
3. a process-wide memo: the second net of the same (graph, anchors, batch, options) in a process gets the
   first one's table, so two instances never differ;
4. the on-device autotune of ``ssd_net_finalize`` (hipEvent timing of every candidate) -- only on a
   miss, and then only for the layers the table does not cover.  ``SSD_HIP_AUTOTUNE=0`` turns a miss
   into an error (deployments that require reproducible kernel selection).

A table is text: one ``layer config split_k`` line per conv layer, ``block image 0|1`` per whole-image
block candidate, ``__launch graph 0|1``; ``#key=value`` header lines carry provenance."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SHIPPED_DIR = os.path.join(HERE, "tables")

_MEMO = {}


def table_key(backbone, img_size, total_labels, aspect_ratios, batch):
    ars = "-".join(str(len(a)) for a in aspect_ratios)
    return "%s_%d_%d_a%s_b%d" % (backbone, int(img_size), int(total_labels), ars, int(batch))


def options_key(options):
    return ",".join("%s=%d" % kv for kv in sorted(options.items()))


def body(text):
    """The lines the native parser reads (no header / comment lines)."""
    return "".join(l + "\n" for l in text.splitlines() if l.strip() and not l.startswith("#"))


def header(text):
    out = {}
    for l in text.splitlines():
        if l.startswith("#") and "=" in l:
            k, v = l[1:].split("=", 1)
            out[k.strip()] = v.strip()
    return out


def sha16(text):
    return hashlib.sha256(body(text).encode()).hexdigest()[:16]


def shipped_path(key):
    return os.path.join(SHIPPED_DIR, key + ".tune")


def load_shipped(key):
    p = shipped_path(key)
    if os.path.exists(p):
        with open(p) as f:
            return f.read()
    return None


def memo_get(key, opts):
    return _MEMO.get((key, opts))


def memo_put(key, opts, text):
    _MEMO[(key, opts)] = text


def memo_clear():
    _MEMO.clear()


def with_header(text, **kv):
    return "".join("#%s=%s\n" % (k, v) for k, v in kv.items()) + body(text)

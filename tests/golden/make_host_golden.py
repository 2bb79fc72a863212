"""Golden vectors for the pure-Python host logic, produced by EXECUTING the
reference's own functions in this container: the function / dict definitions are extracted from the
reference modules with ``ast`` (their top-level ``import tensorflow`` cannot run here, the extracted
pieces are TF-free) and called on the inputs below.  Only inputs and outputs are stored
(``host_logic.json``); no reference source text is written anywhere.

    python tests/golden/make_host_golden.py        # needs /root/reference (not available on the GPU box)
"""
import ast, copy, json, math, os, sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_logic.json")


def extract(path, names):
    tree = ast.parse(open(path).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            keep.append(node)
    ns = {"math": math, "os": os}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    tu = extract(os.path.join(REF, "utils/train_utils.py"), {"SSD", "get_hyper_params", "scheduler", "get_step_size"})
    io = extract(os.path.join(REF, "utils/io_utils.py"), {"get_model_path", "is_valid_backbone"})
    pristine = copy.deepcopy(tu["SSD"])
    cases = [
        ["mobilenet_v2", {}],
        ["vgg16", {}],
        ["mobilenet_v2", {"img_size": 512, "feature_map_shapes": [32, 16, 8, 4, 2, 1]}],
        ["vgg16", {"iou_threshold": 0.45, "neg_pos_ratio": 2, "loc_loss_alpha": 0.5, "variances": [1, 1, 1, 1]}],
        ["mobilenet_v2", {"iou_threshold": 0, "neg_pos_ratio": 0.0, "img_size": None}],        # falsy overrides are ignored
        ["mobilenet_v2", {"total_labels": 21, "unknown_key": 7}],                               # keys not in the dict are ignored
        ["vgg16", {"aspect_ratios": [[1.0], [1.0, 2.0], [1.0], [1.0], [1.0], [1.0]]}],
    ]
    hyper = []
    for backbone, kw in cases:
        tu["SSD"].clear()
        tu["SSD"].update(copy.deepcopy(pristine))          # each case starts from the module's initial state
        hyper.append({"backbone": backbone, "kwargs": kw, "out": copy.deepcopy(tu["get_hyper_params"](backbone, **kw))})
    # the reference mutates its module-level dict: a later call sees earlier overrides
    tu["SSD"].clear()
    tu["SSD"].update(copy.deepcopy(pristine))
    tu["get_hyper_params"]("mobilenet_v2", img_size=512)
    sticky = copy.deepcopy(tu["get_hyper_params"]("mobilenet_v2"))
    epochs = list(range(0, 200)) + [1000]
    steps = [[1, 1], [8, 8], [9, 8], [5011, 32], [4952, 32], [16551, 32], [16551, 256], [7, 64], [0, 4]]
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            paths = {m: io["get_model_path"](m) for m in ("mobilenet_v2", "vgg16")}
            made_dir = os.path.isdir("trained")
        finally:
            os.chdir(cwd)
    valid = {}
    for b in ("mobilenet_v2", "vgg16", "resnet50", "", "VGG16"):
        try:
            io["is_valid_backbone"](b)
            valid[b] = True
        except AssertionError:
            valid[b] = False
    json.dump({"source": "executed from /root/reference/utils/{train_utils,io_utils}.py via ast extraction (make_host_golden.py)",
               "get_hyper_params": hyper, "sticky_img_size_after_override": sticky,
               "scheduler": [[e, tu["scheduler"](e)] for e in epochs],
               "get_step_size": [[t, b, tu["get_step_size"](t, b)] for t, b in steps],
               "get_model_path": paths, "get_model_path_creates_dir": made_dir, "is_valid_backbone": valid},
              open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())

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


def extract(path, names, extra_ns=None):
    tree = ast.parse(open(path).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in node.targets):
            keep.append(node)
    ns = {"math": math, "os": os}
    ns.update(extra_ns or {})
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
    # ---- the remaining TF-free functions on / beside the hot path
    bb = extract(os.path.join(REF, "utils/bbox_utils.py"), {"get_scale_for_nth_feature_map"})
    scale_cases = [[k, {}] for k in range(1, 7)] + [[k, {"m": 4}] for k in range(1, 5)] + \
                  [[k, {"m": 6, "scale_min": 0.1, "scale_max": 0.95}] for k in (1, 3, 6)] + [[2, {"m": 2, "scale_min": 0.15}], [7, {}], [0, {}]]
    scales = [[k, kw, bb["get_scale_for_nth_feature_map"](k, **kw)] for k, kw in scale_cases]
    try:
        bb["get_scale_for_nth_feature_map"](1, m=1)
        scale_m1 = "no error"
    except ZeroDivisionError:
        scale_m1 = "ZeroDivisionError"
    import argparse, datetime as _dt, types

    class FrozenDatetime(object):                 # `datetime.now()` of the module under test, frozen
        @staticmethod
        def now():
            return _dt.datetime(2020, 1, 2, 3, 4, 5)
    io2 = extract(os.path.join(REF, "utils/io_utils.py"), {"get_log_path", "handle_args"},
                  {"datetime": FrozenDatetime, "argparse": argparse})
    log_paths = [[m, pf, io2["get_log_path"](m, pf)] for m, pf in (("mobilenet_v2", ""), ("vgg16", ""), ("vgg16", "_run7"), ("x", "/y"))]
    argv_cases = [[], ["-handle-gpu"], ["--backbone", "vgg16"], ["--backbone=mobilenet_v2", "-handle-gpu"], ["--backbone", "resnet50"]]
    parsed = []
    old_argv = sys.argv
    for av in argv_cases:
        sys.argv = ["prog"] + av
        a = io2["handle_args"]()
        parsed.append([av, {k: v for k, v in sorted(vars(a).items())}])
    bad_argv = []
    for av in (["--nope"], ["--backbone"]):
        sys.argv = ["prog"] + av
        try:
            devnull = open(os.devnull, "w")
            old_err, sys.stderr = sys.stderr, devnull
            try:
                io2["handle_args"]()
                bad_argv.append([av, "ok"])
            finally:
                sys.stderr = old_err
        except SystemExit as e:
            bad_argv.append([av, "SystemExit %s" % e.code])
    sys.argv = old_argv
    du = extract(os.path.join(REF, "utils/data_utils.py"), {"get_total_item_size", "get_labels", "get_custom_imgs"})
    names = ["aeroplane", "bicycle", "bird"]
    info = types.SimpleNamespace(
        splits={"train": types.SimpleNamespace(num_examples=2501), "validation": types.SimpleNamespace(num_examples=2510),
                "test": types.SimpleNamespace(num_examples=4952)},
        features={"labels": types.SimpleNamespace(names=names)})
    sizes = [[sp, du["get_total_item_size"](info, sp)] for sp in ("train", "train+validation", "validation", "test")]
    try:
        du["get_total_item_size"](info, "all")
        bad_split = "no error"
    except AssertionError:
        bad_split = "AssertionError"
    with tempfile.TemporaryDirectory() as d:
        for f in ("b.jpg", "a.png", "c.txt"):
            open(os.path.join(d, f), "w").close()
        os.makedirs(os.path.join(d, "sub"))
        open(os.path.join(d, "sub", "nested.jpg"), "w").close()
        listed = sorted(os.path.relpath(q, d) for q in du["get_custom_imgs"](d))
        missing = du["get_custom_imgs"](os.path.join(d, "does_not_exist"))
    json.dump({"source": "executed from /root/reference/utils/{train_utils,io_utils,bbox_utils,data_utils}.py via ast extraction (make_host_golden.py)",
               "get_hyper_params": hyper, "sticky_img_size_after_override": sticky,
               "scheduler": [[e, tu["scheduler"](e)] for e in epochs],
               "get_step_size": [[t, b, tu["get_step_size"](t, b)] for t, b in steps],
               "get_model_path": paths, "get_model_path_creates_dir": made_dir, "is_valid_backbone": valid,
               "get_scale_for_nth_feature_map": scales, "get_scale_m1": scale_m1,
               "get_log_path_at_2020_01_02_03_04_05": log_paths, "handle_args": parsed, "handle_args_bad": bad_argv,
               "get_total_item_size": sizes, "get_total_item_size_bad_split": bad_split, "get_labels": du["get_labels"](info),
               "get_custom_imgs_sorted_relative": listed, "get_custom_imgs_missing_dir": missing},
              open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())

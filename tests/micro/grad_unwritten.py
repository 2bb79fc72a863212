"""Diagnostics: which trainable parameters does the native backward never write?  (flat gradient NaN-prefilled)"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tf-ssd_amd", os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import numpy as np, torch
import helpers
for backbone in ("mobilenet_v2", "vgg16"):
    if backbone == "mobilenet_v2":
        from models.ssd_mobilenet_v2 import get_model
    else:
        from models.ssd_vgg16 import get_model
    hp = helpers.hyper_params(backbone)
    m = get_model(hp); m.set_weights(helpers.synthetic_weights(backbone, hp)); m.compile()
    B = 2
    x = helpers.images(B, 300, seed=31)
    from utils import bbox_utils, train_utils
    pri = bbox_utils.generate_prior_boxes(hp["feature_map_shapes"], hp["aspect_ratios"])
    gt, gl = helpers.gt_inputs(B, seed=6)
    yd, yl = train_utils.calculate_actual_outputs(pri, gt, gl, hp)
    m.forward_backward(x, yd, yl)
    if len(sys.argv) > 1:
        print("bucket starts", m._plan_gradient_buckets(B, int(sys.argv[1])))
    m._grads.fill_(float("nan")); torch.cuda.synchronize()
    _, _, g = m.forward_backward(x, yd, yl)
    g = g.cpu().numpy()
    for name, (off, shape) in m.trainable_offsets().items():
        n = int(np.prod(shape)); bad = int(np.isnan(g[off:off + n]).sum())
        if bad: print(backbone, name, shape, "unwritten", bad, "of", n)
    print(backbone, "total nan", int(np.isnan(g).sum()), "of", g.size)

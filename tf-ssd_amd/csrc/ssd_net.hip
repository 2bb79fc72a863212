// Graph runner: builds the SSD300 MobileNetV2 / VGG16 forward graphs natively, owns the
// parameters (Keras names + layouts at the boundary), folds BatchNorm, packs weights for
// the MFMA kernels, plans the activation arena in HBM, autotunes the tile configuration of
// every conv on the device, and replays the layer list on a HIP stream.
//
// Reference graphs: models/ssd_mobilenet_v2.py:7-35 (+ [3P] keras-applications 1.0.8
// MobileNetV2), models/ssd_vgg16.py:33-97, models/header.py:43-67.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ssd_net.h"

using namespace ssd;

namespace ssd {

// ------------------------------------------------------------------ graph builder
struct Builder {
    ssd_net& net;
    explicit Builder(ssd_net& n) : net(n) {}

    int add_param(const std::string& name, std::vector<int> shape) {
        Param p;
        p.name = name;
        p.shape = std::move(shape);
        p.count = 1;
        for (int d : p.shape) p.count *= (size_t)d;
        net.params.push_back(p);
        net.param_index[name] = (int)net.params.size() - 1;
        return (int)net.params.size() - 1;
    }
    int add_bn(const std::string& name, int c) {
        const int g = add_param(name + "/gamma", {c});
        add_param(name + "/beta", {c});
        add_param(name + "/moving_mean", {c});
        add_param(name + "/moving_variance", {c});
        return g;
    }
    int add_tensor(const std::string& name, int H, int W, int C) {
        Tensor t;
        t.name = name;
        t.H = H; t.W = W; t.C = C;
        t.per_image = (size_t)H * W * C;
        net.tensors.push_back(t);
        net.tensor_index[name] = (int)net.tensors.size() - 1;
        return (int)net.tensors.size() - 1;
    }
    // pad_mode: 0 valid, 1 same, 2 keras correct_pad (ZeroPadding2D before a stride-2 VALID conv)
    void pads(int in, int k, int stride, int dil, int mode, int* before, int* after) {
        *before = *after = 0;
        if (mode == 1) {
            ssd_same_pads(in, k, stride, dil, before, after);
        } else if (mode == 2) {
            const int adjust = 1 - in % 2, correct = k / 2;      // [3P] keras-applications correct_pad
            *before = correct - adjust;
            *after = correct;
        }
    }
    // Conv2D [+BN | +bias] +act [+residual].  Returns the output tensor id.
    int conv(const std::string& name, const std::string& out_name, int in, int Cout, int k, int stride,
             int pad_mode, int dil, const std::string& bn_name, bool bias, int act, int res = -1) {
        const Tensor ti = net.tensors[in];
        Layer l;
        l.name = name;
        l.kind = LK_CONV;
        l.in = in; l.res = res;
        l.H = ti.H; l.W = ti.W; l.Cin = ti.C; l.Cout = Cout;
        l.kh = l.kw = k; l.stride = stride; l.dil = dil; l.act = act;
        pads(ti.H, k, stride, dil, pad_mode, &l.pt, &l.pb);
        pads(ti.W, k, stride, dil, pad_mode, &l.pl, &l.pr);
        l.Ho = ssd_conv_out_size(ti.H, k, stride, dil, l.pt, l.pb);
        l.Wo = ssd_conv_out_size(ti.W, k, stride, dil, l.pl, l.pr);
        l.p_kernel = add_param(name + "/kernel", {k, k, ti.C, Cout});
        if (bias) l.p_bias = add_param(name + "/bias", {Cout});
        if (!bn_name.empty()) l.p_bn = add_bn(bn_name, Cout);
        if (!out_name.empty()) l.out = add_tensor(out_name, l.Ho, l.Wo, Cout);
        net.layers.push_back(l);
        return l.out;
    }
    int dwconv(const std::string& name, const std::string& out_name, int in, int stride, int pad_mode,
               const std::string& bn_name, int act) {
        const Tensor ti = net.tensors[in];
        Layer l;
        l.name = name;
        l.kind = LK_DW;
        l.in = in;
        l.H = ti.H; l.W = ti.W; l.Cin = l.Cout = ti.C;
        l.kh = l.kw = 3; l.stride = stride; l.act = act;
        pads(ti.H, 3, stride, 1, pad_mode, &l.pt, &l.pb);
        pads(ti.W, 3, stride, 1, pad_mode, &l.pl, &l.pr);
        l.Ho = ssd_conv_out_size(ti.H, 3, stride, 1, l.pt, l.pb);
        l.Wo = ssd_conv_out_size(ti.W, 3, stride, 1, l.pl, l.pr);
        l.p_kernel = add_param(name + "/depthwise_kernel", {3, 3, ti.C, 1});
        l.p_bn = add_bn(bn_name, ti.C);
        l.out = add_tensor(out_name, l.Ho, l.Wo, ti.C);
        net.layers.push_back(l);
        return l.out;
    }
    int pool(const std::string& name, int in, int k, int stride) {
        const Tensor ti = net.tensors[in];
        Layer l;
        l.name = name;
        l.kind = LK_POOL;
        l.in = in;
        l.H = ti.H; l.W = ti.W; l.Cin = l.Cout = ti.C;
        l.kh = l.kw = k; l.stride = stride;
        pads(ti.H, k, stride, 1, 1, &l.pt, &l.pb);
        pads(ti.W, k, stride, 1, 1, &l.pl, &l.pr);
        l.Ho = ssd_conv_out_size(ti.H, k, stride, 1, l.pt, l.pb);
        l.Wo = ssd_conv_out_size(ti.W, k, stride, 1, l.pl, l.pr);
        l.out = add_tensor(name, l.Ho, l.Wo, ti.C);
        net.layers.push_back(l);
        return l.out;
    }
    // models/header.py:43-67: per level a label conv (A*L) and a box conv (A*4), 3x3 SAME, bias.
    void heads(const std::vector<int>& feats) {
        long off = 0;
        net.fmap.clear();
        net.level_off.clear();
        for (size_t i = 0; i < feats.size(); ++i) {
            const Tensor t = net.tensors[feats[i]];
            net.fmap.push_back(t.H);
            net.level_off.push_back(off);
            off += (long)t.H * t.W * (net.n_ars[i] + 1);
        }
        net.num_priors = (int)off;
        // One fused conv per level: the label conv (A*L channels) and the box conv (A*4) read the
        // same feature map, so they run as ONE implicit GEMM with N = A*(L+4); the epilogue
        // routes columns < A*L to the concatenated label buffer and the rest to the box buffer.
        for (size_t i = 0; i < feats.size(); ++i) {
            const int A = net.n_ars[i] + 1;
            const std::string idx = std::to_string(i + 1);
            const Tensor ti = net.tensors[feats[i]];
            Layer l;
            l.name = idx + "_conv_heads";
            l.kind = LK_CONV;
            l.in = feats[i];
            l.H = ti.H; l.W = ti.W; l.Cin = ti.C;
            l.Cout1 = A * net.L;
            l.Cout = A * (net.L + 4);
            l.kh = l.kw = 3; l.stride = 1; l.dil = 1; l.act = SSD_ACT_NONE;
            pads(ti.H, 3, 1, 1, 1, &l.pt, &l.pb);
            pads(ti.W, 3, 1, 1, 1, &l.pl, &l.pr);
            l.Ho = ti.H; l.Wo = ti.W;
            l.p_kernel = add_param(idx + "_conv_label_output/kernel", {3, 3, ti.C, A * net.L});
            l.p_bias = add_param(idx + "_conv_label_output/bias", {A * net.L});
            l.p_kernel2 = add_param(idx + "_conv_boxes_output/kernel", {3, 3, ti.C, A * 4});
            l.p_bias2 = add_param(idx + "_conv_boxes_output/bias", {A * 4});
            l.head_kind = 3;
            l.head_off = net.level_off[i] * net.L;
            l.head_bs = (long)net.num_priors * net.L;
            l.head_ps = (long)A * net.L;
            l.head2_off = net.level_off[i] * 4;
            l.head2_bs = (long)net.num_priors * 4;
            l.head2_ps = (long)A * 4;
            // the two large head convs share one side stream; the four small, launch-latency-bound
            // ones get their own, so they run under the level-2 head instead of queueing behind it
            l.side = i < 2 ? 1 : 2;
            net.layers.push_back(l);
        }
        Layer sm;
        sm.name = "conf";
        sm.kind = LK_SOFTMAX;
        sm.Cout = net.L;
        net.layers.push_back(sm);
    }
};

static void build_mobilenet_v2(ssd_net& net) {
    Builder b(net);
    const int S = net.img_size;
    int x = b.add_tensor("input", S, S, 3);
    x = b.conv("Conv1", "Conv1_relu", x, 32, 3, 2, 2, 1, "bn_Conv1", false, SSD_ACT_RELU6);
    x = b.dwconv("expanded_conv_depthwise", "expanded_conv_depthwise_relu", x, 1, 1, "expanded_conv_depthwise_BN",
                 SSD_ACT_RELU6);
    x = b.conv("expanded_conv_project", "expanded_conv_project_BN", x, 16, 1, 1, 1, 1, "expanded_conv_project_BN",
               false, SSD_ACT_NONE);
    {   // fused stem: Conv1 -> expanded_conv_depthwise -> expanded_conv_project in one kernel
        Layer f;
        f.name = "stem_fused";
        f.kind = LK_FUSED;
        f.f_type = 1;
        f.in = 0; f.out = x;
        const Layer& l1 = net.layers[0];
        f.H = l1.H; f.W = l1.W; f.Cin = 3; f.Ho = l1.Ho; f.Wo = l1.Wo; f.Cout = 16;
        f.stride = 2; f.pt = l1.pt; f.pl = l1.pl;
        net.layers.insert(net.layers.begin(), f);
        net.layers[0].f_expand = 1; net.layers[0].f_dw = 2; net.layers[0].f_project = 3;
        for (int j = 1; j <= 3; ++j) net.layers[j].fused_by = 0;
    }
    static const int blocks[16][2] = {{24, 2}, {24, 1}, {32, 2}, {32, 1}, {32, 1}, {64, 2}, {64, 1}, {64, 1},
                                      {64, 1}, {96, 1}, {96, 1}, {96, 1}, {160, 2}, {160, 1}, {160, 1}, {320, 1}};
    int cin = 16, tap1 = -1;
    for (int k = 1; k <= 16; ++k) {
        const int cout = blocks[k - 1][0], s = blocks[k - 1][1];
        const std::string p = "block_" + std::to_string(k) + "_";
        const int inp = x;
        x = b.conv(p + "expand", p + "expand_relu", x, 6 * cin, 1, 1, 1, 1, p + "expand_BN", false, SSD_ACT_RELU6);
        if (k == 13) tap1 = x;       // block_13_expand_relu (models/ssd_mobilenet_v2.py:18)
        x = b.dwconv(p + "depthwise", p + "depthwise_relu", x, s, s == 2 ? 2 : 1, p + "depthwise_BN", SSD_ACT_RELU6);
        const bool res = (cin == cout && s == 1);
        x = b.conv(p + "project", p + "out", x, cout, 1, 1, 1, 1, p + "project_BN", false, SSD_ACT_NONE,
                   res ? inp : -1);
        int ie = (int)net.layers.size() - 3;      // expand, depthwise = ie + 1, project = ie + 2
        {
            Layer f;
            f.name = p + "fused";
            f.kind = LK_FUSED;
            f.in = inp; f.out = x;
            // block 13's expanded map is SSD feature map #1: it must reach HBM -- only the whole-image
            // kernel takes that block (it writes E out once), the tile kernel never does (Cin = 96)
            if (k == 13) f.e_out = tap1;
            const Layer& le = net.layers[ie];
            const Layer& ld = net.layers[ie + 1];
            f.H = le.H; f.W = le.W; f.Cin = le.Cin; f.Ho = ld.Ho; f.Wo = ld.Wo; f.Cout = cout;
            f.stride = s; f.pt = ld.pt; f.pl = ld.pl;
            net.layers.insert(net.layers.begin() + ie, f);    // fused layer runs first, members follow
            const int fi = ie;
            ++ie;
            net.layers[fi].f_expand = ie; net.layers[fi].f_dw = ie + 1; net.layers[fi].f_project = ie + 2;
            for (int j = 0; j < 3; ++j) net.layers[ie + j].fused_by = fi;
        }
        {   // depthwise + project as one kernel behind the expand GEMM (used when the whole-block
            // kernel above does not take the shape: Cin > 32, and for block 13)
            Layer f;
            f.name = p + "dwproj";
            f.kind = LK_FUSED;
            f.f_type = 2;
            const Layer& ld = net.layers[ie + 1];
            f.in = ld.in; f.out = x;
            f.H = ld.H; f.W = ld.W; f.Cin = ld.Cin; f.Ho = ld.Ho; f.Wo = ld.Wo; f.Cout = cout;
            f.stride = s; f.pt = ld.pt; f.pl = ld.pl;
            net.layers.insert(net.layers.begin() + ie + 1, f);
            const int fi = ie + 1;
            net.layers[fi].f_dw = fi + 1; net.layers[fi].f_project = fi + 2;
            net.layers[fi].fused_by = net.layers[ie].fused_by;     // a whole-block kernel covers it too
            if (net.layers[fi].fused_by >= 0) {                     // member indices of the block layer moved by one
                Layer& blk = net.layers[net.layers[fi].fused_by];
                blk.f_dw = fi + 1; blk.f_project = fi + 2;
            }
            net.layers[fi + 1].fused_by2 = fi;
            net.layers[fi + 2].fused_by2 = fi;
        }
        cin = cout;
    }
    x = b.conv("Conv_1", "out_relu", x, 1280, 1, 1, 1, 1, "Conv_1_bn", false, SSD_ACT_RELU6);
    std::vector<int> feats = {tap1, x};
    static const int extras[4][2] = {{256, 512}, {128, 256}, {128, 256}, {128, 256}};
    for (int i = 1; i <= 4; ++i) {
        const std::string e = "extra" + std::to_string(i);
        x = b.conv(e + "_1", e + "_1", x, extras[i - 1][0], 1, 1, 0, 1, "", true, SSD_ACT_RELU);
        x = b.conv(e + "_2", e + "_2", x, extras[i - 1][1], 3, 2, 1, 1, "", true, SSD_ACT_RELU);
        feats.push_back(x);
    }
    b.heads(feats);
}

static void build_vgg16(ssd_net& net) {
    Builder b(net);
    const int S = net.img_size;
    int x = b.add_tensor("input", S, S, 3);
    auto c3 = [&](const char* name, int cout) { x = b.conv(name, name, x, cout, 3, 1, 1, 1, "", true, SSD_ACT_RELU); };
    c3("conv1_1", 64); c3("conv1_2", 64); x = b.pool("pool1", x, 2, 2);
    c3("conv2_1", 128); c3("conv2_2", 128); x = b.pool("pool2", x, 2, 2);
    c3("conv3_1", 256); c3("conv3_2", 256); c3("conv3_3", 256); x = b.pool("pool3", x, 2, 2);
    c3("conv4_1", 512); c3("conv4_2", 512); c3("conv4_3", 512);
    const int conv4_3 = x;
    x = b.pool("pool4", x, 2, 2);
    c3("conv5_1", 512); c3("conv5_2", 512); c3("conv5_3", 512); x = b.pool("pool5", x, 3, 1);
    x = b.conv("conv6", "conv6", x, 1024, 3, 1, 1, 6, "", true, SSD_ACT_RELU);
    x = b.conv("conv7", "conv7", x, 1024, 1, 1, 1, 1, "", true, SSD_ACT_RELU);
    const int conv7 = x;
    x = b.conv("conv8_1", "conv8_1", x, 256, 1, 1, 0, 1, "", true, SSD_ACT_RELU);
    x = b.conv("conv8_2", "conv8_2", x, 512, 3, 2, 1, 1, "", true, SSD_ACT_RELU);
    const int conv8_2 = x;
    x = b.conv("conv9_1", "conv9_1", x, 128, 1, 1, 0, 1, "", true, SSD_ACT_RELU);
    x = b.conv("conv9_2", "conv9_2", x, 256, 3, 2, 1, 1, "", true, SSD_ACT_RELU);
    const int conv9_2 = x;
    x = b.conv("conv10_1", "conv10_1", x, 128, 1, 1, 0, 1, "", true, SSD_ACT_RELU);
    x = b.conv("conv10_2", "conv10_2", x, 256, 3, 1, 0, 1, "", true, SSD_ACT_RELU);
    const int conv10_2 = x;
    x = b.conv("conv11_1", "conv11_1", x, 128, 1, 1, 0, 1, "", true, SSD_ACT_RELU);
    x = b.conv("conv11_2", "conv11_2", x, 256, 3, 1, 0, 1, "", true, SSD_ACT_RELU);
    const int conv11_2 = x;
    // L2Normalization(20)(conv4_3)  (models/ssd_vgg16.py:94)
    Layer l;
    l.name = "l2_normalization";
    l.kind = LK_L2NORM;
    l.in = conv4_3;
    const Tensor t = net.tensors[conv4_3];
    l.H = l.Ho = t.H; l.W = l.Wo = t.W; l.Cin = l.Cout = t.C;
    l.p_gamma = b.add_param("l2_normalization/scale", {t.C});
    l.out = b.add_tensor("l2_normalization", t.H, t.W, t.C);
    net.layers.push_back(l);
    b.heads({l.out, conv7, conv8_2, conv9_2, conv10_2, conv11_2});
}

static ConvParams layer_conv_params(const ssd_net& net, const Layer& l, int B, const float* in, float* out,
                                    const float* res, float* deltas_out, float* probs_out) {
    ConvParams p{};
    p.in = in;
    p.w = l.packed;
    p.w3 = conv_split_planes(l.packed, l.kh * l.kw * l.Cin, l.Cout);
    p.bf16 = net.precision;
    if (l.in >= 0 && net.tensors[l.in].planes) {       // the LDS-DMA tiles read the input's bf16 planes
        const Tensor& ti = net.tensors[l.in];
        p.xp = ti.planes; p.xp_plane = ti.plane_stride; p.xp_np = ti.planes_np;
    }
    if (l.out >= 0 && l.head_kind == 0 && net.tensors[l.out].planes_live && (l.Cout & 3) == 0) {
        const Tensor& to = net.tensors[l.out];      // ... and a consumer of this layer's output runs on them: the epilogue writes them
        p.op = to.planes; p.op_plane = to.plane_stride; p.op_np = to.planes_np;
    }
    p.wino_w = l.wino;
    p.scale = l.scale;
    p.shift = l.shift;
    p.residual = res;
    p.B = B; p.H = l.H; p.W = l.W; p.Cin = l.Cin; p.Ho = l.Ho; p.Wo = l.Wo; p.Cout = l.Cout;
    p.kh = l.kh; p.kw = l.kw; p.stride = l.stride; p.dil = l.dil; p.pad_t = l.pt; p.pad_l = l.pl;
    p.K = l.kh * l.kw * l.Cin;
    p.Kpad = conv_kpad(p.K);
    p.Npad = conv_npad(l.Cout);
    p.M = (long)B * l.Ho * l.Wo;
    p.act = l.act;
    p.split_k = l.split_k;
    p.partial = l.splitk_part ? l.splitk_part : net.splitk_ws;
    if (l.head_kind == 0) {
        p.out = out;
        p.out_pixel_stride = l.Cout;
        p.out_batch_stride = (long)l.Ho * l.Wo * l.Cout;
    } else {
        p.out = probs_out + l.head_off;
        p.out_pixel_stride = l.head_ps;
        p.out_batch_stride = l.head_bs;
        p.n_split = l.Cout1;
        p.out2 = deltas_out + l.head2_off;
        p.out2_pixel_stride = l.head2_ps;
        p.out2_batch_stride = l.head2_bs;
        p.vec_store2 = (p.out2_pixel_stride % 4 == 0) && (p.out2_batch_stride % 4 == 0);
    }
    p.vec_store = (((uintptr_t)p.out & 15) == 0) && (p.out_pixel_stride % 4 == 0) && (p.out_batch_stride % 4 == 0);
    return p;
}

static FusedBlockParams fused_params(const ssd_net& net, const Layer& f, int B) {
    const Layer& le = net.layers[f.f_expand];
    const Layer& ld = net.layers[f.f_dw];
    const Layer& lp = net.layers[f.f_project];
    FusedBlockParams p{};
    p.x = net.tensors[f.in].dev;
    p.y = net.tensors[f.out].dev;
    // scales are folded into the fz_* weight copies; the kernel only adds the shifts
    p.we = f.fz_we; p.es = le.scale; p.eh = le.shift;
    p.wd = f.fz_wd; p.ds = ld.scale; p.dh = ld.shift;
    p.wp = f.fz_wp; p.ps = lp.scale; p.ph = lp.shift;
    p.residual = lp.res >= 0 ? 1 : 0;
    p.B = B; p.H = f.H; p.W = f.W; p.Cin = f.Cin; p.Ce = le.Cout; p.Cout = f.Cout;
    p.Ho = f.Ho; p.Wo = f.Wo; p.stride = f.stride; p.pad_t = f.pt; p.pad_l = f.pl;
    p.kpad_e = conv_kpad(le.Cin);
    p.kpad_p = conv_kpad(lp.Cin);
    p.npad_p = conv_npad(lp.Cout);
    p.e_out = f.e_out >= 0 ? net.tensors[f.e_out].dev : nullptr;
    if (f.out >= 0 && net.tensors[f.out].planes_live) {           // (honoured by the whole-image kernel; the band kernels take a split pass)
        const Tensor& to = net.tensors[f.out];
        p.y_planes = to.planes; p.y_plane = to.plane_stride; p.planes_np = to.planes_np;
    }
    if (f.e_out >= 0 && net.tensors[f.e_out].planes_live) {
        const Tensor& te = net.tensors[f.e_out];
        p.e_planes = te.planes; p.e_plane = te.plane_stride; p.planes_np = te.planes_np;
    }
    p.we3 = reinterpret_cast<const short*>(f.fz_we3);        // band kernel's plane layout (blocks 1-6) ...
    p.wp3 = reinterpret_cast<const short*>(f.fz_wp3);
    p.bf16 = net.precision;
    if (!fused_block_supported(p) && image_block_supported(p)) {
        p.we3 = conv_split_planes(f.fz_we, le.Cin, le.Cout);  // ... whole-image kernel: plain [4][Ce][kpad_e] / [4][npad_p][kpad_p] planes
        p.wp3 = conv_split_planes(f.fz_wp, lp.Cin, lp.Cout);
        p.groups = net.img_slabs ? image_block_groups(p, B * net.lanes_hint) : 1;
        p.slabs = net.img_slabs;
        // (the in-launch ticketed combine exists for the fp32-MFMA form only: a no-op for the bf16 / split-bf16 forms)
        p.tickets = net.image_ticket && !p.bf16 ? net.img_tickets : nullptr;
        if (!p.bf16 && f.img_choice == 2) { p.bf16 = 3; p.tickets = nullptr; }    // the split-bf16 form of the image kernel (fp32 results)
    }
    p.form2 = net.image_v2 ? 1 : 0;          // second forms: whole-image kernel (ssd_imgblock2.hip), LDS-DMA weight staging of the split row-band kernel
    return p;
}

static DwProjParams dwproj_params(const ssd_net& net, const Layer& f, int B) {
    const Layer& ld = net.layers[f.f_dw];
    const Layer& lp = net.layers[f.f_project];
    DwProjParams p{};
    p.e = net.tensors[f.in].dev;
    p.y = net.tensors[f.out].dev;
    p.wd = f.fz_wd; p.dh = ld.shift;
    p.wp = f.fz_wp; p.ph = lp.shift;
    p.res = lp.res >= 0 ? net.tensors[lp.res].dev : nullptr;
    p.B = B; p.H = f.H; p.W = f.W; p.Ce = ld.Cout; p.Cout = f.Cout;
    p.Ho = f.Ho; p.Wo = f.Wo; p.stride = f.stride; p.pad_t = f.pt; p.pad_l = f.pl;
    p.kpad_p = conv_kpad(lp.Cin);
    p.npad_p = conv_npad(lp.Cout);
    p.bf16 = net.precision;
    return p;
}

static StemParams stem_params(const ssd_net& net, const Layer& f, int B) {
    const Layer& l1 = net.layers[f.f_expand];
    const Layer& ld = net.layers[f.f_dw];
    const Layer& lp = net.layers[f.f_project];
    StemParams p{};
    p.x = net.tensors[f.in].dev;
    p.y = net.tensors[f.out].dev;
    p.w1 = l1.packed; p.s1 = l1.scale; p.h1 = l1.shift;
    p.wd = net.params[ld.p_kernel].dev; p.sd = ld.scale; p.hd = ld.shift;
    p.wp = lp.packed; p.sp = lp.scale; p.hp = lp.shift;
    p.B = B; p.H = f.H; p.W = f.W; p.H1 = f.Ho; p.W1 = f.Wo; p.pad_t = f.pt; p.pad_l = f.pl;
    p.kpad1 = conv_kpad(27);
    p.kpadp = conv_kpad(32);
    p.bf16 = net.precision;
    return p;
}

static int run_layer_impl(ssd_net& net, const Layer& l, int B, float* deltas_out, float* probs_out, hipStream_t st,
                          int cfg_override);

static int run_layer(ssd_net& net, const Layer& l, int B, float* deltas_out, float* probs_out, hipStream_t st,
                     int cfg_override = -1) {
    static const bool dbg_sync = getenv("SSD_HIP_DEBUG_SYNC") != nullptr;
    if (!dbg_sync) return run_layer_impl(net, l, B, deltas_out, probs_out, st, cfg_override);
    fprintf(stderr, "[ssd dbg] run %s\n", l.name.c_str());
    const int rc = run_layer_impl(net, l, B, deltas_out, probs_out, st, cfg_override);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (!rc && cs == hipStreamCaptureStatusNone) SSD_HIP(hipStreamSynchronize(st));
    return rc;
}

// fp32 activation -> its bf16 planes (producers without a plane epilogue)
static int planes_pass(ssd_net& net, int tensor, int B, hipStream_t st) {
    Tensor& t = net.tensors[tensor];
    return launch_split_planes(t.dev, (long)B * (long)t.per_image, t.C, t.planes_np, t.planes, t.plane_stride, st);
}

static int run_layer_kernels(ssd_net& net, const Layer& l, int B, float* deltas_out, float* probs_out, hipStream_t st,
                             int cfg_override);

static int run_layer_impl(ssd_net& net, const Layer& l, int B, float* deltas_out, float* probs_out, hipStream_t st,
                          int cfg_override) {
    int rc = run_layer_kernels(net, l, B, deltas_out, probs_out, st, cfg_override);
    // convs, pools, the L2 normalisation and the whole-image block kernel write their planes themselves
    if (rc || l.kind == LK_CONV || l.kind == LK_POOL || l.kind == LK_L2NORM || l.kind == LK_SOFTMAX) return rc;
    const bool image_form = l.kind == LK_FUSED && l.f_type == 0 && !net.image_ticket && !fused_block_supported(fused_params(net, l, B));
    if (image_form) return rc;
    if (l.out >= 0 && net.tensors[l.out].planes_live) rc = planes_pass(net, l.out, B, st);
    if (!rc && l.kind == LK_FUSED && l.e_out >= 0 && net.tensors[l.e_out].planes_live) rc = planes_pass(net, l.e_out, B, st);
    return rc;
}

static int run_layer_kernels(ssd_net& net, const Layer& l, int B, float* deltas_out, float* probs_out, hipStream_t st,
                             int cfg_override) {
    const float* in = l.in >= 0 ? net.tensors[l.in].dev : nullptr;
    float* out = l.out >= 0 ? net.tensors[l.out].dev : nullptr;
    switch (l.kind) {
        case LK_CONV: {
            const float* res = l.res >= 0 ? net.tensors[l.res].dev : nullptr;
            ConvParams p = layer_conv_params(net, l, B, in, out, res, deltas_out, probs_out);
            const int cfg = cfg_override >= 0 ? cfg_override : l.cfg;
            const int rc = conv_launch(p, cfg, st);
            // families without the shared epilogue (Winograd, skinny, VALU direct): the planes come from a pass of their own
            if (!rc && p.op && !conv_config_writes_planes(cfg, p)) return planes_pass(net, l.out, B, st);
            return rc;
        }
        case LK_DW:
            return launch_dwconv3x3(in, B, l.H, l.W, l.Cin, l.stride, l.pt, l.pl, l.Ho, l.Wo,
                                    net.params[l.p_kernel].dev, l.scale, l.shift, l.act, out, st);
        case LK_POOL: {
            const Tensor& to = net.tensors[l.out];
            return launch_maxpool(in, B, l.H, l.W, l.Cin, l.kh, l.stride, l.pt, l.pl, l.Ho, l.Wo, out, st,
                                  to.planes_live ? to.planes : nullptr, to.plane_stride, to.planes_np);
        }
        case LK_L2NORM: {
            const Tensor& to = net.tensors[l.out];
            return launch_l2norm(in, (long)B * l.H * l.W, l.Cin, net.params[l.p_gamma].dev, out, st,
                                 to.planes_live ? to.planes : nullptr, to.plane_stride, to.planes_np);
        }
        case LK_SOFTMAX:
            return launch_softmax(probs_out, (long)B * net.num_priors, net.L, probs_out, st);
        case LK_FUSED:
            if (l.f_type == 1) return launch_stem(stem_params(net, l, B), st);
            if (l.f_type == 2) return launch_dwproj(dwproj_params(net, l, B), st);
            {
                const FusedBlockParams p = fused_params(net, l, B);
                if (fused_block_supported(p))       // blocks 1-6: row-band kernel where it applies, else the 8x8-tile kernel
{
                    if (net.fuse_band == 2 && p.we3 && band3_block_supported(p)) return launch_band3_block(p, st);
                    return net.fuse_band && band_block_supported(p) ? launch_band_block(p, st) : launch_fused_block(p, st);
                }
                return launch_image_block(p, st);
            }
    }
    return SSD_OK;
}

// A fused layer is active when fusion is on and the kernel supports its shape; then its
// three member layers are skipped (and vice versa).
static bool fused_active(const ssd_net& net, const Layer& f) {
    if (!net.fuse_blocks) return false;
    if (f.f_type == 1) return stem_supported(stem_params(net, f, 1));
    if (f.f_type == 2) {
        if (!net.fuse_dwproj) return false;
        if (f.fused_by >= 0 && fused_active(net, net.layers[f.fused_by])) return false;   // whole block fused
        return dwproj_supported(dwproj_params(net, f, 1));
    }
    const FusedBlockParams p = fused_params(net, f, 1);
    if (fused_block_supported(p)) return true;
    if (!net.fuse_image || (net.fuse_image == 1 && f.img_choice < 1)) return false;
    return image_block_supported(p);
}
static bool layer_runs(const ssd_net& net, const Layer& l) {
    if (l.kind == LK_FUSED) return fused_active(net, l);
    if (l.fused_by >= 0 && fused_active(net, net.layers[l.fused_by])) return false;
    if (l.fused_by2 >= 0 && fused_active(net, net.layers[l.fused_by2])) return false;
    return true;
}

static int dev_alloc(ssd_net& net, size_t floats, float** out) {
    *out = nullptr;
    if (floats == 0) return SSD_OK;
    SSD_HIP(hipMalloc((void**)out, floats * sizeof(float)));
    net.owned.push_back(*out);
    return SSD_OK;
}

// Time each valid (tile configuration, split-K factor) of every conv layer on the device and
// keep the best.  Split-K is tried only where the plain grid cannot fill the 256 CUs.
// Scope guards: the SSD_HIP early returns of the tuning / profiling entry points must not leak
struct ScopedDev {
    float* p = nullptr;
    ~ScopedDev() { if (p) (void)hipFree(p); }
};
struct ScopedEvent {
    hipEvent_t e = nullptr;
    ~ScopedEvent() { if (e) (void)hipEventDestroy(e); }
};

// Whole-image block kernel vs the layer kernels (expand GEMM + depthwise/project kernel) of the
// same block, timed on the device at batch B: at B = 64 the 4-way channel-group reduction costs
// what the E round trip through HBM costs, at large batches (one group) the image kernel wins.
static int tune_image_blocks(ssd_net& net, int B, hipStream_t st) {
    ScopedEvent se0, se1;
    SSD_HIP(hipEventCreate(&se0.e));
    SSD_HIP(hipEventCreate(&se1.e));
    struct ModeGuard {          // the race runs in "tuned choice" mode; early error returns restore the caller's mode too
        ssd_net& n;
        int mode;
        ~ModeGuard() { n.fuse_image = mode; }
    } guard{net, net.fuse_image};
    net.fuse_image = 1;
    int rc = SSD_OK;
    for (size_t fi = 0; fi < net.layers.size() && !rc; ++fi) {
        Layer& f = net.layers[fi];
        if (f.kind != LK_FUSED || f.f_type != 0) continue;
        const FusedBlockParams p = fused_params(net, f, B);
        if (fused_block_supported(p) || !image_block_supported(p)) { f.img_choice = 0; continue; }
        if (f.img_choice >= 0) continue;        // preset line
        float ms[3] = {1e30f, 1e30f, 1e30f};
        const int nchoice = net.precision == 0 && net.image_split ? 3 : 2;     // 2: the fp32 net's split-bf16 form
        for (int choice = 0; choice < nchoice && !rc; ++choice) {
            f.img_choice = choice;
            if (choice == 2 && !image_block_split_fits(fused_params(net, f, B))) continue;
            for (int trial = 0; trial < 4 && !rc; ++trial) {        // trial 0 warms up
                (void)hipEventRecord(se0.e, st);
                for (int r = 0; r < 4 && !rc; ++r)
                    for (int j = (int)fi; j <= f.f_project && !rc; ++j)
                        if (layer_runs(net, net.layers[j])) rc = run_layer(net, net.layers[j], B, nullptr, nullptr, st);
                (void)hipEventRecord(se1.e, st);
                if (rc) break;
                SSD_HIP(hipEventSynchronize(se1.e));
                float t = 0.f;
                (void)hipEventElapsedTime(&t, se0.e, se1.e);
                if (trial && t < ms[choice]) ms[choice] = t;
            }
        }
        f.img_choice = ms[2] < ms[1] && ms[2] < ms[0] ? 2 : ms[1] < ms[0] ? 1 : 0;
    }
    return rc;
}

static int autotune(ssd_net& net, int B, hipStream_t st) {
    ScopedEvent se0, se1;
    SSD_HIP(hipEventCreate(&se0.e));
    SSD_HIP(hipEventCreate(&se1.e));
    hipEvent_t e0 = se0.e, e1 = se1.e;
    ScopedDev sd, spr;                     // scratch outputs for the head convs
    SSD_HIP(hipMalloc((void**)&sd.p, (size_t)B * net.num_priors * 4 * sizeof(float)));
    SSD_HIP(hipMalloc((void**)&spr.p, (size_t)B * net.num_priors * net.L * sizeof(float)));
    float *d = sd.p, *pr = spr.p;
    // dense around small factors: the best split is the one whose M-blocks x split just fills a
    // whole number of CU rounds (e.g. 100 M-blocks x 5 = 500 blocks on 512 slots for head 2)
    static const int kSplits[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32};
    constexpr int kNumSplits = sizeof(kSplits) / sizeof(kSplits[0]);
    // split-K workspace: the largest [split][M][Cout] slab any candidate may need
    size_t ws_floats = 0;
    for (auto& l : net.layers) {
        if (l.kind != LK_CONV) continue;
        const size_t mc = (size_t)B * l.Ho * l.Wo * l.Cout;
        if (mc <= (size_t)4 << 20) ws_floats = std::max(ws_floats, mc * 32);
    }
    if (ws_floats > net.splitk_floats) {
        if (net.splitk_ws) (void)hipFree(net.splitk_ws);
        net.splitk_ws = nullptr;
        SSD_HIP(hipMalloc((void**)&net.splitk_ws, ws_floats * sizeof(float)));
        net.splitk_floats = ws_floats;
    }
    int rc = SSD_OK;
    const bool dbg_sync = getenv("SSD_HIP_DEBUG_SYNC") != nullptr;   // name the launch a GPU fault belongs to
    for (auto& l : net.layers) {
        if (l.kind != LK_CONV) continue;
        if (l.cfg >= 0) continue;           // chosen by a valid preset line (ssd_net_set_tuning)
        const float* in = net.tensors[l.in].dev;
        float* out = l.out >= 0 ? net.tensors[l.out].dev : nullptr;
        const float* res = l.res >= 0 ? net.tensors[l.res].dev : nullptr;
        float best = 1e30f;
        int best_cfg = -1, best_split = 1;
        for (int c = 0; c < conv_num_configs() && !rc; ++c) {
            for (int si = 0; si < kNumSplits && !rc; ++si) {
                const int split = kSplits[si];
                l.split_k = split;
                ConvParams p = layer_conv_params(net, l, B, in, out, res, d, pr);
                if (!conv_config_valid(c, p) || !conv_config_allowed(c, net.precision)) break;
                const bool direct = (c == conv_num_configs() - 1);
                if (direct && (best_cfg >= 0 || split > 1)) break;     // direct only as a last resort
                if (split > 1) {
                    const long blocks = conv_grid_blocks(c, p);
                    const int nkt = conv_k_tiles(c, p);
                    if (blocks > 384 || nkt < 4 * split) break;
                    if ((size_t)split * p.M * p.Cout > net.splitk_floats) break;
                }
                // min over 3 trials of 4 back-to-back launches: robust against clock ramps / noise
                const int reps = 4;
                if (dbg_sync) fprintf(stderr, "[ssd dbg] tune %s %s split %d\n", l.name.c_str(), conv_config_name(c), split);
                rc = conv_launch(p, c, st);     // warm-up
                if (rc) break;
                if (dbg_sync) SSD_HIP(hipStreamSynchronize(st));
                float ms = 1e30f;
                for (int trial = 0; trial < 3 && !rc; ++trial) {
                    (void)hipEventRecord(e0, st);
                    for (int r = 0; r < reps && !rc; ++r) rc = conv_launch(p, c, st);
                    (void)hipEventRecord(e1, st);
                    if (rc) break;
                    SSD_HIP(hipEventSynchronize(e1));
                    float t = 0.f;
                    (void)hipEventElapsedTime(&t, e0, e1);
                    ms = t < ms ? t : ms;
                    if (trial == 0 && ms > 1.5f * best) break;      // clearly slower than the incumbent
                }
                if (rc) break;
                // an LDS-DMA tile makes the producer of its input write the bf16 planes as well (6 or 2 bytes per element
                // at the store rate the epilogues reach): charged to the candidate, the race stays a per-layer one
                if (conv_config_is_dma(c))
                    ms += reps * (float)((double)B * net.tensors[l.in].per_image * 2.0 * net.tensors[l.in].planes_np / 3.0e12 * 1e3);
                if (ms < best) { best = ms; best_cfg = c; best_split = split; }
            }
        }
        if (rc) break;
        if (best_cfg < 0) {
            set_error("finalize: no conv kernel can run layer %s", l.name.c_str());
            rc = SSD_E_UNSUPPORTED;
            break;
        }
        l.cfg = best_cfg;
        l.split_k = best_split;
    }
    return rc;
}

}  // namespace ssd

// Capture `body` (a sequence of launches on st) into a hipGraph the first time a key is seen
// and replay it afterwards: one host call per step instead of ~70 launches.
template <typename F>
static int run_graphed(ssd_net* net, std::vector<const void*> key, hipStream_t& st, F body) {
    if (!net->use_graph || net->timing) return body();
    // fewer than four hardware queues: capturing the FORKED step (head convs / tail on side streams) segfaults inside the
    // runtime (tests/micro/graph_queues_net.py: 2 queues always, 3 queues on a native stream); the single in-order stream
    // of a lane captures and replays fine at any queue count
    if (net->graphs_unsafe && net->overlap_heads) return body();
    const hipStream_t caller = st;
    if (st == nullptr) {
        if (!net->gstream && hipStreamCreateWithFlags(&net->gstream, hipStreamDefault) != hipSuccess) {
            (void)hipGetLastError();
            return body();
        }
        st = net->gstream;
    }
    key.push_back((const void*)st);
    for (auto& g : net->graphs)
        if (g.key == key) {
            SSD_HIP(hipGraphLaunch(g.exec, st));
            // the replay bypasses forward_impl: keep the debug/parity hooks' view of the last call
            net->tensors[0].dev = const_cast<float*>(g.image);
            net->last_batch = g.batch;
            return SSD_OK;
        }
    if (net->graphs.size() >= 16) net->drop_graphs();
    // first sight of this key: run eagerly once on the caller's stream (validates arguments,
    // sets kernel attributes), then capture the same launches
    st = caller;
    int rc = body();
    if (rc) return rc;
    st = caller ? caller : net->gstream;
    ssd_net::GraphEntry e;
    e.key = std::move(key);
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return SSD_OK;               // capture unavailable: stay eager
    }
    rc = body();
    const hipError_t ce = hipStreamEndCapture(st, &e.graph);
    if (rc || ce != hipSuccess || !e.graph) {
        (void)hipGetLastError();
        if (e.graph) (void)hipGraphDestroy(e.graph);
        net->use_graph = false;      // fall back to eager launches for this net
        return rc ? rc : SSD_OK;
    }
    if (hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(e.graph);
        net->use_graph = false;
        return SSD_OK;
    }
    e.image = net->tensors[0].dev;
    e.batch = net->last_batch;
    net->graphs.push_back(e);
    return SSD_OK;
}

extern "C" {

ssd_net* ssd_net_create(int backbone, int img_size, int levels, const int* n_ars, int total_labels) {
    if ((backbone != SSD_MOBILENET_V2 && backbone != SSD_VGG16) || img_size < 32 || levels != 6 || !n_ars ||
        total_labels < 1) {
        set_error("ssd_net_create: bad arguments (backbone=%d img_size=%d levels=%d labels=%d)", backbone, img_size,
                  levels, total_labels);
        return nullptr;
    }
    auto net = std::make_unique<ssd_net>();
    net->backbone = backbone;
    net->img_size = img_size;
    net->levels = levels;
    net->L = total_labels;
    net->n_ars.assign(n_ars, n_ars + levels);
    for (int a : net->n_ars)
        if (a < 0 || a > 15) {
            set_error("ssd_net_create: bad aspect-ratio count %d", a);
            return nullptr;
        }
    if (backbone == SSD_MOBILENET_V2) build_mobilenet_v2(*net);
    else build_vgg16(*net);
    // hipGraph capture / replay of a step (forks onto the side streams) segfaults inside the HIP runtime when it is
    // limited to fewer than four hardware queues (GPU_MAX_HW_QUEUES=2, the two-lane serving setup: measured on
    // ROCm 7.2): such processes launch directly, which is also what the launch-mode race picks at B >= 64
    // (SSD_HIP_GRAPH_FORCE=1: diagnostics -- tests/micro/graph_queues_net.py looks for what exactly crashes)
    const bool graph_force = getenv("SSD_HIP_GRAPH_FORCE") && atoi(getenv("SSD_HIP_GRAPH_FORCE")) != 0;
    if (const char* q = getenv("GPU_MAX_HW_QUEUES"))
        if (atoi(q) > 0 && atoi(q) < 4 && !graph_force) {
            net->use_graph = false;
            net->use_graph_auto = false;
            net->graphs_unsafe = true;
        }
    if (const char* g = getenv("SSD_TAIL_PRIO")) net->tail_prio = atoi(g) < 0 ? 0 : (atoi(g) > 2 ? 2 : atoi(g));    // diagnostics
    if (const char* g = getenv("SSD_FUSE_SOFTMAX")) net->fuse_softmax = atoi(g) != 0;    // diagnostics (A/B of the decoder tail)
    if (const char* g = getenv("SSD_IMAGE_SPLIT")) net->image_split = atoi(g) != 0;      // diagnostics (A/B of the image kernel's forms)
    if (const char* g = getenv("SSD_HIP_CONV_DMA")) net->conv_dma = atoi(g) != 0;         // diagnostics (A/B of the LDS-DMA conv tiles)
    if (const char* g = getenv("SSD_HIP_USE_GRAPH")) {      // diagnostics: pin the launch mode (0 direct, 1 graph replay)
        net->use_graph = atoi(g) != 0 && !net->graphs_unsafe;
        net->use_graph_auto = false;
    }
    return net.release();
}

void ssd_net_destroy(ssd_net* net) { delete net; }

int ssd_net_num_params(const ssd_net* net) { return net ? (int)net->params.size() : 0; }
const char* ssd_net_param_name(const ssd_net* net, int i) {
    return (net && i >= 0 && i < (int)net->params.size()) ? net->params[i].name.c_str() : "";
}
int ssd_net_param_rank(const ssd_net* net, int i) {
    return (net && i >= 0 && i < (int)net->params.size()) ? (int)net->params[i].shape.size() : 0;
}
const int* ssd_net_param_shape(const ssd_net* net, int i) {
    return (net && i >= 0 && i < (int)net->params.size()) ? net->params[i].shape.data() : nullptr;
}

int ssd_net_set_param(ssd_net* net, const char* name, const float* host_data, size_t count) {
    SSD_CHECK_ARG(net && name && host_data, "ssd_net_set_param: NULL argument");
    auto it = net->param_index.find(name);
    SSD_CHECK_ARG(it != net->param_index.end(), "ssd_net_set_param: unknown parameter '%s'", name);
    Param& p = net->params[it->second];
    SSD_CHECK_ARG(count == p.count, "ssd_net_set_param: '%s' expects %zu values, got %zu", name, p.count, count);
    if (!p.dev) SSD_HIP(hipMalloc((void**)&p.dev, p.count * sizeof(float)));
    SSD_HIP(hipMemcpy(p.dev, host_data, p.count * sizeof(float), hipMemcpyHostToDevice));
    p.set = true;
    net->finalized = false;
    net->drop_graphs();
    return SSD_OK;
}

int ssd_net_get_param(const ssd_net* net, const char* name, float* host_out, size_t count) {
    SSD_CHECK_ARG(net && name && host_out, "ssd_net_get_param: NULL argument");
    auto it = net->param_index.find(name);
    SSD_CHECK_ARG(it != net->param_index.end(), "ssd_net_get_param: unknown parameter '%s'", name);
    const Param& p = net->params[it->second];
    SSD_CHECK_ARG(count == p.count, "ssd_net_get_param: '%s' holds %zu values, asked %zu", name, p.count, count);
    if (!p.set) {
        set_error("ssd_net_get_param: '%s' was never set", name);
        return SSD_E_STATE;
    }
    SSD_HIP(hipMemcpy(host_out, p.dev, p.count * sizeof(float), hipMemcpyDeviceToHost));
    return SSD_OK;
}

// hipGraph replay vs direct launches of the whole forward, raced on the device at max_batch on a zero
// image (measured at B=64: direct launches 1.950 ms, replay 1.972 ms per step on a non-default stream).
// Skipped once "use_graph" was set explicitly.
static int tune_launch_mode(ssd_net* net, int B) {
    if (!net->use_graph_auto || net->timing) return SSD_OK;
    {
        auto it = net->preset.find("__launch");          // "__launch graph 0|1": the recorded outcome of this race
        if (it != net->preset.end() && it->second.first == "graph") {
            net->use_graph = it->second.second != 0;
            net->launch_raced = true;
            return SSD_OK;
        }
    }
    ++net->n_autotuned;
    const size_t n_img = (size_t)B * net->img_size * net->img_size * 3;
    ScopedDev img, del, prb;
    SSD_HIP(hipMalloc((void**)&img.p, n_img * sizeof(float)));
    SSD_HIP(hipMalloc((void**)&del.p, (size_t)B * net->num_priors * 4 * sizeof(float)));
    SSD_HIP(hipMalloc((void**)&prb.p, (size_t)B * net->num_priors * net->L * sizeof(float)));
    SSD_HIP(hipMemset(img.p, 0, n_img * sizeof(float)));
    hipStream_t st = nullptr;
    SSD_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ScopedEvent e0, e1;
    int rc = SSD_OK;
    float ms[2] = {1e30f, 1e30f};
    if (hipEventCreate(&e0.e) != hipSuccess || hipEventCreate(&e1.e) != hipSuccess) rc = SSD_E_HIP;
    for (int mode = 0; mode < 2 && !rc; ++mode) {        // warm both modes: eager run, capture, replay
        net->use_graph = mode != 0;
        for (int w = 0; w < 3 && !rc; ++w) rc = ssd_net_forward(net, img.p, B, del.p, prb.p, st);
    }
    for (int trial = 0; trial < 4 && !rc; ++trial)       // interleaved trials (clock drift cancels), best of 4 each
        for (int mode = 0; mode < 2 && !rc; ++mode) {
            net->use_graph = mode != 0;
            (void)hipEventRecord(e0.e, st);
            for (int r = 0; r < 6 && !rc; ++r) rc = ssd_net_forward(net, img.p, B, del.p, prb.p, st);
            (void)hipEventRecord(e1.e, st);
            if (rc || hipEventSynchronize(e1.e) != hipSuccess) { rc = rc ? rc : SSD_E_HIP; break; }
            float t = 0.f;
            (void)hipEventElapsedTime(&t, e0.e, e1.e);
            if (t < ms[mode]) ms[mode] = t;
        }
    (void)hipStreamSynchronize(st);
    net->drop_graphs();
    (void)hipStreamDestroy(st);
    net->tensors[0].dev = nullptr;
    // the two are within ~1 % at B=64 and direct launches measured faster on real batches (heads overlap the
    // backbone earlier): replay only where it is clearly ahead (small batches, launch-bound hosts)
    net->use_graph = rc ? true : ms[1] < ms[0] * 0.99f;
    net->launch_raced = !rc;
    if (getenv("SSD_HIP_DEBUG_TUNE"))
        fprintf(stderr, "[ssd] launch mode race at B=%d: direct %.4f ms, graph replay %.4f ms per forward -> %s\n", B, ms[0] / 6,
                ms[1] / 6, net->use_graph ? "replay" : "direct");
    return rc;
}

int ssd_net_finalize(ssd_net* net, int max_batch) {
    SSD_CHECK_ARG(net && max_batch >= 1, "ssd_net_finalize: bad arguments");
    for (const auto& p : net->params)
        if (!p.set) {
            set_error("ssd_net_finalize: parameter '%s' was never set", p.name.c_str());
            return SSD_E_STATE;
        }
    hipStream_t st = nullptr;
    // (re)build derived weights
    for (float* p : net->owned)
        if (p) (void)hipFree(p);
    net->owned.clear();
    if (net->splitk_layers) (void)hipFree(net->splitk_layers);
    net->splitk_layers = nullptr;
    for (auto& l : net->layers) {
        l.packed = l.scale = l.shift = nullptr;
        l.wino = nullptr;
        l.splitk_part = nullptr;        // autotune runs on the shared slab; per-layer slabs are re-planned below
        if (l.kind == LK_CONV) {
            const int K = l.kh * l.kw * l.Cin;
            int rc = dev_alloc(*net, conv_packed_floats(K, l.Cout), &l.packed);
            if (rc) return rc;
            if (l.p_kernel2 < 0) {
                rc = launch_pack_weights(net->params[l.p_kernel].dev, K, l.Cout, conv_kpad(K), conv_npad(l.Cout),
                                         l.packed, st);
            } else {    // fused head: rows [0, Cout1) = label kernel, [Cout1, Cout) = box kernel
                SSD_HIP(hipMemsetAsync(l.packed, 0, (size_t)conv_kpad(K) * conv_npad(l.Cout) * sizeof(float), st));
                rc = launch_pack_weights(net->params[l.p_kernel].dev, K, l.Cout1, conv_kpad(K), l.Cout1, l.packed, st);
                if (!rc)
                    rc = launch_pack_weights(net->params[l.p_kernel2].dev, K, l.Cout - l.Cout1, conv_kpad(K),
                                             l.Cout - l.Cout1, l.packed + (size_t)l.Cout1 * conv_kpad(K), st);
            }
            if (!rc) rc = launch_pack_split(l.packed, K, l.Cout, st);
            if (rc) return rc;
            // Winograd form of the 3x3 stride-1 convs (heads, VGG16 backbone): the autotune decides
            if (l.kh == 3 && l.kw == 3 && l.stride == 1 && l.dil == 1 && l.Cin % 16 == 0 && l.res < 0 && net->use_wino) {
                const int np = conv_npad(l.Cout);
                rc = dev_alloc(*net, wino_weight_floats(l.Cin, l.Cout), &l.wino);
                if (rc) return rc;
                SSD_HIP(hipMemsetAsync(l.wino, 0, wino_weight_floats(l.Cin, l.Cout) * sizeof(float), st));
                if (l.p_kernel2 < 0) {
                    rc = launch_wino_pack(net->params[l.p_kernel].dev, l.Cin, l.Cout, np, 0, l.wino, st);
                } else {
                    rc = launch_wino_pack(net->params[l.p_kernel].dev, l.Cin, l.Cout1, np, 0, l.wino, st);
                    if (!rc) rc = launch_wino_pack(net->params[l.p_kernel2].dev, l.Cin, l.Cout - l.Cout1, np, l.Cout1, l.wino, st);
                }
                if (rc) return rc;
            }
        }
        if (l.kind == LK_CONV || l.kind == LK_DW) {
            if (l.p_bn >= 0) {
                int rc = dev_alloc(*net, l.Cout, &l.scale);
                if (!rc) rc = dev_alloc(*net, l.Cout, &l.shift);
                if (rc) return rc;
                rc = launch_fold_bn(net->params[l.p_bn].dev, net->params[l.p_bn + 1].dev, net->params[l.p_bn + 2].dev,
                                    net->params[l.p_bn + 3].dev, 1e-3f, l.Cout, l.scale, l.shift, st);
                if (rc) return rc;
            } else if (l.p_bias2 >= 0) {
                int rc = dev_alloc(*net, l.Cout, &l.shift);
                if (rc) return rc;
                SSD_HIP(hipMemcpyAsync(l.shift, net->params[l.p_bias].dev, (size_t)l.Cout1 * sizeof(float),
                                       hipMemcpyDeviceToDevice, st));
                SSD_HIP(hipMemcpyAsync(l.shift + l.Cout1, net->params[l.p_bias2].dev,
                                       (size_t)(l.Cout - l.Cout1) * sizeof(float), hipMemcpyDeviceToDevice, st));
            } else if (l.p_bias >= 0) {
                l.shift = net->params[l.p_bias].dev;     // bias-only epilogue: scale == 1
            }
        }
    }
    // fused blocks: weight copies with the BatchNorm scale folded in
    for (auto& f : net->layers) {
        if (f.kind == LK_FUSED && f.f_type == 2) {
            const Layer& ld = net->layers[f.f_dw];
            const Layer& lp = net->layers[f.f_project];
            const size_t nd = (size_t)9 * ld.Cout, np = (size_t)conv_kpad(lp.Cin) * conv_npad(lp.Cout);
            int rc = dev_alloc(*net, nd, &f.fz_wd);
            if (!rc) rc = dev_alloc(*net, np, &f.fz_wp);
            if (rc) return rc;
            rc = launch_scale_cols(net->params[ld.p_kernel].dev, ld.scale, 9, ld.Cout, f.fz_wd, st);
            if (!rc) rc = launch_scale_rows(lp.packed, lp.scale, conv_npad(lp.Cout), lp.Cout, conv_kpad(lp.Cin), f.fz_wp, st);
            if (rc) return rc;
            continue;
        }
        if (f.kind != LK_FUSED || f.f_type != 0) continue;
        const Layer& le = net->layers[f.f_expand];
        const Layer& ld = net->layers[f.f_dw];
        const Layer& lp = net->layers[f.f_project];
        const size_t nd = (size_t)9 * ld.Cout;
        // (the two 1x1 matrices with room for their four bf16 planes behind them, like every packed conv weight)
        int rc = dev_alloc(*net, conv_packed_floats(le.Cin, le.Cout), &f.fz_we);
        if (!rc) rc = dev_alloc(*net, nd, &f.fz_wd);
        if (!rc) rc = dev_alloc(*net, conv_packed_floats(lp.Cin, lp.Cout), &f.fz_wp);
        if (rc) return rc;
        // packed [n][kpad]: row n scaled by scale[n]; depthwise [9][C]: column c scaled by scale[c]
        rc = launch_scale_rows(le.packed, le.scale, conv_npad(le.Cout), le.Cout, conv_kpad(le.Cin), f.fz_we, st);
        if (!rc) rc = launch_scale_cols(net->params[ld.p_kernel].dev, ld.scale, 9, ld.Cout, f.fz_wd, st);
        if (!rc) rc = launch_scale_rows(lp.packed, lp.scale, conv_npad(lp.Cout), lp.Cout, conv_kpad(lp.Cin), f.fz_wp, st);
        // bf16 planes (h, m, l exact split + r rounding) of both: the bf16 form of the whole-image kernel reads plane r
        if (!rc) rc = launch_pack_split(f.fz_we, le.Cin, le.Cout, st, true);
        if (!rc) rc = launch_pack_split(f.fz_wp, lp.Cin, lp.Cout, st, true);
        if (rc) return rc;
        // split-bf16 band kernel: the same two matrices as three bf16 planes each (exact split, see ssd_band3.hip)
        f.fz_we3 = f.fz_wp3 = nullptr;
        if (band3_block_supported(fused_params(*net, f, 1))) {
            rc = dev_alloc(*net, (band3_we_shorts(le.Cout) + 1) / 2, &f.fz_we3);
            if (!rc) rc = dev_alloc(*net, (band3_wp_shorts(conv_npad(lp.Cout), le.Cout) + 1) / 2, &f.fz_wp3);
            if (!rc) rc = launch_band3_pack(f.fz_we, le.Cout, le.Cin, conv_kpad(le.Cin), reinterpret_cast<short*>(f.fz_we3), f.fz_wp,
                                            conv_npad(lp.Cout), conv_kpad(lp.Cin), reinterpret_cast<short*>(f.fz_wp3), st);
            if (rc) return rc;
        }
    }
    // whole-image block kernel: slab workspace for the largest (groups x batch) product any batch
    // up to max_batch can ask for, and the arrival tickets (zero between launches)
    {
        size_t slab = 0;
        for (auto& f : net->layers) {
            if (f.kind != LK_FUSED || f.f_type != 0) continue;
            const FusedBlockParams p = fused_params(*net, f, 1);
            if (fused_block_supported(p) || !image_block_supported(p)) continue;
            for (int b = 1; b <= max_batch; ++b) slab = std::max(slab, image_block_slab_floats(p, b));
        }
        net->img_slabs = nullptr;
        net->img_tickets = nullptr;
        net->slab_bytes = slab * sizeof(float);
        if (slab) {
            float* tk = nullptr;
            int rc = dev_alloc(*net, slab, &net->img_slabs);
            if (!rc) rc = dev_alloc(*net, (size_t)max_batch, &tk);
            if (rc) return rc;
            net->img_tickets = reinterpret_cast<unsigned*>(tk);
            SSD_HIP(hipMemsetAsync(tk, 0, (size_t)max_batch * sizeof(unsigned), st));
        }
    }
    // activation arena: one slot per tensor (288 GB of HBM: no aliasing needed, every
    // activation of the last forward stays inspectable)
    if (net->arena) (void)hipFree(net->arena);
    net->arena = nullptr;
    // SSD_HIP_DEBUG_POISON: NaN-filled arena with a NaN gap around every activation, so a
    // kernel that reads outside its input (or relies on zero-initialised memory) fails parity
    const bool poison = getenv("SSD_HIP_DEBUG_POISON") != nullptr;
    const size_t gap = poison ? 1024 : 0;
    size_t total = gap;
    for (auto& t : net->tensors) total += align_up(t.per_image * max_batch, 64) + gap;
    SSD_HIP(hipMalloc((void**)&net->arena, total * sizeof(float)));
    size_t off = gap;
    for (auto& t : net->tensors) {
        t.dev = net->arena + off;
        off += align_up(t.per_image * max_batch, 64) + gap;
    }
    SSD_HIP(hipMemset(net->arena, poison ? 0xFF : 0, total * sizeof(float)));
    // bf16 planes of every activation a dense conv with Cin % 32 == 0 reads (the LDS-DMA tiles' operand, ssd_convdma.hip);
    // written only while a consumer's chosen configuration asks for them (planes_live)
    for (auto& t : net->tensors) { t.planes = nullptr; t.planes_live = false; t.planes_np = 0; t.plane_stride = 0; }
    if (net->conv_dma)
        for (const auto& l : net->layers) {
            if (l.kind != LK_CONV || l.in < 0 || l.Cin % 32 != 0) continue;
            Tensor& t = net->tensors[l.in];
            if (t.planes) continue;
            t.planes_np = net->precision ? 1 : 3;
            t.plane_stride = (long)align_up(t.per_image * max_batch, 64);
            float* pl = nullptr;
            const int rcp = dev_alloc(*net, (size_t)t.planes_np * t.plane_stride / 2 + 64, &pl);
            if (rcp) return rcp;
            t.planes = reinterpret_cast<short*>(pl);
            SSD_HIP(hipMemsetAsync(pl, 0, ((size_t)t.planes_np * t.plane_stride / 2 + 64) * sizeof(float), st));
        }
    net->max_batch = max_batch;
    // pick tile configurations on the device
    for (auto& l : net->layers) { l.cfg = -1; l.split_k = 1; }
    // preset lines (ssd_net_set_tuning) replace the on-device autotune LAYER BY LAYER: a layer whose line
    // is missing or names a configuration that cannot run it here is timed on the device, the others are
    // not -- a complete table means no timing at all and therefore the same kernels (and the same bits)
    // in every process that loads it
    size_t ws_need = 0;
    int n_tuned = 0, n_preset = 0;
    for (auto& l : net->layers) {
        if (l.kind != LK_CONV) continue;
        auto it = net->preset.find(l.name);
        if (it == net->preset.end()) { ++n_tuned; continue; }
        int cfg = -1;
        for (int c = 0; c < conv_num_configs(); ++c)
            if (it->second.first == conv_config_name(c)) cfg = c;
        if (cfg < 0 || !conv_config_allowed(cfg, net->precision)) { ++n_tuned; continue; }
        l.cfg = cfg;
        l.split_k = it->second.second;
        ConvParams p = layer_conv_params(*net, l, max_batch, net->tensors[l.in].dev,
                                         l.out >= 0 ? net->tensors[l.out].dev : nullptr, nullptr,
                                         net->arena, net->arena);
        const bool split_ok = l.split_k == 1 || conv_k_tiles(cfg, p) >= l.split_k;
        if (l.split_k < 1 || !split_ok || !conv_config_valid(cfg, p)) { l.cfg = -1; l.split_k = 1; ++n_tuned; continue; }
        ++n_preset;
        if (l.split_k > 1) ws_need = std::max(ws_need, (size_t)l.split_k * p.M * p.Cout);
    }
    bool image_preset_ok = true;
    for (auto& f : net->layers) {       // whole-block layers the image kernel can run: "name image 0|1"
        if (f.kind != LK_FUSED || f.f_type != 0) continue;
        f.img_choice = -1;
        const FusedBlockParams p = fused_params(*net, f, 1);
        if (fused_block_supported(p) || !image_block_supported(p)) continue;
        auto it = net->preset.find(f.name);
        if (it == net->preset.end() || it->second.first != "image") { image_preset_ok = false; continue; }
        f.img_choice = it->second.second < 0 ? 0 : it->second.second > 2 ? 2 : it->second.second;
        if (f.img_choice == 2 && (net->precision != 0 || !net->image_split)) f.img_choice = 1;
    }
    int rc = SSD_OK;
    if (ws_need > net->splitk_floats) {
        if (net->splitk_ws) (void)hipFree(net->splitk_ws);
        net->splitk_ws = nullptr;
        SSD_HIP(hipMalloc((void**)&net->splitk_ws, ws_need * sizeof(float)));
        net->splitk_floats = ws_need;
    }
    net->n_autotuned = n_tuned;
    net->n_preset = n_preset;
    if (n_tuned) rc = autotune(*net, max_batch, st);
    if (rc) return rc;
    // the planes were allocated for the race; keep them only where a CHOSEN tile reads them (6 bytes per element in fp32 nets,
    // 2 in the bf16 mode, on top of the fp32 slot: a table without LDS-DMA tiles leaves none)
    {
        std::vector<char> read(net->tensors.size(), 0);
        for (const auto& l : net->layers)
            if (l.kind == LK_CONV && l.in >= 0 && conv_config_is_dma(l.cfg)) read[l.in] = 1;
        SSD_HIP(hipStreamSynchronize(st));
        net->plane_bytes = 0;
        net->arena_bytes = total * sizeof(float);
        for (size_t i = 0; i < net->tensors.size(); ++i) {
            Tensor& t = net->tensors[i];
            if (!t.planes) continue;
            const size_t bytes = ((size_t)t.planes_np * t.plane_stride / 2 + 64) * sizeof(float);
            if (read[i]) { net->plane_bytes += bytes; continue; }
            float* pl = reinterpret_cast<float*>(t.planes);
            net->owned.erase(std::remove(net->owned.begin(), net->owned.end(), pl), net->owned.end());
            (void)hipFree(pl);
            t.planes = nullptr; t.planes_np = 0; t.plane_stride = 0;
        }
    }
    // every split-K layer gets its own slab (layers on different streams may overlap)
    {
        if (net->splitk_layers) (void)hipFree(net->splitk_layers);
        net->splitk_layers = nullptr;
        size_t total_ws = 0;
        for (auto& l : net->layers) {
            l.splitk_part = nullptr;
            if (l.kind == LK_CONV && l.split_k > 1)
                total_ws += align_up((size_t)l.split_k * max_batch * l.Ho * l.Wo * l.Cout, 64);
        }
        if (total_ws) {
            SSD_HIP(hipMalloc((void**)&net->splitk_layers, total_ws * sizeof(float)));
            size_t o = 0;
            for (auto& l : net->layers)
                if (l.kind == LK_CONV && l.split_k > 1) {
                    l.splitk_part = net->splitk_layers + o;
                    o += align_up((size_t)l.split_k * max_batch * l.Ho * l.Wo * l.Cout, 64);
                }
        }
        for (int k = 0; k < ssd_net::kSides; ++k) {
            if (!net->side[k]) {
                if (k == 2) {       // the tail's stream: highest priority the device offers
                    int lo = 0, hi = 0;
                    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                    SSD_HIP(hipStreamCreateWithPriority(&net->side[k], hipStreamNonBlocking, hi));
                } else {
                    SSD_HIP(hipStreamCreateWithFlags(&net->side[k], hipStreamNonBlocking));
                }
            }
            if (!net->ev_side_done[k]) SSD_HIP(hipEventCreateWithFlags(&net->ev_side_done[k], hipEventDisableTiming));
        }
        for (auto& l : net->layers)
            if (!l.ev_ready) SSD_HIP(hipEventCreateWithFlags(&l.ev_ready, hipEventDisableTiming));
    }
    if (!image_preset_ok) {
        rc = tune_image_blocks(*net, max_batch, st);
        if (rc) return rc;
        ++net->n_autotuned;
    }
    SSD_HIP(hipDeviceSynchronize());
    net->drop_graphs();
    net->finalized = true;
    return tune_launch_mode(net, max_batch);
}

// Tuning table as text, one "layer config split_k" line per conv layer.
long ssd_net_get_tuning(const ssd_net* net, char* buf, size_t cap) {
    if (!net) return SSD_E_INVALID;
    std::string out;
    for (const auto& l : net->layers)
        if (l.kind == LK_CONV && l.cfg >= 0)
            out += l.name + " " + conv_config_name(l.cfg) + " " + std::to_string(l.split_k) + "\n";
    for (const auto& l : net->layers)
        if (l.kind == LK_FUSED && l.f_type == 0 && l.img_choice >= 0) {
            const FusedBlockParams p = fused_params(*net, l, 1);
            if (!fused_block_supported(p) && image_block_supported(p))
                out += l.name + " image " + std::to_string(l.img_choice) + "\n";
        }
    if (net->finalized && net->use_graph_auto && net->launch_raced)
        out += std::string("__launch graph ") + (net->use_graph ? "1" : "0") + "\n";
    if (buf && cap > out.size()) memcpy(buf, out.c_str(), out.size() + 1);
    return (long)out.size();
}

// device memory the finalized net holds, in bytes: [0] activation arena, [1] bf16 planes of the activations its chosen LDS-DMA
// tiles read, [2] whole-image kernel slabs, [3] split-K slabs
int ssd_net_memory_bytes(const ssd_net* net, size_t out[4]) {
    SSD_CHECK_ARG(net != nullptr && out != nullptr, "ssd_net_memory_bytes: NULL argument");
    SSD_CHECK_ARG(net->finalized, "ssd_net_memory_bytes: net is not finalized");
    out[0] = net->arena_bytes;
    out[1] = net->plane_bytes;
    out[2] = net->slab_bytes;
    size_t sk = net->splitk_floats;
    for (const auto& l : net->layers)
        if (l.kind == LK_CONV && l.split_k > 1) sk += align_up((size_t)l.split_k * net->max_batch * l.Ho * l.Wo * l.Cout, 64);
    out[3] = sk * sizeof(float);
    return SSD_OK;
}

int ssd_net_tuning_stats(const ssd_net* net, int* from_table, int* timed) {
    SSD_CHECK_ARG(net != nullptr, "ssd_net_tuning_stats: net is NULL");
    if (from_table) *from_table = net->n_preset;
    if (timed) *timed = net->n_autotuned;
    return SSD_OK;
}

int ssd_net_set_tuning(ssd_net* net, const char* text) {
    SSD_CHECK_ARG(net && text, "ssd_net_set_tuning: NULL argument");
    net->preset.clear();
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t e = s.find('\n', pos);
        if (e == std::string::npos) e = s.size();
        const std::string line = s.substr(pos, e - pos);
        pos = e + 1;
        char name[128], cfg[128];
        int split = 1;
        if (line.empty() || line[0] == '#') continue;
        const bool flag = sscanf(line.c_str(), "%127s %127s %d", name, cfg, &split) == 3 &&
                          (std::string(cfg) == "image" || std::string(cfg) == "graph");
        if (sscanf(line.c_str(), "%127s %127s %d", name, cfg, &split) == 3 && split >= (flag ? 0 : 1) && split <= 64)
            net->preset[name] = {cfg, split};
    }
    net->finalized = false;
    return SSD_OK;
}

int ssd_net_num_priors(const ssd_net* net) { return net ? net->num_priors : 0; }
int ssd_net_feature_map_size(const ssd_net* net, int level) {
    return (net && level >= 0 && level < (int)net->fmap.size()) ? net->fmap[level] : 0;
}

// logits_only: the softmax layer is left out (ssd_net_predict: the decoder's compaction kernel applies it on its
// LDS-staged slab); probs_out then holds the head convs' logits
static int forward_impl(ssd_net* net, const float* image_dev, int B, float* deltas_out, float* probs_out,
                        hipStream_t st, bool logits_only = false) {
    SSD_CHECK_ARG(net != nullptr, "ssd_net_forward: net is NULL");
    if (!net->finalized) {
        set_error("ssd_net_forward: call ssd_net_finalize() first");
        return SSD_E_STATE;
    }
    SSD_CHECK_ARG(B >= 0 && B <= net->max_batch, "ssd_net_forward: batch %d exceeds max_batch %d", B, net->max_batch);
    if (B == 0) return SSD_OK;
    // which activations have to exist as bf16 planes for this forward: the inputs of the running dense convs whose chosen
    // configuration is an LDS-DMA tile (their producers' epilogues -- or a split pass -- write them)
    for (auto& t : net->tensors) t.planes_live = false;
    for (const auto& l : net->layers)
        if (l.kind == LK_CONV && l.in >= 0 && conv_config_is_dma(l.cfg) && net->tensors[l.in].planes && layer_runs(*net, l))
            net->tensors[l.in].planes_live = true;
    SSD_CHECK_ARG(image_dev && deltas_out && probs_out, "ssd_net_forward: NULL pointer");
    SSD_CHECK_ARG(((uintptr_t)deltas_out & 15) == 0 && ((uintptr_t)probs_out & 15) == 0,
                  "ssd_net_forward: outputs must be 16-byte aligned");
    // the input tensor aliases the caller's image buffer (no copy)
    net->tensors[0].dev = const_cast<float*>(image_dev);
    std::vector<hipEvent_t>* ev = nullptr;
    if (net->timing) {
        net->timing_events.emplace_back(net->layers.size() + 2);
        ev = &net->timing_events.back();
        for (auto& e : *ev) SSD_HIP(hipEventCreate(&e));
        (void)hipEventRecord((*ev)[0], st);
    }
    const bool overlap = net->overlap_heads && !net->timing && net->side[0];
    // Stream plan (matters under hipGraph replay too: ready nodes start in capture order).  Every
    // kernel of the heavy part of the graph fills the whole GPU, so overlapping two of them only splits
    // the machine (measured: head 1 beside block 13's depthwise+project -> 239 + 212 us instead of
    // 187 + 37).  What CAN hide is the latency-bound tail behind feature map 3 (extras 2-4, heads 3-6:
    // ~150 us of 5-20 us kernels).  Shipped plan:
    //   main stream : backbone ... producer of feature map 3, then the extras chain, (join) softmax
    //   side[0]     : the two big head convs (side == 1), captured right behind feature map 3's producer
    //                 (launching their split-K reduces behind BOTH of them instead of in between: no change)
    //   side[1]     : the small heads (side == 2), each right behind its producer
    // SSD_TAIL_ON_SIDE=1 / option "tail_on_side" swaps the roles (big heads back to back on the main stream, the tail on the
    // side streams): measured 2.07 instead of 1.99 ms -- beside a head conv whose workgroups hold every
    // CU's LDS / VGPRs a 10 us tail kernel waits for retiring workgroups (120-150 us each), and the
    // dependent tail chain then outlives the heads.  In the shipped plan the executor starts head 1
    // ~80 us after the backbone, i.e. the first third of the tail runs alone and un-starved.
    const int nl = (int)net->layers.size();
    static const bool tail_env = getenv("SSD_TAIL_ON_SIDE") ? atoi(getenv("SSD_TAIL_ON_SIDE")) != 0 : false;
    const bool tail_on_side = tail_env || net->tail_on_side;
    std::vector<int> sid(nl, -1);           // -1 main, 0 / 1 side stream
    std::vector<size_t> order;
    order.reserve(nl);
    if (overlap) {
        int split = -1;                     // producer of the first small head's input
        for (int j = 0; j < nl && split < 0; ++j) {
            const Layer& h = net->layers[j];
            if (h.side != 2 || !layer_runs(*net, h)) continue;
            for (int q = j - 1; q >= 0; --q)
                if ((net->layers[q].out == h.in || net->layers[q].e_out == h.in) && layer_runs(*net, net->layers[q])) { split = q; break; }
        }
        std::vector<char> placed(nl, 0);
        auto place_small_heads = [&](int out_tensor) {
            for (int j = 0; j < nl; ++j) {
                const Layer& c = net->layers[j];
                if (placed[j] || c.side != 2 || c.in != out_tensor || !layer_runs(*net, c)) continue;
                placed[j] = 1;
                sid[j] = 1;
                order.push_back(j);
            }
        };
        for (int i = 0; i < nl; ++i) {      // main chain up to the split
            const Layer& l = net->layers[i];
            if (l.side || l.kind == LK_SOFTMAX || (split >= 0 && i > split)) continue;
            order.push_back(i);
            placed[i] = 1;
        }
        for (int i = 0; i < nl; ++i) {      // big heads: main stream, right behind the split layer
            if (net->layers[i].side != 1) continue;
            // (VGG16: the L2 normalisation feeding head 1 sits behind the extras in the layer list)
            for (int j = i - 1; j >= 0; --j)
                if ((net->layers[j].out == net->layers[i].in || net->layers[j].e_out == net->layers[i].in) && layer_runs(*net, net->layers[j])) {
                    if (!placed[j]) { order.push_back(j); placed[j] = 1; }
                    break;
                }
            order.push_back(i);
            placed[i] = 1;
            if (!tail_on_side) sid[i] = 0;
        }
        if (split >= 0) place_small_heads(net->layers[split].out);
        for (int i = 0; i < nl; ++i) {      // tail chain (+ its small heads right behind their producers)
            const Layer& l = net->layers[i];
            if (placed[i] || l.side || l.kind == LK_SOFTMAX) continue;
            sid[i] = tail_on_side ? 0 : (net->tail_prio ? 2 : -1);
            order.push_back(i);
            placed[i] = 1;
            if (l.out > 0 && layer_runs(*net, l)) {
                const size_t o0 = order.size();
                place_small_heads(l.out);
                if (net->tail_prio == 2)
                    for (size_t oo = o0; oo < order.size(); ++oo) sid[order[oo]] = 2;
            }
        }
        for (int i = 0; i < nl; ++i)        // anything left (small heads without a running producer), then softmax
            if (!placed[i] && net->layers[i].kind != LK_SOFTMAX) { sid[i] = net->layers[i].side == 2 ? 1 : -1; order.push_back(i); placed[i] = 1; }
        for (int i = 0; i < nl; ++i)
            if (!placed[i]) order.push_back(i);
    } else {
        for (int i = 0; i < nl; ++i) order.push_back(i);
    }
    auto stream_of = [&](int i) { return sid[i] < 0 ? st : net->side[sid[i]]; };
    auto producer_of = [&](int i, int tensor) {
        for (int j = i - 1; j >= 0; --j)
            if ((net->layers[j].out == tensor || net->layers[j].e_out == tensor) && layer_runs(*net, net->layers[j])) return j;      // (e_out: block 13's whole-image kernel also produces its expanded map)
        return -1;
    };
    // a layer publishes its completion (ev_ready on ITS stream) when a consumer runs on another stream
    std::vector<char> publishes(nl, 0);
    if (overlap)
        for (int i = 0; i < nl; ++i) {
            const Layer& l = net->layers[i];
            if (!layer_runs(*net, l)) continue;
            for (int tensor : {l.in, l.res}) {
                if (tensor < 0) continue;
                const int pr = producer_of(i, tensor);
                if (pr >= 0 && sid[pr] != sid[i]) publishes[pr] = 1;
            }
        }
    bool side_used[ssd_net::kSides] = {false, false, false};
    auto join_sides = [&]() -> int {
        for (int k = 0; k < ssd_net::kSides; ++k)
            if (side_used[k]) {
                SSD_HIP(hipEventRecord(net->ev_side_done[k], net->side[k]));
                SSD_HIP(hipStreamWaitEvent(st, net->ev_side_done[k], 0));
                side_used[k] = false;
            }
        return SSD_OK;
    };
    for (size_t oi = 0; oi < order.size(); ++oi) {
        const int i = (int)order[oi];
        Layer& l = net->layers[i];
        if (layer_runs(*net, l) && !(logits_only && l.kind == LK_SOFTMAX)) {
            hipStream_t ls = stream_of(i);
            if (overlap) {
                if (l.kind == LK_SOFTMAX) {        // join before the softmax
                    const int rcj = join_sides();
                    if (rcj) return rcj;
                }
                bool forked = false;
                for (int tensor : {l.in, l.res}) {
                    if (tensor < 0) continue;
                    const int pr = producer_of(i, tensor);
                    if (pr >= 0 && sid[pr] != sid[i]) { SSD_HIP(hipStreamWaitEvent(ls, net->layers[pr].ev_ready, 0)); forked = true; }
                }
                if (sid[i] >= 0 && !side_used[sid[i]] && !forked) {
                    // first work of a side stream whose input is not produced by a layer (the image):
                    // order it after the current main-stream work
                    SSD_HIP(hipEventRecord(l.ev_ready, st));
                    SSD_HIP(hipStreamWaitEvent(ls, l.ev_ready, 0));
                }
                if (sid[i] >= 0) side_used[sid[i]] = true;
            }
            const int rc = run_layer(*net, l, B, deltas_out, probs_out, ls);
            if (rc) return rc;
            if (overlap && publishes[i]) SSD_HIP(hipEventRecord(l.ev_ready, ls));
        }
        if (ev) (void)hipEventRecord((*ev)[i + 1], st);
    }
    {   // no softmax layer after the side work: still join
        const int rcj = join_sides();
        if (rcj) return rcj;
    }
    net->last_batch = B;
    return SSD_OK;
}

int ssd_net_forward(ssd_net* net, const float* image_dev, int B, float* deltas_out_dev, float* probs_out_dev,
                    void* stream) {
    SSD_CHECK_ARG(net != nullptr, "ssd_net_forward: net is NULL");
    if (!net->finalized) {
        set_error("ssd_net_forward: call ssd_net_finalize() first");
        return SSD_E_STATE;
    }
    SSD_CHECK_ARG(B >= 0 && B <= net->max_batch, "ssd_net_forward: batch %d exceeds max_batch %d", B, net->max_batch);
    hipStream_t st = (hipStream_t)stream;
    return run_graphed(net, {image_dev, deltas_out_dev, probs_out_dev, (const void*)(intptr_t)B, (const void*)1}, st,
                       [&]() { return forward_impl(net, image_dev, B, deltas_out_dev, probs_out_dev, st); });
}

int ssd_net_predict(ssd_net* net, const float* image_dev, int B, const float* priors_dev, const float* var,
                    int max_total, float iou_thr, float score_thr, float* boxes_dev, float* labels_dev,
                    float* scores_dev, int* valid_dev, void* stream) {
    SSD_CHECK_ARG(net != nullptr, "ssd_net_predict: net is NULL");
    if (!net->finalized) {
        set_error("ssd_net_predict: call ssd_net_finalize() first");
        return SSD_E_STATE;
    }
    const int N = net->num_priors, L = net->L;
    SSD_CHECK_ARG(B >= 0 && B <= net->max_batch, "ssd_net_predict: batch %d exceeds max_batch %d", B, net->max_batch);
    SSD_CHECK_ARG(max_total >= 1, "ssd_net_predict: max_total must be >= 1");
    // head-output scratch follows max_batch (a re-finalize with a larger batch regrows it); every
    // reallocation invalidates the captured graphs, which have the old pointers baked in
    if (net->scratch_batch < net->max_batch) {
        (void)hipDeviceSynchronize();
        net->drop_graphs();
        if (net->deltas) (void)hipFree(net->deltas);
        if (net->probs) (void)hipFree(net->probs);
        net->deltas = net->probs = nullptr;
        net->scratch_batch = 0;
        SSD_HIP(hipMalloc((void**)&net->deltas, (size_t)net->max_batch * N * 4 * sizeof(float)));
        SSD_HIP(hipMalloc((void**)&net->probs, (size_t)net->max_batch * N * L * sizeof(float)));
        net->scratch_batch = net->max_batch;
    }
    // the fused decoder carves the workspace for nms_ws_batch images whatever B a call runs: a re-finalize with another
    // max_batch, or a call with a larger max_total, re-carves -- and re-zeroes -- it (never a layout for one batch over
    // counters left by another)
    const int ws_batch = std::max(net->max_batch, net->nms_ws_batch);
    const size_t need = ssd_decode_nms_workspace_bytes(ws_batch, N, L, max_total);
    if (need > net->nms_ws_bytes || ws_batch != net->nms_ws_batch || max_total != net->nms_ws_total) {
        (void)hipDeviceSynchronize();
        net->drop_graphs();
        if (net->nms_ws) (void)hipFree(net->nms_ws);
        net->nms_ws = nullptr;
        net->nms_ws_bytes = 0;
        SSD_HIP(hipMalloc(&net->nms_ws, need));
        SSD_HIP(hipMemset(net->nms_ws, 0, need));      // candidate counters start at zero and are re-zeroed by every call (nms_class_kernel)
        net->nms_ws_bytes = need;
        net->nms_ws_batch = ws_batch;
        net->nms_ws_total = max_total;
    }
    hipStream_t st = (hipStream_t)stream;
    SSD_CHECK_ARG(var != nullptr, "ssd_net_predict: variances pointer is NULL");
    const float v4[4] = {var[0], var[1], var[2], var[3]};
    union { float f; intptr_t i; } k1{}, k2{};
    k1.f = iou_thr; k2.f = score_thr;
    std::vector<const void*> key = {image_dev, priors_dev, boxes_dev, labels_dev, scores_dev, valid_dev,
                                    (const void*)(intptr_t)B, (const void*)(intptr_t)max_total,
                                    (const void*)k1.i, (const void*)k2.i, (const void*)2};
    for (int i = 0; i < 4; ++i) { union { float f; intptr_t i; } k{}; k.f = v4[i]; key.push_back((const void*)k.i); }
    // softmax folded into the decoder's compaction (one pass over the [B, N, L] buffer and two launches -- softmax, counter
    // memset -- less); option "fuse_softmax" 0 keeps the layer-by-layer form
    const bool fused_sm = net->fuse_softmax && decode_nms_fused_ok(L) && N >= 1 && net->nms_ws_batch >= B;
    int rc = run_graphed(net, key, st, [&]() {
        int r = forward_impl(net, image_dev, B, net->deltas, net->probs, st, fused_sm);
        if (r) return r;
        if (fused_sm)
            return decode_nms_fused(net->deltas, net->probs, priors_dev, v4, B, N, L, max_total, max_total, iou_thr, score_thr,
                                    boxes_dev, labels_dev, scores_dev, valid_dev, net->nms_ws, net->nms_ws_bytes, net->nms_ws_batch, st);
        return ssd_decode_nms(net->deltas, net->probs, priors_dev, v4, B, N, L, max_total, max_total, iou_thr,
                              score_thr, boxes_dev, labels_dev, scores_dev, valid_dev, nullptr, net->nms_ws,
                              net->nms_ws_bytes, (void*)st);
    });
    if (rc && fused_sm && net->nms_ws) {
        // a step that failed between the compaction and the per-class kernel leaves candidate counters behind: the next
        // call must not read them as keys
        (void)hipDeviceSynchronize();
        (void)hipMemset(net->nms_ws, 0, net->nms_ws_bytes);
    }
    if (!rc && net->timing && !net->timing_events.empty())
        (void)hipEventRecord(net->timing_events.back().back(), st);
    return rc;
}

// Diagnostics: run one fused block layer with the kernel's per-phase clock64() counters on
// and return, per phase, the mean cycles per wave (6 phases: prologue, expand, depthwise,
// project, weight stage, epilogue).  Requires a previous forward at batch >= B.
int ssd_net_profile_fused(ssd_net* net, const char* layer, int B, double* cycles_out6) {
    SSD_CHECK_ARG(net && layer && cycles_out6 && B >= 1, "ssd_net_profile_fused: bad arguments");
    if (!net->finalized) { set_error("ssd_net_profile_fused: not finalized"); return SSD_E_STATE; }
    const Layer* f = nullptr;
    for (const auto& l : net->layers)
        if (l.kind == LK_FUSED && l.f_type == 0 && l.name == layer) f = &l;
    SSD_CHECK_ARG(f != nullptr, "ssd_net_profile_fused: unknown fused layer '%s'", layer);
    FusedBlockParams p = fused_params(*net, *f, B);
    const bool image = !fused_block_supported(p) && image_block_supported(p);
    SSD_CHECK_ARG(fused_block_supported(p) || image, "ssd_net_profile_fused: layer not supported by the fused kernels");
    const bool band = !image && net->fuse_band == 1 && band_block_supported(p);
    auto launch = [&](const FusedBlockParams& q) {
        return image ? launch_image_block(q, nullptr) : band ? launch_band_block(q, nullptr) : launch_fused_block(q, nullptr);
    };
    if (const char* ab = getenv("SSD_FUSED_ABLATE")) {      // diagnostics: time the kernel with phases removed
        p.ablate = atoi(ab);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)launch(p);
        (void)hipEventRecord(e0, nullptr);
        for (int r = 0; r < 10; ++r) (void)launch(p);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        for (int i = 0; i < 6; ++i) cycles_out6[i] = 0;
        cycles_out6[0] = ms * 100.0;        // microseconds per launch
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return SSD_OK;
    }
    const size_t n = (size_t)B * 4096 * 4 * 6;     // upper bound: <= 4096 tiles per image
    long long* d = nullptr;
    SSD_HIP(hipMalloc((void**)&d, n * sizeof(long long)));
    SSD_HIP(hipMemset(d, 0, n * sizeof(long long)));
    p.dbg = d;
    int rc = launch(p);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = SSD_E_HIP;
    if (!rc) {
        std::vector<long long> h(n);
        SSD_HIP(hipMemcpy(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost));
        double sum[6] = {0, 0, 0, 0, 0, 0};
        size_t waves = 0;
        for (size_t w = 0; w + 5 < n; w += 6) {
            if (h[w] == 0 && h[w + 5] == 0) continue;
            for (int i = 0; i < 6; ++i) sum[i] += (double)h[w + i];
            ++waves;
        }
        for (int i = 0; i < 6; ++i) cycles_out6[i] = waves ? sum[i] / waves : 0;
    }
    (void)hipFree(d);
    return rc;
}

int ssd_net_set_option(ssd_net* net, const char* name, int value) {
    SSD_CHECK_ARG(net && name, "ssd_net_set_option: NULL argument");
    if (std::string(name) == "fuse_blocks") {
        net->fuse_blocks = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "precision") {      // 0 fp32 (default), 1 bf16 matrix operands (see ssd_net.h); takes effect at the next finalize
        SSD_CHECK_ARG(value == 0 || value == 1, "ssd_net_set_option: precision must be 0 (fp32) or 1 (bf16)");
        if (net->precision != value) net->finalized = false;
        net->precision = value;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "fuse_band") {      // blocks 1-6: 2 (default) split-bf16 row-band kernel, 1 fp32-MFMA row-band kernel, 0 the 8x8-tile kernel
        net->fuse_band = value < 0 ? 0 : (value > 2 ? 2 : value);
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "fuse_image") {
        net->fuse_image = value < 0 ? 0 : (value > 2 ? 2 : value);
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "lanes_hint") {     // number of replicas in flight beside this one (models/decoder.py lanes)
        net->lanes_hint = value < 1 ? 1 : (value > 8 ? 8 : value);
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "tail_prio") {      // extras tail (2: + its small heads) on a highest-priority stream
        net->tail_prio = value < 0 ? 0 : (value > 2 ? 2 : value);
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "tail_on_side") {
        net->tail_on_side = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "image_split") {    // fp32 nets: let the split-bf16 form of the image kernel into the finalize-time race (default 1)
        if (net->image_split != (value != 0)) net->finalized = false;
        net->image_split = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "image_ticket") {
        net->image_ticket = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "fuse_softmax") {   // ssd_net_predict: softmax inside the decoder's compaction kernel (default 1)
        net->fuse_softmax = value != 0;
        net->drop_graphs();
        // the layer-by-layer decoder carves the workspace per call: back to all-zero counters for the fused form
        if (net->nms_ws) {
            (void)hipDeviceSynchronize();
            SSD_HIP(hipMemset(net->nms_ws, 0, net->nms_ws_bytes));
        }
        return SSD_OK;
    }
    if (std::string(name) == "fuse_dwproj") {
        net->fuse_dwproj = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "overlap_heads") {
        net->overlap_heads = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "use_wino") {       // Winograd F(2x2,3x3) candidates in the autotune (default 1)
        net->use_wino = value != 0;
        net->finalized = false;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "image_v2") {       // whole-image kernel: second form (default 1) / first form (0); bitwise equal
        net->image_v2 = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "conv_dma") {       // LDS-DMA conv tiles over pre-split activation planes (default 1)
        if (net->conv_dma != (value != 0)) net->finalized = false;
        net->conv_dma = value != 0;
        net->drop_graphs();
        return SSD_OK;
    }
    if (std::string(name) == "use_graph") {
        if (value && net->graphs_unsafe && net->overlap_heads) {
            set_error("ssd_net_set_option: use_graph 1 needs overlap_heads 0 with GPU_MAX_HW_QUEUES < 4 (hipGraph capture of the forked step crashes in the runtime; a single in-order stream replays fine)");
            return SSD_E_UNSUPPORTED;
        }
        net->use_graph = value != 0;
        net->use_graph_auto = false;
        net->drop_graphs();
        return SSD_OK;
    }
    set_error("ssd_net_set_option: unknown option '%s'", name);
    return SSD_E_INVALID;
}

int ssd_net_set_timing(ssd_net* net, int enabled) {
    SSD_CHECK_ARG(net != nullptr, "ssd_net_set_timing: net is NULL");
    net->timing = enabled != 0;
    return SSD_OK;
}

// Sum of per-layer durations (ms) over the forwards recorded since the last read;
// ms_sum_out has num_layers + 1 entries, the last one being decode+NMS (predict only).
int ssd_net_read_timing(ssd_net* net, float* ms_sum_out, int* forwards_out) {
    SSD_CHECK_ARG(net && ms_sum_out && forwards_out, "ssd_net_read_timing: NULL argument");
    const size_t nl = net->layers.size();
    for (size_t i = 0; i <= nl; ++i) ms_sum_out[i] = 0.f;
    SSD_HIP(hipDeviceSynchronize());
    for (auto& ev : net->timing_events) {
        for (size_t i = 0; i < nl; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) ms_sum_out[i] += ms;
        }
        float ms = 0.f;
        if (hipEventQuery(ev[nl + 1]) == hipSuccess && hipEventElapsedTime(&ms, ev[nl], ev[nl + 1]) == hipSuccess)
            ms_sum_out[nl] += ms;
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    *forwards_out = (int)net->timing_events.size();
    net->timing_events.clear();
    return SSD_OK;
}

long ssd_net_fetch_activation(ssd_net* net, const char* layer, float* host_out, size_t cap) {
    if (!net || !layer) return SSD_E_INVALID;
    auto it = net->tensor_index.find(layer);
    if (it == net->tensor_index.end()) {
        set_error("ssd_net_fetch_activation: unknown tensor '%s'", layer);
        return SSD_E_INVALID;
    }
    const Tensor& t = net->tensors[it->second];
    const size_t n = t.per_image * (size_t)net->last_batch;
    if (!host_out) return (long)n;
    if (cap < n) {
        set_error("ssd_net_fetch_activation: buffer holds %zu floats, need %zu", cap, n);
        return SSD_E_INVALID;
    }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, t.dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("ssd_net_fetch_activation: copy failed");
        return SSD_E_HIP;
    }
    return (long)n;
}

// ... and its bf16 planes (the LDS-DMA conv tiles' operand) joined back to fp32: 0 when no running consumer asked for
// them in the last forward; *planes_out = 3 (exact split) or 1 (bf16 rounding)
long ssd_net_fetch_planes(ssd_net* net, const char* layer, float* host_out, size_t cap, int* planes_out) {
    if (!net || !layer) return SSD_E_INVALID;
    auto it = net->tensor_index.find(layer);
    if (it == net->tensor_index.end()) {
        set_error("ssd_net_fetch_planes: unknown tensor '%s'", layer);
        return SSD_E_INVALID;
    }
    const Tensor& t = net->tensors[it->second];
    if (planes_out) *planes_out = t.planes_np;
    if (!t.planes || !t.planes_live) return 0;
    const size_t n = t.per_image * (size_t)net->last_batch;
    if (!host_out) return (long)n;
    if (cap < n) {
        set_error("ssd_net_fetch_planes: buffer holds %zu floats, need %zu", cap, n);
        return SSD_E_INVALID;
    }
    ScopedDev tmp;
    SSD_HIP(hipMalloc((void**)&tmp.p, n * sizeof(float)));
    SSD_HIP(hipDeviceSynchronize());
    const int rc = launch_join_planes(t.planes, (long)n, t.C, t.planes_np, t.plane_stride, tmp.p, nullptr);
    if (rc) return rc;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, tmp.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("ssd_net_fetch_planes: copy failed");
        return SSD_E_HIP;
    }
    return (long)n;
}

int ssd_net_num_layers(const ssd_net* net) { return net ? (int)net->layers.size() : 0; }
const char* ssd_net_layer_name(const ssd_net* net, int i) {
    return (net && i >= 0 && i < (int)net->layers.size()) ? net->layers[i].name.c_str() : "";
}
const char* ssd_net_layer_kind(const ssd_net* net, int i) {
    return (net && i >= 0 && i < (int)net->layers.size()) ? kKindName[net->layers[i].kind] : "";
}
const char* ssd_net_layer_config(const ssd_net* net, int i) {
    if (!net || i < 0 || i >= (int)net->layers.size()) return "";
    static thread_local char buf[64];
    const Layer& l = net->layers[i];
    if (l.kind == LK_FUSED) {        // which fused kernel family runs the block (bench.py prices each at its matrix instruction)
        if (!layer_runs(*net, l)) return "";
        if (l.f_type == 1) { const int f = stem_form(net->precision); return f == 1 ? "stem_bf16" : f == 3 ? "stem_split" : "stem"; }
        if (l.f_type == 2) return net->precision ? "dwproj_bf16" : "dwproj";
        const FusedBlockParams p = fused_params(*net, l, 1);
        if (!fused_block_supported(p)) return net->precision ? "image_bf16" : l.img_choice == 2 ? "image_split" : "image";
        if (net->fuse_band == 2 && p.we3 && band3_block_supported(p)) return net->precision ? "band_bf16" : "band3";
        return net->fuse_band && band_block_supported(p) ? "band" : "tile";
    }
    if (l.kind != LK_CONV) return "";
    if (l.split_k > 1) snprintf(buf, sizeof(buf), "%s/s%d", conv_config_name(l.cfg), l.split_k);
    else snprintf(buf, sizeof(buf), "%s", conv_config_name(l.cfg));
    return buf;
}
double ssd_net_layer_flops(const ssd_net* net, int i, int B) {
    if (!net || i < 0 || i >= (int)net->layers.size()) return 0;
    const Layer& l = net->layers[i];
    if (!layer_runs(*net, l)) return 0;
    if (l.kind == LK_FUSED) {
        double f = 0;
        const_cast<ssd_net*>(net)->fuse_blocks = false;     // member flops (algorithmic, no halo recompute)
        f = ssd_net_layer_flops(net, l.f_expand, B) + ssd_net_layer_flops(net, l.f_dw, B) +
            ssd_net_layer_flops(net, l.f_project, B);
        const_cast<ssd_net*>(net)->fuse_blocks = true;
        return f;
    }
    const double px = (double)B * l.Ho * l.Wo;
    if (l.kind == LK_CONV) return 2.0 * px * l.kh * l.kw * l.Cin * l.Cout;
    if (l.kind == LK_DW) return 2.0 * px * 9 * l.Cout;
    return 0;
}
// FLOPs the chosen kernel actually issues to the matrix cores: equal to the algorithmic figure
// except for Winograd F(2x2,3x3) layers (16 multiplies per 2x2 tile and channel pair instead of 36;
// border tiles are computed whole).
double ssd_net_layer_executed_flops(const ssd_net* net, int i, int B) {
    if (!net || i < 0 || i >= (int)net->layers.size()) return 0;
    const Layer& l = net->layers[i];
    if (!layer_runs(*net, l)) return 0;
    if (l.kind == LK_CONV && l.cfg >= conv_num_mfma_configs() && l.cfg < conv_num_mfma_configs() + wino_num_configs())
        return 2.0 * B * ((l.Ho + 1) / 2) * ((l.Wo + 1) / 2) * 16.0 * l.Cin * l.Cout;
    return ssd_net_layer_flops(net, i, B);
}
double ssd_net_layer_bytes(const ssd_net* net, int i, int B) {
    if (!net || i < 0 || i >= (int)net->layers.size()) return 0;
    const Layer& l = net->layers[i];
    if (!layer_runs(*net, l)) return 0;
    if (l.kind == LK_FUSED)
        return 4.0 * B * ((double)l.H * l.W * l.Cin + (double)l.Ho * l.Wo * l.Cout);
    if (l.kind == LK_SOFTMAX) return 2.0 * 4.0 * B * net->num_priors * net->L;
    double b = 4.0 * B * ((double)l.H * l.W * l.Cin + (double)l.Ho * l.Wo * l.Cout);
    if (l.res >= 0) b += 4.0 * B * (double)l.Ho * l.Wo * l.Cout;
    if (l.kind == LK_CONV) b += 4.0 * l.kh * l.kw * l.Cin * l.Cout;
    if (l.kind == LK_DW) b += 4.0 * 9 * l.Cout;
    return b;
}

int ssd_net_profile_layers(ssd_net* net, const float* image_dev, int B, int reps, float* ms_out, void* stream) {
    SSD_CHECK_ARG(net && image_dev && ms_out && reps >= 1, "ssd_net_profile_layers: bad arguments");
    if (!net->finalized) {
        set_error("ssd_net_profile_layers: call ssd_net_finalize() first");
        return SSD_E_STATE;
    }
    SSD_CHECK_ARG(B >= 1 && B <= net->max_batch, "ssd_net_profile_layers: bad batch %d", B);
    hipStream_t st = (hipStream_t)stream;
    const int N = net->num_priors, L = net->L;
    ScopedDev sd, spr;
    SSD_HIP(hipMalloc((void**)&sd.p, (size_t)B * N * 4 * sizeof(float)));
    SSD_HIP(hipMalloc((void**)&spr.p, (size_t)B * N * L * sizeof(float)));
    float *d = sd.p, *pr = spr.p;
    int rc = forward_impl(net, image_dev, B, d, pr, st);   // warm-up + valid inputs for each layer
    ScopedEvent se0, se1;
    SSD_HIP(hipEventCreate(&se0.e));
    SSD_HIP(hipEventCreate(&se1.e));
    hipEvent_t e0 = se0.e, e1 = se1.e;
    for (size_t i = 0; i < net->layers.size() && !rc; ++i) {
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < reps && !rc && layer_runs(*net, net->layers[i]); ++r)
            rc = run_layer(*net, net->layers[i], B, d, pr, st);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms_out[i] = ms / reps;
    }
    return rc;
}

}  // extern "C"

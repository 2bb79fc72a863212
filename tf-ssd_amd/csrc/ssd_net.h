// Shared definitions of the graph runner (ssd_net.hip) and the training step (ssd_train.hip):
// parameter table, activation tensors, layer list and the net object behind `ssd_net*`.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "ssd_conv.h"

struct ssd_train_state;      // csrc/ssd_train.hip
void ssd_train_state_free(ssd_train_state*);

namespace ssd {

enum LayerKind { LK_CONV = 0, LK_DW, LK_POOL, LK_L2NORM, LK_SOFTMAX, LK_FUSED };
static const char* kKindName[] = {"conv", "dw", "pool", "l2norm", "softmax", "fused"};

struct Param {
    std::string name;
    std::vector<int> shape;
    size_t count = 0;
    float* dev = nullptr;
    bool set = false;
    bool in_flat = false;     // dev points into the training step's flat parameter buffer (not freed per param)
};

struct Tensor {
    std::string name;
    int H = 0, W = 0, C = 0;
    size_t per_image = 0;     // floats
    float* dev = nullptr;     // arena slot, max_batch images
    // the same activation as bf16 planes [planes_np][plane_stride] for the LDS-DMA conv tiles of its consumers
    // (csrc/ssd_convdma.hip): allocated at finalize for every tensor a dense conv with Cin % 32 == 0 reads (option
    // "conv_dma"), WRITTEN -- by the producer's epilogue or by split_planes_kernel -- only while a running consumer's
    // chosen configuration is an LDS-DMA tile (planes_live, recomputed per forward)
    short* planes = nullptr;
    long plane_stride = 0;    // elements between planes
    int planes_np = 0;        // 3: exact split h, m, l (fp32 nets); 1: bf16 rounding (the bf16 mode)
    bool planes_live = false;
};

struct Layer {
    std::string name;
    LayerKind kind = LK_CONV;
    int in = -1, out = -1, res = -1;          // tensor ids (out == -1: head conv -> net outputs)
    int H = 0, W = 0, Cin = 0, Ho = 0, Wo = 0, Cout = 0;
    int kh = 1, kw = 1, stride = 1, dil = 1, pt = 0, pl = 0, pb = 0, pr = 0, act = 0;
    int p_kernel = -1, p_bias = -1, p_bn = -1;  // p_bn: index of gamma (beta, mean, var follow)
    int p_kernel2 = -1, p_bias2 = -1, Cout1 = 0; // fused head conv: second (boxes) kernel/bias, label width
    int p_gamma = -1;                           // l2norm scale
    // head routing (floats): level offset, batch stride, pixel stride for the label part
    // (head_*) and the box part (head2_*); head_kind 3 = fused label+box head conv of one level
    int head_kind = 0;
    long head_off = 0, head_bs = 0, head_ps = 0;
    long head2_off = 0, head2_bs = 0, head2_ps = 0;
    // derived at finalize
    float* packed = nullptr;
    float* wino = nullptr;          // Winograd F(2x2,3x3) weights (3x3 stride-1 convs), see ssd_wino.hip
    float* scale = nullptr;
    float* shift = nullptr;
    int cfg = -1;
    int split_k = 1;
    // fused inverted-residual block: LK_FUSED layer points at its three member layers;
    // members carry the index of their LK_FUSED layer in `fused_by`
    int f_expand = -1, f_dw = -1, f_project = -1;
    int f_type = 0;                 // 0: inverted-residual block, 1: stem (Conv1 -> dw -> project),
                                    // 2: depthwise + project (the expand conv stays a GEMM of its own)
    int fused_by = -1;
 int e_out = -1;                 // whole-block LK_FUSED layer that must ALSO materialise its expanded map: that tensor
    int img_choice = -1;            // whole-block LK_FUSED layers the image kernel can run: 1 / 2 = its fp32-MFMA / split-bf16 form won the finalize-time race against the layer kernels, 0 = it lost, -1 = not timed
    int fused_by2 = -1;             // depthwise / project members: their type-2 LK_FUSED layer
    float* splitk_part = nullptr;   // this layer's own split-K slab (layers may run concurrently)
    // LK_FUSED: weight copies with the folded BatchNorm scale multiplied in (per output channel)
    float *fz_we = nullptr, *fz_wd = nullptr, *fz_wp = nullptr;
    float *fz_we3 = nullptr, *fz_wp3 = nullptr;      // bf16 planes of fz_we / fz_wp for the split-bf16 band kernel (ssd_band3.hip)
    int side = 0;                   // 1, 2: runs on that side stream (SSD head convs)
    hipEvent_t ev_ready = nullptr;  // recorded on the main stream when this layer's OUTPUT is complete
};

}  // namespace ssd

struct ssd_net {
    int backbone = 0, img_size = 300, levels = 6, L = 21;
    std::vector<int> n_ars;
    std::vector<ssd::Param> params;
    std::map<std::string, int> param_index;
    std::vector<ssd::Tensor> tensors;
    std::map<std::string, int> tensor_index;
    std::vector<ssd::Layer> layers;
    std::vector<int> fmap;          // feature-map size per level
    std::vector<long> level_off;    // prior offset per level
    int num_priors = 0;
    bool finalized = false;
    bool fuse_blocks = true;        // run eligible inverted-residual blocks as one fused kernel
    bool fuse_softmax = true;       // ssd_net_predict: the softmax runs inside the decoder's compaction kernel (csrc/ssd_bbox.hip) instead of as a pass of its own
    bool fuse_dwproj = true;        // ... and depthwise + project of the others as one kernel
    bool use_wino = true;           // offer the Winograd F(2x2,3x3) kernels to the autotune
    int precision = 0;              // 0: fp32 results everywhere (the reference's arithmetic); 1: "bf16" -- every matrix operand of the dense / 1x1 convs rounded once to bf16, one bf16 MFMA per product, fp32 accumulation and epilogues (BASELINE.json configs[3] / [4])
    int fuse_band = 2;              // blocks 1-6: 2 row-band kernel with the 1x1 convs on the bf16 matrix cores through an exact 3-way split (ssd_band3.hip), 1 row-band kernel on the fp32 MFMA (ssd_bandblock.hip), 0 the 8x8-tile kernel
    int fuse_image = 1;             // whole-image block kernel (ssd_imgblock.hip): 0 never, 1 where it won the finalize-time race, 2 wherever it applies
    int lanes_hint = 1;             // replicas of this net running concurrently (lanes): the whole-image kernel then splits an image's expanded channels over fewer workgroups (B x groups x lanes fills the CUs; fewer slab passes)
    int tail_prio = 0;              // 1: extras tail on side[2] (highest priority); 2: its small heads too
    bool tail_on_side = false;      // diagnostics: big heads on the main stream, extras tail + small heads on the side streams
    bool image_split = true;        // fp32 nets: the finalize-time race also times the image kernel's split-bf16 form (img_choice 2; SSD_IMAGE_SPLIT=0 / option "image_split" 0: leave it out)
    bool image_v2 = true;           // whole-image kernel: the second form (ssd_imgblock2.hip: compile-time geometry, adjacent pixels per lane) where it has a configuration; 0 = the first form (A/B, bitwise equal)
    bool conv_dma = true;           // offer the LDS-DMA tiles over pre-split activation planes (ssd_convdma.hip) to the autotune / accept them from tables
    bool image_ticket = false;      // combine the channel-group slabs inside the launch (arrival ticket) instead of by a second launch
    float* img_slabs = nullptr;     // its partial-sum slabs and arrival tickets (sized for max_batch)
    unsigned* img_tickets = nullptr;
    int max_batch = 0;
    int last_batch = 0;
    std::vector<float*> owned;      // device allocations to free
    float* arena = nullptr;
    size_t arena_bytes = 0, plane_bytes = 0, slab_bytes = 0;    // ssd_net_memory_bytes
    float* splitk_ws = nullptr;
    size_t splitk_floats = 0;
    // predict() scratch
    float* deltas = nullptr;
    float* probs = nullptr;
    void* nms_ws = nullptr;
    size_t nms_ws_bytes = 0;
    int nms_ws_batch = 0;           // the batch the workspace was carved (and its candidate counters zeroed) for
    int nms_ws_total = 0;           // ... and the max_total it was carved for
    int scratch_batch = 0;          // batch capacity deltas/probs were allocated for
    // optional per-layer hipEvent timing of forward()/predict() (bench.py roofline leg)
    std::map<std::string, std::pair<std::string, int>> preset;   // layer -> (config name, split_k)
    int n_preset = 0;               // conv layers the last finalize took from preset lines
    int n_autotuned = 0;            // choices the last finalize timed on the device (0 = fully reproducible table)
    bool launch_raced = false;      // the graph-replay / direct-launch choice was made by a race or a preset line
    // hipGraph replay of a whole forward/predict step, keyed by every pointer baked into it
    bool use_graph = true;
    bool graphs_unsafe = false;     // GPU_MAX_HW_QUEUES < 4 in the environment: never capture / replay
    bool use_graph_auto = true;     // finalize races graph replay against direct launches (until "use_graph" is set explicitly)
    struct GraphEntry {
        std::vector<const void*> key;
        hipGraphExec_t exec = nullptr;
        hipGraph_t graph = nullptr;
        const float* image = nullptr;   // restored on replay (fetch_activation / profile hooks read them)
        int batch = 0;
    };
    std::vector<GraphEntry> graphs;
    // graphs cannot be captured on the legacy NULL stream (PyTorch's default stream): such
    // calls are captured/replayed on this BLOCKING stream, which the NULL stream implicitly
    // orders with (legacy default-stream semantics), so callers see the same ordering.
    hipStream_t gstream = nullptr;
    // the head convs only depend on their feature map: they run on `side` concurrently with the
    // rest of the backbone / extras (fork after the producer, join before the softmax)
    bool overlap_heads = true;
    static constexpr int kSides = 3;       // [2]: highest-priority stream for the latency-bound extras tail (option tail_prio)
    hipStream_t side[kSides] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_side_done[kSides] = {nullptr, nullptr, nullptr};
    float* splitk_layers = nullptr;     // per-layer split-K slabs (post-autotune)
    bool timing = false;
    ssd_train_state* train = nullptr;    // training step state (csrc/ssd_train.hip), lazily created
    std::vector<std::vector<hipEvent_t>> timing_events;   // one vector of (layers + 2) events per forward

    ~ssd_net() {
        for (auto& p : params)
            if (p.dev && !p.in_flat) (void)hipFree(p.dev);
        if (train) ssd_train_state_free(train);
        for (float* p : owned)
            if (p) (void)hipFree(p);
        if (arena) (void)hipFree(arena);
        if (splitk_ws) (void)hipFree(splitk_ws);
        if (deltas) (void)hipFree(deltas);
        if (probs) (void)hipFree(probs);
        if (nms_ws) (void)hipFree(nms_ws);
        for (auto& v : timing_events)
            for (auto e : v) (void)hipEventDestroy(e);
        drop_graphs();
        if (gstream) (void)hipStreamDestroy(gstream);
        for (int k = 0; k < kSides; ++k) {
            if (side[k]) (void)hipStreamDestroy(side[k]);
            if (ev_side_done[k]) (void)hipEventDestroy(ev_side_done[k]);
        }
        for (auto& l : layers)
            if (l.ev_ready) (void)hipEventDestroy(l.ev_ready);
        if (splitk_layers) (void)hipFree(splitk_layers);
    }
    void drop_graphs() {
        for (auto& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
        graphs.clear();
    }
};


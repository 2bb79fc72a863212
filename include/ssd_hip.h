/*
 * ssd_hip.h -- C ABI of libssd_hip.so: the MI355X (gfx950) replacement for the SSD
 * forward + decode/NMS hot path of FurkanOM/tf-ssd.
 *
 * The reference has no FFI boundary (it is Python on TensorFlow); the de-facto operator
 * API is its Python surface.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference root).  The Python host in
 * tf-ssd_amd/ binds these with ctypes (tf-ssd_amd/ssd_hip.py) and keeps the reference's
 * function names and signatures.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - Every `const float* / float* / int*` named *_dev or documented "device" is a device
 *     pointer owned by the caller (e.g. torch.Tensor.data_ptr()).  The library never
 *     frees caller memory.  Scratch comes from a caller-provided workspace whose size is
 *     reported by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All compute
 *     entry points are asynchronous on that stream.
 *   - Return value: 0 = ok, negative = error (SSD_E_*); text via ssd_last_error()
 *     (thread-local).  Nothing aborts or throws across the ABI.
 *   - Layouts: activations NHWC fp32; boxes [y1,x1,y2,x2] fp32; Keras weight layouts
 *     (Conv2D HWIO, DepthwiseConv2D [kh,kw,C,1]) at the set_param boundary.
 */
#ifndef SSD_HIP_H
#define SSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSD_OK 0
#define SSD_E_INVALID (-1)     /* bad argument (Python shim raises ValueError)          */
#define SSD_E_HIP (-2)         /* a HIP runtime call failed                              */
#define SSD_E_UNSUPPORTED (-3) /* valid in the reference but outside this build's limits */
#define SSD_E_STATE (-4)       /* object used in the wrong state (e.g. not finalized)    */

/* ---- library ------------------------------------------------------------------- */
const char* ssd_version(void);
const char* ssd_last_error(void);
/* Select + probe the device (must be gfx950).  Replaces utils/io_utils.py:54-61
 * (handle_gpu_compatibility) as the one-off per-process device hook. */
int ssd_init(int device);
/* A stream for keeping several batches in flight (DecoderModel lanes): created
 * hipStreamNonBlocking, so work on it is NOT ordered against the legacy NULL stream -- an event recorded on the
 * NULL stream (what torch's wait_stream does when the caller sits on the default stream) completes without
 * waiting for the lanes, and two lanes really overlap.  (torch.cuda.Stream() objects are ordered against
 * the NULL stream on this stack: recording the per-step dependency there serialised the lanes, measured.)
 * high_priority != 0: the device's greatest stream priority.  Returns NULL on failure (ssd_last_error). */
void* ssd_stream_create(int high_priority);
/* ... restricted to the compute units whose bits are set in cu_mask[0 .. words) (hipExtStreamCreateWithCUMask): lanes on
 * DISJOINT sets of XCDs run truly concurrently, each with a whole number of workgroup rounds of its own. */
void* ssd_stream_create_masked(const unsigned* cu_mask, int words);
int ssd_stream_destroy(void* stream);

/* ---- prior boxes: utils/bbox_utils.py:115-176 (A1-A3) ------------------------------
 * fmaps[levels], n_ars[levels] and ars[levels][n_ars[l]] are HOST arrays; out_dev is
 * [N,4] with N = sum f^2 * (n_ars+1).  Bit-exact vs the NumPy restatement. */
int ssd_priors_count(const int* fmaps, const int* n_ars, int levels);
int ssd_priors(const int* fmaps, const float* const* ars, const int* n_ars, int levels,
               float* out_dev, void* stream);

/* ---- box decode: utils/bbox_utils.py:61-85 (D1) -------------------------------------
 * deltas [B,N,4] (device), priors [N,4] (device) -> out [B,N,4].  var (HOST, 4 floats)
 * may be NULL; when given, deltas are multiplied by it first (models/decoder.py:41). */
int ssd_decode_boxes(const float* priors_dev, const float* deltas_dev, const float* var,
                     int B, int N, float* out_dev, void* stream);

/* ---- SSDDecoder.call: models/decoder.py:36-55 + utils/bbox_utils.py:3-25 (D2,D3) ----
 * and the [3P] tf.image.combined_non_max_suppression semantics.
 * deltas [B,N,4], probs [B,N,L], priors [N,4] device; var HOST[4].
 * Outputs (device, fully overwritten incl. zero padding rows): boxes [B,T,4] clipped to
 * [0,1], labels [B,T] (float class id), scores [B,T], valid [B] (int32).  kept_idx_dev
 * (nullable) [B,T] int32 receives the anchor index behind each row (-1 on padding).
 * Tie rule (TF leaves it unspecified): equal scores -> lower anchor index, then lower
 * class index. */
size_t ssd_decode_nms_workspace_bytes(int B, int N, int L, int max_per_class);
int ssd_decode_nms(const float* deltas_dev, const float* probs_dev, const float* priors_dev,
                   const float* var, int B, int N, int L, int max_per_class, int max_total,
                   float iou_thr, float score_thr,
                   float* boxes_dev, float* labels_dev, float* scores_dev, int* valid_dev,
                   int* kept_idx_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- bbox_utils.non_max_suppression: utils/bbox_utils.py:3-25 (D2) -------------------
 * Generic combined NMS on already-decoded boxes [B,N,1,4] (q == 1) and scores [B,N,C];
 * output order as TF returns it: boxes, scores, classes, valid.  clip_boxes as TF. */
int ssd_combined_nms(const float* boxes_dev, const float* scores_dev, int B, int N, int C,
                     int max_per_class, int max_total, float iou_thr, float score_thr,
                     int clip_boxes, float* boxes_out_dev, float* scores_out_dev,
                     float* classes_out_dev, int* valid_dev, int* kept_idx_dev,
                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- pairwise IoU: utils/bbox_utils.py:27-59 (M1) -----------------------------------
 * boxes [N,4] (boxes_batched == 0, shared across the batch) or [B,N,4]; gt [B,G,4];
 * out [B,N,G].  No epsilon: 0/0 -> NaN exactly like the reference. */
int ssd_iou_map(const float* boxes_dev, int boxes_batched, const float* gt_dev,
                int B, int N, int G, float* out_dev, void* stream);

/* ---- box encode: utils/bbox_utils.py:87-113 (M3): bboxes [N,4], gt [B,N,4] -> [B,N,4] */
int ssd_encode_deltas(const float* bboxes_dev, const float* gt_dev, int B, int N,
                      float* out_dev, void* stream);

/* ---- target assignment: utils/train_utils.py:90-127 (M2) ----------------------------
 * priors [N,4], gt_boxes [B,G,4], gt_labels [B,G] int32 (device); var HOST[4].
 * deltas_out [B,N,4]; label_idx_out [B,N] int32; match_idx_out [B,N] int32 (argmax over
 * G, first max wins) -- both bit-exact; onehot_out (nullable) [B,N,L] fp32. */
int ssd_match_encode(const float* priors_dev, const float* gt_boxes_dev,
                     const int* gt_labels_dev, const float* var, float iou_thr,
                     int B, int N, int G, int L, float* deltas_out_dev,
                     int* label_idx_out_dev, int* match_idx_out_dev, float* onehot_out_dev,
                     void* stream);

/* ---- input pipeline: utils/data_utils.py:22-23 (N4) ---------------------------------------
 * tf.image.convert_image_dtype(uint8 -> float32, x 1/255) + tf.image.resize(bilinear, TF2
 * half-pixel centres, no antialias) in one kernel.  image_u8 [B,H,W,C] uint8 (device) ->
 * out [B,out_h,out_w,C] float32 in [0,1].  Images of different sizes go one call each (B = 1)
 * into their slot of the batch tensor. */
int ssd_preprocess(const unsigned char* image_u8_dev, int B, int H, int W, int C, int out_h, int out_w,
                   float* out_dev, void* stream);

/* ---- augmentation: augmentation.py:4-183 (used at trainer.py:42), the deterministic pieces; the random draws of the
 * reference's tf.random.uniform / sample_distorted_bounding_box calls are INPUTS (host side: tf-ssd_amd/augmentation.py).
 * Images float32 [B,H,W,C] in [0,1] (the reference augments after convert + resize).
 * ssd_image_mean: mean_out[b][c] = mean over H, W of (img + add[b]) (add NULL = 0): expand_image's fill colour
 *   (augmentation.py:142) and adjust_contrast's pivot.
 * ssd_augment_geometry: params [B][10] int32 = {canvas_h, canvas_w, pad_top, pad_left, crop_y, crop_x, crop_h, crop_w,
 *   flip, use_crop}: expand_image (:123-151) as a virtual canvas filled with fill[b][c], patch's tf.slice + tf.image.resize
 *   to out_h x out_w (:176-177, bilinear, half-pixel centres), flip_horizontally (:106) -- one gather kernel, out != img;
 *   use_crop 0 (flip only) needs out_h x out_w == H x W; the whole canvas as the window at its own size materialises it.
 * ssd_augment_color (C = 3, in place): params [B][4] = {brightness delta, contrast factor, hue delta, saturation factor},
 *   flags [B] bits 0..3 = which run (:51-93, reference order), mean [B][3] = contrast's pivot; ends with clip [0,1] (:27). */
int ssd_image_mean(const float* img_dev, int B, int H, int W, int C, const float* add_dev, float* mean_out_dev, void* stream);
int ssd_augment_geometry(const float* img_dev, int B, int H, int W, int C, int out_h, int out_w, const int* params_dev,
                         const float* fill_dev, float* out_dev, void* stream);
int ssd_augment_color(float* img_dev, int B, int H, int W, const float* params_dev, const int* flags_dev,
                      const float* mean_dev, void* stream);

/* ---- training loss: ssd_loss.py:8-65 (N1) ----------------------------------------------
 * CustomLoss.loc_loss_fn + conf_loss_fn in one kernel per image.
 *   actual_deltas / pred_deltas [B,N,4]; actual_labels (one-hot) / pred_labels (probabilities)
 *   [B,N,L] (device).  Either pair may be NULL to evaluate only the other term.
 *   loc_loss [B]  = alpha * sum_pos Huber_{delta=1}(pred - actual summed over 4) / max(#pos, 1),
 *                   positives = anchors with any non-zero actual delta           (ssd_loss.py:18-33)
 *   conf_loss [B] = sum_n (pos + neg)_n * CE_n / max(#pos, 1); CE on probabilities renormalised
 *                   by their sum and clipped to [1e-7, 1-1e-7] ([3P] Keras categorical_crossentropy
 *                   on a non-Softmax-op tensor); pos = any one-hot column 1.. set; neg = rank of
 *                   CE*y0 in descending order (ties: lower anchor index) < int(#pos * ratio)
 *                                                                                  (ssd_loss.py:45-63)
 * Optional outputs: ce_out [B,N], mask_out [B,N] (pos + neg, the reference's final_mask; 2 where a
 * positive also ranks as negative), and the gradients of grad_scale * (loc_loss[b] + conf_loss[b])
 * w.r.t. pred_deltas (grad_deltas [B,N,4]) and w.r.t. the LOGITS behind pred_labels = softmax(z)
 * (grad_logits [B,N,L]); grad_scale = 1/batch gives the Keras batch-mean objective. */
size_t ssd_loss_workspace_bytes(int B, int N);
int ssd_loss(const float* actual_deltas_dev, const float* pred_deltas_dev,
             const float* actual_labels_dev, const float* pred_labels_dev, int B, int N, int L,
             float neg_pos_ratio, float loc_loss_alpha, float* loc_loss_dev, float* conf_loss_dev,
             float* ce_out_dev, float* mask_out_dev, float* grad_deltas_dev, float* grad_logits_dev,
             float grad_scale, void* workspace_dev, size_t workspace_bytes, void* stream);

/* =====================================================================================
 * Conv family (the TF/Keras ops the models dispatch: SURVEY.md 2.3 K1-K7).
 * All tensors NHWC fp32 on the device.
 * ================================================================================== */

enum ssd_act { SSD_ACT_NONE = 0, SSD_ACT_RELU = 1, SSD_ACT_RELU6 = 2 };

/* Geometry of one convolution.  pad_* are explicit (TF SAME / ZeroPadding2D asymmetry is
 * resolved by the caller; ssd_same_pads() below restates the TF rule). */
typedef struct ssd_conv_desc {
    int B, H, W, Cin;          /* input  [B,H,W,Cin]                       */
    int Cout, kh, kw;          /* kernel [kh,kw,Cin,Cout] (Keras HWIO)     */
    int stride, dilation;
    int pad_t, pad_l, pad_b, pad_r;
    int act;                   /* enum ssd_act, applied after scale/shift  */
    int has_residual;          /* add residual [B,Ho,Wo,Cout] before store */
} ssd_conv_desc;

/* TF SAME rule: out=ceil(in/s), p=max((out-1)*s+(k-1)*d+1-in,0),
 * before=p/2, after=p-before.  Returns out size. */
int ssd_same_pads(int in, int k, int stride, int dilation, int* before, int* after);
int ssd_conv_out_size(int in, int k, int stride, int dilation, int pad_before, int pad_after);

/* Packed dense-conv weights: [Npad][Kpad] fp32, K = (ky*kw+kx)*Cin+ci, zero padded (device), followed by four
 * bf16 planes [4][Npad][Kpad] of the same matrix: h, m, l = its EXACT three-way split (x = h + m + l; the "mfma3_*"
 * tile configs run every product as six bf16 MFMAs at fp32 accuracy, csrc/ssd_bf16x3.h) and r = its bf16 rounding
 * (round to nearest even; the "bf16_*" tile configs of the net's precision-1 mode run ONE bf16 MFMA per product of
 * once-rounded operands).  ssd_conv_packed_weight_floats() covers all of it.  scale/shift [Cout] are the folded
 * BatchNorm (or 1/bias) epilogue vectors. */
size_t ssd_conv_packed_weight_floats(int kh, int kw, int Cin, int Cout);
int ssd_conv_pack_weights(const float* hwio_dev, int kh, int kw, int Cin, int Cout,
                          float* packed_dev, void* stream);

/* Conv2D (+BatchNorm +activation +residual): the MFMA implicit-GEMM kernel (K1,K3).
 * Replaces keras Conv2D / BatchNormalization / ReLU call sites
 * (models/ssd_mobilenet_v2.py:16-32, models/ssd_vgg16.py:52-91, models/header.py:60-61).
 * out address of pixel (b, p=oy*Wo+ox), channel n:
 *     out_dev + b*out_batch_stride + p*out_pixel_stride + n
 * (pass 0 for the strides to get the dense [B,Ho,Wo,Cout] default), which is how the
 * head convs write straight into the concatenated [B,N,K] buffers (models/header.py:34-41). */
int ssd_conv2d(const ssd_conv_desc* d, const float* in_dev, const float* packed_w_dev,
               const float* scale_dev, const float* shift_dev, const float* residual_dev,
               float* out_dev, long out_batch_stride, long out_pixel_stride, void* stream);

/* Tuning / test hooks of the same kernel family: explicit tile configuration (-1 = cost
 * model), optional deterministic split-K (split_k > 1 needs split_k*M*Cout floats). */
int ssd_conv_num_configs(void);
const char* ssd_conv_config_name(int config);
int ssd_conv2d_ex(const ssd_conv_desc* d, const float* in_dev, const float* packed_w_dev,
                  const float* scale_dev, const float* shift_dev, const float* residual_dev,
                  float* out_dev, long out_batch_stride, long out_pixel_stride, int config,
                  int split_k, float* splitk_ws_dev, void* stream);

/* The same op on the LDS-DMA tiles ("dma3_*" / "dmab_*" configs, csrc/ssd_convdma.hip): the INPUT is handed over as
 * bf16 planes [planes][n] (planes = 3: the exact split x = h + m + l of the fp32 activation, fp32 results as above;
 * planes = 1: its bf16 rounding -- the bf16 mode's storage format), `in_plane_stride` ELEMENTS between planes; both
 * operands then reach LDS by `buffer_load ... lds` without passing through registers.  Inside a plane the NHWC tensor
 * of `channels` channels (a multiple of 32) and P = n / channels pixels is stored SLICE-MAJOR, [channels / 32][P][32]:
 * element (pixel, c) at ((c / 32) * P + pixel) * 32 + c % 32 -- the 16 rows x 64 bytes of one copy instruction are then
 * 1 KB of contiguous memory.  ssd_split_planes writes such
 * planes from an fp32 tensor (n % 4 == 0), ssd_join_planes restores fp32 (exactly, for planes = 3); the conv can write
 * its own output as planes too (out_planes_dev != NULL: dense outputs with Cout % 32 == 0) for the next layer. */
int ssd_split_planes(const float* x_dev, long n, int channels, int planes, void* planes_dev, long plane_stride, void* stream);
int ssd_join_planes(const void* planes_dev, long n, int channels, int planes, long plane_stride, float* x_dev, void* stream);
int ssd_conv2d_planes(const ssd_conv_desc* d, const void* in_planes_dev, int planes, long in_plane_stride,
                      const float* packed_w_dev, const float* scale_dev, const float* shift_dev,
                      const float* residual_dev, float* out_dev, long out_batch_stride, long out_pixel_stride,
                      void* out_planes_dev, long out_plane_stride, int config, int split_k,
                      float* splitk_ws_dev, void* stream);

/* Winograd F(2x2,3x3) variant of the same op for 3x3 stride-1 dilation-1 convs with Cin % 16 == 0
 * (the SSD head convs, models/header.py:60-61, and VGG16's 3x3 backbone, models/ssd_vgg16.py:52-72):
 * 2.25x fewer multiplications on the same fp32 MFMA, fused input / output transforms.  Weights are
 * transformed once ([16][Npad][Cin] floats).  The graph runner's autotune picks it per layer. */
size_t ssd_conv_wino_weight_floats(int Cin, int Cout);
int ssd_conv_wino_pack_weights(const float* hwio_dev, int Cin, int Cout, float* wino_w_dev, void* stream);
int ssd_conv_wino_num_configs(void);
int ssd_conv2d_wino(const ssd_conv_desc* d, const float* in_dev, const float* wino_w_dev,
                    const float* scale_dev, const float* shift_dev, float* out_dev,
                    long out_batch_stride, long out_pixel_stride, int wino_config, int split_k,
                    float* splitk_ws_dev, void* stream);

/* DepthwiseConv2D 3x3 (+BN +act): weights [3,3,C] (Keras [3,3,C,1]) device (K2). */
int ssd_dwconv3x3(const float* in_dev, int B, int H, int W, int C, int stride,
                  int pad_t, int pad_l, int pad_b, int pad_r,
                  const float* w_dev, const float* scale_dev, const float* shift_dev,
                  int act, float* out_dev, void* stream);

/* MaxPool2D with TF SAME semantics (padded cells ignored): models/ssd_vgg16.py:54-73 (K4) */
int ssd_maxpool2d(const float* in_dev, int B, int H, int W, int C, int k, int stride,
                  int pad_t, int pad_l, int pad_b, int pad_r, float* out_dev, void* stream);

/* L2Normalization: models/ssd_vgg16.py:7-31 (K5): x * rsqrt(max(sum_c x^2, 1e-12)) * gamma_c */
int ssd_l2norm(const float* in_dev, long pixels, int C, const float* gamma_dev,
               float* out_dev, void* stream);

/* softmax over the last dim (models/header.py:64) (K7); in-place allowed. */
int ssd_softmax(const float* in_dev, long rows, int L, float* out_dev, void* stream);

/* =====================================================================================
 * Graph runner: models/ssd_mobilenet_v2.py:7-35 (F3) / models/ssd_vgg16.py:33-97 (F6)
 * + models/header.py:43-67 (F1,F2) [+ models/decoder.py:57-69 (D4)].
 * A net owns its packed weights, activation arena and (optionally) a captured hipGraph.
 * ================================================================================== */
typedef struct ssd_net ssd_net;

enum ssd_backbone { SSD_MOBILENET_V2 = 0, SSD_VGG16 = 1 };

/* n_ars[levels] = len(aspect_ratios[l]) (anchors per cell = n_ars+1), total_labels = L. */
ssd_net* ssd_net_create(int backbone, int img_size, int levels, const int* n_ars,
                        int total_labels);
void ssd_net_destroy(ssd_net* net);

/* Parameter table in Keras order / names (e.g. "Conv1/kernel", "bn_Conv1/gamma",
 * "block_1_expand/kernel", "extra1_1/bias", "1_conv_label_output/kernel" ...). */
int ssd_net_num_params(const ssd_net* net);
const char* ssd_net_param_name(const ssd_net* net, int i);
int ssd_net_param_rank(const ssd_net* net, int i);
const int* ssd_net_param_shape(const ssd_net* net, int i);
/* Copy one parameter from HOST memory in its Keras layout (count = product of shape). */
int ssd_net_set_param(ssd_net* net, const char* name, const float* host_data, size_t count);
/* Read a parameter back (HOST), Keras layout. */
int ssd_net_get_param(const ssd_net* net, const char* name, float* host_out, size_t count);
/* Fold BN, pack weights for the MFMA kernels, size the arena for `max_batch`. */
int ssd_net_finalize(ssd_net* net, int max_batch);
/* Tile-configuration table chosen by finalize's on-device autotune, as text (one
 * "layer config split_k" line per conv).  get returns the length (buf may be NULL);
 * set installs a table that the next finalize uses instead of re-tuning (if complete/valid). */
long ssd_net_get_tuning(const ssd_net* net, char* buf, size_t cap);
int ssd_net_set_tuning(ssd_net* net, const char* text);
/* How the last finalize chose its kernels: *from_table = conv layers taken from preset lines,
 * *timed = choices timed on the device (conv layers + whole-image blocks + launch mode).  timed == 0
 * means the table was complete: nothing depends on timing noise, results are reproducible bit for bit
 * across processes.  (The reference's Keras graph is deterministic in which kernels it runs; this is
 * the counterpart guarantee -- models/ssd_mobilenet_v2.py:7-35.) */
int ssd_net_tuning_stats(const ssd_net* net, int* from_table, int* timed);
/* Device memory a finalized net holds, bytes: out[0] activation arena (one fp32 slot per tensor x max_batch), out[1] bf16
 * planes of the activations its CHOSEN LDS-DMA conv tiles read (allocated for the finalize-time race, freed again where
 * no chosen tile reads them), out[2] partial-sum slabs of the whole-image block kernel, out[3] split-K slabs.  A lane
 * replica (models/decoder.py) holds the same again. */
int ssd_net_memory_bytes(const ssd_net* net, size_t out[4]);
/* sha256 (first 16 hex digits) of the kernel sources this library was built from (csrc/build.sh);
 * shipped tuning tables record the build they were measured on. */
const char* ssd_build_id(void);
int ssd_net_num_priors(const ssd_net* net);
int ssd_net_feature_map_size(const ssd_net* net, int level);

/* image [B,S,S,3] fp32 in [0,1] (device) -> pred_deltas [B,N,4], pred_labels [B,N,L]
 * (softmax probabilities), exactly the pair the reference model outputs. */
int ssd_net_forward(ssd_net* net, const float* image_dev, int B, float* deltas_out_dev,
                    float* probs_out_dev, void* stream);

/* Forward + SSDDecoder (get_decoder_model(...).predict on one batch): priors [N,4] dev. */
int ssd_net_predict(ssd_net* net, const float* image_dev, int B, const float* priors_dev,
                    const float* var, int max_total, float iou_thr, float score_thr,
                    float* boxes_dev, float* labels_dev, float* scores_dev, int* valid_dev,
                    void* stream);

/* Debug/test hook: copy the activation of a named layer of the LAST forward to the host.
 * Returns the element count (or negative error); host_out may be NULL to query size. */
long ssd_net_fetch_activation(ssd_net* net, const char* layer, float* host_out, size_t cap);
/* ... and the same activation as the bf16 planes the LDS-DMA conv tiles read (csrc/ssd_convdma.hip), joined back to
 * fp32: *planes_out = 3 (exact split: equals the fp32 activation bit for bit) or 1 (its bf16 rounding); returns 0 when
 * no running layer asked for this tensor's planes in the last forward. */
long ssd_net_fetch_planes(ssd_net* net, const char* layer, float* host_out, size_t cap, int* planes_out);

/* Per-layer algorithmic work of one forward at batch B (for roofline accounting). */
int ssd_net_num_layers(const ssd_net* net);
const char* ssd_net_layer_name(const ssd_net* net, int i);
const char* ssd_net_layer_kind(const ssd_net* net, int i);   /* "conv","dw","pool",... */
const char* ssd_net_layer_config(const ssd_net* net, int i); /* autotuned conv tile config */
double ssd_net_layer_flops(const ssd_net* net, int i, int B); /* 2*MACs                  */
double ssd_net_layer_bytes(const ssd_net* net, int i, int B); /* in + out + weights      */
/* FLOPs the chosen kernel issues: = layer_flops except Winograd layers (x 16/36, whole border tiles) */
double ssd_net_layer_executed_flops(const ssd_net* net, int i, int B);
/* Options: "use_graph" replays each forward/predict step as one captured hipGraph (keyed by the
 * pointers/sizes of the call; the NULL stream is served through an internal stream) -- by default
 * ssd_net_finalize races replay against direct launches on the device at max_batch and uses replay only
 * where it is more than 1 % ahead; setting the option pins the mode;
 * "fuse_blocks" (default 1) runs eligible MobileNetV2 inverted-residual blocks
 * (expand -> depthwise -> project) as one fused kernel; 0 runs them as three layers (then
 * every intermediate activation is inspectable); "fuse_dwproj" (default 1, needs fuse_blocks)
 * runs depthwise -> project of the remaining blocks (7-16) as one kernel behind the expand
 * GEMM; "fuse_image" (default 1, needs fuse_blocks) lets the whole-image block kernel (blocks
 * 7-12 / 14-16: expand -> depthwise -> project with the expanded map kept on the CU) replace expand
 * GEMM + depthwise/project where it won finalize's on-device race, 2 forces it wherever it applies,
 * 0 disables it; "image_split" (default 1; fp32 nets) also lets that kernel's split-bf16 form (fp32 results from six bf16
 * matrix products per product, weights staged through LDS) into the race -- table line "<block> image 2" --, 0 leaves
 * it out; "image_ticket" (default 0) makes that kernel combine its channel-group partial sums
 * inside the launch (arrival ticket, last arriver) instead of by a second launch; "overlap_heads" (default 1) runs the SSD head convs on side streams ("tail_on_side", default 0,
 * swaps the roles: big head convs on the caller's stream, the small tail layers on the side streams -- measured slower); "use_wino" (default 1)
 * offers the Winograd F(2x2,3x3) kernels to finalize's autotune for the 3x3 stride-1 convs;
 * "precision" (default 0 = fp32, the reference's arithmetic: trainer.py:50-54 has no mixed precision; 1 = bf16, this
 * build's extension for BASELINE.json configs[3] / [4]): every matrix operand of the dense / 1x1 convolutions is rounded
 * once to bf16 (nearest even), each product is one bf16 MFMA with fp32 accumulation ("bf16_*" tiles, the bf16 forms of
 * the stem, row-band and whole-image block kernels); BatchNorm shifts, activations, residual adds, depthwise taps, softmax and
 * the box math stay fp32, activations stay fp32 in HBM; the training step then runs its forward / backward-data convs
 * on the bf16 tiles (fp32 master weights, weight gradients and Adam).  Takes effect at the next finalize. */
int ssd_net_set_option(ssd_net* net, const char* name, int value);
/* Diagnostics: per-phase mean cycles per wave of one fused block layer (clock64 inside the
 * kernel): prologue, expand, depthwise, project, weight staging, epilogue. */
int ssd_net_profile_fused(ssd_net* net, const char* layer, int B, double* cycles_out6);
/* Live per-layer hipEvent timing of ssd_net_forward / ssd_net_predict on their stream.
 * read_timing sums the durations (ms) of the forwards recorded since the last read into
 * ms_sum_out[num_layers + 1] (last entry: decode+NMS of predict) and reports their count. */
int ssd_net_set_timing(ssd_net* net, int enabled);
int ssd_net_read_timing(ssd_net* net, float* ms_sum_out, int* forwards_out);
/* Time every layer with hipEvents on `stream` (reps forwards); ms_out[num_layers]. */
int ssd_net_profile_layers(ssd_net* net, const float* image_dev, int B, int reps,
                           float* ms_out, void* stream);

/* =====================================================================================
 * Training step: what Keras runs for the reference's
 * `ssd_model.compile(optimizer=Adam(1e-3), loss=[loc_loss_fn, conf_loss_fn])` +
 * `ssd_model.fit(...)` (trainer.py:50-76), one call per phase so that the host can all-reduce
 * the flat gradient vector over RCCL between backward and the optimiser.
 * MobileNetV2 graph (BASELINE configs[3]); training-mode BatchNorm (batch statistics, moving
 * averages updated with momentum 0.999 / eps 1e-3 of keras-applications MobileNetV2).
 * ================================================================================== */
/* Plan the training buffers for `batch` images per step; moves the trainable parameters into
 * one flat vector (parameter-table order, Keras layouts; moving_mean / moving_variance are not
 * trainable).  Keeps the Adam state when re-planned for a larger batch. */
int ssd_net_train_begin(ssd_net* net, int batch);
size_t ssd_net_trainable_floats(const ssd_net* net);
/* Offset of a parameter inside the flat trainable vector (-1: unknown / not trainable). */
long ssd_net_trainable_offset(const ssd_net* net, const char* name);
/* Training-mode forward + ssd_loss + backward on one batch (device pointers):
 * image [B,S,S,3], actual_deltas [B,N,4], actual_labels [B,N,L] (calculate_actual_outputs).
 * grads_flat [ssd_net_trainable_floats] <- d mean_b(loc_b + conf_b) / d parameter (caller-owned);
 * loc_loss / conf_loss [B] <- per-image loss terms (nullable). */
int ssd_net_train_forward_backward(ssd_net* net, const float* image_dev, int B,
                                   const float* actual_deltas_dev, const float* actual_labels_dev,
                                   float neg_pos_ratio, float loc_loss_alpha, float* grads_flat_dev,
                                   float* loc_loss_dev, float* conf_loss_dev, void* stream);
/* Gradient buckets for the batch data-parallel step (the reference trains on one device, trainer.py:50-76; the
 * RCCL exchange is this build's SURVEY.md 8e row 2): bucket k = flat offsets [lo[k], lo[k + 1]), lo ascending
 * from 0.  The backward finishes the flat gradient vector from its end (heads first, stem last) and records an
 * event per bucket when it is final; ssd_net_train_wait_bucket(net, k, stream) orders `stream` -- the one the
 * all-reduce of bucket k is issued on -- behind that point, so the exchange overlaps the rest of the backward. */
int ssd_net_train_set_buckets(ssd_net* net, int n, const long* lo);
int ssd_net_train_wait_bucket(ssd_net* net, int k, void* stream);
/* Adam (Keras defaults beta1 0.9, beta2 0.999, eps 1e-7; TF ApplyAdam form) on every trainable
 * parameter; grads are multiplied by grad_scale first (1/world_size after a SUM all-reduce). */
int ssd_net_adam_step(ssd_net* net, const float* grads_flat_dev, float lr, float beta1, float beta2,
                      float eps, float grad_scale, void* stream);
long ssd_net_train_steps(const ssd_net* net);
/* Measurement: matrix-core FLOPs the last ssd_net_train_forward_backward issued, per instruction family --
 * out3[0] conv forward + backward-data on fp32-MFMA tiles, out3[1] the same on split-bf16 tiles (six bf16 MFMAs per
 * fp32 product), out3[2] weight gradients (fp32 MFMA).  bench.py --train prices each family at its own peak. */
int ssd_net_train_matrix_flops(const ssd_net* net, double* out3);
/* Sum of the layers' regularisation losses at the current weights: 5e-4 * sum(kernel^2) over VGG16's
 * backbone / extra convs (reference models/ssd_vgg16.py:44-45), 0 for MobileNetV2.  Keras adds this term to
 * the `loss` and `val_loss` that fit() logs and ModelCheckpoint(save_best_only) monitors (trainer.py:56-63).
 * Synchronous (legacy stream, one float copied to the host). */
int ssd_net_regularization_loss(ssd_net* net, float* host_out);
/* Debug / parity hook: copy a buffer of the last training forward/backward (batch B) to the host:
 * "probs", "deltas", "grad_logits", "grad_deltas", "<tensor>", "grad:<tensor>", "pre:<layer>",
 * "mean:<layer>", "var:<layer>".  Returns the element count (host_out NULL: query). */
long ssd_net_train_fetch(ssd_net* net, const char* what, int B, float* host_out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SSD_HIP_H */

// Internal interface between the conv kernels (ssd_conv.hip / ssd_ops.hip) and the graph
// runner (ssd_net.hip).
#pragma once
#include "common.h"

namespace ssd {

// Fully resolved launch parameters of one dense convolution (implicit GEMM view:
// out[M = B*Ho*Wo, N = Cout] = X[M, K = kh*kw*Cin] * W[K, N]).
struct ConvParams {
    const float* in;
    const float* w;        // packed [Npad][Kpad]
    const short* w3;       // the same as four bf16 planes [4][Npad][Kpad]: the exact split h, m, l and the bf16 rounding r (ssd_conv3.hip) or nullptr
    const float* scale;    // [Cout] or nullptr (== 1)
    const float* shift;    // [Cout] or nullptr (== 0)
    const float* residual; // dense [M][Cout] or nullptr
    float* out;
    int B, H, W, Cin, Ho, Wo, Cout;
    int kh, kw, stride, dil, pad_t, pad_l;
    int K, Kpad, Npad;
    long M;
    long out_batch_stride, out_pixel_stride;
    // optional second destination: output channels n >= n_split go to out2 (column n - n_split)
    // with their own strides -- the fused SSD label+box head conv of one level.
    int n_split;           // 0: single destination
    float* out2;
    long out2_batch_stride, out2_pixel_stride;
    int vec_store2;
    int act;
    int vec_store;         // 1: float4 stores are aligned
    int split_k;           // >1: partial sums to `partial` [split][M][Cout], epilogue deferred
    float* partial;
    const float* wino_w;   // Winograd F(2x2,3x3) weights U [16][Npad][Cin] (ssd_wino.hip) or nullptr
    int bf16;              // the net's precision-1 mode: the cost model (conv_pick_config) may take the bf16 tiles
    // LDS-DMA tiles (csrc/ssd_convdma.hip): the INPUT as bf16 planes [np][B*H*W*Cin] (np = 3: exact split h, m, l of the
    // fp32 activation; np = 1: its bf16 rounding), `xp_plane` elements between planes; nullptr = not available
    const short* xp;
    long xp_plane;
    int xp_np;
    // ... and the OUTPUT also written as planes by the epilogue (dense [M][Cout] outputs only; Cout % 4 == 0) for the
    // LDS-DMA tiles of this layer's consumers; nullptr = fp32 only
    short* op;
    long op_plane;
    int op_np;
};
// may a net of this precision (0 fp32, 1 bf16) choose config `cfg`?  fp32 nets never take the bf16 (one-product) tiles;
// bf16 nets take them instead of the split-bf16 / Winograd tiles (the fp32-MFMA and skinny tiles serve the small tail layers
// in both modes: more accurate than asked for)
bool conv_config_allowed(int cfg, int precision);

// One fused MobileNetV2 inverted-residual block (csrc/ssd_fused.hip).
struct FusedBlockParams {
    const float* x;
    float* y;
    const float* we;            // expand weights, packed [Ce][kpad_e]
    const float *es, *eh;       // folded expand BN [Ce]
    const float* wd;            // depthwise weights [9][Ce]
    const float *ds, *dh;       // folded depthwise BN [Ce]
    const float* wp;            // project weights, packed [npad_p][kpad_p]
    const float *ps, *ph;       // folded project BN [Cout]
    int residual;               // y += x (stride 1, Cin == Cout)
    int B, H, W, Cin, Ce, Cout, Ho, Wo, stride, pad_t, pad_l;
    int kpad_e, kpad_p, npad_p;
    int tiles_y, tiles_x;       // filled by the launcher
    // whole-image kernel (csrc/ssd_imgblock.hip): expanded-channel groups per image and their meeting point
    int bf16;                   // the net's precision-1 mode: bf16 forms of the band / whole-image kernels (operands rounded once to bf16, one MFMA per product)
    int groups;                 // G >= 1 (filled by the caller from image_block_groups)
    const short* we3;           // split-bf16 band kernel (csrc/ssd_band3.hip): bf16 planes of we [4][Ce][32] (h, m, l, r) ...
    const short* wp3;           // ... and of wp [4][npad_p][pairs][4][8]
    int bands;                  // row-band kernel (csrc/ssd_bandblock.hip): bands per image (filled by the launcher)
    float* slabs;               // [G][B][Ho*Wo][Cout] partial sums (G > 1)
    unsigned* tickets;          // [B] arrival counters, zero between launches
    float* e_out;               // optional: the expanded map [B,H,W,Ce] is ALSO written to HBM (block 13: SSD feature map 1)
    // optional bf16 planes of the outputs for the LDS-DMA conv tiles that read them (csrc/ssd_convdma.hip): y (written by
    // the direct epilogue / the combine launch) and the expanded map e_out; planes_np = 3 exact split, 1 bf16 rounding
    short* y_planes;
    long y_plane;
    short* e_planes;
    long e_plane;
    int planes_np;
    int form2;                  // whole-image kernel: 1 = take the second form (csrc/ssd_imgblock2.hip) where it has a configuration
    long long* dbg;             // optional per-phase cycle counters [blocks][8] (profiling builds)
    int ablate;                 // diagnostics: 1 skip expand MFMAs, 2 skip depthwise math, 4 skip project MFMAs, 8 skip expand epilogue math
};
// Fused MobileNetV2 stem (Conv1 -> expanded_conv_depthwise -> expanded_conv_project).
struct StemParams {
    const float* x;             // image [B,H,W,3]
    float* y;                   // [B,H1,W1,16]
    const float* w1;            // Conv1 weights packed [32][kpad1]
    const float *s1, *h1;       // folded Conv1 BN [32]
    const float* wd;            // depthwise weights [9][32]
    const float *sd, *hd;
    const float* wp;            // project weights packed [16][kpadp]
    const float *sp, *hp;       // folded project BN [16]
    int B, H, W, H1, W1, pad_t, pad_l, kpad1, kpadp;
    int tiles_y, tiles_x;
    int ablate;                 // diagnostics (SSD_STEM_ABLATE): 1 skip Conv1 math, 2 depthwise, 4 project MFMAs, 8 patch loads
    int bf16;                   // the net's precision: 1 = Conv1 / project operands rounded once to bf16
};
// Depthwise 3x3 + BN + ReLU6 -> project 1x1 + BN (+ residual) of one MobileNetV2 block
// (csrc/ssd_dwproj.hip); BN scales are folded into wd / wp by the caller.
struct DwProjParams {
    const float* e;             // expanded activations [B,H,W,Ce] (after expand BN + ReLU6)
    const float* wd;            // depthwise weights [9][Ce], BN scale folded in
    const float* dh;            // depthwise BN shift [Ce]
    const float* wp;            // project weights packed [npad_p][kpad_p], BN scale folded in
    const float* ph;            // project BN shift [Cout]
    const float* res;           // residual [B,Ho,Wo,Cout] or nullptr
    float* y;                   // [B,Ho,Wo,Cout]
    int B, H, W, Ce, Cout, Ho, Wo, stride, pad_t, pad_l;
    int kpad_p, npad_p;
    int tiles_y, tiles_x;       // filled by the launcher
    int ablate;                 // diagnostics (SSD_DWPROJ_ABLATE): 1 skip depthwise math, 2 skip MFMAs, 4 skip chunk loads
    int bf16;                   // the net's precision-1 mode: project on the bf16 matrix cores (operands rounded once, fp32 accumulation)
};
bool dwproj_supported(const DwProjParams& p);
int launch_dwproj(DwProjParams p, hipStream_t st);
bool stem_supported(const StemParams& p);
int launch_stem(StemParams p, hipStream_t st);
bool image_block2_supported(const FusedBlockParams& p);
int launch_image_block2(FusedBlockParams p, hipStream_t st);
bool fused_block_supported(const FusedBlockParams& p);
int launch_fused_block(FusedBlockParams p, hipStream_t st);
bool band3_block_supported(const FusedBlockParams& p);
size_t band3_we_shorts(int Ce);
size_t band3_wp_shorts(int npad_p, int Ce);
int launch_band3_pack(const float* we, int Ce, int Cin, int kpad_e, short* we3, const float* wp, int npad_p, int kpad_p,
                      short* wp3, hipStream_t st);
int launch_band3_block(FusedBlockParams p, hipStream_t st);
bool band_block_supported(const FusedBlockParams& p);
int launch_band_block(FusedBlockParams p, hipStream_t st);
bool image_block_supported(const FusedBlockParams& p);
bool image_block_split_fits(FusedBlockParams p);     // the split-bf16 form's LDS tiles fit at p.groups
int image_block_groups(const FusedBlockParams& p, int B);
size_t image_block_slab_floats(const FusedBlockParams& p, int B);
int launch_image_block(FusedBlockParams p, hipStream_t st);
int stem_form(int precision);        // 0 fp32-MFMA, 3 split-bf16 (fp32 results), 1 bf16 operands

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline int conv_kpad(int K) { return round_up(K, 32); }
inline int conv_npad(int Cout) { return round_up(Cout, 16); }

int conv_num_configs();
const char* conv_config_name(int cfg);
// true if config `cfg` can run these parameters (alignment / shape constraints).
bool conv_config_valid(int cfg, const ConvParams& p);
int conv_pick_config(const ConvParams& p);                 // heuristic
long conv_grid_blocks(int cfg, const ConvParams& p);       // blocks of the un-split grid
int conv_k_tiles(int cfg, const ConvParams& p);            // K tiles per block without split
int conv_launch(const ConvParams& p, int cfg, hipStream_t st);
size_t conv_splitk_workspace_floats(const ConvParams& p, int cfg);

int fill_conv_params(const ssd_conv_desc* d, ConvParams* p);   // geometry + validation
int launch_splitk_reduce(const ConvParams& p, hipStream_t st);

// Winograd F(2x2, 3x3) path (csrc/ssd_wino.hip): config ids [conv_num_mfma_configs(), +wino_num_configs())
int conv_num_mfma_configs();
int wino_num_configs();
const char* wino_config_name(int i);
bool wino_applicable(const ConvParams& p);
bool wino_config_valid(int i, const ConvParams& p);
long wino_grid_blocks(int i, const ConvParams& p);
int wino_k_tiles(const ConvParams& p);
int wino_launch(const ConvParams& p, int i, hipStream_t st);
size_t wino_weight_floats(int Cin, int Cout);
int launch_wino_pack(const float* hwio, int Cin, int Cout, int Npad, int row_off, float* U, hipStream_t st);

// Skinny path (csrc/ssd_skinny.hip): small-M / long-K convs with the K split inside the workgroup;
// config ids [conv_num_mfma_configs() + wino_num_configs(), + skinny_num_configs())
int skinny_num_configs();
const char* skinny_config_name(int i);
bool skinny_config_valid(int i, const ConvParams& p);
long skinny_grid_blocks(int i, const ConvParams& p);
int skinny_launch(const ConvParams& p, int i, hipStream_t st);

// Split-bf16 implicit-GEMM tiles (csrc/ssd_conv3.hip); config ids behind the skinny ones
int mfma3_num_configs();
const char* mfma3_config_name(int i);
bool mfma3_config_valid(int i, const ConvParams& p);
long mfma3_grid_blocks(int i, const ConvParams& p);
int mfma3_k_tiles(const ConvParams& p);
void mfma3_tile(int i, int* BM, int* BN);
int mfma3_launch(const ConvParams& p, int i, hipStream_t st, bool bf16 = false);
const char* bf16_config_name(int i);     // bf16 (one-product) form of split-bf16 tile i; config ids behind the mfma3 ones

// LDS-DMA tiles over pre-split bf16 activation planes (csrc/ssd_convdma.hip); config ids behind the bf16 ones:
// "dma3_*" (np = 3, fp32 nets) then "dmab_*" (np = 1, the bf16 mode)
int dma_num_configs();
const char* dma_config_name(int i, int np);
bool dma_config_valid(int i, int np, const ConvParams& p);
long dma_grid_blocks(int i, const ConvParams& p);
int dma_k_tiles(int np, const ConvParams& p);
void dma_tile(int i, int* BM, int* BN);
int dma_launch(const ConvParams& p, int i, int np, hipStream_t st);
int launch_split_planes(const float* x, long n, int C, int np, short* planes, long plane, hipStream_t st);
int launch_join_planes(const short* planes, long n, int C, int np, long plane, float* x, hipStream_t st);
bool conv_config_is_dma(int cfg);          // either family
bool conv_config_writes_planes(int cfg, const ConvParams& p);   // the family's epilogue (conv_epilogue / splitk_reduce_kernel) honours ConvParams::op

int launch_dwconv3x3(const float* in, int B, int H, int W, int C, int stride, int pad_t, int pad_l,
                     int Ho, int Wo, const float* w, const float* scale, const float* shift, int act,
                     float* out, hipStream_t st);
int launch_maxpool(const float* in, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                   int Ho, int Wo, float* out, hipStream_t st, short* planes = nullptr, long plane = 0, int np = 0);
int launch_l2norm(const float* in, long pixels, int C, const float* gamma, float* out, hipStream_t st,
                  short* planes = nullptr, long plane = 0, int np = 0);
int launch_softmax(const float* in, long rows, int L, float* out, hipStream_t st);
// (csrc/ssd_bbox.hip) the SSDDecoder with the softmax fused into its compaction kernel: head LOGITS in; `ws` is a
// workspace of ssd_decode_nms_workspace_bytes(ws_batch, ...) -- carved for ws_batch >= B images whatever B a call runs --
// whose candidate counters are zero (zeroed once at allocation; every call leaves them zero again)
bool decode_nms_fused_ok(int L);
int decode_nms_fused(const float* deltas, const float* logits, const float* priors, const float* var, int B, int N, int L,
                     int max_per_class, int max_total, float iou_thr, float score_thr, float* boxes, float* labels,
                     float* scores, int* valid, void* ws, size_t ws_bytes, int ws_batch, hipStream_t st);
int launch_pack_weights(const float* hwio, int K, int Cout, int Kpad, int Npad, float* packed, hipStream_t st);
// packed fp32 weights + room for their four bf16 planes (conv_split_planes points at them)
inline size_t conv_packed_floats(int K, int Cout) {
    const size_t n = (size_t)conv_kpad(K) * conv_npad(Cout);
    return n + 2 * n;
}
inline const short* conv_split_planes(const float* packed, int K, int Cout) {
    return reinterpret_cast<const short*>(packed + (size_t)conv_kpad(K) * conv_npad(Cout));
}
// after ALL launch_pack_weights calls of a buffer; planes slice-major [Kpad / 32][Npad][32] for the conv tiles (ssd_bf16x3.h),
// row_major: plain [Npad][Kpad] (the whole-image block kernels' expand / project matrices)
int launch_pack_split(float* packed, int K, int Cout, hipStream_t st, bool row_major = false);
// out[r][c] = in[r][c] * scale[r] (rows >= nvalid copied unscaled); out[r][c] = in[r][c] * scale[c]
int launch_scale_rows(const float* in, const float* scale, int rows, int nvalid, int cols, float* out, hipStream_t st);
int launch_scale_cols(const float* in, const float* scale, int rows, int cols, float* out, hipStream_t st);
int launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                   int C, float* scale, float* shift, hipStream_t st);

}  // namespace ssd

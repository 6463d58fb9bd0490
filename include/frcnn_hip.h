/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so, the MI355X (gfx950) implementation of the
 * Faster R-CNN hot path of Kelicious/faster_rcnn.
 *
 * The reference has no FFI of its own: its "operator API" for this path is a set of
 * Python call signatures over numpy arrays and the Keras model duck type
 * (SURVEY.md 8(b)).  Each entry point below names the reference function it
 * replaces (paths relative to faster_rcnn/ in the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures.
 *   - Every pointer is a DEVICE pointer unless its name ends in _h (host).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     performs no allocation and no host synchronisation (hipGraph-capturable).
 *     Scratch memory is supplied by the caller: ask frcnn_*_workspace_bytes().
 *   - Return value: 0 = ok, <0 = error (FRCNN_E_*); frcnn_last_error() gives the text
 *     of the calling thread's last failure.  No exceptions cross the boundary.
 *   - Box layout is always [x1, y1, x2, y2]; activations are NHWC f32; flat anchor
 *     index i = (y*cols + x)*A + a (rpn_util.py:143-156).
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK          0
#define FRCNN_E_ARG      -1   /* bad argument (null pointer, size out of range) */
#define FRCNN_E_WORKSPACE -2  /* workspace too small */
#define FRCNN_E_HIP      -3   /* a HIP runtime call failed */
#define FRCNN_E_UNSUPPORTED -4

#define FRCNN_MAX_ANCHORS 32
#define FRCNN_NMS_MAX_BOXES 12288   /* pre-NMS candidates one frcnn_nms_* call accepts */

const char* frcnn_last_error(void);
/* The ABI revision this header describes; frcnn_version() returns the library's.  A host built against another revision must not
 * call into the library (the Python mirror checks at load): 100 = rounds 1-3; 101 = frcnn_detections takes det_threshold as a double
 * (round 4); 102 = the f16x3 conv engine, magnitude records, frcnn_conv2d_engine, the RPN sampling entry points (round 5);
 * 103 = frcnn_refresh_h3_planes, frcnn_roi_crop_resize_fwd_batch (additions only).
 * 105 = padded canvases: frcnn_preprocess_u8_canvas, frcnn_zero_outside, frcnn_decode_proposals_canvas (additions only).
 * 106 = frcnn_preprocess_u8_canvas takes the image's offset in the canvas (even canvases for every parity).
 * 107 = frcnn_conv2d_fwd_h3_planes_res (addition only).
 * 108 = frcnn_pool2d_fwd_planes (addition only).
 * 104 = the f16x3 engine's fences: frcnn_h3_planes.status (a THIRD field: recompile hosts that pass the struct), status word in a
 *       magnitude record, frcnn_amax_status. */
#define FRCNN_ABI_VERSION 108
int frcnn_version(void);
/* number of HIP devices visible; does not initialise a context */
int frcnn_device_count(void);

/* ------------------------------------------------------------------ anchors */
/* resnet.preprocess / vgg.preprocess (resnet.py:64-75; vgg.py:52-57): out[p][c] = (float)((double)img[p][c] - mean[c])
 * for an interleaved 3-channel u8 image (BGR as cv2.imread delivers it); mean3_h is a HOST array of 3 doubles.
 * Bit-identical to the host path (float64 subtraction, one rounding to f32). */
int frcnn_preprocess_u8(const uint8_t* img_hwc, size_t n_pixels, const double* mean3_h, float* out, void* stream);
/* shapes.Image.data (shapes.py:19-29): cv2.resize(img, (width, height), interpolation=cv2.INTER_CUBIC) of the decoded uint8
 * frame, then the optional horizontal flip (data[:, ::-1]).  OpenCV's 8-bit algorithm: four taps per axis (border replicated),
 * cubic weights (A = -0.75) computed in f32 and rounded to 11-bit fixed point, both passes in integers, (v + 2^21) >> 22,
 * saturate.  frcnn_resize_cubic_taps is a HOST function filling one axis' table [dst][8] = {4 source indices, 4 weights};
 * frcnn_resize_cubic_u8 takes DEVICE copies of the x and y tables and writes dst [dst_h][dst_w][3].  Bit-identical to the
 * integer restatement in shapes._resize (tests); like it, not pinned against cv2 itself (absent offline).
 * `flip`: bit 0 = horizontal flip; bit 1 (round 6) = the source frame is RGB as a JPEG decoder delivers it and the destination gets
 * B, G, R -- cv2.imread's order (shapes.py:23) -- so a host need not reverse the channels before the upload (the same bytes as
 * reversing first: the three channels are resampled independently).  With dst == src sizes the taps are {0, 2048, 0, 0}: a copy. */
int frcnn_resize_cubic_taps(int dst, int src, int32_t* tab_h);
int frcnn_resize_cubic_u8(const uint8_t* src_hwc, int src_h, int src_w, const int32_t* tab_x, const int32_t* tab_y,
                          int dst_h, int dst_w, int flip, uint8_t* dst_hwc, void* stream);
/* rpn_util._get_all_anchor_coords (rpn_util.py:276-298): all anchors in image pixels.
 * anchor_hw_h: host [A][2] = {height, width} (util.get_anchors, util.py:242-253).
 * out: [rows*cols*A][4] f32. */
int frcnn_anchors_image(int rows, int cols, const int32_t* anchor_hw_h, int A, int stride,
                        float* out, void* stream);

/* det_util._get_anchor_coords (det_util.py:162-175): anchors in conv-cell units;
 * anchor_hw_conv_h = anchor dims already floor-divided by the stride (det_util.py:374). */
int frcnn_anchors_conv(int rows, int cols, const int32_t* anchor_hw_conv_h, int A,
                       float* out, void* stream);

/* ------------------------------------------------------------------ IoU */
/* util.cross_ious (util.py:146-177), no "+1" convention, f32 result [M][G].
 * _f32: boxes1 f32 (rpn_util.py:66).  _i16: boxes1 int16 (det_util.py:314). */
int frcnn_cross_ious_f32(const float* boxes1, int M, const float* boxes2, int G, float* out, void* stream);
int frcnn_cross_ious_i16(const int16_t* boxes1, int M, const float* boxes2, int G, float* out, void* stream);

/* ------------------------------------------------------------------ RPN targets */
/* RpnTrainingManager._process (rpn_util.py:54-103), everything before the host-RNG
 * sampling step: anchors -> IoU vs gt -> positives / negatives / out-of-bounds ->
 * regression targets.  gt: [G][4] f32 image pixels.  Outputs (N = rows*cols*A):
 * can_use[N] u8, is_pos[N] u8, bbreg[N][4] f32, argmax_gt[N] i32 (may be NULL).
 * G == 0 is allowed (no positives; every in-bounds anchor is a usable negative). */
size_t frcnn_rpn_assign_workspace_bytes(int rows, int cols, int A, int G);
int frcnn_rpn_assign(int rows, int cols, const int32_t* anchor_hw_h, int A, int stride,
                     const float* gt, int G, int img_w, int img_h,
                     uint8_t* can_use, uint8_t* is_pos, float* bbreg, int32_t* argmax_gt,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The batch sampling of rpn_y_true (rpn_util.py:106-140 over _apply_sampling, :324-350) with the host RNG kept and the tensors left on
 * the device.  frcnn_rpn_sample_lists: the ascending lists np.where(is_pos == 1 & can_use == 1) / np.where(is_pos == 0 & can_use == 1)
 * return (pos_locs / neg_locs, N int32 each, filled from the front) and counts = {num_pos, num_neg} -- the only two numbers
 * `random.sample(range(num_pos), num_pos - 128)` / `random.sample(range(num_neg), num_neg + num_pos - 256)` need.
 * frcnn_rpn_pack_targets: can_use[locs[off[j]]] = 0 for the positions the host drew (off_pos / off_neg: device int32 lists), then
 * y_class [cells][2A] = [can_use | is_pos] and y_bbreg [cells][8A] = [(is_pos & can_use) x 4 | bbreg] as float32 -- the layout
 * np.concatenate gives rpn_y_true's outputs, cast the way train_on_batch's feed casts them.  can_use is modified in place. */
int frcnn_rpn_sample_lists(const uint8_t* can_use, const uint8_t* is_pos, int n, int32_t* pos_locs, int32_t* neg_locs, int32_t* counts, void* stream);
int frcnn_rpn_pack_targets(uint8_t* can_use, const uint8_t* is_pos, const float* bbreg, int cells, int A,
                           const int32_t* pos_locs, int n_pos, const int32_t* off_pos, int n_off_pos,
                           const int32_t* neg_locs, int n_neg, const int32_t* off_neg, int n_off_neg,
                           float* y_class, float* y_bbreg, void* stream);
/* HOST function (no device call): CPython's `random.sample(range(n), k)` replayed on the interpreter's own Mersenne-Twister state
 * (`random.getstate()`: 624 words + index, both updated in place; write them back with `random.setstate`).  The same generator words
 * (MT19937 genrand_uint32), the same rejection loop (`_randbelow_with_getrandbits`) and the same selection (Lib/random.py sample():
 * use_pool != 0 is its `n <= setsize` branch, decided by the caller with the interpreter's own arithmetic): out[k] equals the list the
 * interpreter would return and the stream continues where it would.  The interpreter takes 7-80 ms for the 20 000-60 000 negative
 * anchors of an image (rpn_util.py:343-348); this takes a few hundred microseconds. */
int frcnn_host_mt_sample_range(uint32_t* mt_state, int32_t* mt_index, int n, int k, int use_pool, int32_t* out);

/* ------------------------------------------------------------------ proposals */
/* det_util._get_rois + _get_valid_box_idxs (det_util.py:370-380, 179-205) with
 * util.transform_np_inplace (util.py:111-142) inside: regr [rows][cols][4A] f32 ->
 * rois [N][4] f32 (integer valued, conv-cell units, sanitised), valid[N] u8. */
int frcnn_decode_proposals(const float* regr, int rows, int cols, const int32_t* anchor_hw_conv_h, int A,
                           float* rois, uint8_t* valid, void* stream);

/* Padded canvases: images of different true sizes in ONE pass of fixed shape (shapes.py:106-123 gives every source size its own resized
 * geometry; voc_dets.py:91-111 walks a list of them).  Image i sits at offset (oy, ox) = (h & 1, w & 1) of a canvas [hc][wc] with EVEN
 * sides, zero elsewhere: SAME padding at stride 2 under conv1's 7x7 window (resnet.py:408) puts one more zero row / column in front of an
 * odd side than of an even one, and the offset supplies it -- conv1's output cell (i, j) is the image's own cell (i, j) for every parity.
 * frcnn_preprocess_u8_canvas: frcnn_preprocess_u8 at that offset, zeros outside.  frcnn_zero_outside: x [n][hc][wc][row_bytes]:
 * zero every cell at or beyond true_hw[i] = {rows, cols} (DEVICE int32 [n][2]) -- issue it behind every layer whose output feeds a
 * convolution with taps (bias / BatchNorm shift / ReLU make the outside non-zero; the reference pads with zeros there).
 * frcnn_decode_proposals_canvas: frcnn_decode_proposals on canvas-shaped RPN outputs with the image's true map size true_rows_cols
 * (DEVICE int32 [2]): cells outside it are no candidates, boxes clip to the true extent (det_util.py:179-192). */
int frcnn_preprocess_u8_canvas(const uint8_t* img_hwc, int h, int w, int hc, int wc, int oy, int ox, const double* mean3_h, float* out, void* stream);
int frcnn_zero_outside(void* x, int n, int hc, int wc, int row_bytes, const int32_t* true_hw, void* stream);
int frcnn_decode_proposals_canvas(const float* regr, int rows_c, int cols_c, const int32_t* anchor_hw_conv_h, int A, const int32_t* true_rows_cols,
                                  float* rois, uint8_t* valid, void* stream);
/* util.transform_np_inplace (util.py:111-142) on arbitrary boxes: coords [n][4] f32 is
 * updated in place from deltas [n][4] f32. */
int frcnn_transform_inplace(float* coords, const float* deltas, int n, void* stream);

/* probs.argsort()[::-1][:K] over the valid entries (det_util.py:68-74, 148-154).
 * Tie rule (the reference's is unspecified): descending score, ascending index.
 * order[K] i32 receives the indices (entries >= *n_out are set to -1);
 * n_out (device i32) = min(K, number of valid entries).  valid may be NULL (all valid). */
size_t frcnn_topk_workspace_bytes(int N);
int frcnn_topk_order(const float* scores, const uint8_t* valid, int N, int K,
                     int32_t* order, int32_t* n_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* rois[order].astype('int16') (det_util.py:75-76, 154-155) plus the matching scores.
 * Rows >= *n are zero filled. */
int frcnn_gather_candidates(const float* rois, const float* scores, const int32_t* order, const int32_t* n, int K,
                            int16_t* cand, float* cand_scores, void* stream);

/* det_util.nms (det_util.py:209-256) on candidates ALREADY in descending score order.
 * "+1" pixel convention, keep while overlap <= thresh, stop at max_boxes.
 * n (device i32) = number of live rows (<= K <= FRCNN_NMS_MAX_BOXES).
 * keep[max_boxes] i32 receives positions into the candidate list in pick order (entries >= *n_keep are set
 * to -1: the call defines every slot, the caller need not clear the buffers), n_keep (device i32) their count.
 * _i16: int16 boxes (proposal NMS).  _f64: float64 boxes (voc_dets.py:76). */
size_t frcnn_nms_workspace_bytes(int K);
int frcnn_nms_i16(const int16_t* boxes, const int32_t* n, int K, double thresh, int max_boxes,
                  int32_t* keep, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream);
int frcnn_nms_f64(const double* boxes, const int32_t* n, int K, double thresh, int max_boxes,
                  int32_t* keep, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream);

/* boxes[pick] for the proposal path, written as the f32 RoI list the detector consumes
 * (voc_dets.py:31-47): rows [0,*n_keep) = cand[keep[k]]; rows up to the next multiple of
 * `batch` repeat row (k/batch)*batch (the reference pads its last batch with that batch's
 * first RoI); any further rows up to out_rows repeat row 0.  out: [out_rows][4] f32. */
int frcnn_gather_rois(const int16_t* cand, const int32_t* keep, const int32_t* n_keep,
                      int batch, int out_rows, float* out, void* stream);

/* ------------------------------------------------------------------ detector targets */
/* det_util._rois_to_truth (det_util.py:310-366) per RoI, before host compaction:
 * rois [E][4] int16 (conv units); gt_f32 [G][4] = GT/stride rounded to f32 (IoU input,
 * util.py:229-238); gt_f64 [G][4] = the same boxes in f64 (regression input,
 * det_util.py:349); gt_cls [G] i32.
 * eligible[E] u8 (max IoU >= .1), cls[E] i32 (GT class if max IoU >= .5 else bg_idx),
 * targets[E][4] f32 (already * [10,10,5,5]; zeros for non-positives). */
int frcnn_roi_targets(const int16_t* rois, int E, const float* gt_f32, const double* gt_f64,
                      const int32_t* gt_cls, int G, int bg_idx,
                      uint8_t* eligible, int32_t* cls, float* targets, void* stream);

/* ------------------------------------------------------------------ RoI crop + resize */
/* custom_layers.RoiResizeConv.call (custom_layers.py:35-56): for each RoI, int32-truncate
 * the corners, crop feat[y1:y2, x1:x2, :] (x2/y2 exclusive) and resize it to pool x pool
 * with TF-1.3 bilinear (align_corners=False, no half-pixel offset).
 * feat [rows][cols][C] f32, rois [n][4] f32, out [n][pool][pool][C] f32.  C % 4 == 0. */
int frcnn_roi_crop_resize_fwd(const float* feat, int rows, int cols, int C,
                              const float* rois, int n, int pool, float* out, void* stream);
/* Same resampling with two extras used by the hoisted detector head: `fill` ([c] or NULL = zeros) is
 * what an invalid RoI yields, `relu` != 0 clamps the result at 0.  A 1x1 convolution + folded
 * BatchNorm (resnet.py:351-356, 380-385: res5a_branch2a / branch1, strides (1,1)) commutes with this
 * linear resampling, so the head may apply them once to the conv4 map and resample their outputs;
 * `fill` = that layer's epilogue shift reproduces what an all-zero crop gives in the reference order.
 * layout 1 writes out[pool][pool][n][c] (position-major, frcnn_conv_desc.layout) instead of out[n][pool][pool][c]. */
int frcnn_roi_crop_resize_fwd_ex(const float* feat, int rows, int cols, int c, const float* rois, int n, int pool,
                                 const float* fill, int relu, int layout, float* out, void* stream);
/* Gradient of the above w.r.t. feat: dfeat [rows][cols][C] is OVERWRITTEN (cells no RoI touches get 0).  A gather per
 * feature cell: the taps that land on it are summed in sample order (roi, py, px; top-left, top-right, bottom-left,
 * bottom-right), the order of TF's CPU ResizeBilinearGrad -- deterministic, no atomics. */
int frcnn_roi_crop_resize_bwd(const float* dout, int rows, int cols, int C,
                              const float* rois, int n, int pool, float* dfeat, void* stream);

/* ------------------------------------------------------------------ conv engine (MFMA) */
/* One NHWC f32 convolution with a fused epilogue: the unit the Keras graphs of
 * resnet.py / vgg.py are lowered to.
 *   Conv2D(+bias) -> BatchNormalization(training=False) [-> Scale] -> [add shortcut] -> Activation
 *   (resnet.py:150-176 identity_block, :218-247 conv_block, :408-412 stem, :464-474 RPN heads,
 *    :282-313 / :351-392 TimeDistributed blocks with the RoI axis folded into n;
 *    vgg.py:96-137; Dense layers = 1x1 conv over an n x 1 x 1 x cin tensor).
 * y[m][c] = act( conv(x, w)[m][c] * scale[c] + shift[c] + residual[m][c] ),  m = (img, ho, wo).
 * Keras padding='same' = TF SAME: ho = ceil(h/stride), pad_total = max((ho-1)*stride + kh - h, 0),
 * pad_top = pad_total / 2 (extra pixel at the end); 'valid': pad 0, ho = (h - kh)/stride + 1. */
#define FRCNN_ACT_NONE 0
#define FRCNN_ACT_RELU 1
#define FRCNN_ACT_SIGMOID 2
typedef struct frcnn_conv_desc {
    int32_t n, h, w, cin;          /* input  [n][h][w][cin]                                   */
    int32_t cout, kh, kw, stride;  /* filter [kh][kw][cin][cout], same stride in h and w      */
    int32_t pad_top, pad_left;     /* implicit zero padding before the first row / column     */
    int32_t ho, wo;                /* output [n][ho][wo][cout]                                */
    int32_t act;                   /* FRCNN_ACT_*                                             */
    int32_t ldy, ldres;            /* row strides (elements) of y / residual; 0 = cout        */
    int32_t tile;                  /* 0 = auto; 1: 128x128, 2: 64x64, 3: 128x64, 4: 256x128;
                                      11..14: the same tiles with the pipelined v2 main loop;
                                      21, 22: 128x128 / 64x64 v2 with the late-LDS-store schedule;
                                      23..26: 64x64 / 64x128 / 128x64 / 128x128 with the mid-chunk-barrier schedule
                                          (what auto picks for every 64x64 launch and the 1x1 128x128 ones);
                                      41..44 (bf16 path): 128x128 (4x2 / 2x4 waves), 128x64 with 8 waves, 128x128 with 16;
                                      45..48 (bf16 path): operands staged straight into LDS (buffer_load ... lds):
                                          256x256 / 128x256 / 128x128 on 8 waves, 64x64 on 4; auto takes 47 where it took 42;
                                      50: auto for a launch that SHARES the chip with other streams' launches
                                          (several images in flight): prefers the larger tiles;
                                      61, 62: the balanced (stream-K) form of 21 / 22 when the shape
                                          allows it (needs the workspace; see frcnn_conv2d_fwd_ws);
                                      + 100*s: force s split-K slices (frcnn_conv2d_fwd_ws)       */
    int32_t layout;                /* 0: x [n][h][w][cin], y [n][ho][wo][cout] (NHWC).
                                      1: position-major, x [h][w][n][cin], y [ho][wo][n][cout] (forward only,
                                         cin % 32 == 0): the detector head keeps its RoI crops this way so a
                                         128-row tile covers one or two output positions and the filter taps
                                         that only meet zero padding are skipped (bit-identical result)     */
} frcnn_conv_desc;

/* k extent of a packed filter row: kh*kw*cin rounded up to the kernel's k-chunk (32); cin == 3 (the image
 * stems, resnet.py:408 conv1, vgg.py:96 block1_conv1) is laid out four channels wide: kh*kw*4 rounded up. */
int frcnn_conv_packed_k(int kh, int kw, int cin);
/* Keras HWIO kernel [kh][kw][cin][cout] -> packed [cout][packed_k].  The packed order is private to the
 * library (what frcnn_conv2d_fwd* read): cin % 32 == 0: k = ((c/32)*kh*kw + r*kw+s)*32 + c%32;
 * cin == 3: k = (r*kw+s)*4 + c with a zero fourth column; otherwise k = (r*kw+s)*cin + c; zero padded. */
int frcnn_pack_conv_weights(const float* w_hwio, int kh, int kw, int cin, int cout, float* packed, void* stream);
/* scale / shift / residual may be NULL (1, 0, none). */
int frcnn_conv2d_fwd(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                     const float* scale, const float* shift, const float* residual, float* y, void* stream);
/* Same as frcnn_conv2d_fwd with an extra mask [M][cout]: outputs are zeroed where mask <= 0.  Used
 * as the INPUT-GRADIENT pass of a stride-1 convolution: x := gradient w.r.t. the layer output,
 * w_packed := frcnn_pack_conv_weights_dgrad(...), residual := gradient arriving over the identity
 * shortcut, mask := the forward activation feeding this layer (fused ReLU backward).  This is what
 * Keras' train_on_batch derives for every Conv2D/ReLU/add group (train_util.py:54, 118). */
int frcnn_conv2d_fwd_masked(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                            const float* scale, const float* shift, const float* residual, const float* mask,
                            float* y, void* stream);
/* Split-K form of the same launch for small grids (ResNet stage 4, the RPN heads, the dense layers):
 * K is cut into slices, one workgroup each; the workgroup that arrives last sums the slices' f32
 * partial tiles in slice order (deterministic) and runs the fused epilogue -- still ONE launch.
 * frcnn_conv2d_workspace_bytes returns 0 when the shape does not split.  Workspace contract: its
 * first 16 KiB (arrival tickets) must be ZERO on entry and are left zero on exit, so a buffer zeroed
 * once after allocation serves any sequence of calls on ONE stream; concurrent streams / hipGraphs
 * need one workspace each.  workspace == NULL selects the plain launch.  desc.tile / 100, when
 * non-zero, forces the number of slices (100 = never split); desc.tile % 100 is the tile code.
 * Balanced form (the same workspace, tickets and contract): when a 128x128-tile launch ALONE on the chip would
 * leave >= 6 % of its last round of CU slots empty and has >= 32 k-chunks (the detector head's 460-tile
 * launches at 300 RoIs), the launch takes rounds x 512 workgroups that each run an equal share of ALL k-chunks,
 * crossing tile boundaries; tiles met by several workgroups are summed from up to four partial slots in slot
 * order by the workgroup that completes the tile's chunk count.  Deterministic; not selected for tile code 50. */
size_t frcnn_conv2d_workspace_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_fwd_ws(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                        const float* scale, const float* shift, const float* residual, const float* mask,
                        float* y, void* workspace, size_t workspace_bytes, void* stream);
/* TWO layers that read the SAME input with the same geometry in ONE launch: the output channels of the two filters
 * concatenated, [0, n1) = layer 1 -> y1 [M][n1] with activation act1, [n1, cout) = layer 2 -> y2 [M][cout - n1] with act2.
 * Replaces the pairs conv_block's `branch2a` + shortcut `branch1` (both 1x1, strides (s, s), resnet.py:218-241, the
 * TimeDistributed twins :353-386 and the hoisted res5a pair) and rpn_out_cls (sigmoid) + rpn_out_bbreg (linear)
 * (resnet.py:464-474; vgg.py:172-185): the shared input is read once and one launch boundary disappears.
 * d->cout = n1 + n2, d->act is ignored, ldy / ldres must be 0; w_packed / scale / shift are those of the concatenated
 * filter (frcnn_pack_conv_weights on np.concatenate([k1, k2], axis=3)).  Per output element the arithmetic is that of
 * the single-layer launch with the same tile and split-K choice (the k order does not depend on cout): bit-identical to
 * two frcnn_conv2d_fwd_ws calls whenever those would have made the same split-K choice (always without a workspace).
 * cin % 32 == 0, at most 32 taps.  Workspace contract as frcnn_conv2d_fwd_ws (split-K only, never the balanced form). */
size_t frcnn_conv2d_dual_workspace_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_fwd_dual(const frcnn_conv_desc* d, const float* x, const float* w_packed, const float* scale, const float* shift,
                          float* y1, int n1, int act1, float* y2, int act2,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The tile / main-loop code frcnn_conv2d_fwd_dual runs for this descriptor (frcnn_conv2d_config's codes; the two-layer
 * launch has no balanced form and only the 2x2-wave tiles): profiling tools name the kernel of a paired launch with it. */
int frcnn_conv2d_dual_config(const frcnn_conv_desc* d, int has_workspace);

/* ---- fp32 convolution on the bf16 matrix cores by EXACT three-way operand splitting ("bf16x6", csrc/conv_x6.hip).
 * Same operation, operands, layouts and epilogue as frcnn_conv2d_fwd_masked / frcnn_conv2d_fwd_dual (resnet.py:150-176, 218-247,
 * 508-533: Conv2D + BatchNormalization(training=False) [+ Scale] [+ add] + Activation) with f32 activations in and out.  Every
 * f32 operand is the exact sum of three bf16 values; the six partial products that matter are multiplied on
 * v_mfma_f32_32x32x16_bf16 (each exact in f32) and accumulated in f32: the error against fp64 is that of the native f32 MFMA
 * kernel (measured 2.2e-7 vs 2.7e-7 of sum|ab|), at up to 2.67x its matrix rate.  A different summation order: results agree
 * with frcnn_conv2d_fwd to f32 rounding, not bit for bit.  cin % 32 == 0.
 * frcnn_pack_conv_weights_x6: f32 packed filter [cout][packed_k] (frcnn_pack_conv_weights, scale folded or not) -> three
 * bf16 planes [3][cout][packed_k] (6 bytes per weight).  tile: 0 = auto, 71: 128x128 on 8 waves, 72: 256x128, 73: 128x128 on
 * 4 waves, 74: 64x64, 75: 64x128. */
int frcnn_pack_conv_weights_x6(const float* w_packed, int cout, int packed_k, void* planes_bf16, void* stream);
/* workspace (may be NULL): small grids with k >= 2048 cut K over several workgroups like frcnn_conv2d_fwd_ws does (tickets zero on
 * entry, left zero; frcnn_conv2d_x6_workspace_bytes: 0 = this shape runs unsplit). */
/* The three bf16 planes of MANY packed f32 filters in one launch (frcnn_pack_conv_weights_x6 per job): a training step calls it
 * after frcnn_refresh_packed has rewritten the trainable layers' forward and input-gradient filters.  `jobs` is a HOST array. */
typedef struct frcnn_x6_job {
    const float* w_packed;         /* [rows][kpad] f32 (frcnn_pack_conv_weights / _dgrad layout), 16-byte aligned */
    void* planes_bf16;             /* [3][rows][kpad] bf16 */
    int32_t rows, kpad;            /* kpad % 32 == 0 */
} frcnn_x6_job;
int frcnn_refresh_x6_planes(const frcnn_x6_job* jobs, int n_jobs, void* stream);
/* The f16x3 engine's twin (frcnn_pack_conv_weights_h3 per job, three launches per 48 jobs): `planes_bf16` of a job points at
 * frcnn_conv_h3_planes_bytes(rows, kpad) bytes, 16-byte aligned -- the 16-byte header (max|w|, re-measured) and the two fp16 planes. */
int frcnn_refresh_h3_planes(const frcnn_x6_job* jobs, int n_jobs, void* stream);
/* The tile code (71..77) frcnn_conv2d_fwd_x6 / frcnn_conv2d_fwd_dual_x6 (n1 > 0) run for this descriptor when no split-K workspace
 * applies: profiling tools name the kernel with it. */
int frcnn_conv2d_x6_config(const frcnn_conv_desc* d, int n1);
size_t frcnn_conv2d_x6_workspace_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_fwd_x6(const frcnn_conv_desc* d, const float* x, const void* w_planes_bf16,
                        const float* scale, const float* shift, const float* residual, const float* mask, float* y,
                        void* workspace, size_t workspace_bytes, void* stream);
int frcnn_conv2d_fwd_dual_x6(const frcnn_conv_desc* d, const float* x, const void* w_planes_bf16, const float* scale, const float* shift,
                             float* y1, int n1, int act1, float* y2, int act2, void* stream);
/* ---- which fp32 matrix path a forward launch should take (the policy behind the Python mirror's ops.f32_engine scope, for hosts in
 * any language).  prefer: the split engine the caller holds filter planes for; returns FRCNN_ENGINE_NATIVE (frcnn_conv2d_fwd_ws),
 * FRCNN_ENGINE_X6 (frcnn_conv2d_fwd_x6) or FRCNN_ENGINE_H3 (frcnn_conv2d_fwd_h3) for THIS descriptor: the split engines take the
 * launches where they measure faster than the native kernels on MI355X (cin % 32 == 0, <= 32 taps, >= 64 columns, >= 256 output tiles
 * of 64x64; or >= 128 columns with a split-K workspace at hand when the engine's own split-K form applies); an explicit tile code in
 * d->tile (71..78 / 81..88) picks its engine.  Negative on a bad argument. */
#define FRCNN_ENGINE_NATIVE 0
#define FRCNN_ENGINE_X6 1
#define FRCNN_ENGINE_H3 2
int frcnn_conv2d_engine(const frcnn_conv_desc* d, int prefer, int workspace_present);

/* ---- fp32 convolution on the fp16 matrix cores by a TWO-way operand split with a scaled low part ("f16x3", csrc/conv_h3.hip).
 * Same operation, operands, layouts and epilogue as frcnn_conv2d_fwd_x6 (resnet.py:150-176, 218-247, 508-533: Conv2D +
 * BatchNormalization(training=False) [+ Scale] [+ add] + Activation), f32 activations in and out.  Each operand tensor is scaled by
 * one power of two into fp16's range and every value split into ah = f16(a') and al = f16((a' - ah) * 2^11) (23-24 of its 24 bits);
 * ah*bh and ah*bl + al*bh are accumulated in f32 on v_mfma_f32_32x32x16_f16 in two accumulators and recombined: THREE matrix
 * instructions per block of products where frcnn_conv2d_fwd_x6 needs six.  Error against fp64 at or under the native f32 MFMA's on
 * every operand class measured (7e-8 of sum|ab| on mixed-sign operands against 2.8e-7; exact on integers below 2048); not bit-equal
 * to either other engine (another summation order).  cin % 32 == 0, at most 32 taps.
 *
 * Magnitude records: the activation scale comes from an UPPER BOUND of max|x| kept on the device in a record of
 * frcnn_amax_record_floats() floats (several atomic slots; a reader takes their maximum; 16-byte aligned).  frcnn_amax_clear zeroes
 * records (a kernel, safe inside a captured graph); every frcnn_conv2d_fwd_h3 / _dual_h3 / frcnn_conv2d_fwd_ws_amax launch folds
 * max|y| of what it stores into y_amax (NULL: not tracked), so a chain of layers carries its bounds along without extra passes;
 * frcnn_amax_f32 measures a tensor that has no producer record; frcnn_amax_merge(dst, src, floor, exponent_out) sets dst = max(dst, src, floor)
 * (and, when exponent_out is given, writes the power of two planes under that bound are scaled by: frcnn_roi_crop_resize_fwd_planes)
 * for tensors derived by maps that cannot exceed max(|input|, floor): max-pooling (resnet.py:412), the bilinear RoI resampling
 * with a fill vector (custom_layers.py:35-56), ReLU.  A bound that is too LARGE by up to 2^8 costs no precision; one that is too
 * small overflows fp16 (inf / NaN in the output, as an f32 overflow would give).
 * frcnn_pack_conv_weights_h3: f32 packed filter [cout][packed_k] -> 16-byte header (max|w|) + two fp16 planes [2][cout][packed_k]
 * (frcnn_conv_h3_planes_bytes bytes, 16-byte aligned).  tile: 0 = auto, 81: 128x128 on 8 waves, 82 / 86: 256x128 on 8 / 16 waves (85: 86 without
 * the three-stage ring that plane-input launches with long reductions walk: same bits, the comparand of the ring's bitwise test)
 * with two LDS buffers, 83: 128x128 on 4 waves, 84: 64x64, 87: 128x64.  Workspace (may be NULL): the split-K contract of
 * frcnn_conv2d_fwd_x6 (frcnn_conv2d_h3_workspace_bytes: 0 = this shape runs unsplit). */
size_t frcnn_conv_h3_planes_bytes(int cout, int packed_k);
int frcnn_pack_conv_weights_h3(const float* w_packed, int cout, int packed_k, void* planes_f16, void* stream);
int frcnn_amax_record_floats(void);
int frcnn_amax_clear(float* records, int n_records, void* stream);
int frcnn_amax_f32(const float* x, size_t n, float* record, void* stream);
int frcnn_amax_merge(float* dst_record, const float* src_record, float floor_value, int32_t* exponent_out, void* stream);
/* Status word of a magnitude record (float index 1 of the record, an uint32 bit set; zeroed with the record by frcnn_amax_clear).  A
 * launch that scales a tensor into fp16's range under a record ORs in
 *   FRCNN_H3_UNDER      a value at or above 2^15 after scaling went by: the record was not an upper bound (stale, or a producer's bug);
 *   FRCNN_H3_SATURATED  ... and at fp16's largest finite value: it was clamped to +-65504 (the kernels run with MODE.FP16_OVFL set), so
 *                       the output is finite but WRONG where this bit is set;
 *   FRCNN_H3_NONFINITE  an infinity went by, or the record itself is not finite (the tensor it was measured on holds an infinity: one
 *                       power of two cannot serve such a tensor).
 * Any bit set: the launch's results are not to be used (the native kernels carry +-Inf / NaN through the receptive fields; this engine
 * reports instead -- measured on gfx950: a NaN or Inf element sets all three bits, tests/test_h3_fences_gpu.py).
 * frcnn_amax_status ORs the status words of `n_records` consecutive records (a pass's arena) into *out (device int32, overwritten):
 * issue it behind the pass and read the word with the outputs; non-zero = the pass's f16x3 results are not to be trusted. */
#define FRCNN_H3_UNDER 1
#define FRCNN_H3_SATURATED 2
#define FRCNN_H3_NONFINITE 4
int frcnn_amax_status(const float* records, int n_records, int32_t* out, void* stream);
int frcnn_conv2d_h3_config(const frcnn_conv_desc* d, int n1);
size_t frcnn_conv2d_h3_workspace_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_fwd_h3(const frcnn_conv_desc* d, const float* x, const float* x_amax, const void* w_planes_f16,
                        const float* scale, const float* shift, const float* residual, const float* mask, float* y, float* y_amax,
                        void* workspace, size_t workspace_bytes, void* stream);
int frcnn_conv2d_fwd_dual_h3(const frcnn_conv_desc* d, const float* x, const float* x_amax, const void* w_planes_f16,
                             const float* scale, const float* shift, float* y1, int n1, int act1, float* y1_amax,
                             float* y2, int act2, float* y2_amax, void* stream);
/* The ResNet stem as ONE f16x3 launch: conv1 7x7 / 2 'same' (3 -> 64) + BatchNormalization (+ Scale) + ReLU + MaxPooling2D((3,3),
 * strides (2,2)) (resnet.py:408-412, :565-568), f32 image in, f32 pooled map [n][hp][wp][64] out (hp = ((h+1)/2 - 3)/2 + 1), max|y| into
 * y_amax.  The fp32 twin of frcnn_stem_bf16_fwd: per conv pixel the arithmetic of frcnn_conv2d_fwd_h3 (the image split into two fp16
 * planes in LDS under the scale its magnitude record x_amax gives), the pool taken in LDS on f32 values; replaces a
 * frcnn_conv2d_fwd_ws + frcnn_pool2d_fwd pair whose 38 MB intermediate map does not stay in L2 with several images in flight.
 * frcnn_pack_stem_weights_h3: HWIO [7][7][3][64] f32 -> 16-byte header (max|w|) + two fp16 planes (frcnn_stem_h3_packed_bytes). */
size_t frcnn_stem_h3_packed_bytes(void);
int frcnn_pack_stem_weights_h3(const float* w_hwio, void* packed, void* stream);
int frcnn_stem_h3_fwd(const float* x, const float* x_amax, int n, int h, int w, const void* w_packed, const float* scale, const float* shift,
                      float* out, float* y_amax, void* stream);
/* A chain of f16x3 layers may hand its activations on ALREADY split: `y_planes` makes the launch write two fp16 planes [2][M][cout]
 * (hi, lo as the engine splits them) under the scale 2^*exponent beside -- or, with y == NULL, instead of -- the f32 tensor, and a
 * following launch given them as `x_planes` (x == NULL) stages them into LDS unchanged: no conversion and no arithmetic in its loader
 * (the detector head's 14 700-row GEMMs, resnet.py:508-533: 195 / 103 / 89 us instead of 246 / 127 / 104).  The producer cannot know
 * max|y| before it has finished, so it scales by a bound every workgroup derives alike: |y| <= bound_c * max|x| + bound_d (+ max|residual|),
 * bound_c = max over output channels of |scale[c]| * sum |w[.,.,.,c]|, bound_d = max |shift[c]| (host constants of the filter) -- it cannot
 * overflow whatever the data, and it is loose by about sqrt(K) (2^4..2^6 for these layers), which costs nothing (see above).  The
 * 256x128 tile forms only (frcnn_conv2d_h3_config 86 / 82), dense single layers, no mask; x_amax is always required; residual needs its
 * record when planes are written. */
typedef struct frcnn_h3_planes {
    void* planes;                  /* [2][rows][channels] fp16, 16-byte aligned */
    int32_t* exponent;             /* device int32: stored value = true value * 2^exponent (written by the producing launch) */
    int32_t* status;               /* sticky status word a launch WRITING these planes ORs FRCNN_H3_* bits into (may be NULL): word 1 of the
                                    * tensor's magnitude record, so that frcnn_amax_status finds it */
} frcnn_h3_planes;
int frcnn_conv2d_fwd_h3_planes(const frcnn_conv_desc* d, const float* x, const frcnn_h3_planes* x_planes, const float* x_amax,
                               const void* w_planes_f16, const float* scale, const float* shift,
                               const float* residual, const float* residual_amax,
                               float* y, float* y_amax, const frcnn_h3_planes* y_planes, float bound_c, float bound_d, void* stream);
/* The same launch with the RESIDUAL handed over as planes (residual == NULL then): a bottleneck block of the detector head whose output
 * has two readers -- the next block's branch2a and its shortcut, resnet.py:282-313 -- writes it ONCE, as planes only (y == NULL); branch2a
 * stages them unchanged and the closing 1x1 of that block reads the shortcut back from them, value = (hi + lo / 2048) * 2^-exponent:
 * the tensor's elements to 22-24 bits, i.e. what the engine multiplies anyway.  Same bytes as the f32 tensor, and the 2048 -> 512 layers
 * behind it lose the split in their loader (and, 64 chunks long, walk the three-stage ring). */
int frcnn_conv2d_fwd_h3_planes_res(const frcnn_conv_desc* d, const float* x, const frcnn_h3_planes* x_planes, const float* x_amax,
                                   const void* w_planes_f16, const float* scale, const float* shift,
                                   const float* residual, const frcnn_h3_planes* residual_planes, const float* residual_amax,
                                   float* y, float* y_amax, const frcnn_h3_planes* y_planes, float bound_c, float bound_d, void* stream);
/* RoiResizeConv (custom_layers.py:35-56; frcnn_roi_crop_resize_fwd_ex) writing its result as f16x3 planes for the convolution behind it
 * (res5a_branch2b's 3x3 over the crops, resnet.py:508-512).  out->exponent is an INPUT here: the scale comes from a bound known before the
 * launch -- a bilinear sample cannot exceed the map's largest magnitude, a rejected RoI yields `fill` -- i.e. from
 * frcnn_amax_merge(record, map_record, max|fill|, out->exponent) issued in front of it on the same stream. */
int frcnn_roi_crop_resize_fwd_planes(const float* feat, int rows, int cols, int c, const float* rois, int n, int pool,
                                     const float* fill, int relu, int layout, const frcnn_h3_planes* out, void* stream);
/* The RoIs of a BATCH of images in one launch (fp32 twin of frcnn_roi_crop_resize_fwd_bf16_batch): `feat` holds one (rows, cols, C) map per
 * image, RoI r crops the map of image r / n_per_img.  Exactly one of `out` (f32, as frcnn_roi_crop_resize_fwd_ex) and `planes_out` (two fp16
 * planes under *exponent, as frcnn_roi_crop_resize_fwd_planes: the exponent of the bound over ALL the maps) is given.  Per RoI the
 * arithmetic is the single-image call's. */
int frcnn_roi_crop_resize_fwd_batch(const float* feat, int rows, int cols, int C, const float* rois, int n, int n_per_img, int pool,
                                    const float* fill, int relu, int layout, float* out, const frcnn_h3_planes* planes_out, void* stream);
/* frcnn_conv2d_fwd_ws (the native f32 MFMA kernels) that also folds max|y| into y_amax: a layer that stays on the native path
 * (the 3-channel stem, small grids) in front of an f16x3 layer. */
int frcnn_conv2d_fwd_ws_amax(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                             const float* scale, const float* shift, const float* residual, const float* mask, float* y, float* y_amax,
                             void* workspace, size_t workspace_bytes, void* stream);
/* Filter of the input-gradient convolution: transposed (cin <-> cout), flipped in both taps, input
 * channel co scaled by scale[co] (the forward epilogue scale = folded BatchNorm; NULL = 1).
 * packed: [cin][frcnn_conv_packed_k(kh, kw, cout)]. */
int frcnn_pack_conv_weights_dgrad(const float* w_hwio, const float* scale, int kh, int kw, int cin, int cout,
                                  float* packed, void* stream);
/* After an optimiser step: re-derive the device forms of MANY trainable layers from their fp32
 * master weights in ONE launch (Keras keeps one variable per layer and needs no such step; here
 * the conv engine reads packed filters).  `jobs` is a HOST array (it rides in the kernel arguments).
 * Per job: packed := frcnn_pack_conv_weights(w_hwio), packed_dgrad := frcnn_pack_conv_weights_dgrad
 * (w_hwio, scale), shift[n] := bias[n]*scale[n] + shift_const[n]; NULL outputs are skipped, NULL
 * bias / scale / shift_const read as 0 / 1 / 0. */
typedef struct frcnn_pack_job {
    const float* w_hwio;
    float* packed;
    float* packed_dgrad;
    const float* bias;
    const float* scale;
    const float* shift_const;
    float* shift;
    int32_t kh, kw, cin, cout;
} frcnn_pack_job;
int frcnn_refresh_packed(const frcnn_pack_job* jobs, int n_jobs, void* stream);
/* Bias gradients of many layers in one launch: out[co] = scale[co] * sum_m g[m][co] (g [m][cout],
 * scale may be NULL).  `jobs` is a HOST array.  Fixed summation order (reproducible). */
typedef struct frcnn_colsum_job {
    const void* g;                 /* f32, or bf16 when g_is_bf16 != 0 */
    const float* scale;
    float* out;
    int32_t m, cout;
    int32_t g_is_bf16;
    int32_t reserved;
} frcnn_colsum_job;
int frcnn_colsum_batch(const frcnn_colsum_job* jobs, int n_jobs, void* stream);
/* Weight / bias gradient of the convolution described by d (forward geometry):
 *   dw[kh][kw][cin][cout] = scale[co] * sum_m im2col(x)[m][(tap,ci)] * g[m][co],  dbias[co] = scale[co] * sum_m g[m][co]
 * g [M][cout] = gradient w.r.t. the layer's post-BatchNorm, pre-activation output.  Deterministic
 * (slice partials reduced in a fixed order).  dbias may be NULL.
 * d->tile % 100 in 71..77 asks for the split-bf16 engine (round 4): on layers with cin, cout >= 128 both f32 operands are split
 * exactly into three bf16 pieces inside the kernel and the six partial products that matter run on v_mfma_f32_32x32x16_bf16
 * (same tiles, slabs and fixed-order reduction; error against fp64 at the native kernel's level, results agree with the
 * native engine to f32 rounding, not bit for bit); other layers ignore the request.  Also honoured per job by
 * frcnn_conv2d_wgrad_batch. */
size_t frcnn_conv2d_wgrad_workspace_bytes(const frcnn_conv_desc* d);
int frcnn_conv2d_wgrad(const frcnn_conv_desc* d, const float* x, const float* g, const float* scale,
                       float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* The weight gradients of MANY layers in one go (what Keras' train_on_batch derives for every trainable Conv2D / Dense
 * of the model, train_util.py:54, 118): one launch per operand kind (f32; bf16 on the bf16 matrix cores; bf16 widened
 * for channel counts that are not multiples of 8) over all jobs' workgroups, then one launch for all slice reductions.
 * Per job exactly frcnn_conv2d_wgrad / frcnn_conv2d_wgrad_bf16 without the bias gradient (frcnn_colsum_batch does
 * those): same slices, same fixed summation order, bit-identical dw.  `jobs` is a HOST array; x / g: f32, or bf16
 * when in_bf16 != 0.  The workspace (frcnn_conv2d_wgrad_batch_workspace_bytes) holds every job's partial slabs. */
typedef struct frcnn_wgrad_job {
    frcnn_conv_desc d;             /* forward geometry of the layer */
    const void* x;                 /* layer input  [n][h][w][cin]            */
    const void* g;                 /* gradient w.r.t. the post-BatchNorm, pre-activation output [M][cout] */
    const float* scale;            /* folded BatchNorm scale [cout] or NULL  */
    float* dw;                     /* out: [kh][kw][cin][cout] f32           */
    int32_t in_bf16;
    int32_t reserved;
} frcnn_wgrad_job;
size_t frcnn_conv2d_wgrad_batch_workspace_bytes(const frcnn_wgrad_job* jobs, int n_jobs);
int frcnn_conv2d_wgrad_batch(const frcnn_wgrad_job* jobs, int n_jobs, void* workspace, size_t workspace_bytes, void* stream);
/* The tile code (see frcnn_conv_desc.tile) frcnn_conv2d_fwd will run for this descriptor (30 = the 3-channel
 * stem kernel, which cin == 3 always takes): lets a profiler attribute a launch to its kernel instantiation. */
int frcnn_conv2d_config(const frcnn_conv_desc* d);
/* MaxPooling2D / AveragePooling2D, 'valid' (resnet.py:412, 515; vgg.py:100-128). c % 4 == 0. */
int frcnn_pool2d_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int is_max, float* y, void* stream);
/* The same pooling with the result written as the f16x3 engine's fp16 planes for the convolution behind it (VGG's block<n>_conv1 after a
 * max-pool, vgg.py:100-128).  out->exponent is an INPUT, as for frcnn_roi_crop_resize_fwd_planes: a window's maximum / mean cannot exceed the
 * largest |input|, so the scale comes from frcnn_amax_merge(record, input_record, 0, out->exponent) issued in front on the same stream. */
int frcnn_pool2d_fwd_planes(const float* x, int n, int h, int w, int c, int k, int stride, int is_max, const frcnn_h3_planes* out, void* stream);
/* AveragePooling2D over all `npos` positions of a position-major tensor x[npos][n][c] -> y[n][c]
 * (resnet.py:515 on frcnn_conv_desc.layout == 1 tensors; same additions and division as frcnn_pool2d_fwd). */
int frcnn_avgpool_pos_major(const float* x, int npos, int n, int c, float* y, void* stream);
/* softmax over the first `cols` entries of each row (Dense(activation='softmax'), resnet.py:522). */
int frcnn_softmax_rows(const float* x, int rows, int cols, int ldx, float* y, int ldy, void* stream);
/* Output rows of the detector's two dense heads computed as ONE GEMM (kernels concatenated): cls[r][0..cols) =
 * softmax(x[r][0..cols)) (dense_class_C, resnet.py:522-527), reg[r][0..tail) = x[r][cols..cols+tail) (dense_reg_C,
 * :528-533; vgg.py:241-247).  Same arithmetic as frcnn_softmax_rows; both outputs dense. */
int frcnn_dense_heads_split(const float* x, int rows, int cols, int tail, int ldx, float* cls, float* reg, void* stream);

/* ------------------------------------------------------------------ detections */
/* voc_dets.get_dets, everything after detector.predict (voc_dets.py:51-88): per scored RoI
 * arg-max class and confidence, skip background / f64(confidence) < det_threshold (the reference's pinned numpy compares an
 * np.float32 scalar with a Python float in f64), decode the class's
 * regression with util.transform (util.py:55-74) x stride, per-class det_util.nms(thresh, 2000)
 * on the float boxes, divide by resize_ratio and round half-to-even.
 * rois [max_rows][4] f32 (conv units), *n_rois live rows (<= max_rows <= 512),
 * out_cls [max_rows][C] f32, out_reg [max_rows][4(C-1)] f32.
 * Outputs, in the reference's emission order (classes in first-seen order, NMS pick order
 * inside a class): det_cls[max_rows] i32, det_prob[max_rows] f32, det_bbox[max_rows][4] i32,
 * det_roi[max_rows] i32 (source RoI row), *n_dets; rows >= *n_dets are written too (det_cls = det_roi = -1,
 * det_prob = 0, det_bbox = 0), so the caller need not clear the buffers.  The reference's padded duplicate RoIs
 * (voc_dets.py:42-46) are suppressed by its own NMS (IoU 1), so only live rows are scored. */
int frcnn_detections(const float* rois, const int32_t* n_rois, int max_rows, const float* out_cls, const float* out_reg,
                     int num_classes, int bg_idx, double det_threshold, double stride, double resize_ratio, double nms_thresh,
                     int32_t* det_cls, float* det_prob, int32_t* det_bbox, int32_t* det_roi, int32_t* n_dets, void* stream);
/* The same post-process for a CAPTURED pass that serves every image of one size (voc_dets.get_dets_by_cls, voc_dets.py:91-111,
 * walks a list of images whose resize ratios differ): the two per-image scalars are read from DEVICE memory,
 * dyn[0] = resize_ratio, dyn[1] = det_threshold (f64, 8-byte aligned; the comparison `confidence < det_threshold` is made in
 * f64, numpy 1.13's promotion of an np.float32 scalar against a Python float, voc_dets.py:57).  roi_batch > 0: the rows
 * scored are the reference's PADDED list -- *n_rois rounded up to a multiple of roi_batch, the fill rows being the copies of
 * the last batch's first RoI that frcnn_gather_rois laid out (voc_dets.py:42-46, :51) -- so det_roi indexes that list;
 * roi_batch = 0 scores the live rows only.  counts[0] = detections emitted, counts[1] = *n_rois (the "num rois" line,
 * voc_dets.py:26): both reach the host in the one copy that carries the detections. */
int frcnn_detections_dyn(const float* rois, const int32_t* n_rois, int roi_batch, int max_rows, const float* out_cls, const float* out_reg,
                         int num_classes, int bg_idx, double stride, double nms_thresh, const double* dyn,
                         int32_t* det_cls, float* det_prob, int32_t* det_bbox, int32_t* det_roi, int32_t* counts, void* stream);

/* ------------------------------------------------------------------ training: losses, optimisers */
/* The reference's four Keras loss functions (loss_functions.py:15-76) as Keras 2.0.8 evaluates them
 * inside train_on_batch (train_util.py:54, 118): value into *loss (device f32) and, when the grad
 * pointer is not NULL, the gradient of that loss w.r.t. the network output it is fed with.
 *   rpn_cls: y_true [cells][2A] f32 = [can_use | is_pos], y_pred [cells][A] sigmoid outputs;
 *            grad is w.r.t. the PRE-sigmoid logits.  loss = sum(sel*BCE)/256 (:24).
 *   rpn_reg: y_true [cells][8A] = [mask | targets], y_pred [cells][4A];
 *            loss = mean(mask)*10*S/2400 with S over ALL anchors (the reference's quirk, :44).
 *   det_cls: y_true/y_pred [n][C]; grad w.r.t. the PRE-softmax logits, written with row stride ldg (:76).
 *   det_reg: y_true [n][8K] = [mask | targets], y_pred [n][4K], K = classes excl. background (:65). */
int frcnn_loss_rpn_cls(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_logit, void* stream);
int frcnn_loss_rpn_reg(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_pred, void* stream);
/* The same two losses spread over many workgroups (what the training step calls): per-workgroup f64 partials in
 * `workspace` (frcnn_loss_workspace_bytes(), private to the call until the stream has passed it) summed in index order --
 * reproducible; equal to the one-workgroup forms up to the last ulp of the f64 sums. */
size_t frcnn_loss_workspace_bytes(void);
int frcnn_loss_rpn_cls_ws(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_logit, void* workspace, void* stream);
int frcnn_loss_rpn_reg_ws(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_pred, void* workspace, void* stream);
int frcnn_loss_det_cls(const float* y_true, const float* y_pred, int n_rois, int C, float* loss, float* grad_logit, int ldg, void* stream);
int frcnn_loss_det_reg(const float* y_true, const float* y_pred, int n_rois, int num_classes_excl_bg, float* loss, float* grad_pred, int ldg, void* stream);
/* g *= (y > 0): ReLU backward where no conv epilogue can carry it.  n % 4 == 0. */
int frcnn_relu_bwd_inplace(float* g, const float* y, size_t n, void* stream);
/* Backward of AveragePooling2D(k) on a k x k map (resnet.py:515) fused with the ReLU in front of it:
 * gx[n][k][k][c] = (y > 0) * g_pooled[n][c] / k^2. */
int frcnn_avgpool_bwd_masked(const float* g_pooled, const float* y, int n, int k, int c, float* gx, void* stream);
/* Backward of MaxPooling2D((k,k), strides=(k,k)) (vgg.py:100-128): x [n][h][w][c] the pool input, y its
 * output [n][h/k][w/k][c], gy the gradient w.r.t. y; the gradient goes to the first maximum of each window. */
int frcnn_maxpool_bwd(const float* x, const float* y, const float* gy, int n, int h, int w, int c, int k, float* gx, void* stream);
/* Keras optimisers (args_util.py:48-59) over a flat parameter buffer; l2 = the regulariser factor of
 * resnet.py:26-27 (its gradient 2*l2*w is added here), grad_scale = 1/world_size after an all-reduce sum. */
int frcnn_sgd_momentum(float* w, const float* g, float* v, size_t n, float lr, float momentum, float l2, float grad_scale, void* stream);
int frcnn_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int t,
               float l2, float grad_scale, void* stream);
/* Epilogue shift of a layer whose conv bias trains while its BatchNorm is frozen (resnet.py:150-153):
 * out[c] = bias[c]*scale[c] + shift_const[c] (NULL bias = 0, NULL scale = 1, NULL shift_const = 0). */
int frcnn_fold_bias(const float* bias, const float* scale, const float* shift_const, float* out, int n, void* stream);
/* *out = sum(w^2): the value of the l2 regularisation term is l2 * that. */
size_t frcnn_sumsq_workspace_bytes(void);
int frcnn_sumsq(const float* w, size_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ bf16 conv path */
/* BASELINE configs[3] ("bf16 conv + fp32 NMS"): the same implicit-GEMM convolution with bf16
 * activations/filters on v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 scale/shift, one RNE rounding
 * at the store.  cin % 64 == 0 (the 3-channel stem stays on the f32 kernel).  Layouts as the f32 path;
 * packed filter row length = kh*kw*cin, k = ((c/64)*kh*kw + tap)*64 + c%64. */
/* The ResNet stem in ONE launch on the bf16 matrix cores: conv1 7x7 / 2 'same' (3 -> 64) + folded BatchNorm (+ Scale) + ReLU +
 * MaxPooling2D(3x3, strides 2, VALID) + the bf16 store that opens the bf16 trunk (resnet.py:408-412; resnet101: :565-568).
 * x [n][h][w][3] f32 (preprocessed image, rounded to bf16 once inside), packed filter from frcnn_pack_stem_weights_bf16
 * (HWIO f32 [7][7][3][64] -> bf16 [64][176], k = r*24 + s*3 + c), scale / shift [64] f32 (folded bias + BatchNorm + Scale);
 * out [n][hp][wp][64] bf16 with hp = ((h+1)/2 - 3)/2 + 1.  f32 accumulate, f32 epilogue, one rounding; the pool runs on the
 * rounded values (rounding is monotonic: same bits as pooling first). */
int frcnn_stem_bf16_packed_elems(void);
int frcnn_pack_stem_weights_bf16(const float* w_hwio, void* packed_bf16, void* stream);
int frcnn_stem_bf16_fwd(const float* x, int n, int h, int w, const void* w_packed_bf16, const float* scale, const float* shift,
                        void* out_bf16, void* stream);
int frcnn_conv_packed_k_bf16(int kh, int kw, int cin);
int frcnn_pack_conv_weights_bf16(const float* w_hwio, int kh, int kw, int cin, int cout, void* packed_bf16, void* stream);
/* y is bf16 [M][cout], or f32 when y_is_f32 != 0 (network outputs: RPN scores/regressions). */
int frcnn_conv2d_fwd_bf16(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                          const float* scale, const float* shift, const void* residual_bf16, void* y, int y_is_f32, void* stream);
/* Split-K form (same workspace contract as frcnn_conv2d_fwd_ws; 0 bytes = the shape does not split). */
size_t frcnn_conv2d_workspace_bytes_bf16(const frcnn_conv_desc* d);
int frcnn_conv2d_fwd_bf16_ws(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                             const float* scale, const float* shift, const void* residual_bf16, void* y, int y_is_f32,
                             void* workspace, size_t workspace_bytes, void* stream);
/* Same with a mask [M][cout] (bf16): outputs are zeroed where mask <= 0.  The input-gradient pass of a
 * stride-1 bf16 layer, exactly as frcnn_conv2d_fwd_masked is for f32 (mixed-precision training, BASELINE
 * configs[4]): x := bf16 gradient, w := frcnn_refresh_packed_bf16's packed_dgrad, residual := gradient arriving
 * over the identity shortcut, mask := the forward activation in front of the layer. */
int frcnn_conv2d_fwd_bf16_masked(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                                 const float* scale, const float* shift, const void* residual_bf16, const void* mask_bf16,
                                 void* y, int y_is_f32, void* workspace, size_t workspace_bytes, void* stream);
/* frcnn_refresh_packed for layers that run in bf16: `packed` / `packed_dgrad` of each job point at bf16
 * storage ([cout][kh*kw*cin] and [cin][kh*kw*cout], 64-channel k-chunks); masters, bias, scale and the
 * folded shift stay f32.  cin (and cout, when packed_dgrad is set) must be multiples of 64. */
int frcnn_refresh_packed_bf16(const frcnn_pack_job* jobs, int n_jobs, void* stream);
/* Weight gradient from bf16 activations / gradients: as frcnn_conv2d_wgrad with x and g in bf16; the
 * products are accumulated in f32 and dw / dbias are f32 (they feed the f32 master weights). */
int frcnn_conv2d_wgrad_bf16(const frcnn_conv_desc* d, const void* x_bf16, const void* g_bf16, const float* scale,
                            float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* small bf16 pieces of the backward pass */
int frcnn_cast_bf16_to_f32(const void* x_bf16, size_t n, float* y, void* stream);
int frcnn_relu_bwd_inplace_bf16(void* g_bf16, const void* y_bf16, size_t n, void* stream);
int frcnn_avgpool_bwd_masked_bf16(const float* g_pooled, const void* y_bf16, int n, int k, int c, void* gx_bf16, void* stream);
int frcnn_roi_crop_resize_bwd_bf16(const void* dout_bf16, int rows, int cols, int c, const float* rois, int n, int pool,
                                   float* dfeat, void* stream);
int frcnn_cast_f32_to_bf16(const float* x, size_t n, void* y_bf16, void* stream);
/* AveragePooling2D(k) of a k x k bf16 map -> f32 [n][c] (resnet.py:515). */
int frcnn_avgpool_bf16_to_f32(const void* x_bf16, int n, int k, int c, float* y, void* stream);
int frcnn_avgpool_bf16_to_f32_ex(const void* x_bf16, int n, int k, int c, int layout, float* y, void* stream);
/* frcnn_roi_crop_resize_fwd on a bf16 feature map (f32 interpolation, bf16 result). */
int frcnn_roi_crop_resize_fwd_bf16(const void* feat_bf16, int rows, int cols, int C, const float* rois, int n, int pool,
                                   void* out_bf16, void* stream);
int frcnn_roi_crop_resize_fwd_bf16_ex(const void* feat_bf16, int rows, int cols, int c, const float* rois, int n, int pool,
                                      const float* fill, int relu, int layout, void* out_bf16, void* stream);
/* The same for the RoIs of n_img images in ONE launch (batched inference: the detector head then makes one GEMM pass over
 * every image's RoIs): feat [n_img][rows][cols][C] bf16, rois [n_img * n_per_img][4] f32 (RoI r reads image r / n_per_img),
 * out over all RoIs in the chosen layout.  C % 8 == 0.  Per RoI bit-identical to frcnn_roi_crop_resize_fwd_bf16_ex
 * (custom_layers.py:35-56). */
int frcnn_roi_crop_resize_fwd_bf16_batch(const void* feat_bf16, int n_img, int rows, int cols, int C, const float* rois, int n_per_img, int pool,
                                         const float* fill, int relu, int layout, void* out_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */

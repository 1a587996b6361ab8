/* libyolo_hip.so -- C ABI of the MI355X (gfx950) YOLO inference hot path.
 *
 * Drop-in boundary for the reference's "process -> runtime" call (`sess.run([...], feed_dict)`:
 * V3/YOLO_V3_inference.py:106-107, V3/convert_ckpt_and_inference.py:88, V2/YOLO_v2.py:55,
 * D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:588) and, on the C side, for the entry points the
 * reference's own ctypes binding uses (D2T/darknet.py:48-115 over D2T/include/darknet.h):
 *
 *   reference                                              this library
 *   ---------------------------------------------------    ---------------------------------------
 *   load_network(cfg, weights, clear)   DN/network.c:53     yolo_create + yolo_load_darknet_weights
 *   load_weights(var_list, file)        V3/yolo_v3.py:270   yolo_load_darknet_weights / yolo_set_weights
 *   network_predict(net, float*)        DN/network.c:497    yolo_forward
 *   sess.run(detections)                V3/yolo_v3.py:266   yolo_forward(..., detections_out)
 *   get_network_boxes + do_nms_sort     DN/network.c:562,   yolo_postprocess / yolo_detect
 *     / tf.image.non_max_suppression    DN/box.c:58, V3/YOLOV3.py:364-379
 *   free_network                        DN/network.c:716    yolo_destroy
 *   error(): perror+exit                DN/utils.c:253      int return code + yolo_last_error()
 *
 * Conventions: extern "C", plain pointers and sizes, caller-owned output buffers (no callee malloc
 * on the hot path, unlike make_network_boxes DN/network.c:526-540), every call returns YOLO_OK or a
 * negative code and never exits the process.  A context is single-threaded and bound to one HIP
 * device + stream; one context per GPU.  Activations are NHWC; images are NHWC RGB.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yolo_ctx yolo_ctx;

enum yolo_status {
    YOLO_OK = 0,
    YOLO_ERR_INVALID = -1,     /* bad argument / malformed cfg */
    YOLO_ERR_HIP = -2,         /* HIP runtime error (text in yolo_last_error) */
    YOLO_ERR_STATE = -3,       /* call order (e.g. forward before weights) */
    YOLO_ERR_NOMEM = -4,
    YOLO_ERR_IO = -5,          /* weights file missing / wrong length */
    YOLO_ERR_UNSUPPORTED = -6
};

/* storage/compute type of the conv stack.  YOLO_FP8 (BASELINE config 5): OCP e4m3 filters and activations on the
 * fp8 MFMA (fp32 accumulation, heads in fp32, first conv in bf16); scheme and scales: DESIGN.md "fp8 scheme". */
enum yolo_dtype { YOLO_BF16 = 0, YOLO_FP32 = 1, YOLO_FP8 = 2,
                  YOLO_FP16 = 3,    /* IEEE fp16 filters and activations on v_mfma_f32_16x16x32_f16: the bf16 configuration's kernels,
                                     * tile plans and MFMA rate with an 11-bit significand instead of 8 (values saturate at +-65504);
                                     * what it buys in box accuracy: DESIGN.md section 4 */
                  YOLO_FP16X2 = 4 };/* split fp16 (round 4): every filter and stored activation is a PAIR of fp16 numbers hi = f16(v),
                                     * lo = f16(v - hi) -- 22 significant bits -- and a conv forms W_hi x_hi + W_hi x_lo + W_lo x_hi on the
                                     * fp16 MFMA with fp32 accumulation (three products per algorithmic one, W_lo x_lo is dropped).  The
                                     * configuration that meets north_star's IoU >= 0.999 on weights with a trained file's statistics at
                                     * several times the exact-fp32 path's rate; shortcuts are folded into the conv epilogues, the stem /
                                     * residual-block / 1x1-tail fusions are off on tensors stored as pairs; [connected] / [local] layers are
                                     * not served.  Mixed plans (round 5): `yolo_pair=0` on a [convolutional] section stores that layer's
                                     * output -- and what is derived from it without arithmetic -- as PLAIN fp16 (`yolo_pair_input=0` in
                                     * [net]: the image); a conv that reads a plain tensor runs one MFMA product per algorithmic one, and
                                     * stretches of plain tensors are served by the fp16 configuration's fused kernels.  Both operands of
                                     * a shortcut and all inputs of a concatenation share one form.  DESIGN.md section 3.6 */
enum yolo_semantics { YOLO_SEM_TF = 0, YOLO_SEM_DARKNET = 1 };
/* TF: bilinear `_upsample` (V3/yolo_v3.py:162-192) + tf.space_to_depth (V2/model_darknet19_slim.py:44);
 * DARKNET: nearest upsample (DN/blas.c:334) + reorg_cpu (DN/blas.c:9) -- lets the whole network be
 * checked against the compiled reference. */
enum yolo_decode { YOLO_DECODE_RATIO = 0, YOLO_DECODE_PIXEL = 1 };
/* RATIO: `_ratio_detection_layer` V3/YOLOV3.py:168-238 (normalised); PIXEL: `_detection_layer`
 * V3/yolo_v3.py:111-159 (input pixels).  Region (v2) heads are always normalised (V2/decode.py:13). */
enum yolo_location { YOLO_HOST = 0, YOLO_DEVICE = 1 };
enum yolo_image_format {
    YOLO_IMG_U8 = 0,           /* uint8  [n,S,S,3] already at network size */
    YOLO_IMG_F32 = 1,          /* float32 [n,S,S,3] already at network size */
    YOLO_IMG_F32_CHW = 2       /* float32 [n,3,S,S] planar: darknet's `image` layout (DN/image.c get_pixel) */
};
enum yolo_nms_mode {
    YOLO_NMS_TF = 0,           /* tf.image.non_max_suppression: class-agnostic, `>` iou, top max_out (row N1) */
    YOLO_NMS_PER_CLASS = 1,    /* V2 bboxes_nms (V2/utils.py:176-187): same-class, drop unless iou < thr (row N3) */
    YOLO_NMS_DARKNET = 2,      /* do_nms_sort (DN/box.c:58-89) on max-class score: same-class, drop if iou > thr */
    YOLO_NMS_TF_V1 = 4,        /* YOLOv1's call of tf.image.non_max_suppression (V1/YOLO_V1_Inference.py:259-266): as YOLO_NMS_TF on
                                * boxes whose horizontal extent is built from h and vertical extent from w (the reference's swap);
                                * the records hold those corners: centre = their midpoint, w = y1 - y0, h = x1 - x0 */
    YOLO_NMS_NUMPY_V3 = 3      /* `non_max_suppression` V3/yolo_v3.py:376-420 (row N2): gate on objectness > thr, class =
                                * argmax cls, per class by objectness, keep iou < thr with the unclamped `_iou`; reported
                                * score reproduces the reference's shifted-index behaviour.  Records come out class by
                                * class (ascending), in selection order; boxes are corners (x - w/2 form). */
};
enum yolo_select_mode {
    YOLO_SELECT_GT = 0,        /* max(obj*cls) >  thr  (V3/YOLOV3.py:358) */
    YOLO_SELECT_GE = 1         /* max(obj*cls) >= thr  (V2/postprocess.py:61) */
};

typedef struct yolo_config {
    uint32_t struct_size;      /* = sizeof(yolo_config) */
    const char *cfg_text;      /* darknet cfg text: [net] + convolutional/shortcut/route/upsample/maxpool/reorg/yolo/region */
    int32_t max_batch;
    int32_t dtype;             /* enum yolo_dtype */
    int32_t semantics;         /* enum yolo_semantics */
    int32_t decode;            /* enum yolo_decode */
    int32_t device;            /* HIP device ordinal */
    int32_t keep_layers;       /* 1: every layer output keeps its own buffer (yolo_layer_output valid) */
    void *stream;              /* hipStream_t to launch on, or NULL for a context-owned stream */
} yolo_config;

/* One detection: corners in the decode's units, max class score, class id (24 bytes). */
typedef struct yolo_box {
    float x0, y0, x1, y1;
    float score;
    int32_t cls;
} yolo_box;

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* Parses the topology, plans buffers on `device`.  On failure returns NULL and writes a message to
 * err (if non-NULL).  Replaces load_network's cfg half (DN/parser.c:730-875). */
yolo_ctx *yolo_create(const yolo_config *cfg, char *err, size_t err_len);
void yolo_destroy(yolo_ctx *ctx);
const char *yolo_last_error(const yolo_ctx *ctx);

/* ---- weights (row L) ------------------------------------------------------------------------ */
/* Reads a Darknet .weights file (header rule DN/parser.c:1259-1265; header_ints 4 or 5 forces the
 * reference loaders' fixed counts V3/yolo_v3.py:278 / D2T V2 :351, 0 = auto), folds BN
 * (W*g/sqrt(v+1e-5), b - m*g/sqrt(v+1e-5), fp32), packs and uploads. */
int yolo_load_darknet_weights(yolo_ctx *ctx, const char *path, int header_ints);
/* Export artifact: ONE self-describing file holding the cfg text, the run configuration (dtype, semantics,
 * decode), the folded + packed device-ready parameters of every conv (fp8: codes and scales) and the tile plan.  It is
 * to this library what the frozen `.pb` (input -> detected_boxes / detected_scores / detected_classes,
 * D2T/YOLO_V3_convert_darkenet_to_Tensorflow.py:99-104, D2T/object_detect.py:64-99) is to the reference: a caller
 * needs nothing else to run detections.  The file is checksummed; a mismatch, truncation or a packing that does not
 * fit the topology is an error, never a partial load. */
int yolo_export(yolo_ctx *ctx, const char *path);
yolo_ctx *yolo_create_from_file(const char *path, int max_batch, int device, void *stream, int keep_layers, char *err, size_t err_len);
/* Same from the float stream that follows the header (n floats, must match the topology). */
int yolo_set_weights(yolo_ctx *ctx, const float *flat, size_t n);
size_t yolo_weights_count(const yolo_ctx *ctx);
/* YOLO_FP8 only: per-layer activation scales (stored code = e4m3(value / scale)), one float per cfg layer; entries of
 * layers that only move data (route, upsample, maxpool, reorg, yolo/region) are ignored.  Default: all 1.  Call BEFORE
 * loading weights (filters absorb their input scales; a later call invalidates loaded weights).  No reference
 * counterpart: the reference has no reduced-precision path. */
int yolo_set_act_scales(yolo_ctx *ctx, const float *scales, int n_layers);

/* ---- geometry ------------------------------------------------------------------------------- */
int yolo_input_size(const yolo_ctx *ctx, int *height, int *width, int *channels);
int yolo_num_rows(const yolo_ctx *ctx);      /* candidates per image: 10647 @416 v3, 845 v2 */
int yolo_num_attrs(const yolo_ctx *ctx);     /* 5 + classes */
int yolo_num_layers(const yolo_ctx *ctx);
/* detection head number `head` (0-based, cfg order): kind 0 = [yolo], 1 = [region]; grid side; anchors per cell; first row of
 * this head in the decoded tensor (rows are cell-major, anchor inner).  YOLO_ERR_INVALID past the last head. */
int yolo_head_geometry(const yolo_ctx *ctx, int head, int *kind, int *grid, int *anchors, int *row_offset);
double yolo_conv_flops(const yolo_ctx *ctx); /* 2*k*k*Cin*Cout*Ho*Wo summed (DN/convolutional_layer.c:325), per image */
double yolo_conv_bytes(const yolo_ctx *ctx, int n); /* algorithmic HBM bytes of the conv stack for n images */

/* ---- hot path ------------------------------------------------------------------------------- */
/* images: [n,S,S,3] in `fmt`, at `loc`; each value is multiplied by `scale` on the way in
 * (1/255 for the `inputs / 255` of V3/yolo_v3.py:215, 1 for pre-normalised input).  Runs the conv stack
 * and the head decode; the decoded tensor [n, rows, attrs] (V3/yolo_v3.py:266) stays resident and, when
 * detections_out != NULL, is also copied there (fp32, at out_loc).  Asynchronous on the context stream
 * unless a host buffer is involved. */
int yolo_forward(yolo_ctx *ctx, const void *images, int n, int fmt, int loc, float scale,
                 float *detections_out, int out_loc);

/* Row P: ONE uint8 RGB image of any size [h,w,3] -> /255 -> legacy bilinear stretch to
 * S x S on the device (D2T/YOLO_V3_convert...py:106-111), then as yolo_forward with n = 1. */
int yolo_forward_image_u8(yolo_ctx *ctx, const uint8_t *image, int h, int w, int loc,
                          float *detections_out, int out_loc);
/* darknet's network_predict_image (DN/network.c:579-586): ONE planar float image [3,h,w] (0..1) of any size ->
 * letterbox_image (DN/image.c:960-981: aspect-preserving resize_image DN/image.c:1347-1393, 0.5 fill, centred) on the
 * device, then as yolo_forward with n = 1. */
int yolo_forward_letterbox_chw(yolo_ctx *ctx, const float *image_chw, int w, int h, int loc,
                               float *detections_out, int out_loc);

/* darknet's get_network_boxes on the device (DN/network.c:536-567 = num_detections + fill_network_boxes ->
 * get_yolo_detections DN/yolo_layer.c:316-343 / get_region_detections DN/region_layer.c:364-437 (softmax heads, no tree),
 * then correct_yolo_boxes DN/yolo_layer.c:247-273) over image 0 of the last forward: boxes above `thresh`, compacted in
 * darknet's order, un-letterboxed for a w x h source image.  records: host [cap][5 + classes] = x, y, w, h, objectness,
 * prob[classes]; *count = number of detections (may exceed cap; cap = 0 / records = NULL: count only = num_detections). */
int yolo_darknet_boxes(yolo_ctx *ctx, int w, int h, float thresh, int relative, float *records, int cap, int *count);
/* What darknet's network_predict returns (DN/network.c:497-508, net->output): the LAST layer's output of image 0 in darknet's
 * own layout -- for a [yolo] / [region] layer the planar [anchors * (5 + classes)][grid * grid] tensor with that layer's
 * activations applied (DN/yolo_layer.c:143-152, DN/region_layer.c:160-186).  host buffer of yolo_last_layer_size() floats. */
size_t yolo_last_layer_size(const yolo_ctx *ctx);
int yolo_last_layer_output(yolo_ctx *ctx, float *out, size_t out_floats);
/* The same for images 0..n-1 of the last forward, image after image (darknet's net->output is batch * outputs contiguous floats,
 * DN/network.c:497-508 with l.output sized batch * l.outputs, DN/yolo_layer.c:50): out holds n * yolo_last_layer_size() floats. */
int yolo_last_layer_output_batch(yolo_ctx *ctx, int n, float *out, size_t out_floats);

/* The raw tensor detection head `head` (0-based, cfg order) decodes, for images 0..n-1 of the last forward: the head conv's fp32
 * output [n, grid, grid, anchors * (5 + classes)] dense, to host -- what the reference's graph builders return before any decode
 * (`build_network` V2/model_darknet19_slim.py:198-200; the per-scale `predictions` of V3/yolo_v3.py:239-263). */
int yolo_head_raw(yolo_ctx *ctx, int head, int n, float *out, size_t out_floats);

/* Threshold + NMS on the resident decoded tensor of the last forward (rows S, N1/N3).
 * boxes_out: [n * max_out] caller-owned, counts_out: [n]; both at out_loc.
 * Replaces get_network_boxes + do_nms_* (DN/network.c:562, DN/box.c:21-89) and the TF tail
 * V3/YOLOV3.py:347-379. */
int yolo_postprocess(yolo_ctx *ctx, int n, float score_thr, float iou_thr, int max_out,
                     int nms_mode, int select_mode, yolo_box *boxes_out, int32_t *counts_out, int out_loc);

/* The same, and for every box record the ROW of the decoded tensor it was formed from (0 .. yolo_num_rows()-1 within its image):
 * rows_out [n * max_out] int32 at out_loc, -1 in the unused slots; NULL = not reported.  `self.boxes` of the reference's detectors
 * is the decoded row itself (V1/YOLO_V1_Inference.py:255-268 gathers `_boxes` with the NMS indices): the index is what
 * tf.image.non_max_suppression returns, mapped back through the threshold mask. */
int yolo_postprocess_rows(yolo_ctx *ctx, int n, float score_thr, float iou_thr, int max_out,
                          int nms_mode, int select_mode, yolo_box *boxes_out, int32_t *counts_out, int32_t *rows_out, int out_loc);

/* forward + postprocess. */
int yolo_detect(yolo_ctx *ctx, const void *images, int n, int fmt, int loc, float scale,
                float score_thr, float iou_thr, int max_out, int nms_mode, int select_mode,
                yolo_box *boxes_out, int32_t *counts_out, int out_loc);

/* yolo_detect for device-resident inputs and outputs, replayed from a HIP graph: the first call with a given argument
 * tuple runs eagerly, the second captures the launch sequence (preprocess, ~80 conv/ew launches, decode, NMS) into a
 * hipGraph, later calls replay it (removes the per-launch gaps of a launch-bound step).  Any change of argument
 * re-captures.  images / boxes_out / counts_out must be device pointers that stay valid between calls. */
int yolo_detect_graph(yolo_ctx *ctx, const void *images, int n, int fmt, float scale, float score_thr, float iou_thr,
                      int max_out, int nms_mode, int select_mode, yolo_box *boxes_out, int32_t *counts_out);

int yolo_synchronize(yolo_ctx *ctx);

/* ---- introspection / measurement ------------------------------------------------------------ */
/* Copies layer `index`'s output of the last forward as fp32 NHWC [n,H,W,C] to host (needs
 * keep_layers=1).  dims_out receives H,W,C.  Head (yolo/region) layers return the raw conv tensor. */
int yolo_layer_output(yolo_ctx *ctx, int index, int n, float *out, size_t out_floats, int *dims_out);
/* Times `iters` forwards of batch n on the context stream with HIP events:
 * total_ms = wall per forward; conv_ms = the conv launches' share of it (all layers minus all-but-conv, bulk-timed)
 * (events around every conv launch, separate pass).  Either may be NULL.
 * The timing entry points (this one, yolo_time_layers, yolo_autotune) run the network on the INPUT OF THE LAST FORWARD.  When that call
 * was given a device-resident uint8 batch, the fused first layers read the caller's buffer in place: it must still be valid here.  A
 * timing pass over more images than that batch held reads the context's own (zero-initialised or previously staged) input instead. */
int yolo_time_forward(yolo_ctx *ctx, int n, int iters, float *total_ms, float *conv_ms);
/* What this chip sustains on a register-resident MFMA loop (v_mfma_f32_16x16x32_bf16, or _f16 when f16 != 0; random operands, 8 waves per
 * CU, back-to-back launches for `seconds`, the second half measured): TFLOP/s and the shader clock the waves held (GHz; s_memtime over
 * s_memrealtime).  bench.py prints both next to its roofline fraction: the same binary reads 9 % apart by box, and `frac / (tflops / peak)`
 * is what compares across boxes.  No counterpart in the reference.  stream: a hipStream_t or NULL. */
int yolo_calibrate(int device, void *stream, int f16, double seconds, float *tflops, float *clock_ghz);
/* The memory side of the yardstick: GB/s (bytes read + bytes written) of a streaming device-to-device copy between two 1 GiB buffers,
 * launched back to back for `seconds`.  The boxes of one pool differ in their matrix-pipe clock AND in what their memory system sustains;
 * bench.py prints both numbers (roofline.calib_tflops, roofline.calib_copy_gbs). */
int yolo_calibrate_copy(int device, void *stream, double seconds, float *gbs);
/* Per-layer kernel time (ms, averaged over iters) for batch n into ms_out[num_layers]. */
int yolo_time_layers(yolo_ctx *ctx, int n, int iters, float *ms_out);
/* Tries every conv tile configuration on every conv layer at batch n and keeps the fastest. */
int yolo_autotune(yolo_ctx *ctx, int n, int iters);
/* Read / restore the per-layer tile choices (one int per layer, -1 for non-conv layers or "library default") so a
 * tuned plan can be persisted by the caller.  cfgs has yolo_num_layers() entries. */
int yolo_get_tile_configs(const yolo_ctx *ctx, int32_t *cfgs);
int yolo_set_tile_configs(yolo_ctx *ctx, const int32_t *cfgs);

/* ---- single operators on host buffers (parity tests call the production kernels through these) - */
/* x [n,h,w,cin] fp32 NHWC, w_hwio [k,k,cin,cout], bias [cout] (or NULL), residual [n,ho,wo,cout] or NULL.
 * act: 0 linear, 1 leaky(0.1).  pad = k/2.  out [n,ho,wo,cout] fp32.  dtype selects the kernel family;
 * tile_cfg < 0 lets the library choose, otherwise forces one tile configuration (0..yolo_op_conv_num_cfgs()-1). */
int yolo_op_conv2d(const float *x, int n, int h, int w, int cin, const float *w_hwio, const float *bias,
                   int k, int stride, int cout, int act, const float *residual, float *out,
                   int dtype, int tile_cfg, int device);
int yolo_op_conv_num_cfgs(void);
int yolo_op_upsample2x(const float *x, int n, int h, int w, int c, int semantics, float *out, int device);
int yolo_op_reorg(const float *x, int n, int h, int w, int c, int stride, int semantics, float *out, int device);
int yolo_op_maxpool(const float *x, int n, int h, int w, int c, int size, int stride, float *out, int device);
/* legacy-bilinear stretch of one uint8 image to [s,s,3] fp32: (value/255 then resize) * post_scale. */
int yolo_op_resize_u8(const uint8_t *img, int h, int w, int s, float post_scale, float *out, int device);
/* V2/utils.py:13-27 `preprocess_image`'s arithmetic on the device: cv2.resize(float32 image, (ow, oh)) -- INTER_LINEAR, half-pixel centres,
 * restated from OpenCV's published CV_32F linear resize -- of one uint8 [h,w,3] image, optionally after BGR -> RGB (swap_rb), then
 * / divisor (the reference divides by 225.0, a typo kept).  out [oh,ow,3] fp32. */
int yolo_op_resize_cv2(const uint8_t *img, int h, int w, int oh, int ow, int swap_rb, float divisor, float *out, int device);
/* `detections_boxes` (V3/yolo_v3.py:329-347): (cx,cy,w,h,...) -> (x0,y0,x1,y1,...) over [n,rows,attrs] fp32 */
int yolo_op_detections_boxes(const float *det, int n, int rows, int attrs, float *out, int device);
/* head decode of raw [n,g,g,na*(5+classes)] fp32: yolo (logistic) or region (softmax) */
int yolo_op_decode(const float *raw, int n, int g, int na, int classes, const float *anchors_wh,
                   int img_size, int decode, int region, float *out, int device);
/* threshold + NMS over det [n,rows,attrs] fp32.  nms_mode bits 8..19 / 20..31 carry image height / width for
 * YOLO_NMS_PER_CLASS (V2 pixel boxes); select_mode bit 8 set = rows already hold corners (x0,y0,x1,y1). */
/* darknet letterbox_image (embed = 1, DN/image.c:960-981) / resize_image (embed = 0, DN/image.c:1347-1389) of a planar
 * float image [3][ih][iw] into a planar w x h image, on the device.  Host buffers. */
int yolo_op_letterbox(const float *image_chw, int iw, int ih, int w, int h, int embed, float *out_chw, int device);
/* darknet's do_nms_sort (by_objectness = 0) / do_nms_obj (1), DN/box.c:21-89, on caller arrays (host): boxes [n] (cx,cy,w,h),
 * prob [n][classes], objectness [n]; suppressed entries are zeroed IN PLACE (prob[j][k], or objectness[j] and all of
 * prob[j]); detections whose objectness is 0 do not take part.  n <= 4096.  Array order is left unchanged (the
 * reference qsorts its array; callers only read the surviving probabilities). */
int yolo_op_nms_detections(const float *boxes_xywh, float *prob, float *objectness, int n, int classes, float thresh,
                           int by_objectness, int device);
int yolo_op_postprocess(const float *det, int n, int rows, int attrs, float score_thr, float iou_thr,
                        int max_out, int nms_mode, int select_mode, yolo_box *boxes_out,
                        int32_t *counts_out, int device);
/* ... with the source row of every record (see yolo_postprocess_rows); rows_out [n * max_out] host, or NULL */
int yolo_op_postprocess_rows(const float *det, int n, int rows, int attrs, float score_thr, float iou_thr,
                             int max_out, int nms_mode, int select_mode, yolo_box *boxes_out,
                             int32_t *counts_out, int32_t *rows_out, int device);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */

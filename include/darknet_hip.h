/* darknet veneer over libyolo_hip.so: every symbol of libdarknet's C ABI that the reference's own Python
 * binding resolves -- D2T/darknet.py:48-115 (`network_width/height`, `network_predict`, `cuda_set_device`, `make_image`,
 * `get_network_boxes`, `make_network_boxes`, `free_detections`, `free_ptrs`, `reset_rnn`, `load_network`, `do_nms_obj`,
 * `do_nms_sort`, `free_image`, `letterbox_image`, `get_metadata`, `load_image_color`, `rgbgr_image`, `network_predict_image`)
 * -- with the same names, struct layouts (DN include/darknet.h:35-38, 505-525) and argument meaning, so that `darknet.py`
 * imports and runs `detect()` after changing only the library it dlopens (libdarknet_hip.so).  The network, the letterbox /
 * resize, the head activations, the thresholding + box correction of get_network_boxes and both NMS flavours run on the GPU
 * through the C ABI of include/yolo_hip.h; this layer marshals darknet's structs and parses its small text files.
 *
 * Differences a caller can observe:
 *   - load_image_color decodes binary PPM / PGM only (the reference uses the vendored stb_image for JPEG / PNG);
 *   - do_nms_* leave the array order unchanged (the reference qsorts it in place);
 *   - [region] heads are served in their softmax form (no tree / map / mask coefficients), hier_thresh and map are ignored;
 *   - network_predict* return net->output for networks whose last layer is a [yolo] / [region] / [detection] head (every
 *     topology the reference's detectors use); precision is bf16 unless DARKNET_HIP_DTYPE=fp32 is set in the environment;
 *   - get_network_boxes reports the first image of a batch, as the reference does. */
#ifndef DARKNET_HIP_H
#define DARKNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, w, h; } box;                                /* DN include/darknet.h:514-516 */
typedef struct detection { box bbox; int classes; float *prob; float *mask; float objectness; int sort_class; } detection;   /* :518-525 */
typedef struct { int w, h, c; float *data; } image;                      /* :505-510, planar [c][h][w] */
typedef struct { int classes; char **names; } metadata;                  /* :35-38 */
typedef struct network network;                                          /* opaque */

void cuda_set_device(int n);                                             /* DN/cuda.c:12 */
network *load_network(char *cfg, char *weights, int clear);              /* DN/network.c:53-62 */
void free_network(network *net);                                         /* DN/network.c:716 */
int network_width(network *net);                                         /* DN/network.c:600 */
int network_height(network *net);
void set_batch_network(network *net, int b);                             /* DN/network.c:339 */
void reset_rnn(network *net);                                            /* DN/network.c:85 (no recurrent layers on this path) */
float *network_predict(network *net, float *input);                      /* DN/network.c:497; batch x planar [3][h][w] at network size */
float *network_predict_image(network *net, image im);                    /* DN/network.c:579-586 (letterbox + predict) */
detection *make_network_boxes(network *net, float thresh, int *num);     /* DN/network.c:526 */
detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num);   /* DN/network.c:562-567 */
void free_detections(detection *dets, int n);                            /* DN/network.c:569-577 */
void free_ptrs(void **ptrs, int n);                                      /* DN/utils.c:328 */
void do_nms_sort(detection *dets, int total, int classes, float thresh); /* DN/box.c:58-89 */
void do_nms_obj(detection *dets, int total, int classes, float thresh);  /* DN/box.c:21-55 */
image make_image(int w, int h, int c);                                   /* DN/image.c:798 */
void free_image(image m);
image letterbox_image(image im, int w, int h);                           /* DN/image.c:960-981 */
image load_image_color(char *filename, int w, int h);                    /* DN/image.c:1482 (PPM / PGM) */
void rgbgr_image(image im);                                              /* DN/image.c:527 */
metadata get_metadata(char *file);                                       /* DN/option_list.c:35 */

#ifdef __cplusplus
}
#endif
#endif

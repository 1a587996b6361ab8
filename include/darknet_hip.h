/* darknet veneer over libyolo_hip.so: the subset of libdarknet's C ABI that the reference's own
 * Python binding uses to run detections -- D2T/darknet.py:48-142 (`load_net`, `predict_image`, `get_network_boxes`,
 * `do_nms_obj` / `do_nms_sort`, `free_detections`, `network_width/height`, `make_image` / `free_image`) -- with the
 * same names, struct layouts (DN include/darknet.h:505-525) and argument meaning, so a `darknet.py`-style caller only
 * changes the library it dlopens (libdarknet_hip.so).  The network, the letterbox resize, the head decode and both
 * NMS flavours run on the GPU through the C ABI of include/yolo_hip.h; this layer only marshals darknet's structs.
 *
 * Differences, all visible to a caller that looks: `network_predict*` return the DECODED rows
 * [rows][5 + classes] = (cx, cy, w, h, objectness, class probabilities) of all heads instead of the raw activations of
 * the last layer; do_nms_* leave the array order unchanged (the reference qsorts it); [region] heads are served in their
 * softmax form (no tree / map, no mask coefficients); precision is bf16 unless DARKNET_HIP_DTYPE=fp32 is set in the environment. */
#ifndef DARKNET_HIP_H
#define DARKNET_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, w, h; } box;                                /* DN include/darknet.h:514-516 */
typedef struct detection { box bbox; int classes; float *prob; float *mask; float objectness; int sort_class; } detection;   /* :518-525 */
typedef struct { int w, h, c; float *data; } image;                      /* :505-510, planar [c][h][w] */
typedef struct network network;                                          /* opaque */

network *load_network(char *cfg, char *weights, int clear);              /* DN/network.c:53-62 */
void free_network(network *net);                                         /* DN/network.c:716 */
int network_width(network *net);                                         /* DN/network.c:600 */
int network_height(network *net);
void set_batch_network(network *net, int b);                             /* DN/network.c:335; only b == 1 */
float *network_predict(network *net, float *input);                      /* DN/network.c:497; planar [3][h][w] at network size */
float *network_predict_image(network *net, image im);                    /* DN/network.c:579-586 (letterbox + predict) */
detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num);   /* DN/network.c:562-567 */
void free_detections(detection *dets, int n);                            /* DN/network.c:569-577 */
void do_nms_sort(detection *dets, int total, int classes, float thresh); /* DN/box.c:58-89 */
void do_nms_obj(detection *dets, int total, int classes, float thresh);  /* DN/box.c:21-55 */
image make_image(int w, int h, int c);                                   /* DN/image.c:798 */
void free_image(image m);

#ifdef __cplusplus
}
#endif
#endif

/* Accessors compiled INTO oracle/_ref/libdarknet_ref.so next to the reference's own objects.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference's `layer` / `network` structs
 * (Darknet2Tensorflow/darknet-master/include/darknet.h:118-423, :429-495) are too large to mirror
 * in ctypes, so this file -- our code, compiled against the reference header where it lies --
 * exposes the handful of fields the parity tests read.  It adds no arithmetic.
 */
#include "darknet.h"

int ref_num_layers(network *net) { return net->n; }
float *ref_layer_output(network *net, int i) { return net->layers[i].output; }
int ref_layer_outputs(network *net, int i) { return net->layers[i].outputs; }
int ref_layer_type(network *net, int i) { return (int)net->layers[i].type; }
void ref_layer_dims(network *net, int i, int *whc)
{
    whc[0] = net->layers[i].out_w; whc[1] = net->layers[i].out_h; whc[2] = net->layers[i].out_c;
}
int ref_net_w(network *net) { return net->w; }
int ref_net_h(network *net) { return net->h; }
int ref_type_yolo(void) { return (int)YOLO; }
int ref_type_region(void) { return (int)REGION; }

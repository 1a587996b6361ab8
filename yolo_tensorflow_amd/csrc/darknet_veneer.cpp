// libdarknet_hip.so: darknet's detection ABI (include/darknet_hip.h) marshalled onto the C ABI of libyolo_hip.so.
// Host code only -- every computation (network, letterbox, decode, NMS) is a call into include/yolo_hip.h.
#include "../../include/darknet_hip.h"
#include "../../include/yolo_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct network {
    yolo_ctx *ctx = nullptr;
    int w = 0, h = 0, rows = 0, attrs = 0;
    std::vector<float> det;              // decoded rows of the last predict
    bool have = false;
};

namespace {
bool read_text(const char *path, std::string &out)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return false;
    char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
}  // namespace

extern "C" {

network *load_network(char *cfg, char *weights, int clear)
{
    (void)clear;                                   // `*net->seen = 0`: training state, nothing to clear here
    std::string text;
    if (!read_text(cfg, text)) { fprintf(stderr, "darknet_hip: cannot open cfg '%s'\n", cfg ? cfg : "(null)"); return nullptr; }
    yolo_config yc; memset(&yc, 0, sizeof yc);
    const char *dt = getenv("DARKNET_HIP_DTYPE");
    yc.struct_size = sizeof yc; yc.cfg_text = text.c_str(); yc.max_batch = 1; yc.dtype = dt && !strcmp(dt, "fp32") ? YOLO_FP32 : YOLO_BF16;
    yc.semantics = YOLO_SEM_DARKNET; yc.decode = YOLO_DECODE_RATIO; yc.device = 0;
    char err[512] = {0};
    network *net = new network();
    net->ctx = yolo_create(&yc, err, sizeof err);
    if (!net->ctx) { fprintf(stderr, "darknet_hip: %s\n", err); delete net; return nullptr; }
    if (weights && weights[0] && yolo_load_darknet_weights(net->ctx, weights, 0) != YOLO_OK) {
        fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); yolo_destroy(net->ctx); delete net; return nullptr;
    }
    yolo_input_size(net->ctx, &net->h, &net->w, nullptr);
    net->rows = yolo_num_rows(net->ctx); net->attrs = yolo_num_attrs(net->ctx);
    net->det.assign((size_t)net->rows * net->attrs, 0.f);
    return net;
}

void free_network(network *net) { if (!net) return; yolo_destroy(net->ctx); delete net; }
int network_width(network *net) { return net ? net->w : 0; }
int network_height(network *net) { return net ? net->h : 0; }
void set_batch_network(network *net, int b) { (void)net; if (b != 1) fprintf(stderr, "darknet_hip: the veneer serves batch 1 (use yolo_hip.h for batches)\n"); }

float *network_predict(network *net, float *input)
{
    if (!net || !input) return nullptr;
    net->have = yolo_forward(net->ctx, input, 1, YOLO_IMG_F32_CHW, YOLO_HOST, 1.0f, net->det.data(), YOLO_HOST) == YOLO_OK;
    if (!net->have) { fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr; }
    return net->det.data();
}

float *network_predict_image(network *net, image im)
{
    if (!net || !im.data || im.c != 3) return nullptr;
    net->have = yolo_forward_letterbox_chw(net->ctx, im.data, im.w, im.h, YOLO_HOST, net->det.data(), YOLO_HOST) == YOLO_OK;
    if (!net->have) { fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr; }
    return net->det.data();
}

// get_yolo_detections (DN/yolo_layer.c:316-343) / get_region_detections (DN/region_layer.c:364-437, softmax heads
// without a tree) over every head in network order, then correct_yolo_boxes / correct_region_boxes (identical
// arithmetic, DN/yolo_layer.c:247-273, DN/region_layer.c:336-362).  The decoded rows already hold get_yolo_box's /
// get_region_box's (x, y, w, h) relative to the network input.
detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num)
{
    (void)hier; (void)map;
    if (num) *num = 0;
    if (!net || !net->have) return nullptr;
    const int A = net->attrs, C = A - 5;
    // pass 1: how many detections (a yolo head reports objectness > thresh, a region head every box)
    struct Head { int kind, grid, na, off; };
    std::vector<Head> heads;
    for (int k = 0;; ++k) { Head hd; if (yolo_head_geometry(net->ctx, k, &hd.kind, &hd.grid, &hd.na, &hd.off) != YOLO_OK) break; heads.push_back(hd); }
    int count = 0;
    for (auto &hd : heads) {
        const int rows = hd.grid * hd.grid * hd.na;
        if (hd.kind == 1) count += rows;
        else for (int r = 0; r < rows; ++r) if (net->det[(size_t)(hd.off + r) * A + 4] > thresh) ++count;
    }
    detection *dets = (detection *)calloc(count > 0 ? count : 1, sizeof(detection));
    int k = 0;
    for (auto &hd : heads) {
        const int cells = hd.grid * hd.grid;
        if (hd.kind == 1) {
            // region: index = anchor * cells + cell (DN/region_layer.c:395); objectness and probabilities gated by thresh
            for (int n = 0; n < hd.na; ++n)
                for (int i = 0; i < cells; ++i) {
                    const float *p = &net->det[(size_t)(hd.off + i * hd.na + n) * A];
                    detection &d = dets[k++];
                    const float scale = p[4];
                    d.bbox = box{p[0], p[1], p[2], p[3]}; d.classes = C; d.objectness = scale > thresh ? scale : 0;
                    d.prob = (float *)calloc(C, sizeof(float));
                    if (d.objectness) for (int j = 0; j < C; ++j) { const float prob = scale * p[5 + j]; d.prob[j] = prob > thresh ? prob : 0; }
                }
        } else {
            for (int r = 0; r < cells * hd.na; ++r) {
                const float *p = &net->det[(size_t)(hd.off + r) * A];
                const float objectness = p[4];
                if (!(objectness > thresh)) continue;
                detection &d = dets[k++];
                d.bbox = box{p[0], p[1], p[2], p[3]}; d.classes = C; d.objectness = objectness;
                d.prob = (float *)calloc(C, sizeof(float));
                for (int j = 0; j < C; ++j) { const float prob = objectness * p[5 + j]; d.prob[j] = prob > thresh ? prob : 0; }
            }
        }
    }
    const int netw = net->w, neth = net->h;
    int new_w, new_h;
    if (((float)netw / w) < ((float)neth / h)) { new_w = netw; new_h = (h * netw) / w; } else { new_h = neth; new_w = (w * neth) / h; }
    for (int i = 0; i < count; ++i) {
        box b = dets[i].bbox;
        b.x = (b.x - (netw - new_w) / 2. / netw) / ((float)new_w / netw);
        b.y = (b.y - (neth - new_h) / 2. / neth) / ((float)new_h / neth);
        b.w *= (float)netw / new_w;
        b.h *= (float)neth / new_h;
        if (!relative) { b.x *= w; b.w *= w; b.y *= h; b.h *= h; }
        dets[i].bbox = b;
    }
    if (num) *num = count;
    return dets;
}

void free_detections(detection *dets, int n)
{
    if (!dets) return;
    for (int i = 0; i < n; ++i) { free(dets[i].prob); if (dets[i].mask) free(dets[i].mask); }
    free(dets);
}

static void nms_arrays(detection *dets, int total, int classes, float thresh, int by_obj)
{
    if (!dets || total < 1 || classes < 1) return;
    std::vector<float> b((size_t)total * 4), p((size_t)total * classes), o(total);
    for (int i = 0; i < total; ++i) {
        b[4 * i] = dets[i].bbox.x; b[4 * i + 1] = dets[i].bbox.y; b[4 * i + 2] = dets[i].bbox.w; b[4 * i + 3] = dets[i].bbox.h;
        o[i] = dets[i].objectness; memcpy(&p[(size_t)i * classes], dets[i].prob, (size_t)classes * 4);
    }
    if (yolo_op_nms_detections(b.data(), p.data(), o.data(), total, classes, thresh, by_obj, 0) != YOLO_OK) {
        fprintf(stderr, "darknet_hip: nms: %s\n", yolo_last_error(nullptr)); return;
    }
    for (int i = 0; i < total; ++i) { dets[i].objectness = o[i]; memcpy(dets[i].prob, &p[(size_t)i * classes], (size_t)classes * 4); }
}
void do_nms_sort(detection *dets, int total, int classes, float thresh) { nms_arrays(dets, total, classes, thresh, 0); }
void do_nms_obj(detection *dets, int total, int classes, float thresh) { nms_arrays(dets, total, classes, thresh, 1); }

image make_image(int w, int h, int c) { image m; m.w = w; m.h = h; m.c = c; m.data = (float *)calloc((size_t)w * h * c, sizeof(float)); return m; }
void free_image(image m) { free(m.data); }

}  // extern "C"

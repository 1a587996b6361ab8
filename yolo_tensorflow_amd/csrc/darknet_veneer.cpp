// libdarknet_hip.so: darknet's detection ABI (include/darknet_hip.h) marshalled onto the C ABI of libyolo_hip.so.
// Host code only.  What computes -- the network, the letterbox / resize, the head activations, thresholding + box correction of
// get_network_boxes, both NMS flavours -- is a call into include/yolo_hip.h and runs on the GPU; this file allocates and fills
// darknet's structs, parses the two small text formats darknet.py hands over (.data files, class-name lists) and decodes PPM
// images (the reference decodes JPEG/PNG with the vendored stb_image, which is outside the inference path and not re-implemented).
#include "../../include/darknet_hip.h"
#include "../../include/yolo_hip.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct network {
    yolo_ctx *ctx = nullptr;
    int w = 0, h = 0, rows = 0, attrs = 0, batch = 1;
    std::string cfg_text, weights_path;
    std::vector<float> out;              // last layer's output in darknet layout (what network_predict returns)
    std::vector<float> rec;              // get_network_boxes staging: [count][5 + classes]
    bool have = false;
    int fwd_n = 1;                       // images of the last forward (network_predict: batch; network_predict_image: 1)
};

namespace {
int g_device = 0;                        // cuda_set_device

bool read_text(const char *path, std::string &out)
{
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) return false;
    char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

bool open_ctx(network *net)
{
    yolo_config yc; memset(&yc, 0, sizeof yc);
    const char *dt = getenv("DARKNET_HIP_DTYPE");
    yc.struct_size = sizeof yc; yc.cfg_text = net->cfg_text.c_str(); yc.max_batch = net->batch;
    yc.dtype = dt && !strcmp(dt, "fp32") ? YOLO_FP32 : dt && !strcmp(dt, "fp16") ? YOLO_FP16 : dt && !strcmp(dt, "fp16x2") ? YOLO_FP16X2 : YOLO_BF16;
    yc.semantics = YOLO_SEM_DARKNET; yc.decode = YOLO_DECODE_RATIO; yc.device = g_device;
    char err[512] = {0};
    net->ctx = yolo_create(&yc, err, sizeof err);
    if (!net->ctx) { fprintf(stderr, "darknet_hip: %s\n", err); return false; }
    if (!net->weights_path.empty() && yolo_load_darknet_weights(net->ctx, net->weights_path.c_str(), 0) != YOLO_OK) {
        fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); yolo_destroy(net->ctx); net->ctx = nullptr; return false;
    }
    yolo_input_size(net->ctx, &net->h, &net->w, nullptr);
    net->rows = yolo_num_rows(net->ctx); net->attrs = yolo_num_attrs(net->ctx);
    net->out.assign(yolo_last_layer_size(net->ctx) * (size_t)net->batch, 0.f);      // net->output: batch * outputs floats
    net->have = false;
    return true;
}

float *fetch_output(network *net)
{
    net->have = yolo_last_layer_output_batch(net->ctx, net->fwd_n, net->out.data(), net->out.size()) == YOLO_OK;
    if (!net->have) { fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr; }
    return net->out.data();
}
}  // namespace

extern "C" {

void cuda_set_device(int n) { g_device = n; }                                        /* DN/cuda.c:12 */

network *load_network(char *cfg, char *weights, int clear)
{
    (void)clear;                                   // `*net->seen = 0`: training state, nothing to clear here
    network *net = new network();
    if (!read_text(cfg, net->cfg_text)) { fprintf(stderr, "darknet_hip: cannot open cfg '%s'\n", cfg ? cfg : "(null)"); delete net; return nullptr; }
    if (weights && weights[0]) net->weights_path = weights;
    if (!open_ctx(net)) { delete net; return nullptr; }
    return net;
}

void free_network(network *net) { if (!net) return; if (net->ctx) yolo_destroy(net->ctx); delete net; }
int network_width(network *net) { return net ? net->w : 0; }
int network_height(network *net) { return net ? net->h : 0; }
void reset_rnn(network *net) { (void)net; }                                          /* DN/network.c:85: no recurrent state here */

// DN/network.c:339-356 resizes every layer for `b` images; here the plan is rebuilt for that batch (weights are re-read)
void set_batch_network(network *net, int b)
{
    if (!net || b < 1 || b == net->batch) return;
    const int old = net->batch;
    yolo_ctx *prev = net->ctx;
    net->batch = b; net->ctx = nullptr;
    if (!open_ctx(net)) { fprintf(stderr, "darknet_hip: set_batch_network(%d) failed, keeping batch %d\n", b, old); net->batch = old; net->ctx = prev; return; }
    yolo_destroy(prev);
}

// `input`: batch x planar [3][h][w] at network size.  Returns net->output: the last layer's output of every image of the batch,
// image after image (batch * outputs floats, DN/network.c:497-508), see include/darknet_hip.h.
float *network_predict(network *net, float *input)
{
    if (!net || !input) return nullptr;
    if (yolo_forward(net->ctx, input, net->batch, YOLO_IMG_F32_CHW, YOLO_HOST, 1.0f, nullptr, YOLO_HOST) != YOLO_OK) {
        net->have = false; fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr;
    }
    net->fwd_n = net->batch;
    return fetch_output(net);
}

float *network_predict_image(network *net, image im)
{
    if (!net || !im.data || im.c != 3) return nullptr;
    if (yolo_forward_letterbox_chw(net->ctx, im.data, im.w, im.h, YOLO_HOST, nullptr, YOLO_HOST) != YOLO_OK) {
        net->have = false; fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr;
    }
    net->fwd_n = 1;
    return fetch_output(net);
}

// DN/network.c:526-540: `num` boxes above thresh, each with a zeroed prob array
detection *make_network_boxes(network *net, float thresh, int *num)
{
    if (num) *num = 0;
    if (!net || !net->have) return nullptr;
    int count = 0;
    if (yolo_darknet_boxes(net->ctx, net->w, net->h, thresh, 1, nullptr, 0, &count) != YOLO_OK) { fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr; }
    const int C = net->attrs - 5;
    detection *dets = (detection *)calloc(count > 0 ? count : 1, sizeof(detection));
    for (int i = 0; i < count; ++i) { dets[i].classes = C; dets[i].prob = (float *)calloc(C, sizeof(float)); }
    if (num) *num = count;
    return dets;
}

// DN/network.c:562-567.  Thresholding, ordering and the un-letterboxing run on the device (yolo_darknet_boxes); only the
// surviving boxes cross PCIe and are copied into darknet's structs here.
detection *get_network_boxes(network *net, int w, int h, float thresh, float hier, int *map, int relative, int *num)
{
    (void)hier; (void)map;
    if (num) *num = 0;
    if (!net || !net->have) return nullptr;
    const int A = net->attrs, C = A - 5;
    net->rec.resize((size_t)net->rows * A);
    int count = 0;
    if (yolo_darknet_boxes(net->ctx, w, h, thresh, relative, net->rec.data(), net->rows, &count) != YOLO_OK) {
        fprintf(stderr, "darknet_hip: %s\n", yolo_last_error(net->ctx)); return nullptr;
    }
    detection *dets = (detection *)calloc(count > 0 ? count : 1, sizeof(detection));
    for (int i = 0; i < count; ++i) {
        const float *p = &net->rec[(size_t)i * A];
        detection &d = dets[i];
        d.bbox = box{p[0], p[1], p[2], p[3]}; d.classes = C; d.objectness = p[4];
        d.prob = (float *)malloc((size_t)C * sizeof(float));
        memcpy(d.prob, p + 5, (size_t)C * sizeof(float));
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

void free_ptrs(void **ptrs, int n)                                                   /* DN/utils.c:328 */
{
    if (!ptrs) return;
    for (int i = 0; i < n; ++i) free(ptrs[i]);
    free(ptrs);
}

static void nms_arrays(detection *dets, int total, int classes, float thresh, int by_obj)
{
    if (!dets || total < 1 || classes < 1) return;
    std::vector<float> b((size_t)total * 4), p((size_t)total * classes), o(total);
    for (int i = 0; i < total; ++i) {
        b[4 * i] = dets[i].bbox.x; b[4 * i + 1] = dets[i].bbox.y; b[4 * i + 2] = dets[i].bbox.w; b[4 * i + 3] = dets[i].bbox.h;
        o[i] = dets[i].objectness; memcpy(&p[(size_t)i * classes], dets[i].prob, (size_t)classes * 4);
    }
    if (yolo_op_nms_detections(b.data(), p.data(), o.data(), total, classes, thresh, by_obj, g_device) != YOLO_OK) {
        fprintf(stderr, "darknet_hip: nms: %s\n", yolo_last_error(NULL)); return;
    }
    for (int i = 0; i < total; ++i) { dets[i].objectness = o[i]; memcpy(dets[i].prob, &p[(size_t)i * classes], (size_t)classes * 4); }
}
void do_nms_sort(detection *dets, int total, int classes, float thresh) { nms_arrays(dets, total, classes, thresh, 0); }
void do_nms_obj(detection *dets, int total, int classes, float thresh) { nms_arrays(dets, total, classes, thresh, 1); }

image make_image(int w, int h, int c) { image m; m.w = w; m.h = h; m.c = c; m.data = (float *)calloc((size_t)w * h * c, sizeof(float)); return m; }
void free_image(image m) { free(m.data); }

image letterbox_image(image im, int w, int h)                                        /* DN/image.c:960-981, on the device */
{
    image out = make_image(w, h, im.c);
    if (!im.data || im.c != 3 || yolo_op_letterbox(im.data, im.w, im.h, w, h, 1, out.data, g_device) != YOLO_OK)
        fprintf(stderr, "darknet_hip: letterbox_image: %s\n", im.c != 3 ? "3-channel images only" : yolo_last_error(NULL));
    return out;
}

void rgbgr_image(image im)                                                           /* DN/image.c:527 */
{
    if (!im.data || im.c < 3) return;
    const size_t n = (size_t)im.w * im.h;
    for (size_t i = 0; i < n; ++i) { const float t = im.data[i]; im.data[i] = im.data[i + 2 * n]; im.data[i + 2 * n] = t; }
}

// DN/image.c:1442-1485 (load_image_stb + optional resize_image).  Binary PPM (P6) and PGM (P5) only, 8 bits per sample.
image load_image_color(char *filename, int w, int h)
{
    image bad = {0, 0, 0, nullptr};
    FILE *f = filename ? fopen(filename, "rb") : nullptr;
    if (!f) { fprintf(stderr, "darknet_hip: cannot load image \"%s\"\n", filename ? filename : "(null)"); return bad; }
    auto token = [&](std::string &t) {
        t.clear(); int ch;
        for (;;) {
            ch = fgetc(f);
            if (ch == '#') { while ((ch = fgetc(f)) != EOF && ch != '\n') {} continue; }
            if (ch == EOF || !isspace(ch)) break;
        }
        while (ch != EOF && !isspace(ch)) { t.push_back((char)ch); ch = fgetc(f); }
        return !t.empty();
    };
    std::string magic, sw, sh, smax;
    if (!token(magic) || (magic != "P6" && magic != "P5") || !token(sw) || !token(sh) || !token(smax) || atoi(smax.c_str()) != 255) {
        fclose(f); fprintf(stderr, "darknet_hip: \"%s\" is not an 8-bit binary PPM/PGM (JPEG/PNG decoding is outside the inference path: convert the image first)\n", filename); return bad;
    }
    const int iw = atoi(sw.c_str()), ih = atoi(sh.c_str()), ch = magic == "P6" ? 3 : 1;
    if (iw < 1 || ih < 1 || (long)iw * ih > (1L << 28)) { fclose(f); return bad; }
    std::vector<unsigned char> raw((size_t)iw * ih * ch);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) { fprintf(stderr, "darknet_hip: short read on \"%s\"\n", filename); return bad; }
    image im = make_image(iw, ih, 3);
    const size_t n = (size_t)iw * ih;
    for (int k = 0; k < 3; ++k)
        for (size_t i = 0; i < n; ++i) im.data[k * n + i] = (float)raw[i * ch + (ch == 3 ? k : 0)] / 255.;       /* DN/image.c:1458 */
    if (h && w && (h != im.h || w != im.w)) {
        image r = make_image(w, h, 3);
        if (yolo_op_letterbox(im.data, im.w, im.h, w, h, 0, r.data, g_device) != YOLO_OK) fprintf(stderr, "darknet_hip: resize: %s\n", yolo_last_error(NULL));
        free_image(im); im = r;
    }
    return im;
}

// DN/option_list.c:35-50 + get_labels DN/data.c:618: `classes = N`, `names = path` (or `labels = path`); one class name per line
metadata get_metadata(char *file)
{
    metadata m = {0, nullptr};
    std::string text;
    if (!read_text(file, text)) { fprintf(stderr, "darknet_hip: couldn't open file: %s\n", file ? file : "(null)"); return m; }
    std::string names_path; int classes = 2;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos); if (e == std::string::npos) e = text.size();
        std::string line = trim(text.substr(pos, e - pos)); pos = e + 1;
        if (line.empty() || line[0] == '#' || line[0] == ';') continue;
        const size_t eq = line.find('='); if (eq == std::string::npos) continue;
        const std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        if (k == "classes") classes = atoi(v.c_str());
        else if (k == "names" || (k == "labels" && names_path.empty())) names_path = v;
    }
    m.classes = classes;
    if (names_path.empty()) { fprintf(stderr, "No names or labels found\n"); return m; }
    std::string list;
    if (!read_text(names_path.c_str(), list)) { fprintf(stderr, "darknet_hip: couldn't open file: %s\n", names_path.c_str()); return m; }
    std::vector<std::string> names;
    pos = 0;
    while (pos < list.size()) {
        size_t e = list.find('\n', pos); if (e == std::string::npos) e = list.size();
        std::string line = list.substr(pos, e - pos); pos = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() || pos < list.size()) names.push_back(line);
    }
    m.names = (char **)calloc(names.size() ? names.size() : 1, sizeof(char *));
    for (size_t i = 0; i < names.size(); ++i) m.names[i] = strdup(names[i].c_str());
    return m;
}

}  // extern "C"

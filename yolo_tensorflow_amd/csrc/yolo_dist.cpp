// include/yolo_dist.h: the image-sharded detect step.  One contiguous slice of the global batch per rank, a full
// weight replica per rank, ONE all-gather of fixed-capacity record buffers per step.  RCCL is bound at run time with dlopen -- the
// prototypes below are rccl.h's (ncclUniqueId is 128 opaque bytes passed by value, ncclInt32 = 2) -- so libyolo_hip.so carries no
// link-time dependency on it.  Layout and split are the ones yolo_tensorflow_amd/dist.py uses over torch.distributed
// (shard_bounds, alloc_flat_records, split_flat_records_ragged): tests/test_host.py checks the two against each other.
#include "yolo_ctx.h"
#include "../../include/yolo_dist.h"

#include <dlfcn.h>

namespace {

struct UniqueId { char internal[128]; };
typedef void *Comm;
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl load_rccl()
{
    Rccl r;
    const char *names[] = {getenv("YOLO_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n || !*n) continue;
        if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!r.so) { const char *e = dlerror(); r.why = "RCCL not found (set YOLO_RCCL_LIB): "; r.why += e ? e : "dlopen failed"; return r; }
    auto sym = [&](const char *s) { void *p = dlsym(r.so, s); if (!p && r.why.empty()) { r.why = "RCCL lacks "; r.why += s; } return p; };
    r.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
    r.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
    r.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    r.AllGather = (int (*)(const void *, void *, size_t, int, Comm, hipStream_t))sym("ncclAllGather");
    r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    if (!r.why.empty()) { dlclose(r.so); r.so = nullptr; }
    return r;
}

// bound once, on first use (a function-local static: initialised exactly once even when contexts on several threads get here together)
const Rccl &rccl()
{
    static const Rccl r = load_rccl();
    return r;
}

constexpr int kRecordWords = sizeof(yolo_box) / 4;      // 6

struct ErrBuf { char *p; size_t n; };
__attribute__((format(printf, 2, 3))) yolo_dist *bad(ErrBuf e, const char *fmt, ...)
{
    if (e.p && e.n) { va_list ap; va_start(ap, fmt); vsnprintf(e.p, e.n, fmt, ap); va_end(ap); }
    return nullptr;
}
constexpr int kNcclInt32 = 2;

}  // namespace

struct yolo_dist {
    yolo_ctx *ctx = nullptr;
    int device = 0;                  // cached: yolo_dist_destroy must not look into a context the caller may already have destroyed
    int world = 1, rank = 0, global_batch = 0, max_out = 0, per = 0, first = 0, count = 0;
    Comm comm = nullptr; bool own_comm = false;
    int32_t *d_rec = nullptr, *d_all = nullptr, *h_all = nullptr;      // [flat], [world][flat] device; [world][flat] pinned host
    size_t flat = 0;
};

extern "C" {

int yolo_shard_bounds(int global_batch, int world_size, int rank, int *first, int *count)
{
    if (global_batch < 0 || world_size < 1 || rank < 0 || rank >= world_size) return YOLO_ERR_INVALID;
    const int base = global_batch / world_size, rem = global_batch % world_size;
    if (first) *first = rank * base + std::min(rank, rem);
    if (count) *count = base + (rank < rem ? 1 : 0);
    return YOLO_OK;
}

size_t yolo_dist_flat_words(int per, int max_out)
{
    if (per < 0 || max_out < 0) return 0;
    return (size_t)per * max_out * kRecordWords + per;
}

int yolo_dist_split_records(const int32_t *gathered, int world_size, int global_batch, int max_out, yolo_box *boxes_out, int32_t *counts_out)
{
    if (!gathered || !boxes_out || !counts_out || world_size < 1 || global_batch < 0 || max_out < 1) return YOLO_ERR_INVALID;
    const int per = (global_batch + world_size - 1) / world_size;
    const size_t flat = yolo_dist_flat_words(per, max_out);
    for (int r = 0; r < world_size; ++r) {
        int first = 0, count = 0;
        yolo_shard_bounds(global_batch, world_size, r, &first, &count);
        const int32_t *rec = gathered + (size_t)r * flat;
        memcpy(boxes_out + (size_t)first * max_out, rec, (size_t)count * max_out * sizeof(yolo_box));
        memcpy(counts_out + first, rec + (size_t)per * max_out * kRecordWords, (size_t)count * sizeof(int32_t));
    }
    return YOLO_OK;
}

int yolo_dist_unique_id(uint8_t id[128])
{
    if (!id) return YOLO_ERR_INVALID;
    const Rccl &r = rccl();
    if (!r.so) return YOLO_ERR_UNSUPPORTED;
    UniqueId u;
    if (r.GetUniqueId(&u) != 0) return YOLO_ERR_HIP;
    memcpy(id, u.internal, 128);
    return YOLO_OK;
}

yolo_dist *yolo_dist_create(yolo_ctx *ctx, int world_size, int rank, const uint8_t id[128], void *comm, int global_batch, int max_out,
                            char *err, size_t err_len)
{
    const ErrBuf eb{err, err_len};
    if (!ctx) return bad(eb, "yolo_dist_create: no context");
    int first = 0, count = 0;
    if (yolo_shard_bounds(global_batch, world_size, rank, &first, &count) != YOLO_OK || global_batch < 1 || max_out < 1)
        return bad(eb, "yolo_dist_create: bad world_size %d / rank %d / global_batch %d / max_out %d", world_size, rank, global_batch, max_out);
    const int per = (global_batch + world_size - 1) / world_size;
    if (per > ctx->max_batch) return bad(eb, "yolo_dist_create: %d images per rank, the context was planned for %d", per, ctx->max_batch);
    if (!comm && !id) return bad(eb, "yolo_dist_create needs the caller's ncclComm_t or the 128-byte id of yolo_dist_unique_id");
    const Rccl &r = rccl();
    if (!r.so) return bad(eb, "yolo_dist_create: %s", r.why.c_str());
    if (hipSetDevice(ctx->device) != hipSuccess) return bad(eb, "yolo_dist_create: hipSetDevice(%d) failed", ctx->device);
    yolo_dist *d = new yolo_dist;
    d->ctx = ctx; d->device = ctx->device; d->world = world_size; d->rank = rank; d->global_batch = global_batch; d->max_out = max_out;
    d->per = per; d->first = first; d->count = count; d->flat = yolo_dist_flat_words(per, max_out);
    // The communicator FIRST: ncclCommInitRank is collective, and a rank that returned on a failed allocation before joining it would leave
    // the others waiting in theirs.
    if (comm) d->comm = comm;
    else {
        UniqueId u; memcpy(u.internal, id, 128);
        const int e = r.CommInitRank(&d->comm, world_size, u, rank);
        if (e != 0) { d->comm = nullptr; const char *s = r.GetErrorString(e); yolo_dist_destroy(d); return bad(eb, "ncclCommInitRank: %s", s ? s : "?"); }
        d->own_comm = true;
    }
    const size_t bytes = d->flat * sizeof(int32_t);
    // the record buffer is zeroed once: the padding rows of a short rank (count < per) are never written and travel as zeros
    if (hipMalloc((void **)&d->d_rec, bytes) != hipSuccess || hipMemset(d->d_rec, 0, bytes) != hipSuccess ||
        hipMalloc((void **)&d->d_all, bytes * world_size) != hipSuccess || hipHostMalloc((void **)&d->h_all, bytes * world_size) != hipSuccess) {
        (void)hipGetLastError(); yolo_dist_destroy(d); return bad(eb, "yolo_dist_create: out of memory (%zu bytes x %d ranks)", bytes, world_size);
    }
    return d;
}

int yolo_dist_detect_async(yolo_dist *d, const void *images, int fmt, float scale, float score_thr, float iou_thr, int nms_mode,
                           int select_mode, const int32_t **gathered_dev)
{
    if (!d || !d->ctx) return YOLO_ERR_INVALID;
    yolo_ctx *c = d->ctx;
    HIPCK(c, hipSetDevice(c->device));
    // A rank whose LOCAL step fails still joins the exchange (its records zeroed: counts 0), and reports the failure afterwards: leaving a
    // collective early would hang every other rank.
    int local = YOLO_OK;
    if (d->count > 0) {
        yolo_box *boxes = (yolo_box *)d->d_rec;
        int32_t *counts = d->d_rec + (size_t)d->per * d->max_out * kRecordWords;
        if (!images) local = fail(c, YOLO_ERR_INVALID, "yolo_dist_detect: no images for this rank's %d-image slice", d->count);
        else local = yolo_detect_graph(c, images, d->count, fmt, scale, score_thr, iou_thr, d->max_out, nms_mode, select_mode, boxes, counts);
        if (local != YOLO_OK) { (void)hipGetLastError(); (void)hipMemsetAsync(d->d_rec, 0, d->flat * sizeof(int32_t), c->stream); }
    }
    const std::string local_msg = local != YOLO_OK ? std::string(yolo_last_error(c)) : std::string();
    const int e = rccl().AllGather(d->d_rec, d->d_all, d->flat, kNcclInt32, d->comm, c->stream);
    if (local != YOLO_OK) return fail(c, local, "%s (this rank joined the exchange with zero records)", local_msg.c_str());
    if (e != 0) { const char *s = rccl().GetErrorString(e); return fail(c, YOLO_ERR_HIP, "ncclAllGather: %s", s ? s : "?"); }
    if (gathered_dev) *gathered_dev = d->d_all;
    return YOLO_OK;
}

int yolo_dist_detect(yolo_dist *d, const void *images, int fmt, float scale, float score_thr, float iou_thr, int nms_mode, int select_mode,
                     yolo_box *boxes_all, int32_t *counts_all)
{
    if (!d || !d->ctx) return YOLO_ERR_INVALID;
    yolo_ctx *c = d->ctx;
    if (!boxes_all || !counts_all) return fail(c, YOLO_ERR_INVALID, "yolo_dist_detect needs boxes_all and counts_all");
    if (int r = yolo_dist_detect_async(d, images, fmt, scale, score_thr, iou_thr, nms_mode, select_mode, nullptr)) return r;
    HIPCK(c, hipMemcpyAsync(d->h_all, d->d_all, d->flat * sizeof(int32_t) * d->world, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return yolo_dist_split_records(d->h_all, d->world, d->global_batch, d->max_out, boxes_all, counts_all);
}

void yolo_dist_destroy(yolo_dist *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->own_comm && d->comm && rccl().so) rccl().CommDestroy(d->comm);
    if (d->d_rec) (void)hipFree(d->d_rec);
    if (d->d_all) (void)hipFree(d->d_all);
    if (d->h_all) (void)hipHostFree(d->h_all);
    delete d;
}

}  // extern "C"

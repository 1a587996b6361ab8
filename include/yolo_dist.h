/* libyolo_hip.so -- the image-sharded detect step behind the C ABI.
 *
 * The reference has no inference-time parallelism to mirror (one `tf.Session` on one device,
 * V3/YOLO_V3_inference.py:97-107; its only multi-GPU code averages weights during training,
 * DN/network.c:857-1121).  The path shards by image: every rank holds a full replica of the folded
 * weights in its own yolo_ctx, runs conv stack + decode + threshold + NMS on a contiguous slice of
 * the global batch, and the ranks exchange ONE message per step -- each rank's fixed-capacity record
 * buffer ([per][max_out] yolo_box, then [per] int32 counts; per = ceil(global_batch / world_size)) in
 * an all-gather over RCCL.  The Python host does the same with torch.distributed
 * (yolo_tensorflow_amd/dist.py: same split, same buffer layout); this header is what a C / cgo / JNI
 * caller binds instead.
 *
 * RCCL is resolved at run time (dlopen of $YOLO_RCCL_LIB, librccl.so.1, librccl.so): the library has
 * no link-time dependency on it and every other entry point works without it.
 */
#ifndef YOLO_DIST_H
#define YOLO_DIST_H

#include "yolo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yolo_dist yolo_dist;

/* Contiguous split of the global batch: rank r serves images [*first, *first + *count); the earlier
 * ranks take the remainder (world 3, batch 8: 3 + 3 + 2).  Host arithmetic, no device needed. */
int yolo_shard_bounds(int global_batch, int world_size, int rank, int *first, int *count);

/* int32 words of one rank's record buffer for `per` images: per * max_out * 6 (records) + per (counts). */
size_t yolo_dist_flat_words(int per, int max_out);

/* gathered: [world_size][yolo_dist_flat_words(per, max_out)] int32 as the all-gather leaves it (host).
 * Writes the global batch in image order: boxes_out [global_batch * max_out], counts_out
 * [global_batch]; the padding rows of the short ranks are dropped.  Host arithmetic. */
int yolo_dist_split_records(const int32_t *gathered, int world_size, int global_batch, int max_out,
                            yolo_box *boxes_out, int32_t *counts_out);

/* ncclGetUniqueId for callers without rccl.h: 128 bytes that rank 0 creates and hands to the other
 * ranks by its own means (environment, file, MPI, a socket). */
int yolo_dist_unique_id(uint8_t id[128]);

/* Binds `ctx` (this rank's context, max_batch >= per) to a communicator of world_size ranks: either
 * the caller's own `comm` (an ncclComm_t made on ctx's device; the caller keeps ownership), or, when
 * comm is NULL, one initialised here from `id` (ncclCommInitRank: collective over all ranks, destroyed
 * by yolo_dist_destroy).  Allocates the record buffer, the gather buffer and a pinned host mirror.
 * On failure returns NULL and writes a message to err.  The communicator is joined before anything is
 * allocated, so a rank that fails locally does not leave the others inside ncclCommInitRank.
 * Life time: destroy the yolo_dist BEFORE its context (a step uses the context; yolo_dist_destroy itself
 * only needs the device number, which it keeps). */
yolo_dist *yolo_dist_create(yolo_ctx *ctx, int world_size, int rank, const uint8_t id[128], void *comm,
                            int global_batch, int max_out, char *err, size_t err_len);

/* One step.  images: THIS rank's slice ([count] images in `fmt`, device-resident, stable between
 * calls: the step is replayed from a HIP graph as yolo_detect_graph does).  Runs the local detect
 * step into the record buffer, all-gathers it on the context's stream, and returns the WHOLE batch
 * in image order on every rank: boxes_all [global_batch * max_out], counts_all [global_batch] (host).
 * Arguments as yolo_detect. */
int yolo_dist_detect(yolo_dist *d, const void *images, int fmt, float scale, float score_thr, float iou_thr,
                     int nms_mode, int select_mode, yolo_box *boxes_all, int32_t *counts_all);

/* The same without the host copy: the step is enqueued and the gathered buffer stays on the device
 * ([world_size][flat words] int32, valid until the next step) -- for callers that consume the records
 * on the GPU or overlap the copy themselves.  *gathered_dev receives the pointer. */
int yolo_dist_detect_async(yolo_dist *d, const void *images, int fmt, float scale, float score_thr, float iou_thr,
                           int nms_mode, int select_mode, const int32_t **gathered_dev);

void yolo_dist_destroy(yolo_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_DIST_H */

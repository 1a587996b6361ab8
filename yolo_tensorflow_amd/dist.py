"""Multi-GPU data parallelism for the detect path: one process per GPU, images sharded by rank, ONE
exchange step -- an all-gather of fixed-capacity detection records.

The reference has no inference-time parallelism at all (single `tf.Session`, single device; its only
multi-GPU code is host-mediated weight averaging for training, DN/network.c:857-1121), so there is no call
pattern to mirror: each rank holds a full replica of the folded weights, runs conv stack + decode + NMS on
its own images, and the ranks exchange `[B/G, max_out] x 24-byte records + [B/G] counts` (a few KB; latency
bound).  `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import numpy as np

RECORD_FLOATS = 6        # x0, y0, x1, y1, score, class(as int32 bits)


def shard_bounds(global_batch, world_size, rank):
    """Contiguous split of the global batch; earlier ranks take the remainder."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(boxes, counts, max_out):
    """structured yolo_box array [n,max_out] + counts [n] -> one int32 buffer [n, 1 + max_out*6] (count first),
    so a single collective moves both."""
    n = boxes.shape[0]
    buf = np.zeros((n, 1 + max_out * RECORD_FLOATS), dtype=np.int32)
    buf[:, 0] = counts
    buf[:, 1:] = boxes.view(np.int32).reshape(n, max_out * RECORD_FLOATS)
    return buf


def unpack_records(buf, box_dtype, max_out):
    n = buf.shape[0]
    counts = buf[:, 0].copy()
    boxes = np.ascontiguousarray(buf[:, 1:]).view(box_dtype).reshape(n, max_out)
    return [boxes[i, :counts[i]].copy() for i in range(n)]


def all_gather_detections(local_buf, global_batch, group=None):
    """local_buf: torch int32 tensor [n_local, 1 + max_out*6] on the rank's device (CPU for gloo).
    Every rank contributes a block padded to ceil(B/G) rows (all_gather needs equal sizes); returns the
    [global_batch, ...] tensor in global image order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = -(-global_batch // world)
    width = local_buf.shape[1]
    padded = torch.zeros((per, width), dtype=local_buf.dtype, device=local_buf.device)
    padded[:local_buf.shape[0]] = local_buf
    out = torch.empty((world * per, width), dtype=local_buf.dtype, device=local_buf.device)
    dist.all_gather_into_tensor(out, padded, group=group) if hasattr(dist, "all_gather_into_tensor") and local_buf.is_cuda \
        else _all_gather_list(out, padded, world, per, group)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(global_batch, world, r)
        rows.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(rows, dim=0)


def _all_gather_list(out, padded, world, per, group):
    import torch.distributed as dist
    parts = [out[r * per:(r + 1) * per] for r in range(world)]
    dist.all_gather(parts, padded, group=group)


# ---- in-place exchange used on the hot path (bench.py): the library writes box records and counts straight into ONE
#      flat int32 buffer per rank, which is all-gathered as is (equal shards) ----
def alloc_flat_records(n_local, max_out, device):
    """-> (rec, boxes, counts): rec int32 [n_local*max_out*6 + n_local]; boxes = rec's leading [n_local, max_out*6] view
    (what yolo_detect* fills as yolo_box records), counts = its trailing [n_local] view."""
    import torch
    rec = torch.zeros((n_local * max_out * RECORD_FLOATS + n_local,), dtype=torch.int32, device=device)
    return rec, rec[:n_local * max_out * RECORD_FLOATS].view(n_local, max_out * RECORD_FLOATS), rec[n_local * max_out * RECORD_FLOATS:]


def gather_flat_records(rec, out=None, group=None):
    """One collective: every rank's flat buffer -> [world, len(rec)] on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, rec.numel()), dtype=rec.dtype, device=rec.device)
    if rec.is_cuda:
        dist.all_gather_into_tensor(out, rec, group=group)
    else:
        dist.all_gather([out[r] for r in range(world)], rec, group=group)
    return out


class PipelinedGather(object):
    """The same exchange taken off the critical path: step n's records are copied (a few KB, on the compute stream) into a staging
    buffer and gathered from there asynchronously -- RCCL runs the collective on its own stream -- while step n+1 computes into the
    library's output buffer again; the collective is waited for one step later.  Every step is still exchanged; `result()` returns
    the gathered records of the last submitted step (it makes the current stream wait for that collective), so a consumer sees them
    one step behind the compute.

    Staging and output buffers are DOUBLE-BUFFERED (alternating per submit): the tensor `result()` returns is not touched by the
    next `submit()`, only by the one after it -- a consumer may keep reading it (on the current stream, or after a stream sync on
    the host) while the next step's collective is in flight."""

    def __init__(self, rec, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.stage = [torch.empty_like(rec) for _ in range(2)]
        self.out = [torch.empty((self.world, rec.numel()), dtype=rec.dtype, device=rec.device) for _ in range(2)]
        self.work = [None, None]
        self.k = 0            # buffer pair the NEXT submit uses
        self.last = None      # buffer pair of the last submit

    def submit(self, rec):
        import torch.distributed as dist
        k = self.k
        if self.work[k] is not None:
            self.work[k].wait(); self.work[k] = None      # the gather that used this pair two submits ago (stream-ordered for RCCL)
        self.stage[k].copy_(rec, non_blocking=True)
        if rec.is_cuda:
            self.work[k] = dist.all_gather_into_tensor(self.out[k], self.stage[k], group=self.group, async_op=True)
        else:
            self.work[k] = dist.all_gather([self.out[k][r] for r in range(self.world)], self.stage[k], group=self.group, async_op=True)
        self.last = k; self.k = 1 - k

    def result(self):
        """Gathered records [world, len(rec)] of the last submitted step; valid until the submit after the next one."""
        if self.last is None:
            return None
        k = self.last
        if self.work[k] is not None:
            self.work[k].wait(); self.work[k] = None
        return self.out[k]


class HostStagedGather(object):
    """PipelinedGather's interface for a process group WITHOUT device collectives (backend "gloo": bench.py's functional multi-rank mode, two
    or more ranks sharing one GPU, BENCH_BACKEND=gloo): step n's records are copied device -> pinned host on the current stream, and the
    collective -- `all_gather` of the host buffers -- is started one submit later, when that copy has long completed, and waited for at the
    submit after that (or by `result()`).  Same double buffering, same one-step-behind contract; `result()` returns a HOST tensor."""

    def __init__(self, rec, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.stage = [torch.empty(rec.shape, dtype=rec.dtype).pin_memory() if rec.is_cuda else torch.empty_like(rec) for _ in range(2)]
        self.out = [torch.empty((self.world, rec.numel()), dtype=rec.dtype) for _ in range(2)]
        self.copied = [torch.cuda.Event() if rec.is_cuda else None for _ in range(2)]
        self.state = [0, 0]       # 0 free, 1 copy enqueued, 2 collective in flight
        self.work = [None, None]
        self.k = 0; self.last = None

    def _start(self, k):
        import torch.distributed as dist
        if self.state[k] == 1:
            if self.copied[k] is not None:
                self.copied[k].synchronize()
            self.work[k] = dist.all_gather([self.out[k][r] for r in range(self.world)], self.stage[k], group=self.group, async_op=True)
            self.state[k] = 2

    def _finish(self, k):
        self._start(k)
        if self.state[k] == 2:
            self.work[k].wait(); self.work[k] = None; self.state[k] = 0

    def submit(self, rec):
        k = self.k
        self._finish(k)                     # the exchange that used this pair two submits ago
        if self.last is not None:
            self._start(self.last)          # the previous step's copy is done by now: its collective runs under this step
        self.stage[k].copy_(rec, non_blocking=True)
        if self.copied[k] is not None:
            self.copied[k].record()
        self.state[k] = 1
        self.last = k; self.k = 1 - k

    def result(self):
        if self.last is None:
            return None
        self._finish(1 - self.last); self._finish(self.last)
        return self.out[self.last]


def split_flat_records(rec_all, n_local, max_out):
    """[world, flat] -> (boxes [world*n_local, max_out*6], counts [world*n_local]) in rank (= image) order."""
    world = rec_all.shape[0]
    nb = n_local * max_out * RECORD_FLOATS
    return rec_all[:, :nb].reshape(world * n_local, max_out * RECORD_FLOATS), rec_all[:, nb:].reshape(world * n_local)


def split_flat_records_ragged(rec_all, global_batch, max_out):
    """Strong scaling (bench.py --global-batch): every rank's flat buffer is sized for per = ceil(global_batch / world) images and rank r
    filled the first shard_bounds(global_batch, world, r) of them.  [world, flat] -> (boxes [global_batch, max_out*6], counts
    [global_batch]) in global image order (the padding rows of the short ranks are dropped)."""
    import torch
    world = rec_all.shape[0]
    per = -(-global_batch // world)
    boxes, counts = split_flat_records(rec_all, per, max_out)
    keep = []
    for r in range(world):
        lo, hi = shard_bounds(global_batch, world, r)
        keep.append(torch.arange(r * per, r * per + (hi - lo), device=rec_all.device))
    keep = torch.cat(keep)
    return boxes[keep], counts[keep]

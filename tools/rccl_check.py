"""RCCL exchange check on real GPUs: every rank runs the detect path on its shard of a global batch (device-resident images,
box records written in place by the library), then ONE all_gather_into_tensor of the flat record buffers
(yolo_tensorflow_amd/dist.py, the call bench.py makes); every rank must end up with exactly the records a single engine
produces for the whole batch.  The same step is then run behind the C ABI (include/yolo_dist.h: yolo_dist_create /
yolo_dist_detect on a communicator the library initialises itself) on a ragged global batch.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_check.py

Fabric stress mode (the raw-tensor all-gather used to MEASURE xGMI, never on the product path): every rank
contributes its [B/G, rows, 85] bf16 decoded tensor (YOLOv3 416: 10647 rows -> 1.8 MB per image) to one all_gather_into_tensor;
prints bytes, time and the per-link rate of a ring over xGMI (7 links x ~153 GB/s per GPU), next to the box-record exchange:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_check.py --stress [--batch 64] [--iters 20]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run_check(rank, world, local_rank, n_local=4, size=160, max_out=10):
    import torch
    import torch.distributed as dist
    from yolo_tensorflow_amd import hip, darknet_io as IO, dist as ydist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    txt = IO.with_input_size(IO.cfg_text("yolov3-tiny"), size)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=1, obj_bias=1.0)
    all_img = np.random.default_rng(7).integers(0, 256, (n_local * world, size, size, 3), dtype=np.uint8)
    lo, hi = ydist.shard_bounds(n_local * world, world, rank)
    stream = torch.cuda.current_stream(dev)
    eng = hip.Engine(txt, max_batch=n_local * world, device=local_rank, stream=stream.cuda_stream)
    eng.set_weights(flat)
    # reference: the whole global batch on this rank's engine, host round trip
    eng.forward(all_img, want_detections=False)
    want = eng.postprocess(n_local * world, score_thr=0.3, iou_thr=0.5, max_out=max_out)
    assert sum(len(w) for w in want) > 0
    # sharded: device images in, records written in place, one collective
    images = torch.from_numpy(all_img[lo:hi]).to(dev)
    rec, boxes, counts = ydist.alloc_flat_records(n_local, max_out, dev)
    for _ in range(3):                     # eager, capture, replay
        eng.detect_graph(images, boxes, counts, score_thr=0.3, iou_thr=0.5, max_out=max_out)
        rec_all = ydist.gather_flat_records(rec)
    torch.cuda.synchronize(dev)
    gb, gc = ydist.split_flat_records(rec_all, n_local, max_out)
    gb = gb.cpu().numpy().view(hip.BOX_DTYPE).reshape(n_local * world, max_out); gc = gc.cpu().numpy()
    for i in range(n_local * world):
        assert gc[i] == len(want[i]) and np.array_equal(gb[i, :gc[i]], want[i]), "rank %d: image %d differs after the all-gather" % (rank, i)
    # the pipelined form bench.py uses: staged copy + asynchronous collective, waited for one step later
    pg = ydist.PipelinedGather(rec)
    for _ in range(3):
        eng.detect_graph(images, boxes, counts, score_thr=0.3, iou_thr=0.5, max_out=max_out)
        pg.submit(rec)
    rec_p = pg.result()
    torch.cuda.synchronize(dev)
    assert torch.equal(rec_p, rec_all), "rank %d: pipelined gather differs from the blocking one" % rank
    # the ragged-shard form (padded blocks) on device tensors too
    buf = torch.from_numpy(ydist.pack_records(gb[lo:hi], gc[lo:hi].astype(np.int32), max_out)).to(dev)
    full = ydist.all_gather_detections(buf, n_local * world).cpu().numpy()
    dets = ydist.unpack_records(full, hip.BOX_DTYPE, max_out)
    for i in range(n_local * world):
        assert np.array_equal(dets[i], want[i])
    # the same step behind the C ABI (include/yolo_dist.h): the library's own RCCL communicator, a RAGGED global batch (one image
    # fewer than the ranks can hold), the 128-byte id handed out through the process group
    B = n_local * world - (1 if world > 1 else 0)
    ids = [hip.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    sd = hip.ShardedDetector(eng, world, rank, ids[0], global_batch=B, max_out=max_out)
    clo, chi = hip.shard_bounds(B, world, rank)
    cimg = torch.from_numpy(all_img[clo:chi]).to(dev)
    for _ in range(3):
        got = sd.detect(cimg, score_thr=0.3, iou_thr=0.5)
    for i in range(B):
        assert np.array_equal(got[i], want[i]), "rank %d: image %d differs after yolo_dist_detect" % (rank, i)
    sd.close()
    eng.close()
    return True


def run_stress(rank, world, local_rank, global_batch=64, rows=10647, attrs=85, iters=20, max_out=20):
    """All-gather of the raw bf16 decoded tensors (what a naive multi-GPU detector would exchange) against the 24-byte box records the
    product exchanges: one line per rank 0 with both timings.  Data is random: this measures the fabric, not the network."""
    import torch
    import torch.distributed as dist
    from yolo_tensorflow_amd import dist as ydist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lo, hi = ydist.shard_bounds(global_batch, world, rank)
    per = -(-global_batch // world)
    local = torch.randn((per, rows, attrs), device=dev).to(torch.bfloat16)
    out = torch.empty((world * per, rows, attrs), dtype=torch.bfloat16, device=dev)
    rec, _, _ = ydist.alloc_flat_records(per, max_out, dev)
    rec_all = torch.empty((world, rec.numel()), dtype=rec.dtype, device=dev)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize(dev)
        t = torch.tensor([e0.elapsed_time(e1) / iters], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ms_raw = timed(lambda: dist.all_gather_into_tensor(out, local))
    ms_rec = timed(lambda: dist.all_gather_into_tensor(rec_all, rec))
    # every rank holds every rank's block afterwards
    chk = torch.stack([out[r * per] .float().abs().sum() for r in range(world)])
    assert torch.isfinite(chk).all() and (chk > 0).all()
    if rank == 0:
        raw_bytes = local.numel() * 2
        # ring all-gather: every GPU sends (and receives) (world - 1) blocks over its ring link
        per_link = (world - 1) * raw_bytes / (ms_raw * 1e-3) / 1e9 if world > 1 else 0.0
        print("rccl_check stress: world %d, global batch %d | raw decoded tensors: %.1f MB per rank, all-gather %.3f ms (%.1f GB/s per ring "
              "link; xGMI link ~153 GB/s) | box records: %d B per rank, all-gather %.3f ms" % (world, global_batch, raw_bytes / 1e6, ms_raw, per_link, rec.numel() * 4, ms_rec))
    return ms_raw, ms_rec


if __name__ == "__main__":
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lr = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    if "--stress" in sys.argv:
        arg = lambda k, d: int(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
        run_stress(rank, world, lr, global_batch=arg("--batch", 64), iters=arg("--iters", 20))
    else:
        run_check(rank, world, lr)
    dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print("rccl_check ok: world %d" % world)

"""The N>1 path on CPU: two processes, gloo backend, sharded batch, one all-gather of detection records."""
import os
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from yolo_tensorflow_amd import dist as ydist
from yolo_tensorflow_amd.hip import BOX_DTYPE


def test_shard_bounds_cover_batch():
    for B in (1, 7, 32, 64):
        for G in (1, 2, 3, 8):
            spans = [ydist.shard_bounds(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _fake_boxes(img_index, max_out):
    rng = np.random.default_rng(1000 + img_index)
    cnt = int(rng.integers(0, max_out + 1))
    b = np.zeros(max_out, dtype=BOX_DTYPE)
    b["x0"][:cnt] = rng.random(cnt); b["y0"][:cnt] = rng.random(cnt); b["x1"][:cnt] = rng.random(cnt); b["y1"][:cnt] = rng.random(cnt)
    b["score"][:cnt] = rng.random(cnt); b["cls"][:cnt] = rng.integers(0, 80, cnt)
    return b, cnt


def _worker(rank, world, port, B, max_out, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ydist.shard_bounds(B, world, rank)
    boxes = np.zeros((hi - lo, max_out), dtype=BOX_DTYPE); counts = np.zeros(hi - lo, np.int32)
    for i in range(lo, hi):
        boxes[i - lo], counts[i - lo] = _fake_boxes(i, max_out)
    buf = torch.from_numpy(ydist.pack_records(boxes, counts, max_out))
    full = ydist.all_gather_detections(buf, B)
    q.put((rank, full.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("B,world", [(7, 2), (30, 4)])      # ragged splits: 4 + 3; 8 + 8 + 7 + 7 (batch not a multiple of the GPU count)
def test_gather_matches_single_process(B, world):
    max_out = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17 * world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, max_out, q)) for r in range(world)]
    for p in procs: p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(results[0], results[1]) and results[0].shape == (B, 1 + max_out * 6)
    dets = ydist.unpack_records(results[0], BOX_DTYPE, max_out)
    for i in range(B):
        want, cnt = _fake_boxes(i, max_out)
        assert len(dets[i]) == cnt and np.array_equal(dets[i], want[:cnt])


def _flat_worker(rank, world, port, n_local, max_out, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec, boxes, counts = ydist.alloc_flat_records(n_local, max_out, "cpu")
    for i in range(n_local):                                  # what the library would have written in place
        b, cnt = _fake_boxes(rank * n_local + i, max_out)
        boxes[i] = torch.from_numpy(b.view(np.int32).reshape(-1)); counts[i] = cnt
    rec_all = ydist.gather_flat_records(rec)
    gb, gc = ydist.split_flat_records(rec_all, n_local, max_out)
    q.put((rank, gb.numpy().copy(), gc.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_flat_record_exchange():
    """The hot-path exchange of bench.py: equal shards, ONE collective on the flat [records | counts] buffer."""
    n_local, max_out, world = 3, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, n_local, max_out, q)) for r in range(world)]
    for p in procs: p.start()
    results = {r: (b, c) for r, b, c in (q.get(timeout=120) for _ in range(world))}
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    gb, gc = results[0]
    assert gb.shape == (world * n_local, max_out * 6) and gc.shape == (world * n_local,)
    for i in range(world * n_local):
        want, cnt = _fake_boxes(i, max_out)
        assert gc[i] == cnt and np.array_equal(gb[i], want.view(np.int32).reshape(-1))


def _pipe_worker(rank, world, port, n_local, max_out, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rec, boxes, counts = ydist.alloc_flat_records(n_local, max_out, "cpu")
    g = ydist.PipelinedGather(rec)
    seen = []
    for step in range(3):                                     # the library overwrites `rec` every step; each step is exchanged
        for i in range(n_local):
            b, cnt = _fake_boxes(100 * step + rank * n_local + i, max_out)
            boxes[i] = torch.from_numpy(b.view(np.int32).reshape(-1)); counts[i] = cnt
        g.submit(rec)
        rec.zero_()                                           # the next step's compute may clobber `rec` while the gather is in flight
        if step == 1:
            held = g.result()                                 # a consumer takes step 1's records and KEEPS the tensor (no copy) across
                                                              # the next submit: the buffers alternate, so step 2's gather must not touch it
    seen.append(g.result().clone())
    seen.insert(0, held.clone())
    q.put((rank, [t.numpy() for t in seen]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_pipelined_gather():
    """bench.py's exchange for N > 1: copy to a staging buffer, asynchronous all-gather, waited for one step later."""
    n_local, max_out, world = 2, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, n_local, max_out, q)) for r in range(world)]
    for p in procs: p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    for k, step in enumerate((1, 2)):
        assert np.array_equal(results[0][k], results[1][k])
        gb, gc = ydist.split_flat_records(torch.from_numpy(results[0][k]), n_local, max_out)
        for i in range(world * n_local):
            want, cnt = _fake_boxes(100 * step + i, max_out)
            assert gc[i] == cnt and np.array_equal(gb[i].numpy(), want.view(np.int32).reshape(-1))



def _strong_worker(rank, world, port, GB, max_out, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = -(-GB // world)
    lo, hi = ydist.shard_bounds(GB, world, rank)
    rec, boxes, counts = ydist.alloc_flat_records(per, max_out, "cpu")       # every rank's buffer is sized for the largest share
    for i in range(lo, hi):                                                   # the library fills the first hi - lo images in place
        b, cnt = _fake_boxes(i, max_out)
        boxes[i - lo] = torch.from_numpy(b.view(np.int32).reshape(-1)); counts[i - lo] = cnt
    g = ydist.PipelinedGather(rec); g.submit(rec)
    gb, gc = ydist.split_flat_records_ragged(g.result(), GB, max_out)
    q.put((rank, gb.numpy().copy(), gc.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("GB,world", [(32, 2), (7, 3)])      # bench.py --global-batch: 16 + 16; ragged 3 + 2 + 2
def test_strong_scaling_split_of_a_fixed_global_batch(GB, world):
    """bench.py's strong-scaling mode: a fixed global batch split contiguously, every rank exchanging a buffer sized for the largest
    share; the gathered records come back in global image order with the short ranks' padding rows dropped."""
    max_out = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() + 13 * world) % 2000
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, GB, max_out, q)) for r in range(world)]
    for p in procs: p.start()
    results = {r: (b, c) for r, b, c in (q.get(timeout=120) for _ in range(world))}
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    gb, gc = results[0]
    assert gb.shape == (GB, max_out * 6) and gc.shape == (GB,)
    for r in range(1, world):
        assert np.array_equal(results[r][0], gb) and np.array_equal(results[r][1], gc)
    for i in range(GB):
        want, cnt = _fake_boxes(i, max_out)
        assert gc[i] == cnt and np.array_equal(gb[i], want.view(np.int32).reshape(-1))

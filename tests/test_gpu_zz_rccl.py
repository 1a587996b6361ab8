"""The RCCL exchange of the detect path on a real GPU: `nccl` backend (RCCL on ROCm), device tensors, the same
`dist.gather_flat_records` call bench.py makes for N > 1.  World size 1 runs in this process on any box; the 2-rank variant
needs two GPUs and is launched by `python -m torch.distributed.run --nproc-per-node 2 tools/rccl_check.py` (a pytest process
that has initialised the GPU must not spawn/exec children on this pool).  Named zz so that it runs after the other GPU tests:
it creates and destroys the default process group."""
import json
import os
import sys

import numpy as np
import pytest
from yolo_tensorflow_amd import darknet_io as IO, dist as ydist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_sharded_detect_world1(hiplib):
    """include/yolo_dist.h on the GPU: yolo_dist_unique_id -> yolo_dist_create (ncclCommInitRank through the library's run-time RCCL
    binding) -> yolo_dist_detect (graph-replayed local step into the flat record buffer, ncclAllGather on the context's stream, split on
    the host) returns exactly what the plain detect step returns.  World size 1 is the only world a one-GPU box offers; the split of a
    gathered buffer over many ranks is covered on the CPU (tests/test_host.py) against the torch.distributed host."""
    import torch
    txt = IO.with_input_size(IO.cfg_text("yolov3"), 96)
    secs = IO.parse_cfg(txt); flat = IO.synth_weights(secs, seed=12)
    img = np.random.default_rng(16).integers(0, 256, (3, 96, 96, 3), dtype=np.uint8)
    eng = hiplib.Engine(txt, max_batch=4)
    eng.set_weights(flat)
    want = eng.detect(img, score_thr=0.3, iou_thr=0.5, max_out=20, nms_mode=hiplib.NMS_TF)
    assert sum(len(w) for w in want) > 5
    sd = hiplib.ShardedDetector(eng, 1, 0, hiplib.dist_unique_id(), global_batch=3, max_out=20)
    dimg = torch.from_numpy(img).cuda()
    for _ in range(4):                                            # eager, capture, replays
        got = sd.detect(dimg, score_thr=0.3, iou_thr=0.5, nms_mode=hiplib.NMS_TF)
    assert len(got) == 3 and all(np.array_equal(g, w) for g, w in zip(got, want))
    with pytest.raises(hiplib.YoloError, match="images"):
        sd.detect(dimg[:2], score_thr=0.3)
    sd.close()
    with pytest.raises(hiplib.YoloError, match="planned for"):
        hiplib.ShardedDetector(eng, 1, 0, hiplib.dist_unique_id(), global_batch=5)
    eng.close()


def test_rccl_gather_of_device_records_world1(hiplib):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.fail("torch does not see the GPU")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rccl_check
    port = 29700 + os.getpid() % 1000
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rccl_check.run_check(rank=0, world=1, local_rank=0)
        ms_raw, ms_rec = rccl_check.run_stress(rank=0, world=1, local_rank=0, global_batch=4, iters=3)      # the fabric stress mode's code path
        assert ms_raw > 0 and ms_rec > 0
    finally:
        dist.destroy_process_group()


def test_bench_force_dist_path(hiplib, capsys, monkeypatch):
    """bench.py's own N > 1 code path (process group, all_gather_into_tensor of the flat record buffer inside the timed loop)
    at world size 1: BENCH_FORCE_DIST=1, in process."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("torch does not see the GPU")
    monkeypatch.setenv("BENCH_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_PORT", str(30700 + os.getpid() % 1000))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"])
    sys.path.insert(0, ROOT)
    import bench
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["achieved"] > 0


def test_bench_strong_scaling_and_parity_fields(hiplib, capsys, monkeypatch):
    """bench.py --global-batch (strong scaling: a fixed global batch split over the ranks) through the same forced distributed path, and the
    `parity` object of the line: the timed configuration's boxes against the fp32 oracle on the first images of the timed batch."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("torch does not see the GPU")
    monkeypatch.setenv("BENCH_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_PORT", str(31700 + os.getpid() % 1000))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--global-batch", "8", "--parity-images", "1"])
    sys.path.insert(0, ROOT)
    import bench
    bench.main()
    out = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 8 and "global batch=8" in out["config"]["workload"]
    assert out["value"] > 0 and abs(out["value"] - 8 * 1e3 / out["ms_per_step"]) / out["value"] < 0.01
    p = out["parity"]
    assert p["images"] == 1 and p["candidates"] > 20 and p["min_iou"] >= 0.99 and p["max_dscore"] <= 1e-2 and p["lost"] == 0


def test_bench_two_ranks_one_gpu(hiplib, tmp_path):
    """bench.py's rank > 0 branch on a real device.  The GPU boxes have ONE GPU, so two ranks are launched by the
    contract's launcher (`python -m torch.distributed.run --nproc-per-node 2 ...`) with BENCH_BACKEND=gloo: both ranks compute on cuda:0,
    the box records are exchanged through pinned host buffers by gloo (dist.HostStagedGather).  A FUNCTIONAL run -- shard bounds of a ragged
    global batch (33 = 17 + 16), per-rank seeds, the pipelined exchange and its ordering, the MAX-reduce, the rank-0-only line --, not a
    scaling measurement, and the line says so.  The launcher and its workers are child processes started before they touch the GPU (no exec
    of a process that has initialised it)."""
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.fail("torch does not see the GPU")
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    port = str(32700 + os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--global-batch", "33", "--no-tune", "--no-cpu-baseline", "--parity-images", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                          # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 33
    assert "17 image(s) on rank 0" in out["config"]["workload"] and out["config"]["per_rank"]["images_on_rank0"] == 17
    assert out["value"] > 0 and abs(out["value"] - 33 * 1e3 / out["ms_per_step"]) / out["value"] < 0.01
    assert out["backend"].startswith("gloo: FUNCTIONAL") and out["exchange_check"] is True


"""The data-parallel step on the DEVICE with two ranks: both processes drive cuda:0 and exchange gradients
over gloo (RCCL needs one GPU per rank; the box has one) — this exercises what the CPU gloo test cannot: the
bf16 native block path announcing its gradients from the side stream (`block._notify`), the stem / tail nodes,
the reducer's communication stream and event ordering, and NativeAdamW on the reducer's arena.
Checks: reducer gradients == mean of the two ranks' local gradients, every bucket sent exactly once,
replicas identical after optimizer steps."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, backend="gloo"):
    local = rank if backend == "nccl" else 0                   # RCCL: one GPU per rank; gloo: both ranks drive the one GPU of the box
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    try:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        from cream_amd import comm
        from cream_amd.autoformer import engine
        dist.init_process_group(backend, rank=rank, world_size=world)
        torch.manual_seed(11 + rank)                           # different initial weights per rank (reducer broadcasts)
        model = engine.build_supernet("S", drop_path_rate=0.0, depth=3).to(dev)
        choices = dict(mlp_ratio=[3.0, 3.5, 4.0], num_heads=[5, 6, 7], depth=[2, 3], embed_dim=[320, 384, 448])
        opt = engine.build_optimizer(model, lr=1e-3, batch_size=8, world_size=world)
        reducer = comm.GradReducer(model)
        tr = engine.SupernetTrainer(model, opt, choices, reducer)
        g = torch.Generator().manual_seed(100 + rank)          # different data per rank
        images = torch.randn(8, 3, 224, 224, generator=g).to(dev)
        target = torch.softmax(torch.randn(8, 1000, generator=g), -1).to(dev)
        tr.start_epoch(1)
        cfg = tr.sample()
        loss = tr.forward_backward(images, target)
        torch.cuda.synchronize()
        assert all(v == 0 for v in reducer.pending.values()), reducer.pending
        mine = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
        sent = reducer.bytes_sent
        # local (un-averaged) gradients of the same sub-network through the same native path
        reducer.zero_grad()
        reducer.prepare(cfg)
        reducer.pending = {}                                   # hooks become no-ops: nothing is sent
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(images)
        engine.soft_target_cross_entropy(out, target).backward()
        torch.cuda.synchronize()
        local = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        want = sum(both) / world
        err = float((mine - want).abs().max() / want.abs().max())
        # optimizer steps keep the replicas identical
        for _ in range(2):
            tr.step(images, target)
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in model.parameters()])
        allp = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(allp, flat)
        in_sync = bool(torch.equal(allp[0], allp[1]))
        full = sum(buf.numel() * 4 for b, buf in reducer.flat.items() if not (b.startswith("block") and int(b[5:]) >= cfg["layer_num"]))
        # active-slice messages (csrc/slices.hip on the device): exactly the elements this configuration can write
        active = sum(4 * r * c for r, c in (comm.autoformer_active_slice(n, p, cfg, comm.attention_layout(model)) for n, p in model.named_parameters()))
        assert sent == active, (sent, active)
        assert reducer.use_avg == (backend == "nccl")          # RCCL: one AVG all-reduce per message
        q.put((rank, err, in_sync, sent, full, float(loss.detach()), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # report instead of hanging the parent
        import traceback
        q.put((rank, None, False, 0, 0, 0.0, traceback.format_exc()))
        raise e


def test_two_ranks_on_the_device_native_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    for rank, err, in_sync, sent, full, loss, tb in res:
        assert tb is None, f"rank {rank} failed:\n{tb}"
        print(f"[2 ranks, one device] rank {rank}: reducer vs mean of local gradients {err:.2e}, in sync {in_sync}, "
              f"sent {sent / 1e6:.1f} MB of {full / 1e6:.1f} MB active, loss {loss:.4f}")
        assert err < 1e-5, err                                  # the kernels are atomics-free and bit-reproducible: measured 0.0
        assert in_sync
        assert sent <= full                                     # slices of the sampled sub-network, not whole buckets
    assert all(p.exitcode == 0 for p in procs)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (the gpurun box has one; the 8-GPU node runs this)")
def test_two_ranks_rccl():
    """The same two-rank step over the `nccl` backend (= RCCL over xGMI): one process per GPU, the reducer's AVG all-reduce
    of the active-slice messages on its communication stream, NativeAdamW on the arena — gradients equal the mean of the
    ranks' local gradients, replicas identical after optimizer steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30900 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    for rank, err, in_sync, sent, full, loss, tb in res:
        assert tb is None, f"rank {rank} failed:\n{tb}"
        print(f"[2 ranks, RCCL] rank {rank}: reducer vs mean of local gradients {err:.2e}, in sync {in_sync}, sent {sent / 1e6:.1f} MB")
        assert err < 1e-5 and in_sync and 0 < sent <= full
    assert all(p.exitcode == 0 for p in procs)


def test_bench_n4_code_path_rehearsed_over_gloo(tmp_path):
    """bench.py's N > 1 path end to end with FOUR ranks — all on cuda:0, gradients over gloo (CREAM_DIST_BACKEND): init,
    rank-0 broadcast, the timed region with its barriers and the MAX over ranks, the in-step kernel-timing pass (every rank
    runs it: its steps contain collectives), the gathered device list and the compact headline.  The first real 8-GPU run
    of the driver must not die on a code path nobody executed; the numbers mean nothing."""
    import json
    import subprocess
    world = 4
    port = 31300 + (os.getpid() % 300)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), LOCAL_RANK="0",
               CREAM_DIST_BACKEND="gloo", CREAM_BENCH_EXTRA=str(tmp_path / "extra.json"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-host-leg"]
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exit {p.returncode}\n{se[-3000:]}"
    # (the gloo transport prints its own "[Gloo] Rank r is connected ..." banner on stdout: not bench.py's output)
    own = [[ln for ln in so.strip().splitlines() if ln.strip() and not ln.startswith("[Gloo]")] for so, _ in outs]
    line = json.loads(own[0][-1])
    assert line["n_gpus"] == world and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8 * world and line["config"]["comm"]["world"] == world
    assert line["config"]["comm"]["backend"] == "gloo"
    assert line["roofline"] and line["roofline"]["launches"] > 0          # the timing pass ran on every rank
    assert len(json.dumps(line)) < 4096                                   # the compact headline
    for r in range(1, world):
        assert own[r] == [], f"only rank 0 prints: {own[r][:3]}"
    extra = json.load(open(tmp_path / "extra.json"))
    assert len(extra["config_full"]["comm"]["devices"]) == world

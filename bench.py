#!/usr/bin/env python
"""bench.py — images/sec of the AutoFormer-S supernet train step @224^2 on N MI355X.

    python bench.py --gpus 1 --steps 40 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 40 --warmup 10

One "step" = one pass of the hot path over one batch of synthetic input already resident in
HBM: sample a random sub-network (random.seed(epoch) discipline, identical on all ranks),
set_sample_config, forward (bf16 autocast), soft-target cross entropy, backward with the
bucketed gradient all-reduce overlapped on a side stream, AdamW over the full supernet.
Per-GPU batch 128 (AutoFormer/README.md:73-75) => weak scaling.  W untimed warm-up steps,
then exactly K timed steps bracketed by barrier + synchronize; MAX over ranks; rank 0
prints ONE JSON line.

Extra objects on the line:
  roofline      — the dominant hand-written kernel family of the step BY IN-STEP TIME: algorithmic flops / HIP-event
                  time, events recorded inside the native block calls (csrc/block_seq.cpp: cream_block_prof_*) on the
                  stream each kernel is launched on, in a pass of the SAME two-stream native step right after the timed
                  region (the timed region itself carries no events); `mfma_util` / `traffic` of the same kernels come
                  from the committed rocprofv3 PMC passes under profiles/ (counters cannot be read in-process);
  rpe_index_config4 — the rpe_index gather / scatter-add at BASELINE config 4 (B=64, H=12, L=577, 50 buckets), fp32 and
                  bf16: algorithmic GB/s and fraction of the 8 TB/s HBM peak (HIP events, median of 20 launches);
  roofline_step — whole-step algorithmic rate: FLOPs of the sub-networks sampled in the timed region (SURVEY 8d formula) / time / 2.5 PF;
  per_embed_dim — mean GPU ms per step by sampled embed dim (events between steps, no host sync);
  host_unstalled — the same step at batch 4 (host-bound: same launches, ~30x less device work): what the host needs to
                  enqueue a step when the launch queue is never full (`host_enqueue_ms_per_step` of the timed region
                  includes waiting for queue slots whenever the device is the bottleneck);
  cpu_baseline  — kind "port": the oracle (CPU fp32 restatement of the reference step, oracle/autoformer_oracle.py)
                  timed on this box's host cores on a bounded sample at batch 64: best of 16 / 64 / all hardware
                  threads (rank 0, N=1 only), CPU model stated; plus one RPEAttention layer at config-4 shapes in
                  the reference's own pure-PyTorch formulation (oracle/irpe_oracle.py: flat-index gather, irpe.py:646).
  input_transform — the device input transform of lib/datasets.py:189-220 (crop, Pillow's bicubic resize, window / mirror, ToTensor,
                  Normalize) on 128 decoded frames resident in HBM: ms per batch, images/s, algorithmic GB/s (N=1 default run only);
  tinyclip_config5 — BASELINE config 5's distillation step on one device (student ViT-39M/16 + Text-19M, teacher ViT-B/16);
  irpe_config4  — BASELINE config 4 on the device: one RPEAttention layer (DeiT-B-384 + iRPE, L = 577) fwd+bwd
                  through the fused kernels of csrc/irpe_attn.hip, ms per layer (N=1 default run only); `model`: the whole
                  DeiT-base-384 + iRPE training step at batch 64 with the run of RPEBlocks as one node on the own kernels
                  (cream_amd/deit_native.py; tools/bench_deit_irpe.py measures the framework / composed variants beside it).
`--subnet T|S` benchmarks ONE fixed published sub-network instead of random sampling (BASELINE
config 2: AutoFormer-T subnet, bf16, batch 128).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory: shortens the dispatch of back-to-back kernels (a step is ~330 launches);
# measured +3.3 % images/s in a same-box A/B.  Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# RCCL between the ranks of a node shares device memory through dmabuf handles; the host driver of these boxes has no
# legacy IPC (without this, multi-process runs fail with `hipIpcGetMemHandle: invalid argument`)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA
PEAK_F32_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (README recipe: 128)")
    ap.add_argument("--supernet", default="S", choices=["T", "S", "B"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--impl", default="auto", choices=["auto", "fused", "bucketed"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=28.0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the tiny-batch pass that measures the unstalled host cost of a step")
    ap.add_argument("--no-mixup", action="store_true", help="feed precomputed soft targets instead of running the step body's mixup_fn "
                    "(supernet_engine.py:52-53) on the device inside the timed region")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-wgrad-stream", action="store_true", help="weight-gradient GEMMs on the main stream")
    ap.add_argument("--comm-mode", default="auto", choices=["auto", "allreduce", "rs_ag"],
                    help="gradient exchange per bucket: one all-reduce, or reduce-scatter + all-gather (all 7 xGMI links at once)")
    ap.add_argument("--subnet", default=None, choices=["T", "S"],
                    help="train ONE fixed published sub-network of that supernet (BASELINE config 2: T)")
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-irpe", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# published sub-networks: AutoFormer/experiments/subnet/AutoFormer-{T,S}.yaml (RETRAIN sections)
SUBNETS = {
    "T": dict(layer_num=13, embed_dim=[192] * 13,
              mlp_ratio=[3.5, 3.5, 3.0, 3.5, 3.0, 3.0, 4.0, 4.0, 3.5, 4.0, 3.5, 4.0, 3.5],
              num_heads=[3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4, 3, 3]),
    "S": dict(layer_num=13, embed_dim=[384] * 13,
              mlp_ratio=[3.0, 3.5, 3.0, 3.5, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 3.5, 4.0],
              num_heads=[6, 6, 5, 7, 5, 5, 5, 6, 6, 7, 7, 6, 7]),
}


def step_flops_per_image(cfg, n_tokens=197, patch_in=768, num_classes=1000):
    """Algorithmic FLOPs of ONE train step per image for a sampled sub-network: forward =
    2*196*768*E (patch) + sum over layers [2 N E 3Q (qkv) + 4 H N^2 64 (QK^T, PV) + 4 N Q 60 (bucketed RPE, both sides)
    + 2 N Q E (proj) + 4 N E F (fc1 + fc2)] + 2 E 1000 (head), Q = 64 H, F = int(E R); train = 3 x forward."""
    N, E = n_tokens, cfg["embed_dim"][0]
    f = 2.0 * (N - 1) * patch_in * E + 2.0 * E * num_classes
    for i in range(cfg["layer_num"]):
        H = cfg["num_heads"][i]
        Q, F = 64 * H, int(E * cfg["mlp_ratio"][i])
        f += 2.0 * N * E * 3 * Q + 4.0 * H * N * N * 64 + 4.0 * N * Q * 60 + 2.0 * N * Q * E + 4.0 * N * E * F
    return 3.0 * f


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_irpe_leg(seconds, threads=0):
    """The iRPE half of the path on the host, kind "port": one RPEAttention layer (DeiT-base-384 geometry, C = 768, H = 12,
    L = 577, product / contextual, 50 buckets, rpe on k) forward + backward at B = 2 in the reference's own PURE-PYTORCH
    formulation — lookup = q W, gathered with the flat index i * nb + bucket[i, j] (irpe.py:573-583, :646-647), which is
    what the reference runs on a host where its rpe_index_cpp extension is not built (oracle/irpe_oracle.py, pinned
    against reference-made fixtures)."""
    from oracle import irpe_oracle as IO
    if threads > 0:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    C, H, L, B = 768, 12, 577, 2
    ids, nb = IO.product_bucket_ids(24, 24, 1)
    p = {"qkv.weight": torch.randn(3 * C, C) * C ** -0.5, "qkv.bias": torch.zeros(3 * C),
         "rpe_k.lookup_table_weight": 0.02 * torch.randn(1, 64, nb), "proj.weight": torch.randn(C, C) * C ** -0.5,
         "proj.bias": torch.zeros(C)}
    p = {k: v.requires_grad_() for k, v in p.items()}
    x = torch.randn(B, L, C, requires_grad=True)

    def once():
        IO.rpe_attention_layer(p, x, H, ids, nb).sum().backward()
        x.grad = None
        for v in p.values():
            v.grad = None

    once()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and n < 20:
        once()
        n += 1
    dt = (time.perf_counter() - t0) / max(1, n)
    return dict(ms_per_layer_fwd_bwd=round(dt * 1e3, 1), batch=B, cores=torch.get_num_threads(), kind="port",
                sample=f"{n} RPEAttention fwd+bwd, B={B} H={H} L={L} fp32, reference's pure-PyTorch formulation (irpe.py:646 flat-index "
                       "gather; oracle/irpe_oracle.py)")


REFERENCE_AUTOFORMER = "/root/reference/AutoFormer"


def cpu_baseline_reference(size, seconds, threads=0, B=64):
    """kind "reference": the reference's OWN modules (AutoFormer/model/supernet_transformer.py Vision_TransformerSuper and
    supernet_engine.py sample_configs, imported read-only through tests/refshim.py) doing the step on the host cores —
    only where the reference checkout exists (the build container; the GPU box has none and times the port).  The loss
    and the optimizer are torch's (timm is not vendored): soft-target CE and AdamW as in cpu_baseline."""
    import random
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import refshim
    from cream_amd.autoformer import engine
    if threads > 0:
        torch.set_num_threads(threads)
    ns = refshim.load_autoformer_reference()
    sample_configs = refshim.reference_sample_configs()
    space = engine.SEARCH_SPACES[size]
    torch.manual_seed(0)
    model = ns.Vision_TransformerSuper(img_size=224, patch_size=16, embed_dim=space["embed_dim"], depth=space["depth"],
                                       num_heads=space["num_heads"], mlp_ratio=space["mlp_ratio"], qkv_bias=True, drop_rate=0.0,
                                       drop_path_rate=0.0, gp=True, num_classes=1000, max_relative_position=14,
                                       relative_position=True, change_qkv=True, abs_pos=True).float()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4 * B / 512, weight_decay=0.05)
    images = torch.randn(B, 3, 224, 224)
    target = torch.zeros(B, 1000).scatter_(1, torch.randint(0, 1000, (B, 1)), 1.0)
    random.seed(0)
    n, t0, elapsed = 0, None, 0.0
    while True:
        cfg = sample_configs(choices=space["choices"])
        model.set_sample_config(config=cfg)
        opt.zero_grad(set_to_none=False)
        loss = torch.sum(-target * torch.nn.functional.log_softmax(model(images), dim=-1), dim=-1).mean()
        loss.backward()
        opt.step()
        if t0 is None:
            t0 = time.perf_counter()
            continue
        n += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= seconds or n >= 50:
            break
    return dict(value=round(n * B / elapsed, 2), unit="images/sec", cores=torch.get_num_threads(), kind="reference",
                sample=f"{n} AutoFormer-{size} supernet steps of batch {B} (fp32) on the reference's own Vision_TransformerSuper + "
                       "sample_configs (imported read-only via tests/refshim.py); soft-target CE / AdamW from torch (timm not vendored)")


def cpu_baseline(size, seconds, threads=0, B=64):
    """Reference step restated on the CPU (oracle/autoformer_oracle.py), fp32, `threads` intra-op
    threads (0 = torch's default = all cores), batch B, random sub-networks from the same draw
    sequence, AdamW over the full supernet.  Bounded: warm-up 1 step, then steps until `seconds`.
    Where the reference checkout exists its own modules are timed instead (cpu_baseline_reference)."""
    if os.path.isdir(REFERENCE_AUTOFORMER) and os.environ.get("CREAM_CPU_BASELINE", "") != "port":
        try:
            return cpu_baseline_reference(size, seconds, threads, B)
        except Exception as e:                   # (an import shim that no longer fits): the port still gives a line
            sys.stderr.write(f"[bench] reference CPU leg failed ({e}); timing the port\n")
    import random
    from oracle import autoformer_oracle as AO
    from cream_amd.autoformer import engine
    if threads > 0:
        torch.set_num_threads(threads)
    space = engine.SEARCH_SPACES[size]
    torch.manual_seed(0)
    model = engine.build_supernet(size, drop_path_rate=0.0).float()      # parameter container
    params = {k: v.detach().clone().requires_grad_(True) for k, v in model.named_parameters()}
    opt = torch.optim.AdamW(list(params.values()), lr=5e-4 * B / 512, weight_decay=0.05)
    images = torch.randn(B, 3, 224, 224)
    target = torch.zeros(B, 1000).scatter_(1, torch.randint(0, 1000, (B, 1)), 1.0)
    random.seed(0)
    n, t0, elapsed = 0, None, 0.0
    while True:
        cfg = AO.sample_configs(space["choices"])
        opt.zero_grad(set_to_none=False)
        loss = AO.soft_target_cross_entropy(AO.forward(params, cfg, images), target)
        loss.backward()
        opt.step()
        if t0 is None:
            t0 = time.perf_counter()         # first step = warm-up
            continue
        n += 1
        elapsed = time.perf_counter() - t0
        if elapsed >= seconds or n >= 50:
            break
    return dict(value=round(n * B / elapsed, 2), unit="images/sec", cores=torch.get_num_threads(),
                kind="port", sample=f"{n} AutoFormer-{size} supernet steps of batch {B} (fp32, oracle/autoformer_oracle.py: the "
                                    "reference's dense formulation restated; NOT the reference's own code — timm is not vendored)")


def cpu_baseline_best(size, seconds):
    """Best of 16 / 64 / all hardware threads at batch 64 (an oversubscribed pool at batch 16 was 3x slower than 8 threads
    in round 1; a larger batch gives the wide pools enough work), each in its own process, plus the iRPE leg; CPU model
    and counts stated."""
    ncpu = os.cpu_count() or 1
    # all 256 hardware threads of the GPU box: > 160 s for two steps of batch 64 (gpurun_out/r03a_bench.err) — the oracle's
    # small GEMMs do not scale past one CCD group; 64 threads are already 2x slower than 16
    counts = sorted({c for c in (16, 32, 64) if c <= ncpu}) or [ncpu]
    per = max(4.0, seconds / (len(counts) + 1))
    tried, best = {}, None
    for c in counts:
        r = cpu_baseline_subprocess(size, per, threads=c)
        if r:
            tried[str(c)] = r["value"]
            if best is None or r["value"] > best["value"]:
                best = r
    if best is None:
        return None
    best["threads_tried"] = tried
    best["threads_note"] = "all 256 hardware threads: more than 160 s for two batch-64 steps (round-3 run), excluded from the sweep"
    best["cpu_model"] = cpu_model()
    best["host_cores"] = ncpu
    if best.get("kind") == "port":
        # the port against the reference's OWN code, same box, same 8 threads, in the build container (the GPU box has no reference
        # checkout): profiles/r05_cpu_port_vs_reference.jsonl — reference 12.17 images/s, port 13.79
        best["port_over_reference"] = dict(ratio=1.13, measured_on="build container, 8 threads, batch 64, same process setup",
                                           reference_images_per_sec=12.17, port_images_per_sec=13.79,
                                           source="profiles/r05_cpu_port_vs_reference.jsonl",
                                           reference_equivalent=round(best["value"] / 1.13, 2))
    irpe = cpu_baseline_subprocess(size, per, threads=min(64, ncpu), irpe=True)
    if irpe:
        best["irpe_config4_cpu"] = irpe
    return best


def cpu_baseline_subprocess(size, seconds, threads=0, irpe=False):
    """The CPU leg runs in its OWN process, after the GPU measurement: its intra-op thread pool
    (all host cores, spinning between parallel regions) must not share a process — or a time
    window — with the thread that launches the GPU kernels (measured: 13.8 -> 19.1 ms/step when
    the pool of a finished CPU run was still alive in the benchmark process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--supernet", size,
           "--cpu-seconds", str(seconds), "--cpu-threads", str(threads)] + (["--cpu-irpe"] if irpe else [])
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=seconds * 6 + 120)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        sys.stderr.write(f"[bench] cpu baseline produced no result:\n{out.stderr[-2000:]}\n")
    except Exception as e:  # the GPU line must not be lost to a failing baseline leg
        sys.stderr.write(f"[bench] cpu baseline failed: {e}\n")
    return None


def pmc_traffic(region):
    """HBM bytes per launch of a timed region from the committed rocprofv3 PMC passes
    (profiles/*_kernels_pmc.json, written by tools/summarize_pmc.py: FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md + WRITE_SIZE, mean over the launches of the sampled
    sub-networks).  PMC counters cannot be read from inside this process, so the number is the one
    measured by tools/gpu_round.sh for the same kernels; None if no such file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernels_pmc.json")))
    if not files:
        return None, None
    rec = json.load(open(files[-1]))
    val = rec.get("traffic_bytes_per_launch_by_timed_region", {}).get(region)
    return (val, os.path.relpath(files[-1], ROOT)) if val else (None, None)


def pmc_kernels(pattern):
    """{kernel name: {mfma_util, hbm_read_MB_corrected, hbm_write_MB, ...}} of the kernels whose name contains `pattern`, from the
    newest committed profiles/*_kernels_pmc.json (rocprofv3 --pmc passes of this bench command, tools/summarize_pmc.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernels_pmc.json")))
    if not files:
        return None
    rec = json.load(open(files[-1]))
    out = {k: v for k, v in rec.get("kernels", {}).items() if pattern in k}
    return dict(source=os.path.relpath(files[-1], ROOT), kernels=out) if out else None


def native_kernel_choice():
    """the process-wide kernel-choice switches of the library as the timed region ran them (include/cream_amd.h)"""
    from cream_amd import _lib
    lib = _lib.load()
    return {"cream_gemm_nt8": lib.cream_gemm_nt8(-1), "cream_gemm_nt256": lib.cream_gemm_nt256(-1), "cream_gemm_tn8": lib.cream_gemm_tn8(-1),
            "cream_gemm_nthalf": lib.cream_gemm_nthalf(-1), "cream_gemm_ntopt": lib.cream_gemm_ntopt(-1),
            "cream_block_wgrad_bf16": lib.cream_block_wgrad_bf16(-1), "cream_attn_rpe2d_bwd_mode": lib.cream_attn_rpe2d_bwd_mode(-1)}


def native_prof_summary():
    """Collect the in-step HIP-event records of csrc/block_seq.cpp -> {name: dict(launches, total_ms, avg_ms, flops, bytes)}."""
    import ctypes
    from cream_amd import _lib
    lib = _lib.load()
    n = lib.cream_block_prof_kinds()
    ms, cnt, fl, by = (ctypes.c_double * n)(), (ctypes.c_int64 * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    _lib.check(lib.cream_block_prof_collect(ms, cnt, fl, by), "cream_block_prof_collect")
    out = {}
    for k in range(n):
        if cnt[k]:
            out[lib.cream_block_prof_name(k).decode()] = dict(launches=int(cnt[k]), total_ms=ms[k], avg_ms=ms[k] / cnt[k],
                                                              flops=fl[k], bytes=by[k])
    return out


def rpe_index_config4_leg(iters=20, warmup=3):
    """north_star's first rocprof figure, measured live: the rpe_index gather (cream_rpe_index_fwd) and scatter-add
    (cream_rpe_index_bwd) at BASELINE config 4 — DeiT-B-384 + iRPE: B = 64, H = 12, L = 577, 50 buckets — through the
    Python operator (iRPE's transposed input view), fp32 and bf16.  Algorithmic bytes: lookup rows
    B H L nb s + index L L 4 + output B H L L s; median of `iters` launches between HIP events on the launch stream."""
    from cream_amd import rpe_index as R
    dev = torch.device("cuda")
    B, H, L, nb = 64, 12, 577, 50
    index = torch.randint(0, nb, (L, L), dtype=torch.int32, device=dev)

    def med(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2]

    out = {"workload": "rpe_index gather / scatter-add, B=64 H=12 Lq=Lk=577 nb=50 (DeiT-B-384 + iRPE product-ctx)", "peak_GBps": PEAK_HBM_GBS}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        s_ = torch.empty((), dtype=dt).element_size()
        x = torch.randn(H, B, L, nb, device=dev).to(dt).transpose(0, 1)
        g = torch.randn(B, H, L, L, device=dev).to(dt)
        gin = torch.zeros(B, H, L, nb, device=dev, dtype=dt)
        nbytes = B * H * L * nb * s_ + L * L * 4 + B * H * L * L * s_
        f = med(lambda: R.forward_gpu(x, index))
        b = med(lambda: R.backward_gpu(gin, g, index, accumulate=False))
        out[name] = {"algorithmic_MB": round(nbytes / 1e6, 1),
                     "gather": {"us": round(f * 1e3, 1), "GBps": round(nbytes / f / 1e6, 1), "frac": round(nbytes / f / 1e6 / PEAK_HBM_GBS, 4)},
                     "scatter": {"us": round(b * 1e3, 1), "GBps": round(nbytes / b / 1e6, 1), "frac": round(nbytes / b / 1e6 / PEAK_HBM_GBS, 4)}}
        del x, g, gin
    pmc = pmc_kernels("rpe_")
    if pmc:
        out["pmc"] = pmc
    return out


def irpe_config4_leg(iters=10):
    """BASELINE config 4 on the device: ONE RPEAttention layer of DeiT-B-384 with iRPE
    product / contextual (B = 64, H = 12, L = 577, 50 buckets) forward + backward under bf16 autocast —
    the fused kernels of csrc/irpe_attn.hip — for rpe on keys and on q, k and v; ms per layer and the
    kernels' share (HIP events on the launch stream)."""
    from cream_amd import timing
    from cream_amd.irpe import get_rpe_config
    from cream_amd.rpe_attention import RPEAttention
    dev = torch.device("cuda")
    B, L, C, H = 64, 577, 768, 12
    out = {}
    for rpe_on in ("k", "qkv"):
        torch.manual_seed(0)
        cfg = get_rpe_config(ratio=1.9, method="product", mode="ctx", shared_head=True, skip=1, rpe_on=rpe_on)
        m = RPEAttention(C, num_heads=H, qkv_bias=True, rpe_config=cfg).to(dev)
        x = torch.randn(B, L, C, device=dev, requires_grad=True)
        g = torch.randn(B, L, C, device=dev)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            y.backward(g)
            x.grad = None
            for p in m.parameters():
                p.grad = None

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        timing.reset()
        timing.enable(True, only=("irpe_attn_fwd", "irpe_attn_bwd"))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            step()
        b.record()
        torch.cuda.synchronize()
        timing.enable(False)
        ks = timing.summary()
        out[rpe_on] = dict(ms_per_layer_fwd_bwd=round(a.elapsed_time(b) / iters, 3),
                           kernels={k: dict(avg_us=round(v["avg_ms"] * 1e3, 1),
                                            tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1)) for k, v in sorted(ks.items())})
        del m, x, g
    timing.reset()
    return dict(workload="RPEAttention layer fwd+bwd, DeiT-B-384 iRPE product-ctx, B=64 H=12 L=577, bf16 autocast", **out)


def deit_config4_model_leg(iters=6, batch=64):
    """BASELINE config 4 as a whole model: DeiT-base-384 + iRPE (product, contextual, rpe on k), batch 64, one training step
    (forward + backward + AdamW) under bf16 autocast with the run of RPEBlocks as one node on the own kernels
    (cream_amd/deit_native.py); tools/bench_deit_irpe.py measures the framework / composed variants beside it."""
    from cream_amd.rpe_attention import deit_irpe
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = deit_irpe("base", img_size=384, rpe_on="k").to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)
    x = torch.randn(batch, 3, 384, 384, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(model(x).float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    del model, opt, x, y
    torch.cuda.empty_cache()
    return dict(workload="DeiT-base-384 + iRPE product-ctx (rpe on k) train step, batch 64, L = 577, bf16 autocast, blocks on the own kernels",
                ms_per_step=round(ms, 2), images_per_s=round(batch / ms * 1e3, 1))


def input_transform_leg(B=128, iters=20):
    """The device input transform of lib/datasets.py:189-220 (csrc/image_transform.hip) on B synthetic decoded frames of ImageNet's
    typical sizes, frames resident in HBM: both pipelines' kernels (HIP events on the launch stream), images/s and the algorithmic
    bytes per second (box rows read + intermediate written and read + fp32 batch written) against the 8 TB/s HBM peak."""
    import ctypes
    import random as _random
    import numpy as np
    from cream_amd import _lib
    from cream_amd.autoformer import data as D
    dev, size = "cuda:0", 224
    rng, pr = np.random.default_rng(0), _random.Random(0)
    shapes = [[(375, 500), (500, 375), (333, 500), (480, 640), (768, 1024)][i % 5] for i in range(B)]
    frames = [torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in shapes]
    T = D.DeviceTransform(size, device=dev)
    lib = _lib.load()
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=dev)
    res = {"workload": f"{B} decoded frames (333x500 .. 768x1024, HWC uint8) -> ({B}, 3, {size}, {size}) fp32, frames resident in HBM"}
    for name in ("train", "eval"):
        params = [D.eval_crop_params(h, w) + (False,) if name == "eval" else D.train_crop_params(h, w, pr) for h, w in shapes]
        descs, nbytes, ws = T.plan(shapes, params)
        pix = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for d, f in zip(descs, frames):
            pix[d.offset:d.offset + f.numel()] = f.reshape(-1).to(dev)
        dd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        wsb = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        call = lambda: lib.cream_image_batch_transform(p(out), p(pix), nbytes, descs, p(dd), B, size, size, T._mean, T._std, p(wsb),
                                                       wsb.numel(), ctypes.c_void_p(st.cuda_stream))
        for _ in range(3):
            _lib.check(call(), "cream_image_batch_transform")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            call()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        algo = (sum(d.nrows * d.box_w * 3 for d in descs) + 2 * sum(d.nrows * size * 3 for d in descs) + B * 3 * size * size * 4)
        res[name] = {"ms_per_batch": round(ms, 4), "images_per_sec": round(B / ms * 1e3), "algorithmic_bytes": int(algo),
                     "GBps": round(algo / ms / 1e6, 1), "frac_of_hbm_peak": round(algo / ms / 1e6 / 8000, 3)}
    return res


def tinyclip_config5_leg(batch=256, iters=10):
    """BASELINE config 5 on ONE device: the affinity-mimicking distillation step of TinyCLIP — student
    TinyCLIP-ViT-39M/16 + Text-19M, frozen teacher ViT-B/16, ClipSoftLoss, gradient clipping 5, AdamW — on synthetic
    image / token batches, bf16 autocast; the image towers' attention on the fused kernels of csrc/irpe_attn.hip.  The
    reference quotes this configuration on 8 GPUs at 1024 pairs per GPU; this is the per-GPU building block."""
    from cream_amd import timing
    from cream_amd.tinyclip import model as M
    from cream_amd.tinyclip.distill import DistillStep
    dev = torch.device("cuda")
    torch.manual_seed(0)
    student = M.create_model("TinyCLIP-ViT-39M-16-Text-19M").to(dev)
    teacher = M.create_model("ViT-B-16").to(dev)
    opt = torch.optim.AdamW(student.parameters(), lr=1e-4, weight_decay=0.2, fused=True)
    step = DistillStep(student, teacher, opt, logit_scale=50.0, norm_gradient_clip=5.0, amp_dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(5)
    images = torch.randn(batch, 3, 224, 224, device=dev, generator=g)
    texts = torch.randint(1, 49406, (batch, 77), device=dev, generator=g)
    texts[:, 40] = 49407                                        # end-of-text
    texts[:, 41:] = 0
    for _ in range(3):
        loss = step.step(images, texts)
    torch.cuda.synchronize()
    timing.reset()
    timing.enable(True, only=("irpe_attn_fwd", "irpe_attn_bwd"))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        loss = step.step(images, texts)
    b.record()
    torch.cuda.synchronize()
    timing.enable(False)
    ks = timing.summary()
    timing.reset()
    ms = a.elapsed_time(b) / iters
    assert torch.isfinite(loss).item()
    return dict(workload="TinyCLIP distillation step: ViT-39M/16 + Text-19M student, ViT-B/16 teacher, ClipSoftLoss, bf16 autocast, 1 GPU",
                batch=batch, ms_per_step=round(ms, 2), pairs_per_s=round(batch / ms * 1e3, 1), loss=round(float(loss), 4),
                kernels={k: dict(launches_per_step=round(v["launches"] / iters, 1), avg_us=round(v["avg_ms"] * 1e3, 1)) for k, v in sorted(ks.items())})


def compact_line(line):
    """(headline, extra): the headline keeps the driver's contract keys, `roofline` and `cpu_baseline` in short form (< 4 KB: the
    driver's record truncated the 14 KB line of round 5); everything else — per-kernel table, config-4 / config-5 / host legs,
    method notes — goes to bench_extra.json, which the headline names."""
    extra = {}
    head = dict(line)
    for k in ("rpe_index_config4", "irpe_config4", "tinyclip_config5", "input_transform", "host_unstalled", "per_embed_dim", "parity_unpinned",
              "host_enqueue_ms_per_step"):
        if k in head:
            extra[k] = head.pop(k)
    roof = head.get("roofline")
    if roof:
        extra["roofline_full"] = roof
        head["roofline"] = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_us")}
        head["roofline"]["timing"] = "in-step HIP event pairs on each kernel's own dispatch packet, both streams live (bench_extra.json: roofline_full)"
        kern = roof.get("kernels") or {}
        head["roofline"]["attention_us"] = {k: kern[k]["avg_us"] for k in ("attn_rpe2d_fwd", "attn_rpe2d_bwd") if k in kern}
    rs = head.get("roofline_step")
    if rs:
        extra["roofline_step_full"] = rs
        head["roofline_step"] = {k: rs.get(k) for k in ("achieved", "peak", "unit", "frac")}
    cpu = head.get("cpu_baseline")
    if cpu:
        extra["cpu_baseline_full"] = cpu
        short = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
        short["sample"] = (cpu.get("sample") or "")[:160]
        if "port_over_reference" in cpu:
            short["port_over_reference"] = cpu["port_over_reference"]["ratio"]
            short["reference_equivalent"] = cpu["port_over_reference"]["reference_equivalent"]
        head["cpu_baseline"] = short
    cfg = dict(head.get("config") or {})
    extra["config_full"] = cfg
    comm = cfg.get("comm") or {}
    head["config"] = {"workload": cfg.get("workload"), "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism"),
                      "comm": {k: comm.get(k) for k in ("world", "backend", "mode", "cu_budget", "grad_bytes_per_step_last") if k in comm},
                      "wgrad_partials": "bf16 split-K partial tiles (within 1.9e-3 of fp32 partials)",
                      "mixup": "on-device, in the timed region" if str(cfg.get("mixup", "")).startswith("in the timed") else "off",
                      "kernel_choice": cfg.get("kernel_choice")}
    head["extra"] = "bench_extra.json (side legs: rpe_index / iRPE config 4, TinyCLIP config 5, host leg, per-kernel table, notes)"
    return head, extra


def main():
    a = parse()
    if a.cpu_baseline_only:
        res = cpu_irpe_leg(a.cpu_seconds, a.cpu_threads) if a.cpu_irpe else cpu_baseline(a.supernet, a.cpu_seconds, a.cpu_threads)
        print(json.dumps(res), flush=True)
        return
    from cream_amd import comm, timing
    from cream_amd.autoformer import engine
    rank, local, world = comm.init_distributed()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    cpu = None

    if a.no_wgrad_stream:
        from cream_amd.autoformer import block as _blk
        _blk.WGRAD_SIDE_STREAM = False
    torch.manual_seed(0 + rank)                               # supernet_train.py:196-198
    model = engine.build_supernet(a.supernet, drop_path_rate=0.1).to(dev)
    for m in model.modules():
        if hasattr(m, "attention_impl"):
            m.attention_impl = a.impl
    opt = engine.build_optimizer(model, lr=5e-4, batch_size=a.batch, world_size=world)
    if a.comm_mode == "auto":
        a.comm_mode = comm.default_comm_mode(world)
    reducer = comm.GradReducer(model, mode=a.comm_mode)
    amp = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    mixup_fn = None
    if not a.no_mixup:
        # the step body's `samples, targets = mixup_fn(samples, targets)` (supernet_engine.py:52-53) with the recipe of
        # supernet_train.py:245-251 (mixup 0.8, cutmix 1.0, prob 1.0, switch 0.5, batch mode, label smoothing 0.1): the resident
        # batch is mixed in place by one launch per step, which also builds the soft targets (csrc/mixup.hip)
        from cream_amd.autoformer.data import Mixup
        import numpy as _np
        _np.random.seed(0 + rank)                             # supernet_train.py:197-199 seeds numpy with seed + rank
        mixup_fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000)
    trainer = engine.SupernetTrainer(model, opt, engine.SEARCH_SPACES[a.supernet]["choices"], reducer, amp_dtype=amp, mixup_fn=mixup_fn)
    if a.subnet:                                              # BASELINE config 2: one fixed sub-network
        assert a.subnet == a.supernet, "--subnet X needs --supernet X"
        fixed = SUBNETS[a.subnet]

        def sample_fixed():
            trainer.config = fixed
            model.set_sample_config(fixed)
            return fixed
        trainer.sample = sample_fixed

    # synthetic ImageNet-shaped batch, generated on the device, resident before timing
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(a.batch, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 1000, (a.batch,), device=dev, generator=g)
    if mixup_fn is not None:
        target = labels                                       # class indices: the mixup launch builds the smoothed soft targets
    else:
        target = torch.full((a.batch, 1000), 0.1 / 1000, device=dev)
        target[torch.arange(a.batch, device=dev), labels] += 0.9

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trainer.start_epoch(0)
    for _ in range(a.warmup):
        loss = trainer.step(images, target)
    sync()
    marks, dims = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)], []
    flops_timed = 0.0                        # algorithmic FLOPs per image summed over the configs ACTUALLY sampled in the timed region
    t0 = time.perf_counter()
    for i in range(a.steps):
        marks[i].record()                    # (asynchronous: no host sync inside the timed region)
        loss = trainer.step(images, target)
        dims.append(trainer.config["embed_dim"][0])
        flops_timed += step_flops_per_image(trainer.config)
    marks[a.steps].record()
    t_issue = time.perf_counter() - t0       # host time to ENQUEUE the steps (== dt when launch-bound)
    sync()
    dt = time.perf_counter() - t0
    per_e = {}
    for i, e in enumerate(dims):
        per_e.setdefault(e, []).append(marks[i].elapsed_time(marks[i + 1]))
    per_e = {str(e): dict(steps=len(v), gpu_ms_per_step=round(sum(v) / len(v), 3)) for e, v in sorted(per_e.items())}
    assert torch.isfinite(loss).item(), "loss is not finite"     # supernet_engine.py:87-89

    # Host cost of a step WITHOUT back-pressure from the device: when the GPU is the bottleneck the launch queue fills
    # and `t_issue` above includes the host waiting for queue slots.  The same code path at a tiny batch (same launches,
    # same descriptors, ~30x less device work) is host-bound, so its time per step is what the host needs to enqueue one.
    host_leg = None
    if not a.no_host_leg:
        hb = min(4, a.batch)
        im4, tg4 = images[:hb].contiguous(), target[:hb].contiguous()
        for _ in range(3):
            trainer.step(im4, tg4)
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            trainer.step(im4, tg4)
        t_enq = time.perf_counter() - t0
        sync()
        host_leg = {"batch": hb, "steps": 20, "enqueue_ms_per_step": round(t_enq / 20 * 1e3, 3),
                    "ms_per_step": round((time.perf_counter() - t0) / 20 * 1e3, 3)}

    # Kernel-level timing for the roofline entry: HIP events recorded INSIDE the native block calls (csrc/block_seq.cpp,
    # cream_block_prof_*), each pair on the stream its kernel is launched on, over a pass of the same two-stream native
    # step right after the timed region — so durations include the contention between the weight-gradient GEMMs on the
    # side stream and the main chain, as in the timed steps (the op-by-op single-stream pass of round 2 did not).
    ksum = {}
    if not a.no_kernel_timing:                # every rank runs the pass (its steps contain collectives)
        from cream_amd import _lib
        lib = _lib.load()
        try:
            native_prof_summary()                 # (drop stale records)
            trainer.start_epoch(0)
            lib.cream_block_prof_enable(1)
            pm = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            nprof = max(4, min(a.steps, 12))
            pm[0].record()
            for _ in range(nprof):
                trainer.step(images, target)
            pm[1].record()
            torch.cuda.synchronize()
            lib.cream_block_prof_enable(0)
            ksum = native_prof_summary()
            prof_ms_per_step = pm[0].elapsed_time(pm[1]) / nprof
        except Exception as e:                    # the throughput line must not be lost to the timing pass
            if world > 1:
                raise                             # (ranks would fall out of step with each other)
            sys.stderr.write(f"[bench] in-step kernel timing failed: {e}\n")
            lib.cream_block_prof_enable(0)
            ksum = {}

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    devices = [f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(dev)}"]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])       # proof that `world` ranks on distinct devices took part
        devices = gathered
    dt = t.item()

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        torch.cuda.synchronize()
        cpu = cpu_baseline_best(a.supernet, a.cpu_seconds)            # after the GPU measurement, own processes

    if rank == 0:
        roof = None
        if ksum:
            # dominant hand-written kernel family by total IN-STEP time (both streams live)
            name, st = max(ksum.items(), key=lambda kv: kv[1]["total_ms"])
            if st["flops"]:
                peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
                ach = st["flops"] / (st["total_ms"] * 1e-3) / 1e12
                roof = dict(kernel=name, bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                            frac=round(ach / peak, 4), traffic=None)
            else:
                ach = st["bytes"] / (st["total_ms"] * 1e-3) / 1e9
                roof = dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(ach / PEAK_HBM_GBS, 4), traffic=None)
            roof["traffic"], src = pmc_traffic(name) if a.supernet == "S" else (None, None)   # counters were taken on S shapes
            if src:
                roof["traffic_unit"] = "HBM bytes per launch (mean over the sampled sub-networks), rocprofv3 PMC pass: " + src
            roof["launches"] = st["launches"]
            roof["avg_us"] = round(st["avg_ms"] * 1e3, 2)
            roof["timing"] = ("in-step: a start / stop HIP event pair carried by each kernel's own dispatch packet (hipExtLaunchKernelGGL, "
                              "csrc/launch_ev.hpp) inside the native block calls, on the kernel's launch stream, two streams live; "
                              f"{nprof} steps right after the timed region at {round(prof_ms_per_step, 3)} ms/step with the events in place; "
                              "multi-kernel operators are timed on their last kernel; the pass with events runs ~10 % slower per step than "
                              "the timed region, so these per-kernel rates are slightly pessimistic (they agree with rocprofv3's)")
            roof["kernels"] = {k: dict(launches=v["launches"], avg_us=round(v["avg_ms"] * 1e3, 2),
                                       total_ms=round(v["total_ms"], 3), ms_per_step=round(v["total_ms"] / nprof, 3),
                                       tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                       GBps=round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None)
                               for k, v in sorted(ksum.items())}
            qk = pmc_kernels("attn_rpe2d")
            if qk and a.supernet == "S":
                roof["qkt_mfma_util"] = dict(source=qk["source"], note="SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs), "
                                             "rocprofv3 PMC pass of this command", kernels={k: v.get("mfma_util") for k, v in qk["kernels"].items()})
        ips = a.steps * a.batch * world / dt
        what = (f"AutoFormer-{a.subnet} published sub-network (experiments/subnet/AutoFormer-{a.subnet}.yaml) train step"
                if a.subnet else f"AutoFormer-{a.supernet} supernet train step, random-path sampling (random.seed(epoch))")
        line = {
            "metric": f"images/sec (whole node) AutoFormer-{a.supernet} {'subnet' if a.subnet else 'supernet'} step @224^2",
            "value": round(ips, 1),
            "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "host_enqueue_ms_per_step": round(t_issue / a.steps * 1e3, 3),
            "host_unstalled": host_leg,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{what}, per-GPU batch {a.batch}, 224x224, AdamW, grad all-reduce RCCL",
                       "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "attention_impl": a.impl,
                       "comm": {"world": world, "backend": (dist.get_backend() if dist.is_initialized() else "none"),
                                "devices": devices, "grad_bytes_per_step_last": reducer.bytes_sent,
                                "grad_bytes_full_buckets": int(reducer.arena.numel() * 4),
                                "message": "active slices of the sampled sub-network per block bucket (csrc/slices.hip), side stream",
                                "mode": a.comm_mode, "cu_budget": comm.comm_cu_budget(),
                                # what RCCL was told (unset = library defaults): read a SCALE record against these
                                "rccl_env": {k: os.environ[k] for k in sorted(os.environ)
                                             if k.startswith(("NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HIP_FORCE_DEV_KERNARG")}},
                       "mixup": ("in the timed region: batch-mode Mixup 0.8 / CutMix 1.0 + label smoothing 0.1 on the device, one launch "
                                 "per step (csrc/mixup.hip; supernet_engine.py:52-53)" if mixup_fn is not None else "off (precomputed soft targets)"),
                       "gemm": "own MFMA kernels (csrc/gemm_mfma.hpp, gemm_nt8.hpp, gemm_tn8.hpp), no vendor GEMM library",
                       "wgrad_partials": "bf16 token-sliced partial tiles of the weight gradients, added in fp32 in fixed order "
                                         "(cream_block_wgrad_bf16; within 1.9e-3 of fp32 partials, tests/test_block_gpu.py)",
                       "kernel_choice": native_kernel_choice()},
            "roofline": roof,
            # whole-step algorithmic rate PER GPU from the FLOPs of the sub-networks actually sampled in the timed region
            # (SURVEY 8d formula, step_flops_per_image), not from the search-space mean
            "roofline_step": ({"achieved": round(flops_timed * a.batch / dt / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(flops_timed * a.batch / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
                               "gflop_per_image_timed_mean": round(flops_timed / a.steps / 1e9, 2),
                               "flops_per_image": "sum over the timed steps of 3 x forward FLOPs of the sampled config (SURVEY 8d); "
                                                  "search-space mean for reference: 28.6 GFLOP"}
                              if a.dtype == "bf16" else None),
            "per_embed_dim": per_e,
            "parity_unpinned": "AdamW parameter-group rule, soft-target CE, Mixup (timm, not vendored in the reference)",
            "cpu_baseline": cpu,
        }
        if world == 1 and a.supernet == "S" and not a.subnet and not a.no_kernel_timing and a.dtype == "bf16":
            try:                                   # after and outside the timed region; must not lose the line
                line["rpe_index_config4"] = rpe_index_config4_leg()
            except Exception as e:
                sys.stderr.write(f"[bench] rpe_index config-4 leg failed: {e}\n")
            try:
                line["irpe_config4"] = irpe_config4_leg()
                pm_ = pmc_kernels("irpe_attn")
                if pm_:
                    line["irpe_config4"]["qkt_mfma_util"] = dict(source=pm_["source"], kernels={k: v.get("mfma_util") for k, v in pm_["kernels"].items()})
            except Exception as e:
                sys.stderr.write(f"[bench] iRPE config-4 leg failed: {e}\n")
            try:
                del trainer, model, opt, reducer, images, target
                torch.cuda.empty_cache()
                line["irpe_config4"]["model"] = deit_config4_model_leg()
            except Exception as e:
                sys.stderr.write(f"[bench] DeiT config-4 model leg failed: {e}\n")
            try:
                line["tinyclip_config5"] = tinyclip_config5_leg()
            except Exception as e:
                sys.stderr.write(f"[bench] TinyCLIP config-5 leg failed: {e}\n")
            try:
                line["input_transform"] = input_transform_leg()
            except Exception as e:
                sys.stderr.write(f"[bench] input-transform leg failed: {e}\n")
        head, extra = compact_line(line)
        try:                                       # the side legs and the long-form notes: next to the headline, not in it
            with open(os.environ.get("CREAM_BENCH_EXTRA", "bench_extra.json"), "w") as fh:
                json.dump(extra, fh, indent=1)
        except OSError as e:
            sys.stderr.write(f"[bench] bench_extra.json not written: {e}\n")
        print(json.dumps(head), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

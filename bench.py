#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the SAVi PPO rollout-and-update hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the one the metric is quoted on): NUM_ENVS=64 per GPU, full
pi_g / pi_l / pi_q three-policy stack with the CLIP ViT-B/32 text tower frozen, synthetic 128x128 RGB-D +
2x257x101 binaural-spectrogram observations resident in HBM, T=150-step rollouts, then the pi_q PPO update
(2 epochs x 2 minibatches, interactive 1st-stage yaml).  One "step" = one full rollout+update cycle
(N*T env-steps per GPU).  Environments shard across GPUs (weak scaling); the only collective is the RCCL
all-reduce of pi_q's flat gradient, once per optimiser step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--rollout", type=int, default=150)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--spectrogram", default="257x101")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-share", action="store_true", help="do not batch the three policies' visual towers")
    return ap.parse_args()


def kernel_roofline(prec_name):
    """Roofline of the DOMINANT kernel by GPU time (profiles/r01_rocprof_summary.md): g2_kernel<128,16,4,1,2>, the 3x3
    16->16 implicit-GEMM conv of the ResNet towers' layer1 with fused GroupNorm statistics, as launched in a rollout step
    (six towers x 64 envs = 384 images of 64x64x16).  HBM-bound: N=16 output channels give 36 FLOP per byte.
    `achieved` = algorithmic bytes (bf16 activation read once + fp32 raw output written once) / duration measured live with
    HIP events on the launch stream; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json, FETCH_SIZE x2 per the gfx950 note).  The MFMA-bound GEMM of the CLIP MLP is reported
    beside it as `mfma_gemm`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_probe as rp
    sec = rp.measure()
    ab = rp.algorithmic_bytes()
    traffic = None
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["traffic_bytes_gfx950_corrected"]
    except Exception:
        pass
    ach = ab / sec / 1e9
    out = {"bound": "hbm", "kernel": "g2_kernel<128,16,4,1,2> (tower layer1 conv3x3 16->16 @64x64, 384 images/launch, bf16 in, "
                                     "fp32 out, fused GN stats)", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
           "frac": round(ach / 8000.0, 4), "traffic": traffic, "algorithmic_bytes": ab, "us_per_launch": round(sec * 1e6, 2)}
    out["mfma_gemm"] = mfma_gemm_rate(prec_name)
    return out


def mfma_gemm_rate(prec_name):
    """bf16 MFMA GEMM of the CLIP MLP up-projection (M = 64 envs x 77 tokens = 4928, K = 512, N = 2048, QuickGELU)."""
    from avlen_amd import _lib as L
    from avlen_amd.engine import P
    M, N, K = 64 * 77, 2048, 512
    dev = "cuda"
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    st = L.stream()
    if prec_name == "bf16":
        A16, W16 = A.bfloat16(), W.bfloat16()
        C16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        nb = L.lib.avlen_gemm_bf16_workspace_bytes(M, N)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda: L.call("avlen_gemm_bf16", P(A16), K, P(W16), K, None, N, P(C16), N, P(b), None, 0, M, N, K, 2, P(ws),
                             nb, st)
        name, peak = "g2_kernel<128,128,2,2,2> bf16 glds", 2500.0
    else:
        Cc = torch.empty(M, N, device=dev)
        nb = L.lib.avlen_gemm_workspace_bytes(M, N, K, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda: L.call("avlen_gemm", P(A), K, 0, P(W), K, 0, P(Cc), N, P(b), None, 0, M, N, K, 2, L.PREC_FP32, 1, 0.0,
                             P(ws), nb, st)
        name, peak = "igemm_kernel<128,fp32>", 157.3
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 50
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / iters
    ach = 2.0 * M * N * K / sec / 1e12
    return {"kernel": f"{name} (CLIP c_fc M=4928 N=2048 K=512)", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "us_per_launch": round(sec * 1e6, 2)}


def cpu_baseline(spec_hw):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow                                            # the oracle: checker/baseline only, never the product
    specs = json.load(open(os.path.join(ROOT, "tests", "golden", "param_specs.json")))
    cores = os.cpu_count() or 1
    n, t = 16, 3
    eps, sec, thr = flow.cpu_baseline(specs, N=n, T=t, spectrogram=spec_hw, pretraining=True, threads=min(cores, 64))
    return {"value": round(eps, 3), "unit": "env-steps/s", "cores": thr, "kind": "port",
            "sample": f"oracle (plain PyTorch fp32 restatement), {n} envs x {t} steps of the 3-policy rollout incl. CLIP "
                      f"text + one pi_q PPO update (2 epochs x 2 minibatches), {sec:.1f} s"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")                    # RCCL over xGMI
    from avlen_amd.harness import Workload
    H, W = (int(x) for x in a.spectrogram.split("x"))
    wl = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=True, seed=rank,
                  use_graphs=not a.no_graphs, share_encoders=not a.no_share)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.cycle()
    barrier()
    t0 = time.perf_counter()
    t_roll = 0.0
    for _ in range(a.steps):
        s0 = time.perf_counter()
        for _ in range(wl.T):
            wl.rollout_step()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - s0
        wl.update()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    env_steps = a.steps * a.envs * a.rollout * world
    out = {
        "metric": "env-steps/sec (encoder+GRU+PPO update) at NUM_ENVS=64", "value": round(env_steps / dt, 2),
        "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": "savi_interactive_1st_stage: pi_g+pi_l+pi_q rollout (CLIP ViT-B/32 text frozen) + pi_q "
                               "PPO update 2x2", "num_envs_per_gpu": a.envs, "rollout_steps": a.rollout,
                   "spectrogram": a.spectrogram, "parallelism": f"env-shard x{world}, RCCL grad all-reduce",
                   "rollout_fraction_of_time": round(t_roll / dt, 3)},
    }
    if rank == 0:
        if not a.no_roofline:
            out["roofline"] = kernel_roofline(a.precision)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline((H, W))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

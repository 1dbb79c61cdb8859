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

# The stream layout of the rollout (main + text + pi_g/pi_l streams) is tuned for the runtime's default of 4 hardware queues
# per process (measured: 2/3/5/6/8 queues give 22k/22k/16k/25k/25k env-steps/s against 29k): pin it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--rollout", type=int, default=150)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--spectrogram", default="257x101")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-share", action="store_true", help="do not batch the three policies' visual towers")
    ap.add_argument("--no-launch-ahead", action="store_true", help="call the three policies strictly one after the other")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2],
                    help="1 = savi_interactive_1st_stage (pretraining=True, the metric's config); 2 = 2nd stage: pi_q attends over "
                         "its 300-slot memory history in rollout and update (BASELINE configs[3] runs it at 32 envs per GPU)")
    ap.add_argument("--belief", action="store_true",
                    help="also run BeliefPredictor.update every step (SURVEY 8f rank 1; needs --spectrogram 65x26, the only size "
                         "the reference's predictor.fc accepts)")
    return ap.parse_args()


def kernel_roofline(prec_name):
    """Roofline of the DOMINANT kernel by GPU time in the rollout (profiles/r01_rocprof_summary.md):
    g2_kernel<64,128,2,4,2,512>, the bf16 MFMA GEMM (8-wave ping-pong tile, global_load_lds staging) on its heaviest call site, the
    CLIP text MLP up-projection c_fc of one rollout step on the ragged batch (M = 2464 live rows, N = 2048, K = 512, bias +
    QuickGELU, bf16 out); the block's other three GEMMs are listed under `other_call_sites`.  `achieved` = 2*M*N*K / duration measured live with HIP events on the launch stream; `peak` = dense
    bf16 MFMA; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes (profiles/r01_pmc_traffic.json: 2 x FETCH_SIZE +
    WRITE_SIZE, gfx950 correction).  The HBM-bound kernel class (direct 3x3 conv of the towers' layer 1) is reported beside it
    as `hbm_conv`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_probe as rp
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    except Exception:
        pass
    if prec_name != "bf16":
        return None
    sg, sc = rp.measure(rp.make_gemm), rp.measure(rp.make_conv)
    gw, cw = rp.gemm_work(), rp.conv_work()
    tf = gw["flops"] / sg / 1e12
    gb = cw["bytes"] / sc / 1e9
    return {"bound": "mfma", "kernel": "g2_kernel<64,128,2,4,2,512> bf16 glds GEMM (CLIP c_fc, ragged M=2464 N=2048 K=512)",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "traffic": pmc.get("gemm", {}).get("traffic_bytes"), "mfma_util_pmc_percent": pmc.get("gemm", {}).get("MfmaUtil_percent"),
            "algorithmic_flops": gw["flops"],
            "algorithmic_bytes": gw["bytes"], "us_per_launch": round(sg * 1e6, 2),
            "other_call_sites": rp.clip_call_sites(),
            "hbm_conv": {"bound": "hbm", "kernel": "dconv3x3_kernel<16,16,64,3> (tower layer-1 conv, 384 images/launch, bf16 "
                                                   "in/out, fused GN statistics)", "achieved": round(gb, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(gb / 8000.0, 4), "traffic": pmc.get("dconv", {}).get("traffic_bytes"),
                         "algorithmic_bytes": cw["bytes"], "us_per_launch": round(sc * 1e6, 2)}}


def cpu_baseline(spec_hw):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow                                            # the oracle: checker/baseline only, never the product
    specs = json.load(open(os.path.join(ROOT, "tests", "golden", "param_specs.json")))
    cores = os.cpu_count() or 1
    n, t = 32, 6                      # ~20 s of host work on the GPU box's cores
    eps, sec, thr = flow.cpu_baseline(specs, N=n, T=t, spectrogram=spec_hw, pretraining=True, threads=min(cores, 64))
    # SURVEY 8(d) asks for both thread settings: the reference pins torch to ONE thread (run.py:113)
    n1, t1 = 8, 8
    eps1, sec1, _ = flow.cpu_baseline(specs, N=n1, T=t1, spectrogram=spec_hw, pretraining=True, threads=1)
    return {"value": round(eps, 3), "unit": "env-steps/s", "cores": thr, "kind": "port",
            "sample": f"oracle (plain PyTorch fp32 restatement), {n} envs x {t} steps of the 3-policy rollout incl. CLIP "
                      f"text + one pi_q PPO update (2 epochs x 2 minibatches), {sec:.1f} s",
            "single_thread": {"value": round(eps1, 3), "cores": 1,
                              "sample": f"same flow, torch.set_num_threads(1) as the reference runs it, {n1} envs x {t1} steps, {sec1:.1f} s"}}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        # RCCL over xGMI ("nccl"); AVLEN_DIST_BACKEND=gloo rehearses the multi-rank flow on a single-GPU box
        dist.init_process_group(os.environ.get("AVLEN_DIST_BACKEND", "nccl"))
    from avlen_amd.harness import Workload
    H, W = (int(x) for x in a.spectrogram.split("x"))
    wl = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=(a.stage == 1), seed=rank,
                  use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead, belief_predictor=a.belief)

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
        last = wl.update()
    barrier()
    dt = time.perf_counter() - t0
    # outside the timed region: the run must have produced numbers (a kernel race shows up as NaN losses / memories, not a crash)
    import math
    ro = wl.rollouts
    finite = all(math.isfinite(float(x)) for x in last) and bool(torch.isfinite(ro.value_preds).all()) and \
        bool(torch.isfinite(ro.em_vln_dialog.memory).all()) and bool(torch.isfinite(ro.em.memory).all())
    assert finite, f"non-finite results after the timed cycles: losses {last}"
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
        "config": {"workload": f"savi_interactive_{'1st' if a.stage == 1 else '2nd'}_stage: pi_g+pi_l+pi_q rollout (CLIP ViT-B/32 text "
                               "frozen) + pi_q PPO update 2x2", "num_envs_per_gpu": a.envs, "rollout_steps": a.rollout,
                   "spectrogram": a.spectrogram, "parallelism": f"env-shard x{world}, RCCL grad all-reduce",
                   "rollout_fraction_of_time": round(t_roll / dt, 3), "belief_predictor": bool(a.belief)},
    }
    if rank == 0:
        if not a.no_roofline:
            rl = kernel_roofline(a.precision)
            if rl is not None:
                out["roofline"] = rl
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline((H, W))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                      # rank 0 may still be timing the roofline kernel: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

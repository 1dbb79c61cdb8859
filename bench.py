#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the SAVi PPO rollout-and-update hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1 without a launcher: bench.py starts the N ranks itself, one process per GPU, before the parent touches the GPU;
   under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it reads RANK/LOCAL_RANK/WORLD_SIZE.)

Workload (BASELINE.json configs[2], the one the metric is quoted on): NUM_ENVS=64 per GPU, full
pi_g / pi_l / pi_q three-policy stack with the CLIP ViT-B/32 text tower frozen, synthetic 128x128 RGB-D +
2x257x101 binaural-spectrogram observations resident in HBM, T=150-step rollouts, then the pi_q PPO update
(2 epochs x 2 minibatches, interactive 1st-stage yaml).  One "step" = one full rollout+update cycle
(N*T env-steps per GPU).  Environments shard across GPUs (weak scaling); the only collective is the RCCL
all-reduce of pi_q's flat gradient, once per optimiser step.

Besides the headline line, rank 0 of a 1-GPU run adds (each skippable, see the flags):
  roofline          dominant kernel, isolated AND in situ (FLOPs routed through it per rollout step / its per-step time)
  cpu_baseline      the oracle timed on the host cores at N=64 (bounded T), 1 thread and all threads
  bf16x3_vs_fp32,   the benched mode (bf16x3) and the plain-bf16 mode against the fp32 parity mode on identical weights /
  bf16_vs_fp32      observations / seeds: max |value|, |prob| differences and the sampled-action flips, per step on the SAME state
  bf16_fast_mode    env-steps/s of the same cycle with precision="bf16" (plain bf16 operands: faster, outside the 1e-3 tolerance)
  fp32_parity_mode  env-steps/s of the same cycle with precision="fp32" (the mode the bit-exact tests run)
  integration       env-steps/s with the tokens-known-ahead ordering of round 3 (text_ahead_synthetic), cached step views
                    (round-1 harness), share_encoders only, and the imports-only integration (no sharing, no launch-ahead)
  reference_dialog_process   the same cycle when the tokens follow the trainer's dialog process (3 steps after a_q == 1, else
                    zeros): the memoised text tower runs only the rows that changed
  gru_baseline      BASELINE configs[1]: N=16 GRU policy rollout + PPO 4x2 update
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The stream layout of the rollout (main + text + pi_g/pi_l streams) is tuned for the runtime's default of 4 hardware queues
# per process (measured: 2/3/5/6/8 queues give 22k/22k/16k/25k/25k env-steps/s against 29k): pin it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--rollout", type=int, default=150)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16", "fp32"],
                    help="bf16x3 (default, the mode that meets north_star's 1e-3 / no-flip tolerance): compensated bf16 (hi + lo operands, "
                         "3 MFMAs per product) towers and state encoders, fp16 CLIP text tower / AudioCNN; bf16: plain bf16 operands "
                         "(faster, ~0.1 off on the values); fp32: exact fp32 MFMA (the bit-exact parity mode)")
    ap.add_argument("--spectrogram", default="257x101")
    ap.add_argument("--config", default="interactive", choices=["interactive", "gru"],
                    help="interactive = BASELINE configs[2] (default); gru = configs[1] (N=16 GRU baseline, PPO 4x2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip bf16_vs_fp32 / fp32_parity_mode / integration / gru records")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-share", action="store_true", help="do not batch the three policies' visual towers")
    ap.add_argument("--no-launch-ahead", action="store_true", help="call the three policies strictly one after the other")
    ap.add_argument("--cached-views", action="store_true",
                    help="keep the per-step view objects of the storage (the reference trainer slices fresh ones every step)")
    ap.add_argument("--distractor", action="store_true", help="BASELINE configs[4]: use_category_input (F = 297 / 329)")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2],
                    help="1 = savi_interactive_1st_stage (pretraining=True, the metric's config); 2 = 2nd stage: pi_q attends over "
                         "its 300-slot memory history in rollout and update (BASELINE configs[3] runs it at 32 envs per GPU)")
    ap.add_argument("--dialog-tokens", default="after_option", choices=["after_option", "ahead"],
                    help="after_option (default) = the reference's data flow: the step's dialog tokens exist only after pi_q's "
                         "action was sampled (ppo_trainer.py:347, 449-593); ahead = tokens known at the start of the step (synthetic)")
    ap.add_argument("--dialog-process", default="fresh", choices=["fresh", "reference"],
                    help="fresh = a new random dialog for every env every step (SURVEY 8d input); reference = the trainer's process: "
                         "tokens persist NUM_DIALOG_STEPS = 3 steps after a_q == 1, zeros otherwise")
    ap.add_argument("--belief", action="store_true",
                    help="also run BeliefPredictor.update every step (SURVEY 8f rank 1; needs --spectrogram 65x26, the only size "
                         "the reference's predictor.fc accepts)")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------------
# multi-rank entry: the parent never touches the GPU
# --------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N ranks (one process per GPU), wait, propagate the worst exit code.
    Rank 0 prints the JSON line on the inherited stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# --------------------------------------------------------------------------------------------------------------------
# roofline of the dominant kernel
# --------------------------------------------------------------------------------------------------------------------
def kernel_roofline(prec_name, in_situ=None, towers=None):
    """Roofline of the DOMINANT kernel by GPU time in the rollout (profiles/): the bf16 MFMA GEMM on its heaviest call site, the
    CLIP text MLP up-projection c_fc of one rollout step on the ragged batch (M = 2464 live rows, N = 2048, K = 512, bias +
    QuickGELU, bf16 out); the block's other three GEMMs are listed under `other_call_sites`.  `achieved` = 2*M*N*K / duration
    measured live with HIP events on the launch stream; `peak` = dense bf16 MFMA; `traffic` = HBM bytes per launch from the
    rocprofv3 PMC passes (profiles/*_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_probe as rp
    pmc = {}
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            break
        except Exception:
            pass
    if prec_name == "fp32":
        return None
    rp.FMT = 1 if prec_name == "bf16x3" else 0          # the CLIP GEMMs of the bf16x3 mode run on fp16 operands
    h16 = "fp16" if rp.FMT else "bf16"
    sg = rp.measure(rp.make_gemm)
    gw = rp.gemm_work()
    tf = gw["flops"] / sg / 1e12
    out = {"bound": "mfma", "kernel": rp.GEMM_KERNEL_NAME + f" {h16} glds GEMM (CLIP c_fc, ragged M=2464 N=2048 K=512)",
           "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
           "traffic": pmc.get("gemm", {}).get("traffic_bytes"), "mfma_util_pmc_percent": pmc.get("gemm", {}).get("MfmaUtil_percent"),
           "algorithmic_flops": gw["flops"],
           "algorithmic_bytes": gw["bytes"], "us_per_launch": round(sg * 1e6, 2),
           "other_call_sites": rp.clip_call_sites()}
    if in_situ:
        ct = pmc.get("clip_tower", {})
        in_situ = dict(in_situ, traffic=ct.get("traffic_bytes"), mfma_util_pmc_percent=ct.get("MfmaUtil_percent"))
        out["in_situ"] = in_situ
    if towers:
        out["towers_fused"] = towers
        # The dominant kernel of the cycle is the visual-tower launch (bf16x3: tower_x3_kernel, 150 x 0.5 ms in the rollout + 4 x 11 ms
        # in the update = 36 % of the cycle's GPU time; bf16: tower_head + tower_tail): IT is the top-level record, measured in situ
        # (the product's grouped call at the benched batch, HIP events on its launch stream).  `achieved` counts the ALGORITHMIC
        # conv FLOPs (581.4 MF per tower pair and sample) -- in bf16x3 every product is three bf16 MFMAs, so the matrix
        # pipe issues 3x that (`mfma_issue_frac`).  The stand-alone GEMM probe that used to be the top-level record is `gemm_probe`.
        probe = {k: out[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mfma_util_pmc_percent",
                                     "algorithmic_flops", "algorithmic_bytes", "us_per_launch", "other_call_sites")}
        for k in probe:
            out.pop(k, None)
        x3 = prec_name == "bf16x3"
        tpmc = pmc.get("tower_x3" if x3 else "towers", {})
        top = {"bound": "mfma",
               "kernel": ("tower_x3_kernel (persistent work queue: stem + layer 1 items, layer 2-4 items; compensated bf16) + fc GEMM"
                          if x3 else "tower_head_kernel + tower_tail_kernel + fc GEMM") + ": " + towers["what"],
               "achieved": towers["TFLOPs"], "peak": 2500.0, "unit": "TFLOP/s", "frac": towers["frac_of_bf16_peak"],
               "traffic": tpmc.get("traffic_bytes"), "mfma_util_pmc_percent": tpmc.get("MfmaUtil_percent"),
               "algorithmic_flops": towers["flops"], "algorithmic_bytes": towers["algorithmic_hbm_bytes"],
               "us_per_launch": towers["us"], "measured": "the product's grouped call alone on its stream, HIP events on the launch stream"}
        step = towers.get("in_step")
        if step:
            # THE figure: the kernel's duration inside the timed rollout steps (stamped by the kernel itself, avlen_tower_x3_timing:
            # ticket-0 workgroup's start to the last workgroup's exit, device wall clock), beside the other streams' work; the
            # stand-alone figure (grouped call incl. its fc GEMM, nothing else running) stays as `isolated`
            tf = towers["flops"] / (step["us"] * 1e-6) / 1e12
            top["isolated"] = {"achieved": top["achieved"], "frac": top["frac"], "us_per_launch": top["us_per_launch"],
                               "measured": top["measured"]}
            top.update(achieved=round(tf, 1), frac=round(tf / 2500.0, 4), us_per_launch=step["us"],
                       measured="in situ: mean over the %d tower launches of the timed rollout steps, stamped by the kernel on the "
                                "device wall clock (first ticket to last exit)" % step["launches"])
        if x3:
            top["mfma_issue_frac"] = round(3 * top["frac"], 4)
            try:                                     # per conv INSIDE the fused bodies: from the committed phase-stamp profile (a lab
                import tower_x3_phase_table as tpt   # build of the same kernel, tools/x3_lab.hip), not re-measured by this run
                ph = next(f for f in ("r05_tower_x3_phases.txt", "r04_tower_x3_phases.txt", "r03_tower_x3_phases.txt")
                          if os.path.exists(os.path.join(ROOT, "profiles", f)))
                t = tpt.table(os.path.join(ROOT, "profiles", ph))
                top["tower_convs_in_situ"] = {"source": "profiles/" + ph + " (tools/x3_lab: phase stamps of the product kernel, "
                                                        "384 images; algorithmic FLOPs of a conv over its whole phase incl. GroupNorm "
                                                        "statistics and the normalise / split pass, per CU)",
                                              "whole_tower_frac_mfma_algorithmic": t["whole_tower_frac_mfma_algorithmic"],
                                              "whole_tower_frac_mfma_issue": t["whole_tower_frac_mfma_issue"],
                                              "rows": [{"conv": r["conv"], "us": r["us"], "frac_mfma": r["frac_mfma_algorithmic"],
                                                        "frac_mfma_issue": r["frac_mfma_issue"]} for r in t["rows"]]}
            except Exception as e:
                top["tower_convs_in_situ_error"] = repr(e)
        top.update(out)
        top["gemm_probe"] = probe
        out = top
    return out


def towers_fused(wl):
    """The visual towers of ONE rollout step at the benched batch as the product runs them -- one grouped call for the six towers
    (rgb + depth of pi_q, pi_g, pi_l) x N images: tower_head (preprocessing + stem + layers 1-2), tower_tail (layers 3-4), fc
    GEMM -- timed with HIP events on the launch stream while nothing else runs.  FLOPs: SURVEY 8(d), 581.4 MF per tower pair
    and sample; algorithmic HBM bytes: the images read once (uint8 rgb, fp32 depth) + the (N, 64) features written."""
    import torch
    grp = getattr(wl.pi_q, "_enc_group", None)
    if grp is None:
        return None
    obs = {k: v[0] for k, v in wl.rollouts.observations.items()}
    rgb, depth = obs["rgb"], obs["depth"]
    fn = lambda: grp.run_all(wl.pi_q, rgb, depth)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / it
    N, pairs = rgb.shape[0], len(grp.members)
    flops = 581.4e6 * N * pairs
    by = pairs * N * (rgb[0].numel() * rgb.element_size() + depth[0].numel() * depth.element_size() + 2 * 64 * 4)
    return {"what": "%d towers x %d images, one grouped call, alone on its stream" % (2 * pairs, N),
            "us": round(sec * 1e6, 1), "flops": flops, "TFLOPs": round(flops / sec / 1e12, 1),
            "frac_of_bf16_peak": round(flops / sec / 2.5e15, 4), "us_per_tower_image": round(sec * 1e6 / (2 * pairs * N), 3),
            "algorithmic_hbm_bytes": by, "GBps": round(by / sec / 1e9, 1)}


def text_tower_in_situ(wl):
    """CLIP text tower of ONE rollout step at the benched batch, timed with HIP events on the stream it is launched on while
    nothing else runs: FLOPs through the GEMM family per step / that time.  (The rocprof per-step table in profiles/ gives the
    same quantity with the towers running beside it.)"""
    import torch
    pol = wl.pi_l
    if pol is None:
        return None
    toks = wl.dialog[0]
    st = torch.cuda.Stream()
    live = int((toks.argmax(-1) + 1).sum())                      # rows that reach the GEMMs (EOT position + 1 per dialog)
    N = toks.shape[0]
    w = 512
    # per live row and layer: in_proj 3w^2, out_proj w^2, c_fc 4w^2, c_proj 4w^2 MACs; the last layer's MLP and out_proj run on
    # the N pooled rows only
    flops = 2.0 * (11 * live * 12 * w * w + live * 3 * w * w + N * 9 * w * w)
    def once(after):
        pol.net.invalidate_text_cache()                  # every dialog new, as in the fresh-token rollout (the memo would skip them all)
        pol.prefetch_text(toks, st, after_current=after)
    for _ in range(3):
        once(True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    with torch.cuda.stream(st):
        e0.record(st)
    for _ in range(it):
        once(False)
    with torch.cuda.stream(st):
        e1.record(st)
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / it
    impl = ("work list + one sequence-stationary launch (clip_tower_kernel: dialogs packed into <= 4-tile groups, 2- or 4-way column "
            "split) + ln_final + folded projection; memo emptied before every call" if pol._engine()["clip"].wstream
            else "launch-per-GEMM chain")
    return {"what": "CLIP text tower graph of one rollout step (" + impl + "), alone on its stream", "live_rows": live, "ms": round(sec * 1e3, 4),
            "gemm_flops": flops, "TFLOPs": round(flops / sec / 1e12, 1), "frac_of_bf16_peak": round(flops / sec / 2.5e15, 4)}


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, "port")
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(spec_hw, envs):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import flow                                            # the oracle: checker/baseline only, never the product
    specs = json.load(open(os.path.join(ROOT, "tests", "golden", "param_specs.json")))
    cores = os.cpu_count() or 1
    thr = min(cores, 64)
    # N = NUM_ENVS of the metric, 1 warm-up, >= 3 timed cycles, same call order; T is shortened (150 steps of the
    # oracle at N=64 would take ~15 min) -- the rollout cost per step does not depend on T, the update's does (T*N samples), so the
    # shortened cycle keeps the rollout:update proportion of samples
    def timed(n, t, threads, reps):
        vals, secs = [], 0.0
        for _ in range(reps):
            eps, sec, _ = flow.cpu_baseline(specs, N=n, T=t, spectrogram=spec_hw, pretraining=True, threads=threads)
            vals.append(eps); secs += sec
        vals.sort()
        return vals[len(vals) // 2], [round(v, 3) for v in vals], secs
    flow.cpu_baseline(specs, N=4, T=1, spectrogram=spec_hw, pretraining=True, threads=thr)                       # warm-up
    n, t = envs, 2
    med, allv, sec = timed(n, t, thr, 3)
    flow.cpu_baseline(specs, N=2, T=1, spectrogram=spec_hw, pretraining=True, threads=1)                         # warm-up
    n1, t1 = envs, 1
    med1, all1, sec1 = timed(n1, t1, 1, 2)
    # BASELINE configs[0]: the reference's own CPU-runnable case, NUM_ENVS = 1 (num_mini_batch = 1), one thread (run.py:113)
    e0 = [flow.cpu_baseline(specs, N=1, T=8, spectrogram=spec_hw, pretraining=True, threads=1, mini_batches=1) for _ in range(4)][1:]
    v0 = sorted(x[0] for x in e0)
    return {"value": round(med, 3), "unit": "env-steps/s", "cores": thr, "kind": "port",
            "sample": f"oracle (plain PyTorch fp32 restatement), {n} envs x {t} steps of the 3-policy rollout incl. CLIP "
                      f"text + one pi_q PPO update (2 epochs x 2 minibatches); median of 3 timed cycles after a warm-up pass "
                      f"({sec:.1f} s)", "timed_cycles": allv,
            "single_thread": {"value": round(med1, 3), "cores": 1, "timed_cycles": all1,
                              "sample": f"same flow, torch.set_num_threads(1) as the reference runs it (run.py:113), {n1} envs x "
                                        f"{t1} step + update; median of 2 timed cycles ({sec1:.1f} s)"},
            "cfg1_single_env": {"value": round(v0[len(v0) // 2], 3), "cores": 1, "timed_cycles": [round(v, 3) for v in v0],
                                "sample": "BASELINE configs[0]: NUM_ENVS=1, num_mini_batch=1, one thread; 1 env x 8 steps of the "
                                          "3-policy rollout + one pi_q update (2 epochs x 1 minibatch); median of 3 timed cycles "
                                          "after one warm-up cycle"}}


# --------------------------------------------------------------------------------------------------------------------
# bf16 (benched) vs fp32 (parity) mode
# --------------------------------------------------------------------------------------------------------------------
def modes_vs_fp32(a, H, W, wls):
    """The fast modes against the fp32 parity mode: same weights (same weight seed + the headline's trained pi_q), same synthetic
    observations, same host RNG state.  Per step t the fp32 workload's state (storage views, memories written by fp32 forwards) is
    fed to every mode's policies with the host generator rewound in between, so `flips` counts steps where that mode's arithmetic
    alone changed the sampled action (no trajectory divergence mixed in).  wls: {mode name: Workload}."""
    import torch
    from avlen_amd.harness import Workload
    wl32 = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision="fp32", pretraining=(a.stage == 1), seed=0,
                    use_graphs=not a.no_graphs, share_encoders=False, launch_ahead=False, distractor=a.distractor)
    wl32.pi_q.load_state_dict(next(iter(wls.values())).pi_q.state_dict())      # the timed cycles have trained pi_q
    T = a.rollout
    keys = ("q_value", "q_prob", "g_prob", "l_prob", "g_value", "l_value")
    mv = {m: {k: 0.0 for k in keys} for m in wls}
    flips = {m: {"q": 0, "g": 0, "l": 0} for m in wls}
    torch.manual_seed(4242)
    for t in range(T):
        rng = torch.get_rng_state()
        outs = {}
        for m, wl in wls.items():
            torch.set_rng_state(rng)
            outs[m] = {k: v.clone() for k, v in wl.policies_on(wl32, t).items()}
        torch.set_rng_state(rng)
        o32 = wl32.rollout_step(return_outs=True)
        for m in wls:
            for k in keys:
                mv[m][k] = max(mv[m][k], float((outs[m][k] - o32[k]).abs().max()))
            for k in flips[m]:
                flips[m][k] += int((outs[m]["a_" + k] != o32["a_" + k]).sum())
    torch.cuda.synchronize()
    n = T * a.envs
    # fp32 parity mode throughput: one more full cycle (the rollout above + this update warmed everything up)
    wl32.update()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    wl32.cycle()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    recs = {}
    for m in wls:
        v, f = mv[m], flips[m]
        recs[m] = {"samples": n, "how": "per step on the fp32 workload's state, host RNG rewound between the modes; N, T as benched; "
                                        "every mode holds the weights the timed cycles trained",
                   "max_abs_value": round(max(v["q_value"], v["g_value"], v["l_value"]), 6),
                   "max_abs_prob": round(max(v["q_prob"], v["g_prob"], v["l_prob"]), 7),
                   "sampled_action_flips": f["q"] + f["g"] + f["l"],
                   "action_flip_rate": round((f["q"] + f["g"] + f["l"]) / (3.0 * n), 6),
                   "within_1e-3_and_no_flips": bool(max(v.values()) <= 1e-3 and f["q"] + f["g"] + f["l"] == 0),
                   "per_policy": {"pi_" + k: {"max_abs_value": round(v[k + "_value"], 6), "max_abs_prob": round(v[k + "_prob"], 7),
                                              "flips": f[k]} for k in ("q", "g", "l")}}
    fp32 = {"value": round(a.envs * T / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2),
            "what": "same cycle, precision='fp32' (fp32 MFMA, deterministic reductions; the mode of the bit-exact parity "
                    "tests), graphs on, no encoder sharing / launch-ahead"}
    del wl32
    torch.cuda.empty_cache()
    return recs, fp32


def time_cycles(wl, warmup, steps):
    import torch
    for _ in range(warmup):
        wl.cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.cycle()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def integration_records(a, H, W):
    import torch
    from avlen_amd.harness import Workload
    out = {}
    for name, kw in (("text_ahead_synthetic", dict(dialog_tokens="ahead")),
                     ("cached_views", dict(cached_views=True)),
                     ("share_only", dict(share_encoders=True, launch_ahead=False)),
                     ("imports_only", dict(share_encoders=False, launch_ahead=False))):
        kws = dict(spectrogram=(H, W, 2), precision=a.precision, pretraining=(a.stage == 1), seed=0, use_graphs=not a.no_graphs,
                   share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead, distractor=a.distractor,
                   dialog_tokens=a.dialog_tokens, dialog_process=a.dialog_process)
        kws.update(kw)
        if kws["dialog_tokens"] == "ahead":
            kws["dialog_process"] = "fresh"
        wl = Workload(a.envs, a.rollout, **kws)
        dt = time_cycles(wl, 2, 3)
        out[name] = {"value": round(a.envs * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2)}
        del wl
        torch.cuda.empty_cache()
    out["what"] = ("headline = fresh storage views every step (as ppo_trainer.py:375-391 slices them), encoder sharing + "
                   "launch-ahead calls added to the trainer, dialog tokens issued after act_option as the reference's data flow "
                   "dictates; text_ahead_synthetic = the round-3 ordering: the step's tokens are known before act_option and the "
                   "text tower is launched beside the visual towers (NOT obtainable from the reference trainer, "
                   "ppo_trainer.py:449-593); cached_views = the round-1 harness (view objects kept per step "
                   "slot); share_only = the imports + ONE share_encoders(pi_q, pi_g, pi_l, rollouts=rollouts) call where the trainer builds the policies, "
                   "no per-step prefetch_* calls; imports_only = the three-import-lines integration: no share_encoders, no "
                   "prefetch_* calls")
    return out


def reference_dialog_record(a, H, W):
    """The same cycle under the reference's dialog PROCESS (ppo_trainer.py:347, 463-469, 582-587, 763-765): an env outside a dialog
    that samples a_q == 1 gets a new instruction, keeps it for NUM_DIALOG_STEPS = 3 steps, every other env presents all-zero
    tokens -- so the memoised text tower runs only the rows that changed.  pi_q is untrained here (p(query) ~ 0.5): about one env in
    five starts a dialog per step, far more than a trained pi_q asks for."""
    import torch
    from avlen_amd.harness import Workload
    wl = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=(a.stage == 1), seed=0,
                  use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                  distractor=a.distractor, dialog_tokens="after_option", dialog_process="reference")
    dt = time_cycles(wl, 2, 3)
    ds = wl.dialog_stats
    rec = {"value": round(a.envs * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2),
           "new_dialogs_per_step": round(ds["new_dialogs"] / max(ds["steps"], 1), 2),
           "rows_with_a_dialog_per_step": round(ds["active_rows"] / max(ds["steps"], 1), 2), "num_envs": a.envs,
           "what": "dialog tokens persist 3 steps after a_q == 1 and are zero otherwise, written by a host loop after act_option "
                   "(H2D of current_dialog / agent_step per step); the text tower is memoised per row "
                   "(avlen_clip_text_cached_fwd): only the new dialogs of a step run the 12 blocks"}
    del wl
    torch.cuda.empty_cache()
    return rec


def feature_reuse_record(a, H, W):
    """Opt-in, reported beside the headline and never as it: pi_q's update reads the encoder feature columns back from the rows the
    rollout wrote into the option memory ring instead of re-running the frozen towers on the 4 x 4800 stored observations
    (PPO.feature_reuse; exact up to GEMM tiling order: tests/test_gpu_feature_reuse.py).  The reference recomputes them
    (ppo.py:207-262), and so does the headline."""
    import torch
    from avlen_amd.harness import Workload
    wl = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=(a.stage == 1), seed=0,
                  use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                  distractor=a.distractor, dialog_tokens=a.dialog_tokens)
    wl.agent.feature_reuse = True
    dt = time_cycles(wl, 2, 3)
    for _ in range(wl.T):
        wl.rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.update()
    torch.cuda.synchronize()
    up = time.perf_counter() - t0
    rec = {"value": round(a.envs * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2),
           "update_ms": round(up * 1e3, 2),
           "what": "same cycle with PPO.feature_reuse = True: the update's frozen-encoder features come from the stored memory rows "
                   "(policy.py:1062-1065) instead of a recompute; skips work the metric names, hence not the headline"}
    del wl
    torch.cuda.empty_cache()
    return rec


def stage2_record(a, H, W):
    """BASELINE configs[3]'s per-GPU share: 2nd stage (pi_q attends over its 300-slot memory history in rollout AND update), 32 envs.
    The update is the SMT transformer forward / backward over (M + 1) x T x N_mb = 722 k token rows per minibatch: 9600 sample-passes
    x 3 (fwd + bwd) x 498.6 MF = 14.4 TFLOP per update."""
    import torch
    from avlen_amd.harness import Workload
    N = 32
    wl = Workload(N, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=False, seed=0,
                  use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                  dialog_tokens=a.dialog_tokens)
    dt = time_cycles(wl, 1, 2)
    for _ in range(wl.T):
        wl.rollout_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.update()
    torch.cuda.synchronize()
    up = time.perf_counter() - t0
    flops = 2 * 2 * (N // 2) * a.rollout * 3 * 498.6e6           # epochs x minibatches x samples x (fwd + bwd) x SMT block
    rec = {"value": round(N * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2), "num_envs": N,
           "update_ms": round(up * 1e3, 2),
           "update_roofline": {"bound": "mfma", "achieved": round(flops / up / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                               "frac": round(flops / up / 2.5e15, 4), "algorithmic_flops": flops,
                               "what": "pi_q PPO update 2 x 2 over 32 x 150 samples with M = 300 memory rows each (SMT fusion MLP + "
                                       "encoder / decoder layer forward + backward), whole update incl. the frozen towers' forward, "
                                       "GAE, losses, Adam"},
           "what": "savi_interactive_2nd_stage at NUM_ENVS = 32 (one rank's share of BASELINE configs[3]), dtype as the headline"}
    del wl
    torch.cuda.empty_cache()
    return rec


def belief_record(a):
    """BeliefPredictor.update every rollout step (65x26 spectrogram: the only size the reference's predictor.fc
    accepts), with and without it."""
    import torch
    from avlen_amd.harness import Workload
    out = {}
    for name, on in (("without", False), ("with_belief_predictor", True)):
        wl = Workload(a.envs, a.rollout, spectrogram=(65, 26, 2), precision=a.precision, pretraining=(a.stage == 1), seed=0,
                      use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                      belief_predictor=on, dialog_tokens=a.dialog_tokens)
        dt = time_cycles(wl, 1, 2)
        out[name] = {"value": round(a.envs * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2)}
        del wl
        torch.cuda.empty_cache()
    out["cost_percent"] = round(100.0 * (1.0 - out["with_belief_predictor"]["value"] / out["without"]["value"]), 1)
    out["what"] = "same cycle at the 65x26 spectrogram, BeliefPredictor.update (two ResNet-18s + the per-env filter) between the env step and insert"
    return out


def gru_record(a):
    """BASELINE configs[1]: NUM_ENVS=16, AudioCNN + VisualCNN + single-layer GRU pi_g, PPO 4 epochs x 2 minibatches, bf16."""
    import torch
    from avlen_amd.harness import GruWorkload
    H, W = (int(x) for x in a.spectrogram.split("x"))
    prec = a.precision
    wl = GruWorkload(16, a.rollout, spectrogram=(H, W, 2), precision=prec)
    dt = time_cycles(wl, 1, 3)
    rec = {"value": round(16 * a.rollout / dt, 2), "unit": "env-steps/s", "ms_per_step": round(dt * 1e3, 2),
           "config": {"workload": "AudioNavBaselinePolicy (AudioCNN + VisualCNN + GRU-512) rollout + PPO 4 epochs x 2 minibatches",
                      "num_envs": 16, "rollout_steps": a.rollout, "spectrogram": a.spectrogram, "dtype": prec}}
    if prec != "fp32":
        rec["vs_fp32"] = gru_vs_fp32(wl, a, H, W)
        other = "bf16" if prec == "bf16x3" else "bf16x3"
        wlo = GruWorkload(16, a.rollout, spectrogram=(H, W, 2), precision=other)
        dto = time_cycles(wlo, 1, 3)
        rec[other + "_mode"] = {"value": round(16 * a.rollout / dto, 2), "unit": "env-steps/s", "ms_per_step": round(dto * 1e3, 2)}
        del wlo
    del wl
    torch.cuda.empty_cache()
    return rec


def gru_vs_fp32(wl, a, H, W):
    """cfg2's benched mode against the fp32 parity mode on identical weights (incl. what the timed cycles trained), observations and
    host RNG: a whole rollout of the fp32 workload, the fast policy evaluated on the fp32 workload's state at every step."""
    import torch
    from avlen_amd.harness import GruWorkload
    w32 = GruWorkload(16, a.rollout, spectrogram=(H, W, 2), precision="fp32")
    w32.pol.load_state_dict(wl.pol.state_dict())
    mv = {"value": 0.0, "prob": 0.0, "hidden": 0.0}
    flips = 0
    torch.manual_seed(777)
    for t in range(w32.T):
        ro = w32.rollouts
        obs = {k: v[t] for k, v in ro.observations.items()}
        rng = torch.get_rng_state()
        v1, a1, _, h1, _, p1 = wl.pol.act(obs, ro.recurrent_hidden_states[t], ro.prev_actions[t], ro.masks[t], None, None)
        v1, a1, h1, p1 = v1.clone(), a1.clone(), h1.clone(), p1.clone()
        torch.set_rng_state(rng)
        v0, a0, lp0, h0, _, p0 = w32.pol.act(obs, ro.recurrent_hidden_states[t], ro.prev_actions[t], ro.masks[t], None, None)
        mv["value"] = max(mv["value"], float((v1 - v0).abs().max()))
        mv["prob"] = max(mv["prob"], float((p1 - p0).abs().max()))
        mv["hidden"] = max(mv["hidden"], float((h1 - h0).abs().max()))
        flips += int((a1 != a0).sum())
        ro.insert({k: w32.sim[k][t + 1] for k in ro.observations}, h0, a0, lp0, v0, w32.rewards[t], w32.not_done[t])
    torch.cuda.synchronize()
    n = w32.T * 16
    del w32
    torch.cuda.empty_cache()
    return {"samples": n, "max_abs_value": round(mv["value"], 6), "max_abs_prob": round(mv["prob"], 7),
            "max_abs_hidden": round(mv["hidden"], 6), "sampled_action_flips": flips,
            "within_1e-3_and_no_flips": bool(max(mv.values()) <= 1e-3 and flips == 0),
            "how": "per step on the fp32 workload's state (N = 16, T as benched), host RNG rewound between the modes; both hold the "
                   "weights the timed cycles trained"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus or a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    backend = os.environ.get("AVLEN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    assert backend != "nccl" or world <= ndev, f"{world} ranks need {world} GPUs, {ndev} visible"
    torch.cuda.set_device(local % ndev)
    if world > 1:
        import torch.distributed as dist
        # RCCL over xGMI ("nccl"); AVLEN_DIST_BACKEND=gloo rehearses the multi-rank flow on a single-GPU box
        dist.init_process_group(backend)
        assert dist.get_world_size() == world
    from avlen_amd.harness import Workload
    H, W = (int(x) for x in a.spectrogram.split("x"))
    if a.config == "gru":
        from avlen_amd.harness import GruWorkload
        wl = GruWorkload(a.envs if a.envs != 64 else 16, a.rollout, spectrogram=(H, W, 2), precision=a.precision, seed=rank)
        a.envs = wl.N
    else:
        wl = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=a.precision, pretraining=(a.stage == 1), seed=rank,
                      use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                      belief_predictor=a.belief, cached_views=a.cached_views, distractor=a.distractor,
                      dialog_tokens=a.dialog_tokens, dialog_process=a.dialog_process if a.dialog_tokens == "after_option" else "fresh")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.cycle()
    barrier()
    import ctypes
    from avlen_amd import _lib as L
    x3_us, x3_n = ctypes.c_double(0.0), ctypes.c_longlong(0)
    step_us = step_n = 0.0
    probe_x3 = rank == 0 and a.precision == "bf16x3" and a.config == "interactive" and not a.no_roofline
    if probe_x3:
        L.lib.avlen_tower_x3_timing(None, None, 1)
    t0 = time.perf_counter()
    t_roll = 0.0
    for _ in range(a.steps):
        s0 = time.perf_counter()
        for _ in range(wl.T):
            wl.rollout_step()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - s0
        if probe_x3:                                      # the rollout's tower launches only (the update's run 25x the images)
            L.lib.avlen_tower_x3_timing(ctypes.byref(x3_us), ctypes.byref(x3_n), 1)
            step_us += x3_us.value * x3_n.value
            step_n += x3_n.value
        last = wl.update()
        if probe_x3:
            L.lib.avlen_tower_x3_timing(None, None, 1)
    barrier()
    dt = time.perf_counter() - t0
    # outside the timed region: the run must have produced numbers (a kernel race shows up as NaN losses / memories, not a crash)
    import math
    assert all(math.isfinite(float(x)) for x in last) and wl.finite(), f"non-finite results after the timed cycles: losses {last}"
    if world > 1:
        tt = torch.tensor([dt], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    env_steps = a.steps * a.envs * a.rollout * world
    if a.config == "gru":
        workload = "AudioNavBaselinePolicy (AudioCNN + VisualCNN + GRU-512) rollout + PPO update 4x2 (BASELINE configs[1])"
    else:
        workload = (f"savi_interactive_{'1st' if a.stage == 1 else '2nd'}_stage{' (distractor)' if a.distractor else ''}: "
                    "pi_g+pi_l+pi_q rollout (CLIP ViT-B/32 text frozen) + pi_q PPO update 2x2")
    out = {
        "metric": f"env-steps/sec (encoder+GRU+PPO update) at NUM_ENVS={a.envs}", "value": round(env_steps / dt, 2),
        "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.precision, "data": "synthetic",
        "config": {"workload": workload, "num_envs_per_gpu": a.envs, "rollout_steps": a.rollout,
                   "spectrogram": a.spectrogram, "parallelism": f"env-shard x{world}, RCCL grad all-reduce",
                   "rollout_fraction_of_time": round(t_roll / dt, 3), "belief_predictor": bool(a.belief),
                   "step_views": "cached" if a.cached_views else "fresh", "encoder_sharing": not a.no_share,
                   "launch_ahead": not a.no_launch_ahead, "dialog_tokens": a.dialog_tokens,
                   "dialog_process": a.dialog_process},
    }
    if rank == 0:
        interactive = a.config == "interactive"
        fast = a.precision in ("bf16", "bf16x3")
        if not a.no_roofline:
            situ = text_tower_in_situ(wl) if (interactive and fast) else None
            tw = towers_fused(wl) if (interactive and fast) else None
            if tw is not None and step_n:
                tw["in_step"] = {"us": round(step_us / step_n, 1), "launches": int(step_n)}
            rl = kernel_roofline(a.precision, situ, tw)
            if rl is not None:
                out["roofline"] = rl
        if world == 1 and interactive and not a.no_extras and fast and not a.belief:
            wls = {a.precision: wl}
            other = "bf16" if a.precision == "bf16x3" else "bf16x3"
            wlo = Workload(a.envs, a.rollout, spectrogram=(H, W, 2), precision=other, pretraining=(a.stage == 1), seed=0,
                           use_graphs=not a.no_graphs, share_encoders=not a.no_share, launch_ahead=not a.no_launch_ahead,
                           distractor=a.distractor, dialog_tokens=a.dialog_tokens)
            wlo.pi_q.load_state_dict(wl.pi_q.state_dict())      # the timed cycles have trained pi_q: every mode holds the SAME weights
            wls[other] = wlo
            recs, out["fp32_parity_mode"] = modes_vs_fp32(a, H, W, wls)
            for m, r in recs.items():
                out[m + "_vs_fp32"] = r
            dto = time_cycles(wlo, 2, 3)
            out["bf16_fast_mode" if other == "bf16" else "bf16x3_mode"] = {
                "value": round(a.envs * a.rollout / dto, 2), "unit": "env-steps/s", "ms_per_step": round(dto * 1e3, 2),
                "what": f"same cycle, precision='{other}'"}
            del wl, wlo, wls
            torch.cuda.empty_cache()
            out["integration"] = integration_records(a, H, W)
            out["reference_dialog_process"] = reference_dialog_record(a, H, W)
            out["update_feature_reuse"] = feature_reuse_record(a, H, W)
            if a.stage == 1 and not a.distractor:
                out["stage2_envs32"] = stage2_record(a, H, W)
                out["belief_65x26"] = belief_record(a)
            import avlen_amd.harness as hz
            if hasattr(hz, "GruWorkload"):
                out["gru_baseline"] = gru_record(a)
        if world == 1 and not a.no_cpu_baseline and interactive:
            out["cpu_baseline"] = cpu_baseline((H, W), a.envs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                      # rank 0 may still be timing the roofline kernel: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

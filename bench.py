#!/usr/bin/env python3
"""bench.py — env-steps/s of one full PPO iteration (rollout + GAE + epochs x minibatches update) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 either under a launcher that sets
WORLD_SIZE (torch.distributed.run: one rank per GPU) or plain, in which case this script starts its own N ranks
(launch_ranks below).  A "step" = one PPO iteration of BASELINE.json configs[1]:
CartPole-v1, n_envs = 65 536 per GPU, hidden [64,64], PPO defaults, n_steps = 2048, synthetic fixed-length
episodes (termination disabled, truncation every 500 steps), batch_size = N/32 (BASELINE.json leaves it open;
SURVEY.md §8d) — all state resident in HBM before the timed region.  Prints ONE JSON line on rank 0, kept under 8 KB
(the driver keeps the last 8 KB of stdout): the other configs' runs appear in it as compact `secondary` entries and
in full on stderr (`--full`: in full in the line).

N > 1: envs are sharded (weak scaling, 65 536 per GPU), the only data-path collective is the RCCL all-reduce of
[gradients | loss sums] per optimiser step (+ the 3-double advantage-moment reduce), issued inside the library
on its own stream.  torch.distributed (gloo) is used only for rendezvous, the barrier and the max-over-ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

FLOP_FWD = 17_792            # 2*(4*64+64*64+64*2 + 4*64+64*64+64*1)
FLOP_FWD_BWD = 3 * FLOP_FWD  # a16: backward ~ 2x forward


def flop_fwd(D: int, H: int, A: int) -> int:
    return 2 * ((D * H + H * H + H * A) + (D * H + H * H + H))
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
# an fp32-equivalent contraction on the 16-bit matrix cores spends several MFMAs per product: SIX bf16 ones with three-piece operands (the generic contractions; the fused
# kernels until the end of round 3), THREE f16 ones with two-piece operands (the fused kernels since) — the kernel's own ceiling in delivered f32 flops
# is the dense 16-bit peak (the same for f16 and bf16) over that count
PEAK_BF16_SPLIT6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6
# HBM traffic (PMC) + rocprofv3 average of the dominant kernel, replayed from the committed profile of exactly that workload (each file names the commit it was taken at)
PMC_FILES = {("cartpole", 64): "r05_ppo_grad_pmc.json", ("pendulum", 256): "r05_wide_split_pmc.json"}
PEAK_HBM_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s (~ 6.3 TB/s is what a streaming kernel reaches)


def hbm_kernels(prof: dict, *, N: int, D: int, A: int, discrete: bool, P: int, epochs: int, steps: int, normalize: bool) -> list:
    """the bandwidth-class kernels of the path one by one — ALGORITHMIC bytes per launch class (what the data structure obliges the kernel to move,
    not what the caches saw) / HIP-event time of that class in the timed region / the 8 TB/s HBM peak.  Units: N = env-steps of one rollout (per GPU), P = parameters."""
    act = 4 if discrete else 4 * A
    recq = 2 if D <= 4 else 3
    rows = [
        ("rollout_kernel", 4 * D + act + 4 + 4 + 4 + 2, N, "buffer WRITE per env-step: obs 4D + action + reward + logprob + value + flags (the kernel is bound by its two MLP forwards, not by this)"),
        ("gae_kernel", 18, N, "read r, V, flags (+ V of the next row from cache) 10 B, write A, R 8 B"),
        ("pack_records_kernel", 32 + 16 * recq, N, f"read obs 4D + action + adv + logp + ret, write {recq} float4 per sample"),
        ("adv_moments_kernel", 8 * epochs, N, "per epoch: epoch_index_kernel writes 4 B per sample (the DataLoader order as int32), epoch_moments_kernel reads 4 B per sample"),
        ("explained_var_kernel", 8, N, "read V, R"),
        ("adam_kernel", 28, P, "per launch: read g, p, m, v, write p, m, v (a few hundred KB: launch-latency-bound, not bandwidth-bound)"),
    ]
    out = []
    for name, bpu, units, what in rows:
        k = prof.get(name)
        if not k or not k["launches"]:
            continue
        per_launch = name == "adam_kernel"
        total_bytes = bpu * units * (k["launches"] if per_launch else steps)
        gbps = total_bytes / (k["total_ms"] * 1e-3) / 1e9
        out.append({"kernel": name, "bytes_per_unit": bpu, "unit": "parameter, per launch" if per_launch else "env-step, per rollout", "what": what, "launches": k["launches"],
                    "ms_per_step": k["total_ms"] / steps, "achieved_GBps": gbps, "peak_GBps": PEAK_HBM_GBPS, "frac": gbps / PEAK_HBM_GBPS})
    return out


def mfma_roofline(ach_tflops: float, arith: str) -> dict:
    """peak / frac of a dense-contraction kernel.  A kernel that computes its fp32-equivalent products as THREE f16 MFMAs (two-piece operand split) or SIX bf16 MFMAs (three-piece
    split) is priced against ITS pipe: dense 16-bit peak / 3 = 838.9 or / 6 = 419.4 TFLOP/s of delivered f32 flops; the figure against the f32-MFMA peak (157.3, the pipe the exact-f32
    kernels run on and the roof SURVEY.md section 8d names) is kept beside it as frac_vs_f32_peak — it may exceed 1 and is never `frac`."""
    if "f16x2" in arith:          # two-piece f16 operands: three f16 MFMAs per fp32-equivalent product
        peak, note = PEAK_BF16_SPLIT6_TFLOPS * 2.0, "dense f16 MFMA peak 2516.6 / 3 MFMAs per fp32-equivalent product"
    elif "bf16x3" in arith:
        peak, note = PEAK_BF16_SPLIT6_TFLOPS, "dense bf16 MFMA peak 2516.6 / 6 MFMAs per fp32-equivalent product"
    else:
        peak, note = PEAK_F32_MFMA_TFLOPS, "dense f32 MFMA peak (v_mfma_f32_32x32x2_f32)"
    return {"bound": "mfma", "achieved": ach_tflops, "peak": peak, "unit": "TFLOP/s", "frac": ach_tflops / peak,
            "peak_note": note,
            "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "frac_vs_f32_peak": ach_tflops / PEAK_F32_MFMA_TFLOPS, "arithmetic": arith}


def cpu_baseline(pkg, seed: int) -> dict:
    """The CPU oracle (a C restatement of the reference algorithm, OpenMP over envs/samples like the reference's
    @threads) timed on this box's host cores on a bounded sample of the same workload.  kind = "port": the real
    reference is Julia and cannot run here."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    capi = pkg._capi
    E, T, EP = 65536, 96, 2
    cfg = capi.default_config(capi.ENV_CARTPOLE)
    cfg.n_envs, cfg.n_steps, cfg.episode_len, cfg.fixed_length_episodes = E, T, 500, 1
    cfg.batch_size, cfg.epochs, cfg.seed = E * T // 2, EP, seed
    o = oracle_lib.Oracle(cfg)
    layer = pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space())
    o.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(seed))))
    o.env_reset(seed)
    t0 = time.perf_counter(); o.collect_rollout(); t1 = time.perf_counter()
    o.ppo_update(); t2 = time.perf_counter()
    per_step_roll = (t1 - t0) / (E * T)
    per_sample_epoch = (t2 - t1) / (E * T * EP)
    epochs = 10
    value = 1.0 / (per_step_roll + epochs * per_sample_epoch)
    return {"value": value, "unit": "env-steps/s", "cores": int(oracle_lib.lib().orc_num_threads()), "kind": "port",
            "sample": f"C/OpenMP oracle: rollout of {T} steps x {E} envs ({t1 - t0:.2f}s) + {EP} PPO epochs over those {E * T} samples in 2 minibatches each "
                      f"({t2 - t1:.2f}s); extrapolated to 1 rollout step + {epochs} epochs per env-step",
            "rollout_only": 1.0 / per_step_roll}


def sac_flops(D: int, A: int, H: int, B: int, E: int) -> tuple:
    """ALGORITHMIC flops (2 x MACs of the dense layers; head math excluded) of one gradient step and one collection step.
    update! (sac.jl:299-404): actor forward on obs and next obs (2B), target-Q forward (2 nets), Q forward + backward (3x forward, 2 nets),
    Q forward + input-gradient backward on the actor's actions (2x forward, 2 nets), actor backward (2x forward)."""
    fa = 2 * (D * H + H * H + H * A)            # actor forward / sample
    fq = 2 * ((D + A) * H + H * H + H)          # one Q net forward / sample
    upd = B * (2 * fa + 2 * fq + 2 * 3 * fq + 2 * 2 * fq + 2 * fa)
    return upd, E * fa


def run_sac(pkg, *, steps: int, warmup: int, iters: int, E: int = 4096, H: int = 512, events: bool = True, cpu: bool = True) -> dict:
    """BASELINE.json configs[4]: SAC on Pendulum-v1, n_envs = 4096, SACLayer [512,512] relu, SAC() defaults (batch 256, train_freq 1,
    gradient_steps 1).  One bench "step" = `iters` iterations of train!'s loop body (sac.jl:464-535): collect one env step over all
    envs into the device replay ring, then one update!.  Single GPU (the path has one learner; N > 1 would be replicas)."""
    B = 256
    env = pkg.PendulumEnv(max_steps=200)
    alg = pkg.SAC(batch_size=B, buffer_capacity=1_000_000)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(H, H))
    cfg = pkg.make_sac_config(env, E, alg, layer, seed=42, profile_events=events)
    h = pkg.SacHandle(cfg)
    flat = pkg.sac_flatten_params(layer.initialparameters(np.random.default_rng(42)))
    h.set_params(flat); h.env_reset(42)
    h.collect_rollout(max(1, alg.start_steps // E), True)          # train!'s first, random-action collection (sac.jl:436-440)

    # the loop body of train! as dril_sac_train runs it after its first iteration (dril_sac_iterate): {collect train_freq env steps, the gradient steps} enqueued
    # back to back, the stream drained every 64 iterations; statistics of every gradient step and fps of every iteration come back with each call
    for _ in range(warmup):
        h.iterate(iters)
    h.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.iterate(iters)
    dt = time.perf_counter() - t0
    prof = h.profile()
    n_it = steps * iters
    f_upd, f_col = sac_flops(3, 1, H, B, E)
    out = {"metric": "env-steps/s (SAC collect + update!) at n_envs=4096", "value": E * n_it / dt, "unit": "env-steps/s", "n_gpus": 1,
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 (collection contraction: f16 two-piece split, f32 accumulate)", "data": "synthetic",
           "config": {"workload": f"Pendulum-v1 configs[4]: SAC, device-resident envs + replay ring, SACLayer hidden_dims=[{H},{H}] relu, batch {B}, "
                                  f"train_freq 1, gradient_steps 1; one step = {iters} iterations of (1 env step x {E} envs, 1 update!)",
                      "n_envs": E, "batch_size": B, "iterations_per_step": iters, "buffer_capacity": alg.buffer_capacity}}
    if prof["updates"]:
        u_ms, c_ms = prof["update_ms"] / prof["updates"], prof["collect_ms"] / max(1, prof["collect_steps"])
        ach = (f_upd + f_col) / ((u_ms + c_ms) * 1e-3) / 1e12
        # priced against the f32-MFMA peak: update!'s sixteen launches run v_mfma_f32_32x32x2_f32; only the collection's 512 x 4096 x 512 contraction runs on two-piece f16 operands
        out["roofline"] = dict(mfma_roofline(ach, "f32 (update!: v_mfma_f32_32x32x2_f32; the collection's second layer: two f16 pieces per operand, three v_mfma_f32_32x32x16_f16 per k16 step, f32 accumulate)"), traffic=None,
                               kernel="all launches of one SAC iteration (sac_gemm_lds_kernel / sac_gemm_multi_kernel / head kernels of update!; sac_collect_l1 / l2 / env kernels of the collection step); in-kernel wall-clock stamps at the phase boundaries",
                               update_ms=u_ms, collect_step_ms=c_ms, flops_per_update=f_upd, flops_per_collect_step=f_col)
    if cpu:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_lib
        ccfg = pkg.make_sac_config(env, E, pkg.SAC(batch_size=B, buffer_capacity=200_000), layer, seed=42)
        o = oracle_lib.sac_oracle(ccfg)
        o.set_params(flat); o.env_reset(42); o.collect_rollout(1, True)
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            o.collect_rollout(1, False); o.update(1)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": E * n / dtc, "unit": "env-steps/s", "cores": int(oracle_lib.lib().orc_num_threads()), "kind": "port",
                               "sample": f"C/OpenMP oracle: {n} iterations of (1 env step x {E} envs, 1 update! at batch {B}) in {dtc:.2f}s"}
    h.close()
    return out


def run_ppo(pkg, *, env_name: str, E: int, T: int, hidden: int, minibatches: int, epochs: int, normalize: bool, steps: int, warmup: int, events: int = 7,
            batch_size: int | None = None, fixed_length: bool = True, label: str = "", rank: int = 0, local_rank: int = 0, world: int = 1, dist=None,
            grad_variant: str | None = None) -> dict | None:
    """one PPO workload: `warmup` untimed iterations, then exactly `steps` iterations (rollout + GAE + epochs x minibatches update) between a barrier +
    stream synchronisation on both sides, max over ranks; returns the result dict on rank 0."""
    N_local = E * T
    B_global = batch_size if batch_size else (N_local // minibatches) * world
    env = pkg.CartPoleEnv(max_steps=500) if env_name == "cartpole" else pkg.PendulumEnv(max_steps=200)
    alg = pkg.PPO(n_steps=T, batch_size=B_global, epochs=epochs)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(hidden, hidden))
    cfg = pkg.make_config(env, E, alg, layer, seed=42, fixed_length_episodes=fixed_length, device=local_rank, rank=rank, world_size=world,
                          profile_events=events, normalize={} if normalize else None)
    old_gv = os.environ.get("DRIL_GRAD_VARIANT")
    if grad_variant is not None:
        os.environ["DRIL_GRAD_VARIANT"] = grad_variant      # latched by dril_create: "0" = the exact-f32 kernels (v_mfma_f32_32x32x2_f32) for update AND forward
    try:
        h = pkg.Handle(cfg)
    finally:
        if grad_variant is not None:
            os.environ.pop("DRIL_GRAD_VARIANT") if old_gv is None else os.environ.__setitem__("DRIL_GRAD_VARIANT", old_gv)
    h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(42))))   # random-init weights of the named architecture
    force_ar = os.environ.get("DRIL_FORCE_ALLREDUCE", "0") not in ("", "0")      # one rank through RCCL (tests): the same all-reduce call sites, a 1-rank communicator
    if world > 1 or force_ar:
        # banner BEFORE the communicator: if ncclCommInitRank hangs or fails, the record already says which device every rank had bound
        rank_banner(rank, world, local_rank, h.device_info())
        uid = [h.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        h.comm_init(uid[0])
        progress(rank, f"RCCL communicator up: ncclCommCount = {h.comm_ranks()} ranks")
        if h.comm_ranks() != world:        # what the communicator itself counts (ncclCommCount), not what the launcher asked for
            raise SystemExit(f"rank {rank}: the RCCL communicator has {h.comm_ranks()} ranks, --gpus asked for {world}")
    h.env_reset(42)

    def iteration():
        h.set_learning_rate(alg.learning_rate)
        h.lib.dril_collect_rollout(h._h, None)      # no fps query: keeps the iteration free of host syncs until the update's end
        return h.ppo_update()

    for i in range(warmup):
        iteration()
        if world > 1: progress(rank, f"warm-up iteration {i + 1}/{warmup} done")
    h.synchronize(); h.profile_reset()
    calls0 = h.comm_allreduce_calls()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = iteration()
        if world > 1: progress(rank, f"iteration {i + 1}/{steps} done")      # (stderr, after the iteration's own end-of-update sync: no extra device sync in the timed region)
    h.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        import torch
        t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    prof = h.profile()
    if world > 1:
        # a data-parallel result must have BEEN data-parallel: one gradient all-reduce per optimiser step of the timed region at least (+ the per-epoch moment tables);
        # a KL stop (target_kl, off in PPO() defaults) would legitimately lower it, the bench workloads have none
        need = epochs * (-(-N_local * world // B_global)) * steps
        got = h.comm_allreduce_calls() - calls0
        if got < need:
            raise SystemExit(f"rank {rank}: {got} all-reduces in the timed region, expected at least {need} (epochs x minibatches x steps): not a data-parallel run")
    out = None
    if rank == 0:
        total_env_steps = N_local * world * steps
        value = total_env_steps / dt
        out = {
            "metric": f"env-steps/s (rollout+PPO update) at n_envs={E}", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (label or (("CartPole-v1 configs[1]" if env_name == "cartpole" else "Pendulum-v1 configs[2]" + (" + NormalizeWrapperEnv" if normalize else ""))))
                                   + f": device-resident envs, ActorCritic hidden_dims=[{hidden},{hidden}], PPO rollout + update",
                       "n_envs_per_gpu": E, "n_steps": T, "epochs": epochs, "batch_size": B_global,
                       "optimizer_steps_per_iteration": epochs * (-(-N_local * world // B_global)),
                       "episodes": f"fixed length {env.max_steps} (termination disabled)" if fixed_length else f"real episodes (time limit {env.max_steps})",
                       "parallelism": f"dp{world} (env shards)"},
            "loss_last": last.loss, "n_updates_last": last.n_updates,
            # what the communicator itself reports (ncclCommCount), not the launcher's WORLD_SIZE; 1 = no communicator (single GPU)
            "rccl_ranks": h.comm_ranks(), "allreduce_calls": h.comm_allreduce_calls(),
            # updates that left the f16-piece arithmetic (dril_f32_fallback_info): retries = redone on the exact-f32 kernels, direct = run on them at once; 0 / 0 for this workload
            "f32_retries": h.f32_retries(), "f32_fallback": h.f32_fallback_info(),
            "value_per_gpu": value / world,      # N = 1-equivalent figure, to be read against the N = 1 BENCH line
        }
        gk = prof.get("ppo_grad_kernel", {"total_ms": 0, "launches": 0})
        if gk["launches"]:
            # dominant kernel: the gradient kernel. ALGORITHMIC flops per launch = B_local samples x 3 x forward flops (fwd + bwd of both MLPs,
            # 53 376 at [64,64]); duration = HIP events on the library's stream around each launch in the timed region.
            avg_ms = gk["total_ms"] / gk["launches"]
            info = h.grad_kernel_info()                  # "<kernel>: <arithmetic>" of the kernel the last optimiser step actually ran
            flops = (B_global // world) * 3 * flop_fwd(h.D, hidden, h.A)
            if info.startswith("ppo_update_small_kernel"):   # one launch = a run of optimiser steps (all of an iteration's, up to 16 384): every sample of the buffer, `epochs` times
                flops = N_local * epochs * 3 * flop_fwd(h.D, hidden, h.A) * steps // gk["launches"]
            ach = flops / (avg_ms * 1e-3) / 1e12
            # traffic and the rocprofv3 average are NOT measured in this run (PMC needs its own rocprofv3 passes): they are replayed from the
            # committed profile of exactly this workload, and the line says so (traffic_source: file + the commit the profile was taken at)
            traffic = traffic_source = rocprof_ms = None
            pmc_name = PMC_FILES.get((env_name, hidden))
            pmc = ROOT / "profiles" / pmc_name if pmc_name else None
            if pmc and pmc.exists() and minibatches == 32 and E == 65536 and T == 2048 and not batch_size and grad_variant is None and world == 1:
                rec = json.loads(pmc.read_text())
                traffic, rocprof_ms = rec.get("hbm_bytes_per_launch"), rec.get("rocprof_avg_launch_ms")
                traffic_source = f"replayed from profiles/{pmc_name} (rocprofv3 --pmc passes at commit {rec.get('commit', '?')}); not measured in this run"
            kname, arith = info.split(": ", 1)
            out["dtype"] = "f32 (f16x2 split, f32 accumulate)" if "f16x2" in arith else "f32 (bf16x3 split, f32 accumulate)" if "bf16x3" in arith else "f32"
            out["roofline"] = dict(mfma_roofline(ach, arith), traffic=traffic, traffic_source=traffic_source, kernel=kname, avg_launch_ms=avg_ms,
                                   avg_launch_ms_source=f"HIP events on the library's stream around every {'' if events == 1 else str(int(events)) + 'th '}launch of the timed region"
                                                        + ("" if events == 1 else f" (launch i of the class is bracketed when i % {int(events)} == 0, counted from the profile reset)"),
                                   timed_launches=gk["timed_launches"],
                                   rocprof_avg_launch_ms=rocprof_ms, launches=gk["launches"], flops_per_launch=flops,
                                   record_bytes_per_launch=(B_global // world) * 64)      # one 32-byte record per sample and net (the algorithmic gather volume of the record path)
            out["kernel_ms_per_step"] = {k: v["total_ms"] / steps for k, v in prof.items() if v["launches"]}
            out["hbm_kernels"] = hbm_kernels(prof, N=N_local, D=h.D, A=h.A, discrete=bool(h.discrete), P=h.P, epochs=epochs, steps=steps, normalize=normalize)
            ar = prof.get("ncclAllReduce", {"total_ms": 0, "launches": 0})
            if ar["launches"]:
                # HIP events around the library's ncclAllReduce launches (the same thinning as the other per-step classes): time the stream spent in all-reduces per
                # optimiser step — the gradient all-reduce of [P + 8] floats plus, amortised, the per-epoch moment table and the explained-variance sums
                opt_steps = epochs * (-(-N_local * world // B_global)) * steps
                out["allreduce_us_per_step"] = 1e3 * ar["total_ms"] / opt_steps
                out["allreduce"] = {"launches": ar["launches"], "timed_launches": ar["timed_launches"], "avg_us_per_launch": 1e3 * ar["total_ms"] / ar["launches"],
                                    "gradient_bytes": 4 * (h.P + 8), "ranks": h.comm_ranks()}
            rk = prof.get("rollout_kernel", {"total_ms": 0, "launches": 0})
            if rk["launches"]:                           # what the reference logs as env/fps (rollout_buffer.jl:60-64, ppo.jl:176): env steps per second of the collection alone (HIP events around the rollout)
                out["rollout_only_env_steps_per_s"] = N_local * world * steps / (rk["total_ms"] * 1e-3)
    h.close()
    return out


def secondary_runs(pkg) -> list:
    """short, timed runs of the other single-GPU configs of BASELINE.json beside the headline line (a driver-observed number for each):
    configs[1] on the exact-f32 kernels (1 warm-up + 2 iterations), configs[2] Pendulum [256,256] + NormalizeWrapperEnv at full size (1 warm-up + 2 iterations), configs[4] SAC (1 + 2 steps of 500 iterations),
    configs[0] the reference's README quick-start (4 envs, PPO() defaults, real CartPole episodes).  Each entry has its own config / roofline."""
    out = []
    # configs[1] again on the exact-f32 kernels (DRIL_GRAD_VARIANT=0: v_mfma_f32_32x32x2_f32 for update and forward) — the literal reading of "all arithmetic is fp32"
    # (SURVEY.md section 8 preamble); its roofline.frac is against the 157.3 TFLOP/s f32-MFMA peak
    out.append(run_ppo(pkg, env_name="cartpole", E=65536, T=2048, hidden=64, minibatches=32, epochs=10, normalize=False, steps=2, warmup=1, grad_variant="0",
                       label="CartPole-v1 configs[1] on the exact-f32 kernels (DRIL_GRAD_VARIANT=0)"))
    out.append(run_ppo(pkg, env_name="pendulum", E=65536, T=2048, hidden=256, minibatches=32, epochs=10, normalize=True, steps=2, warmup=1))
    out.append(run_sac(pkg, steps=2, warmup=1, iters=500, cpu=False))
    c0 = run_ppo(pkg, env_name="cartpole", E=4, T=2048, hidden=64, minibatches=0, epochs=10, normalize=False, steps=5, warmup=2, batch_size=64, fixed_length=False, events=1,
                 label="CartPole-v1 configs[0] (README quick-start: MultiThreadedParallelEnv n_envs=4, PPO() defaults, batch_size 64)")
    c0["cpu_baseline"] = cpu_baseline_configs0(pkg, 42)      # the one shape where a CPU is close: the reference's own README case, the same iteration on the host cores
    out.append(c0)
    return out


def cpu_baseline_configs0(pkg, seed: int) -> dict:
    """configs[0] on the CPU oracle: one full iteration of the README quick-start (4 envs x 2048 steps, real CartPole episodes, batch 64, 10 epochs = 1 280 optimiser
    steps), timed whole — no extrapolation.  kind "port" (C restatement; the Julia reference cannot run here)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib
    capi = pkg._capi
    cfg = capi.default_config(capi.ENV_CARTPOLE)
    cfg.n_envs, cfg.n_steps, cfg.episode_len, cfg.fixed_length_episodes, cfg.batch_size, cfg.epochs, cfg.seed = 4, 2048, 500, 0, 64, 10, seed
    o = oracle_lib.Oracle(cfg)
    layer = pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space())
    o.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(seed))))
    o.env_reset(seed)
    o.collect_rollout(); o.ppo_update()          # warm-up iteration (page faults, OpenMP team start)
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        o.collect_rollout(); o.ppo_update()
    dt = time.perf_counter() - t0
    return {"value": 4 * 2048 * n / dt, "unit": "env-steps/s", "cores": int(oracle_lib.lib().orc_num_threads()), "kind": "port",
            "sample": f"C/OpenMP oracle: {n} full iterations of configs[0] (4 envs x 2048 steps + 10 epochs x 128 minibatches of 64) in {dt:.3f}s after one warm-up iteration"}


def compact_entry(e: dict) -> dict:
    """a secondary run as it appears inside the headline line: workload, value, its roofline numbers, its CPU figure — the full entry goes to stderr"""
    r = e.get("roofline") or {}
    c = {"workload": e["config"]["workload"].split(":")[0], "value": e["value"], "unit": e["unit"], "ms_per_step": e["ms_per_step"], "steps": e["steps"], "dtype": e.get("dtype")}
    if r:
        c["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "update_ms", "collect_step_ms") if k in r}
        c["roofline"]["kernel"] = str(c["roofline"].get("kernel", ""))[:60]
    if "cpu_baseline" in e:
        c["cpu_baseline"] = {k: e["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")}
    for k in ("f32_retries", "allreduce_us_per_step"):
        if k in e: c[k] = e[k]
    return c


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: `python bench.py --gpus N` starts its own ranks.  The parent makes NO GPU call, imports neither the package nor torch and
# loads no shared library: it starts N fresh children of this script (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and HSA_ENABLE_IPC_MODE_LEGACY=0 in their
# environment, each in its own session), relays their stderr, watches them (exit code, a total time limit, a silence limit per rank — every rank prints a progress line per
# phase and per iteration), on any failure ends exactly the process groups it started and exits non-zero, and on success prints rank 0's JSON line after checking that
# the line really is an N-rank result.  Under torch.distributed.run (WORLD_SIZE already set) none of this runs.
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def _end_groups(procs) -> None:
    import signal
    for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
        live = [p for p in procs if p.poll() is None]
        if not live:
            return
        for p in live:
            try:
                os.killpg(p.pid, sig)          # the child's own session (start_new_session): exactly the group this launcher created
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.monotonic() + grace
        while time.monotonic() < t_end and any(p.poll() is None for p in live):
            time.sleep(0.05)


def launch_ranks(n: int, argv: list, total_timeout: float, silent_timeout: float) -> int:
    import subprocess
    import threading
    port = _free_port()
    procs, out_lines, last_seen, threads = [], [[] for _ in range(n)], [time.monotonic()] * n, []

    def pump(r: int, stream, is_err: bool) -> None:
        for line in stream:
            last_seen[r] = time.monotonic()
            if is_err:
                sys.stderr.write(f"[rank {r}] {line}"); sys.stderr.flush()
            else:
                out_lines[r].append(line)

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONUNBUFFERED="1")
        env.setdefault("NCCL_DEBUG", "WARN")      # a failing ncclCommInitRank / first collective must say why on stderr (which rank, device, transport); the caller's own setting wins
        p = subprocess.Popen([sys.executable, "-u", str(Path(__file__).resolve())] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             start_new_session=True, cwd=str(ROOT))
        procs.append(p)
        for stream, is_err in ((p.stdout, False), (p.stderr, True)):
            t = threading.Thread(target=pump, args=(r, stream, is_err), daemon=True); t.start(); threads.append(t)
    t0 = time.monotonic()
    why = None
    while why is None:
        time.sleep(0.1)
        now = time.monotonic()
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            why = "rank %d exited with code %d" % bad[0]
        elif all(c == 0 for c in codes):
            break
        elif now - t0 > total_timeout:
            why = f"the ranks did not finish within {total_timeout:.0f} s"
        else:
            quiet = [r for r, c in enumerate(codes) if c is None and now - last_seen[r] > silent_timeout]
            if quiet:
                why = f"rank {quiet[0]} printed nothing for {silent_timeout:.0f} s"
    if why is not None:
        _end_groups(procs)
        sys.stderr.write(f"bench.py --gpus {n}: {why}; all ranks ended, no result line\n")
        return 1
    for t in threads:
        t.join(timeout=5.0)
    line = None
    for cand in reversed(out_lines[0]):
        try:
            rec = json.loads(cand)
        except ValueError:
            continue
        if isinstance(rec, dict) and "metric" in rec:
            line = rec; break
    if line is None:
        sys.stderr.write(f"bench.py --gpus {n}: rank 0 exited 0 without a JSON result line\n")
        return 1
    if line.get("n_gpus") != n or line.get("rccl_ranks") != n:
        sys.stderr.write(f"bench.py --gpus {n}: rank 0 reports n_gpus={line.get('n_gpus')} rccl_ranks={line.get('rccl_ranks')}: not an {n}-rank result\n")
        return 1
    line["launcher"] = f"bench.py started its own {n} ranks (no WORLD_SIZE in the environment)"
    print(json.dumps(line), flush=True)
    return 0


def progress(rank: int, msg: str) -> None:
    """one line on stderr per phase / iteration: the launcher's silence watchdog reads these (stdout carries only the JSON line)"""
    sys.stderr.write(f"[bench rank {rank} +{time.perf_counter() - _T_START:7.1f}s] {msg}\n"); sys.stderr.flush()


_T_START = time.perf_counter()


def rank_banner(rank: int, world: int, local_rank: int, device_info: str) -> None:
    """what every rank of a data-parallel job says BEFORE it enters ncclCommInitRank: which device it bound (ordinal, PCI bus id, visibility masks) and the two environment
    settings a failing first contact is usually about — so that a hang or failure inside RCCL leaves a record that names the rank -> device map"""
    progress(rank, f"BANNER rank {rank}/{world} local_rank {local_rank} -> {device_info}; NCCL_DEBUG={os.environ.get('NCCL_DEBUG', '(unset)')} "
                   f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '(unset)')}")


def stub_worker(args) -> int:
    """DRIL_BENCH_STUB=1 (tests/test_bench_launcher.py): a rank that loads NO library and touches no GPU — it proves the launcher's plumbing (environment of every
    rank, a real gloo rendezvous on MASTER_ADDR:MASTER_PORT, barrier + max over ranks, a failing rank, a silent rank, a rank-count mismatch)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")
    sys.stderr.write("STUBENV " + json.dumps({k: os.environ.get(k) for k in keys}) + "\n"); sys.stderr.flush()
    if world > 1:
        rank_banner(rank, world, int(os.environ.get("LOCAL_RANK", "0")), "(stub worker: no library loaded, no device bound)")
    if os.environ.get("DRIL_BENCH_STUB_FAIL_RANK") == str(rank):
        progress(rank, "stub: failing on purpose"); return 7
    if os.environ.get("DRIL_BENCH_STUB_SILENT_RANK") == str(rank):
        time.sleep(3600); return 0
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = 1 if os.environ.get("DRIL_BENCH_STUB_WRONG_RANKS") else world
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": 0.0, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) * 1e3,
                          "rccl_ranks": ranks, "stub": True}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    return 0


def main() -> None:
    # RCCL across processes needs dmabuf IPC on this platform (legacy IPC: hipIpcGetMemHandle "invalid argument").  ROCr reads the variable when it initialises, so it is
    # exported FIRST — before the package import, before libdril_hip.so is loaded (fat-binary registration), before any HIP call of this process or of its children
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", choices=["ppo", "sac"], default="ppo", help="sac = BASELINE configs[4] (single GPU)")
    ap.add_argument("--sac-iters", type=int, default=500, help="train! iterations per bench step in --algo sac")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-envs", type=int, default=65536)
    ap.add_argument("--n-steps", type=int, default=2048)
    ap.add_argument("--minibatches", type=int, default=32, help="optimiser steps per epoch; batch_size = N / this")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--env", choices=["cartpole", "pendulum"], default="cartpole", help="pendulum + --hidden 256 + --normalize = BASELINE configs[2]")
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--normalize", action="store_true", help="NormalizeWrapperEnv on device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of configs[2] / configs[4] / configs[0] appended to the default line")
    ap.add_argument("--no-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--full", action="store_true", help="the secondary runs and the per-kernel bandwidth table in full inside the JSON line (default: compact there, full on stderr)")
    ap.add_argument("--event-stride", type=int, default=7, help="bracket every K-th launch of the per-optimiser-step kernels with HIP events (1 = every launch; "
                    "an event record costs the stream ~3.5 us, 960 launches an iteration: 2 %% of configs[1] at K = 1).  7 is coprime to the 32 minibatches of an epoch: the "
                    "sampled launches visit every position of the epoch in turn (a stride of 8 always sampled positions 0, 8, 16, 24 — position 0 follows the epoch's index / moments kernels)")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="--gpus N > 1 without WORLD_SIZE: seconds the self-started ranks may take in total")
    ap.add_argument("--silent-timeout", type=float, default=420.0, help="... and seconds one rank may stay without a progress line (the first import of a fresh box takes 1-2 min)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU (no package import, no library load above this line)
        if args.algo == "sac":
            raise SystemExit("--algo sac is a single-learner path: --gpus 1 only")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout, args.silent_timeout))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    if os.environ.get("DRIL_BENCH_STUB"):
        sys.exit(stub_worker(args))
    progress(int(os.environ.get("RANK", "0")), "loading libdril_hip.so")
    pkg = g.load_package()           # loads libdril_hip.so first (binds /opt/rocm's HIP runtime); no CPU fallback exists
    pkg._capi.load_library()
    if args.algo == "sac":
        if args.gpus != 1:
            raise SystemExit("--algo sac is a single-learner path: --gpus 1 only")
        print(json.dumps(run_sac(pkg, steps=args.steps, warmup=args.warmup, iters=args.sac_iters, E=args.n_envs if args.n_envs != 65536 else 4096,
                                 H=args.hidden if args.hidden != 64 else 512, events=not args.no_events, cpu=not args.no_cpu_baseline)), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist  # rendezvous / barrier only; never touches torch.cuda
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        progress(rank, f"gloo rendezvous of {world} ranks done")

    out = run_ppo(pkg, env_name=args.env, E=args.n_envs, T=args.n_steps, hidden=args.hidden, minibatches=args.minibatches, epochs=args.epochs, normalize=args.normalize,
                  steps=args.steps, warmup=args.warmup, events=0 if args.no_events else max(1, args.event_stride), rank=rank, local_rank=local_rank, world=world, dist=dist)
    if rank == 0:
        default_workload = (args.env, args.n_envs, args.n_steps, args.hidden, args.minibatches, args.epochs, args.normalize) == ("cartpole", 65536, 2048, 64, 32, 10, False)
        out["metric"] = "env-steps/s (rollout+PPO update) at n_envs=65536" if args.n_envs == 65536 else out["metric"]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, 42)
        if world == 1 and default_workload and not args.no_secondary:
            full = secondary_runs(pkg)
            for e in full:
                sys.stderr.write("[bench secondary, full entry] " + json.dumps(e) + "\n")
            sys.stderr.flush()
            out["secondary"] = full if args.full else [compact_entry(e) for e in full]
        if not args.full:
            for row in out.get("hbm_kernels", []):
                row.pop("what", None); row.pop("peak_GBps", None)      # the definitions are in docs/measurement.md; the line stays under the driver's 8 KB tail
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

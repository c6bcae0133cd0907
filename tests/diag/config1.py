"""BASELINE configs[0] (the reference's README quick-start: CartPole, n_envs = 4, PPO() defaults: n_steps 2048, batch 64, 10 epochs) on the device path and
on the CPU oracle — a data point for DESIGN.md (this size is launch-latency-bound on any GPU)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g
pkg = g.load_package()
import oracle_lib
env = pkg.CartPoleEnv(max_steps=500); alg = pkg.PPO()
layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
cfg = pkg.make_config(env, 4, alg, layer, seed=42)
flat = pkg.flatten_params(layer.initialparameters(np.random.default_rng(0)))
for name, h in (("device", pkg.Handle(cfg)), ("oracle", oracle_lib.Oracle(cfg))):
    h.set_params(flat); h.env_reset(42)
    h.collect_rollout(); h.ppo_update()
    t0 = time.perf_counter(); n = 3
    tr = tu = 0.0
    for _ in range(n):
        a = time.perf_counter(); h.collect_rollout(); b = time.perf_counter(); h.ppo_update(); c = time.perf_counter()
        tr += b - a; tu += c - b
    dt = time.perf_counter() - t0
    print(f"{name}: {4 * 2048 * n / dt:.3e} env-steps/s  (rollout {1e3 * tr / n:.1f} ms, update {1e3 * tu / n:.1f} ms per iteration of 8192 steps, 1280 optimiser steps)")

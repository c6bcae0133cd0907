#!/usr/bin/env python3
"""create / use / destroy every handle type in a loop and watch free device memory (hipMemGetInfo): leaks show up as a drift"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0; return f.value / 2**20
def ppo(kind, **kw):
    c = capi.default_config(kind); c.n_envs, c.n_steps, c.batch_size, c.epochs = 64, 16, 256, 1
    for k, v in kw.items(): setattr(c, k, v)
    h = pkg.Handle(c); h.set_params((np.random.default_rng(0).standard_normal(h.P) * 0.1).astype(np.float32))
    if kind != capi.ENV_EXTERNAL:
        h.env_reset(1); h.collect_rollout(); h.ppo_update(); h.evaluate_agent(2)
    else:
        for t in range(16):
            h.ext_act(np.zeros((64, h.D), np.float32)); h.ext_record(np.zeros(64, np.float32), np.zeros(64, np.uint8), np.ones(64, np.uint8) * (t % 4 == 3), np.zeros((64, h.D), np.float32))
        h.ext_finish(np.zeros((64, h.D), np.float32)); h.ppo_update()
    h.close()
def sac():
    env = pkg.PendulumEnv(); alg = pkg.SAC(batch_size=64, buffer_capacity=4096, start_steps=64)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(64, 64))
    h = pkg.SacHandle(pkg.make_sac_config(env, 16, alg, layer)); h.env_reset(0); h.train(400); h.close()
base = None
for rep in range(6):
    for _ in range(10):
        ppo(capi.ENV_CARTPOLE); ppo(capi.ENV_PENDULUM, hidden1=256, hidden2=256, norm_obs=1, norm_reward=1, norm_training=1, monitor_window=10)
        ppo(capi.ENV_EXTERNAL, ext_obs_dim=20, ext_action_dim=5, ext_discrete=0, hidden1=96, hidden2=48); sac()
    f = free_mb(); base = base or f
    print(f"after {10 * (rep + 1):3d} rounds of 4 handles: free {f:10.1f} MiB (drift {f - base:+.1f})")
assert abs(free_mb() - base) < 64, "device memory drifts"
print("no leak")

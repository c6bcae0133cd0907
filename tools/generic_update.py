#!/usr/bin/env python3
"""Cost of the generic (DRIL_ENV_EXTERNAL) update path next to the fused kernels on the SAME buffer: CartPole's spaces, hidden [64,64],
N = n_envs * n_steps samples, 32 minibatches x 10 epochs.  Also times a few other shapes the fused kernels do not cover.
usage: python tools/generic_update.py [n_envs=4096] [n_steps=256]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package(); capi = pkg._capi
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = E * T


def ext_cfg(D, A, disc, H1, H2):
    c = capi.default_config(capi.ENV_EXTERNAL)
    c.ext_obs_dim, c.ext_action_dim, c.ext_discrete, c.hidden1, c.hidden2 = D, A, int(disc), H1, H2
    c.n_envs, c.n_steps, c.batch_size, c.epochs, c.profile_events = E, T, N // 32, 10, 1
    return c


def fill(h, seed=0):
    rng = np.random.default_rng(seed)
    h.set_params((rng.standard_normal(h.P) * 0.2).astype(np.float32))
    obs = rng.uniform(-1, 1, (N, h.D)).astype(np.float32)
    act = (rng.integers(0, h.A, N) + h.cfg.action_start).astype(np.int32) if h.discrete else rng.standard_normal((N, h.A)).astype(np.float32)
    h.set_buffer(capi.BUF_OBSERVATIONS, obs); h.set_buffer(capi.BUF_ACTIONS, act)
    for which in (capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_VALUES):
        h.set_buffer(which, rng.standard_normal(N).astype(np.float32))
    B = 1 << 18
    lp = np.concatenate([h.evaluate_actions(obs[i:i + B], act[i:i + B])[1] for i in range(0, N, B)])
    h.set_buffer(capi.BUF_LOGPROBS, lp + rng.normal(0, 0.05, N).astype(np.float32))


def flops(D, A, H1, H2):   # forward + backward = 3 x forward, both nets
    f = lambda o: 2 * (D * H1 + H1 * H2 + H2 * o)
    return 3 * (f(A) + f(1))


def timed(h, label, fl):
    fill(h)
    h.ppo_update(); h.profile_reset()
    t0 = time.perf_counter(); st = h.ppo_update(); dt = time.perf_counter() - t0
    pr = h.profile()
    gk = pr.get("ppo_grad_kernel", {"total_ms": 0, "launches": 1})
    tf = fl * N * 10 / dt / 1e12
    print(f"{label:58s} update {dt * 1e3:8.1f} ms  {N * 10 / dt / 1e6:8.1f} M samples/s  {tf:6.1f} TFLOP/s whole update; grad stage {gk['total_ms'] / max(gk['launches'], 1):7.3f} ms/minibatch; n_updates {st.n_updates}")
    return dt


import os
if os.environ.get("GENERIC_SHAPE"):   # "D,A,discrete,H1,H2": only this shape (for rocprofv3 runs)
    D, A, disc, H1, H2 = (int(x) for x in os.environ["GENERIC_SHAPE"].split(","))
    timed(pkg.Handle(ext_cfg(D, A, bool(disc), H1, H2)), f"generic obs [{D}] {'Discrete' if disc else 'Box'}({A}) hidden [{H1},{H2}]", flops(D, A, H1, H2))
    sys.exit(0)
cf = capi.default_config(capi.ENV_CARTPOLE)
cf.n_envs, cf.n_steps, cf.batch_size, cf.epochs, cf.profile_events = E, T, N // 32, 10, 1
a = timed(pkg.Handle(cf), "fused   CartPole [4] Discrete(2) hidden [64,64]", flops(4, 2, 64, 64))
b = timed(pkg.Handle(ext_cfg(4, 2, True, 64, 64)), "generic same spaces (DRIL_ENV_EXTERNAL)", flops(4, 2, 64, 64))
print(f"generic / fused = {b / a:.2f}x")
import os
for (D, A, disc, H1, H2) in () if os.environ.get('GENERIC_FIRST_ONLY') else ((17, 6, False, 64, 64), (24, 4, False, 256, 256), (64, 18, True, 512, 512), (376, 17, False, 400, 300)):
    timed(pkg.Handle(ext_cfg(D, A, disc, H1, H2)), f"generic obs [{D}] {'Discrete' if disc else 'Box'}({A}) hidden [{H1},{H2}]", flops(D, A, H1, H2))

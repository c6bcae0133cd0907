#!/usr/bin/env python3
"""Host-env boundary cost (DRIL_ENV_EXTERNAL): wall time of one dril_ext_act + dril_ext_record pair (host obs in over PCIe, policy forward +
sampling on the device, actions back, rewards / flags in) and of one optimiser step, for a few env counts.  No env is stepped: this is the
library's share of an env step.   usage: python tools/ext_latency.py"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package(); capi = pkg._capi
D, A, H = 24, 4, 64
for E in (8, 64, 1024, 16384):
    T = 64
    c = capi.default_config(capi.ENV_EXTERNAL)
    c.ext_obs_dim, c.ext_action_dim, c.ext_discrete, c.hidden1, c.hidden2 = D, A, 0, H, H
    c.ext_action_low, c.ext_action_high = -1.0, 1.0
    c.n_envs, c.n_steps, c.batch_size, c.epochs = E, T, max(64, E * T // 32), 2
    h = pkg.Handle(c)
    rng = np.random.default_rng(0)
    h.set_params((rng.standard_normal(h.P) * 0.2).astype(np.float32))
    obs = rng.standard_normal((E, D)).astype(np.float32); rew = np.zeros(E, np.float32); fl = np.zeros(E, np.uint8)
    tr = fl.copy(); tr[::7] = 1
    for rep in range(2):
        t_act = t_rec = t_rec_tr = 0.0
        for t in range(T):
            a = time.perf_counter(); h.ext_act(obs); b = time.perf_counter()
            if t % 8 == 7:
                h.ext_record(rew, fl, tr, obs); t_rec_tr += time.perf_counter() - b
            else:
                h.ext_record(rew, fl, fl); t_rec += time.perf_counter() - b
            t_act += b - a
        a = time.perf_counter(); h.ext_finish(obs); t_fin = time.perf_counter() - a
        a = time.perf_counter(); st = h.ppo_update(); t_upd = time.perf_counter() - a
    print(f"E = {E:6d}: ext_act {t_act / T * 1e6:7.1f} us  ext_record {t_rec / (T - T // 8) * 1e6:6.1f} us (with truncations {t_rec_tr / (T // 8) * 1e6:6.1f} us)  "
          f"= {E / ((t_act + t_rec + t_rec_tr) / T) / 1e6:7.2f} M env-steps/s boundary ceiling;  ext_finish {t_fin * 1e3:5.2f} ms;  update {t_upd * 1e3:7.2f} ms for {st.n_updates} optimiser steps "
          f"({t_upd / max(st.n_updates, 1) * 1e6:6.1f} us each)")

"""Diagnostic (GPU): where do NaNs appear in a kind-2 rollout (device vs oracle)?  Prints the first offending buffer / index per repetition."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g
import oracle_lib
pkg = g.load_package(); capi = pkg._capi
names = {capi.BUF_OBSERVATIONS: "obs", capi.BUF_VALUES: "val", capi.BUF_LOGPROBS: "logp", capi.BUF_REWARDS: "rew", capi.BUF_ADVANTAGES: "adv", capi.BUF_RETURNS: "ret",
         capi.BUF_BOOTSTRAP: "boot", capi.BUF_LAST_VALUES: "last", capi.BUF_ACTIONS: "act"}
kind, E, T, L = 2, 40, 30, 12
bad = 0
for rep in range(200):
    cfg = capi.default_config(kind); cfg.n_envs, cfg.n_steps, cfg.episode_len, cfg.batch_size, cfg.epochs = E, T, L, max(2, (E * T) // 4), 2
    h, o = pkg.Handle(cfg), oracle_lib.Oracle(cfg)
    flat = (np.random.default_rng(17 + E).standard_normal(h.P) * 0.4).astype(np.float32)
    h.set_params(flat); o.set_params(flat); h.env_reset(17 + E); o.env_reset(17 + E)
    if rep % 2 == 0:
        noise = np.random.default_rng(E).standard_normal((E * T, h.A)).astype(np.float32); h.set_noise(noise); o.set_noise(noise)
    for rollout in range(2):
        h.collect_rollout(); o.collect_rollout()
        fl = h.buffer(capi.BUF_FLAGS)
        for w, nm in names.items():
            a, b = h.buffer(w), o.buffer(w)
            na, nb = np.isnan(a.astype(np.float64)).reshape(-1), np.isnan(b.astype(np.float64)).reshape(-1)
            if nm == "boot":
                na = na.reshape(-1) & ((fl & 2) != 0); nb = nb & ((o.buffer(capi.BUF_FLAGS) & 2) != 0)
            if na.any() or nb.any():
                bad += 1
                print(f"rep {rep} rollout {rollout} {nm}: device NaNs {int(na.sum())} at {np.flatnonzero(na)[:8]}, oracle NaNs {int(nb.sum())} at {np.flatnonzero(nb)[:8]}", flush=True)
    h.close()
print("done, offending buffers:", bad)

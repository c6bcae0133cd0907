#!/usr/bin/env python3
"""Writes tests/golden/reference_ppo_input.json: the inputs tests/golden/gen_reference_golden.jl feeds to DRiL.jl (flat parameters in the layout of include/dril_hip.h,
one minibatch per case) — the `_batch(seed 0)` inputs of tests/test_gpu_parity.py at B = 64 and 1 024 (4 096 would be a 2 MB fixture; 1 024 already spans 32 sample tiles).  Needs only numpy and the CPU oracle (for the old log-probabilities,
which `_batch` derives from the policy's own); deterministic: rerunning it reproduces the committed file."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g  # noqa: E402
import oracle_lib  # noqa: E402

pkg = g.load_package()
capi = pkg._capi
CASES = [  # name, env kind, B, overrides of PPO()
    ("cartpole_B64_defaults", 0, 64, {}),
    ("cartpole_B1024_ent_vfclip", 0, 1024, dict(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)),
    ("pendulum_B64_defaults", 1, 64, {}),
]
SPACES = {0: dict(obs_low=[-4.8, -3.4e38, -0.418, -3.4e38], obs_high=[4.8, 3.4e38, 0.418, 3.4e38]),      # CartPole-v1 Box, Discrete(2) starting at 1 (Julia's default)
          1: dict(obs_low=[-1, -1, -8], obs_high=[1, 1, 8], act_low=[-2.0], act_high=[2.0])}               # Pendulum-v1


def params(P, seed, scale):
    return (np.random.default_rng(seed).standard_normal(P) * scale).astype(np.float32)


def main():
    out = {"generator": "tests/golden/make_reference_input.py", "layout": "include/dril_hip.h dril_set_params: actor {W1 b1 W2 b2 W3 b3} critic {...} log_std; W (out x in) column-major",
           "cases": []}
    for name, kind, B, kw in CASES:
        cfg = capi.default_config(kind)
        cfg.n_envs, cfg.n_steps, cfg.batch_size = 2, 2, 2
        for k, v in kw.items():
            setattr(cfg, k, v)
        o = oracle_lib.Oracle(cfg)
        flat = params(o.P, 40, 0.25)
        if not o.discrete:
            flat[-o.A:] = cfg.log_std_init          # log_std as the layer constructor would leave it; the gradient test does not need a special value
        o.set_params(flat)
        rng = np.random.default_rng(0)
        obs = rng.uniform(-1, 1, (B, o.D)).astype(np.float32)
        act = (rng.integers(0, o.A, B) + cfg.action_start).astype(np.int32) if o.discrete else rng.normal(0, 1, (B, o.A)).astype(np.float32)
        adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
        _, lp, _ = o.evaluate_actions(obs, act)
        olp = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
        fl = lambda a: [float('%.9g' % x) for x in np.asarray(a, np.float32).ravel()]      # nine significant digits round-trip a float32
        case = dict(name=name, env_kind=kind, D=o.D, A=o.A, B=B, discrete=bool(o.discrete), action_start=int(cfg.action_start), hidden=[int(cfg.hidden1), int(cfg.hidden2)],
                    log_std_init=float(cfg.log_std_init), **SPACES[kind],
                    hyper=dict(clip_range=float(cfg.clip_range), clip_range_vf=(float(cfg.clip_range_vf) if cfg.has_clip_range_vf else None), ent_coef=float(cfg.ent_coef),
                               vf_coef=float(cfg.vf_coef), max_grad_norm=float(cfg.max_grad_norm), normalize_advantage=bool(cfg.normalize_advantage),
                               learning_rate=float(cfg.learning_rate), adam_eps=float(cfg.adam_eps), adam_beta1=float(cfg.adam_beta1), adam_beta2=float(cfg.adam_beta2)),
                    params=fl(flat), obs=fl(obs), actions=([int(x) for x in act] if o.discrete else fl(act)), advantages=fl(adv), returns=fl(ret), old_logprobs=fl(olp), old_values=fl(ov))
        out["cases"].append(case)
    p = ROOT / "tests" / "golden" / "reference_ppo_input.json"
    p.write_text(json.dumps(out, separators=(",", ":")))
    print("wrote", p, p.stat().st_size, "bytes")


if __name__ == "__main__":
    main()

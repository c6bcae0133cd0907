"""Error budget of the operand split against a float64 gradient (test infrastructure).

The update kernels that run on the 16-bit matrix cores (ppo_grad_pair_kernel, ppo_grad_wide_split_kernel) cut every f32 operand of the H x H
contractions into pieces — two f16 pieces and three of the four piece products since the end of round 3 (dril_device.h), three bf16 pieces and
six of nine before.  `measure` runs one minibatch through
  * the split kernel (what the size rule / the default selects, or forced),
  * the exact-f32 kernel of the same handle shape (DRIL_GRAD_VARIANT=0),
  * a float64 torch-autograd restatement of ppo.jl:365-407 (tests/test_oracle_crosschecks.py),
and reports each kernel's distance from the float64 gradient, the loss errors, and the signed statistics of the dW2 error (a split by truncation drops
one-signed terms: a bias would show as a non-zero mean / a shrinkage of the gradient).

Run as a script it prints the numbers as one JSON line; tests/test_gpu_split_arith.py starts it with DRIL_HIP_LIBRARY pointing at the
negative-control build (libdril_hip_droplo.so: the `lo` products dropped) and requires the budget to be BROKEN there.
"""
from __future__ import annotations

import contextlib
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


@contextlib.contextmanager
def grad_variant(v):
    """DRIL_GRAD_VARIANT is latched by dril_create: set it around the Handle construction only"""
    old = os.environ.get("DRIL_GRAD_VARIANT")
    if v is None:
        os.environ.pop("DRIL_GRAD_VARIANT", None)
    else:
        os.environ["DRIL_GRAD_VARIANT"] = str(v)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("DRIL_GRAD_VARIANT", None)
        else:
            os.environ["DRIL_GRAD_VARIANT"] = old


def w2_slices(D, H, A, discrete):
    """index ranges of W2 of the actor and of the critic in the flat parameter vector (include/dril_hip.h layout)"""
    out, off = [], 0
    for O in (A, 1):
        off += D * H + H
        out.append(slice(off, off + H * H))
        off += H * H + H + O * H + O
    return out


def measure(pkg, oracle_mod, kind, H, B, split_variant, seed=0, ent_coef=0.01, scale=1.0):
    """-> dict of error norms; split_variant: None = the library's own selection, 2 = force the pair kernel (hidden 64); scale: advantages, returns and old values
    are N(0, scale^2) instead of N(0, 1) (the budget must hold away from benign inputs — un-normalised rewards give returns of 1e3, sparse ones 1e-3)"""
    from test_oracle_crosschecks import make_batch, torch_ppo_loss
    capi = pkg._capi
    cfg = capi.default_config(kind)
    cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.hidden1, cfg.hidden2, cfg.ent_coef = 2, 2, 2, H, H, ent_coef
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(40 + seed).standard_normal(o.P) * (0.25 if H == 64 else 0.08)).astype(np.float32)
    o.set_params(flat)
    batch = make_batch(o, cfg, B, seed, o.discrete, o.A)
    if scale != 1.0:
        obs_, act_, adv_, ret_, lp_, ov_ = batch
        batch = (obs_, act_, (adv_ * np.float32(scale)).astype(np.float32), (ret_ * np.float32(scale)).astype(np.float32), lp_, (ov_ * np.float32(scale)).astype(np.float32))
    l64, s64, g64 = torch_ppo_loss(flat, cfg, *batch, o.discrete, o.A)
    res = {}
    for name, v in (("split", split_variant), ("f32", 0)):
        with grad_variant(v):
            h = pkg.Handle(cfg)
        h.set_params(flat)
        l, s, g = h.ppo_loss_grad(*batch)
        res[name] = (float(l), g.astype(np.float64), h.grad_kernel_info().split(":")[0])
        h.close()
    lo, so, go = o.ppo_loss_grad(*batch)
    gn = float(np.linalg.norm(g64))
    sl = w2_slices(o.D, H, o.A, o.discrete)
    idx = np.r_[sl[0], sl[1]]
    out = {"kind": kind, "H": H, "B": B, "scale": scale, "kernel_split": res["split"][2], "kernel_f32": res["f32"][2], "grad_norm": gn, "loss_f64": l64}
    for name in ("split", "f32"):
        l, g, _ = res[name]
        e = g - g64
        out[f"grad_err_{name}"] = float(np.linalg.norm(e)) / gn
        out[f"loss_err_{name}"] = abs(l - l64) / abs(l64)
        ew, gw = e[idx], g64[idx]
        out[f"dW2_mean_err_{name}"] = float(ew.mean())
        out[f"dW2_std_err_{name}"] = float(ew.std())
        out[f"dW2_shrink_{name}"] = float(np.dot(ew, gw) / np.dot(gw, gw))          # e = shrink * g + noise
    out["grad_err_oracle"] = float(np.linalg.norm(go.astype(np.float64) - g64)) / gn
    out["loss_err_oracle"] = abs(float(lo) - l64) / abs(l64)
    return out


def within_budget(m):
    """the criteria of the error-budget test:
      grad : the split kernel is no further from the float64 gradient than twice the exact-f32 kernel;
      loss : the same for the loss (floor of one f32 ulp of the loss: both errors are single roundings at that level);
      bias : the signed mean of the dW2 error is a small fraction (<= 1/4) of its spread, and the error has no component along the gradient itself
             (shrinkage e = s g + noise) beyond the f32 kernel's by more than 1e-7 — round 2's truncation split measured s = -2.1e-7 against -2.3e-9"""
    ok_grad = m["grad_err_split"] <= 2.0 * m["grad_err_f32"]
    ok_loss = m["loss_err_split"] <= max(2.0 * m["loss_err_f32"], 2.0 ** -23)
    ok_bias = abs(m["dW2_mean_err_split"]) <= 0.25 * m["dW2_std_err_split"] and abs(m["dW2_shrink_split"] - m["dW2_shrink_f32"]) <= 1e-7
    return bool(ok_grad), bool(ok_loss), bool(ok_bias)


if __name__ == "__main__":
    import __graft_entry__ as g
    import oracle_lib
    pkg = g.load_package()
    cases = json.loads(sys.argv[1])
    print(json.dumps([measure(pkg, oracle_lib, *c) for c in cases]))

#!/usr/bin/env python3
"""Writes tests/golden/sac_kats.json: known answers for the SAC path taken from the reference's own tests and closed forms.
Nothing here imports the oracle or the product; numbers are either literal values of the reference's tests or f64 closed forms.

  polyak            test/test_utils.jl:4-25 (literal inputs and expected outputs of the reference test)
  squashed_logpdf   src/DRiLDistributions/squashedDiagGaussian.jl:36-46 — the change-of-variables identity its comment cites:
                    log p(x) = log N(atanh x; mean, exp(log_std)) - sum log(1 - x^2), evaluated in float64
  schedule          src/algorithms/sac.jl:436-447,59-65 — [n_steps of the first collection, iterations, total_steps, updates per iteration]
                    worked by hand from those lines (Julia div truncates toward zero)
"""
import json
import math
from pathlib import Path

import numpy as np

out = {"polyak": [
    {"target": [1.0, 2.0, 3.0], "source": [0.0, 0.0, 0.0], "tau": 0.5, "expected": [0.5, 1.0, 1.5]},
    {"target": [0.0, 0.0, 0.0], "source": [1.0, 2.0, 3.0], "tau": 0.01, "expected": [0.01, 0.02, 0.03]},
]}
rng = np.random.default_rng(20240607)
cases = []
for k in (1, 1, 2, 3, 6):
    mean = rng.normal(0, 0.8, k); log_std = rng.uniform(-3.0, 0.3, k)
    x = np.tanh(mean + np.exp(log_std) * rng.normal(0, 1, k))
    x32, m32, l32 = x.astype(np.float32), mean.astype(np.float32), log_std.astype(np.float32)
    g = np.arctanh(x32.astype(np.float64))
    lp = -0.5 * (2 * l32.astype(np.float64).sum() + (((g - m32) ** 2) * np.exp(-2.0 * l32.astype(np.float64))).sum() + k * math.log(2 * math.pi))
    lp -= np.log1p(-x32.astype(np.float64) ** 2).sum()
    cases.append({"x": x32.tolist(), "mean": m32.tolist(), "log_std": l32.tolist(), "expected": float(lp)})
out["squashed_logpdf"] = cases
out["schedule"] = [
    {"max_steps": 1000, "n_envs": 4, "start_steps": 100, "train_freq": 1, "gradient_steps": 1, "expected": [25, 226, 1000, 1]},
    {"max_steps": 144, "n_envs": 8, "start_steps": 64, "train_freq": 2, "gradient_steps": 3, "expected": [8, 6, 144, 3]},
    {"max_steps": 100, "n_envs": 8, "start_steps": 0, "train_freq": 2, "gradient_steps": -1, "expected": [2, 6, 96, 16]},
    {"max_steps": 5, "n_envs": 8, "start_steps": 3, "train_freq": 1, "gradient_steps": 1, "expected": [1, 1, 8, 1]},
    {"max_steps": 64, "n_envs": 8, "start_steps": 64, "train_freq": 2, "gradient_steps": -1, "expected": [8, 1, 64, 16]},
]
Path(__file__).with_name("sac_kats.json").write_text(json.dumps(out, indent=1))

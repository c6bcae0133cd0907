#!/usr/bin/env python3
"""Writes tests/golden/scaling_kats.json: the literal inputs and expected outputs of the reference's own ScalingWrapperEnv tests
(test/test_scaling_wrapper.jl:42-130,208-244).  Data only — nothing here imports the oracle or the product."""
import json
from pathlib import Path

out = {
    "observation": [   # test_scaling_wrapper.jl:42-84: Box([0,-10,5],[10,10,25])
        {"low": [0.0, -10.0, 5.0], "high": [10.0, 10.0, 25.0], "x": [5.0, 0.0, 15.0], "expected": [0.0, 0.0, 0.0], "atol": 1e-6},
        {"low": [0.0, -10.0, 5.0], "high": [10.0, 10.0, 25.0], "x": [0.0, -10.0, 5.0], "expected": [-1.0, -1.0, -1.0], "atol": 1e-6},
        {"low": [0.0, -10.0, 5.0], "high": [10.0, 10.0, 25.0], "x": [10.0, 10.0, 25.0], "expected": [1.0, 1.0, 1.0], "atol": 1e-6},
        {"low": [-1000.0, -500.0], "high": [2000.0, 1500.0], "x": [500.0, 500.0], "expected": [0.0, 0.0], "atol": 1e-5},   # :208-236
    ],
    "action": [        # :86-130: Box([2,-5,0],[8,15,10]); :238-243: Box([-100],[300])
        {"low": [2.0, -5.0, 0.0], "high": [8.0, 15.0, 10.0], "x": [0.0, 0.0, 0.0], "expected": [5.0, 5.0, 5.0], "atol": 1e-6},
        {"low": [2.0, -5.0, 0.0], "high": [8.0, 15.0, 10.0], "x": [-1.0, -1.0, -1.0], "expected": [2.0, -5.0, 0.0], "atol": 1e-6},
        {"low": [2.0, -5.0, 0.0], "high": [8.0, 15.0, 10.0], "x": [1.0, 1.0, 1.0], "expected": [8.0, 15.0, 10.0], "atol": 1e-6},
        {"low": [-100.0], "high": [300.0], "x": [0.5], "expected": [200.0], "atol": 1e-5},
    ],
}
Path(__file__).with_name("scaling_kats.json").write_text(json.dumps(out, indent=1))

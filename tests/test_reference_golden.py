"""A vector from DRiL.jl ITSELF for the PPO loss, its gradient and the clip + Adam step — when someone with Julia has produced it.

The reference's own tests hold no such vector, so loss / gradient / Adam are pinned here by torch-f64 autograd only (tests/test_oracle_crosschecks.py).
`tests/golden/gen_reference_golden.jl` turns that into a one-command job: it loads DRiL.jl, feeds it the committed inputs (`reference_ppo_input.json`, written by
`make_reference_input.py`) and writes `reference_ppo.json`.  This module consumes that file when it is present — the CPU oracle here, the device through the C ABI under
`-m gpu` — and SKIPS with this message when it is absent (the build image has no Julia).  Nothing else depends on it."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
SKIP = ("tests/golden/reference_ppo.json is absent: run `julia --project=<DRiL.jl> tests/golden/gen_reference_golden.jl` once on a machine with Julia "
        "(inputs: tests/golden/reference_ppo_input.json) and commit its output")


def _inputs():
    return json.loads((GOLD / "reference_ppo_input.json").read_text())["cases"]


def _golden():
    p = GOLD / "reference_ppo.json"
    if not p.exists():
        pytest.skip(SKIP)
    return {c["name"]: c for c in json.loads(p.read_text())["cases"]}


def _cfg_of(pkg, case):
    capi = pkg._capi
    cfg = capi.default_config(case["env_kind"])
    cfg.n_envs, cfg.n_steps, cfg.batch_size = 2, 2, 2
    hp = case["hyper"]
    cfg.clip_range, cfg.ent_coef, cfg.vf_coef, cfg.max_grad_norm = hp["clip_range"], hp["ent_coef"], hp["vf_coef"], hp["max_grad_norm"]
    cfg.normalize_advantage, cfg.learning_rate = int(hp["normalize_advantage"]), hp["learning_rate"]
    if hp["clip_range_vf"] is not None:
        cfg.has_clip_range_vf, cfg.clip_range_vf = 1, hp["clip_range_vf"]
    return cfg


def _batch_of(case):
    f = lambda k: np.asarray(case[k], np.float32)
    B, D, A = case["B"], case["D"], case["A"]
    act = np.asarray(case["actions"], np.int32) if case["discrete"] else f("actions").reshape(B, A)
    return f("obs").reshape(B, D), act, f("advantages"), f("returns"), f("old_logprobs"), f("old_values")


def _check(make, case, gold):
    x = make()
    flat = np.asarray(case["params"], np.float32)
    x.set_params(flat)
    batch = _batch_of(case)
    loss, _, grad = x.ppo_loss_grad(*batch)
    gref = np.asarray(gold["grad"], np.float64)
    assert loss == pytest.approx(gold["loss"], rel=1e-4)                                  # north_star: PPO loss to 1e-4 relative
    assert np.linalg.norm(grad - gref) <= 3e-4 * np.linalg.norm(gref)
    assert np.linalg.norm(grad) == pytest.approx(gold["grad_norm"], rel=1e-4)
    for k, step in enumerate(gold["adam_steps"]):                                         # the batch loop's step, three times on this minibatch: gradient, clip, Adam(3e-4, (0.9, 0.999), 1e-5)
        lk, _, g = x.ppo_loss_grad(*batch)
        assert lk == pytest.approx(step["loss_before"], rel=1e-4), k
        n = x.apply_gradients(g)
        assert n == pytest.approx(step["grad_norm"], rel=1e-4), k
        np.testing.assert_allclose(x.get_params(), np.asarray(step["params_after"], np.float32), rtol=1e-4, atol=2e-6, err_msg=f"step {k}")


def test_the_committed_input_is_what_its_generator_writes(tmp_path):
    """reference_ppo_input.json must stay reproducible from make_reference_input.py (seeded numpy + the CPU oracle): a hand-edited input would pin nothing"""
    before = (GOLD / "reference_ppo_input.json").read_text()
    r = subprocess.run([sys.executable, str(GOLD / "make_reference_input.py")], capture_output=True, text=True, timeout=600)
    try:
        assert r.returncode == 0, r.stderr[-2000:]
        assert (GOLD / "reference_ppo_input.json").read_text() == before
    finally:
        (GOLD / "reference_ppo_input.json").write_text(before)
    cases = _inputs()
    assert [c["name"] for c in cases] == ["cartpole_B64_defaults", "cartpole_B1024_ent_vfclip", "pendulum_B64_defaults"]
    assert all(len(c["obs"]) == c["B"] * c["D"] and len(c["old_logprobs"]) == c["B"] for c in cases)


def test_the_generator_script_names_what_the_reference_defines():
    """static check of gen_reference_golden.jl (it cannot run here): every DRiL name it uses is defined at the cited place of the reference — skipped on the GPU box"""
    ref = Path("/root/reference/src")
    if not ref.exists():
        pytest.skip("needs the reference sources")
    jl = (GOLD / "gen_reference_golden.jl").read_text()
    src = "\n".join(p.read_text() for p in ref.rglob("*.jl"))
    for name in ("nested_norm", "nested_scale!", "make_optimizer", "DiscreteActorCriticLayer", "ContinuousActorCriticLayer"):
        assert name in jl and ("function " + name in src or "struct " + name in src), name
    assert "Optimisers.update!" in jl and "Zygote.gradient" in jl
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_shim", ROOT / "tools" / "check_shim.py")
    cs = importlib.util.module_from_spec(spec); spec.loader.exec_module(cs)
    o, e = cs.block_balance(jl)
    assert o == e, (o, e)


@pytest.mark.parametrize("name", ["cartpole_B64_defaults", "cartpole_B1024_ent_vfclip", "pendulum_B64_defaults"])
def test_oracle_matches_the_reference_vector(pkg, oracle_mod, name):
    gold = _golden()
    case = {c["name"]: c for c in _inputs()}[name]
    _check(lambda: oracle_mod.Oracle(_cfg_of(pkg, case)), case, gold[name])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cartpole_B64_defaults", "cartpole_B1024_ent_vfclip", "pendulum_B64_defaults"])
def test_device_matches_the_reference_vector(pkg, name):
    gold = _golden()
    case = {c["name"]: c for c in _inputs()}[name]
    _check(lambda: pkg.Handle(_cfg_of(pkg, case)), case, gold[name])

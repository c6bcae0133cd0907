"""CPU: pins the oracle against the reference's analytic tests (tests/golden, see make_golden.py)."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).parent / "golden"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("case", json.loads((G / "gae.json").read_text()), ids=lambda c: c["name"])
def test_gae_known_answers(oracle_mod, case):
    L = oracle_mod.lib()
    E, T = case["n_envs"], case["n_steps"]
    f32 = lambda k: np.asarray(case[k], np.float32)
    adv = np.zeros(E * T, np.float32); ret = np.zeros(E * T, np.float32)
    flags = np.asarray(case["flags"], np.uint8)
    r, v, b, lv = f32("rewards"), f32("values"), f32("bootstrap"), f32("last_values")
    assert L.orc_gae(E, T, case["gamma"], case["gae_lambda"], _p(r), _p(v), _p(flags), _p(b), _p(lv), _p(adv), _p(ret)) == 0
    np.testing.assert_allclose(adv, case["expected_advantages"], atol=case["atol"], rtol=0)
    np.testing.assert_allclose(ret, case["expected_returns"], atol=case["atol"], rtol=0)  # returns = adv + values, rollout_buffer.jl:87


def test_compute_advantages_terminated_vs_bootstrap(oracle_mod):
    """test/test_buffers.jl:60-115 through compute_advantages! itself (trajectory.jl:80-102)."""
    L = oracle_mod.lib()
    r = np.array([0, 0, 0, 0, 0, 1], np.float32); v = np.full(6, 0.7, np.float32)
    a_term = np.zeros(6, np.float32); a_trunc = np.zeros(6, np.float32)
    L.orc_compute_advantages(_p(a_term), _p(r), _p(v), 6, 1, 0, 0.0, 0.9, 0.8)
    L.orc_compute_advantages(_p(a_trunc), _p(r), _p(v), 6, 0, 1, 0.2, 0.9, 0.8)
    cases = {c["name"]: c for c in json.loads((G / "gae.json").read_text())}
    np.testing.assert_allclose(a_term, cases["buffers_terminated"]["expected_advantages"], atol=1e-4)
    np.testing.assert_allclose(a_trunc, cases["buffers_truncated_boot0.2"]["expected_advantages"], atol=1e-4)
    assert not np.allclose(a_term, a_trunc, atol=1e-3)


def test_distributions_closed_form(oracle_mod):
    """test/test_distributions.jl:1-39 (DiagGaussian ~ MvNormal), :94-119 (Categorical)."""
    L = oracle_mod.lib()
    d = json.loads((G / "distributions.json").read_text())
    for g in d["diag_gaussian"]:
        x, mu, ls = (np.asarray(g[k], np.float32) for k in ("x", "mean", "log_std"))
        lp = L.orc_gauss_logpdf(_p(x), _p(mu), _p(ls), g["k"])
        ent = L.orc_gauss_entropy(_p(ls), g["k"])
        assert lp == pytest.approx(g["logpdf"], rel=2e-5, abs=2e-5)   # Julia's ≈ is rtol sqrt(eps(Float32)) ~ 3.4e-4
        assert ent == pytest.approx(g["entropy"], rel=2e-5, abs=2e-5)
    for c in d["categorical"]:
        p = np.asarray(c["p"], np.float32)
        assert L.orc_categorical_logpdf(_p(p), p.size, 1, 1) == pytest.approx(c["logpdf_first"], rel=1e-5, abs=1e-6)
        assert L.orc_categorical_entropy(_p(p), p.size) == pytest.approx(c["entropy"], rel=1e-5, abs=1e-6)
        for s in c["samples"]:
            assert L.orc_categorical_sample(_p(p), p.size, s["u"], 1) == s["index"] + 1   # start = 1 default, spaces.jl:160


@pytest.mark.parametrize("case", json.loads((G / "running_mean_std.json").read_text()), ids=lambda c: c["name"])
def test_running_mean_std(oracle_mod, case):
    """test/test_normalize_wrapper.jl:3-70."""
    L = oracle_mod.lib()
    d = case["dims"]
    mean = np.zeros(d, np.float32); var = np.ones(d, np.float32); cnt = C.c_int64(0)
    for i, b in enumerate(case["batches"]):
        b = np.ascontiguousarray(b, np.float32)
        L.orc_rms_update(_p(mean), _p(var), C.byref(cnt), d, _p(b), b.shape[0])
        assert cnt.value == case["count_after"][i]
        np.testing.assert_allclose(mean, case["mean_after"][i], atol=case["atol"][i] * max(1.0, np.abs(mean).max()))
        np.testing.assert_allclose(var, case["var_after"][i], atol=case["atol"][i] * max(1.0, np.abs(var).max()))


def test_param_counts(oracle_mod, pkg):
    """test/test_policies.jl:36-64 + SURVEY.md §8 a6 (9 155 / 134 147)."""
    for c in json.loads((G / "param_counts.json").read_text()):
        env = pkg.CartPoleEnv() if c["discrete"] else pkg.PendulumEnv()
        layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=tuple(c["hidden"]))
        assert layer.parameterlength() == c["total"]
        assert pkg.flatten_params(layer.initialparameters(np.random.default_rng(0))).size == c["total"]
        cfg = pkg._capi.default_config(env.kind); cfg.hidden1, cfg.hidden2 = c["hidden"]; cfg.n_envs = 2; cfg.n_steps = 2
        assert oracle_mod.Oracle(cfg).P == c["total"]


def test_rollout_logprob_value_consistency(oracle_mod, pkg):
    """test/test_buffers.jl:166-278: stored logprobs/values == evaluate_actions(obs, actions) recomputed (1e-5)."""
    for kind in (0, 1):
        cfg = pkg._capi.default_config(kind); cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs = 3, 40, 8, 1
        cfg.episode_len = 17
        o = oracle_mod.Oracle(cfg)
        rng = np.random.default_rng(kind)
        o.set_params((rng.standard_normal(o.P) * 0.3).astype(np.float32))
        o.env_reset(11); o.collect_rollout()
        capi = pkg._capi
        obs, act = o.buffer(capi.BUF_OBSERVATIONS), o.buffer(capi.BUF_ACTIONS)
        val, lp, ent = o.evaluate_actions(obs, act)
        np.testing.assert_allclose(val, o.buffer(capi.BUF_VALUES), atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(lp, o.buffer(capi.BUF_LOGPROBS), atol=1e-5, rtol=1e-5)
        ret, adv = o.buffer(capi.BUF_RETURNS), o.buffer(capi.BUF_ADVANTAGES)
        np.testing.assert_allclose(ret, adv + o.buffer(capi.BUF_VALUES), atol=1e-5)   # test_buffers.jl:117-164
        assert np.all(np.isfinite(adv))
        if kind == 0:
            assert act.dtype == np.int32 and set(np.unique(act)) <= {1, 2} and np.all(ent >= 0)   # actions in Discrete(2,1), test_buffers.jl:185-198
            assert np.all(o.buffer(capi.BUF_REWARDS) >= 0)
        # the reference-order map covers every time-major slot exactly once (rollout_buffer.jl:70-80)
        assert np.array_equal(np.sort(o.ref_order()), np.arange(o.N))
        flags = o.buffer(capi.BUF_FLAGS)
        assert (flags & 2).sum() >= 2 * cfg.n_envs   # episode_len 17 over 40 steps: two truncations per env


def test_scaling_wrapper_known_answers(pkg, oracle_mod):
    """the reference's own ScalingWrapperEnv cases (test/test_scaling_wrapper.jl:42-130,208-244) through the oracle's scale!/unscale!
    restatement (scalingWrapperEnv.jl:71-79) and through the host mirror's helpers"""
    import ctypes as C
    import json
    from pathlib import Path
    G = json.loads((Path(__file__).parent / "golden" / "scaling_kats.json").read_text())
    L = oracle_mod.lib()
    for fn in (L.orc_scale, L.orc_unscale):
        fn.restype = None; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for kind, fn in (("observation", L.orc_scale), ("action", L.orc_unscale)):
        for c in G[kind]:
            x, lo, hi = (np.array(c[k], np.float32) for k in ("x", "low", "high"))
            fn(x.ctypes.data, lo.ctypes.data, hi.ctypes.data, x.size)
            np.testing.assert_allclose(x, c["expected"], atol=c["atol"], rtol=0)
    w = pkg.ScalingWrapperEnv(pkg.PendulumEnv())
    assert w.observation_space() == pkg.Box((-1.0,) * 3, (1.0,) * 3) and w.action_space() == pkg.Box((-1.0,), (1.0,))      # :26-33
    np.testing.assert_allclose(w.scale_observation([1.0, -1.0, 0.0]), [1.0, -1.0, 0.0], atol=1e-7)
    np.testing.assert_allclose(w.unscale_observation(w.scale_observation([0.3, -0.7, 5.0])), [0.3, -0.7, 5.0], atol=1e-6)
    np.testing.assert_allclose(w.unscale_action([-1.0, 0.0, 0.25, 1.0]), [-2.0, 0.0, 0.5, 2.0], atol=1e-7)
    with pytest.raises(NotImplementedError):
        pkg.ScalingWrapperEnv(pkg.CartPoleEnv())                      # Box / Box only (scalingWrapperEnv.jl:22)


def test_scaled_pendulum_oracle_semantics(pkg, oracle_mod):
    """ScalingWrapperEnv(Pendulum) in the oracle: observations are the scaled Pendulum observations, a wrapper-space action a is the torque
    2a, and the rollout stores raw actions while the env sees the ClampAdapter'ed action of the wrapper's Box(-1,1)"""
    capi = pkg._capi
    def mk(kind):
        c = capi.default_config(kind); c.n_envs, c.n_steps, c.episode_len, c.batch_size = 6, 4, 3, 24
        o = oracle_mod.Oracle(c); o.env_reset(5); return o
    a, b = mk(capi.ENV_PENDULUM), mk(capi.ENV_PENDULUM_SCALED)
    oa, ob = a.env_observe(), b.env_observe()
    w = pkg.ScalingWrapperEnv(pkg.PendulumEnv())
    np.testing.assert_allclose(ob, w.scale_observation(oa), atol=1e-7)
    act = np.linspace(-1.3, 1.3, 6, dtype=np.float32).reshape(6, 1)
    ra, ta, ua, xa = a.env_step(np.clip(2 * act, -2, 2))
    rb, tb, ub, xb = b.env_step(act)
    np.testing.assert_allclose(rb, ra, rtol=1e-6, atol=1e-6); np.testing.assert_array_equal(ub, ua)
    np.testing.assert_allclose(b.env_observe(), w.scale_observation(a.env_observe()), atol=1e-6)
    for _ in range(2):
        ra, ta, ua, xa = a.env_step(np.zeros((6, 1), np.float32)); rb, tb, ub, xb = b.env_step(np.zeros((6, 1), np.float32))
    assert ub.all()
    np.testing.assert_allclose(xb, w.scale_observation(xa), atol=1e-6)                          # terminal observation passes through the wrapper too

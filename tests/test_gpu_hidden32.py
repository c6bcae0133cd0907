"""GPU (-m gpu): hidden_dims = [32, 32] — the shape of the reference's own benchmark suite (benchmark/bench_utils.jl:31,49: CartPole / Pendulum, hidden [32,32];
`rollouts/rollout_buffer` 2 envs x 64 steps, `training/ppo_cartpole` n_steps 32, batch 32, 1 epoch) — on the FUSED kernels (rollout_duo_kernel / rollout_kernel /
policy_kernel / ppo_grad_kernel instantiated at H = 32: one m-tile per layer, f32 MFMA) instead of the generic layer-by-layer path it took until round 3.
Checker: the CPU oracle, tolerances of tests/test_gpu_parity.py; and the generic path of the same library (DRIL_FORCE_GENERIC) as a second implementation.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(pkg, kind, **kw):
    c = pkg._capi.default_config(kind)
    c.hidden1 = c.hidden2 = 32
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _params(P, seed, scale=0.4):
    return (np.random.default_rng(seed).standard_normal(P) * scale).astype(np.float32)


@pytest.mark.parametrize("kind,B", [(0, 33), (1, 100), (3, 70), (4, 45), (6, 77), (0, 4096)])
def test_forward_evaluate_predict(pkg, oracle_mod, kind, B):
    cfg = _cfg(pkg, kind, n_envs=2, n_steps=2, batch_size=2)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    assert h.P == o.P == (h.D * 32 + 32 + 32 * 32 + 32) * 2 + 32 * h.A + h.A + 32 + 1 + (0 if h.discrete else h.A)      # Lux.parameterlength of the two [32,32] heads
    flat = _params(h.P, 10 + kind); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(3)
    obs = rng.uniform(-1.5, 1.5, (B, h.D)).astype(np.float32)
    noise = rng.random(B) if h.discrete else rng.standard_normal((B, h.A)).astype(np.float32)
    ah, vh, lh = h.policy_forward(obs, noise); ao, vo, lo = o.policy_forward(obs, noise)
    np.testing.assert_allclose(vh, vo, atol=2e-5, rtol=2e-5)
    same = (ah == ao) if h.discrete else np.ones(B, bool)
    assert same.mean() >= 0.99
    if not h.discrete:
        np.testing.assert_allclose(ah, ao, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(lh[same], lo[same], atol=5e-5, rtol=5e-5)
    ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
    np.testing.assert_allclose(ve, vo2, atol=2e-5, rtol=2e-5); np.testing.assert_allclose(le, lo2, atol=5e-5, rtol=5e-5); np.testing.assert_allclose(ee, eo2, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(h.predict_values(obs), o.predict_values(obs), atol=2e-5, rtol=2e-5)
    assert h.f32_fallback_info()["forward_exact_f32"] == 1            # hidden 32 exists on the f32-MFMA forward only


@pytest.mark.parametrize("kind,B,variant", [(0, 32, "default"), (0, 333, "ent_vfclip"), (1, 64, "default"), (1, 1000, "ent_vfclip"), (3, 200, "default"), (4, 129, "ent_vfclip"), (6, 150, "default"), (0, 16403, "default")])
def test_ppo_loss_and_gradient(pkg, oracle_mod, kind, B, variant):
    kw = dict(n_envs=2, n_steps=2, batch_size=2)
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    cfg = _cfg(pkg, kind, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 20 + kind, 0.3); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(B)
    obs = rng.uniform(-1, 1, (B, h.D)).astype(np.float32)
    act = (rng.integers(0, h.A, B) + cfg.action_start).astype(np.int32) if h.discrete else rng.normal(0, 1, (B, h.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    lp = (o.evaluate_actions(obs, act)[1] + rng.normal(0, 0.1, B)).astype(np.float32)
    lh, sh, gh = h.ppo_loss_grad(obs, act, adv, ret, lp, ov); lo, so, go = o.ppo_loss_grad(obs, act, adv, ret, lp, ov)
    assert h.grad_kernel_info().split(":")[0] == "ppo_grad_kernel"
    assert lh == pytest.approx(lo, rel=1e-4)                           # BASELINE.json: PPO loss rel-err <= 1e-4
    np.testing.assert_allclose(sh, so, rtol=2e-4, atol=2e-6)
    assert np.linalg.norm(gh - go) <= 2e-4 * np.linalg.norm(go)
    lh2, _, gh2 = h.ppo_loss_grad(obs, act, adv, ret, lp, ov)
    assert lh2 == lh and np.array_equal(gh, gh2)                        # deterministic slabs


@pytest.mark.parametrize("kind,E,T,B", [(0, 2, 64, 32), (1, 2, 64, 32), (0, 300, 40, 1000), (1, 20000, 12, 40000), (6, 64, 32, 512)])
def test_rollout_and_update(pkg, oracle_mod, kind, E, T, B):
    """collect_rollout! (rollout_duo_kernel below 16 384 envs, rollout_kernel above) with truncation bootstraps, GAE, then the update on the oracle's buffer with an
    injected DataLoader order; (2, 64, 32) is benchmark/bench_utils.jl's own shape"""
    capi = pkg._capi
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=11, batch_size=B, epochs=2, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 30 + kind, 0.4); h.set_params(flat); o.set_params(flat)
    h.env_reset(4); o.env_reset(4)
    nz = np.random.default_rng(0).random(E * T) if h.discrete else np.random.default_rng(0).standard_normal((E * T, h.A)).astype(np.float32)
    h.set_noise(nz); o.set_noise(nz)
    h.collect_rollout(); o.collect_rollout()
    assert (o.buffer(capi.BUF_FLAGS) & 2).any()
    if h.discrete:
        ok = np.cumprod(h.buffer(capi.BUF_ACTIONS).reshape(T, E) == o.buffer(capi.BUF_ACTIONS).reshape(T, E), axis=0).astype(bool).all(axis=0)
        assert ok.mean() >= 0.9
    else:
        ok = np.ones(E, bool)
    for which, tol in ((capi.BUF_OBSERVATIONS, 2e-4), (capi.BUF_VALUES, 2e-4), (capi.BUF_LOGPROBS, 2e-4), (capi.BUF_REWARDS, 2e-4), (capi.BUF_ADVANTAGES, 2e-3), (capi.BUF_RETURNS, 2e-3)):
        a, b = h.buffer(which).reshape(T, E, -1), o.buffer(which).reshape(T, E, -1)
        np.testing.assert_allclose(a[:, ok], b[:, ok], atol=tol, rtol=tol, err_msg=str(which))
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(2)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates and sh.f32_path == 0
    for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance"):
        assert getattr(sh, f) == pytest.approx(getattr(so, f), rel=5e-4, abs=2e-6), f
    assert sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    h.close()


def test_fused_and_generic_paths_agree_and_the_mirror_trains(pkg, monkeypatch):
    """the same [32,32] agent on the fused kernels and on the generic layer-by-layer kernels (DRIL_FORCE_GENERIC): two implementations inside the library, one answer;
    and train! through the reference-shaped mirror on the benchmark suite's PPO (n_steps 32, batch 32, 1 epoch)"""
    res = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("DRIL_FORCE_GENERIC", "1")
        cfg = _cfg(pkg, 0, n_envs=64, n_steps=32, episode_len=9, batch_size=256, epochs=2)
        h = pkg.Handle(cfg)
        monkeypatch.delenv("DRIL_FORCE_GENERIC", raising=False)
        h.set_params(_params(h.P, 5, 0.4)); h.env_reset(2)
        nz = np.random.default_rng(1).random(64 * 32); h.set_noise(nz)
        h.collect_rollout()
        perm = np.stack([np.random.default_rng(e).permutation(64 * 32) for e in range(2)]).astype(np.int64); h.set_permutation(perm)
        st = h.ppo_update()
        res.append((h.buffer(pkg._capi.BUF_VALUES), st.loss, h.get_params(), h.grad_kernel_info().split(":")[0]))
        h.close()
    assert res[0][3] == "ppo_grad_kernel" and res[1][3] != "ppo_grad_kernel"
    np.testing.assert_allclose(res[0][0], res[1][0], atol=2e-5, rtol=2e-5)
    assert res[0][1] == pytest.approx(res[1][1], rel=1e-4)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=2e-4, atol=3e-6)
    env = pkg.DeviceParallelEnv(pkg.CartPoleEnv(), 2, seed=42)
    alg = pkg.PPO(n_steps=32, batch_size=32, epochs=1)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(32, 32)), alg)
    stats, _ = pkg.train_(agent, env, alg, 256)                       # benchmark/bench_utils.jl:78-83: 256 total steps
    assert len(stats["losses"]) == 4 and np.isfinite(stats["losses"]).all()
    assert not env.handle.grad_kernel_info().startswith("generic")

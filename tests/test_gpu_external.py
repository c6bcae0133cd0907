"""GPU (-m gpu): DRIL_ENV_EXTERNAL — the caller's own host envs with any obs / action / hidden width — through the C ABI vs the CPU oracle.
The generic path runs layer by layer on the strided fp32-MFMA contraction (dril_generic.hip / dril_gemm.hip); same fp32 tolerances as
test_gpu_parity.py.  One property test ties it to the fused path: with CartPole's spaces it must reproduce the fused kernels' rollout."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (obs dim, action dim, discrete, hidden1, hidden2): odd sizes on purpose (no multiple of 4 / 32 anywhere), unequal hidden widths, wide nets
SHAPES = [(6, 3, True, 64, 64), (11, 5, False, 48, 80), (1, 1, False, 7, 5), (33, 17, True, 100, 36), (8, 2, False, 256, 256), (128, 64, True, 32, 32)]


def _ext_cfg(pkg, D, A, discrete, H1, H2, **kw):
    c = pkg._capi.default_config(pkg._capi.ENV_EXTERNAL)
    c.ext_obs_dim, c.ext_action_dim, c.ext_discrete, c.hidden1, c.hidden2 = D, A, int(discrete), H1, H2
    c.ext_action_low, c.ext_action_high = -1.0, 1.0
    c.n_envs, c.n_steps, c.batch_size = 2, 2, 2
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _params(P, seed, scale=0.3):
    return (np.random.default_rng(seed).standard_normal(P) * scale).astype(np.float32)


def _noise(rng, h, n):
    return rng.random(n) if h.discrete else rng.standard_normal((n, h.A)).astype(np.float32)


@pytest.mark.parametrize("D,A,discrete,H1,H2", SHAPES)
def test_generic_forward_evaluate_predict(pkg, oracle_mod, D, A, discrete, H1, H2):
    """layer(obs, ps, st), evaluate_actions, predict_values (layer_forward.jl:3-13,30-39; layer_methods.jl:28-61) for arbitrary spaces"""
    cfg = _ext_cfg(pkg, D, A, discrete, H1, H2)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    assert (h.D, h.A, h.discrete) == (D, A, discrete)
    net = lambda out: D * H1 + H1 + H1 * H2 + H2 + H2 * out + out
    assert h.P == o.P == net(A) + net(1) + (0 if discrete else A)            # Lux.parameterlength, test/test_policies.jl:55-57
    flat = _params(h.P, 1, 0.2); h.set_params(flat); o.set_params(flat)
    assert np.array_equal(h.get_params(), flat)
    rng = np.random.default_rng(D)
    for B in (1, 37, 1000):
        obs = rng.uniform(-2, 2, (B, D)).astype(np.float32); nz = _noise(rng, h, B)
        ah, vh, lh = h.policy_forward(obs, nz); ao, vo, lo = o.policy_forward(obs, nz)
        np.testing.assert_allclose(vh, vo, atol=5e-5, rtol=5e-5)
        if discrete:
            assert (ah == ao).mean() >= 0.99 and ah.min() >= cfg.action_start and ah.max() < cfg.action_start + A
            same = ah == ao
            np.testing.assert_allclose(lh[same], lo[same], atol=1e-4, rtol=1e-4)
        else:
            np.testing.assert_allclose(ah, ao, atol=5e-5, rtol=5e-5); np.testing.assert_allclose(lh, lo, atol=3e-4, rtol=3e-4)
        ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
        np.testing.assert_allclose(ve, vo2, atol=5e-5, rtol=5e-5); np.testing.assert_allclose(le, lo2, atol=3e-4, rtol=3e-4)
        np.testing.assert_allclose(ee, eo2, atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(h.predict_values(obs), vo, atol=5e-5, rtol=5e-5)
        # predict_actions(...; deterministic) (layer_methods.jl:3-26): mode(d) / rand(d)
        dh, do = h.predict_actions(obs, True), o.predict_actions(obs, True)
        sh_, so_ = h.predict_actions(obs, False, nz), o.predict_actions(obs, False, nz)
        if discrete:
            assert (dh == do).mean() >= 0.99 and (sh_ == ah).all() and (so_ == ao).all()
        else:
            np.testing.assert_allclose(dh, do, atol=5e-5, rtol=5e-5); np.testing.assert_allclose(sh_, ah, atol=0, rtol=0)


@pytest.mark.parametrize("variant", ["default", "ent_vfclip", "no_norm"])
@pytest.mark.parametrize("D,A,discrete,H1,H2", SHAPES)
def test_generic_ppo_loss_and_gradient(pkg, oracle_mod, D, A, discrete, H1, H2, variant):
    """(alg::PPO)(...) ppo.jl:365-407 + its gradient through the generic path: loss within 1e-4 rel (north_star), gradient within fp32 noise"""
    kw = {}
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    if variant == "no_norm":
        kw.update(normalize_advantage=0)
    cfg = _ext_cfg(pkg, D, A, discrete, H1, H2, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 40, 0.2); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(A)
    for B in (2, 333, 20000):                                                        # 20000 rows: three row chunks (slabs), the last one ragged
        obs = rng.uniform(-1, 1, (B, D)).astype(np.float32)
        act = (rng.integers(0, A, B) + cfg.action_start).astype(np.int32) if discrete else rng.normal(0, 1, (B, A)).astype(np.float32)
        adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
        lp = (o.evaluate_actions(obs, act)[1] + rng.normal(0, 0.1, B)).astype(np.float32)
        lh, sh, gh = h.ppo_loss_grad(obs, act, adv, ret, lp, ov); lo, so, go = o.ppo_loss_grad(obs, act, adv, ret, lp, ov)
        assert lh == pytest.approx(lo, rel=1e-4, abs=1e-6)
        np.testing.assert_allclose(sh, so, rtol=3e-4, atol=3e-6)
        assert np.linalg.norm(gh - go) <= 3e-4 * np.linalg.norm(go) + 1e-7
        if not discrete:                                                             # the log_std gradient sits behind both nets
            np.testing.assert_allclose(gh[-A:], go[-A:], rtol=2e-3, atol=2e-6)
        lh2, _, gh2 = h.ppo_loss_grad(obs, act, adv, ret, lp, ov)
        assert lh2 == lh and np.array_equal(gh, gh2)                                 # fixed summation order: bitwise reproducible


class _ToyEnvs:
    """E deterministic host envs with D-dim observations: x' = 0.9 x + 0.1 f(action) + drift, reward = -|x|^2 mean, terminated when |x_0| > 1.5,
    truncated every `limit` steps (per-env phase); auto-reset + terminal_observation on truncation like BroadcastedParallelEnv"""

    def __init__(self, E, D, A, discrete, limit, seed):
        self.E, self.D, self.A, self.discrete, self.limit = E, D, A, discrete, limit
        self.rng = np.random.default_rng(seed)
        self.W = self.rng.standard_normal((A, D)).astype(np.float32) * 0.5
        self.x = self.rng.uniform(-1, 1, (E, D)).astype(np.float32)
        self.t = (np.arange(E) % limit).astype(np.int64)

    def observe(self):
        return self.x.copy()

    def step(self, env_actions, action_start):
        if self.discrete:
            f = self.W[np.asarray(env_actions) - action_start]
        else:
            f = np.asarray(env_actions, np.float32).reshape(self.E, self.A) @ self.W
        self.x = (0.9 * self.x + 0.3 * f + 0.01).astype(np.float32)
        self.t += 1
        rew = -(self.x ** 2).mean(axis=1).astype(np.float32)
        term = np.abs(self.x[:, 0]) > 1.5
        trunc = self.t >= self.limit
        tobs = self.x.copy()
        done = term | trunc
        self.x[done] = self.rng.uniform(-1, 1, (int(done.sum()), self.D)).astype(np.float32)
        self.t[done] = 0
        return rew, term, trunc, tobs


@pytest.mark.parametrize("D,A,discrete,H1,H2", [(6, 3, True, 64, 64), (11, 5, False, 48, 80), (33, 17, True, 100, 36)])
def test_external_rollout_and_update_vs_oracle(pkg, oracle_mod, D, A, discrete, H1, H2):
    """collect_trajectories over host envs (trajectory.jl:22-78) + GAE + the PPO update, step by step against the oracle: same observations in,
    same actions out, same buffers, same parameters after update! (two iterations, so the second rollout runs on updated weights)"""
    capi = pkg._capi
    E, T = 24, 20
    cfg = _ext_cfg(pkg, D, A, discrete, H1, H2, n_envs=E, n_steps=T, batch_size=E * T // 3, epochs=2, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 5, 0.15); h.set_params(flat); o.set_params(flat)
    envs = _ToyEnvs(E, D, A, discrete, limit=7, seed=3)
    rng = np.random.default_rng(0)
    for it in range(2):
        nz = _noise(rng, h, E * T)
        h.set_noise(nz)
        n_trunc = 0
        for t in range(T):
            obs = envs.observe()
            raw_h, env_h = h.ext_act(obs); raw_o, env_o = o.ext_act(obs, nz[t * E:(t + 1) * E])
            if discrete:
                assert (raw_h == raw_o).all() and (env_h == raw_h).all()              # DiscreteAdapter is the identity, default_adapters.jl:34-38
            else:
                np.testing.assert_allclose(raw_h, raw_o, atol=1e-4, rtol=1e-4)
                assert env_h.min() >= -1.0 and env_h.max() <= 1.0                    # ClampAdapter, default_adapters.jl:4-11
                np.testing.assert_allclose(env_h, np.clip(raw_h, -1, 1), atol=0, rtol=0)
            rew, term, trunc, tobs = envs.step(env_o, cfg.action_start)              # the oracle's actions drive the envs (both paths then see the same obs)
            n_trunc += int(trunc.sum())
            h.ext_record(rew, term, trunc, tobs if trunc.any() else None); o.ext_record(rew, term, trunc, tobs if trunc.any() else None)
            assert h.ext_steps() == t + 1
        assert n_trunc > 0
        last = envs.observe()
        h.ext_finish(last); o.ext_finish(last)
        assert h.ext_steps() == 0
        for which, tol in ((capi.BUF_OBSERVATIONS, 0), (capi.BUF_REWARDS, 0), (capi.BUF_VALUES, 1e-4), (capi.BUF_LOGPROBS, 3e-4), (capi.BUF_BOOTSTRAP, 1e-4),
                           (capi.BUF_LAST_VALUES, 1e-4), (capi.BUF_ADVANTAGES, 1e-3), (capi.BUF_RETURNS, 1e-3)):
            np.testing.assert_allclose(h.buffer(which), o.buffer(which), atol=tol, rtol=tol)
        assert np.array_equal(h.buffer(capi.BUF_FLAGS), o.buffer(capi.BUF_FLAGS))
        for which in (capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))
        perm = np.stack([np.random.default_rng(10 * it + e).permutation(E * T) for e in range(cfg.epochs)]).astype(np.int64)
        h.set_permutation(perm); o.set_permutation(perm)
        sh, so = h.ppo_update(), o.ppo_update()
        assert sh.n_updates == so.n_updates == 6
        assert sh.loss == pytest.approx(so.loss, rel=2e-4, abs=1e-6) and sh.grad_norm == pytest.approx(so.grad_norm, rel=1e-3)
        assert sh.explained_variance == pytest.approx(so.explained_variance, abs=1e-3)
        np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)
        h.set_params(o.get_params())


def test_external_matches_the_fused_cartpole_path(pkg):
    """property: with CartPole's spaces ([4] obs, Discrete(2), hidden [64,64]) the generic path fed by the device CartPole simulator through the
    step-granular env verbs reproduces the fused rollout_kernel + ppo_grad_kernel path: same buffers, same update"""
    capi = pkg._capi
    E, T = 64, 32
    common = dict(n_envs=E, n_steps=T, batch_size=E * T // 2, epochs=2, episode_len=9, seed=7)
    cf = capi.default_config(capi.ENV_CARTPOLE)
    cx = _ext_cfg(pkg, 4, 2, True, 64, 64)
    for k, v in common.items():
        setattr(cf, k, v); setattr(cx, k, v)
    cx.action_start = cf.action_start
    fused, sim, ext = pkg.Handle(cf), pkg.Handle(cf), pkg.Handle(cx)
    flat = _params(fused.P, 3, 0.4)
    assert ext.P == fused.P
    for hh in (fused, sim, ext):
        hh.set_params(flat)
    nz = np.random.default_rng(1).random(E * T)
    fused.env_reset(11); sim.env_reset(11)
    fused.set_noise(nz); fused.collect_rollout()
    ext.set_noise(nz)
    for t in range(T):
        raw, ea = ext.ext_act(sim.env_observe())
        rew, term, trunc, tobs = sim.env_step(ea)
        ext.ext_record(rew, term, trunc, tobs)
    ext.ext_finish(sim.env_observe())
    assert (ext.buffer(capi.BUF_ACTIONS) == fused.buffer(capi.BUF_ACTIONS)).all()
    assert np.array_equal(ext.buffer(capi.BUF_FLAGS), fused.buffer(capi.BUF_FLAGS)) and (ext.buffer(capi.BUF_FLAGS) & 2).any()
    for which, tol in ((capi.BUF_OBSERVATIONS, 1e-6), (capi.BUF_REWARDS, 0), (capi.BUF_VALUES, 2e-5), (capi.BUF_LOGPROBS, 2e-5), (capi.BUF_BOOTSTRAP, 2e-5),
                       (capi.BUF_ADVANTAGES, 2e-4), (capi.BUF_RETURNS, 2e-4)):
        np.testing.assert_allclose(ext.buffer(which), fused.buffer(which), atol=tol, rtol=tol)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        ext.set_buffer(which, fused.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(2)]).astype(np.int64)
    ext.set_permutation(perm); fused.set_permutation(perm)
    se, sf = ext.ppo_update(), fused.ppo_update()
    assert se.n_updates == sf.n_updates == 4 and se.loss == pytest.approx(sf.loss, rel=1e-4)
    np.testing.assert_allclose(ext.get_params(), fused.get_params(), rtol=2e-4, atol=2e-6)


def test_external_call_order_and_unsupported_verbs(pkg):
    """the act / record / finish protocol is checked, the device-env verbs refuse an external handle, and nothing falls back silently"""
    cfg = _ext_cfg(pkg, 5, 2, True, 16, 16, n_envs=3, n_steps=2, batch_size=6)
    h = pkg.Handle(cfg)
    h.set_params(_params(h.P, 0))
    obs = np.zeros((3, 5), np.float32); z = np.zeros(3, np.float32); f = np.zeros(3, np.uint8)
    with pytest.raises(pkg.DrilError):
        h.ext_record(z, f, f)                                                        # record before act
    h.ext_act(obs)
    with pytest.raises(pkg.DrilError):
        h.ext_act(obs)                                                               # act twice
    with pytest.raises(pkg.DrilError):
        h.ext_record(z, f, np.ones(3, np.uint8))                                     # truncated without terminal_obs
    h.ext_record(z, f, f)
    with pytest.raises(pkg.DrilError):
        h.ext_finish(obs)                                                            # only 1 of 2 steps recorded
    h.ext_act(obs); h.ext_record(z, f, f)
    with pytest.raises(pkg.DrilError):
        h.ext_act(obs)                                                               # rollout full
    h.ext_finish(obs)
    for call in (lambda: h.env_reset(0), h.env_observe, h.collect_rollout, lambda: h.train(12), lambda: h.evaluate_agent(1)):
        with pytest.raises(pkg.DrilError) as e:
            call()
        assert e.value.code == pkg._capi.ERR_UNSUPPORTED
    bad = _ext_cfg(pkg, 0, 2, True, 16, 16)
    with pytest.raises(pkg.DrilError):
        pkg.Handle(bad)
    with pytest.raises(pkg.DrilError):
        pkg.Handle(pkg._capi.default_config(pkg._capi.ENV_CARTPOLE)).ext_act(np.zeros((4, 4), np.float32))   # device-env handle


class _PyPointEnv:
    """a host env with the reference's AbstractEnv verbs: 2-D point, 6-dim observation, Box(-1,1)^2 action, reward = -distance to the origin"""

    def __init__(self, pkg, seed):
        self.pkg, self.rng, self.limit = pkg, np.random.default_rng(seed), 25
        self.reset_()

    def observation_space(self):
        return self.pkg.Box(low=[-4.0] * 6, high=[4.0] * 6)

    def action_space(self):
        return self.pkg.Box(low=[-1.0, -1.0], high=[1.0, 1.0])

    def reset_(self):
        self.p = self.rng.uniform(-2, 2, 2).astype(np.float32); self.t = 0

    def observe(self):
        return np.concatenate([self.p, self.p ** 2 / 4, np.sin(self.p)]).astype(np.float32)

    def act_(self, a):
        self.p = np.clip(self.p + 0.3 * np.asarray(a, np.float32), -4, 4); self.t += 1
        return float(-np.linalg.norm(self.p))

    def terminated(self):
        return bool(np.linalg.norm(self.p) < 0.1)

    def truncated(self):
        return self.t >= self.limit


def test_train_on_host_envs_learns(pkg):
    """train!(agent, env, alg, max_steps) over the caller's own Python envs (HostParallelEnv = BroadcastedParallelEnv): the mean reward per step
    improves, statistics have the reference's keys, parameters come back to the agent"""
    env = pkg.HostParallelEnv([_PyPointEnv(pkg, s) for s in range(16)], seed=0)
    alg = pkg.PPO(n_steps=64, batch_size=256, epochs=6, learning_rate=1e-3, ent_coef=0.0)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(48, 32), log_std_init=-0.5), alg, seed=0)
    p0 = pkg.flatten_params(agent.train_state.parameters).copy()
    buf = pkg.RolloutBuffer(alg.n_steps, 16, alg.gae_lambda, alg.gamma)
    pkg.collect_rollout_(buf, agent, alg, env); r0 = float(buf.rewards.mean())
    stats, timer = pkg.train_(agent, env, alg, 25 * 64 * 16)
    assert len(stats["losses"]) == 25 and np.isfinite(stats["losses"]).all() and agent.gradient_updates == 25 * 6 * 4
    pkg.collect_rollout_(buf, agent, alg, env); r1 = float(buf.rewards.mean())
    assert r1 > r0 + 0.3, (r0, r1)
    assert not np.array_equal(p0, pkg.flatten_params(agent.train_state.parameters))
    ev = pkg.evaluate_agent(agent, env, n_eval_episodes=12, deterministic=True)        # evaluation.jl:54-143 on the host envs, mode(d) from the device
    assert set(ev) == {"mean_reward", "std_reward", "mean_length", "std_length"} and 1 <= ev["mean_length"] <= 25 and ev["mean_reward"] < 0
    er, el = pkg.evaluate_agent(agent, env, n_eval_episodes=5, return_stats=False)
    assert len(er) == len(el) == 5


# ---- device envs with hidden_dims the fused kernels are not built for: the same generic kernels, step-granular rollout on the device -----------------
@pytest.mark.parametrize("kind,H1,H2,norm", [(0, 32, 48, 0), (1, 100, 36, 1), (3, 64, 128, 0), (4, 512, 512, 0), (2, 24, 24, 1), (6, 64, 64, 0), (6, 48, 32, 1)])   # 6 = Acrobot-v1 (six observation dims): fused at [64,64] since round 3, generic for other hidden_dims
def test_device_env_with_other_hidden_dims(pkg, oracle_mod, kind, H1, H2, norm):
    """CartPole / Pendulum / MountainCar with hidden_dims outside {[64,64], [128,128], [256,256]}: rollout (env-keyed or injected noise, truncation
    bootstraps, NormalizeWrapperEnv) and the PPO update against the oracle"""
    capi = pkg._capi
    E, T = 40, 24
    c = capi.default_config(kind)
    for k, v in dict(n_envs=E, n_steps=T, episode_len=10, batch_size=E * T // 3, epochs=2, hidden1=H1, hidden2=H2, norm_training=norm, norm_obs=norm, norm_reward=norm).items():
        setattr(c, k, v)
    h, o = pkg.Handle(c), oracle_mod.Oracle(c)
    flat = _params(h.P, 12, 0.08); h.set_params(flat); o.set_params(flat)
    h.env_reset(4); o.env_reset(4)
    nz = np.random.default_rng(0).random(E * T) if h.discrete else np.random.default_rng(0).standard_normal((E * T, h.A)).astype(np.float32)
    h.set_noise(nz); o.set_noise(nz)
    h.collect_rollout(); o.collect_rollout()
    fl = o.buffer(capi.BUF_FLAGS)
    assert (fl & 2).any()
    if h.discrete:
        same = (h.buffer(capi.BUF_ACTIONS).reshape(T, E) == o.buffer(capi.BUF_ACTIONS).reshape(T, E))
        ok = np.cumprod(same, axis=0).astype(bool).all(axis=0)                        # envs whose action sequence never flipped on a rounding tie
        assert ok.mean() >= 0.9
    else:
        ok = np.ones(E, bool)
    for which, tol in ((capi.BUF_OBSERVATIONS, 2e-4), (capi.BUF_VALUES, 3e-4), (capi.BUF_LOGPROBS, 3e-4), (capi.BUF_REWARDS, 3e-4), (capi.BUF_ADVANTAGES, 3e-3), (capi.BUF_RETURNS, 3e-3)):
        a, b = h.buffer(which).reshape(T, E, -1), o.buffer(which).reshape(T, E, -1)
        np.testing.assert_allclose(a[:, ok], b[:, ok], atol=tol, rtol=tol)
    tr = ((fl & 2) != 0).reshape(T, E) & ok[None, :]
    np.testing.assert_allclose(h.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], o.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], atol=3e-4, rtol=3e-4)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(c.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates == 6 and sh.loss == pytest.approx(so.loss, rel=2e-4, abs=1e-6)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)
    # the env-keyed Philox stream (no injected noise), train and evaluate_agent run on the same kernels
    st, fps = h.train(2 * E * T)
    assert len(st) == 2 and np.isfinite([s.loss for s in st]).all()
    ev, er, el = h.evaluate_agent(5, True)
    assert len(er) == 5 and (el >= 1).all() and (el <= 10).all()


# ---- ActorCriticLayer(...; hidden_dims, activation) in full: any depth 1..4 and relu (dril_config v2) on the generic kernels ------------------------------
@pytest.mark.parametrize("kind,hidden,act,norm", [(0, (48,), 0, 0), (1, (40, 24, 56), 0, 1), (0, (32, 32, 16, 8), 1, 0), (1, (64, 64), 1, 0), (3, (96, 20, 33), 1, 0), (4, (128, 128, 128), 0, 0),
                                                  (0, (64, 64), 2, 0), (1, (40, 24, 56), 3, 1), (3, (96, 20), 4, 0), (4, (64, 64), 5, 0), (6, (32, 48), 3, 0),
                                                  (6, (128, 128), 0, 0), (6, (256, 256), 0, 1), (0, (64, 64), 6, 0), (1, (40, 24, 56), 7, 1), (3, (512, 512), 6, 0)])   # sigmoid, elu, leakyrelu, softplus (NNlib); Acrobot with elu
def test_any_depth_and_relu_on_device(pkg, oracle_mod, kind, hidden, act, norm):
    """hidden_dims of length 1, 3, 4 and relu (the reference accepts any, layer_constructors.jl:6-10,55-56; layer_helpers.jl:27-57) on device envs: forward /
    evaluate / loss + gradient / rollout (truncation bootstraps, NormalizeWrapperEnv) / update against the oracle, whose any-depth MLP is pinned to torch
    autograd (tests/test_oracle_crosschecks.py::test_any_depth_and_relu_vs_torch_autograd)"""
    capi = pkg._capi
    E, T = 40, 24
    c = capi.default_config(kind)
    for k, v in dict(n_envs=E, n_steps=T, episode_len=10, batch_size=E * T // 3, epochs=2, norm_training=norm, norm_obs=norm, norm_reward=norm, ent_coef=0.01,
                     n_hidden=len(hidden), activation=act).items():
        setattr(c, k, v)
    for i, w in enumerate(hidden):
        c.hidden[i] = w
    h, o = pkg.Handle(c), oracle_mod.Oracle(c)
    assert h.P == o.P
    flat = _params(h.P, 12, 0.15); h.set_params(flat); o.set_params(flat)
    # layer(obs, ps, st), evaluate_actions, predict_values
    rng = np.random.default_rng(3)
    B = 333
    obs = rng.uniform(-1.5, 1.5, (B, h.D)).astype(np.float32)
    noise = rng.random(B) if h.discrete else rng.standard_normal((B, h.A)).astype(np.float32)
    ah, vh, lh = h.policy_forward(obs, noise); ao, vo, lo = o.policy_forward(obs, noise)
    np.testing.assert_allclose(vh, vo, atol=5e-5, rtol=5e-5)
    same = (ah == ao) if h.discrete else np.ones(B, bool)
    assert same.mean() >= 0.99
    np.testing.assert_allclose(lh[same], lo[same], atol=1e-4, rtol=1e-4)
    ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
    np.testing.assert_allclose(le, lo2, atol=1e-4, rtol=1e-4); np.testing.assert_allclose(ee, eo2, atol=5e-5, rtol=5e-5)
    # (alg::PPO)(...) and its gradient, every layer
    act_b = (rng.integers(0, h.A, B) + c.action_start).astype(np.int32) if h.discrete else rng.normal(0, 1, (B, h.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    lp = (o.evaluate_actions(obs, act_b)[1] + rng.normal(0, 0.1, B)).astype(np.float32)
    lh_, sh_, gh = h.ppo_loss_grad(obs, act_b, adv, ret, lp, ov); lo_, so_, go = o.ppo_loss_grad(obs, act_b, adv, ret, lp, ov)
    assert lh_ == pytest.approx(lo_, rel=1e-4)
    np.testing.assert_allclose(sh_, so_, rtol=3e-4, atol=3e-6)
    assert np.linalg.norm(gh - go) <= 3e-4 * np.linalg.norm(go)
    if act == 1:
        assert (gh == 0).sum() < gh.size // 2                                          # relu: dead units give exact zeros, but not everywhere
    # rollout + update
    h.env_reset(4); o.env_reset(4)
    nz = np.random.default_rng(0).random(E * T) if h.discrete else np.random.default_rng(0).standard_normal((E * T, h.A)).astype(np.float32)
    h.set_noise(nz); o.set_noise(nz)
    h.collect_rollout(); o.collect_rollout()
    fl = o.buffer(capi.BUF_FLAGS)
    assert (fl & 2).any()
    if h.discrete:
        ok = np.cumprod(h.buffer(capi.BUF_ACTIONS).reshape(T, E) == o.buffer(capi.BUF_ACTIONS).reshape(T, E), axis=0).astype(bool).all(axis=0)
        assert ok.mean() >= 0.9
    else:
        ok = np.ones(E, bool)
    for which, tol in ((capi.BUF_OBSERVATIONS, 2e-4), (capi.BUF_VALUES, 3e-4), (capi.BUF_LOGPROBS, 3e-4), (capi.BUF_REWARDS, 3e-4), (capi.BUF_ADVANTAGES, 3e-3), (capi.BUF_RETURNS, 3e-3)):
        a, b = h.buffer(which).reshape(T, E, -1), o.buffer(which).reshape(T, E, -1)
        np.testing.assert_allclose(a[:, ok], b[:, ok], atol=tol, rtol=tol)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(c.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates == 6 and sh.loss == pytest.approx(so.loss, rel=2e-4, abs=1e-6)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)


@pytest.mark.parametrize("bias", [-10.0, -15.0])
def test_softplus_far_negative_preactivations_on_device(pkg, oracle_mod, bias):
    """ADVICE r3 (dril_gemm.hip EPI_MASK_SOFTPLUS): sigmoid(x) from y = softplus(x) as -expm1(-y): the first layer's gradient behind far-negative pre-activations
    against the oracle (pinned to float64 torch autograd for this very case, tests/test_oracle_crosschecks.py) to 1e-3 of its own size"""
    c = pkg._capi.default_config(4)
    for k, v in dict(n_envs=8, n_steps=8, batch_size=32, epochs=1, ent_coef=0.01, n_hidden=2, activation=5).items():
        setattr(c, k, v)
    c.hidden[0] = c.hidden[1] = 48
    h, o = pkg.Handle(c), oracle_mod.Oracle(c)
    flat = _params(h.P, 3, 0.3)
    D, H = h.D, 48
    per_net = [D * H + H + H * H + H + H * O + O for O in (h.A, 1)]
    l1 = []
    for base in (0, per_net[0]):
        flat[base + D * H: base + D * H + H] = bias
        l1.append(slice(base, base + D * H + H))
    h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(5)
    B = 333
    obs = rng.uniform(-1.5, 1.5, (B, D)).astype(np.float32)
    act_b = rng.normal(0, 1, (B, h.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    lp = (o.evaluate_actions(obs, act_b)[1] + rng.normal(0, 0.1, B)).astype(np.float32)
    lh, _, gh = h.ppo_loss_grad(obs, act_b, adv, ret, lp, ov); lo, _, go = o.ppo_loss_grad(obs, act_b, adv, ret, lp, ov)
    assert lh == pytest.approx(lo, rel=1e-4)
    for sl in l1:
        assert np.linalg.norm(go[sl]) > 0 and np.linalg.norm(gh[sl] - go[sl]) <= 1e-3 * np.linalg.norm(go[sl])
    assert np.linalg.norm(gh - go) <= 3e-4 * np.linalg.norm(go)
    h.close()


def test_relu_three_layer_agent_learns_through_the_mirror(pkg):
    """the host mirror carries hidden_dims / activation end to end: ActorCriticLayer(...; hidden_dims = [32, 32, 32], activation = relu) trains on CartPole"""
    env = pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=200), 256, seed=1)
    alg = pkg.PPO(n_steps=64, batch_size=2048, epochs=4, learning_rate=1e-3)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(32, 32, 32), activation="relu")
    agent = pkg.Agent(layer, alg, seed=0)
    assert layer.parameterlength() == pkg.flatten_params(agent.train_state.parameters).size
    assert sorted(agent.train_state.parameters["actor_head"]) == ["layer_1", "layer_2", "layer_3", "layer_4"]
    base = pkg.evaluate_agent(agent, env, n_eval_episodes=64, deterministic=True)
    stats, _ = pkg.train_(agent, env, alg, 30 * 64 * 256)
    assert np.isfinite(stats["losses"]).all()
    trained = pkg.evaluate_agent(agent, env, n_eval_episodes=64, deterministic=True)
    print(f"[relu 3-layer] CartPole mean episode reward {base['mean_reward']:.1f} -> {trained['mean_reward']:.1f}")
    assert trained["mean_reward"] > max(2 * base["mean_reward"], 80.0)
    with pytest.raises(ValueError):
        pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(8, 8, 8, 8, 8))
    with pytest.raises(ValueError):
        pkg.ActorCriticLayer(env.observation_space(), env.action_space(), activation="mish")      # not one of the eight


def test_forced_generic_equals_fused(pkg, monkeypatch):
    """DRIL_FORCE_GENERIC=1 runs a [64,64] CartPole handle on the generic kernels: same rollout (same env-keyed noise) and update as the fused kernels"""
    capi = pkg._capi
    E, T = 64, 16
    c = capi.default_config(capi.ENV_CARTPOLE)
    c.n_envs, c.n_steps, c.batch_size, c.epochs, c.episode_len, c.seed = E, T, E * T // 2, 2, 9, 3
    fused = pkg.Handle(c)
    monkeypatch.setenv("DRIL_FORCE_GENERIC", "1")
    gen = pkg.Handle(c)
    monkeypatch.delenv("DRIL_FORCE_GENERIC")
    flat = _params(fused.P, 3, 0.4)
    for hh in (fused, gen):
        hh.set_params(flat); hh.env_reset(11); hh.collect_rollout()
    assert (gen.buffer(capi.BUF_ACTIONS) == fused.buffer(capi.BUF_ACTIONS)).mean() > 0.99
    same = (gen.buffer(capi.BUF_ACTIONS).reshape(T, E) == fused.buffer(capi.BUF_ACTIONS).reshape(T, E)).all(axis=0)
    for which, tol in ((capi.BUF_OBSERVATIONS, 1e-6), (capi.BUF_VALUES, 3e-5), (capi.BUF_LOGPROBS, 3e-5), (capi.BUF_ADVANTAGES, 3e-4)):
        np.testing.assert_allclose(gen.buffer(which).reshape(T, E, -1)[:, same], fused.buffer(which).reshape(T, E, -1)[:, same], atol=tol, rtol=tol)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        gen.set_buffer(which, fused.buffer(which))
    sg, sf = gen.ppo_update(), fused.ppo_update()                                         # device-generated minibatch order: the same keyed bijection on both paths
    assert sg.n_updates == sf.n_updates == 4 and sg.loss == pytest.approx(sf.loss, rel=1e-4)
    np.testing.assert_allclose(gen.get_params(), fused.get_params(), rtol=2e-4, atol=2e-6)


# ---- callbacks (test/test_callbacks.jl): locals keys, early stops at every hook, on_step through the step-granular path ------------------------
def test_callbacks_locals_and_early_stops(pkg):
    def setup():
        env = pkg.NormalizeWrapperEnv(pkg.MonitorWrapperEnv(pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=50), 8, seed=1)), gamma=0.99)
        alg = pkg.PPO(ent_coef=0.1, n_steps=64, batch_size=64, epochs=2)
        return pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=0), env, alg

    seen = {}

    class CheckLocals:
        def on_training_start(self, loc):
            seen["start"] = set(loc); return True

        def on_rollout_start(self, loc):
            seen.setdefault("rollout", set(loc)); return True

    agent, env, alg = setup()
    out = pkg.train_(agent, env, alg, 2 * 64 * 8, callbacks=[CheckLocals()])
    assert out is not None and agent.steps_taken == 1024
    assert {"agent", "env", "alg", "iterations", "total_steps", "max_steps", "n_steps", "n_envs", "roll_buffer", "total_fps", "callbacks"} <= seen["start"]   # :25-27
    assert {"agent", "env", "alg", "iterations", "total_steps", "max_steps", "i", "learning_rate"} <= seen["rollout"]                                      # :36-39

    class StopAt:
        def __init__(self, name):
            setattr(self, name, lambda loc: False)

    for name in ("on_training_start", "on_rollout_start"):                       # test_callbacks.jl:70-88
        agent, env, alg = setup()
        assert pkg.train_(agent, env, alg, 3000, callbacks=[StopAt(name)]) is None and agent.steps_taken == 0

    class OnStepStopEarly:                                                        # :91-99: the first rollout (64 x 8 = 512 steps) completes, the second stops at its first step
        def __init__(self):
            self.calls = 0

        def on_step(self, loc):
            self.calls += 1
            return loc["agent"].steps_taken < 500

    agent, env, alg = setup()
    cb = OnStepStopEarly()
    assert pkg.train_(agent, env, alg, 3000, callbacks=[cb]) is None
    assert agent.steps_taken == 512 and cb.calls == 65 and agent.gradient_updates > 0


def test_stepwise_rollout_fills_the_same_buffer_as_gae_oracle(pkg, oracle_mod):
    """the step-granular rollout used for on_step callbacks leaves a consistent buffer: advantages / returns equal the oracle's GAE of the stored
    rewards / values / flags / bootstraps, and logprobs / values equal evaluate_actions of the stored observations and actions"""
    capi = pkg._capi
    env = pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=9), 12, seed=3)
    alg = pkg.PPO(n_steps=20, batch_size=60, epochs=1)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=0)
    h = env.bind(alg, agent.layer); h.set_params(pkg.flatten_params(agent.train_state.parameters))
    from dril_jl_amd.host import _stepwise_rollout
    assert _stepwise_rollout(h, env, lambda: True) > 0
    B = {n: h.buffer(getattr(capi, "BUF_" + n)) for n in ("OBSERVATIONS", "ACTIONS", "REWARDS", "VALUES", "LOGPROBS", "FLAGS", "BOOTSTRAP", "LAST_VALUES", "ADVANTAGES", "RETURNS")}
    assert (B["FLAGS"] & 2).any()
    v, lp, _ = h.evaluate_actions(B["OBSERVATIONS"], B["ACTIONS"])
    np.testing.assert_allclose(v, B["VALUES"], atol=1e-5, rtol=1e-5); np.testing.assert_allclose(lp, B["LOGPROBS"], atol=1e-5, rtol=1e-5)
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    adv = np.zeros(240, np.float32); ret = np.zeros(240, np.float32)
    assert oracle_mod.lib().orc_gae(12, 20, alg.gamma, alg.gae_lambda, p(B["REWARDS"]), p(B["VALUES"]), p(B["FLAGS"]), p(B["BOOTSTRAP"]), p(B["LAST_VALUES"]), p(adv), p(ret)) == 0
    np.testing.assert_allclose(B["ADVANTAGES"], adv, atol=1e-4, rtol=1e-5); np.testing.assert_allclose(B["RETURNS"], ret, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("E,T,B,kw", [(1, 3, 10, {}), (5, 4, 7, {}), (3, 2, 6, dict(has_target_kl=1, target_kl=1e-6)), (2, 2, 4, dict(epochs=0)),
                                      (4, 5, 20, dict(has_max_grad_norm=0, normalize_advantage=0))])
def test_external_edge_sizes_and_control_flow(pkg, oracle_mod, E, T, B, kw):
    """edge cases of the update loop on the generic path (ppo.jl:188-254): one env / one partial minibatch larger than the buffer, ragged last minibatch,
    target_kl stopping both loops before the first apply, zero epochs, no gradient clipping / no advantage normalisation"""
    capi = pkg._capi
    cfg = _ext_cfg(pkg, 5, 3, True, 16, 24, n_envs=E, n_steps=T, batch_size=B, epochs=kw.pop("epochs", 2), **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 2, 0.3); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(E * 10 + T)
    N = E * T
    bufs = {capi.BUF_OBSERVATIONS: rng.standard_normal((N, 5)).astype(np.float32), capi.BUF_ACTIONS: (rng.integers(0, 3, N) + cfg.action_start).astype(np.int32),
            capi.BUF_ADVANTAGES: rng.standard_normal(N).astype(np.float32), capi.BUF_RETURNS: rng.standard_normal(N).astype(np.float32),
            capi.BUF_VALUES: rng.standard_normal(N).astype(np.float32)}
    bufs[capi.BUF_LOGPROBS] = (o.evaluate_actions(bufs[capi.BUF_OBSERVATIONS], bufs[capi.BUF_ACTIONS])[1] + rng.normal(0, 0.2, N)).astype(np.float32)
    for which, arr in bufs.items():
        h.set_buffer(which, arr); o.set_buffer(which, arr)
    if cfg.epochs:
        perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(cfg.epochs)]).astype(np.int64)
        h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert (sh.n_updates, sh.early_stopped, sh.nan_or_inf) == (so.n_updates, so.early_stopped, so.nan_or_inf)
    if cfg.has_target_kl:
        assert sh.early_stopped == 1 and sh.n_updates <= 1
    if so.n_updates:
        assert sh.loss == pytest.approx(so.loss, rel=2e-4, abs=1e-6) and sh.grad_norm == pytest.approx(so.grad_norm, rel=1e-3)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)
    if cfg.epochs == 0:
        assert np.array_equal(h.get_params(), flat)

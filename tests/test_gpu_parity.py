"""GPU (-m gpu): the HIP path through the C ABI vs the CPU oracle on the same seeded inputs, the committed golden
vectors, and size-independent properties at larger sizes.  Tolerances are fp32 (reference arithmetic is Float32):
  GAE returns            atol 1e-4   (the reference's own test tolerance, test/test_gae.jl:66,70)
  forward values/logp    atol/rtol 1e-5 scale (test/test_buffers.jl:166-214 uses 1e-5)
  PPO loss               rel 1e-4    (BASELINE.json north_star)
"""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _cfg(pkg, kind, **kw):
    c = pkg._capi.default_config(kind)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _params(P, seed, scale=0.3):
    return (np.random.default_rng(seed).standard_normal(P) * scale).astype(np.float32)


def _flip_report(o, cfg, obs, u, a_dev, a_orc, where=""):
    """Discrete actions come from an inverse-CDF draw (categorical.jl:47-52: findfirst(cumsum(p) .>= u)); device and oracle compute p in fp32 with
    different instruction sequences, so an action may differ only where u sits within fp32 rounding of a CDF edge.  Returns the number of
    flips and asserts that EVERY flipped sample has |u - edge| <= 1e-6 (the oracle's probabilities give the edges); prints the count."""
    flip = np.flatnonzero(np.asarray(a_dev).reshape(-1) != np.asarray(a_orc).reshape(-1))
    if flip.size == 0:
        return 0
    ob = np.ascontiguousarray(np.asarray(obs, np.float32).reshape(-1, o.D)[flip])
    probs = np.stack([np.exp(o.evaluate_actions(ob, np.full(flip.size, cfg.action_start + a, np.int32))[1].astype(np.float64)) for a in range(o.A)], axis=1)
    edges = np.cumsum(probs, axis=1)[:, :-1]
    margin = np.abs(edges - np.asarray(u, np.float64).reshape(-1)[flip][:, None]).min(axis=1)
    print(f"[flips]{where} {flip.size} of {np.asarray(a_orc).size} actions differ; max |u - CDF edge| = {margin.max():.2e}")
    assert margin.max() <= 1e-6, f"{where}: an action differs with u {margin.max():.3e} away from the nearest CDF edge"
    return int(flip.size)


@pytest.fixture(scope="module")
def lib(hip):
    return hip


@pytest.mark.parametrize("case", json.loads((G / "gae.json").read_text()), ids=lambda c: c["name"])
def test_gae_golden_on_device(pkg, lib, case):
    """the reference's analytic GAE tests (test/test_gae.jl, test/test_buffers.jl:60-115) through dril_gae"""
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    E, T = case["n_envs"], case["n_steps"]
    f32 = lambda k: np.asarray(case[k], np.float32)
    r, v, b, lv = f32("rewards"), f32("values"), f32("bootstrap"), f32("last_values")
    fl = np.asarray(case["flags"], np.uint8)
    adv = np.zeros(E * T, np.float32); ret = np.zeros(E * T, np.float32)
    assert lib.dril_gae(E, T, case["gamma"], case["gae_lambda"], p(r), p(v), p(fl), p(b), p(lv), p(adv), p(ret)) == 0
    np.testing.assert_allclose(adv, case["expected_advantages"], atol=case["atol"], rtol=0)
    np.testing.assert_allclose(ret, case["expected_returns"], atol=case["atol"], rtol=0)


def test_gae_random_vs_oracle(pkg, lib, oracle_mod):
    """gae_scan_kernel cuts the time axis into chunks of 32 rows, one workgroup per (chunk, 256 envs), the carry between chunks travelling through L2: shapes that are
    ragged in both directions (a last chunk of 1 ... 31 rows, a last workgroup with a few live lanes), one chunk only, hundreds of chunks with few envs (the reference's
    own scale: a chain of T / 32 dependent hops), and — gamma = lambda = 1 with no trajectory end anywhere — carries that run through EVERY chunk undamped.
    The chain inside a chunk is the serial kernel's, so the agreement with the serial oracle is at rounding level"""
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(0)
    for E, T, gamma, lam, cuts in ((1, 1, 0.99, 0.95, True), (3, 7, 0.99, 0.95, True), (257, 33, 0.99, 0.95, True), (1000, 129, 0.99, 0.95, True), (4, 2048, 0.99, 0.95, True),
                                   (513, 32, 0.99, 0.95, True), (70, 64, 0.99, 0.95, True), (300, 1000, 1.0, 1.0, False), (5, 4096, 1.0, 1.0, False), (4096, 97, 0.9, 0.8, True)):
        r = rng.standard_normal(E * T).astype(np.float32); v = rng.standard_normal(E * T).astype(np.float32)
        fl = (rng.choice([0, 0, 0, 0, 1, 2, 3], E * T) if cuts else np.zeros(E * T)).astype(np.uint8)
        if not cuts:
            r *= np.float32(0.01)                                   # undamped sums over thousands of steps stay O(1)
        b = rng.standard_normal(E * T).astype(np.float32); lv = rng.standard_normal(E).astype(np.float32)
        out = [np.full(E * T, np.nan, np.float32) for _ in range(4)]
        assert lib.dril_gae(E, T, gamma, lam, p(r), p(v), p(fl), p(b), p(lv), p(out[0]), p(out[1])) == 0
        assert oracle_mod.lib().orc_gae(E, T, gamma, lam, p(r), p(v), p(fl), p(b), p(lv), p(out[2]), p(out[3])) == 0
        np.testing.assert_allclose(out[0], out[2], atol=2e-6 * max(1.0, float(np.abs(out[2]).max())), rtol=1e-5, err_msg=f"advantages E {E} T {T}")
        np.testing.assert_allclose(out[1], out[3], atol=2e-6 * max(1.0, float(np.abs(out[3]).max())), rtol=1e-5, err_msg=f"returns E {E} T {T}")


def test_gae_twice_on_one_handle_uses_fresh_carries(pkg, oracle_mod):
    """the carry words of gae_scan_kernel are validated by a per-launch tag: a second rollout on the same handle (other rewards) must not read the first one's carries"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=300, n_steps=200, episode_len=500, batch_size=300, fixed_length_episodes=1)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 1, 0.3); h.set_params(flat); o.set_params(flat)
    h.env_reset(2); o.env_reset(2); h.collect_rollout(); o.collect_rollout()
    for k in range(3):
        rw = np.random.default_rng(k).standard_normal(300 * 200).astype(np.float32)
        h.set_buffer(capi.BUF_REWARDS, rw); o.set_buffer(capi.BUF_REWARDS, rw)
        h.set_buffer(capi.BUF_VALUES, o.buffer(capi.BUF_VALUES))
        h.compute_gae(); o.compute_gae()
        np.testing.assert_allclose(h.buffer(capi.BUF_ADVANTAGES), o.buffer(capi.BUF_ADVANTAGES), atol=1e-4, rtol=1e-5)   # (the two rollouts' last values agree to 1e-5)
        np.testing.assert_allclose(h.buffer(capi.BUF_RETURNS), o.buffer(capi.BUF_RETURNS), atol=1e-4, rtol=1e-5)
    h.close()


@pytest.mark.parametrize("kind", [0, 1, 2, 3, 4, 6, 7])           # 2 = ScalingWrapperEnv(Pendulum); 3 / 4 = MountainCar-v0 / MountainCarContinuous-v0; 6 = Acrobot-v1; 7 = ScalingWrapperEnv(MountainCarContinuous)
def test_env_verbs_match_oracle(pkg, oracle_mod, kind):
    """reset!/observe/act! with auto-reset and terminal_observation (multithreadedParallelEnv.jl:12-74)"""
    cfg = _cfg(pkg, kind, n_envs=300, n_steps=4, episode_len=7, batch_size=4)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    h.env_reset(123); o.env_reset(123)
    np.testing.assert_array_equal(h.env_get_state()[0], o.env_get_state()[0])       # Philox reset noise is bit-exact
    rng = np.random.default_rng(1)
    saw_trunc = False
    for step in range(20):
        np.testing.assert_allclose(h.env_observe(), o.env_observe(), atol=2e-6 if kind != 6 else 2e-5, rtol=2e-6 if kind != 6 else 2e-5)
        a = (rng.integers(0, h.A, cfg.n_envs) + cfg.action_start).astype(np.int32) if h.discrete else rng.uniform(-3, 3, (cfg.n_envs, 1)).astype(np.float32)
        rh, th, uh, oh = h.env_step(a); ro, to, uo, oo = o.env_step(a)
        np.testing.assert_array_equal(th, to); np.testing.assert_array_equal(uh, uo)
        np.testing.assert_allclose(rh, ro, atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(oh[uh], oo[uo], atol=1e-5 if kind != 6 else 1e-4, rtol=1e-5 if kind != 6 else 1e-4)              # terminal_observation only where truncated
        sh, ch = h.env_get_state(); so, co = o.env_get_state()
        np.testing.assert_array_equal(ch, co)
        np.testing.assert_allclose(sh, so, atol=1e-5 if kind != 6 else 1e-4, rtol=1e-5 if kind != 6 else 1e-4)        # Acrobot: one RK4 step = 12 sin/cos per env step
        o.env_set_state(sh, ch)                                                       # teacher forcing: no drift accumulation
        saw_trunc |= bool(uh.any())
    assert saw_trunc


@pytest.mark.parametrize("kind,B", [(0, 1), (0, 31), (0, 32), (0, 1000), (1, 65), (1, 4096), (3, 333), (4, 97), (6, 1), (6, 777)])   # 6 = Acrobot-v1: six observation dims = four first-layer k-steps
def test_policy_forward_evaluate_predict(pkg, oracle_mod, kind, B):
    """layer(obs,ps,st), evaluate_actions, predict_values (layer_forward.jl:3-39, layer_methods.jl:28-61)"""
    cfg = _cfg(pkg, kind, n_envs=2, n_steps=2, batch_size=2)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 5 + B, 0.5); h.set_params(flat); o.set_params(flat)
    np.testing.assert_array_equal(h.get_params(), flat)
    rng = np.random.default_rng(B)
    obs = rng.uniform(-2, 2, (B, h.D)).astype(np.float32)
    noise = rng.random(B) if h.discrete else rng.standard_normal((B, h.A)).astype(np.float32)
    ah, vh, lh = h.policy_forward(obs, noise); ao, vo, lo = o.policy_forward(obs, noise)
    np.testing.assert_allclose(vh, vo, atol=2e-5, rtol=2e-5)
    if h.discrete:
        same = ah == ao
        _flip_report(o, cfg, obs, noise, ah, ao, f" policy_forward kind={kind} B={B}:")     # a flip needs u within 1e-6 of a CDF edge; count printed
        assert same.mean() >= 0.995
        np.testing.assert_allclose(lh[same], lo[same], atol=2e-5, rtol=2e-5)
        assert set(np.unique(ah)) <= {cfg.action_start + i for i in range(h.A)}
    else:
        np.testing.assert_allclose(ah, ao, atol=2e-5, rtol=2e-5)
        np.testing.assert_allclose(lh, lo, atol=1e-4, rtol=1e-4)
    ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
    np.testing.assert_allclose(ve, vo2, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(le, lo2, atol=1e-4 if kind else 2e-5, rtol=1e-4)
    np.testing.assert_allclose(ee, eo2, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(h.predict_values(obs), vo, atol=2e-5, rtol=2e-5)
    # predict_actions(layer, obs, ps, st; deterministic) (layer_methods.jl:3-26): mode(d) = argmax / mean, rand(d) = the sampled action above
    dh, do = h.predict_actions(obs, True), o.predict_actions(obs, True)
    if h.discrete:
        assert (dh == do).mean() >= 0.99 and np.array_equal(h.predict_actions(obs, False, noise), ah)
    else:
        np.testing.assert_allclose(dh, do, atol=2e-5, rtol=2e-5); np.testing.assert_allclose(h.predict_actions(obs, False, noise), ah, atol=0, rtol=0)
    # forward vs evaluate self-consistency, test/test_policies.jl:127-145
    np.testing.assert_allclose(h.evaluate_actions(obs, ah)[1], lh, atol=1e-5, rtol=1e-5)


def _rollout_pair(pkg, oracle_mod, kind, E, T, L, seed, fixed=False):
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=L, batch_size=max(2, (E * T) // 4), epochs=2, fixed_length_episodes=int(fixed))
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, seed, 0.4); h.set_params(flat); o.set_params(flat)
    h.env_reset(seed); o.env_reset(seed)
    return cfg, h, o


@pytest.mark.parametrize("kind,E,T,L,fixed", [(0, 64, 48, 500, False), (0, 100, 40, 9, True), (1, 33, 50, 12, False), (0, 1, 5, 3, False), (2, 40, 30, 12, False), (3, 70, 40, 15, False), (4, 33, 24, 10, False), (6, 50, 30, 9, False)])
def test_collect_rollout_matches_oracle(pkg, oracle_mod, kind, E, T, L, fixed):
    """collect_rollout! (rollout_buffer.jl:46-90): every buffer field vs the trajectory-based oracle, with injected
    sampling noise and with the shared Philox stream; includes terminations, mid-rollout truncations with
    V(terminal_observation) bootstraps and rollout-limited tails (trajectory.jl:52-74)."""
    capi = pkg._capi
    for inject in (True, False):
        cfg, h, o = _rollout_pair(pkg, oracle_mod, kind, E, T, L, seed=17 + E, fixed=fixed)
        if inject:
            rng = np.random.default_rng(E)
            noise = rng.random(E * T) if h.discrete else rng.standard_normal((E * T, h.A)).astype(np.float32)
            h.set_noise(noise); o.set_noise(noise)
        for rollout in range(2):                       # the env is NOT reset between rollouts (trajectory.jl:26)
            fps = h.collect_rollout(); o.collect_rollout()
            assert fps > 0
            ah, ao = h.buffer(capi.BUF_ACTIONS).reshape(T, E, -1), o.buffer(capi.BUF_ACTIONS).reshape(T, E, -1)
            if h.discrete:
                ok = np.cumprod((ah == ao).all(axis=2), axis=0).astype(bool)     # env matches up to its first action flip
                assert ok.all(axis=0).mean() >= 0.98
                if inject and rollout == 0:                                       # (injected noise covers the first rollout) the FIRST difference of an env must be a CDF-edge flip (later ones follow from diverged states)
                    first = ok.copy(); first[1:] = ok[:-1]; first[0] = True       # steps whose inputs still agree
                    fo_obs = o.buffer(capi.BUF_OBSERVATIONS).reshape(T, E, -1)
                    sel = first & ~ok
                    if sel.any():
                        _flip_report(o, cfg, fo_obs[sel], noise.reshape(T, E)[sel], ah[sel], ao[sel], f" rollout kind={kind} E={E}:")
            else:
                ok = np.ones((T, E), bool)
            for which, tol in ((capi.BUF_OBSERVATIONS, 2e-5), (capi.BUF_VALUES, 5e-5), (capi.BUF_LOGPROBS, 1e-4), (capi.BUF_REWARDS, 1e-4),
                               (capi.BUF_ADVANTAGES, 1e-3), (capi.BUF_RETURNS, 1e-3)):
                a, b = h.buffer(which), o.buffer(which)
                a, b = a.reshape(T, E, -1), b.reshape(T, E, -1)
                full = ok.all(axis=0)                   # GAE looks ahead, so compare whole envs that never diverged
                np.testing.assert_allclose(a[:, full], b[:, full], atol=tol, rtol=tol)
            fh, fo = h.buffer(capi.BUF_FLAGS).reshape(T, E), o.buffer(capi.BUF_FLAGS).reshape(T, E)
            full = ok.all(axis=0)
            np.testing.assert_array_equal(fh[:, full], fo[:, full])
            tr = (fo & 2).astype(bool) & full[None, :]
            np.testing.assert_allclose(h.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], o.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], atol=5e-5, rtol=5e-5)
            live = full & (fo[T - 1] == 0)                # V(new_obs) is only consumed for rollout-limited tails (trajectory.jl:65-70)
            np.testing.assert_allclose(h.buffer(capi.BUF_LAST_VALUES)[live], o.buffer(capi.BUF_LAST_VALUES)[live], atol=5e-5, rtol=5e-5)
            np.testing.assert_allclose(h.buffer(capi.BUF_RETURNS), h.buffer(capi.BUF_ADVANTAGES) + h.buffer(capi.BUF_VALUES), atol=1e-5)
            if not inject:
                break
            # keep the two simulators in lock-step for the second rollout
            st, sc = h.env_get_state(); o.env_set_state(st, sc)
        if L < T:
            assert (fh & 2).any()
    # stored logprobs/values == evaluate_actions recomputed on device (test/test_buffers.jl:166-214)
    val, lp, ent = h.evaluate_actions(h.buffer(capi.BUF_OBSERVATIONS), h.buffer(capi.BUF_ACTIONS))
    np.testing.assert_allclose(val, h.buffer(capi.BUF_VALUES), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(lp, h.buffer(capi.BUF_LOGPROBS), atol=1e-5, rtol=1e-5)


def _batch(oracle, cfg, B, seed):
    rng = np.random.default_rng(seed)
    obs = rng.uniform(-1, 1, (B, oracle.D)).astype(np.float32)
    act = (rng.integers(0, oracle.A, B) + cfg.action_start).astype(np.int32) if oracle.discrete else rng.normal(0, 1, (B, oracle.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    _, lp, _ = oracle.evaluate_actions(obs, act)
    return obs, act, adv, ret, (lp + rng.normal(0, 0.1, B)).astype(np.float32), ov


@pytest.mark.parametrize("kind,B,variant", [(0, 64, "default"), (0, 33, "default"), (0, 4096, "ent_vfclip"), (0, 65536, "default"),
                                             (1, 64, "default"), (1, 1000, "ent_vfclip"), (0, 200, "no_norm"), (3, 500, "ent_vfclip"), (4, 129, "default"),
                                             (6, 64, "default"), (6, 1001, "ent_vfclip"), (6, 131072 + 5, "default")])   # Acrobot (D = 6): three-quad records; the exact-f32 kernel below 65 536 samples, ppo_grad_pair_kernel (dW1 on the matrix cores) above
def test_ppo_loss_and_gradient(pkg, oracle_mod, kind, B, variant):
    """(alg::PPO)(...) ppo.jl:365-407 + gradient: loss within 1e-4 rel (north_star), gradient within fp32 noise"""
    kw = dict(n_envs=2, n_steps=2, batch_size=2)
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    if variant == "no_norm":
        kw.update(normalize_advantage=0)
    cfg = _cfg(pkg, kind, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    for seed in range(2):
        flat = _params(h.P, 40 + seed, 0.25); h.set_params(flat); o.set_params(flat)
        batch = _batch(o, cfg, B, seed)
        lh, sh, gh = h.ppo_loss_grad(*batch); lo, so, go = o.ppo_loss_grad(*batch)
        assert lh == pytest.approx(lo, rel=1e-4)
        np.testing.assert_allclose(sh, so, rtol=2e-4, atol=2e-6)
        assert np.linalg.norm(gh - go) <= 2e-4 * np.linalg.norm(go)
        np.testing.assert_allclose(gh, go, rtol=5e-3, atol=1e-5 * np.abs(go).max())
        lh2, _, gh2 = h.ppo_loss_grad(*batch)
        assert lh2 == lh and np.array_equal(gh, gh2)                     # slab reduction is bitwise reproducible


def test_apply_gradients_clip_adam_nan(pkg, oracle_mod):
    """nested_norm / nested_scale! / Adam (ppo.jl:216-239) incl. the NaN assert (:213-214)"""
    cfg = _cfg(pkg, 0, n_envs=2, n_steps=2, batch_size=2)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 9, 0.2); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(3)
    for step in range(7):
        g = (rng.standard_normal(h.P) * (0.02 if step % 2 else 0.001)).astype(np.float32)
        nh, no = h.apply_gradients(g), o.apply_gradients(g)
        assert nh == pytest.approx(no, rel=1e-5)
        np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=1e-5, atol=2e-7)
    before = h.get_params()
    g = np.zeros(h.P, np.float32); g[17] = np.inf
    with pytest.raises(pkg.DrilError) as e:
        h.apply_gradients(g)
    assert e.value.code == pkg._capi.ERR_NAN_IN_GRADS
    np.testing.assert_array_equal(h.get_params(), before)               # a poisoned step is never applied
    h.reset_optimizer(); o.reset_optimizer()
    g = (rng.standard_normal(h.P) * 0.01).astype(np.float32)
    o.set_params(before); h.apply_gradients(g); o.apply_gradients(g)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=1e-5, atol=2e-7)


@pytest.mark.parametrize("kind,E,T,B,kw", [
    (0, 16, 24, 64, {}),
    (0, 10, 13, 16, {}),                                   # N = 130: ragged last minibatch of 2 kept (MLUtils partial=true)
    (1, 12, 20, 60, {"ent_coef": 0.01}),
    (2, 12, 20, 60, {}),                                   # ScalingWrapperEnv(Pendulum): the update shares the Pendulum kernels
    (3, 12, 20, 60, {"ent_coef": 0.01}),                   # MountainCar-v0: Categorical over 3 actions, D = 2
    (4, 12, 20, 48, {}),                                   # MountainCarContinuous-v0
    (7, 12, 20, 48, {"ent_coef": 0.01}),                   # ScalingWrapperEnv(MountainCarContinuous): the update shares kind 4's kernels
    (7, 8, 16, 20, {}),                                    # ... on ppo_update_small_kernel (batch_size <= 64)
    (6, 12, 20, 60, {"ent_coef": 0.01}),                   # Acrobot-v1 on the fused kernels (D = 6)
    (6, 64, 32, 512, {}),
    (0, 16, 24, 96, {"has_target_kl": 1, "target_kl": 0.002}),
    (0, 16, 24, 384, {"has_clip_range_vf": 1, "clip_range_vf": 0.2}),
])
def test_ppo_update_matches_oracle(pkg, oracle_mod, kind, E, T, B, kw):
    """the epoch x minibatch loop (ppo.jl:188-264) on identical buffers and an injected DataLoader order:
    parameters after the update, per-iteration stats, KL early stop"""
    capi = pkg._capi
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, batch_size=B, epochs=3, episode_len=11, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 77, 0.3); h.set_params(flat); o.set_params(flat)
    o.env_reset(5); o.collect_rollout()
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    N = E * T
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(cfg.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert (sh.n_updates, sh.early_stopped) == (so.n_updates, so.early_stopped)
    if "has_target_kl" in kw:
        assert sh.early_stopped and sh.n_updates < cfg.epochs * -(-N // B)
    for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first"):
        assert getattr(sh, f) == pytest.approx(getattr(so, f), rel=5e-4, abs=2e-6), f
    assert sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    if B == 16:                                              # appendix item 5: a 1-sample minibatch -> NaN assert, like the reference
        cfg1 = _cfg(pkg, 0, n_envs=5, n_steps=13, batch_size=16, epochs=1, episode_len=11)
        h1, o1 = pkg.Handle(cfg1), oracle_mod.Oracle(cfg1)
        h1.set_params(flat); o1.set_params(flat); o1.env_reset(5); o1.collect_rollout()
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h1.set_buffer(which, o1.buffer(which))
        p1 = np.arange(65, dtype=np.int64)[None, :]
        h1.set_permutation(p1)
        with pytest.raises(pkg.DrilError) as e:
            h1.ppo_update()
        assert e.value.code == capi.ERR_NAN_IN_GRADS


@pytest.mark.parametrize("E,T,B", [(24, 37, 100), (100, 77, 1111), (64, 32, 256), (3, 683, 100), (4096, 32, 16384)])
def test_epoch_moment_table_equals_the_per_step_moments(pkg, oracle_mod, E, T, B):
    """the advantage moments of all minibatches of an epoch from ONE sequential pass (epoch_moments_kernel: inverse bijection with cycle walking, position / B by a
    reciprocal, replicated LDS bins) against the per-step route (the same order injected as an index array => adv_moments_kernel gathers per minibatch): buffer sizes
    that are not powers of two, a partial last minibatch, 2 049 samples (12 bits, 2 047 rejected positions), a chip-filling size.  Same f32 kernels on both routes, the moments are f64 sums: the two
    updates agree to the last bits of f32"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, batch_size=B, epochs=2, episode_len=11, seed=9)
    a, b, o = pkg.Handle(cfg), pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(a.P, 5, 0.3)
    o.set_params(flat); o.env_reset(5); o.collect_rollout()
    for h in (a, b):
        h.set_params(flat)
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))
    N, L = E * T, oracle_mod.lib()
    if N <= 10000:
        perm = np.asarray([[L.orc_perm_index(p, N, L.orc_perm_key(cfg.seed, 0, ep)) for p in range(N)] for ep in range(cfg.epochs)], np.int64)
    else:                                                     # the bijection vectorised (dril_device.h mix_bij32; bits = 17 here, no rejected positions when N is a power of two)
        assert N & (N - 1) == 0
        bits = N.bit_length() - 1; mask = np.uint64(N - 1); sh = bits // 2; rows = []
        for ep in range(cfg.epochs):
            key = L.orc_perm_key(cfg.seed, 0, ep); x = np.arange(N, dtype=np.uint64)
            for r in range(4):
                x ^= np.uint64((key >> (13 * r)) & (N - 1))
                x = (x * np.uint64(0x7F4A7C15) + np.uint64(0xD192ED03)) & mask; x ^= x >> np.uint64(sh)
                x = (x * np.uint64(0x1CE4E5B9)) & mask; x ^= x >> np.uint64(sh + 1 if sh + 1 < bits else sh)
            rows.append(x.astype(np.int64))
        perm = np.stack(rows)
        assert [L.orc_perm_index(p, N, L.orc_perm_key(cfg.seed, 0, 1)) for p in (0, 1, N - 1)] == [int(perm[1][p]) for p in (0, 1, N - 1)]
    assert all(np.array_equal(np.sort(row), np.arange(N)) for row in perm)
    b.set_permutation(perm)
    sa, sb = a.ppo_update(), b.ppo_update()
    assert sa.n_updates == sb.n_updates == cfg.epochs * -(-N // B)
    for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm"):
        assert getattr(sa, f) == pytest.approx(getattr(sb, f), rel=2e-6, abs=1e-8), f
    np.testing.assert_allclose(a.get_params(), b.get_params(), rtol=2e-6, atol=1e-8)
    assert not np.array_equal(a.get_params(), flat)


def test_device_permutation_matches_oracle_and_train_end_to_end(pkg, oracle_mod):
    """train! (ppo.jl:100-325) for 3 iterations with the device-generated DataLoader order and Philox sampling:
    the oracle draws the same streams, so parameters and learn_stats agree to fp32 noise on a small problem"""
    cfg = _cfg(pkg, 0, n_envs=32, n_steps=16, batch_size=128, epochs=2, episode_len=500, seed=4)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 1, 0.05); h.set_params(flat); o.set_params(flat)   # small logits: p ~ 0.5, flips are measure-zero
    h.env_reset(cfg.seed); o.env_reset(cfg.seed)
    sh, fh = h.train(3 * 32 * 16 + 5); so, _ = o.train(3 * 32 * 16 + 5)  # remainder steps are dropped (ppo.jl:117)
    assert len(sh) == len(so) == 3 and all(f > 0 for f in fh)
    for a, b in zip(sh, so):
        assert a.n_updates == b.n_updates == 2 * 4
        assert a.loss == pytest.approx(b.loss, rel=2e-3, abs=1e-5)
        assert a.explained_variance == pytest.approx(b.explained_variance, rel=2e-3, abs=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-3, atol=2e-5)


def test_configs0_readme_quickstart_matches_oracle(pkg, oracle_mod):
    """BASELINE.json configs[0] — the reference's README quick-start (/root/reference README.md:50-73): CartPole-v1, 4 envs, PPO() defaults
    (n_steps 2048, batch_size 64, 10 epochs => 1 280 optimiser steps per iteration on the small-minibatch path).  One full iteration with injected sampling noise and DataLoader order vs the oracle: every buffer
    field, the learn_stats and the parameters after 1 280 Adam steps; then a second iteration on top (the env is not reset, Adam state carries)."""
    capi = pkg._capi
    env = pkg.CartPoleEnv(max_steps=500); alg = pkg.PPO()
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
    cfg = pkg.make_config(env, 4, alg, layer, seed=42)
    assert (cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.hidden1, cfg.hidden2) == (4, 2048, 64, 10, 64, 64)
    assert (cfg.learning_rate, cfg.clip_range, cfg.vf_coef, cfg.max_grad_norm) == pytest.approx((3e-4, 0.2, 0.5, 0.5))
    E, T, N = 4, 2048, 8192
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = pkg.flatten_params(layer.initialparameters(np.random.default_rng(0)))       # orthogonal init, gains sqrt2 / 0.01 / 1 (layer_constructors.jl:61-65)
    h.set_params(flat); o.set_params(flat); h.env_reset(42); o.env_reset(42)
    rng = np.random.default_rng(7)
    for it in range(2):
        noise = rng.random(N)
        h.set_noise(noise); o.set_noise(noise)
        h.collect_rollout(); o.collect_rollout()
        ah, ao = h.buffer(capi.BUF_ACTIONS).reshape(T, E), o.buffer(capi.BUF_ACTIONS).reshape(T, E)
        ok = np.cumprod(ah == ao, axis=0).astype(bool)
        first = ok.copy(); first[1:] = ok[:-1]; first[0] = True
        sel = first & ~ok
        nflip = _flip_report(o, cfg, o.buffer(capi.BUF_OBSERVATIONS).reshape(T, E, -1)[sel], noise.reshape(T, E)[sel], ah[sel], ao[sel], f" configs[0] iteration {it}:") if sel.any() else 0
        full = ok.all(axis=0)
        assert full.sum() >= 3, f"{nflip} CDF-edge flips diverged more than one of the 4 envs"
        for which, tol in ((capi.BUF_OBSERVATIONS, 2e-5), (capi.BUF_VALUES, 5e-5), (capi.BUF_LOGPROBS, 1e-4), (capi.BUF_REWARDS, 0), (capi.BUF_ADVANTAGES, 1e-3), (capi.BUF_RETURNS, 1e-3)):
            a, b = h.buffer(which).reshape(T, E, -1), o.buffer(which).reshape(T, E, -1)
            np.testing.assert_allclose(a[:, full], b[:, full], atol=tol, rtol=tol)
        np.testing.assert_array_equal(h.buffer(capi.BUF_FLAGS).reshape(T, E)[:, full], o.buffer(capi.BUF_FLAGS).reshape(T, E)[:, full])
        assert (o.buffer(capi.BUF_FLAGS) & 1).any()                                     # real CartPole episodes: the pole falls
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))                                        # (compared above) the UPDATE comparison starts from identical data, also where an env flipped
        perm = np.stack([np.random.default_rng(1000 * it + e).permutation(N) for e in range(cfg.epochs)]).astype(np.int64)
        data = {w: o.buffer(w) for w in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES)}
        if it == 0:
            # the first 256 optimiser steps (2 of the 10 epochs) on their own handles, at the short-update tolerance: individual Adam steps are pinned here
            import copy
            cfg2 = copy.copy(cfg); cfg2.epochs = 2
            h2, o2 = pkg.Handle(cfg2), oracle_mod.Oracle(cfg2)
            h2.set_params(flat); o2.set_params(flat)
            for w, arr in data.items():
                h2.set_buffer(w, arr); o2.set_buffer(w, arr)
            h2.set_permutation(perm[:2]); o2.set_permutation(perm[:2])
            s2h, s2o = h2.ppo_update(), o2.ppo_update()
            assert s2h.n_updates == s2o.n_updates == 256
            print(f"[configs0] first 256 optimiser steps: max abs parameter difference {np.abs(h2.get_params() - o2.get_params()).max():.2e}")
            np.testing.assert_allclose(h2.get_params(), o2.get_params(), rtol=2e-4, atol=1e-5)
            h2.close()
        h.set_permutation(perm); o.set_permutation(perm)
        sh, so = h.ppo_update(), o.ppo_update()
        assert sh.n_updates == so.n_updates == 1280 and not sh.early_stopped
        for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first"):
            assert getattr(sh, f) == pytest.approx(getattr(so, f), rel=2e-3, abs=5e-6), (it, f)
        assert sh.loss == pytest.approx(so.loss, rel=1e-4 if it == 0 else 1e-3)          # north_star: PPO loss within 1e-4 rel (iteration 0: identical inputs)
        # 1 280 sequential Adam steps on a NON-SMOOTH loss: min(r A, clamp(r) A) (ppo.jl:381-382) switches a sample's gradient on or off where its ratio crosses
        # 1 +- clip_range, so two correct fp32 runs stay together (measured below: ~1e-6 over the first 256 steps) until the first sample whose ratio sits within
        # fp32 rounding of the boundary, and differ by one sample's gradient from there on (profiles/r03_configs0_tolerance.md: step 598 of iteration 0, 17 vs 16 of
        # 64 samples clipped, 3.8e-4 of the update afterwards — the same 3.8e-4 for torch-Float32, for f32 gradients with f64 Adam and for f64 gradients with f32 Adam).
        # So the bound is MEASURED here, not chosen: the yardstick is the reference's own precision — the same loop in Float32 (torch autograd, f32 Adam) against the
        # same loop in float64; the device may sit no further from float64 than 3x that (and the oracle's own distance is printed beside it).
        from test_oracle_crosschecks import f64_ppo_update
        p0 = flat if it == 0 else start
        bufs = tuple(data[w].reshape(N, -1) if w == capi.BUF_OBSERVATIONS else data[w].reshape(N) for w in
                     (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES))
        dh, do = h.get_params().astype(np.float64) - p0, o.get_params().astype(np.float64) - p0
        if it == 0:                                                                      # (iteration 1 carries each side's own Adam moments: no common float64 start)
            d64 = f64_ppo_update(p0, cfg, bufs, perm, True, 2) - p0
            d32 = f64_ppo_update(p0, cfg, bufs, perm, True, 2, dtype=np.float32).astype(np.float64) - p0
            n64 = np.linalg.norm(d64)
            r_dev, r_orc, r_f32 = (np.linalg.norm(x - d64) / n64 for x in (dh, do, d32))
            print(f"[configs0] distance of the update from the float64 loop: device {r_dev:.2e}, oracle {r_orc:.2e}, Float32 loop (the reference's precision) {r_f32:.2e}")
            assert r_dev <= max(3.0 * r_f32, 3.0 * r_orc, 1e-5)
        rel = np.linalg.norm(dh - do) / np.linalg.norm(do)
        print(f"[configs0] iteration {it}: |update| = {np.linalg.norm(do):.4f}, relative difference of the update {rel:.2e}, max abs {np.abs(dh - do).max():.2e}")
        assert rel <= 2e-2 and np.abs(dh - do).max() <= 4 * cfg.learning_rate            # iteration 1 (no float64 yardstick): a handful of boundary samples at most
        assert np.abs(dh).max() > 1e-3                                                   # 1 280 Adam steps moved the weights
        st, sc = h.env_get_state(); o.env_set_state(st, sc)
        start = h.get_params().astype(np.float64)
        o.set_params(h.get_params())                                                     # iteration 1 starts from identical weights (Adam moments stay each side's own)


def test_large_size_properties(pkg):
    """BASELINE-size invariants that need no oracle: reproducibility, returns = adv + values, flags pattern of
    fixed-length episodes, finite stats, parameters move."""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=65536, n_steps=64, episode_len=25, fixed_length_episodes=1, batch_size=65536 * 64 // 8, epochs=1)
    flat = _params(9155, 3, 0.3)
    outs = []
    for rep in range(2):
        h = pkg.Handle(cfg); h.set_params(flat); h.env_reset(42)
        h.collect_rollout()
        outs.append((h.buffer(capi.BUF_ADVANTAGES), h.buffer(capi.BUF_ACTIONS)))
        if rep == 0:
            fl = h.buffer(capi.BUF_FLAGS).reshape(64, 65536)
            expect = np.zeros(64, np.uint8); expect[24::25] = 2
            assert (fl == expect[:, None]).all()                          # every env truncates at steps 25, 50 (synthetic fixed-length episodes)
            np.testing.assert_allclose(h.buffer(capi.BUF_RETURNS), h.buffer(capi.BUF_ADVANTAGES) + h.buffer(capi.BUF_VALUES), atol=1e-5)
            assert (h.buffer(capi.BUF_REWARDS) == 1.0).all()
            st = h.ppo_update()
            assert st.n_updates == 8 and np.isfinite([st.loss, st.grad_norm, st.explained_variance]).all()
            assert st.ratio_first == pytest.approx(1.0, abs=1e-5)        # first minibatch of the first epoch: ratio == 1 (ppo.jl:209-212)
            assert not np.array_equal(h.get_params(), flat)
        h.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("config", ["configs[1]", "configs[2]"])
def test_full_size_properties(pkg, config):
    """BASELINE.json configs[1] (CartPole, [64,64]) and configs[2] (Pendulum, [256,256], NormalizeWrapperEnv) at FULL size (65 536 envs x 2048 steps = 134 217 728 samples, minibatch 4 194 304): size-independent properties —
    the truncation pattern of the synthetic fixed-length episodes, returns = advantages + values over the whole buffer, bit-identical buffers on a
    second rollout from the same seed (a checksum of the whole advantage buffer), ratio == 1 on the first minibatch, 32 optimiser steps per epoch"""
    capi = pkg._capi
    E, T = 65536, 2048
    if config == "configs[1]":
        cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, episode_len=500, fixed_length_episodes=1, batch_size=E * T // 32, epochs=1); L = 500
    else:
        cfg = _cfg(pkg, 1, n_envs=E, n_steps=T, episode_len=200, batch_size=E * T // 32, epochs=1, hidden1=256, hidden2=256, norm_training=1, norm_obs=1, norm_reward=1); L = 200
    flat = _params(9155 if config == "configs[1]" else 134147, 3, 0.3 if config == "configs[1]" else 0.05)
    sums = []
    for rep in range(2):
        h = pkg.Handle(cfg); h.set_params(flat); h.env_reset(42)
        h.collect_rollout()
        adv = h.buffer(capi.BUF_ADVANTAGES)
        sums.append((float(adv.astype(np.float64).sum()), int(np.frombuffer(adv.tobytes(), np.uint32).astype(np.uint64).sum())))   # value sum + bit-pattern checksum
        if rep == 0:
            fl = h.buffer(capi.BUF_FLAGS).reshape(T, E)
            expect = np.zeros(T, np.uint8); expect[L - 1::L] = 2                    # Pendulum never terminates: every env truncates every 200 steps
            assert (fl == expect[:, None]).all()
            del fl
            ret = h.buffer(capi.BUF_RETURNS); ret -= adv; ret -= h.buffer(capi.BUF_VALUES)
            assert np.abs(ret).max() <= 1e-4 * max(1.0, float(np.abs(adv).max()))
            del ret
            st = h.ppo_update()
            assert st.n_updates == 32 and np.isfinite([st.loss, st.grad_norm, st.explained_variance]).all()
            assert st.ratio_first == pytest.approx(1.0, abs=1e-5)
        del adv
        h.close()
    assert sums[0] == sums[1]


def test_host_mirror_train(pkg):
    """the reference-shaped API end to end: Agent / ActorCriticLayer / PPO / DeviceParallelEnv / train_ / collect_rollout_"""
    env = pkg.DeviceParallelEnv(pkg.CartPoleEnv(), 64, seed=1)
    alg = pkg.PPO(n_steps=32, batch_size=256, epochs=2)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
    agent = pkg.Agent(layer, alg, seed=3)
    p0 = pkg.flatten_params(agent.train_state.parameters)
    obs = env.observe()
    assert len(obs) == 64 and obs[0].shape == (4,)
    rew, term, trunc, infos = env.act_(np.full(64, 1, np.int32))
    assert rew.shape == (64,) and not trunc.any() and all(i == {} for i in infos)
    stats, timer = pkg.train_(agent, env, alg, 3 * 64 * 32)
    assert set(stats) == {"entropy_losses", "policy_losses", "value_losses", "approx_kl_divs", "clip_fractions", "losses",
                          "explained_variances", "fps", "grad_norms", "learning_rates"}          # ppo.jl:301-312
    assert all(len(v) == 3 for v in stats.values()) and {"setup", "training_loop", "collect_rollout"} <= set(timer)
    assert not np.array_equal(pkg.flatten_params(agent.train_state.parameters), p0)
    buf = pkg.RolloutBuffer(alg.n_steps, 64, alg.gae_lambda, alg.gamma)
    fps, ok = pkg.collect_rollout_(buf, agent, alg, env)
    assert ok and buf.observations.shape == (64 * 32, 4) and buf.actions.dtype == np.int64
    assert np.array_equal(np.sort(buf.to_reference_order()), np.arange(64 * 32))


def test_rccl_plumbing_single_rank(pkg, oracle_mod, monkeypatch):
    """dril_comm_unique_id / dril_comm_init / the in-stream ncclAllReduce of [grads | sums] and of the advantage moments,
    exercised with a 1-rank communicator (DRIL_FORCE_ALLREDUCE=1): results must equal the oracle's single-process update.
    The N>1 arithmetic is covered on CPU by tests/test_distributed_gloo.py and on one device by the loopback ranks of tests/test_gpu_dataparallel.py;
    RCCL across more than one device has not run yet (the driver's multi-GPU bench is the first contact: bench.py's launcher makes it self-diagnosing)."""
    monkeypatch.setenv("DRIL_FORCE_ALLREDUCE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=16, n_steps=24, batch_size=96, epochs=2, episode_len=11)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    uid = h.comm_unique_id()
    assert len(uid) == 128
    h.comm_init(uid)
    flat = _params(h.P, 77, 0.3); h.set_params(flat); o.set_params(flat)
    o.env_reset(5); o.collect_rollout()
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(h.N) for e in range(cfg.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates and sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    # device-generated order: the per-epoch advantage-moment table goes through one all-reduce per epoch
    h.set_permutation(None); o.set_permutation(None)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates and sh.loss == pytest.approx(so.loss, rel=2e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=5e-4, atol=5e-6)


@pytest.mark.parametrize("kind,flags", [(1, (1, 1)), (0, (1, 1)), (1, (1, 0)), (1, (0, 1)), (2, (1, 1)), (3, (1, 1)), (4, (1, 1)), (7, (1, 1))])   # 2: Normalize(Parallel([Scaling(Pendulum)])); 3, 4: MountainCar
@pytest.mark.parametrize("via_rccl", [False, True])
def test_normalize_wrapper_rollout_matches_oracle(pkg, oracle_mod, kind, flags, via_rccl, monkeypatch):
    """NormalizeWrapperEnv on device (normalizeWrapperEnv.jl:21-50,123-197): running obs/return statistics (updated on EVERY
    observe, including the double update at a rollout boundary), normalised + clipped obs and rewards, normalised
    terminal_observation for the truncation bootstrap — vs the oracle's line-by-line restatement.
    via_rccl: the data-parallel form of the statistics (partial sums folded to one row, summed over ranks by ncclAllReduce once per env
    step, merged with n = world * E) on a 1-rank communicator must give the same numbers."""
    capi = pkg._capi
    E, T, L = 48, 30, 9
    if via_rccl:
        monkeypatch.setenv("DRIL_FORCE_ALLREDUCE", "1"); monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=L, batch_size=E * T // 2, epochs=1, norm_training=1, norm_obs=flags[0],
               norm_reward=flags[1], clip_obs=5.0, clip_reward=2.0)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    if via_rccl:
        h.comm_init(h.comm_unique_id())
    flat = _params(h.P, 31, 0.4); h.set_params(flat); o.set_params(flat)
    h.env_reset(9); o.env_reset(9)
    rng = np.random.default_rng(2)
    for rollout in range(2):
        noise = rng.random(E * T) if h.discrete else rng.standard_normal((E * T, h.A)).astype(np.float32)
        h.set_noise(noise); o.set_noise(noise)
        h.collect_rollout(); o.collect_rollout()
        sh = h.norm_get_stats(); om, ov, oc, rm, rv, rc = o.norm_stats()
        assert (sh["obs_count"], sh["ret_count"]) == (oc, rc)
        assert oc == (E * (T + 1) * (rollout + 1) if flags[0] else 0)            # stats update on every observe call (T+1 per rollout)
        np.testing.assert_allclose(sh["obs_mean"], om, rtol=2e-5, atol=2e-6); np.testing.assert_allclose(sh["obs_var"], ov, rtol=1e-4, atol=1e-6)
        assert sh["ret_mean"] == pytest.approx(rm, rel=1e-4, abs=1e-5) and sh["ret_var"] == pytest.approx(rv, rel=1e-4, abs=1e-5)
        ah, ao = h.buffer(capi.BUF_ACTIONS).reshape(T, E, -1), o.buffer(capi.BUF_ACTIONS).reshape(T, E, -1)
        ok = np.cumprod((ah == ao).all(axis=2), axis=0).astype(bool).all(axis=0) if h.discrete else np.ones(E, bool)
        assert ok.mean() >= 0.95
        for which, tol in ((capi.BUF_OBSERVATIONS, 1e-4), (capi.BUF_VALUES, 2e-4), (capi.BUF_LOGPROBS, 2e-4), (capi.BUF_REWARDS, 2e-4),
                           (capi.BUF_ADVANTAGES, 2e-3), (capi.BUF_RETURNS, 2e-3)):
            a, b = h.buffer(which).reshape(T, E, -1), o.buffer(which).reshape(T, E, -1)
            np.testing.assert_allclose(a[:, ok], b[:, ok], atol=tol, rtol=tol)
        fh, fo = h.buffer(capi.BUF_FLAGS).reshape(T, E), o.buffer(capi.BUF_FLAGS).reshape(T, E)
        np.testing.assert_array_equal(fh[:, ok], fo[:, ok])
        tr = (fo & 2).astype(bool) & ok[None, :]
        assert tr.any()
        np.testing.assert_allclose(h.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], o.buffer(capi.BUF_BOOTSTRAP).reshape(T, E)[tr], atol=2e-4, rtol=2e-4)
        if flags[0]:
            assert np.abs(h.buffer(capi.BUF_OBSERVATIONS)).max() <= 5.0 + 1e-6      # clip_obs
        if flags[1]:
            assert np.abs(h.buffer(capi.BUF_REWARDS)).max() <= 2.0 + 1e-6           # clip_reward
        st, sc = h.env_get_state(); o.env_set_state(st, sc)


def test_normalize_wrapper_env_verbs_and_training_flag(pkg, oracle_mod):
    """observe/act! through the wrapper (normalizeWrapperEnv.jl:123-165); training=false freezes the statistics (:281-324 of the
    reference's test file); dril_norm_set_stats round-trips (save/load, :261-297)."""
    cfg = _cfg(pkg, 1, n_envs=100, n_steps=4, episode_len=6, batch_size=4, norm_training=1, norm_obs=1, norm_reward=1)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    h.env_reset(3); o.env_reset(3)
    rng = np.random.default_rng(0)
    for step in range(14):
        np.testing.assert_allclose(h.env_observe(True), o.env_observe(True), atol=2e-5, rtol=2e-5)
        a = rng.uniform(-2.5, 2.5, (100, 1)).astype(np.float32)
        rh, th, uh, oh = h.env_step(a); ro, to, uo, oo = o.env_step(a)
        np.testing.assert_allclose(rh, ro, atol=2e-5, rtol=2e-4); np.testing.assert_array_equal(uh, uo)
        np.testing.assert_allclose(oh[uh], oo[uo], atol=2e-5, rtol=2e-5)            # normalised terminal_observation
        st, sc = h.env_get_state(); o.env_set_state(st, sc)
    s = h.norm_get_stats()
    assert s["obs_count"] == 14 * 100 and s["ret_count"] == 14 * 100
    # frozen statistics
    cfg2 = _cfg(pkg, 1, n_envs=100, n_steps=4, episode_len=6, batch_size=4, norm_training=0, norm_obs=1, norm_reward=1)
    h2 = pkg.Handle(cfg2); h2.env_reset(3)
    h2.norm_set_stats(s["obs_mean"], s["obs_var"], s["obs_count"], s["ret_mean"], s["ret_var"], s["ret_count"])
    before = h2.norm_get_stats()
    h2.env_observe(True); h2.env_step(a); h2.env_observe(True)
    after = h2.norm_get_stats()
    assert before["obs_count"] == after["obs_count"] == 1400 and np.array_equal(before["obs_mean"], after["obs_mean"])
    assert after["ret_var"] == before["ret_var"]


def test_stepwise_path_equals_persistent_kernel(pkg, monkeypatch):
    """the step-granular launch sequence and the fused persistent rollout_kernel are the same algorithm"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=200, n_steps=40, episode_len=13, batch_size=400, epochs=1)
    flat = _params(9155, 8, 0.4)
    outs = []
    for force in ("0", "1"):
        monkeypatch.setenv("DRIL_FORCE_STEPWISE", force)
        h = pkg.Handle(cfg); h.set_params(flat); h.env_reset(21); h.collect_rollout()
        outs.append({w: h.buffer(w) for w in range(10)})
        h.close()
    for w in range(10):
        if w in (capi.BUF_BOOTSTRAP,):
            m = (outs[0][capi.BUF_FLAGS] & 2).astype(bool)
            np.testing.assert_allclose(outs[0][w][m], outs[1][w][m], atol=1e-6)
        elif w == capi.BUF_LAST_VALUES:
            np.testing.assert_allclose(outs[0][w], outs[1][w], atol=1e-6)
        else:
            np.testing.assert_allclose(outs[0][w].astype(np.float64), outs[1][w].astype(np.float64), atol=1e-6)


# ---- hidden_dims = [256, 256] (BASELINE configs[2]) and [128, 128]: wide forward (W2 streamed from L2) and the workgroup-cooperative grad
# kernel (H / 32 waves per workgroup) ----
@pytest.mark.parametrize("H", [256, 128])
@pytest.mark.parametrize("kind,B", [(1, 100), (0, 33), (3, 70), (4, 45), (6, 77)])      # 6 = Acrobot-v1 (D = 6: four first-layer k-steps in the wide kernels, end of round 3)
def test_wide_forward_evaluate(pkg, oracle_mod, kind, B, H):
    cfg = _cfg(pkg, kind, n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    if H == 256 and kind < 2:
        assert h.P == (134147 if kind == 1 else 2 * (256 * 4 + 256 + 256 * 256 + 256) + 256 * 2 + 2 + 256 + 1)
    flat = _params(h.P, 3, 0.12); h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(B)
    obs = rng.uniform(-2, 2, (B, h.D)).astype(np.float32)
    noise = rng.random(B) if kind in (0, 6) else rng.standard_normal((B, h.A)).astype(np.float32)
    ah, vh, lh = h.policy_forward(obs, noise); ao, vo, lo = o.policy_forward(obs, noise)
    np.testing.assert_allclose(vh, vo, atol=5e-5, rtol=5e-5)
    if kind in (0, 6):
        assert (ah == ao).mean() >= 0.99
    else:
        np.testing.assert_allclose(ah, ao, atol=5e-5, rtol=5e-5)
    ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
    np.testing.assert_allclose(ve, vo2, atol=5e-5, rtol=5e-5); np.testing.assert_allclose(le, lo2, atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(h.predict_values(obs), vo, atol=5e-5, rtol=5e-5)


@pytest.mark.parametrize("H", [256, 128])
@pytest.mark.parametrize("kind,B,variant", [(1, 64, "default"), (1, 1000, "ent_vfclip"), (0, 333, "default"), (3, 200, "default"), (4, 129, "ent_vfclip"), (6, 150, "ent_vfclip"), (6, 16403, "default")])   # Acrobot: three-quad records; 16 403 samples = the wide split kernel by the size rule
def test_wide_ppo_loss_and_gradient(pkg, oracle_mod, kind, B, variant, H):
    kw = dict(n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H)
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    cfg = _cfg(pkg, kind, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 40, 0.1); h.set_params(flat); o.set_params(flat)
    batch = _batch(o, cfg, B, 1)
    lh, sh, gh = h.ppo_loss_grad(*batch); lo, so, go = o.ppo_loss_grad(*batch)
    assert lh == pytest.approx(lo, rel=1e-4)
    np.testing.assert_allclose(sh, so, rtol=3e-4, atol=3e-6)
    assert np.linalg.norm(gh - go) <= 3e-4 * np.linalg.norm(go)
    lh2, _, gh2 = h.ppo_loss_grad(*batch)
    assert lh2 == lh and np.array_equal(gh, gh2)


@pytest.mark.parametrize("kind,H", [(1, 256), (1, 128), (4, 128), (2, 128), (7, 128), (7, 256)])
def test_wide_rollout_and_update_config3_shape(pkg, oracle_mod, kind, H):
    """configs[2] at test size: Pendulum, DiagGaussian, hidden [256,256], NormalizeWrapperEnv — rollout then PPO update"""
    capi = pkg._capi
    E, T = 40, 24
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=10, batch_size=E * T // 3, epochs=2, hidden1=H, hidden2=H,
               norm_training=1, norm_obs=1, norm_reward=1)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 12, 0.08); h.set_params(flat); o.set_params(flat)
    h.env_reset(4); o.env_reset(4)
    noise = np.random.default_rng(0).standard_normal((E * T, 1)).astype(np.float32)
    h.set_noise(noise); o.set_noise(noise)
    h.collect_rollout(); o.collect_rollout()
    for which, tol in ((capi.BUF_OBSERVATIONS, 2e-4), (capi.BUF_VALUES, 3e-4), (capi.BUF_LOGPROBS, 3e-4), (capi.BUF_REWARDS, 3e-4),
                       (capi.BUF_ADVANTAGES, 3e-3), (capi.BUF_RETURNS, 3e-3)):
        np.testing.assert_allclose(h.buffer(which), o.buffer(which), atol=tol, rtol=tol)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(cfg.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert sh.n_updates == so.n_updates == 6 and sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)


@pytest.mark.parametrize("start", [0, 1, -1, -2])
def test_discrete_action_start_offsets(pkg, oracle_mod, start):
    """actions lie in Discrete(n, start) for start 0/1/-1/-2 and are stored with the offset (test/test_policies.jl:66-100,
    src/DRiLDistributions/categorical.jl:51, trajectory.jl:48)"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=96, n_steps=12, episode_len=50, batch_size=96 * 12, epochs=1, action_start=start)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 6, 0.4); h.set_params(flat); o.set_params(flat)
    h.env_reset(2); o.env_reset(2)
    noise = np.random.default_rng(1).random(h.N); h.set_noise(noise); o.set_noise(noise)
    h.collect_rollout(); o.collect_rollout()
    a = h.buffer(capi.BUF_ACTIONS)
    assert set(np.unique(a)) == {start, start + 1}
    assert (a == o.buffer(capi.BUF_ACTIONS)).mean() > 0.99
    val, lp, ent = h.evaluate_actions(h.buffer(capi.BUF_OBSERVATIONS), a)
    np.testing.assert_allclose(lp, h.buffer(capi.BUF_LOGPROBS), atol=1e-5, rtol=1e-5)
    st = h.ppo_update(); so = o.ppo_update()
    assert st.loss == pytest.approx(so.loss, rel=2e-3)


def test_env_seeding_rule(pkg):
    """Random.seed!(penv, s) seeds sub-env i with s + i - 1 (wrapper_utils.jl:39-44, test/test_env_seeding.jl:31-173):
    same seed -> same observations; env i under seed s == env 0 under seed s + i"""
    cfg = _cfg(pkg, 0, n_envs=8, n_steps=4, batch_size=4)
    a, b, c = pkg.Handle(cfg), pkg.Handle(cfg), pkg.Handle(cfg)
    a.env_reset(100); b.env_reset(100); c.env_reset(103)
    oa, ob, oc = a.env_observe(), b.env_observe(), c.env_observe()
    assert np.array_equal(oa, ob) and not np.array_equal(oa, oc)
    assert np.array_equal(oa[3], oc[0]) and np.array_equal(oa[7], oc[4])
    act = np.ones(8, np.int32)
    for _ in range(30):                       # auto-reset draws (episode counter) are reproducible too
        ra, ta, ua, _ = a.env_step(act); rb, tb, ub, _ = b.env_step(act)
        assert np.array_equal(ta, tb)
    assert np.array_equal(a.env_get_state()[0], b.env_get_state()[0]) and ta.any() | True


@pytest.mark.parametrize("E,T,B", [(1, 1, 2), (33, 3, 7), (130, 5, 1000), (257, 2, 514)])
def test_ragged_sizes(pkg, oracle_mod, E, T, B):
    """sizes that are not multiples of the 32-sample tile / 128-env workgroup; batch larger than the buffer; single env"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, batch_size=B, epochs=2, episode_len=4)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 2, 0.3); h.set_params(flat); o.set_params(flat)
    h.env_reset(9); o.env_reset(9)
    noise = np.random.default_rng(E).random(E * T); h.set_noise(noise); o.set_noise(noise)
    h.collect_rollout(); o.collect_rollout()
    for which, tol in ((capi.BUF_VALUES, 5e-5), (capi.BUF_ADVANTAGES, 1e-3), (capi.BUF_RETURNS, 1e-3)):
        np.testing.assert_allclose(h.buffer(which), o.buffer(which), atol=tol, rtol=tol)
    if E * T >= 2 and (E * T) % B != 1:
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))
        perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(2)]).astype(np.int64)
        h.set_permutation(perm); o.set_permutation(perm)
        if E * T == 1:
            return
        sh, so = h.ppo_update(), o.ppo_update()
        assert sh.n_updates == so.n_updates
        np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=3e-4, atol=3e-6)


def test_ppo_learns_cartpole(pkg):
    """learning smoke test in the spirit of test/test_ppo_integration.jl:1-40: after training, episodes last longer"""
    env = pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=200), 256, seed=5)
    alg = pkg.PPO(n_steps=64, batch_size=2048, epochs=8, learning_rate=3e-3, ent_coef=0.0)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
    agent = pkg.Agent(layer, alg, seed=1)
    def mean_episode_length():
        buf = pkg.RolloutBuffer(alg.n_steps, 256, alg.gae_lambda, alg.gamma)
        pkg.collect_rollout_(buf, agent, alg, env)
        done = (buf.flags != 0).sum()
        return 256 * 64 / max(int(done), 1)
    before = mean_episode_length()
    stats, _ = pkg.train_(agent, env, alg, 40 * 64 * 256)
    after = mean_episode_length()
    assert np.isfinite(stats["losses"]).all() and len(stats["losses"]) == 40
    assert after > 2.0 * before and after > 40, (before, after)


@pytest.mark.parametrize("kind,norm,L,W", [(0, 0, 500, 100), (0, 0, 7, 100), (1, 1, 9, 25), (0, 1, 500, 10)])
def test_monitor_wrapper_episode_stats(pkg, oracle_mod, kind, norm, L, W):
    """MonitorWrapperEnv (monitorWrapperEnv.jl:46-70): mean return / length over the last `stats_window` finished episodes in
    (step, env) completion order, from RAW rewards, across rollouts and across the fused / step-granular / host-stepped paths."""
    E, T = 70, 40
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=L, batch_size=E * T, epochs=1, monitor_window=W,
               norm_training=norm, norm_obs=norm, norm_reward=norm)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 5, 0.5)
    if kind == 0:
        flat[4608:4610] = (1.5, -1.5)                     # biased policy: CartPole episodes terminate quickly
    h.set_params(flat); o.set_params(flat)
    h.env_reset(8); o.env_reset(8)
    assert h.monitor_stats() == (0.0, 0.0, 0)
    rng = np.random.default_rng(3)
    for rollout in range(3):
        noise = rng.random(E * T) if kind == 0 else rng.standard_normal((E * T, h.A)).astype(np.float32)
        h.set_noise(noise); o.set_noise(noise)
        h.collect_rollout(); o.collect_rollout()
        rh, lh, nh = h.monitor_stats(); ro, lo, no = o.monitor_stats()
        assert nh == no and nh > 0
        assert lh == pytest.approx(lo, rel=1e-6) and rh == pytest.approx(ro, rel=1e-5, abs=1e-5)
        st, sc = h.env_get_state(); o.env_set_state(st, sc)
    # host-stepped path keeps feeding the same window
    for step in range(12):
        a = (rng.integers(0, 2, E) + cfg.action_start).astype(np.int32) if kind == 0 else rng.uniform(-2, 2, (E, 1)).astype(np.float32)
        h.env_step(a); o.env_step(a)
        st, sc = h.env_get_state(); o.env_set_state(st, sc)
    rh, lh, nh = h.monitor_stats(); ro, lo, no = o.monitor_stats()
    assert nh == no == min(W, nh) and lh == pytest.approx(lo, rel=1e-6) and rh == pytest.approx(ro, rel=1e-5, abs=1e-5)
    if kind == 0:
        assert rh == pytest.approx(lh, rel=1e-6)          # CartPole: reward 1 per step => return == length


def test_monitor_off_is_an_error(pkg):
    h = pkg.Handle(_cfg(pkg, 0, n_envs=4, n_steps=2, batch_size=4))
    with pytest.raises(pkg.DrilError) as e:
        h.monitor_stats()
    assert e.value.code == pkg._capi.ERR_NOT_INITIALISED


@pytest.mark.parametrize("kind,det,kw", [(0, True, {}), (0, False, {}), (1, True, dict(norm_training=1, norm_obs=1, norm_reward=1, monitor_window=50)),
                                         (1, False, dict(norm_training=1, norm_obs=1, norm_reward=1)), (2, True, dict(monitor_window=20)), (3, True, {}), (4, False, dict(monitor_window=10))])
def test_evaluate_agent(pkg, oracle_mod, kind, det, kw):
    """evaluate_agent (src/evaluation.jl:54-143): reset, predict(deterministic) / act! / observe until the first n episodes finish,
    mean / corrected std of returns and lengths; monitored envs report RAW returns"""
    cfg = _cfg(pkg, kind, n_envs=24, n_steps=4, episode_len=15 if kind else 60, batch_size=24, **kw)      # MountainCar: every episode ends at the time limit
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 21, 0.5)
    if kind == 0:
        flat[4608:4610] = (0.8, -0.8)
    h.set_params(flat); o.set_params(flat)
    h.env_reset(13); o.env_reset(13)
    n = 40
    sh, rh, lh = h.evaluate_agent(n, det); so, ro, lo = o.evaluate_agent(n, det)
    assert np.array_equal(lh, lo)
    np.testing.assert_allclose(rh, ro, rtol=2e-4, atol=2e-4)
    for k in ("mean_reward", "std_reward", "mean_length", "std_length"):
        assert sh[k] == pytest.approx(so[k], rel=2e-4, abs=2e-4), k
    assert sh["n_steps"] == so["n_steps"] and sh["mean_length"] == pytest.approx(float(np.mean(lh)))
    assert sh["std_reward"] == pytest.approx(float(np.std(rh.astype(np.float64), ddof=1)), rel=1e-5, abs=1e-6)
    if kind == 0:
        assert np.allclose(rh, lh)                       # CartPole: one reward per step
    if det:                                              # deterministic evaluation is repeatable (reset! + mode)
        sh2, rh2, lh2 = h.evaluate_agent(n, det)
        assert np.array_equal(lh, lh2) and (kw or np.array_equal(rh, rh2))


def test_host_mirror_evaluate_and_wrappers(pkg):
    env = pkg.MonitorWrapperEnv(pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=100), 32, seed=3), 20)
    alg = pkg.PPO(n_steps=16, batch_size=128, epochs=1)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=0)
    stats = pkg.evaluate_agent(agent, env, n_eval_episodes=12)
    assert set(stats) == {"mean_reward", "std_reward", "mean_length", "std_length"} and stats["mean_reward"] == stats["mean_length"] > 0
    with pytest.raises(RuntimeError):
        pkg.evaluate_agent(agent, env, n_eval_episodes=5, reward_threshold=1e9)
    r, l, n = env.handle.monitor_stats()
    assert n == 20 and r == l


def test_scaling_wrapper_is_pendulum_in_other_units(pkg):
    """ScalingWrapperEnv(Pendulum) vs plain Pendulum on device: same simulator state trajectory when the wrapper-space action a is the torque
    2a; observations differ by exactly the affine map (scalingWrapperEnv.jl:71-79); wide [256,256] rollout runs on the wrapped env too"""
    capi = pkg._capi
    w = pkg.ScalingWrapperEnv(pkg.PendulumEnv())
    ca, cb = _cfg(pkg, 1, n_envs=200, n_steps=4, episode_len=6, batch_size=8), _cfg(pkg, 2, n_envs=200, n_steps=4, episode_len=6, batch_size=8)
    a, b = pkg.Handle(ca), pkg.Handle(cb)
    a.env_reset(3); b.env_reset(3)
    rng = np.random.default_rng(0)
    for _ in range(8):
        np.testing.assert_allclose(b.env_observe(), w.scale_observation(a.env_observe()), atol=1e-6)
        act = rng.uniform(-1.2, 1.2, (200, 1)).astype(np.float32)
        ra, ta, ua, xa = a.env_step(np.clip(w.unscale_action(act), -2, 2)); rb, tb, ub, xb = b.env_step(act)
        np.testing.assert_allclose(rb, ra, rtol=1e-5, atol=1e-5); np.testing.assert_array_equal(ub, ua)
        np.testing.assert_allclose(xb[ub], w.scale_observation(xa[ua]), atol=1e-6)
        sa, _ = a.env_get_state(); sb, _ = b.env_get_state()
        np.testing.assert_allclose(sb, sa, rtol=1e-5, atol=1e-5)
    env = pkg.DeviceParallelEnv(w, 256, seed=1)
    alg = pkg.PPO(n_steps=16, batch_size=1024, epochs=2)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(256, 256)), alg)
    stats, _ = pkg.train_(agent, env, alg, 2 * 16 * 256)
    assert len(stats["losses"]) == 2 and np.isfinite(stats["losses"]).all()
    obs = env.handle.buffer(capi.BUF_OBSERVATIONS)
    assert np.abs(obs).max() <= 1.0 + 1e-6                                                  # every observation inside the wrapper's Box(-1, 1)


@pytest.mark.parametrize("kind,B", [(0, 320), (1, 320), (3, 320), (4, 320), (0, 300), (1, 77)])   # 300 / 77: partial tiles and a ragged last minibatch on every kernel
def test_grad_kernel_variants_agree(pkg, oracle_mod, monkeypatch, kind, B):
    """hidden [64,64] has two update kernels: ppo_grad_kernel (exact f32 MFMA chain; small minibatches) and ppo_grad_pair_kernel (the three H x H
    contractions on the bf16 matrix cores with 3-piece operand splitting, f32 accumulate, two waves per tile; large minibatches).  Forced onto the same
    rollout and DataLoader order (DRIL_GRAD_VARIANT) they must give the same loss / gradient norm / parameters to fp32 noise, each within the oracle
    tolerances, and each must be bitwise reproducible"""
    capi = pkg._capi
    E, T = 32, 40
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, episode_len=11, batch_size=B, epochs=2, ent_coef=0.01)
    o = oracle_mod.Oracle(cfg)
    flat = _params(o.P, 3, 0.4); o.set_params(flat); o.env_reset(4)
    noise = np.random.default_rng(1).random(E * T) if o.discrete else np.random.default_rng(1).standard_normal((E * T, o.A)).astype(np.float32)
    o.set_noise(noise); o.collect_rollout()
    perm = np.stack([np.random.default_rng(7 + e).permutation(E * T) for e in range(2)]).astype(np.int64)
    o.set_permutation(perm); so = o.ppo_update()
    res = {}
    for variant in (0, 1):
        monkeypatch.setenv("DRIL_GRAD_VARIANT", str(variant))
        runs = []
        for rep in range(2):
            h = pkg.Handle(cfg); h.set_params(flat)
            for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
                h.set_buffer(which, o.buffer(which))
            h.set_permutation(perm)
            st = h.ppo_update()
            runs.append((st.loss, st.grad_norm, h.get_params()))
        assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][2], runs[1][2])            # deterministic slabs: bitwise reproducible
        assert runs[0][0] == pytest.approx(so.loss, rel=1e-4) and runs[0][1] == pytest.approx(so.grad_norm, rel=5e-4)
        np.testing.assert_allclose(runs[0][2], o.get_params(), rtol=2e-4, atol=3e-6)
        res[variant] = runs[0]
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-5) and res[0][1] == pytest.approx(res[1][1], rel=1e-5)
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("kind,B", [(0, 4096), (1, 1000), (0, 33)])
def test_split_kernel_loss_and_gradient(pkg, oracle_mod, monkeypatch, kind, B):
    """ppo_grad_pair_kernel through dril_ppo_loss_grad (forced: the size rule would pick the f32 kernel for these minibatches): loss within 1e-4 rel
    (north_star), gradient within 2e-4 of its norm — the same tolerances as the f32 kernel, i.e. the bf16 x 3 split is fp32-equivalent"""
    monkeypatch.setenv("DRIL_GRAD_VARIANT", "1")
    cfg = _cfg(pkg, kind, n_envs=2, n_steps=2, batch_size=2, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    for seed in range(2):
        flat = _params(h.P, 40 + seed, 0.25); h.set_params(flat); o.set_params(flat)
        batch = _batch(o, cfg, B, seed)
        lh, sh, gh = h.ppo_loss_grad(*batch); lo, so, go = o.ppo_loss_grad(*batch)
        assert lh == pytest.approx(lo, rel=1e-4)
        np.testing.assert_allclose(sh, so, rtol=2e-4, atol=2e-6)
        assert np.linalg.norm(gh - go) <= 2e-4 * np.linalg.norm(go)
        lh2, _, gh2 = h.ppo_loss_grad(*batch)
        assert lh2 == lh and np.array_equal(gh, gh2)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_pair_kernel_at_bench_scale_matches_the_f32_kernel(pkg, monkeypatch, kind):
    """ppo_grad_pair_kernel on a chip-filling minibatch (4 096 tiles: every workgroup slot of the device, unequal actor / critic pair counts, the default size rule):
    rollout + one update of 2 epochs x 2 minibatches against the exact-f32 kernel on the same seed — loss, gradient norm and parameters to fp32 noise; the
    library must report the pair kernel for the default and the f32 kernel when forced"""
    res = {}
    for variant in ("-1", "0"):
        monkeypatch.setenv("DRIL_GRAD_VARIANT", variant)
        env = pkg.CartPoleEnv(max_steps=500) if kind == 0 else pkg.PendulumEnv(max_steps=200)
        E, T = 2048, 128
        alg = pkg.PPO(n_steps=T, batch_size=E * T // 2, epochs=2)
        layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
        h = pkg.Handle(pkg.make_config(env, E, alg, layer, seed=3, fixed_length_episodes=True))
        h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(5))))
        h.env_reset(3); h.collect_rollout(); st = h.ppo_update()
        res[variant] = (h.get_params().copy(), st.loss, st.grad_norm, h.grad_kernel_info().split(":")[0])
        h.close()
    assert res["-1"][3] == "ppo_grad_pair_kernel" and res["0"][3] == "ppo_grad_kernel"
    assert res["-1"][1] == pytest.approx(res["0"][1], rel=1e-5) and res["-1"][2] == pytest.approx(res["0"][2], rel=1e-5)
    np.testing.assert_allclose(res["-1"][0], res["0"][0], rtol=1e-4, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,H,E", [(1, 256, 2048), (0, 256, 4096), (0, 128, 4096), (1, 128, 4096)])
def test_wide_split_kernel_matches_the_f32_wide_kernel(pkg, monkeypatch, kind, H, E):
    """hidden [256,256] (BASELINE configs[2] shape) and [128,128]: ppo_grad_wide_split_kernel (default) against ppo_grad_wide_kernel (DRIL_GRAD_VARIANT=0) on the same seed —
    rollout (Pendulum: under NormalizeWrapperEnv) + one update of 2 epochs x 2 minibatches: loss, gradient norm and parameters to fp32 noise.  The E = 4096 cases fill
    the chip (every workgroup slot; the Categorical head among them: the grid-size rules of the [64,64] kernels must not leak into the wide ones — they once did, and the
    critic lost 8 of its 128 slabs)"""
    res = {}
    for variant in ("-1", "0"):
        monkeypatch.setenv("DRIL_GRAD_VARIANT", variant)
        env = pkg.PendulumEnv(max_steps=200) if kind == 1 else pkg.CartPoleEnv(max_steps=500)
        T = 64
        alg = pkg.PPO(n_steps=T, batch_size=E * T // 2, epochs=2)
        layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(H, H))
        h = pkg.Handle(pkg.make_config(env, E, alg, layer, seed=3, fixed_length_episodes=True, normalize={} if kind == 1 else None))
        h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(5))))
        h.env_reset(3); h.collect_rollout(); st = h.ppo_update()
        res[variant] = (h.get_params().copy(), st.loss, st.grad_norm, h.grad_kernel_info().split(":")[0])
        h.close()
    assert res["-1"][3] == "ppo_grad_wide_split_kernel" and res["0"][3] == "ppo_grad_wide_kernel"
    assert res["-1"][1] == pytest.approx(res["0"][1], rel=1e-5) and res["-1"][2] == pytest.approx(res["0"][2], rel=1e-5)
    np.testing.assert_allclose(res["-1"][0], res["0"][0], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("B,gmax", [(64, None), (20, None), (33, None), (64, "1")])
def test_forced_pair_kernel_on_small_minibatches(pkg, oracle_mod, monkeypatch, B, gmax):
    """ppo_grad_pair_kernel forced (DRIL_GRAD_VARIANT=2) onto minibatches of two tiles, of less than one tile (the second pair of the workgroup owns no tile at all) and
    of one sample in the last tile: it has no in-kernel advantage moments, so ppo_step must route it through the moments launches — the round-2 abort
    (profiles/r02_split_kernel.md "The abort of 09:41") was this kernel reading a null moments pointer on the small path.  DRIL_GRAD_GMAX=1 leaves one slab per net:
    the pair kernel (two slabs per workgroup) must not be selected at all.  Every case against the oracle."""
    capi = pkg._capi
    monkeypatch.setenv("DRIL_GRAD_VARIANT", "2")
    if gmax:
        monkeypatch.setenv("DRIL_DEBUG", "1"); monkeypatch.setenv("DRIL_GRAD_GMAX", gmax)        # (a diagnostic switch: honoured only with DRIL_DEBUG=1)
    E, T = 4, 32
    cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, batch_size=B, epochs=2, episode_len=9)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 5, 0.3); h.set_params(flat); o.set_params(flat)
    o.env_reset(2); o.collect_rollout()
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
        h.set_buffer(which, o.buffer(which))
    perm = np.stack([np.random.default_rng(e).permutation(E * T) for e in range(2)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert (h.grad_kernel_info().split(":")[0] == "ppo_grad_pair_kernel") == (gmax is None)
    assert sh.n_updates == so.n_updates and sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    batch = _batch(o, cfg, B, 1)                                                     # and through the parity entry point (no DataLoader order)
    lh, _, gh = h.ppo_loss_grad(*batch); lo, _, go = o.ppo_loss_grad(*batch)
    assert lh == pytest.approx(lo, rel=1e-4) and np.linalg.norm(gh - go) <= 2e-4 * np.linalg.norm(go)


@pytest.mark.parametrize("kind,E,T,B,kw", [
    (0, 16, 24, 64, {}),
    (0, 16, 24, 48, {"has_target_kl": 1, "target_kl": 0.002}),                         # the KL stop inside the persistent loop: skip that apply, stop (ppo.jl:235-238)
    (0, 8, 16, 32, {"has_clip_range_vf": 1, "clip_range_vf": 0.2, "ent_coef": 0.01}),
    (1, 8, 16, 20, {"ent_coef": 0.01}),                                                # DiagGaussian: the log_std parameter and its staged copy
    (3, 10, 13, 50, {}),                                                               # N = 130: a ragged last minibatch of 30
    (0, 4, 1024, 2, {"epochs": 5}),                                                    # 10 240 optimiser steps of 2 samples
    (6, 12, 20, 64, {"ent_coef": 0.01}),                                               # Acrobot-v1 (D = 6): four first-layer k-steps, three-quad records, 9 476 parameters
    (6, 7, 19, 40, {"has_clip_range_vf": 1, "clip_range_vf": 0.2}),                    # ... ragged: 133 samples = 3 x 40 + 13, eight valid lanes in the second pair's tile
])
def test_persistent_small_update_matches_oracle(pkg, oracle_mod, monkeypatch, kind, E, T, B, kw):
    """ppo_update_small_kernel (batch_size <= 64: the reference's default PPO()): all optimiser steps of an iteration in one persistent workgroup, against the oracle
    on identical buffers and an injected DataLoader order; the first case also against the per-step path (DRIL_NO_PERSISTENT_UPDATE) on the same data"""
    capi = pkg._capi
    kw = dict(kw); epochs = kw.pop("epochs", 3)
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, batch_size=B, epochs=epochs, episode_len=11, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = _params(h.P, 77, 0.3); h.set_params(flat); o.set_params(flat)
    o.env_reset(5); o.collect_rollout()
    bufs = (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES)
    for which in bufs:
        h.set_buffer(which, o.buffer(which))
    N = E * T
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(cfg.epochs)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert h.grad_kernel_info().split(":")[0] == "ppo_update_small_kernel"
    assert (sh.n_updates, sh.early_stopped) == (so.n_updates, so.early_stopped)
    if "has_target_kl" in kw:
        assert sh.early_stopped and sh.n_updates < cfg.epochs * -(-N // B)
    long_run = sh.n_updates > 5000
    if long_run:                                                                       # thousands of 2-sample steps on a non-smooth loss: the update as a whole (see the configs[0] test)
        dh, do = h.get_params().astype(np.float64) - flat, o.get_params().astype(np.float64) - flat
        rel = np.linalg.norm(dh - do) / np.linalg.norm(do)
        print(f"[persistent, {sh.n_updates} steps of {B} samples] relative difference of the update {rel:.2e}; mean loss {sh.loss:.5f} vs {so.loss:.5f}")
        assert rel <= 0.1 and sh.loss == pytest.approx(so.loss, rel=0.1) and sh.value_loss == pytest.approx(so.value_loss, rel=0.05)
    else:
        for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first"):
            assert getattr(sh, f) == pytest.approx(getattr(so, f), rel=5e-4, abs=2e-6), f
        assert sh.loss == pytest.approx(so.loss, rel=1e-4)
        np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    if B == 64:
        monkeypatch.setenv("DRIL_NO_PERSISTENT_UPDATE", "1")
        h2 = pkg.Handle(cfg); h2.set_params(flat)
        for which in bufs:
            h2.set_buffer(which, o.buffer(which))
        h2.set_permutation(perm); s2 = h2.ppo_update()
        assert h2.grad_kernel_info().split(":")[0] == "ppo_grad_kernel"
        assert s2.loss == pytest.approx(sh.loss, rel=1e-5)
        np.testing.assert_allclose(h2.get_params(), h.get_params(), rtol=1e-4, atol=2e-6)
    # the optimiser state left in global memory is the state the next update starts from: a second update on the same data stays with the oracle
    if not long_run and "has_target_kl" not in kw:
        sh2, so2 = h.ppo_update(), o.ppo_update()
        assert sh2.loss == pytest.approx(so2.loss, rel=2e-4)
        np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=4e-4, atol=6e-6)


def test_persistent_small_update_across_launch_boundaries(pkg, oracle_mod, monkeypatch):
    """the persistent kernel runs at most DRIL_SMALL_CHUNK (default 16 384) optimiser steps per launch; parameters, Adam moments and the beta powers pass from one
    launch to the next through global memory.  48 steps in one launch and in launches of 5 must agree BITWISE (same arithmetic, same order)"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=16, n_steps=24, batch_size=24, epochs=3, episode_len=11, ent_coef=0.01)
    o = oracle_mod.Oracle(cfg)
    flat = _params(o.P, 7, 0.3); o.set_params(flat); o.env_reset(5); o.collect_rollout()
    res = []
    for chunk in (None, "5"):
        if chunk: monkeypatch.setenv("DRIL_DEBUG", "1"); monkeypatch.setenv("DRIL_SMALL_CHUNK", chunk)
        h = pkg.Handle(cfg); h.set_params(flat)
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))
        sts = [h.ppo_update(), h.ppo_update()]                                        # twice: the device DataLoader order, update counter 0 and 1
        assert h.grad_kernel_info().split(":")[0] == "ppo_update_small_kernel" and sts[0].n_updates == 48
        res.append((h.get_params(), [(st.loss, st.grad_norm, st.approx_kl_div) for st in sts]))
        h.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]


def test_persistent_small_update_gives_up_instead_of_hanging(pkg, oracle_mod, monkeypatch):
    """the persistent update is TWO workgroups (actor | critic) that exchange one message per optimiser step through L2.  If the partner never runs (the GPU is shared and the
    two were not resident together), a workgroup must leave after a bounded wait — never hang the device.  DRIL_SMALL_DEBUG_SOLO launches the actor's workgroup alone:
      * with the snapshot the library keeps for its redos, nothing is lost: the update is taken back and redone on the per-step kernels (bitwise what
        DRIL_NO_PERSISTENT_UPDATE=1 gives), counted in persistent_fallbacks, and the handle stays on the per-step kernels;
      * without it (DRIL_NO_F32_RETRY=1) the call returns DRIL_ERR_HIP within seconds and leaves the parameters untouched."""
    import time
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=8, n_steps=16, batch_size=32, epochs=1, episode_len=11)
    o = oracle_mod.Oracle(cfg)
    flat = _params(o.P, 3, 0.3); o.set_params(flat); o.env_reset(5); o.collect_rollout()

    def make(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        h = pkg.Handle(cfg); h.set_params(flat)
        for k in env:
            monkeypatch.delenv(k)
        for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
            h.set_buffer(which, o.buffer(which))
        return h
    h, href = make(), make(DRIL_NO_PERSISTENT_UPDATE="1")
    monkeypatch.setenv("DRIL_SMALL_DEBUG_SOLO", "1")
    t0 = time.time()
    st = h.ppo_update()
    assert time.time() - t0 < 30.0
    monkeypatch.delenv("DRIL_SMALL_DEBUG_SOLO")
    sr = href.ppo_update()
    assert h.f32_fallback_info()["persistent_fallbacks"] == 1 and st.n_updates == sr.n_updates == 4 and st.loss == sr.loss
    np.testing.assert_array_equal(h.get_params(), href.get_params())
    assert h.get_optimizer_state()["steps"] == href.get_optimizer_state()["steps"] == 4
    assert not h.grad_kernel_info().startswith("ppo_update_small_kernel")
    st = h.ppo_update(); sr = href.ppo_update()                                       # the handle stays on the per-step kernels
    assert h.f32_fallback_info()["persistent_fallbacks"] == 1 and st.loss == sr.loss and not h.grad_kernel_info().startswith("ppo_update_small_kernel")
    h.close(); href.close()
    h = make(DRIL_NO_F32_RETRY="1")                                                    # no snapshot: the failure surfaces
    monkeypatch.setenv("DRIL_SMALL_DEBUG_SOLO", "1")
    t0 = time.time()
    with pytest.raises(Exception) as ei:
        h.ppo_update()
    assert time.time() - t0 < 30.0 and "partner workgroup" in str(ei.value)
    assert np.array_equal(h.get_params(), flat)
    monkeypatch.delenv("DRIL_SMALL_DEBUG_SOLO")
    st = h.ppo_update()                                                               # the knob is read per update: both workgroups again
    assert st.n_updates == 4 and np.isfinite(st.loss) and not np.array_equal(h.get_params(), flat) and h.grad_kernel_info().startswith("ppo_update_small_kernel")
    h.close()


def test_epoch_index_array_is_the_same_dataloader_order(pkg, monkeypatch):
    """chip-filling minibatches read the epoch's DataLoader order from an index array written once per epoch (epoch_index_kernel) instead of evaluating the keyed
    bijection inside the update kernel: the same order, hence bitwise the same update as with DRIL_NO_EPOCH_INDEX=1"""
    res = []
    for off in ("0", "1"):
        monkeypatch.setenv("DRIL_NO_EPOCH_INDEX", off)
        env = pkg.CartPoleEnv(max_steps=500)
        E, T = 2048, 128
        alg = pkg.PPO(n_steps=T, batch_size=E * T // 2, epochs=2)
        layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
        h = pkg.Handle(pkg.make_config(env, E, alg, layer, seed=3, fixed_length_episodes=True))
        h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(5))))
        h.env_reset(3); h.collect_rollout(); st = h.ppo_update()
        res.append((h.get_params().copy(), st.loss, st.grad_norm)); h.close()
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] and np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("stride", [1, 4])
def test_profile_events_stride(pkg, stride):
    """cfg.profile_events = k: the per-iteration kernels are bracketed at every launch, the per-optimiser-step kernels at every k-th; `launches` counts all of them either way
    and the parameters do not depend on the bracketing"""
    res = []
    for k in (0, stride):
        cfg = pkg._capi.default_config(0)
        cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.profile_events = 256, 16, 512, 3, k
        h = pkg.Handle(cfg)
        h.set_params((np.random.default_rng(1).standard_normal(h.P) * 0.3).astype(np.float32)); h.env_reset(3)
        h.collect_rollout(); st = h.ppo_update()
        assert st.n_updates == 24
        pr = h.profile()
        if k:
            assert pr["rollout_kernel"]["launches"] == pr["rollout_kernel"]["timed_launches"] == 1 and pr["gae_kernel"]["timed_launches"] == 1
            for name in ("ppo_grad_kernel", "adam_kernel"):
                assert pr[name]["launches"] == 24 and pr[name]["timed_launches"] == 24 // k and pr[name]["timed_ms"] > 0
                assert pr[name]["total_ms"] == pytest.approx(pr[name]["timed_ms"] * k)
            h.profile_reset()
            assert h.profile()["ppo_grad_kernel"] == {"total_ms": 0.0, "launches": 0, "timed_ms": 0.0, "timed_launches": 0}
        else:
            assert all(v["launches"] == 0 for v in pr.values())
        res.append(h.get_params()); h.close()
    assert np.array_equal(res[0], res[1])

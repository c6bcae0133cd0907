"""GPU (-m gpu): the SAC path (include/dril_sac.h) through the C ABI vs the CPU oracle on the same inputs.
Both are driven through the same typed wrapper (dril.jl_amd/sac.py) — the product with prefix dril_sac_, the oracle with orc_sac_.
Tolerances (fp32 arithmetic on both sides; the device contracts with fp32 MFMA in a different summation order):
  layer calls (actions, log-probs, Q)     rtol 2e-5 / atol 2e-5
  losses                                  rel 1e-4   (BASELINE.json north_star's loss tolerance)
  gradients / parameters after k steps    rtol 2e-4 / atol 2e-6
"""
import math

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def make_pair(pkg, E=8, hidden=(32, 32), B=16, cap=4096, act="relu", seed=7, max_steps=200, scaled=False, **alg_kw):
    env = pkg.MountainCarContinuousEnv(max_steps=max_steps) if scaled in ("mountaincar", "mountaincar_scaled") else pkg.PendulumEnv(max_steps=max_steps)
    if scaled is True or scaled == "mountaincar_scaled":
        env = pkg.ScalingWrapperEnv(env)
    alg = pkg.SAC(batch_size=B, buffer_capacity=cap, **alg_kw)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=hidden, activation=act)
    cfg = pkg.make_sac_config(env, E, alg, layer, seed=seed)
    return pkg.SacHandle(cfg), O.sac_oracle(cfg), layer, alg


def init_params(pkg, layer, seed=0, scale_out=30.0):
    ps = layer.initialparameters(np.random.default_rng(seed))
    ps["actor_head"]["layer_3"]["weight"] *= scale_out
    rng = np.random.default_rng(seed + 1)
    for head in (ps["actor_head"], ps["critic_head"]["layer_1"], ps["critic_head"]["layer_2"]):
        for l in head.values():
            l["bias"] = rng.normal(0, 0.1, l["bias"].shape).astype(np.float32)
    ps["log_std"] = np.full_like(ps["log_std"], -1.0)
    return pkg.sac_flatten_params(ps)


def random_replay(rng, n, D=3, A=1, p_term=0.2):
    obs, nobs = rng.uniform(-1, 1, (n, D)).astype(np.float32), rng.uniform(-1, 1, (n, D)).astype(np.float32)
    act = np.tanh(rng.normal(0, 1, (n, A))).astype(np.float32)
    rew = rng.normal(-1, 1, n).astype(np.float32)
    term = (rng.uniform(size=n) < p_term).astype(np.uint8)
    return obs, act, rew, term, np.zeros(n, np.uint8), nobs


def close(a, b, rtol=2e-5, atol=2e-5):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("hidden,act", [((32, 32), "relu"), ((64, 32), "tanh"), ((24, 40), "tanh"), ((512, 512), "relu")])
def test_layer_calls_match_oracle(pkg, hidden, act):
    """action_log_prob / predict_actions / predict_values(obs, actions) incl. target networks; batch sizes ragged w.r.t. the 32-wide tiles"""
    h, o, layer, _ = make_pair(pkg, hidden=hidden, act=act, B=48)
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(1)
    for x in (h, o):
        x.set_params(flat)
        x.set_target_params((flat[layer.parameterlength() - 1 - 2 * layer.q_parameterlength():-1] * 0.9).astype(np.float32))
    for n in (1, 37, 96, 257):
        obs = rng.uniform(-1, 1, (n, 3)).astype(np.float32); nz = rng.normal(0, 1, (n, 1)).astype(np.float32)
        act_in = np.tanh(rng.normal(0, 1, (n, 1))).astype(np.float32)
        a1, l1 = h.action_log_prob(obs, nz); a2, l2 = o.action_log_prob(obs, nz)
        close(a1, a2); close(l1, l2, rtol=1e-4, atol=1e-4)
        for det in (False, True):
            r1, e1 = h.predict_actions(obs, det, nz); r2, e2 = o.predict_actions(obs, det, nz)
            close(r1, r2); close(e1, e2)
        for tgt in (False, True):
            close(h.predict_q(obs, act_in, tgt), o.predict_q(obs, act_in, tgt), rtol=5e-5, atol=5e-5)
    assert h.predict_q(obs, act_in, True)[0, 0] != h.predict_q(obs, act_in, False)[0, 0]
    # Philox noise when none is given: deterministic per handle state, finite, inside the squash range
    a, lp = h.action_log_prob(obs)
    assert np.isfinite(lp).all() and np.abs(a).max() <= 1.0


@pytest.mark.parametrize("act,auto_ent,interval,hidden,B", [("relu", True, 1, (32, 32), 16), ("tanh", True, 2, (64, 32), 40),
                                                            ("relu", False, 1, (32, 32), 16), ("relu", True, 1, (512, 512), 256),
                                                            ("tanh", True, 1, (40, 24), 7), ("relu", True, 1, (24, 40), 9),      # ragged: hidden dims and batch off every 32-wide tile; H2 > H1 and H2 < H1
                                                            ("relu", True, 1, (96, 160), 100), ("tanh", True, 1, (320, 288), 200)])   # the LDS-staged split-K body (K >= 64): one ragged pass (K = 96 / 100 / 160), a full + a ragged pass (K = 320 / 288), ragged tiles in m and n
def test_update_matches_oracle(pkg, act, auto_ent, interval, hidden, B):
    """three consecutive update! steps with injected batches: losses, both gradients, parameters, targets, log_ent_coef"""
    ent = pkg.AutoEntropyCoefficient(initial_value=0.7) if auto_ent else pkg.FixedEntropyCoefficient(0.3)
    h, o, layer, alg = make_pair(pkg, hidden=hidden, B=B, act=act, ent_coef=ent, target_update_interval=interval, learning_rate=3e-3, tau=0.05)
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(3)
    rb = random_replay(rng, 300)
    n_upd = 3
    idx = rng.integers(0, 300, (n_upd, B))
    nz = [rng.normal(0, 1, (n_upd, B, 1)).astype(np.float32) for _ in range(3)]
    for x in (h, o):
        x.set_params(flat); x.replay_fill(*rb)
    for k in range(n_upd):
        st = []
        for x in (h, o):
            x.set_batches(1, idx[k:k + 1], *[z[k:k + 1] for z in nz])
            st.append(x.update(1)[0])
        a, b = st
        for f in ("critic_loss", "actor_loss", "mean_q_values", "grad_norm", "entropy_coefficient"):
            assert getattr(a, f) == pytest.approx(getattr(b, f), rel=1e-4, abs=1e-5), (k, f)
        assert a.has_entropy_loss == b.has_entropy_loss == int(auto_ent)
        if auto_ent:
            assert a.entropy_loss == pytest.approx(b.entropy_loss, rel=1e-4, abs=1e-5)
        (gc1, ga1), (gc2, ga2) = h.last_grads(), o.last_grads()
        scale_c, scale_a = np.abs(gc2).max(), np.abs(ga2).max()
        close(gc1, gc2, rtol=2e-4, atol=2e-6 * max(1.0, scale_c)); close(ga1, ga2, rtol=2e-4, atol=2e-6 * max(1.0, scale_a))
        n_actor = layer.parameterlength() - 1 - 2 * layer.q_parameterlength()
        assert not gc1[:n_actor].any() and not gc1[-1:].any() and not ga1[n_actor:-1].any()       # test/test_sac.jl:282-284,337-341
        close(h.get_params(), o.get_params(), rtol=2e-4, atol=5e-6)
        close(h.get_target_params(), o.get_target_params(), rtol=2e-5, atol=2e-6)
        assert h.get_log_ent_coef() == pytest.approx(o.get_log_ent_coef(), rel=1e-5, abs=1e-6)


@pytest.mark.parametrize("act,hidden,B", [("relu", (512, 512), 256), ("tanh", (40, 24), 7), ("relu", (100, 64), 50)])
def test_first_layer_gradient_inside_the_optimiser_kernels_equals_the_contraction_form(pkg, monkeypatch, act, hidden, B):
    """[dW1 | db1] of the narrow first layers, their Adam step and the critics' re-evaluated first layer come from blocks of sac_adam_kernel / sac_step_end_kernel
    (first_layer_opt_block) by default and from launches of their own under DRIL_SAC_NO_FUSED_DW1=1 (the sequence of rounds 1 - 3): the same sums in a different
    order — gradients, statistics, parameters and targets after four steps agree to summation-order rounding"""
    res = []
    rng = np.random.default_rng(5)
    rb = random_replay(rng, 400)
    idx = rng.integers(0, 400, (4, B)); nz = [rng.normal(0, 1, (4, B, 1)).astype(np.float32) for _ in range(3)]
    for off in (False, True):
        if off:
            monkeypatch.setenv("DRIL_SAC_NO_FUSED_DW1", "1")
        h, _, layer, _ = make_pair(pkg, hidden=hidden, B=B, act=act, learning_rate=3e-3, tau=0.05)
        monkeypatch.delenv("DRIL_SAC_NO_FUSED_DW1", raising=False)
        h.set_params(init_params(pkg, layer)); h.replay_fill(*rb); h.set_batches(4, idx, *nz)
        st = h.update(4)
        res.append((st, h.last_grads(), h.get_params(), h.get_target_params()))
    (s1, (gc1, ga1), p1, t1), (s2, (gc2, ga2), p2, t2) = res
    for a, b in zip(s1, s2):
        for f in ("critic_loss", "actor_loss", "mean_q_values", "grad_norm", "entropy_loss"):
            assert getattr(a, f) == pytest.approx(getattr(b, f), rel=2e-5, abs=1e-6), f
    close(gc1, gc2, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(gc2).max())); close(ga1, ga2, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(ga2).max()))
    assert np.abs(gc2).max() > 0 and np.abs(ga2).max() > 0
    close(p1, p2, rtol=5e-5, atol=2e-6); close(t1, t2, rtol=2e-5, atol=1e-6)


def test_many_updates_in_one_call_and_reset_optimizer(pkg):
    h, o, layer, _ = make_pair(pkg, B=32, learning_rate=1e-3)
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(9)
    rb = random_replay(rng, 128)
    idx = rng.integers(0, 128, (6, 32)); nz = [rng.normal(0, 1, (6, 32, 1)).astype(np.float32) for _ in range(3)]
    for x in (h, o):
        x.set_params(flat); x.replay_fill(*rb); x.set_batches(6, idx, *nz)
    s1, s2 = h.update(6), o.update(6)
    for a, b in zip(s1, s2):
        assert a.critic_loss == pytest.approx(b.critic_loss, rel=2e-4) and a.actor_loss == pytest.approx(b.actor_loss, rel=2e-4, abs=1e-5)
    close(h.get_params(), o.get_params(), rtol=5e-4, atol=1e-5)
    for x in (h, o):
        x.reset_optimizer(); x.set_params(flat); x.set_log_ent_coef(0.0); x.set_batches(1, idx[:1], *[z[:1] for z in nz])
    a, b = h.update(1)[0], o.update(1)[0]
    assert a.critic_loss == pytest.approx(s2[0].critic_loss, rel=1e-4) and b.critic_loss == pytest.approx(s2[0].critic_loss, rel=1e-6)
    close(h.get_params(), o.get_params(), rtol=2e-4, atol=5e-6)


@pytest.mark.parametrize("scaled", [False, True, "mountaincar", "mountaincar_scaled"])      # "mountaincar": MountainCarContinuous-v0 as the device env (D = 2, Box(-1, 1)); "_scaled": under ScalingWrapperEnv
def test_collect_matches_oracle(pkg, scaled):
    """off_policy_collection.jl:28-96 with injected noise: the replay contents agree field by field, through a truncation and a ring wrap;
    scaled = under ScalingWrapperEnv (TanhScaleAdapter then maps onto the wrapper's Box(-1, 1))"""
    E, L = 50, 5
    h, o, layer, _ = make_pair(pkg, E=E, max_steps=L, cap=400, scaled=scaled)
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(2)
    for x in (h, o):
        x.set_params(flat); x.env_reset(11)
    close(h.env_observe(), o.env_observe(), rtol=1e-6, atol=1e-6)
    u = rng.uniform(0, 1, (3, E, 1)).astype(np.float32); nz = rng.normal(0, 1, (7, E, 1)).astype(np.float32)
    for x in (h, o):
        x.set_collect_noise(u); x.collect_rollout(3, True)
        x.set_collect_noise(nz); x.collect_rollout(7, False)                # 10 steps: two truncations, 500 pushes into 400 slots
    assert h.replay_size() == o.replay_size() == 400
    C = pkg._capi
    for which in (C.RB_OBSERVATIONS, C.RB_NEXT_OBSERVATIONS, C.RB_ACTIONS, C.RB_REWARDS):
        close(h.replay(which), o.replay(which), rtol=1e-4, atol=2e-5)
    for which in (C.RB_TERMINATED, C.RB_TRUNCATED):
        np.testing.assert_array_equal(h.replay(which), o.replay(which))
    assert h.replay(C.RB_TRUNCATED).sum() == 2 * E
    close(h.env_observe(), o.env_observe(), rtol=1e-4, atol=2e-5)
    # un-injected noise: the env-keyed Philox stream; same uniforms / normals up to libm ulps
    for x in (h, o):
        x.collect_rollout(2, True); x.collect_rollout(2, False)
    close(h.replay(C.RB_ACTIONS), o.replay(C.RB_ACTIONS), rtol=1e-3, atol=1e-4)
    close(h.replay(C.RB_REWARDS), o.replay(C.RB_REWARDS), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("scaled", [False, True, "mountaincar", "mountaincar_scaled"])
def test_one_launch_collection_equals_the_four_launch_sequence(pkg, monkeypatch, scaled):
    """sac_collect_env_kernel (head + act! + observe + push! per env in one launch) against sac_collect_head_kernel -> env_step_kernel -> env_observe_kernel ->
    sac_push_kernel (DRIL_SAC_NO_FUSED_COLLECT=1, latched at create): same device functions in the same order => every replay field, the observation and the env
    state after truncations, a ring wrap, random and policy actions, injected and Philox noise are bit-identical"""
    E, L = 70, 4
    a, _, layer, _ = make_pair(pkg, E=E, max_steps=L, cap=300, scaled=scaled)
    monkeypatch.setenv("DRIL_SAC_NO_FUSED_COLLECT", "1")
    b, _, _, _ = make_pair(pkg, E=E, max_steps=L, cap=300, scaled=scaled)
    monkeypatch.delenv("DRIL_SAC_NO_FUSED_COLLECT")
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(5)
    nz = rng.normal(0, 1, (3, E, 1)).astype(np.float32)
    for x in (a, b):
        x.set_params(flat); x.env_reset(3)
        x.collect_rollout(2, True); x.set_collect_noise(nz); x.collect_rollout(3, False); x.collect_rollout(4, False)      # 9 steps x 70 envs into 300 slots, two truncations
    C = pkg._capi
    assert a.replay_size() == b.replay_size() == 300
    for which in (C.RB_OBSERVATIONS, C.RB_NEXT_OBSERVATIONS, C.RB_ACTIONS, C.RB_REWARDS, C.RB_TERMINATED, C.RB_TRUNCATED):
        np.testing.assert_array_equal(a.replay(which), b.replay(which))
    np.testing.assert_array_equal(a.env_observe(), b.env_observe())
    assert a.replay(C.RB_TRUNCATED).sum() > 0


@pytest.mark.parametrize("E,hidden,act", [(4096, (512, 512), "relu"), (4100, (512, 512), "tanh"), (16400, (96, 160), "relu"), (2050, (128, 256), "relu")])
def test_fused_collection_forward_matches_the_three_contraction_form_and_the_oracle(pkg, monkeypatch, E, hidden, act):
    """the collection forward of configs[4]-sized env counts, three forms inside the library and the oracle's collection (off_policy_collection.jl:28-96):
      a  default: first layer as an elementwise pass that also cuts h1 into two f16 planes, the second layer on the f16 matrix cores from pre-split operands
         (sac_collect_l2_kernel; widths that are multiples of its tiles: (512, 512), (128, 256)), the output layer inside the env kernel;
      b  DRIL_SAC_NO_F16_FWD=1: the same launch structure with the second layer as the generic f32-MFMA contraction;
      c  DRIL_SAC_NO_FUSED_FWD=1: three generic contractions and mu through memory (rounds 1 - 3).
    Same fp32-equivalent products in other summation orders: replay contents agree to rounding, incl. an env count that is no multiple of the tile sizes"""
    a, o, layer, _ = make_pair(pkg, E=E, hidden=hidden, act=act, cap=4 * E, max_steps=3)
    monkeypatch.setenv("DRIL_SAC_NO_F16_FWD", "1")
    b, _, _, _ = make_pair(pkg, E=E, hidden=hidden, act=act, cap=4 * E, max_steps=3)
    monkeypatch.delenv("DRIL_SAC_NO_F16_FWD")
    monkeypatch.setenv("DRIL_SAC_NO_FUSED_FWD", "1")
    c, _, _, _ = make_pair(pkg, E=E, hidden=hidden, act=act, cap=4 * E, max_steps=3)
    monkeypatch.delenv("DRIL_SAC_NO_FUSED_FWD")
    flat = init_params(pkg, layer, scale_out=3.0)
    nz = np.random.default_rng(5).normal(0, 1, (3, E, 1)).astype(np.float32)
    for x in (a, b, c, o):
        x.set_params(flat); x.env_reset(3); x.set_collect_noise(nz); x.collect_rollout(3, False)      # three policy steps, one truncation
    C = pkg._capi
    for which in (C.RB_TERMINATED, C.RB_TRUNCATED):
        for x in (b, c, o):
            np.testing.assert_array_equal(a.replay(which), x.replay(which))
    for which in (C.RB_OBSERVATIONS, C.RB_NEXT_OBSERVATIONS, C.RB_ACTIONS, C.RB_REWARDS):
        close(a.replay(which), b.replay(which), rtol=1e-4, atol=2e-5)
        close(a.replay(which), c.replay(which), rtol=1e-4, atol=2e-5)
        close(a.replay(which), o.replay(which), rtol=1e-4, atol=5e-5)
    da = np.abs(a.replay(C.RB_ACTIONS).astype(np.float64) - b.replay(C.RB_ACTIONS)).max()
    print(f"[collection forward] E {E} hidden {hidden}: max |action(f16 pieces) - action(f32 MFMA)| = {da:.2e}")
    assert da <= 5e-7                                                 # fp32-equivalent: three steps of closed-loop dynamics apart by rounding only (measured 0 - 1.2e-7; hi.hi alone: 2.7e-6, the negative control below)
    assert a.replay(C.RB_TRUNCATED).sum() == E and np.abs(a.replay(C.RB_ACTIONS)).max() < 1.0 and np.std(a.replay(C.RB_ACTIONS)) > 0.1
    for x in (a, b, c):
        x.close()


def test_negative_control_dropping_the_lo_products_breaks_the_collection_forward(pkg):
    """the agreement test above has power: the same comparison (f16-piece second layer vs f32-MFMA second layer, three closed-loop steps) on libdril_hip_droplo.so —
    mfma_split3 without its two `lo` products: an 11-bit product — must miss the 5e-7 bound"""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    so = root / "dril.jl_amd" / "csrc" / "libdril_hip_droplo.so"
    assert so.exists(), f"{so} missing: __graft_entry__.build() compiles it"
    code = (
        "import sys, json, os, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r}); sys.path.insert(0, {str(root / 'tests')!r})\n"
        "import __graft_entry__ as g; pkg = g.load_package()\n"
        "import test_gpu_sac as T\n"
        "res = []\n"
        "for flag in (None, '1'):\n"
        "    if flag: os.environ['DRIL_SAC_NO_F16_FWD'] = flag\n"
        "    h, _, layer, _ = T.make_pair(pkg, E=4096, hidden=(512, 512), act='relu', cap=16384, max_steps=3)\n"
        "    os.environ.pop('DRIL_SAC_NO_F16_FWD', None)\n"
        "    h.set_params(T.init_params(pkg, layer, scale_out=3.0)); h.env_reset(3)\n"
        "    h.set_collect_noise(np.random.default_rng(5).normal(0, 1, (3, 4096, 1)).astype(np.float32)); h.collect_rollout(3, False)\n"
        "    res.append(h.replay(pkg._capi.RB_ACTIONS).astype(np.float64))\n"
        "print(json.dumps(float(np.abs(res[0] - res[1]).max())))\n")
    env = dict(os.environ, DRIL_HIP_LIBRARY=str(so))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"[negative control] max |action(hi.hi only) - action(f32 MFMA)| = {d:.2e}")
    assert d > 1e-6                                                    # 5e-7 is the bound of the real arithmetic (measured 6e-8 - 1.2e-7); an 11-bit product gives 2.7e-6 — the rounding errors of
                                                                       # 512-term sums average out, so the control sits 20 - 45 x above the product, not 2^13 x


@pytest.mark.parametrize("what", ["h1", "w2", "nan"])
def test_collection_forward_outside_the_f16_range_falls_back_inside_the_kernel(pkg, monkeypatch, what):
    """f16 pieces overflow beyond 65 504 / scale: a relu activation >= 4 094 (first-layer bias 5 000: every h1 is) or an actor W2 entry >= 1 023.  The producers flag it
    on the device, sac_collect_l2_kernel computes its block with f32 MFMAs from the f32 operands instead — no host round trip, and never a silently wrong action:
    the replay equals the f32 form's.  A NaN weight must surface as NaN in both forms, not be laundered."""
    E, hidden = 4096, (512, 512)
    a, _, layer, _ = make_pair(pkg, E=E, hidden=hidden, act="relu", cap=2 * E, max_steps=50)
    monkeypatch.setenv("DRIL_SAC_NO_F16_FWD", "1")
    b, _, _, _ = make_pair(pkg, E=E, hidden=hidden, act="relu", cap=2 * E, max_steps=50)
    monkeypatch.delenv("DRIL_SAC_NO_F16_FWD")
    ps = layer.initialparameters(np.random.default_rng(0))
    if what == "h1":
        ps["actor_head"]["layer_1"]["bias"][:] = 5000.0
        ps["actor_head"]["layer_2"]["weight"] *= 1e-3                  # keep h2 (and the sampled action) in a sane range
    elif what == "w2":
        ps["actor_head"]["layer_2"]["weight"][7, 11] = 2000.0
        ps["actor_head"]["layer_3"]["weight"] *= 1e-3
    else:
        ps["actor_head"]["layer_2"]["weight"][3, 5] = np.nan
    flat = pkg.sac_flatten_params(ps)
    nz = np.random.default_rng(5).normal(0, 1, (2, E, 1)).astype(np.float32)
    for x in (a, b):
        x.set_params(flat); x.env_reset(3); x.set_collect_noise(nz); x.collect_rollout(2, False)
    C = pkg._capi
    ra, rb = a.replay(C.RB_ACTIONS), b.replay(C.RB_ACTIONS)
    if what == "nan":
        assert np.isnan(ra).all() and np.isnan(rb).all()
    else:
        assert np.isfinite(ra).all()
        close(ra, rb, rtol=1e-5, atol=2e-6)
        close(a.replay(C.RB_REWARDS), b.replay(C.RB_REWARDS), rtol=1e-5, atol=1e-5)
    # back in range: the very next step runs the f16 form again (the flags are tagged per launch, nothing to clear)
    flat2 = init_params(pkg, layer, scale_out=3.0)
    for x in (a, b):
        x.set_params(flat2); x.env_reset(4); x.set_collect_noise(nz); x.collect_rollout(2, False)
    close(a.replay(C.RB_ACTIONS), b.replay(C.RB_ACTIONS), rtol=1e-4, atol=2e-5)
    assert np.isfinite(a.replay(C.RB_ACTIONS)).all()
    a.close(); b.close()


def test_iterations_without_host_sync_equal_the_step_by_step_loop(pkg):
    """dril_sac_iterate (train!'s loop body enqueued back to back, one drain per 64 iterations; dril_sac_train's own loop) against collect_rollout(train_freq) +
    update(n) per iteration with a drain after each: the same launches in the same order => parameters, targets, entropy coefficient, replay contents and the
    statistics of every gradient step are bit-identical; 70 iterations = one full chunk and a partial one"""
    E = 24
    a, _, layer, _ = make_pair(pkg, E=E, max_steps=9, cap=1000)
    b, _, _, _ = make_pair(pkg, E=E, max_steps=9, cap=1000)
    flat = init_params(pkg, layer)
    for x in (a, b):
        x.set_params(flat); x.env_reset(4); x.collect_rollout(3, True)
    n_it = 70
    sa, fps = a.iterate(n_it)
    sb = []
    for _ in range(n_it):
        b.collect_rollout(b.cfg.train_freq, False); sb += b.update(1)
    assert len(sa) == len(sb) == n_it and len(fps) == n_it and (fps > 0).all()
    for x, y in zip(sa, sb):
        assert (x.actor_loss, x.critic_loss, x.entropy_coefficient, x.mean_q_values, x.grad_norm) == (y.actor_loss, y.critic_loss, y.entropy_coefficient, y.mean_q_values, y.grad_norm)
    np.testing.assert_array_equal(a.get_params(), b.get_params())
    np.testing.assert_array_equal(a.get_target_params(), b.get_target_params())
    assert a.get_log_ent_coef() == b.get_log_ent_coef()
    C = pkg._capi
    for which in (C.RB_OBSERVATIONS, C.RB_NEXT_OBSERVATIONS, C.RB_ACTIONS, C.RB_REWARDS, C.RB_TERMINATED, C.RB_TRUNCATED):
        np.testing.assert_array_equal(a.replay(which), b.replay(which))
    assert not np.array_equal(a.get_params(), flat)


def test_replay_fill_copy_out_and_errors(pkg):
    h, _, layer, _ = make_pair(pkg, cap=64)
    rng = np.random.default_rng(0)
    rb = random_replay(rng, 100)                                            # 100 pushes into 64 slots: the last 64 survive
    h.replay_fill(*rb)
    assert h.replay_size() == 64
    np.testing.assert_array_equal(h.replay(pkg._capi.RB_REWARDS), rb[2][36:])
    np.testing.assert_array_equal(h.replay(pkg._capi.RB_NEXT_OBSERVATIONS), rb[5][36:])
    with pytest.raises(pkg.DrilError):
        h.set_batches(1, np.full((1, 16), 64))                              # index out of range
    with pytest.raises(pkg.DrilError):
        h.collect_rollout(1)                                                # env not reset
    h2, _, _, _ = make_pair(pkg)
    h2.set_params(init_params(pkg, layer))
    with pytest.raises(pkg.DrilError):
        h2.update(1)                                                        # empty replay buffer


def test_train_loop_matches_oracle_statistically(pkg):
    """train! sac.jl:414-549 with device Philox streams on both sides: identical schedule; losses track the oracle (noise differs by libm ulps,
    so the comparison is on the first updates tightly and on the tail loosely)"""
    E = 16
    kw = dict(E=E, B=64, cap=4096, start_steps=128, train_freq=2, gradient_steps=2, learning_rate=1e-3)
    h, o, layer, alg = make_pair(pkg, **kw)
    flat = init_params(pkg, layer, scale_out=1.0)
    res = []
    for x in (h, o):
        x.set_params(flat); x.env_reset(5)
        res.append(x.train(128 + 10 * 2 * E))
    (s1, f1, n1, i1, t1), (s2, f2, n2, i2, t2) = res
    assert (n1, i1, t1) == (n2, i2, t2) == (22, 11, 128 + 20 * E)
    assert h.replay_size() == o.replay_size() == t1
    assert len(f1) == 11 and (f1 > 0).all()
    assert s1[0].critic_loss == pytest.approx(s2[0].critic_loss, rel=1e-3)
    assert s1[0].actor_loss == pytest.approx(s2[0].actor_loss, rel=1e-3, abs=1e-4)
    a = np.array([[s.critic_loss, s.actor_loss, s.entropy_coefficient, s.mean_q_values] for s in s1])
    b = np.array([[s.critic_loss, s.actor_loss, s.entropy_coefficient, s.mean_q_values] for s in s2])
    assert np.isfinite(a).all()
    np.testing.assert_allclose(a, b, rtol=0.05, atol=0.02)
    close(h.get_params(), o.get_params(), rtol=5e-2, atol=2e-3)


def test_host_mirror_and_config5_shape(pkg):
    """SAC / SACLayer / SACAgent / ReplayBuffer / sac_train_ (the reference's user-facing surface) at BASELINE.json configs[4] shape:
    Pendulum, 4096 envs, SACLayer [512, 512] relu, batch 256"""
    env = pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=200), 4096, seed=42)
    alg = pkg.SAC(start_steps=4096 * 2, buffer_capacity=100_000)
    layer = pkg.SACLayer(env.observation_space(), env.action_space())
    agent = pkg.SACAgent(layer, alg, seed=1)
    assert layer.parameterlength() == (3 * 512 + 512 + 512 * 512 + 512 + 512 + 1) + 2 * (4 * 512 + 512 + 512 * 512 + 512 + 512 + 1) + 1
    before = pkg.sac_flatten_params(agent.parameters).copy()
    agent, rb, stats, timer = pkg.sac_train_(agent, env, alg, 4096 * 2 + 30 * 4096)
    assert len(rb) == 100_000 and rb.isfull()                               # 32 steps x 4096 envs pushed into 100 000 slots
    assert len(stats["critic_losses"]) == 31 and agent.gradient_updates == 31 and agent.steps_taken == 32 * 4096
    assert np.isfinite(stats["critic_losses"]).all() and np.isfinite(stats["actor_losses"]).all()
    after = pkg.sac_flatten_params(agent.parameters)
    assert np.isfinite(after).all() and not np.array_equal(before, after)
    # log_std_init = -3: log-probs far above -target_entropy, so c > 0 and the coefficient climbs (sac.jl:326-330)
    assert 1.0 < stats["entropy_coefficients"][-1] < 1.1
    obs = rb.observations
    np.testing.assert_allclose(np.hypot(obs[:, 0], obs[:, 1]), 1.0, rtol=1e-5)   # (cos, sin, theta_dot)
    assert np.abs(rb.actions).max() <= 2.0


def test_sac_callbacks_run_where_the_reference_runs_them(pkg):
    """train!(...; callbacks) for SAC (sac.jl:476-552, off_policy_collection.jl:44-49): the five hooks fire in the reference's places with its locals, the
    step-by-step loop they need is bit-identical to the sync-free loop of the callback-less train, and a false return stops the training with the reference's
    early-return shape (no timer) and the partially trained weights in the agent"""
    E = 64

    def fresh():
        env = pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=200), E, seed=7)
        alg = pkg.SAC(start_steps=3 * E, buffer_capacity=10_000, batch_size=64)
        layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(64, 64))
        return env, alg, pkg.SACAgent(layer, alg, seed=2)

    class Rec:
        def __init__(self, stop_at=None):
            self.n = dict(training_start=0, rollout_start=0, step=0, rollout_end=0, training_end=0); self.stop_at = stop_at; self.seen = {}
        def on_training_start(self, loc):
            self.n["training_start"] += 1; self.seen["start"] = set(loc); return True
        def on_rollout_start(self, loc):
            self.n["rollout_start"] += 1; return self.stop_at is None or loc["training_iteration"] < self.stop_at
        def on_step(self, loc):
            self.n["step"] += 1; self.seen["step"] = set(loc); return True
        def on_rollout_end(self, loc):
            self.n["rollout_end"] += 1; assert loc["fps"] > 0; return True
        def on_training_end(self, loc):
            self.n["training_end"] += 1; return True

    max_steps = 3 * E + 9 * E                                     # 3 start steps, then 9 more iterations of train_freq = 1
    env, alg, agent = fresh(); ref_agent, ref_rb, ref_stats, _ = pkg.sac_train_(agent, env, alg, max_steps)
    env, alg, agent = fresh(); cb = Rec()
    out = pkg.sac_train_(agent, env, alg, max_steps, callbacks=[cb])
    assert len(out) == 4
    agent, rb, stats, timer = out
    assert cb.n == dict(training_start=1, rollout_start=10, step=3 + 9, rollout_end=10, training_end=1) and timer["iterations"] == 10
    assert {"agent", "replay_buffer", "env", "alg", "max_steps", "callbacks", "n_envs", "iterations", "total_steps", "n_steps", "training_stats"} <= cb.seen["start"]
    assert {"i", "use_random_actions", "training_iteration"} <= cb.seen["step"]
    np.testing.assert_array_equal(pkg.sac_flatten_params(agent.parameters), pkg.sac_flatten_params(ref_agent.parameters))
    assert stats["critic_losses"] == ref_stats["critic_losses"] and len(stats["fps"]) == 10
    assert agent.steps_taken == ref_agent.steps_taken == 12 * E and agent.gradient_updates == ref_agent.gradient_updates == 10
    np.testing.assert_array_equal(rb.rewards, ref_rb.rewards)
    # a hook that says no at the start of iteration 4
    env, alg, agent = fresh(); before = pkg.sac_flatten_params(agent.parameters).copy(); cb = Rec(stop_at=4)
    out = pkg.sac_train_(agent, env, alg, max_steps, callbacks=[cb])
    assert len(out) == 3 and cb.n["rollout_start"] == 4 and cb.n["rollout_end"] == 3 and cb.n["training_end"] == 0
    assert len(out[2]["critic_losses"]) == 3 and not np.array_equal(pkg.sac_flatten_params(out[0].parameters), before)


def test_critic_learns_fixed_targets(pkg):
    """learning signal: on a fixed replay with terminated transitions only (target = reward), the critic loss falls by > 10x"""
    h, _, layer, _ = make_pair(pkg, hidden=(64, 64), B=128, learning_rate=3e-3, ent_coef=pkg.FixedEntropyCoefficient(0.01))
    h.set_params(init_params(pkg, layer, scale_out=1.0))
    rng = np.random.default_rng(4)
    obs, act, _, _, trunc, nobs = random_replay(rng, 512)
    rew = (obs[:, 0] * 2 - act[:, 0]).astype(np.float32)
    h.replay_fill(obs, act, rew, np.ones(512, np.uint8), trunc, nobs)
    losses = [s.critic_loss for s in h.update(300)]
    assert losses[-1] < 0.1 * losses[0]


def test_training_is_bitwise_reproducible(pkg):
    """two fresh handles with the same seeds: identical parameters, targets, replay contents and statistics bit for bit (every reduction in the
    path sums in a fixed order; the only atomic elects the block that finalises the statistics)"""
    runs = []
    for _ in range(2):
        env = pkg.PendulumEnv(max_steps=200)
        alg = pkg.SAC(batch_size=256, buffer_capacity=50_000, start_steps=1024)
        layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(128, 128))
        h = pkg.SacHandle(pkg.make_sac_config(env, 512, alg, layer, seed=3))
        h.set_params(pkg.sac_flatten_params(layer.initialparameters(np.random.default_rng(5)))); h.env_reset(3)
        stats, fps, n_upd, iters, total = h.train(1024 + 40 * 512)
        runs.append((h.get_params(), h.get_target_params(), h.replay(pkg._capi.RB_REWARDS), h.replay(pkg._capi.RB_ACTIONS),
                     np.array([[s.critic_loss, s.actor_loss, s.grad_norm] for s in stats]), n_upd))
    a, b = runs
    assert a[5] == b[5] == 41
    for x, y in zip(a[:5], b[:5]):
        np.testing.assert_array_equal(x, y)
    assert np.isfinite(a[4]).all()


# ---- DRIL_ENV_EXTERNAL: the caller's own host envs, any observation width, up to 16 action dimensions --------------------------------------
class _HostSpaces:
    def __init__(self, pkg, D, A, low=-1.0, high=1.0):
        self.kind, self._o, self._a = pkg._capi.ENV_EXTERNAL, pkg.Box(low=(-10.0,) * D, high=(10.0,) * D), pkg.Box(low=(low,) * A, high=(high,) * A)

    def observation_space(self):
        return self._o

    def action_space(self):
        return self._a


def make_ext_pair(pkg, D, A, E=8, hidden=(32, 32), B=16, cap=4096, act="relu", low=-1.0, high=1.0, **alg_kw):
    env = _HostSpaces(pkg, D, A, low, high)
    alg = pkg.SAC(batch_size=B, buffer_capacity=cap, **alg_kw)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=hidden, activation=act)
    cfg = pkg.make_sac_config(env, E, alg, layer, seed=7)
    return pkg.SacHandle(cfg), O.sac_oracle(cfg), layer, alg


@pytest.mark.parametrize("D,A,hidden,B,act", [(11, 5, (24, 40), 16, "tanh"), (17, 6, (256, 256), 256, "relu"), (1, 16, (32, 32), 33, "relu"), (376, 16, (64, 64), 64, "relu")])
def test_external_spaces_layer_calls_and_update(pkg, D, A, hidden, B, act):
    """layer calls and three update! steps for spaces the Pendulum kind does not have (multi-dimensional squashed Gaussian, wide observations)"""
    h, o, layer, alg = make_ext_pair(pkg, D, A, hidden=hidden, B=B, act=act, ent_coef=pkg.AutoEntropyCoefficient(initial_value=0.5), learning_rate=3e-3, tau=0.05,
                                    low=-0.5, high=1.5)
    assert (h.D, h.A, h.P) == (D, A, layer.parameterlength())
    flat = init_params(pkg, layer, scale_out=10.0)
    rng = np.random.default_rng(9)
    rb = random_replay(rng, 300, D, A)
    for x in (h, o):
        x.set_params(flat); x.replay_fill(*rb)
    obs = rng.uniform(-1, 1, (70, D)).astype(np.float32); nz = rng.standard_normal((70, A)).astype(np.float32)
    (a1, l1), (a2, l2) = h.action_log_prob(obs, nz), o.action_log_prob(obs, nz)
    close(a1, a2); close(l1, l2, rtol=1e-4, atol=1e-4 * A)
    (r1, e1), (r2, e2) = h.predict_actions(obs, False, nz), o.predict_actions(obs, False, nz)
    close(r1, r2); close(e1, e2)
    assert e1.min() >= -0.5 - 1e-6 and e1.max() <= 1.5 + 1e-6                          # to_env(TanhScaleAdapter) onto Box(-0.5, 1.5), default_adapters.jl:13-21
    close(h.predict_actions(obs, True)[1], o.predict_actions(obs, True)[1])
    close(h.predict_q(obs, a2), o.predict_q(obs, a2), rtol=1e-4, atol=1e-4)
    n_upd = 3
    idx = rng.integers(0, 300, (n_upd, B)); nzs = [rng.normal(0, 1, (n_upd, B, A)).astype(np.float32) for _ in range(3)]
    for k in range(n_upd):
        st = []
        for x in (h, o):
            x.set_batches(1, idx[k:k + 1], *[z[k:k + 1] for z in nzs])
            st.append(x.update(1)[0])
        a, b = st
        for f in ("critic_loss", "actor_loss", "entropy_loss", "mean_q_values", "grad_norm", "entropy_coefficient"):
            assert getattr(a, f) == pytest.approx(getattr(b, f), rel=2e-4, abs=2e-5), (k, f)
        # Adam turns a gradient entry near zero into a step of up to lr = 3e-3 in either direction: a handful of the 2e5 parameters may differ by a few 1e-5
        close(h.get_params(), o.get_params(), rtol=3e-4, atol=1e-4)
        close(h.get_target_params(), o.get_target_params(), rtol=3e-5, atol=1e-5)


def test_external_push_matches_oracle_and_protocol(pkg):
    """dril_sac_ext_push: same ring as the oracle's after wrapping pushes with truncations; device-env verbs refuse an external handle"""
    capi = pkg._capi
    E, D, A = 5, 7, 3
    h, o, layer, alg = make_ext_pair(pkg, D, A, E=E, cap=23)
    rng = np.random.default_rng(0)
    for t in range(9):
        obs, nobs, tobs = (rng.standard_normal((E, D)).astype(np.float32) for _ in range(3))
        act = rng.uniform(-1, 1, (E, A)).astype(np.float32); rew = rng.standard_normal(E).astype(np.float32)
        term = (rng.random(E) < 0.2); trunc = (rng.random(E) < 0.3) if t % 2 else np.zeros(E, bool)
        for x in (h, o):
            x.ext_push(obs, act, rew, term, trunc, nobs, tobs if trunc.any() else None)
    assert h.replay_size() == o.replay_size() == 23
    for which in (capi.RB_OBSERVATIONS, capi.RB_ACTIONS, capi.RB_REWARDS, capi.RB_TERMINATED, capi.RB_TRUNCATED, capi.RB_NEXT_OBSERVATIONS):
        assert np.array_equal(h.replay(which), o.replay(which)), which
    z = np.zeros(E, np.float32)
    with pytest.raises(pkg.DrilError):
        h.ext_push(np.zeros((E, D)), np.zeros((E, A)), z, z, np.ones(E), np.zeros((E, D)))       # truncated without terminal_obs
    for call in (lambda: h.env_reset(1), h.env_observe, lambda: h.collect_rollout(1), lambda: h.train(100)):
        with pytest.raises(pkg.DrilError) as e:
            call()
        assert e.value.code == capi.ERR_UNSUPPORTED
    dev, _, _, _ = make_pair(pkg)
    with pytest.raises(pkg.DrilError):
        dev.ext_push(np.zeros((8, 3)), np.zeros((8, 1)), np.zeros(8), np.zeros(8), np.zeros(8), np.zeros((8, 3)))   # device-env handle


class _PyPointEnv:
    """host env with the reference's AbstractEnv verbs: 2-D point, 6-dim observation, Box(-1,1)^2 action, reward = -distance to the origin"""

    def __init__(self, pkg, seed):
        self.pkg, self.rng, self.limit = pkg, np.random.default_rng(seed), 25
        self.reset_()

    def observation_space(self):
        return self.pkg.Box(low=(-4.0,) * 6, high=(4.0,) * 6)

    def action_space(self):
        return self.pkg.Box(low=(-1.0, -1.0), high=(1.0, 1.0))

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


def test_sac_trains_on_host_envs(pkg):
    """train!(agent, replay_buffer, env, alg::SAC, max_steps) over the caller's Python envs (HostParallelEnv): schedule counts as in sac.jl:456-466,
    transitions of the random start phase are stored in env space, the mean reward per step improves"""
    E = 8
    env = pkg.HostParallelEnv([_PyPointEnv(pkg, s) for s in range(E)], seed=0)
    alg = pkg.SAC(start_steps=400, buffer_capacity=20000, batch_size=128, gradient_steps=4, learning_rate=1e-3)
    agent = pkg.SACAgent(pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(64, 64)), alg, seed=0)
    agent, rb, stats, timer = pkg.sac_train_(agent, env, alg, 6000)
    iters = (6000 - 400) // E + 1
    assert timer["iterations"] == iters and agent.steps_taken == 400 + E * (iters - 1) and agent.gradient_updates == 4 * iters == len(stats["critic_losses"])
    r = rb.rewards
    assert len(r) == agent.steps_taken and np.isfinite(stats["critic_losses"]).all()
    assert r[-800:].mean() > r[:400].mean() + 0.5, (r[:400].mean(), r[-800:].mean())
    assert np.abs(rb.actions[:400]).max() <= 1.0 and len(stats["entropy_losses"]) == 4 * iters


def test_sac_callbacks_on_host_envs(pkg):
    """the same five hooks over the caller's Python envs (HostParallelEnv): counts per the schedule of sac.jl:456-466; a false on_step in the middle of a
    collection ends the training with the reference's early-return shape and the counters / weights of what was done"""
    E = 4
    calls = dict(training_start=0, rollout_start=0, step=0, rollout_end=0, training_end=0)

    class Cb:
        def __init__(self, stop_step=None): self.stop_step = stop_step
        def on_training_start(self, loc): calls["training_start"] += 1; return True
        def on_rollout_start(self, loc): calls["rollout_start"] += 1; return True
        def on_step(self, loc): calls["step"] += 1; return self.stop_step is None or calls["step"] < self.stop_step
        def on_rollout_end(self, loc): calls["rollout_end"] += 1; return True
        def on_training_end(self, loc): calls["training_end"] += 1; return True

    def run(cb, max_steps=200):
        env = pkg.HostParallelEnv([_PyPointEnv(pkg, s) for s in range(E)], seed=0)
        alg = pkg.SAC(start_steps=40, buffer_capacity=5000, batch_size=32, gradient_steps=1)
        agent = pkg.SACAgent(pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(32, 32)), alg, seed=0)
        return pkg.sac_train_(agent, env, alg, max_steps, callbacks=[cb])

    out = run(Cb())
    iters = (200 - 40) // E + 1
    assert len(out) == 4 and out[3]["iterations"] == iters
    assert calls == dict(training_start=1, rollout_start=iters, step=10 + (iters - 1), rollout_end=iters, training_end=1)
    for k in calls: calls[k] = 0
    out = run(Cb(stop_step=14))                                              # the 14th env step = the 4th iteration after the 10-step start phase
    assert len(out) == 3 and calls["step"] == 14 and calls["rollout_end"] == 4 and calls["training_end"] == 0
    assert out[0].steps_taken == 13 * E and out[0].gradient_updates == 4 == len(out[2]["critic_losses"])

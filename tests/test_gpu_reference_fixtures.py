"""GPU (-m gpu): the reference's OWN end-to-end fixtures for the loss + update path, re-expressed on the device path with the
reference's hyper-parameters and thresholds (VERDICT r1 item 4 — the only reference-held tests that touch loss, gradient, clip
and Adam together):

  * test/test_ppo_integration.jl:1-40   "PPO smoke test improves tracking env performance": TrackingTargetEnv
    (test/test_shared_setup.jl:119-173), 4 envs, hidden [64,64], PPO(n_steps 64, batch_size 32, epochs 10, lr 3e-3), 50 iterations:
    baseline mean reward per step < 0.6; trained > baseline + 0.2 and > 0.75
  * test/test_ppo_integration.jl:42-83  "PPO agent serialization roundtrip": train -> save -> fresh agent -> load -> identical
    parameters and identical deterministic actions; plus (advisor, round 1) the optimiser really is fresh after a load
  * test/test_normalize_wrapper.jl:141-249  un-normalise round trips (obs 1e-5, rewards 1e-4)

The envs below restate the reference's TEST envs in Python (fixtures, not product code); nothing of the reference travels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class TrackingTargetEnv:
    """test/test_shared_setup.jl:119-173: observation u ~ U(0,1) redrawn every step, action Box(-1,1), reward = clamp(1 - |clamp(a) - obs|, 0, 1),
    terminated after max_steps (16) steps, never truncated — the optimal action equals the observed target"""

    def __init__(self, pkg, max_steps=16, seed=0):
        self.pkg, self.max_steps, self.rng = pkg, max_steps, np.random.default_rng(seed)
        self.current_step, self._terminated = 0, False
        self.current_obs = np.float32(self.rng.random())

    def observation_space(self):
        return self.pkg.Box(low=[0.0], high=[1.0])

    def action_space(self):
        return self.pkg.Box(low=[-1.0], high=[1.0])

    def reset_(self):
        self.current_step, self._terminated = 0, False
        self.current_obs = np.float32(self.rng.random())

    def act_(self, action):
        self.current_step += 1
        a = np.clip(np.float32(np.asarray(action).reshape(-1)[0]), -1.0, 1.0)
        reward = float(np.clip(1.0 - abs(a - self.current_obs), 0.0, 1.0))
        self._terminated = self.current_step >= self.max_steps
        self.current_obs = np.float32(self.rng.random())
        return reward

    def observe(self):
        return np.array([self.current_obs], np.float32)

    def terminated(self):
        return self._terminated

    def truncated(self):
        return False


def _parallel(pkg, seed, n_envs):
    return pkg.HostParallelEnv([TrackingTargetEnv(pkg, 16, seed + i + 1) for i in range(n_envs)], seed=seed)


@pytest.mark.parametrize("agent_seed", [42, 7, 2024])
def test_ppo_improves_tracking_env_reference_thresholds(pkg, agent_seed):
    """test_ppo_integration.jl:1-40 with its exact hyper-parameters and thresholds; three initialisation seeds (the reference's comment:
    'min ~0.88, mean ~0.94' across seeds)"""
    n_envs = 4
    train_env, baseline_env, eval_env = _parallel(pkg, 1234, n_envs), _parallel(pkg, 9999, n_envs), _parallel(pkg, 9999, n_envs)
    policy = pkg.ActorCriticLayer(train_env.observation_space(), train_env.action_space(), hidden_dims=(64, 64))
    alg = pkg.PPO(n_steps=64, batch_size=32, epochs=10, learning_rate=3e-3)
    agent = pkg.Agent(policy, alg, seed=agent_seed)
    base = pkg.evaluate_agent(agent, baseline_env, n_eval_episodes=64, deterministic=True)
    baseline_mean_step = base["mean_reward"] / base["mean_length"]
    assert baseline_mean_step < 0.6                                              # "Random policy ~0.5"
    max_steps = alg.n_steps * n_envs * 50
    stats, _ = pkg.train_(agent, train_env, alg, max_steps)
    assert len(stats["losses"]) == 50 and np.isfinite(stats["losses"]).all() and agent.steps_taken == max_steps
    assert agent.gradient_updates == 50 * 10 * (64 * n_envs // 32)               # every minibatch applied (no target_kl)
    trained = pkg.evaluate_agent(agent, eval_env, n_eval_episodes=64, deterministic=True)
    trained_mean_step = trained["mean_reward"] / trained["mean_length"]
    print(f"[tracking] seed {agent_seed}: baseline {baseline_mean_step:.3f} -> trained {trained_mean_step:.3f} per step")
    assert trained["mean_length"] == 16
    assert trained_mean_step > baseline_mean_step + 0.2                         # "Significant improvement"
    assert trained_mean_step > 0.75                                             # "Well above random policy"


def test_ppo_agent_serialization_roundtrip(pkg, tmp_path):
    """test_ppo_integration.jl:42-83 on a device-trained agent: 2 envs, hidden [32,32], PPO(n_steps 16, batch 16, epochs 2, lr 5e-4), 10 iterations,
    save -> new agent (other init seed) -> load: parameters identical, deterministic actions on 5 observations identical"""
    env = _parallel(pkg, 321, 2)
    osp, asp = env.observation_space(), env.action_space()
    alg = pkg.PPO(n_steps=16, batch_size=16, epochs=2, learning_rate=5e-4)
    agent = pkg.Agent(pkg.ActorCriticLayer(osp, asp, hidden_dims=(32, 32)), alg, seed=77)
    before = pkg.flatten_params(agent.train_state.parameters).copy()
    assert pkg.train_(agent, env, alg, alg.n_steps * 2 * 10) is not None
    assert not np.array_equal(before, pkg.flatten_params(agent.train_state.parameters))
    saved = pkg.save_policy_params_and_state(agent, tmp_path / "ppo_agent")
    new_agent = pkg.Agent(pkg.ActorCriticLayer(osp, asp, hidden_dims=(32, 32)), alg, seed=88)
    assert not np.array_equal(pkg.flatten_params(new_agent.train_state.parameters), pkg.flatten_params(agent.train_state.parameters))
    pkg.load_policy_params_and_state_(new_agent, alg, saved)
    np.testing.assert_array_equal(pkg.flatten_params(new_agent.train_state.parameters), pkg.flatten_params(agent.train_state.parameters))
    obs = np.random.default_rng(42).random((5, 1)).astype(np.float32)            # rand(MersenneTwister(42), obs_space) x 5
    h = env.bind(alg, agent.layer)
    h.set_params(pkg.flatten_params(agent.train_state.parameters)); original = h.predict_actions(obs, True)
    h.set_params(pkg.flatten_params(new_agent.train_state.parameters)); loaded = h.predict_actions(obs, True)
    np.testing.assert_array_equal(original, loaded)
    assert np.isfinite(original).all() and original.shape == (5, 1)


@pytest.mark.parametrize("kind", ["cartpole_fused", "pendulum_wide"])
def test_device_env_checkpoint_and_fresh_optimizer_after_load(pkg, oracle_mod, tmp_path, kind):
    """The same round trip on DEVICE envs (fused kernels), and the property the round-1 advisor found missing: after
    load_policy_params_and_state! the optimiser is NEW (ppo.jl:88-91) even though the env is still bound to the handle whose Adam moments
    belong to the previous TrainState.  A callback copies the post-load rollout into the oracle (zero Adam state, same injected DataLoader
    order); the parameters after that update must agree — stale moments would be off by O(lr) in every weight."""
    capi = pkg._capi
    if kind == "cartpole_fused":
        base, layer_kw, alg = pkg.CartPoleEnv(max_steps=50), dict(hidden_dims=(64, 64)), pkg.PPO(n_steps=32, batch_size=128, epochs=2, learning_rate=1e-3)
    else:
        base, layer_kw, alg = pkg.PendulumEnv(max_steps=40), dict(hidden_dims=(128, 128)), pkg.PPO(n_steps=32, batch_size=128, epochs=2, learning_rate=1e-3, ent_coef=0.01)
    E = 16
    env = pkg.DeviceParallelEnv(base, E, seed=5)
    mk = lambda seed: pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), **layer_kw), alg, seed=seed)
    agent = mk(1)
    assert pkg.train_(agent, env, alg, 2 * alg.n_steps * E) is not None            # 2 iterations: Adam moments are now non-zero on the handle
    saved = pkg.save_policy_params_and_state(agent, tmp_path / "agent")
    fresh = mk(2)
    pkg.load_policy_params_and_state_(fresh, alg, saved)
    pa, pf = pkg.flatten_params(agent.train_state.parameters), pkg.flatten_params(fresh.train_state.parameters)
    np.testing.assert_array_equal(pa, pf)
    h = env.bind(alg, agent.layer)
    assert h is env.handle                                                         # still the SAME bound handle (that is the point)
    obs = np.random.default_rng(0).uniform(-1, 1, (64, h.D)).astype(np.float32)
    h.set_params(pa); a0 = h.predict_actions(obs, True)
    h.set_params(pf); a1 = h.predict_actions(obs, True)
    np.testing.assert_array_equal(a0, a1)                                          # identical deterministic actions (test_ppo_integration.jl:77-80)
    o = oracle_mod.Oracle(h.cfg); o.set_params(pf)                                 # the oracle: loaded weights, zero Adam moments

    class CopyRollout:
        def on_rollout_end(self, loc):
            hh = loc["env"].handle
            for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES):
                o.set_buffer(which, hh.buffer(which))
            perm = np.stack([np.random.default_rng(e).permutation(hh.N) for e in range(alg.epochs)]).astype(np.int64)
            hh.set_permutation(perm); o.set_permutation(perm)
            return True
    out = pkg.train_(fresh, env, alg, alg.n_steps * E, callbacks=[CopyRollout()])
    assert out is not None
    so = o.ppo_update()
    assert out[0]["losses"][0] == pytest.approx(so.loss, rel=1e-4)
    got, want = pkg.flatten_params(fresh.train_state.parameters), o.get_params()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=3e-6)
    assert np.abs(got - pf).max() > 1e-4                                           # the update did move the weights
    # and a second train_ on the SAME TrainState keeps its moments (the reference's train_state carries optimizer_state between train! calls)
    h.set_permutation(None)
    steps_before = fresh.gradient_updates
    assert pkg.train_(fresh, env, alg, alg.n_steps * E) is not None and fresh.gradient_updates > steps_before


def test_early_stop_callback_leaves_trained_weights_in_the_agent(pkg):
    """ppo.jl:145-152,170-176: train! mutates agent.train_state in place, so when a callback stops the run after k iterations the agent holds
    the k-iteration weights (round-1 advisor: the mirror returned None with the pre-training weights)"""
    env = pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=50), 16, seed=3)
    alg = pkg.PPO(n_steps=16, batch_size=64, epochs=1)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=0)
    p0 = pkg.flatten_params(agent.train_state.parameters).copy()

    class StopAtThird:
        def on_rollout_start(self, loc):
            return loc["i"] < 3
    assert pkg.train_(agent, env, alg, 10 * 16 * 16, callbacks=[StopAtThird()]) is None
    assert agent.steps_taken == 2 * 16 * 16 and agent.gradient_updates == 2 * 4
    p2 = pkg.flatten_params(agent.train_state.parameters)
    assert not np.array_equal(p0, p2)
    np.testing.assert_array_equal(p2, env.handle.get_params())


def test_unnormalize_round_trips(pkg):
    """test/test_normalize_wrapper.jl:141-195 (observations, tol 1e-5) and :197-249 (rewards, tol 1e-4) through the device wrapper's
    step-granular verbs: unnormalize_obs!(observe(env)) == get_original_obs(env), unnormalize_rewards!(rewards) == get_original_rewards(env)"""
    E = 2
    env = pkg.NormalizeWrapperEnv(pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=200), E, seed=11), training=True, norm_obs=True, norm_reward=False)
    rng = np.random.default_rng(0)
    env.reset_()
    for _ in range(6):
        env.act_(rng.uniform(-2, 2, (E, 1)).astype(np.float32))
    final_obs = np.stack(env.observe())
    original = pkg.get_original_obs(env)
    assert not np.allclose(final_obs, original)                                    # "The normalized observations should be different from original"
    un = pkg.unnormalize_obs_(final_obs.copy(), env)
    assert np.abs(un - original).max() < 1e-5
    # raw Pendulum observations are (cos, sin, theta_dot) of the simulator state: the "original" really is the env's own observation
    st, _ = env.handle.env_get_state()
    np.testing.assert_allclose(original, np.stack([np.cos(st[:, 0]), np.sin(st[:, 0]), st[:, 1]], axis=1), atol=1e-6)

    E = 64                                                                         # enough envs for a meaningful batch variance of the returns from the first step on
    env = pkg.NormalizeWrapperEnv(pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=200), E, seed=12), training=True, norm_obs=False, norm_reward=True)
    env.reset_()
    allr, origr = [], []
    for _ in range(8):
        r, *_ = env.act_(rng.uniform(-2, 2, (E, 1)).astype(np.float32))
        allr.extend(r); origr.extend(pkg.get_original_rewards(env))
    assert np.std(allr, ddof=1) < np.std(origr, ddof=1)                            # "should have lower variance"
    last, *_ = env.act_(rng.uniform(-2, 2, (E, 1)).astype(np.float32))
    last_original = pkg.get_original_rewards(env)
    un = pkg.unnormalize_rewards_(last.copy(), env)
    assert np.abs(un - last_original).max() < 1e-4
    assert (last_original <= 0).all() and (last_original > -17).all()              # Pendulum costs


def test_optimizer_state_follows_the_train_state_across_envs(pkg, tmp_path):
    """Lux.Training.TrainState carries optimizer_state (ppo.jl:52-53,239): the SAME agent trained on env A, then on env B, continues with its Adam moments — it does
    not restart from zero because the handle changed (round-2 advisor).  A TrainState without optimizer_state (a loaded checkpoint, ppo.jl:88-91) starts cold and
    ends somewhere else.  A checkpoint written for another architecture is refused (ppo.jl:90 replaces agent.layer with the stored one; here the shapes would
    silently mismatch)."""
    import copy
    alg = pkg.PPO(n_steps=16, batch_size=64, epochs=2, learning_rate=1e-3)
    E = 16
    mk_env = lambda: pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=50), E, seed=9)
    mk_agent = lambda: pkg.Agent(pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space()), alg, seed=4)
    per = alg.n_steps * E
    agent = mk_agent()
    assert agent.train_state.optimizer_state is None
    assert pkg.train_(agent, mk_env(), alg, per) is not None
    st1 = agent.train_state.optimizer_state
    assert st1["steps"] == 2 * 4 and np.abs(st1["m"]).max() > 0 and st1["beta_powers"] == pytest.approx((0.9 ** 9, 0.999 ** 9), rel=1e-5)
    cold = copy.deepcopy(agent); cold.train_state.optimizer_state = None            # same weights, fresh optimiser
    assert pkg.train_(agent, mk_env(), alg, per) is not None                        # a NEW env = a new handle: the moments arrive with the TrainState
    assert pkg.train_(cold, mk_env(), alg, per) is not None                         # identical env, identical rollout (same weights, same seed), zero moments
    st2 = agent.train_state.optimizer_state
    assert st2["steps"] == 16 and st2["beta_powers"] == pytest.approx((0.9 ** 17, 0.999 ** 17), rel=1e-5)
    assert cold.train_state.optimizer_state["steps"] == 8
    warm_p, cold_p = pkg.flatten_params(agent.train_state.parameters), pkg.flatten_params(cold.train_state.parameters)
    assert np.isfinite(warm_p).all() and np.abs(warm_p - cold_p).max() > 1e-4        # the carried moments changed the update

    small = pkg.Agent(pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space(), hidden_dims=(32, 32)), alg, seed=1)
    path = pkg.save_policy_params_and_state(small, tmp_path / "small")
    with pytest.raises(ValueError):
        pkg.load_policy_params_and_state_(mk_agent(), alg, path)
    loaded = pkg.load_policy_params_and_state_(mk_agent(), alg, pkg.save_policy_params_and_state(agent, tmp_path / "agent"))
    assert loaded.train_state.optimizer_state is None                               # ppo.jl:88-91: a fresh optimiser


def test_optimizer_state_round_trip_through_the_abi(pkg, oracle_mod):
    """dril_get_optimizer_state / dril_set_optimizer_state: moments, beta powers and step count move to a second handle, which then continues bit-identically"""
    capi = pkg._capi
    cfg = capi.default_config(0); cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.episode_len = 16, 24, 128, 3, 11
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(5).standard_normal(o.P) * 0.3).astype(np.float32); o.set_params(flat); o.env_reset(5); o.collect_rollout()
    bufs = (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES)
    perm = np.stack([np.random.default_rng(e).permutation(16 * 24) for e in range(3)]).astype(np.int64)
    def fresh():
        h = pkg.Handle(cfg); h.set_params(flat)
        for w in bufs: h.set_buffer(w, o.buffer(w))
        h.set_permutation(perm)
        return h
    a = fresh(); a.ppo_update(); st = a.get_optimizer_state(); p1 = a.get_params(); a.ppo_update(); want = a.get_params()
    assert st["steps"] == 9 and st["beta_powers"] == pytest.approx((0.9 ** 10, 0.999 ** 10), rel=1e-5)
    b = fresh(); b.set_params(p1); b.set_optimizer_state(st); b.ppo_update()
    np.testing.assert_array_equal(b.get_params(), want)
    c = fresh(); c.set_params(p1); c.ppo_update()                                   # zero moments from the same weights: a different result
    assert not np.array_equal(c.get_params(), want)
    with pytest.raises(ValueError):
        b.set_optimizer_state({"m": st["m"][:-1], "v": st["v"][:-1], "beta_powers": st["beta_powers"], "steps": 1})

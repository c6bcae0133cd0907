"""CPU: parts of the path the reference's tests leave UNPINNED — PPO loss/gradient, grad clip,
Adam, env physics — cross-checked against independent engines: torch-CPU autograd / torch.optim.Adam and a
float64 numpy restatement of the published Gymnasium equations."""
import math

import numpy as np
import pytest
import torch


def _hidden_of(cfg):
    """hidden_dims and activation carried by a dril_config (n_hidden == 0: the two-layer tanh form)"""
    hd = [cfg.hidden[i] for i in range(cfg.n_hidden)] if cfg.n_hidden else [cfg.hidden1, cfg.hidden2]
    F = torch.nn.functional                                   # dril_config.activation codes (NNlib's definitions: elu alpha = 1, leakyrelu 0.01)
    return hd, (torch.tanh, torch.relu, torch.sigmoid, F.elu, lambda z: F.leaky_relu(z, 0.01), F.softplus, lambda z: F.gelu(z, approximate="tanh"), F.silu)[cfg.activation]


def _nets(flat, D, hidden, A, discrete, dtype=torch.float64):
    """split the flat parameter vector (include/dril_hip.h layout: {W_1 b_1 ... W_{n+1} b_{n+1}} per net) into torch tensors (out x in, column-major)"""
    t = torch.tensor(flat, dtype=dtype, requires_grad=True)
    off = 0
    out = []
    for O in (A, 1):
        net = []
        dims = [D, *hidden, O]
        for i, o in zip(dims[:-1], dims[1:]):
            W = t[off:off + o * i].reshape(i, o).T; off += o * i
            b = t[off:off + o]; off += o
            net.append((W, b))
        out.append(net)
    ls = None if discrete else t[off:off + A]
    return t, out[0], out[1], ls


def _mlp(net, x, act=torch.tanh):
    h = x
    for W, b in net[:-1]:
        h = act(h @ W.T + b)
    return h @ net[-1][0].T + net[-1][1]


def torch_ppo_loss(flat, cfg, obs, actions, adv, ret, old_logp, old_val, discrete, A, dtype=torch.float64):
    """(alg::PPO)(policy, ps, st, batch): src/algorithms/ppo.jl:365-407 written with torch ops (float64; dtype=torch.float32 = the precision the reference
    itself computes in: Float32 tensors, Float32 BLAS sums)."""
    D = obs.shape[1]
    hidden, act = _hidden_of(cfg)
    t, actor, critic, ls = _nets(flat, D, hidden, A, discrete, dtype)
    x = torch.tensor(obs, dtype=dtype)
    advt = torch.tensor(adv, dtype=dtype)
    if cfg.normalize_advantage:
        advt = (advt - advt.mean()) / (advt.std(unbiased=True) + 1e-8)      # ppo.jl:350-356
    out = _mlp(actor, x, act)
    values = _mlp(critic, x, act)[:, 0]
    if discrete:
        p = torch.softmax(out, dim=1)
        a = torch.tensor(actions - cfg.action_start, dtype=torch.long)
        logp = torch.log(p.gather(1, a[:, None])[:, 0])
        ent = -(p * torch.log(p)).sum(1)
    else:
        xa = torch.tensor(actions, dtype=dtype)
        k = A
        logp = -0.5 * (2 * ls.sum() + ((xa - out) ** 2 * torch.exp(-2 * ls)).sum(1) + k * math.log(2 * math.pi))
        ent = (0.5 * k * (1 + math.log(2 * math.pi)) + ls.sum()).expand(x.shape[0])
    if cfg.has_clip_range_vf:
        ov = torch.tensor(old_val, dtype=dtype)
        values = ov + torch.clamp(values - ov, -cfg.clip_range_vf, cfg.clip_range_vf)
    r = torch.exp(logp - torch.tensor(old_logp, dtype=dtype))
    rc = torch.clamp(r, 1 - cfg.clip_range, 1 + cfg.clip_range)
    p_loss = -torch.minimum(r * advt, rc * advt).mean()
    ent_loss = -ent.mean()
    v_loss = ((values - torch.tensor(ret, dtype=dtype)) ** 2).mean()
    loss = p_loss + cfg.ent_coef * ent_loss + cfg.vf_coef * v_loss
    loss.backward()
    lr = logp - torch.tensor(old_logp, dtype=dtype)
    stats = [p_loss.item(), v_loss.item(), ent_loss.item(), (r != rc).to(dtype).mean().item(),
             (torch.exp(lr) - 1 - lr).mean().item(), ent.mean().item(), r.mean().item()]
    return loss.item(), np.array(stats), t.grad.numpy()


def make_batch(oracle, cfg, B, seed, discrete, A):
    """SURVEY.md §8d parity inputs: obs~U(-1,1), uniform actions, adv/ret/old_values~N(0,1), old_logp = eval + N(0, 0.1)"""
    rng = np.random.default_rng(seed)
    D = oracle.D
    obs = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    actions = (rng.integers(0, A, B) + cfg.action_start).astype(np.int32) if discrete else rng.normal(0, 1, (B, A)).astype(np.float32)
    adv, ret, old_val = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    _, lp, _ = oracle.evaluate_actions(obs, actions)
    old_logp = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
    return obs, actions, adv, ret, old_logp, old_val


@pytest.mark.parametrize("kind,B,variant", [(0, 64, "default"), (0, 1000, "ent_vfclip"), (1, 64, "default"), (1, 777, "ent_vfclip"), (0, 37, "no_norm")])
def test_ppo_loss_and_gradient_vs_torch_autograd(oracle_mod, pkg, kind, B, variant):
    cfg = pkg._capi.default_config(kind); cfg.n_envs, cfg.n_steps = 2, 2
    if variant == "ent_vfclip":
        cfg.ent_coef = 0.01; cfg.has_clip_range_vf = 1; cfg.clip_range_vf = 0.3; cfg.clip_range = 0.1
    if variant == "no_norm":
        cfg.normalize_advantage = 0
    o = oracle_mod.Oracle(cfg)
    rng = np.random.default_rng(100 + B)
    flat = (rng.standard_normal(o.P) * 0.25).astype(np.float32)
    o.set_params(flat)
    batch = make_batch(o, cfg, B, 7 + B, o.discrete, o.A)
    loss, stats, grads = o.ppo_loss_grad(*batch)
    tl, ts, tg = torch_ppo_loss(flat, cfg, *batch, o.discrete, o.A)
    assert loss == pytest.approx(tl, rel=1e-4)                     # BASELINE.json: PPO loss rel-err <= 1e-4
    np.testing.assert_allclose(stats, ts, rtol=2e-4, atol=2e-6)
    assert 0.0 < stats[3] < 0.95                                  # a real fraction of ratios is clipped
    np.testing.assert_allclose(grads, tg, rtol=2e-3, atol=2e-6)
    assert np.linalg.norm(grads - tg) <= 1e-4 * np.linalg.norm(tg)


@pytest.mark.parametrize("kind,hidden,act,B", [(0, (48,), 0, 100), (1, (40, 24, 56), 0, 257), (0, (32, 32, 16, 8), 1, 64), (1, (64, 64), 1, 129), (3, (96, 20, 33), 1, 77),
                                               (0, (40, 24), 2, 90), (1, (33, 65, 17), 3, 101), (3, (64,), 4, 55), (4, (48, 48), 5, 70),
                                               (0, (48, 32), 6, 80), (1, (40, 24, 56), 7, 97), (6, (64, 64), 6, 60)])   # sigmoid, elu, leakyrelu, softplus; gelu, swish (pre-activation kept)
def test_any_depth_and_relu_vs_torch_autograd(oracle_mod, pkg, kind, hidden, act, B):
    """ActorCriticLayer(...; hidden_dims, activation) beyond two tanh layers (layer_constructors.jl:6-10,55-56; get_mlp layer_helpers.jl:27-57): the oracle's
    any-depth MLP (1 to 4 hidden layers; tanh, relu, sigmoid, elu, leakyrelu, softplus, gelu, swish) — loss, statistics and every layer's gradient against torch autograd"""
    cfg = pkg._capi.default_config(kind); cfg.n_envs, cfg.n_steps = 2, 2; cfg.ent_coef = 0.01
    cfg.n_hidden = len(hidden); cfg.activation = act
    for i, h in enumerate(hidden):
        cfg.hidden[i] = h
    o = oracle_mod.Oracle(cfg)
    dims = lambda O: [o.D, *hidden, O]
    P = sum(i * j + j for O in (o.A, 1) for i, j in zip(dims(O)[:-1], dims(O)[1:])) + (0 if o.discrete else o.A)
    assert o.P == P
    flat = (np.random.default_rng(B).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat)
    batch = make_batch(o, cfg, B, 5 + B, o.discrete, o.A)
    loss, stats, grads = o.ppo_loss_grad(*batch)
    tl, ts, tg = torch_ppo_loss(flat, cfg, *batch, o.discrete, o.A)
    assert loss == pytest.approx(tl, rel=1e-4)
    np.testing.assert_allclose(stats, ts, rtol=2e-4, atol=2e-6)
    assert np.linalg.norm(grads - tg) <= 1e-4 * np.linalg.norm(tg)
    np.testing.assert_allclose(grads, tg, rtol=5e-3, atol=5e-6)


@pytest.mark.parametrize("bias", [-10.0, -15.0])
def test_softplus_far_negative_preactivations_vs_torch_autograd(oracle_mod, pkg, bias):
    """softplus'(x) = sigmoid(x) recovered from the stored activation y = softplus(x) as 1 - e^(-y) cancels for x << 0 (y ~ e^x: 1 % error at x = -10, all digits
    gone near -16).  With first-layer biases of -10 / -15 the first layer's gradient is tiny but must still be RELATIVELY right (-expm1(-y)); checked layer by layer
    against float64 torch autograd — the same case runs on the device in tests/test_gpu_external.py"""
    cfg = pkg._capi.default_config(4); cfg.n_envs, cfg.n_steps = 2, 2; cfg.ent_coef = 0.01      # MountainCarContinuous: D = 2, Box(1)
    cfg.n_hidden = 2; cfg.activation = 5; cfg.hidden[0] = cfg.hidden[1] = 48
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(3).standard_normal(o.P) * 0.3).astype(np.float32)
    D, H = o.D, 48
    per_net = [D * H + H + H * H + H + H * O + O for O in (o.A, 1)]
    l1 = []
    for base in (0, per_net[0]):
        flat[base + D * H: base + D * H + H] = bias                                    # b1 of this net
        l1.append(slice(base, base + D * H + H))
    o.set_params(flat)
    batch = make_batch(o, cfg, 90, 11, o.discrete, o.A)
    loss, stats, grads = o.ppo_loss_grad(*batch)
    tl, ts, tg = torch_ppo_loss(flat, cfg, *batch, o.discrete, o.A)
    assert loss == pytest.approx(tl, rel=1e-4)
    for sl in l1:
        assert 0 < np.linalg.norm(tg[sl]) < 1e-2 * np.linalg.norm(tg)                   # the far-negative units really are nearly closed ...
        assert np.linalg.norm(grads[sl] - tg[sl]) <= 1e-3 * np.linalg.norm(tg[sl])      # ... and their gradient is right to 1e-3 of ITS size (1 - e^(-y) gave 1e-2 at -10, O(1) at -15)
    assert np.linalg.norm(grads - tg) <= 1e-4 * np.linalg.norm(tg)


def _ext_cfg(pkg, D, A, discrete, H1, H2):
    c = pkg._capi.default_config(pkg._capi.ENV_EXTERNAL)
    c.ext_obs_dim, c.ext_action_dim, c.ext_discrete, c.hidden1, c.hidden2, c.n_envs, c.n_steps = D, A, int(discrete), H1, H2, 2, 2
    c.ext_action_low, c.ext_action_high = -1.0, 1.0
    return c


@pytest.mark.parametrize("D,A,discrete,H1,H2,B", [(6, 3, True, 64, 64, 100), (11, 5, False, 48, 80, 257), (33, 17, True, 20, 36, 64), (1, 1, False, 7, 5, 33)])
def test_external_spaces_loss_and_gradient_vs_torch_autograd(oracle_mod, pkg, D, A, discrete, H1, H2, B):
    """DRIL_ENV_EXTERNAL (host envs, any obs / action / hidden width): the oracle's loss and gradient for spaces the built-in env kinds do not
    have — wide Categorical, multi-dimensional DiagGaussian with its log_std gradient, unequal hidden widths — against torch autograd"""
    cfg = _ext_cfg(pkg, D, A, discrete, H1, H2); cfg.ent_coef = 0.01
    o = oracle_mod.Oracle(cfg)
    assert (o.D, o.A, o.discrete) == (D, A, discrete)
    flat = (np.random.default_rng(B).standard_normal(o.P) * 0.25).astype(np.float32)
    o.set_params(flat)
    batch = make_batch(o, cfg, B, 3 + B, discrete, A)
    loss, stats, grads = o.ppo_loss_grad(*batch)
    tl, ts, tg = torch_ppo_loss(flat, cfg, *batch, discrete, A)
    assert loss == pytest.approx(tl, rel=1e-4)
    np.testing.assert_allclose(stats, ts, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(grads, tg, rtol=2e-3, atol=2e-6)
    assert np.linalg.norm(grads - tg) <= 1e-4 * np.linalg.norm(tg)


def test_external_rollout_is_collect_trajectories(oracle_mod, pkg):
    """pins the oracle's step-by-step host-env rollout (orc_ext_act / _record / _finish) to its collect_rollout, which the reference's buffer and
    GAE tests pin (test/test_buffers.jl, test/test_gae.jl): an external context with CartPole's spaces, fed by a second context's CartPole
    simulator through the env verbs, fills the same buffer — including V(terminal_observation) bootstraps and the rollout-end values"""
    capi = pkg._capi
    E, T = 12, 60
    cf = capi.default_config(capi.ENV_CARTPOLE); cx = _ext_cfg(pkg, 4, 2, True, 64, 64)
    for c in (cf, cx):
        c.n_envs, c.n_steps, c.episode_len, c.batch_size = E, T, 16, E * T
    cx.action_start = cf.action_start
    full, sim, ext = oracle_mod.Oracle(cf), oracle_mod.Oracle(cf), oracle_mod.Oracle(cx)
    assert ext.P == full.P
    flat = (np.random.default_rng(0).standard_normal(full.P) * 0.4).astype(np.float32)
    for o in (full, sim, ext):
        o.set_params(flat)
    nz = np.random.default_rng(1).random(E * T)
    full.env_reset(5); sim.env_reset(5)
    full.set_noise(nz); full.collect_rollout()
    for t in range(T):
        raw, ea = ext.ext_act(sim.env_observe(), nz[t * E:(t + 1) * E])
        rew, term, trunc, tobs = sim.env_step(ea)
        ext.ext_record(rew, term, trunc, tobs)
    ext.ext_finish(sim.env_observe())
    fl = ext.buffer(capi.BUF_FLAGS)
    assert (fl & 1).any() and (fl & 2).any()                                          # terminations and truncations both occur
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_REWARDS, capi.BUF_VALUES, capi.BUF_LOGPROBS, capi.BUF_FLAGS, capi.BUF_BOOTSTRAP,
                  capi.BUF_ADVANTAGES, capi.BUF_RETURNS):
        assert np.array_equal(ext.buffer(which), full.buffer(which)), which
    lv_e, lv_f = ext.buffer(capi.BUF_LAST_VALUES), full.buffer(capi.BUF_LAST_VALUES)
    open_end = fl[(T - 1) * E:] == 0                                                  # LAST_VALUES is read only where the last step left the trajectory open
    assert np.array_equal(lv_e[open_end], lv_f[open_end])


def test_grad_clip_and_adam_vs_torch(oracle_mod, pkg):
    """nested_norm / nested_scale! (optimization_utils.jl:74-107) + Adam(eps=1e-5) (ppo.jl:64-66) vs torch.optim.Adam."""
    cfg = pkg._capi.default_config(0); cfg.n_envs, cfg.n_steps = 2, 2
    o = oracle_mod.Oracle(cfg)
    rng = np.random.default_rng(3)
    flat = (rng.standard_normal(o.P) * 0.2).astype(np.float32)
    o.set_params(flat)
    p = torch.tensor(flat.astype(np.float64), requires_grad=True)
    opt = torch.optim.Adam([p], lr=cfg.learning_rate, betas=(0.9, 0.999), eps=1e-5)
    for step in range(6):
        g = (rng.standard_normal(o.P) * (0.02 if step % 2 else 0.001)).astype(np.float32)   # alternately above / below max_grad_norm
        norm = o.apply_gradients(g)
        assert norm == pytest.approx(float(np.linalg.norm(g.astype(np.float64))), rel=1e-5)
        gt = torch.tensor(g.astype(np.float64))
        if norm > cfg.max_grad_norm:
            gt = gt * (cfg.max_grad_norm / norm)
        p.grad = gt
        opt.step()
        np.testing.assert_allclose(o.get_params(), p.detach().numpy(), rtol=1e-5, atol=2e-7)
    g = np.zeros(o.P, np.float32); g[5] = np.nan
    o.apply_gradients(g)
    assert o.last_rc == pkg._capi.ERR_NAN_IN_GRADS                  # @assert !nested_has_nan(grads), ppo.jl:213


def _cartpole_f64(s, a):
    """Gymnasium CartPole-v1 step (Euler), float64"""
    x, xd, th, thd = s
    force = 10.0 if a == 1 else -10.0
    c, sn = math.cos(th), math.sin(th)
    temp = (force + 0.05 * thd * thd * sn) / 1.1
    thacc = (9.8 * sn - c * temp) / (0.5 * (4.0 / 3.0 - 0.1 * c * c / 1.1))
    xacc = temp - 0.05 * thacc * c / 1.1
    return np.array([x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc])


def _pendulum_f64(s, u):
    th, thd = s
    u = min(max(u, -2.0), 2.0)
    an = ((th + math.pi) % (2 * math.pi)) - math.pi
    cost = an * an + 0.1 * thd * thd + 0.001 * u * u
    nthd = min(max(thd + (3 * 10.0 / 2 * math.sin(th) + 3.0 * u) * 0.05, -8.0), 8.0)
    return np.array([th + nthd * 0.05, nthd]), -cost


def test_env_physics_vs_gymnasium_equations(oracle_mod, pkg):
    """physics parity is unpinned by the reference: check the published equations."""
    capi = pkg._capi
    cfg = capi.default_config(0); cfg.n_envs, cfg.n_steps, cfg.episode_len = 16, 4, 500
    o = oracle_mod.Oracle(cfg); o.env_reset(5)
    st, _ = o.env_get_state()
    assert np.all(np.abs(st) <= 0.05)                               # reset ~ U(-0.05, 0.05)^4
    rng = np.random.default_rng(0)
    for _ in range(30):
        a = rng.integers(0, 2, cfg.n_envs).astype(np.int32)
        prev, _ = o.env_get_state()
        rew, term, trunc, _ = o.env_step(a + cfg.action_start)
        cur, sc = o.env_get_state()
        for e in range(cfg.n_envs):
            exp = _cartpole_f64(prev[e].astype(np.float64), a[e])
            t_exp = abs(exp[0]) > 2.4 or abs(exp[2]) > 12 * 2 * math.pi / 360
            assert bool(term[e]) == t_exp and rew[e] == 1.0
            if not term[e]:
                np.testing.assert_allclose(cur[e], exp, rtol=1e-5, atol=1e-6)
            else:
                assert np.all(np.abs(cur[e]) <= 0.05) and sc[e] == 0    # auto-reset, multithreadedParallelEnv.jl:68-70
    cfg = capi.default_config(1); cfg.n_envs, cfg.n_steps, cfg.episode_len = 8, 4, 5
    o = oracle_mod.Oracle(cfg); o.env_reset(9)
    st, _ = o.env_get_state()
    assert np.all(np.abs(st[:, 0]) <= math.pi + 1e-6) and np.all(np.abs(st[:, 1]) <= 1.0)
    for step in range(1, 8):
        u = rng.uniform(-3, 3, (cfg.n_envs, 1)).astype(np.float32)
        prev, _ = o.env_get_state()
        rew, term, trunc, tobs = o.env_step(u)
        cur, _ = o.env_get_state()
        assert not term.any() and trunc.all() == (step % 5 == 0)
        for e in range(cfg.n_envs):
            exp, r = _pendulum_f64(prev[e].astype(np.float64), float(u[e, 0]))
            assert rew[e] == pytest.approx(r, rel=1e-5, abs=1e-5)
            if trunc[e]:    # terminal_observation only on truncation (multithreadedParallelEnv.jl:64-66)
                np.testing.assert_allclose(tobs[e], [math.cos(exp[0]), math.sin(exp[0]), exp[1]], rtol=1e-5, atol=1e-5)
            else:
                np.testing.assert_allclose(cur[e], exp, rtol=1e-5, atol=1e-5)


def _mountaincar_f64(st, action, continuous):
    """Gymnasium MountainCar-v0 / MountainCarContinuous-v0 step in float64 -> (state, reward, terminated)"""
    pos, vel = float(st[0]), float(st[1])
    if continuous:
        force = min(max(float(action), -1.0), 1.0)
        vel += force * 0.0015 - 0.0025 * math.cos(3 * pos)
    else:
        force = 0.0
        vel += (int(action) - 1) * 0.001 + math.cos(3 * pos) * (-0.0025)
    vel = min(max(vel, -0.07), 0.07)
    pos += vel
    pos = min(max(pos, -1.2), 0.6)
    if pos == -1.2 and vel < 0:
        vel = 0.0
    goal = pos >= (0.45 if continuous else 0.5) and vel >= 0
    rew = ((100.0 if goal else 0.0) - 0.1 * force * force) if continuous else -1.0
    return np.array([pos, vel]), rew, goal


@pytest.mark.parametrize("kind", [3, 4])
def test_mountaincar_physics_vs_gymnasium_equations(oracle_mod, pkg, kind):
    """MountainCar-v0 / MountainCarContinuous-v0: reset distribution, step, wall rule, goal termination, time limit"""
    capi = pkg._capi
    cont = kind == capi.ENV_MOUNTAINCAR_CONTINUOUS
    cfg = capi.default_config(kind); cfg.n_envs, cfg.n_steps, cfg.episode_len = 32, 4, 40
    assert capi.default_config(kind).episode_len == (999 if cont else 200)
    o = oracle_mod.Oracle(cfg); o.env_reset(5)
    st, _ = o.env_get_state()
    assert np.all((st[:, 0] >= -0.6) & (st[:, 0] <= -0.4)) and not st[:, 1].any()
    np.testing.assert_array_equal(o.env_observe(), st)
    rng = np.random.default_rng(0)
    # put some cars next to the wall / the goal so that both special cases are hit
    st[:4] = [[-1.199, -0.05], [-1.2, -0.01], [0.49, 0.06], [0.44, 0.069]]
    o.env_set_state(st, np.zeros(32, np.int32))
    saw_goal = saw_wall = False
    for step in range(1, 45):
        a = rng.uniform(-1.5, 1.5, (32, 1)).astype(np.float32) if cont else (rng.integers(0, 3, 32) + cfg.action_start).astype(np.int32)
        if step <= 2:
            a[:2] = -1.0 if cont else cfg.action_start
            a[2:4] = 1.0 if cont else cfg.action_start + 2
        prev, psc = o.env_get_state()
        rew, term, trunc, tobs = o.env_step(a)
        cur, sc = o.env_get_state()
        for e in range(32):
            exp, r, goal = _mountaincar_f64(prev[e].astype(np.float64), a[e, 0] if cont else a[e] - cfg.action_start, cont)
            near = abs(exp[0] - (0.45 if cont else 0.5)) < 1e-6 or abs(exp[0] + 1.2) < 1e-7 and abs(prev[e][0] + 1.2) > 1e-7
            if not near:
                assert bool(term[e]) == goal
                assert rew[e] == pytest.approx(r, rel=1e-5, abs=1e-6)
            saw_goal |= bool(term[e]); saw_wall |= (exp[0] == -1.2 and exp[1] == 0.0)
            assert bool(trunc[e]) == (psc[e] + 1 >= 40)
            if term[e] or trunc[e]:
                assert -0.6 <= cur[e][0] <= -0.4 and cur[e][1] == 0 and sc[e] == 0         # auto-reset
                if trunc[e] and not near:
                    np.testing.assert_allclose(tobs[e], exp, rtol=1e-5, atol=1e-6)
            elif not near:
                np.testing.assert_allclose(cur[e], exp, rtol=1e-5, atol=1e-6)
    assert saw_goal and saw_wall


def _acrobot_f64(s, a):
    """Gymnasium acrobot.py in float64: _dsdt ("book"), rk4 over [0, 0.2], wrap, bound, termination, reward"""
    m1 = m2 = l1 = 1.0; lc1 = lc2 = 0.5; I1 = I2 = 1.0; g = 9.8; pi = math.pi

    def dsdt(y):
        t1, t2, w1, w2 = y
        d1 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * math.cos(t2)) + I1 + I2
        d2 = m2 * (lc2 ** 2 + l1 * lc2 * math.cos(t2)) + I2
        phi2 = m2 * lc2 * g * math.cos(t1 + t2 - pi / 2.0)
        phi1 = -m2 * l1 * lc2 * w2 ** 2 * math.sin(t2) - 2 * m2 * l1 * lc2 * w2 * w1 * math.sin(t2) + (m1 * lc1 + m2 * l1) * g * math.cos(t1 - pi / 2) + phi2
        dd2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * w1 ** 2 * math.sin(t2) - phi2) / (m2 * lc2 ** 2 + I2 - d2 ** 2 / d1)
        dd1 = -(d2 * dd2 + phi1) / d1
        return np.array([w1, w2, dd1, dd2])

    dt = 0.2
    y0 = np.asarray(s, np.float64)
    k1 = dsdt(y0); k2 = dsdt(y0 + dt / 2 * k1); k3 = dsdt(y0 + dt / 2 * k2); k4 = dsdt(y0 + dt * k3)
    ns = y0 + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    def wrap(x):
        while x > pi:
            x -= 2 * pi
        while x < -pi:
            x += 2 * pi
        return x

    ns[0], ns[1] = wrap(ns[0]), wrap(ns[1])
    ns[2], ns[3] = min(max(ns[2], -4 * pi), 4 * pi), min(max(ns[3], -9 * pi), 9 * pi)
    term = -math.cos(ns[0]) - math.cos(ns[1] + ns[0]) > 1.0
    return ns, (0.0 if term else -1.0), term


def test_acrobot_physics_vs_gymnasium_equations(oracle_mod, pkg):
    """Acrobot-v1: reset distribution, one RK4 step of the book dynamics per env step, angle wrap, velocity bounds, the height
    termination with reward 0, time limit + auto-reset"""
    capi = pkg._capi
    assert capi.default_config(capi.ENV_ACROBOT).episode_len == 500
    cfg = capi.default_config(capi.ENV_ACROBOT); cfg.n_envs, cfg.n_steps, cfg.episode_len = 32, 4, 60
    o = oracle_mod.Oracle(cfg); o.env_reset(5)
    assert (o.D, o.A, o.discrete) == (6, 3, True)
    st, _ = o.env_get_state()
    assert st.shape == (32, 4) and np.all(np.abs(st) <= 0.1) and np.abs(st).max() > 0.05
    obs = o.env_observe()
    np.testing.assert_allclose(obs, np.stack([np.cos(st[:, 0]), np.sin(st[:, 0]), np.cos(st[:, 1]), np.sin(st[:, 1]), st[:, 2], st[:, 3]], 1), atol=1e-6)
    rng = np.random.default_rng(0)
    st[:3] = [[3.0, 0.1, 12.4, 1.0], [2.9, 0.2, 0.5, 28.0], [-3.1, -3.0, -12.5, -28.2]]      # next to the wrap and the velocity bounds, above the bar
    o.env_set_state(st, np.zeros(32, np.int32))
    saw_term = saw_wrap = saw_bound = False
    for step in range(1, 70):
        a = (rng.integers(0, 3, 32) + cfg.action_start).astype(np.int32)
        if step > 8:
            a[:16] = np.where(o.env_get_state()[0][:16, 2] > 0, cfg.action_start + 2, cfg.action_start)   # pump energy into half of the arms so that some reach the height
        prev, psc = o.env_get_state()
        rew, term, trunc, tobs = o.env_step(a)
        cur, sc = o.env_get_state()
        for e in range(32):
            exp, r, tm = _acrobot_f64(prev[e], float(a[e] - cfg.action_start - 1))
            height = -math.cos(exp[0]) - math.cos(exp[1] + exp[0])
            edge = abs(height - 1.0) < 1e-4 or min(abs(abs(exp[0]) - math.pi), abs(abs(exp[1]) - math.pi)) < 1e-4    # f32 vs f64 may fall on either side
            if not edge:
                assert bool(term[e]) == tm and rew[e] == r
            saw_term |= bool(term[e]); saw_wrap |= abs(exp[0] - prev[e][0]) > 3.0 or abs(exp[1] - prev[e][1]) > 3.0
            saw_bound |= abs(abs(exp[2]) - 4 * math.pi) < 1e-9 or abs(abs(exp[3]) - 9 * math.pi) < 1e-9
            assert bool(trunc[e]) == (psc[e] + 1 >= 60)
            if term[e] or trunc[e]:
                assert np.all(np.abs(cur[e]) <= 0.1) and sc[e] == 0                                   # auto-reset
                if trunc[e] and not edge:
                    np.testing.assert_allclose(tobs[e], [math.cos(exp[0]), math.sin(exp[0]), math.cos(exp[1]), math.sin(exp[1]), exp[2], exp[3]], rtol=2e-4, atol=2e-4)
            elif not edge:
                np.testing.assert_allclose(cur[e], exp, rtol=2e-4, atol=2e-4)
    assert saw_term and saw_wrap and saw_bound


def f64_ppo_update(flat, cfg, bufs, perm, discrete, A, max_steps=None, dtype=np.float64):
    """The epoch x minibatch loop of train! (ppo.jl:205-239) carried in FLOAT64 end to end: torch-f64 autograd for the loss gradient, global-norm clip
    (optimization_utils.jl:74-107, no eps), Adam(eps = cfg.adam_eps) with the bias-corrected form of SURVEY a19.  bufs = (obs (N, D), actions, adv, ret, logp, val);
    perm (epochs, N).  Returns the float64 parameters after all (or max_steps) optimiser steps — the yardstick for how far fp32 rounding carries a long update
    (the configs[0] test measures the oracle's and the device's distance from it).  dtype=np.float32 carries the SAME loop in Float32 throughout (torch-f32
    autograd, f32 Adam state): the reference's own precision (PPO{Float32}, Float32 BLAS sums) — the yardstick for what "as close as the reference itself" means
    after 1 280 sequential Adam steps."""
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    obs, act, adv, ret, lp, val = bufs
    p = np.asarray(flat, dtype).copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    b1, b2, t = dtype(cfg.adam_beta1), dtype(cfg.adam_beta2), 0
    one, lr, eps = dtype(1), dtype(cfg.learning_rate), dtype(cfg.adam_eps)
    N, B = adv.shape[0], int(cfg.batch_size)
    for ep in range(perm.shape[0]):
        for k in range(0, N, B):
            idx = perm[ep, k:k + B]
            _, _, g = torch_ppo_loss(p, cfg, obs[idx], act[idx], adv[idx], ret[idx], lp[idx], val[idx], discrete, A, tdt)
            norm = np.sqrt((g * g).sum(dtype=dtype))
            if cfg.has_max_grad_norm and norm > cfg.max_grad_norm:
                g = g * (dtype(cfg.max_grad_norm) / norm)
            t += 1
            m = b1 * m + (one - b1) * g; v = b2 * v + (one - b2) * g * g
            p = p - lr * (m / (one - b1 ** dtype(t))) / (np.sqrt(v / (one - b2 ** dtype(t))) + eps)
            if max_steps is not None and t >= max_steps:
                return p
    return p


def test_f64_update_loop_matches_oracle_on_a_short_update(oracle_mod, pkg):
    """pins the float64 yardstick itself: 24 optimiser steps (3 epochs x 8 minibatches) of the oracle against f64_ppo_update at the short-update tolerance"""
    capi = pkg._capi
    cfg = capi.default_config(0); cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.episode_len = 8, 64, 64, 3, 30
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(2).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(3); o.collect_rollout()
    N = 8 * 64
    bufs = tuple(o.buffer(w).reshape(N, -1) if w == capi.BUF_OBSERVATIONS else o.buffer(w).reshape(N) for w in
                 (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_LOGPROBS, capi.BUF_VALUES))
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(3)]).astype(np.int64)
    o.set_permutation(perm); st = o.ppo_update()
    assert st.n_updates == 24
    p64 = f64_ppo_update(flat, cfg, bufs, perm, o.discrete, o.A)
    np.testing.assert_allclose(o.get_params(), p64, rtol=2e-4, atol=3e-6)
    assert np.abs(p64 - flat).max() > 1e-3


def test_update_loop_control_flow(oracle_mod, pkg):
    """ppo.jl:205-254: partial last batch kept, target_kl early stop skips the apply and both loops."""
    capi = pkg._capi
    cfg = capi.default_config(0); cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.episode_len = 5, 14, 16, 3, 6
    o = oracle_mod.Oracle(cfg)
    o.set_params((np.random.default_rng(0).standard_normal(o.P) * 0.3).astype(np.float32))
    o.env_reset(1); o.collect_rollout()
    st = o.ppo_update()
    assert st.n_updates == 3 * math.ceil(70 / 16) and not st.early_stopped     # ceil(N/B) minibatches per epoch
    assert np.isfinite([st.loss, st.grad_norm, st.explained_variance, st.approx_kl_div]).all()
    # a 1-sample trailing minibatch makes std(adv) NaN (ppo.jl:350-356, Julia std of one element) -> the NaN assert fires (:213)
    cfg1 = capi.default_config(0); cfg1.n_envs, cfg1.n_steps, cfg1.batch_size, cfg1.epochs, cfg1.episode_len = 5, 13, 16, 1, 6
    o1 = oracle_mod.Oracle(cfg1)
    o1.set_params((np.random.default_rng(0).standard_normal(o1.P) * 0.3).astype(np.float32))
    o1.env_reset(1); o1.collect_rollout()
    st1 = o1.ppo_update()
    assert o1.last_rc == capi.ERR_NAN_IN_GRADS and st1.nan_or_inf and st1.n_updates == 4
    cfg.has_target_kl = 1; cfg.target_kl = 1e-9
    o2 = oracle_mod.Oracle(cfg)
    p0 = (np.random.default_rng(0).standard_normal(o2.P) * 0.3).astype(np.float32)
    o2.set_params(p0); o2.env_reset(1); o2.collect_rollout()
    st2 = o2.ppo_update()
    assert st2.early_stopped and st2.n_updates <= 1                            # first batch has ratio == 1 -> kl == 0 -> one apply, then stop


def test_monitor_wrapper_oracle_vs_python(oracle_mod, pkg):
    """MonitorWrapperEnv restatement (monitorWrapperEnv.jl:46-70) against a direct Python transcription"""
    from collections import deque
    capi = pkg._capi
    cfg = capi.default_config(1); cfg.n_envs, cfg.n_steps, cfg.episode_len, cfg.monitor_window = 6, 4, 5, 7
    o = oracle_mod.Oracle(cfg); o.env_reset(1)
    rets, lens = deque(maxlen=7), deque(maxlen=7)
    cur_r, cur_l = np.zeros(6), np.zeros(6, int)
    rng = np.random.default_rng(0)
    for step in range(23):
        a = rng.uniform(-2, 2, (6, 1)).astype(np.float32)
        rew, term, trunc, _ = o.env_step(a)
        cur_r += rew; cur_l += 1
        for i in np.nonzero(term | trunc)[0]:
            rets.append(cur_r[i]); lens.append(cur_l[i]); cur_r[i] = 0; cur_l[i] = 0
        r, l, n = o.monitor_stats()
        assert n == len(rets)
        if n:
            assert r == pytest.approx(np.mean(rets), rel=1e-5) and l == pytest.approx(np.mean(lens), rel=1e-6)
    assert n == 7

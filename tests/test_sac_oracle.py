"""CPU: the SAC oracle (oracle/dril_sac_oracle.c) against the reference's known answers and an independent torch-autograd
restatement of src/algorithms/sac.jl.  No GPU, no product compute calls."""
import ctypes as C
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

import oracle_lib as O

ROOT = Path(__file__).resolve().parents[1]
GOLD = json.loads((ROOT / "tests" / "golden" / "sac_kats.json").read_text())


def make(pkg, E=8, hidden=(32, 32), B=16, cap=4096, act="relu", **alg_kw):
    env = pkg.PendulumEnv(max_steps=alg_kw.pop("max_steps", 200))
    alg = pkg.SAC(batch_size=B, buffer_capacity=cap, **alg_kw)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=hidden, activation=act)
    cfg = pkg.make_sac_config(env, E, alg, layer, seed=7)
    return O.sac_oracle(cfg), layer, alg


class _HostSpaces:
    """what make_sac_config needs of a HostParallelEnv: the DRIL_ENV_EXTERNAL kind and the two Box spaces"""

    def __init__(self, pkg, D, A, low=-1.0, high=1.0):
        self.kind, self._o, self._a = pkg._capi.ENV_EXTERNAL, pkg.Box(low=(-10.0,) * D, high=(10.0,) * D), pkg.Box(low=(low,) * A, high=(high,) * A)

    def observation_space(self):
        return self._o

    def action_space(self):
        return self._a


def make_ext(pkg, D, A, E=8, hidden=(32, 32), B=16, cap=4096, act="relu", low=-1.0, high=1.0, **alg_kw):
    env = _HostSpaces(pkg, D, A, low, high)
    alg = pkg.SAC(batch_size=B, buffer_capacity=cap, **alg_kw)
    layer = pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=hidden, activation=act)
    return O.sac_oracle(pkg.make_sac_config(env, E, alg, layer, seed=7)), layer, alg


def init_params(pkg, layer, seed=0, scale_out=30.0):
    """orthogonal init, with the tiny output layers scaled up so that means / Q values are O(1) and every term of the losses matters"""
    ps = layer.initialparameters(np.random.default_rng(seed))
    ps["actor_head"]["layer_3"]["weight"] *= scale_out
    rng = np.random.default_rng(seed + 1)
    for head in (ps["actor_head"], ps["critic_head"]["layer_1"], ps["critic_head"]["layer_2"]):
        for l in head.values():
            l["bias"] = rng.normal(0, 0.1, l["bias"].shape).astype(np.float32)
    ps["log_std"] = np.full_like(ps["log_std"], -1.0)
    return pkg.sac_flatten_params(ps)


# ---------------------------------------------------------------------------------------------------------------------
# known answers
# ---------------------------------------------------------------------------------------------------------------------
def test_polyak_update_known_answers():
    """test/test_utils.jl:4-25"""
    L = O.lib(); O.sac_oracle  # noqa: B018  (types the helper symbols below on first use)
    L.orc_polyak_update.restype = None
    L.orc_polyak_update.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
    for case in GOLD["polyak"]:
        t = np.array(case["target"], np.float32); s = np.array(case["source"], np.float32)
        L.orc_polyak_update(t.ctypes.data, s.ctypes.data, t.size, case["tau"])
        np.testing.assert_allclose(t, np.array(case["expected"], np.float32), rtol=1e-7, atol=0)


def test_squashed_logpdf_closed_form():
    """squashedDiagGaussian.jl:36-46 == log N(atanh x; mu, sigma) - sum log(1 - x^2) (the identity its comment cites), f64 reference"""
    L = O.lib()
    L.orc_squashed_logpdf.restype = C.c_float
    L.orc_squashed_logpdf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for case in GOLD["squashed_logpdf"]:
        x, mu, ls = (np.array(case[k], np.float32) for k in ("x", "mean", "log_std"))
        got = L.orc_squashed_logpdf(x.ctypes.data, mu.ctypes.data, ls.ctypes.data, x.size)
        assert got == pytest.approx(case["expected"], rel=2e-5, abs=2e-5)
    rng = np.random.default_rng(0)
    for k in (1, 2, 4):
        for _ in range(50):
            mu, ls = rng.normal(0, 1, k).astype(np.float32), rng.uniform(-3, 0.5, k).astype(np.float32)
            x = np.tanh(mu + np.exp(ls) * rng.normal(0, 1, k)).astype(np.float32)
            g = np.arctanh(np.clip(x.astype(np.float64), -1 + 1e-6, 1 - 1e-6))
            ref = -0.5 * (2 * ls.sum(dtype=np.float64) + (((g - mu) ** 2) * np.exp(-2.0 * ls)).sum() + k * math.log(2 * math.pi)) - np.log1p(-np.tanh(g) ** 2).sum()
            got = L.orc_squashed_logpdf(x.ctypes.data, mu.ctypes.data, ls.ctypes.data, k)
            assert got == pytest.approx(ref, rel=1e-4, abs=1e-4)


def test_train_schedule_arithmetic():
    """sac.jl:436-447 (and get_gradient_steps :59-65); expected values worked by hand from those lines"""
    L = O.lib()
    L.orc_sac_schedule.restype = None
    L.orc_sac_schedule.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_int64)] * 4
    for case in GOLD["schedule"]:
        o = [C.c_int64() for _ in range(4)]
        L.orc_sac_schedule(case["max_steps"], case["n_envs"], case["start_steps"], case["train_freq"], case["gradient_steps"], *[C.byref(v) for v in o])
        assert [v.value for v in o] == case["expected"], case


# ---------------------------------------------------------------------------------------------------------------------
# an independent restatement of update! in torch (f64): entropy step, critic step, actor step, zero-grad critic step, polyak
# ---------------------------------------------------------------------------------------------------------------------
class TorchSAC:
    def __init__(self, flat, D, A, H, act, alg, log_ent, target_entropy):
        self.D, self.A, self.H, self.act, self.alg = D, A, H, act, alg
        self.shapes = []
        for inp, out in ((D, A), (D + A, 1), (D + A, 1)):
            self.shapes += [(H[0], inp), (H[0],), (H[1], H[0]), (H[1],), (out, H[1]), (out,)]
        self.shapes.append((A,))
        self.p = []
        off = 0
        for sh in self.shapes:
            n = int(np.prod(sh))
            a = flat[off:off + n].astype(np.float64)
            self.p.append(torch.tensor(a.reshape(sh, order="F") if len(sh) == 2 else a, dtype=torch.float64, requires_grad=True))
            off += n
        self.target = [t.detach().clone() for t in self.p[6:18]]
        self.m = [torch.zeros_like(t) for t in self.p]; self.v = [torch.zeros_like(t) for t in self.p]
        self.t_actor = 0; self.t_critic = 0
        self.log_ent = torch.tensor(log_ent, dtype=torch.float64, requires_grad=True)
        self.ent_m = self.ent_v = 0.0; self.ent_t = 0
        self.target_entropy = target_entropy
        self.n = 0

    def mlp(self, ps, x):
        f = torch.relu if self.act == "relu" else torch.tanh
        h = f(x @ ps[0].T + ps[1]); h = f(h @ ps[2].T + ps[3])
        return h @ ps[4].T + ps[5]

    def alp(self, obs, noise, ps=None):
        ps = self.p if ps is None else ps
        mu = self.mlp(ps[0:6], obs); ls = ps[18]
        a = torch.tanh(mu + torch.exp(ls) * noise)
        g = torch.atanh(torch.clamp(a, -1 + 1e-6, 1 - 1e-6))
        glp = -0.5 * (2 * ls.sum() + (((g - mu) ** 2) * torch.exp(-2 * ls)).sum(1) + self.A * math.log(2 * math.pi))
        corr = (2 * (math.log(2) - g - torch.nn.functional.softplus(-2 * g))).sum(1)
        return a, glp - corr

    def q(self, ps12, obs, act):
        x = torch.cat([obs, act], 1)
        return torch.cat([self.mlp(ps12[0:6], x), self.mlp(ps12[6:12], x)], 1)

    def adam(self, idx, grads, t):
        lr, b1, b2, eps = self.alg.learning_rate, 0.9, 0.999, 1e-8
        with torch.no_grad():
            for i, g in zip(idx, grads):
                self.m[i] = b1 * self.m[i] + (1 - b1) * g; self.v[i] = b2 * self.v[i] + (1 - b2) * g * g
                self.p[i] -= lr * (self.m[i] / (1 - b1 ** t)) / (torch.sqrt(self.v[i] / (1 - b2 ** t)) + eps)

    def update(self, obs, act, rew, term, nobs, ne, nn, npi, auto_ent=True):
        T = lambda a: torch.tensor(np.asarray(a, np.float64))
        obs, act, rew, nobs, ne, nn, npi = map(T, (obs, act, rew, nobs, ne, nn, npi))
        term = torch.tensor(np.asarray(term, bool))
        alg, st = self.alg, {}
        if auto_ent:
            with torch.no_grad():
                c = (self.alp(obs, ne)[1] + self.target_entropy).mean()
            loss = -(self.log_ent * c)
            (g,) = torch.autograd.grad(loss, [self.log_ent])
            st["entropy_loss"] = loss.item()
            self.ent_t += 1
            self.ent_m = 0.9 * self.ent_m + 0.1 * g.item(); self.ent_v = 0.999 * self.ent_v + 0.001 * g.item() ** 2
            with torch.no_grad():
                self.log_ent -= alg.learning_rate * (self.ent_m / (1 - 0.9 ** self.ent_t)) / (math.sqrt(self.ent_v / (1 - 0.999 ** self.ent_t)) + 1e-8)
        alpha = torch.exp(self.log_ent.detach())
        with torch.no_grad():
            na, nlp = self.alp(nobs, nn)
            nq = self.q(self.target, nobs, na).min(1).values
            y = rew + torch.where(term, torch.zeros_like(rew), alg.gamma * (nq - alpha * nlp))
        cq = self.q(self.p[6:18], obs, act)
        closs = 0.5 * (((cq[:, 0] - y) ** 2).mean() + ((cq[:, 1] - y) ** 2).mean())
        gc = torch.autograd.grad(closs, self.p[6:18])
        self.t_critic += 1
        self.adam(range(6, 18), gc, self.t_critic)
        a_pi, lp = self.alp(obs, npi)
        aloss = (alpha * lp - self.q(self.p[6:18], obs, a_pi).min(1).values).mean()
        ga = torch.autograd.grad(aloss, self.p[0:6] + [self.p[18]])
        self.t_actor += 1
        self.adam(list(range(0, 6)) + [18], ga, self.t_actor)
        self.t_critic += 1
        self.adam(range(6, 18), [torch.zeros_like(t) for t in self.p[6:18]], self.t_critic)      # zero_critic_grads! then apply_gradients
        if self.n % alg.target_update_interval == 0:
            with torch.no_grad():
                for t, s in zip(self.target, self.p[6:18]):
                    t.mul_(1 - alg.tau).add_(alg.tau * s)
        self.n += 1
        st.update(actor_loss=aloss.item(), critic_loss=closs.item(), mean_q_values=cq.mean().item(), entropy_coefficient=math.exp(self.log_ent.item()),
                  grad_norm=math.sqrt(sum((g ** 2).sum().item() for g in gc) + sum((g ** 2).sum().item() for g in ga)))
        return st, gc, ga

    def flat(self, tensors):
        return np.concatenate([(t.detach().numpy().ravel(order="F") if t.ndim == 2 else t.detach().numpy().ravel()) for t in tensors])


def random_replay(rng, n, D, A, p_term=0.2):
    obs, nobs = rng.uniform(-1, 1, (n, D)).astype(np.float32), rng.uniform(-1, 1, (n, D)).astype(np.float32)
    act = np.tanh(rng.normal(0, 1, (n, A))).astype(np.float32)
    rew = rng.normal(-1, 1, n).astype(np.float32)
    term = (rng.uniform(size=n) < p_term).astype(np.uint8)
    return obs, act, rew, term, np.zeros(n, np.uint8), nobs


@pytest.mark.parametrize("act,auto_ent,interval", [("relu", True, 1), ("tanh", True, 2), ("relu", False, 1)])
def test_update_matches_torch_autograd(pkg, act, auto_ent, interval):
    """three consecutive update! steps: losses, gradients, parameters, targets and log_ent_coef against torch f64 autograd + textbook Adam,
    including the reference's zero-gradient Adam step on the critics (sac.jl:381-382)"""
    B, n_upd = 16, 3
    ent = pkg.AutoEntropyCoefficient(initial_value=0.7) if auto_ent else pkg.FixedEntropyCoefficient(0.3)
    h, layer, alg = make(pkg, hidden=(32, 32), B=B, act=act, ent_coef=ent, target_update_interval=interval, learning_rate=3e-3, tau=0.05)
    flat = init_params(pkg, layer)
    h.set_params(flat)
    rng = np.random.default_rng(3)
    rb = random_replay(rng, 64, h.D, h.A)
    h.replay_fill(*rb)
    idx = rng.integers(0, 64, (n_upd, B))
    ne, nn, npi = (rng.normal(0, 1, (n_upd, B, h.A)).astype(np.float32) for _ in range(3))
    T = TorchSAC(flat, h.D, h.A, (32, 32), act, alg, math.log(0.7 if auto_ent else 0.3), -float(h.A))
    for k in range(n_upd):
        h.set_batches(1, idx[k:k + 1], ne[k:k + 1], nn[k:k + 1], npi[k:k + 1])
        (st,) = h.update(1)
        j = idx[k]
        ref, gc, ga = T.update(rb[0][j], rb[1][j], rb[2][j], rb[3][j], rb[5][j], ne[k], nn[k], npi[k], auto_ent)
        assert st.critic_loss == pytest.approx(ref["critic_loss"], rel=2e-5)
        assert st.actor_loss == pytest.approx(ref["actor_loss"], rel=2e-5, abs=2e-6)
        assert st.mean_q_values == pytest.approx(ref["mean_q_values"], rel=2e-5, abs=2e-6)
        assert st.grad_norm == pytest.approx(ref["grad_norm"], rel=2e-5)
        assert st.entropy_coefficient == pytest.approx(ref["entropy_coefficient"], rel=1e-6)
        assert bool(st.has_entropy_loss) == auto_ent
        if auto_ent:
            assert st.entropy_loss == pytest.approx(ref["entropy_loss"], rel=2e-5, abs=2e-6)
        ogc, oga = h.last_grads()
        n_actor = T.flat(T.p[0:6]).size
        rgc, rga = T.flat(gc), T.flat(ga)
        np.testing.assert_allclose(ogc[n_actor:n_actor + rgc.size], rgc, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(np.concatenate([oga[:n_actor], oga[-h.A:]]), rga, rtol=2e-4, atol=1e-7)
        # the sub-trees test/test_sac.jl:282-284,337-341 require to be zero
        assert not ogc[:n_actor].any() and not ogc[-h.A:].any() and not oga[n_actor:-h.A].any()
        np.testing.assert_allclose(h.get_params(), T.flat(T.p), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(h.get_target_params(), T.flat(T.target), rtol=2e-5, atol=1e-6)
        assert h.get_log_ent_coef() == pytest.approx(T.log_ent.item(), rel=1e-5, abs=1e-6)


def test_external_spaces_update_matches_torch_autograd(pkg):
    """DRIL_ENV_EXTERNAL (host envs): an 11-dim observation, a 5-dim action and unequal hidden widths — shapes the built-in Pendulum kind does not
    have (multi-dimensional squashed Gaussian, target entropy -5) — against the same torch f64 autograd restatement"""
    B, n_upd, D, A, H = 16, 2, 11, 5, (24, 40)
    h, layer, alg = make_ext(pkg, D, A, hidden=H, B=B, act="tanh", ent_coef=pkg.AutoEntropyCoefficient(initial_value=0.5), learning_rate=3e-3, tau=0.05)
    assert (h.D, h.A) == (D, A) and h.P == layer.parameterlength()
    flat = init_params(pkg, layer, scale_out=10.0)
    h.set_params(flat)
    rng = np.random.default_rng(5)
    rb = random_replay(rng, 64, D, A)
    h.replay_fill(*rb)
    idx = rng.integers(0, 64, (n_upd, B))
    ne, nn, npi = (rng.normal(0, 1, (n_upd, B, A)).astype(np.float32) for _ in range(3))
    T = TorchSAC(flat, D, A, H, "tanh", alg, math.log(0.5), -float(A))
    for k in range(n_upd):
        h.set_batches(1, idx[k:k + 1], ne[k:k + 1], nn[k:k + 1], npi[k:k + 1])
        (st,) = h.update(1)
        j = idx[k]
        ref, gc, ga = T.update(rb[0][j], rb[1][j], rb[2][j], rb[3][j], rb[5][j], ne[k], nn[k], npi[k], True)
        assert st.critic_loss == pytest.approx(ref["critic_loss"], rel=5e-5) and st.actor_loss == pytest.approx(ref["actor_loss"], rel=5e-5, abs=5e-6)
        assert st.entropy_loss == pytest.approx(ref["entropy_loss"], rel=5e-5, abs=5e-6) and st.grad_norm == pytest.approx(ref["grad_norm"], rel=5e-5)
        np.testing.assert_allclose(h.get_params(), T.flat(T.p), rtol=3e-4, atol=3e-6)
        np.testing.assert_allclose(h.get_target_params(), T.flat(T.target), rtol=3e-5, atol=1e-6)


def test_external_push_is_collect_rollout(pkg):
    """pins orc_sac_ext_push (one host-env step into the ring) to orc_sac_collect_rollout: an external context with Pendulum's spaces, fed by the
    on-policy oracle's Pendulum simulator through the env verbs, builds the same replay buffer — random start phase (env-space actions stored),
    policy phase (raw squashed actions stored), truncated transitions with their terminal observation, ring overwrite"""
    capi = pkg._capi
    E, L, n_rand, n_pol = 6, 5, 4, 13
    full, layer, alg = make(pkg, E=E, hidden=(16, 16), cap=80, max_steps=L)
    ext, layer_x, _ = make_ext(pkg, 3, 1, E=E, hidden=(16, 16), cap=80, low=-2.0, high=2.0)
    assert ext.P == full.P
    flat = init_params(pkg, layer, scale_out=20.0)
    full.set_params(flat); ext.set_params(flat)
    cs = capi.default_config(capi.ENV_PENDULUM); cs.n_envs, cs.n_steps, cs.episode_len = E, 2, L
    sim = O.Oracle(cs)
    full.env_reset(7); sim.env_reset(7)
    rng = np.random.default_rng(2)
    u = rng.random((n_rand, E, 1)).astype(np.float32); z = rng.standard_normal((n_pol, E, 1)).astype(np.float32)
    full.set_collect_noise(u); full.collect_rollout(n_rand, True)
    full.set_collect_noise(z); full.collect_rollout(n_pol, False)
    obs = sim.env_observe()
    for t in range(n_rand + n_pol):
        if t < n_rand:
            stored = env_a = (-2.0 + u[t] * 4.0).astype(np.float32)                       # rand(rng, act_space) = low + u (high - low)
        else:
            stored, env_a = ext.predict_actions(obs, False, z[t - n_rand])
        rew, term, trunc, tobs = sim.env_step(env_a)
        nobs = sim.env_observe()
        ext.ext_push(obs, stored, rew, term, trunc, nobs, tobs if trunc.any() else None)
        obs = nobs
    assert ext.replay_size() == full.replay_size() == 80 < (n_rand + n_pol) * E          # the ring wrapped
    assert ext.replay(capi.RB_TRUNCATED).any()
    for which in (capi.RB_OBSERVATIONS, capi.RB_ACTIONS, capi.RB_REWARDS, capi.RB_TERMINATED, capi.RB_TRUNCATED, capi.RB_NEXT_OBSERVATIONS):
        assert np.array_equal(ext.replay(which), full.replay(which)), which


def test_injected_batches_equal_one_call_or_many(pkg):
    h1, layer, _ = make(pkg, B=8)
    h2, _, _ = make(pkg, B=8)
    flat = init_params(pkg, layer)
    rng = np.random.default_rng(5)
    rb = random_replay(rng, 40, 3, 1)
    idx = rng.integers(0, 40, (4, 8)); nz = [rng.normal(0, 1, (4, 8, 1)).astype(np.float32) for _ in range(3)]
    for h in (h1, h2):
        h.set_params(flat); h.replay_fill(*rb)
    h1.set_batches(4, idx, *nz); s1 = h1.update(4)
    s2 = []
    for k in range(4):
        h2.set_batches(1, idx[k:k + 1], *[z[k:k + 1] for z in nz]); s2 += h2.update(1)
    np.testing.assert_array_equal(h1.get_params(), h2.get_params())
    assert [s.critic_loss for s in s1] == [s.critic_loss for s in s2]
    # un-injected batches come from the Philox streams: deterministic, and different from step to step
    h1.replay_fill(*rb); h2.replay_fill(*rb)
    a, b = h1.update(2), h2.update(2)
    assert [s.actor_loss for s in a] == [s.actor_loss for s in b] and a[0].actor_loss != a[1].actor_loss


# ---------------------------------------------------------------------------------------------------------------------
# collection + replay semantics
# ---------------------------------------------------------------------------------------------------------------------
def test_collect_semantics(pkg):
    """off_policy_collection.jl:28-96: random actions are stored in env space, policy actions raw (pre-adapter); the next observation of a
    transition is the following observation of the same env, except at a time-limit truncation where it is the terminal observation"""
    E, L = 4, 5
    h, layer, _ = make(pkg, E=E, max_steps=L)
    h.set_params(init_params(pkg, layer))
    h.env_reset(11)
    obs0 = h.env_observe()
    u = np.random.default_rng(0).uniform(0, 1, (3, E, 1)).astype(np.float32)
    h.set_collect_noise(u)
    h.collect_rollout(3, use_random_actions=True)
    assert h.replay_size() == 3 * E
    acts = h.replay(pkg._capi.RB_ACTIONS).reshape(3, E)
    np.testing.assert_allclose(acts, -2 + 4 * u[..., 0], rtol=1e-6)                     # rand(act_space), stored unprocessed (:50-53,72)
    obs = h.replay(pkg._capi.RB_OBSERVATIONS).reshape(3, E, 3); nxt = h.replay(pkg._capi.RB_NEXT_OBSERVATIONS).reshape(3, E, 3)
    np.testing.assert_array_equal(obs[0], obs0)
    np.testing.assert_array_equal(nxt[:2], obs[1:])                                     # mid-trajectory: buffer.observations[i + 1]
    assert not h.replay(pkg._capi.RB_TERMINATED).any() and not h.replay(pkg._capi.RB_TRUNCATED).any()
    nz = np.random.default_rng(1).normal(0, 1, (4, E, 1)).astype(np.float32)
    before = h.env_observe()
    h.set_collect_noise(nz)
    h.collect_rollout(4)                                                                 # steps 4..7 of the episodes: truncation at step 5
    raw = h.replay(pkg._capi.RB_ACTIONS).reshape(7, E)[3:]
    exp_raw, exp_env = h.predict_actions(before, noise=nz[0])
    np.testing.assert_allclose(raw[0], exp_raw[:, 0], rtol=1e-6)
    np.testing.assert_allclose(exp_env, 2 * np.tanh(exp_raw), rtol=1e-6)                 # to_env(TanhScaleAdapter): tanh applied again, default_adapters.jl:13-21
    assert np.abs(raw).max() <= 1.0
    tr = h.replay(pkg._capi.RB_TRUNCATED).reshape(7, E)
    assert tr[4].all() and tr.sum() == E                                                 # the 5th step of every env
    obs = h.replay(pkg._capi.RB_OBSERVATIONS).reshape(7, E, 3); nxt = h.replay(pkg._capi.RB_NEXT_OBSERVATIONS).reshape(7, E, 3)
    assert not np.array_equal(nxt[4], obs[5])                                            # terminal observation, not the reset observation
    np.testing.assert_array_equal(nxt[5], obs[6])
    # the terminal observation continues the physics from obs[4] (|theta_dot| changes by at most max torque * dt * 3 + gravity term)
    assert np.all(np.abs(nxt[4][:, 2] - obs[4][:, 2]) < 1.2)
    np.testing.assert_allclose(np.hypot(nxt[4][:, 0], nxt[4][:, 1]), 1.0, rtol=1e-5)


def test_replay_ring_overwrites_oldest(pkg):
    """CircularBuffer semantics (replay_buffer.jl:14-31): logical index 0 is always the oldest surviving element"""
    E = 4
    h, layer, _ = make(pkg, E=E, cap=10)
    h.set_params(init_params(pkg, layer))
    h.env_reset(3)
    h.collect_rollout(2, use_random_actions=True)
    first = h.replay(pkg._capi.RB_REWARDS)
    assert h.replay_size() == 8 and h.replay_capacity() == 10
    h.collect_rollout(1, use_random_actions=True)
    r = h.replay(pkg._capi.RB_REWARDS)
    assert h.replay_size() == 10
    np.testing.assert_array_equal(r[:6], first[2:])                                      # two oldest dropped
    h.collect_rollout(3, use_random_actions=True)
    r2 = h.replay(pkg._capi.RB_REWARDS)
    assert h.replay_size() == 10 and not np.array_equal(r2, r)


def test_train_loop_counts_and_learning_signal(pkg):
    """train! sac.jl:414-549: first a random-action collection of start_steps / E steps, then (collect train_freq, gradient_steps updates)"""
    E = 8
    h, layer, alg = make(pkg, E=E, B=32, cap=2048, start_steps=64, train_freq=2, gradient_steps=3)
    h.set_params(init_params(pkg, layer, scale_out=1.0))
    h.env_reset(0)
    stats, fps, n_upd, iters, total = h.train(64 + 5 * 2 * E)
    assert iters == 6 and n_upd == 18 and total == 64 + 5 * 16 and h.replay_size() == total
    assert all(np.isfinite([s.actor_loss, s.critic_loss, s.entropy_loss, s.grad_norm]).all() for s in stats)
    assert stats[-1].entropy_coefficient < stats[0].entropy_coefficient < 1.0            # entropy far above the target: log_ent_coef descends
    h2, _, _ = make(pkg, E=E, B=32, cap=2048, start_steps=64, train_freq=2, gradient_steps=-1)
    h2.set_params(init_params(pkg, layer)); h2.env_reset(0)
    assert h2.train(64)[2] == 2 * E                                                     # gradient_steps = -1 -> train_freq * n_envs (:59-65)

"""Generates tests/golden/*.json — the reference's own analytic tests re-expressed as data.

The reference (Julia) cannot run in the build image, so these vectors come from the CLOSED FORMS its tests
assert against (each case cites the test it restates), evaluated here in float64 with formulas that do not
share code with the oracle or the HIP library:
  * GAE: A_t = sum_l (gamma*lambda)^l * delta_{t+l} as an explicit double sum (the reference's tests use the
    same definition: test/test_gae.jl:38-66, test/test_shared_setup.jl:295-318)
  * distributions: scipy.stats (the reference compares against Distributions.jl: test/test_distributions.jl)
  * RunningMeanStd: numpy batch moments of the concatenated batches (test/test_normalize_wrapper.jl:3-70)

Run:  python tests/golden/make_golden.py     (deterministic; commit the JSON it writes)
"""
import json
from pathlib import Path

import numpy as np
from scipy import stats

OUT = Path(__file__).resolve().parent


def gae_direct(rewards, values, gamma, lam, terminated, bootstrap):
    """one trajectory, explicit sum over future TD errors (float64)"""
    n = len(rewards)
    nxt = list(values[1:]) + [0.0 if (terminated or bootstrap is None) else bootstrap]
    delta = [rewards[i] + gamma * nxt[i] - values[i] for i in range(n)]
    return [sum((gamma * lam) ** l * delta[t + l] for l in range(n - t)) for t in range(n)]


def gae_case(name, cite, T, gamma, lam, rewards, values, flags, bootstrap, last_value, atol=1e-4):
    """single env (E=1) time-major arrays; trajectories are cut at flags!=0 and at T-1"""
    exp = [0.0] * T
    start = 0
    for t in range(T):
        term, trunc = flags[t] & 1, (flags[t] >> 1) & 1
        if term or trunc or t == T - 1:
            if term:
                b = None
            elif trunc:
                b = bootstrap[t]
            else:
                b = last_value
            seg = gae_direct(rewards[start:t + 1], values[start:t + 1], gamma, lam, bool(term), b)
            exp[start:t + 1] = seg
            start = t + 1
    return dict(name=name, cite=cite, n_envs=1, n_steps=T, gamma=gamma, gae_lambda=lam, rewards=rewards, values=values,
                flags=flags, bootstrap=bootstrap, last_values=[last_value], expected_advantages=exp,
                expected_returns=[a + v for a, v in zip(exp, values)], atol=atol)


def make_gae():
    cases = []
    # test/test_gae.jl:1-71 — 8 steps, reward only at the end, V=0.5, terminated
    T = 8
    c = gae_case("analytic_8", "test/test_gae.jl:1-71", T, 0.99, 0.95, [0.0] * 7 + [1.0], [0.5] * T, [0] * 7 + [1], [0.0] * T, 0.0)
    # the test's own closed form (test_gae.jl:57-61) must agree with the direct sum
    gl, d = 0.99 * 0.95, -0.005
    closed = [d * ((1 - gl ** (7 - i)) / (1 - gl)) + gl ** (7 - i) * 0.5 for i in range(7)] + [0.5]
    assert np.allclose(closed, c["expected_advantages"], atol=1e-12)
    cases.append(c)
    # test/test_gae.jl:73-115 — five (gamma, lambda) pairs, 4 steps, V=0.3
    for g, l in [(0.95, 0.9), (0.99, 0.95), (1.0, 1.0), (0.9, 0.0), (0.8, 0.5)]:
        cases.append(gae_case(f"param_g{g}_l{l}", "test/test_gae.jl:73-115", 4, g, l, [0.0, 0.0, 0.0, 1.0], [0.3] * 4, [0, 0, 0, 1], [0.0] * 4, 0.0))
    # test/test_gae.jl:117-174 — 64-step episode inside a 64-step rollout
    cases.append(gae_case("customenv_64", "test/test_gae.jl:117-174", 64, 0.99, 0.95, [0.0] * 63 + [1.0], [0.5] * 64, [0] * 63 + [1], [0.0] * 64, 0.0))
    # test/test_gae.jl:176-220 — four 8-step episodes in one 32-step rollout, gamma=lambda=1, V=0 => everything is 1
    r = ([0.0] * 7 + [1.0]) * 4
    f = ([0] * 7 + [1]) * 4
    c = gae_case("multi_episode_32", "test/test_gae.jl:176-220", 32, 1.0, 1.0, r, [0.0] * 32, f, [0.0] * 32, 0.0)
    assert np.allclose(c["expected_returns"], 1.0) and np.allclose(c["expected_advantages"], 1.0)
    cases.append(c)
    # test/test_gae.jl:222-270 — never terminates: rollout-limited bootstrap with V(next obs) = 0.5
    cases.append(gae_case("infinite_horizon_8", "test/test_gae.jl:222-270", 8, 0.9, 0.8, [1.0] * 8, [0.5] * 8, [0] * 8, [0.0] * 8, 0.5))
    # test/test_gae.jl:272-322 — edge cases: 1-step episode, gamma = 0, lambda = 0
    cases.append(gae_case("edge_single_step", "test/test_gae.jl:272-303", 1, 0.9, 0.8, [1.0], [0.3], [1], [0.0], 0.0))
    cases.append(gae_case("edge_gamma0", "test/test_gae.jl:304-310", 1, 0.0, 0.8, [1.0], [0.3], [1], [0.0], 0.0))
    cases.append(gae_case("edge_lambda0_td0", "test/test_gae.jl:311-321", 3, 0.9, 0.0, [0.0, 0.0, 1.0], [0.3] * 3, [0, 0, 1], [0.0] * 3, 0.0))
    # test/test_buffers.jl:60-115 — terminated (no bootstrap) vs truncated with bootstrap 0.2, V=0.7
    cases.append(gae_case("buffers_terminated", "test/test_buffers.jl:60-96", 6, 0.9, 0.8, [0.0] * 5 + [1.0], [0.7] * 6, [0] * 5 + [1], [0.0] * 6, 0.0))
    cases.append(gae_case("buffers_truncated_boot0.2", "test/test_buffers.jl:98-115", 6, 0.9, 0.8, [0.0] * 5 + [1.0], [0.7] * 6, [0] * 5 + [2], [0.0] * 5 + [0.2], 0.0))
    # terminated wins over truncated when both flags are set (src/buffers/trajectory.jl:85)
    cases.append(gae_case("terminated_and_truncated", "src/buffers/trajectory.jl:85", 6, 0.9, 0.8, [0.0] * 5 + [1.0], [0.7] * 6, [0] * 5 + [3], [0.0] * 5 + [0.2], 0.0))
    # a truncation in the middle of a rollout followed by a rollout-limited tail
    cases.append(gae_case("mid_truncation_then_tail", "src/buffers/trajectory.jl:52-74", 7, 0.97, 0.9, [1.0, 0.5, -1.0, 2.0, 0.0, 1.0, 1.0],
                          [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7], [0, 0, 2, 0, 0, 0, 0], [0.0, 0.0, 0.9, 0.0, 0.0, 0.0, 0.0], -0.4))
    (OUT / "gae.json").write_text(json.dumps(cases, indent=1))


def make_distributions():
    rng = np.random.default_rng(20240607)
    gauss = []
    for shape in [(1,), (1, 1), (2,), (2, 3), (2, 3, 1), (2, 3, 4)]:  # test/test_distributions.jl:8
        k = int(np.prod(shape))
        for _ in range(8):
            mean = rng.uniform(-1, 2, k); log_std = rng.uniform(-1, 2, k); x = rng.uniform(-1, 2, k)
            mvn = stats.multivariate_normal(mean, np.diag(np.exp(log_std) ** 2))
            gauss.append(dict(k=k, mean=mean.astype(np.float32).tolist(), log_std=log_std.astype(np.float32).tolist(),
                              x=x.astype(np.float32).tolist(),
                              logpdf=float(stats.multivariate_normal(np.float32(mean).astype(np.float64), np.diag(np.exp(np.float32(log_std).astype(np.float64)) ** 2)).logpdf(np.float32(x).astype(np.float64))),
                              entropy=float(stats.multivariate_normal(np.float32(mean).astype(np.float64), np.diag(np.exp(np.float32(log_std).astype(np.float64)) ** 2)).entropy())))
            del mvn
    cat = []
    for n in (3, 8):  # test/test_distributions.jl:99
        for _ in range(16):
            p = rng.random(n).astype(np.float32); p = (p / p.sum()).astype(np.float32)
            p64 = p.astype(np.float64)
            cat.append(dict(p=p.tolist(), logpdf_first=float(np.log(p64[0])), entropy=float(-(p64 * np.log(p64)).sum()),
                            # Categorical.rand = findfirst(cumsum(p) .>= u) (categorical.jl:47-52)
                            samples=[dict(u=float(u), index=int(np.searchsorted(np.cumsum(p, dtype=np.float32).astype(np.float64), u, side="left")))
                                     for u in rng.random(6)]))
    (OUT / "distributions.json").write_text(json.dumps(dict(cite="test/test_distributions.jl:1-39,94-119", diag_gaussian=gauss, categorical=cat), indent=1))


def make_rms():
    cases = []
    b1 = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.float32)  # (dims=3) x (batch=3), test_normalize_wrapper.jl:17
    b2 = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.float32)  # :27
    both = np.hstack([b1, b2])
    cases.append(dict(name="two_batches", cite="test/test_normalize_wrapper.jl:3-38", dims=3, batches=[b1.T.tolist(), b2.T.tolist()],
                      mean_after=[b1.mean(1).tolist(), both.mean(1).tolist()], var_after=[b1.var(1).tolist(), both.var(1).tolist()],
                      count_after=[3, 6], atol=[1e-6, 1e-5]))
    c = np.array([[5, 5, 5], [3, 3, 3]], np.float32)
    cases.append(dict(name="zero_variance", cite="test/test_normalize_wrapper.jl:40-51", dims=2, batches=[c.T.tolist()], mean_after=[[5.0, 3.0]],
                      var_after=[[0.0, 0.0]], count_after=[3], atol=[1e-6]))
    cases.append(dict(name="single_sample", cite="test/test_normalize_wrapper.jl:53-59", dims=1, batches=[[[42.0]]], mean_after=[[42.0]], var_after=[[0.0]],
                      count_after=[1], atol=[1e-6]))
    cases.append(dict(name="scalar", cite="test/test_normalize_wrapper.jl:61-69", dims=1, batches=[[[1.0], [2.0], [3.0]]], mean_after=[[2.0]],
                      var_after=[[float(np.var([1.0, 2.0, 3.0]))]], count_after=[3], atol=[1e-6]))
    rng = np.random.default_rng(5)
    bs = [rng.normal(3, 2, (n, 4)).astype(np.float32) for n in (7, 1, 64, 33)]
    cat = [np.vstack(bs[:i + 1]).astype(np.float64) for i in range(len(bs))]
    cases.append(dict(name="random_merge", cite="src/environment_wrappers/normalizeWrapperEnv.jl:21-50", dims=4, batches=[b.tolist() for b in bs],
                      mean_after=[c.mean(0).tolist() for c in cat], var_after=[c.var(0).tolist() for c in cat],
                      count_after=[c.shape[0] for c in cat], atol=[1e-5] * 4))
    (OUT / "running_mean_std.json").write_text(json.dumps(cases, indent=1))


def make_param_counts():
    # test/test_policies.jl:36-64 (parameterlength == sum of leaf sizes); closed forms from src/layers/layer_helpers.jl:27-57
    def net(d, h1, h2, o):
        return d * h1 + h1 + h1 * h2 + h2 + h2 * o + o
    cases = [dict(env="CartPole", obs_dim=4, hidden=[64, 64], actor_out=2, discrete=True, total=net(4, 64, 64, 2) + net(4, 64, 64, 1)),
             dict(env="Pendulum", obs_dim=3, hidden=[64, 64], actor_out=1, discrete=False, total=2 * net(3, 64, 64, 1) + 1),
             dict(env="Pendulum", obs_dim=3, hidden=[256, 256], actor_out=1, discrete=False, total=2 * net(3, 256, 256, 1) + 1)]
    assert cases[0]["total"] == 9155 and cases[2]["total"] == 134147
    (OUT / "param_counts.json").write_text(json.dumps(cases, indent=1))


if __name__ == "__main__":
    make_gae(); make_distributions(); make_rms(); make_param_counts()
    print("wrote", sorted(p.name for p in OUT.glob("*.json")))

"""GPU (-m gpu): the LIBRARY's world_size > 1 code (every `reduce` branch of dril_api.hip: per-step advantage moments, the
[grads || 8 sums] all-reduce + grad-norm, the per-epoch moment table, NormalizeWrapperEnv's global batch moments, the
explained-variance sums, dril_train's iteration arithmetic) executed on ONE device.

RCCL refuses two ranks on one device, so the ranks are two handles of this process joined by the loopback communicator
(dril_debug_comm_loopback, include/dril_hip.h): same call sites, counts and dtypes, the transport is an in-process rendezvous
plus one summing kernel.  Each rank is driven from its own host thread, as separate processes would.

What is pinned (the reference has no distributed code, so the contract is north_star's):
  * replicas stay BITWISE identical (parameters, statistics) — every rank applies the same reduced numbers;
  * the 2-rank run equals ONE handle over the union of the shards (which the other GPU tests pin to the oracle): rank r owns
    global envs [r*E/2, (r+1)*E/2), seeded seed + global index; minibatch k of the union = the ranks' local minibatches k;
  * NormalizeWrapperEnv statistics equal the oracle's RunningMeanStd over ALL envs (normalizeWrapperEnv.jl:21-50).
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(pkg, kind, **kw):
    c = pkg._capi.default_config(kind)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _params(P, seed, scale=0.3):
    return (np.random.default_rng(seed).standard_normal(P) * scale).astype(np.float32)


def _ranks(pkg, kind, world, E_total, **kw):
    """world handles (rank r owns E_total / world envs) joined by the loopback communicator"""
    hs = [pkg.Handle(_cfg(pkg, kind, n_envs=E_total // world, rank=r, world_size=world, **kw)) for r in range(world)]
    pkg.Handle.comm_loopback(hs)
    assert all(h.comm_ranks() == world for h in hs)
    return hs


def _each(hs, fn):
    """fn(rank, handle) on one host thread per rank (the loopback all-reduce is a rendezvous); re-raises the first failure"""
    out, err = [None] * len(hs), [None] * len(hs)

    def run(r):
        try:
            out[r] = fn(r, hs[r])
        except BaseException as e:   # noqa: BLE001 - re-raised below
            err[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(len(hs))]
    [t.start() for t in ts]; [t.join() for t in ts]
    for e in err:
        if e is not None:
            raise e
    return out


def _union_perm(local_perms, E_total, world, T, B_global):
    """DataLoader order of ONE handle over the union whose minibatch k is the union of the ranks' local minibatches k.
    local_perms[r]: (epochs, N_local) buffer indices of rank r (time-major over its E_total / world envs)"""
    El = E_total // world; Nl = El * T; Bl = B_global // world
    epochs = local_perms[0].shape[0]
    out = np.zeros((epochs, El * world * T), np.int64)
    for ep in range(epochs):
        pos = 0
        for k0 in range(0, Nl, Bl):
            for r in range(world):
                loc = local_perms[r][ep, k0:k0 + Bl]
                t, e = loc // El, loc % El
                glob = t * E_total + r * El + e
                out[ep, pos:pos + glob.size] = glob; pos += glob.size
        assert pos == out.shape[1]
    return out


STATS = ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first", "entropy")


def _same_stats(a, b):
    return all(np.float32(getattr(a, f)).tobytes() == np.float32(getattr(b, f)).tobytes() for f in STATS) and \
        (a.n_updates, a.early_stopped) == (b.n_updates, b.early_stopped)


@pytest.mark.parametrize("kind,E,T,B,world,kw", [
    (0, 64, 24, 256, 2, {}),
    (0, 64, 13, 192, 2, {}),                                  # N_local = 416, B_local = 96: ragged last minibatch on every rank
    (1, 48, 20, 240, 2, {"ent_coef": 0.01}),                  # Pendulum: DiagGaussian + log_std gradient through the all-reduce
    (0, 96, 16, 384, 4, {}),                                  # four ranks
    (0, 64, 24, 128, 2, {"has_target_kl": 1, "target_kl": 0.003}),   # KL early stop decided identically on every rank
    (0, 4096, 64, 262144, 2, {}),                             # bench-scale minibatches: 4 096 tiles per rank = ppo_grad_pair_kernel with the unequal pair split, through the all-reduce
    (1, 4096, 64, 131072, 2, {"ent_coef": 0.01}),            # the same with DiagGaussian and two minibatches per epoch (2 048 tiles per rank: every pair gets tiles, the second pairs one fewer)
    (0, 2048, 32, 65536, 2, {"hidden1": 256, "hidden2": 256}),   # wide nets: ppo_grad_wide_split_kernel (Categorical head) through the all-reduce
    (1, 2048, 32, 65536, 2, {"hidden1": 128, "hidden2": 128, "ent_coef": 0.01}),
])
def test_two_ranks_equal_one_handle_over_the_union(pkg, oracle_mod, kind, E, T, B, world, kw):
    """rollout (no communication) + update with an injected DataLoader order: per-step advantage-moment all-reduce (3 doubles),
    [grads || 8 sums] all-reduce + grad norm, explained-variance reduce (dril_api.hip ppo_step / ppo_update)"""
    capi = pkg._capi
    common = dict(n_steps=T, batch_size=B, epochs=3, episode_len=11, **kw)
    one = pkg.Handle(_cfg(pkg, kind, n_envs=E, **common))
    hs = _ranks(pkg, kind, world, E, **common)
    El = E // world
    flat = _params(one.P, 77, 0.3)
    rng = np.random.default_rng(E + T)
    noise = rng.random((T, E)) if one.discrete else rng.standard_normal((T, E, one.A)).astype(np.float32)
    one.set_params(flat); one.env_reset(5); one.set_noise(noise.reshape(-1) if one.discrete else noise.reshape(T * E, one.A)); one.collect_rollout()

    def roll(r, h):
        h.set_params(flat); h.env_reset(5)                                   # the library offsets the seed by rank * n_envs (wrapper_utils.jl:39-44)
        nz = np.ascontiguousarray(noise[:, r * El:(r + 1) * El])
        h.set_noise(nz.reshape(-1) if h.discrete else nz.reshape(T * El, h.A)); h.collect_rollout()
    _each(hs, roll)
    # the shards' buffers are the union's buffer, bit for bit (same kernels, same per-env streams)
    for which in (capi.BUF_OBSERVATIONS, capi.BUF_ACTIONS, capi.BUF_REWARDS, capi.BUF_VALUES, capi.BUF_LOGPROBS, capi.BUF_ADVANTAGES, capi.BUF_RETURNS, capi.BUF_FLAGS):
        u = one.buffer(which).reshape(T, E, -1)
        for r, h in enumerate(hs):
            np.testing.assert_array_equal(h.buffer(which).reshape(T, El, -1), u[:, r * El:(r + 1) * El], err_msg=f"buffer {which} rank {r}")
    Nl = El * T
    local = [np.stack([np.random.default_rng(100 * r + ep).permutation(Nl) for ep in range(3)]).astype(np.int64) for r in range(world)]
    one.set_permutation(_union_perm(local, E, world, T, B))
    s1 = one.ppo_update()
    calls0 = [h.comm_allreduce_calls() for h in hs]

    def upd(r, h):
        h.set_permutation(local[r]); return h.ppo_update()
    st = _each(hs, upd)
    nb = -(-Nl // (B // world))
    for r, h in enumerate(hs):
        assert _same_stats(st[r], st[0]), f"rank {r} statistics differ from rank 0"
        np.testing.assert_array_equal(h.get_params(), hs[0].get_params(), err_msg=f"replica {r} diverged")          # bitwise
        if not kw.get("has_target_kl"):
            assert h.comm_allreduce_calls() - calls0[r] == 3 * nb * 2 + 1      # per step: moments + [grads || sums]; + explained variance
    assert (st[0].n_updates, st[0].early_stopped) == (s1.n_updates, s1.early_stopped)
    if kw.get("has_target_kl"):
        assert st[0].early_stopped and st[0].n_updates < 3 * nb
    for f in STATS:
        assert getattr(st[0], f) == pytest.approx(getattr(s1, f), rel=5e-4, abs=2e-6), f
    assert st[0].loss == pytest.approx(s1.loss, rel=1e-4)
    np.testing.assert_allclose(hs[0].get_params(), one.get_params(), rtol=2e-4, atol=3e-6)
    assert not np.array_equal(hs[0].get_params(), flat)


def test_keyed_order_and_epoch_moment_table(pkg, oracle_mod):
    """no injected order: every rank draws its shard-local keyed bijection (key = f(seed + rank, update, epoch)) and the advantage
    moments of ALL minibatches of an epoch go through ONE all-reduce of the (3 x nb) table (dril_api.hip ppo_update).  The union handle
    gets the same order injected (computed with the oracle's statement of the bijection), which sends it down the per-step moments path:
    two different routes to the same minibatch statistics"""
    capi = pkg._capi
    E, T, B, world, epochs = 64, 32, 256, 2, 2
    El, Nl = E // world, (E // world) * T
    common = dict(n_steps=T, batch_size=B, epochs=epochs, episode_len=500, seed=11)
    one = pkg.Handle(_cfg(pkg, 0, n_envs=E, **common)); hs = _ranks(pkg, 0, world, E, **common)
    flat = _params(one.P, 3, 0.2)
    one.set_params(flat); one.env_reset(11); one.collect_rollout()            # Philox sampling keyed by seed + global env index

    def roll(r, h):
        h.set_params(flat); h.env_reset(11); h.collect_rollout()
    _each(hs, roll)
    u = one.buffer(capi.BUF_ACTIONS).reshape(T, E)
    for r, h in enumerate(hs):
        np.testing.assert_array_equal(h.buffer(capi.BUF_ACTIONS).reshape(T, El), u[:, r * El:(r + 1) * El])
    L = oracle_mod.lib()
    local = []
    for r in range(world):
        rows = []
        for ep in range(epochs):
            key = L.orc_perm_key(common["seed"] + r, 0, ep)
            rows.append([L.orc_perm_index(p, Nl, key) for p in range(Nl)])
        local.append(np.asarray(rows, np.int64))
        assert all(sorted(row) == list(range(Nl)) for row in rows)
    one.set_permutation(_union_perm(local, E, world, T, B)); s1 = one.ppo_update()
    calls0 = [h.comm_allreduce_calls() for h in hs]
    st = _each(hs, lambda r, h: h.ppo_update())
    nb = Nl // (B // world)
    for r, h in enumerate(hs):
        assert h.comm_allreduce_calls() - calls0[r] == epochs * (1 + nb) + 1   # per epoch: the moment table; per step: [grads || sums]; + explained variance
        assert _same_stats(st[r], st[0])
        np.testing.assert_array_equal(h.get_params(), hs[0].get_params())
    for f in STATS:
        assert getattr(st[0], f) == pytest.approx(getattr(s1, f), rel=5e-4, abs=2e-6), f
    np.testing.assert_allclose(hs[0].get_params(), one.get_params(), rtol=2e-4, atol=3e-6)


@pytest.mark.parametrize("kind", [1, 0])
def test_normalize_wrapper_statistics_cover_all_ranks(pkg, oracle_mod, kind):
    """NormalizeWrapperEnv with > 1 rank: one 16-double all-reduce per env step carries the batch moments of EVERY rank's envs, so the
    running statistics equal the reference's single vector env over the union (normalizeWrapperEnv.jl:21-50,123-171): compared with the
    oracle's RunningMeanStd over all E envs and with one handle over the union; replicas hold bit-identical statistics"""
    capi = pkg._capi
    E, T, world = 64, 30, 2
    El = E // world
    common = dict(n_steps=T, episode_len=9, batch_size=E * T // 2, epochs=1, norm_training=1, norm_obs=1, norm_reward=1, clip_obs=5.0, clip_reward=2.0)
    cfg1 = _cfg(pkg, kind, n_envs=E, **common)
    one, o = pkg.Handle(cfg1), oracle_mod.Oracle(cfg1)
    hs = _ranks(pkg, kind, world, E, **common)
    flat = _params(one.P, 31, 0.4)
    rng = np.random.default_rng(2)
    one.set_params(flat); o.set_params(flat); one.env_reset(9); o.env_reset(9)
    _each(hs, lambda r, h: (h.set_params(flat), h.env_reset(9)))
    for rollout in range(2):
        noise = rng.random((T, E)) if one.discrete else rng.standard_normal((T, E, one.A)).astype(np.float32)
        flatn = noise.reshape(-1) if one.discrete else noise.reshape(T * E, one.A)
        one.set_noise(flatn); o.set_noise(flatn); one.collect_rollout(); o.collect_rollout()
        calls0 = [h.comm_allreduce_calls() for h in hs]

        def roll(r, h):
            nz = np.ascontiguousarray(noise[:, r * El:(r + 1) * El])
            h.set_noise(nz.reshape(-1) if h.discrete else nz.reshape(T * El, h.A)); h.collect_rollout()
            return h.norm_get_stats()
        sts = _each(hs, roll)
        om, ov, oc, rm, rv, rc = o.norm_stats()
        s1 = one.norm_get_stats()
        for r, s in enumerate(sts):
            assert hs[r].comm_allreduce_calls() - calls0[r] == T + 1             # observe at the start + one fused observe/act! per env step
            assert (s["obs_count"], s["ret_count"]) == (oc, rc) == (E * (T + 1) * (rollout + 1), E * T * (rollout + 1))    # counts cover ALL envs
            for k in ("obs_mean", "obs_var"):
                np.testing.assert_array_equal(s[k], sts[0][k])                 # replicas bit-identical
            assert (s["ret_mean"], s["ret_var"]) == (sts[0]["ret_mean"], sts[0]["ret_var"])
            np.testing.assert_allclose(s["obs_mean"], om, rtol=2e-5, atol=2e-6); np.testing.assert_allclose(s["obs_var"], ov, rtol=1e-4, atol=1e-6)
            assert s["ret_mean"] == pytest.approx(rm, rel=1e-4, abs=1e-5) and s["ret_var"] == pytest.approx(rv, rel=1e-4, abs=1e-5)
            np.testing.assert_allclose(s["obs_mean"], s1["obs_mean"], rtol=1e-5, atol=1e-6); np.testing.assert_allclose(s["obs_var"], s1["obs_var"], rtol=1e-5, atol=1e-6)
        if not one.discrete:                                                       # continuous control: no action flips, shards == union to fp32 noise
            for which, tol in ((capi.BUF_OBSERVATIONS, 1e-4), (capi.BUF_REWARDS, 2e-4), (capi.BUF_VALUES, 2e-4), (capi.BUF_ADVANTAGES, 2e-3)):
                u = one.buffer(which).reshape(T, E, -1)
                for r, h in enumerate(hs):
                    np.testing.assert_allclose(h.buffer(which).reshape(T, El, -1), u[:, r * El:(r + 1) * El], atol=tol, rtol=tol)
        st, sc = one.env_get_state(); o.env_set_state(st, sc)                      # teacher forcing for the oracle (as in test_gpu_parity)


def test_train_two_ranks_end_to_end(pkg):
    """dril_train with world_size = 2: iterations = max_steps / (T * E_local * world) (ppo.jl:117 over the whole job), Philox sampling,
    keyed shuffles, per-epoch moment table — replicas bitwise identical after every iteration, the job learns the same as one handle
    would in distribution (loss finite, parameters moved), and a second run reproduces the first bit for bit"""
    E, T, world = 64, 16, 2
    common = dict(n_steps=T, batch_size=256, epochs=2, seed=4)
    flat = _params(9155, 1, 0.05)
    runs = []
    for rep in range(2):
        hs = _ranks(pkg, 0, world, E, **common)
        _each(hs, lambda r, h: (h.set_params(flat), h.env_reset(4)))
        res = _each(hs, lambda r, h: h.train(3 * E * T + 7))                      # remainder steps dropped
        for r in range(world):
            stats, fps = res[r]
            assert len(stats) == 3 and all(f > 0 for f in fps)
            assert all(s.n_updates == 2 * (E * T // 256) for s in stats)
            assert all(_same_stats(a, b) for a, b in zip(stats, res[0][0]))
            assert np.isfinite([s.loss for s in stats]).all()
        p = [h.get_params() for h in hs]
        np.testing.assert_array_equal(p[0], p[1])
        assert not np.array_equal(p[0], flat)
        runs.append(p[0])
        [h.close() for h in hs]
    np.testing.assert_array_equal(runs[0], runs[1])


def test_configs3_per_rank_shape(pkg):
    """BASELINE.json configs[3] per-rank shape (65 536 envs per rank, sharded CartPole, [grads || sums] all-reduce per optimiser step) on two
    loopback ranks with a short rollout: size-independent properties — shards seeded seed + global index produce different data, ratio == 1 on
    the first minibatch, replicas bitwise identical, 4 optimiser steps per epoch of the GLOBAL minibatch"""
    capi = pkg._capi
    El, T, world = 65536, 32, 2
    common = dict(n_steps=T, batch_size=world * El * T // 4, epochs=2, episode_len=25, fixed_length_episodes=1)
    hs = _ranks(pkg, 0, world, world * El, **common)
    flat = _params(9155, 3, 0.3)
    _each(hs, lambda r, h: (h.set_params(flat), h.env_reset(42), h.collect_rollout()))
    a0, a1 = hs[0].buffer(capi.BUF_OBSERVATIONS), hs[1].buffer(capi.BUF_OBSERVATIONS)
    assert not np.array_equal(a0, a1)                                             # different global env indices -> different streams
    for h in hs:
        np.testing.assert_allclose(h.buffer(capi.BUF_RETURNS), h.buffer(capi.BUF_ADVANTAGES) + h.buffer(capi.BUF_VALUES), atol=1e-5)
    st = _each(hs, lambda r, h: h.ppo_update())
    assert st[0].n_updates == 8 and _same_stats(st[0], st[1])
    assert st[0].ratio_first == pytest.approx(1.0, abs=1e-5) and np.isfinite([st[0].loss, st[0].grad_norm, st[0].explained_variance]).all()
    np.testing.assert_array_equal(hs[0].get_params(), hs[1].get_params())
    assert not np.array_equal(hs[0].get_params(), flat)


def test_loopback_rejects_bad_groups(pkg):
    capi = pkg._capi
    a = pkg.Handle(_cfg(pkg, 0, n_envs=8, n_steps=4, batch_size=8, rank=0, world_size=2))
    b = pkg.Handle(_cfg(pkg, 0, n_envs=8, n_steps=4, batch_size=8, rank=0, world_size=2))    # duplicate rank
    with pytest.raises(pkg.DrilError):
        pkg.Handle.comm_loopback([a, b])
    with pytest.raises(pkg.DrilError) as e:                                               # world_size 2 without any communicator
        a.env_reset(1); a.collect_rollout(); a.ppo_update()
    assert e.value.code == capi.ERR_NOT_INITIALISED
    assert a.comm_ranks() == 1

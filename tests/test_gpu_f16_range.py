"""GPU (-m gpu): the RANGE of the f16-piece arithmetic and what the library does when a workload leaves it.

The fused kernels compute fp32-equivalent products from two f16 pieces per operand: fp32's precision, f16's range.  The reference asks no range of its user
(ppo.jl:213-214 only asserts finiteness; rewards are whatever the env returns), so
  * an update whose f16-piece step met a non-finite gradient is taken back and redone on the exact-f32 kernels (counted: dril_f32_retries, dril_ppo_stats.f32_path = 1);
  * after two consecutive redone updates the next 16 run the exact-f32 kernels directly (f32_path = 2), then one update probes f16 again;
  * a W2 entry beyond f16's range switches rollout / policy forwards to the f32-MFMA instantiations and the update to the exact-f32 kernels directly;
  * data-parallel: when ONE rank's shard overflows every rank redoes (the non-finite value survives the all-reduce), with equal all-reduce counts and bitwise replicas.
The checker is the CPU oracle (ppo.jl:365-407 restated) and the exact-f32 kernels of the same library (DRIL_GRAD_VARIANT=0), which must match BIT FOR BIT whenever
the exact-f32 path produced the result.
"""
import threading

import numpy as np
import pytest

import split_budget

pytestmark = pytest.mark.gpu
BUFS = ("BUF_OBSERVATIONS", "BUF_ACTIONS", "BUF_ADVANTAGES", "BUF_RETURNS", "BUF_LOGPROBS", "BUF_VALUES")
STATS = ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first", "entropy")


def _cfg(pkg, kind, **kw):
    c = pkg._capi.default_config(kind)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _handle(pkg, cfg, variant=None, env=None, monkeypatch=None):
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    with split_budget.grad_variant(variant):
        h = pkg.Handle(cfg)
    for k in (env or {}):
        monkeypatch.delenv(k, raising=False)
    return h


def _load(pkg, h, o, perm, overrides=None):
    capi = pkg._capi
    for name in BUFS:
        which = getattr(capi, name)
        h.set_buffer(which, (overrides or {}).get(name, o.buffer(which)))
    h.set_permutation(perm)


def _bits_equal_stats(a, b):
    return all(np.float32(getattr(a, f)).tobytes() == np.float32(getattr(b, f)).tobytes() for f in STATS) and (a.n_updates, a.early_stopped) == (b.n_updates, b.early_stopped)


# returns / old values far outside what a normalised reward gives: the value head's gradient 2 vf_coef (V - R) / B, scaled by the power of two ~ 4 B that places O(1)
# tiles where f16 is dense, passes 65 504 when |V - R| reaches ~ 1.6e4.  5e3 stays inside (no redo allowed to be needed, none forbidden), 3e5 cannot.
@pytest.mark.parametrize("kind,H,scale,must_redo", [
    (0, 64, 5e3, False), (0, 64, 3e5, True),
    (1, 64, 5e3, False), (1, 64, 3e5, True),
    (1, 256, 3e5, True), (0, 128, 3e5, True),
])
def test_large_returns_update_matches_the_oracle_and_the_exact_f32_kernels(pkg, oracle_mod, kind, H, scale, must_redo):
    capi = pkg._capi
    E, T = (2048, 128) if H == 64 else (512, 64)
    N = E * T
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, batch_size=N // 2, epochs=2, episode_len=25, hidden1=H, hidden2=H, ent_coef=0.01)
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(60 + kind).standard_normal(o.P) * (0.3 if H == 64 else 0.08)).astype(np.float32)
    o.set_params(flat); o.env_reset(5); o.collect_rollout()
    rng = np.random.default_rng(7)
    big = {"BUF_RETURNS": (rng.uniform(-1, 1, N) * scale).astype(np.float32), "BUF_VALUES": (rng.uniform(-1, 1, N) * scale).astype(np.float32)}
    o.set_buffer(capi.BUF_RETURNS, big["BUF_RETURNS"]); o.set_buffer(capi.BUF_VALUES, big["BUF_VALUES"])
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(2)]).astype(np.int64)
    o.set_permutation(perm)
    h, hx = _handle(pkg, cfg), _handle(pkg, cfg, "0")
    for x in (h, hx):
        x.set_params(flat); _load(pkg, x, o, perm, big)
    s, sx, so = h.ppo_update(), hx.ppo_update(), o.ppo_update()
    info = h.f32_fallback_info()
    print(f"[range] kind {kind} H {H} returns +-{scale:g}: f32_path {s.f32_path} retries {info['retries']} loss {s.loss:.6g} oracle {so.loss:.6g} grad_norm {s.grad_norm:.4g}")
    assert s.f32_path in (0, 1) and info["retries"] == s.f32_path == h.f32_retries() and info["direct_updates"] == 0
    if must_redo:
        assert s.f32_path == 1
    if s.f32_path == 1:                                    # the redo IS the exact-f32 path from the restored state: bit for bit what DRIL_GRAD_VARIANT=0 gives
        assert _bits_equal_stats(s, sx)
        np.testing.assert_array_equal(h.get_params(), hx.get_params())
        assert h.get_optimizer_state()["steps"] == hx.get_optimizer_state()["steps"] == 4
    assert (s.n_updates, s.early_stopped) == (so.n_updates, so.early_stopped) == (4, 0)
    for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first"):
        assert getattr(s, f) == pytest.approx(getattr(so, f), rel=5e-4, abs=2e-6), f
    assert s.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    h.close(); hx.close()


@pytest.mark.parametrize("log_std", [-3.0, -5.0])
def test_small_log_std_update(pkg, oracle_mod, log_std):
    """DiagGaussian with a tight sigma: d logp / d mu = (a - mu) / sigma^2 grows like e^(-2 log_std).  With the rollout's actions (drawn at sigma = 1) it is ~ 20 z / sigma
    at log_std = -3 (inside the gradient tiles' f16 range: no redo, and the f16-piece gradient must be as close to the oracle's as the exact-f32 kernel's) and ~ 2e4 z at
    log_std = -5 (outside: the update is redone and equals DRIL_GRAD_VARIANT=0 bit for bit).
    ONE optimiser step (batch_size = N): with sigma this tight a single Adam step moves log-probabilities by hundreds, so a second minibatch overflows exp() in ANY arithmetic
    (the reference's assert fires there too).  At log_std = -5 log p ~ -2e5, where float32 resolves 0.016: ratios differ by per cents between any two f32 evaluation orders,
    so beyond the loss (value-loss dominated) the oracle is no checker there — the exact-f32 kernels are."""
    capi = pkg._capi
    E, T = 2048, 128
    N = E * T
    cfg = _cfg(pkg, 1, n_envs=E, n_steps=T, batch_size=N, epochs=1, episode_len=25, ent_coef=0.01)
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(61).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(5); o.collect_rollout()
    tight = flat.copy()
    tight[-1] = log_std                                   # the flat layout is {actor net, critic net, log_std (A)} (include/dril_hip.h)
    obs, act = o.buffer(capi.BUF_OBSERVATIONS).reshape(N, -1), o.buffer(capi.BUF_ACTIONS).reshape(N, -1)
    o.set_params(tight)
    _, lp, _ = o.evaluate_actions(obs, act)
    lp_old = (lp + np.random.default_rng(3).normal(0, 0.1, N)).astype(np.float32)     # old log-probabilities of the SAME tight policy: ratios near 1, a tenth of them clipped
    o.set_buffer(capi.BUF_LOGPROBS, lp_old)
    perm = np.random.default_rng(0).permutation(N).astype(np.int64)[None]
    o.set_permutation(perm)
    h, hx = _handle(pkg, cfg), _handle(pkg, cfg, "0")
    for x in (h, hx):
        x.set_params(tight); _load(pkg, x, o, perm)
    if log_std == -3.0:                                   # the gradient itself, f16-piece kernel and exact-f32 kernel against the oracle's
        batch = (obs, act, o.buffer(capi.BUF_ADVANTAGES), o.buffer(capi.BUF_RETURNS), lp_old, o.buffer(capi.BUF_VALUES))
        (lh, _, gh), (lx, _, gx), (lo, _, go) = h.ppo_loss_grad(*batch), hx.ppo_loss_grad(*batch), o.ppo_loss_grad(*batch)
        assert h.grad_kernel_info().split(":")[0] == "ppo_grad_pair_kernel" and hx.grad_kernel_info().split(":")[0] == "ppo_grad_kernel"
        eh, ex = np.linalg.norm(gh - go) / np.linalg.norm(go), np.linalg.norm(gx - go) / np.linalg.norm(go)
        print(f"[range] log_std -3 gradient vs oracle: f16 pieces {eh:.2e}, exact-f32 kernel {ex:.2e}; loss rel {abs(lh - lo) / abs(lo):.1e}")
        assert np.isfinite(gh).all() and eh <= max(2.0 * ex, 2e-4) and lh == pytest.approx(lo, rel=1e-4)
    s, sx, so = h.ppo_update(), hx.ppo_update(), o.ppo_update()
    print(f"[range] log_std {log_std}: f32_path {s.f32_path} loss {s.loss:.6g} oracle {so.loss:.6g} grad_norm {s.grad_norm:.4g} (exact f32 {sx.grad_norm:.4g}, oracle {so.grad_norm:.4g})")
    assert s.f32_path == (1 if log_std == -5.0 else 0)
    assert s.n_updates == sx.n_updates == so.n_updates == 1 and np.isfinite(h.get_params()).all()
    assert s.loss == pytest.approx(so.loss, rel=1e-4)
    if s.f32_path == 1:
        assert _bits_equal_stats(s, sx)
        np.testing.assert_array_equal(h.get_params(), hx.get_params())
    else:
        assert s.grad_norm == pytest.approx(sx.grad_norm, rel=5e-4) and s.grad_norm == pytest.approx(so.grad_norm, rel=2e-3)
    h.close(); hx.close()


def test_ragged_tail_does_not_hide_the_overflow(pkg, oracle_mod):
    """the redo was gated on the kernel of the LAST optimiser step; with N % B != 0 and a tail below 8 tiles per CU the last step runs the exact-f32 kernel although
    the earlier, overflowing ones ran the pair kernel — the overflow then surfaced as the reference's NaN error.  The gate is a sticky per-update flag now"""
    capi = pkg._capi
    E, T = 2048, 130
    N = E * T                                            # 266 240 = 2 x 131 072 + 4 096: two pair-kernel minibatches and a 128-tile tail (f32 kernel)
    cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, batch_size=131072, epochs=1, episode_len=25)
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(2).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(5); o.collect_rollout()
    big = {"BUF_RETURNS": (np.random.default_rng(7).uniform(-1, 1, N) * 3e5).astype(np.float32)}
    o.set_buffer(capi.BUF_RETURNS, big["BUF_RETURNS"])
    perm = np.random.default_rng(0).permutation(N).astype(np.int64)[None]
    o.set_permutation(perm)
    h = _handle(pkg, cfg)
    h.set_params(flat); _load(pkg, h, o, perm, big)
    s, so = h.ppo_update(), o.ppo_update()
    assert s.f32_path == 1 and h.f32_retries() == 1 and s.n_updates == so.n_updates == 3
    assert h.grad_kernel_info().split(":")[0] == "ppo_grad_kernel"
    assert s.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    assert h.lib.dril_last_error(h._h).decode() == ""    # the first pass's NaN message does not outlive the successful redo
    h.close()


def test_latch_after_repeated_redos_and_reprobe(pkg, oracle_mod):
    """a workload that overflows every update must not pay an f16 pass + snapshot + redo forever: after 2 consecutive redone updates the next 16 run the exact-f32
    kernels directly (f32_path 2, retries unchanged), the 19th probes f16 again (redone: the buffer still overflows) and re-arms the latch at once; when the data come
    back into range the probe succeeds, the streak ends and f16 is the default again.  Every update equals DRIL_GRAD_VARIANT=0 bit for bit while out of range."""
    capi = pkg._capi
    E, T = 2048, 64
    N = E * T
    cfg = _cfg(pkg, 0, n_envs=E, n_steps=T, batch_size=N, epochs=1, episode_len=25)
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(2).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(5); o.collect_rollout()
    small = o.buffer(capi.BUF_RETURNS).copy()
    big = {"BUF_RETURNS": (np.random.default_rng(7).uniform(-1, 1, N) * 3e5).astype(np.float32)}
    perm = np.random.default_rng(0).permutation(N).astype(np.int64)[None]
    h, hx = _handle(pkg, cfg), _handle(pkg, cfg, "0")
    for x in (h, hx):
        x.set_params(flat); _load(pkg, x, o, perm, big)
    paths = []
    for i in range(20):
        s, sx = h.ppo_update(), hx.ppo_update()
        paths.append(s.f32_path)
        assert _bits_equal_stats(s, sx), i
    np.testing.assert_array_equal(h.get_params(), hx.get_params())
    assert paths == [1, 1] + [2] * 16 + [1] + [2], paths
    info = h.f32_fallback_info()
    assert (info["retries"], info["direct_updates"], info["latch_updates_left"]) == (3, 17, 15)
    # back in range: the latch runs out (15 more direct updates), the probe succeeds on f16, and stays there
    h.set_buffer(capi.BUF_RETURNS, small)
    paths = [h.ppo_update().f32_path for _ in range(18)]
    assert paths == [2] * 15 + [0, 0, 0], paths
    assert h.grad_kernel_info().split(":")[0] == "ppo_grad_pair_kernel" and h.f32_retries() == 3
    h.close(); hx.close()


@pytest.mark.parametrize("kind,H", [(0, 64), (1, 64), (1, 256), (6, 128)])
def test_w2_beyond_f16_range_switches_forward_and_update_to_f32(pkg, oracle_mod, kind, H):
    """the forward of rollout / policy kernels puts kTanhScale kWScale W2 on f16 pieces; |W2| >= 350 gave hi = Inf, lo = -Inf -> NaN values and log-probs, and the
    f32 redo of the update then read the poisoned buffer.  max |W2| is tracked on the host (dril_set_params; after each update with its statistics) and such parameters run
    the f32-MFMA instantiations of the same kernels: rollout, evaluate_actions and the update equal the oracle, and the update is not an f16 attempt at all (f32_path 2)"""
    capi = pkg._capi
    E, T = 256, 16
    N = E * T
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, batch_size=N // 2, epochs=2, episode_len=7, hidden1=H, hidden2=H, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(9).standard_normal(h.P) * (0.3 if H == 64 else 0.08)).astype(np.float32)
    D, A = h.D, h.A
    sl = split_budget.w2_slices(D, H, A, h.discrete)
    h.set_params(flat)
    assert h.f32_fallback_info()["forward_exact_f32"] == 0
    big = flat.copy(); big[sl[0].start + 5] = 1000.0; big[sl[1].start + 70] = -4000.0
    h.set_params(big); o.set_params(big)
    info = h.f32_fallback_info()
    assert info["forward_exact_f32"] == 1 and info["max_abs_w2"] == 4000.0
    rng = np.random.default_rng(1)
    noise = rng.random(N) if h.discrete else rng.standard_normal((N, A)).astype(np.float32)
    h.env_reset(3); o.env_reset(3); h.set_noise(noise); o.set_noise(noise)
    h.collect_rollout(); o.collect_rollout()
    for which, tol in ((capi.BUF_VALUES, 2e-4), (capi.BUF_LOGPROBS, 2e-4), (capi.BUF_ADVANTAGES, 2e-3), (capi.BUF_RETURNS, 2e-3)):
        a, b = h.buffer(which), o.buffer(which)
        assert np.isfinite(a).all(), which
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(1.0, float(np.abs(b).max())), err_msg=str(which))
    obs = rng.uniform(-1, 1, (300, D)).astype(np.float32)
    v_h, v_o = h.predict_values(obs), o.predict_values(obs)
    assert np.isfinite(v_h).all()
    np.testing.assert_allclose(v_h, v_o, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(v_o).max())))
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(2)]).astype(np.int64)
    for name in BUFS:                                     # the update on the oracle's own rollout (sampling at CDF edges may differ by a few actions between the two)
        h.set_buffer(getattr(capi, name), o.buffer(getattr(capi, name)))
    h.set_permutation(perm); o.set_permutation(perm)
    s, so = h.ppo_update(), o.ppo_update()
    assert s.f32_path == 2 and h.f32_retries() == 0 and h.f32_fallback_info()["direct_updates"] == 1
    assert s.n_updates == so.n_updates and s.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    assert h.f32_fallback_info()["forward_exact_f32"] == 1          # the update's read-back: the big entries are still there
    h.set_params(flat)
    assert h.f32_fallback_info()["forward_exact_f32"] == 0
    h.close()


def test_grad_variant_0_runs_the_f32_forward_too(pkg):
    """DESIGN: DRIL_GRAD_VARIANT=0 = the exact-f32 kernels everywhere — the update's AND the rollout / policy forwards (it used to leave the forward on f16 pieces)"""
    cfg = _cfg(pkg, 0, n_envs=64, n_steps=8, batch_size=64, epochs=1)
    with split_budget.grad_variant("0"):
        h = pkg.Handle(cfg)
    h.set_params((np.random.default_rng(0).standard_normal(h.P) * 0.3).astype(np.float32))
    assert h.f32_fallback_info()["forward_exact_f32"] == 1
    h.env_reset(1); h.collect_rollout()
    assert np.isfinite(h.buffer(pkg._capi.BUF_VALUES)).all()
    h.close()


def _each(hs, fn):
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


@pytest.mark.parametrize("kind,H,E,T,B", [(0, 64, 4096, 64, 262144), (1, 256, 1024, 32, 32768)])
def test_rank_local_overflow_is_redone_on_every_rank(pkg, oracle_mod, kind, H, E, T, B):
    """two loopback ranks (the library's world_size > 1 code on one device, tests/test_gpu_dataparallel.py), only rank 1's shard holds returns beyond f16's range: the
    non-finite gradient survives the all-reduce, so BOTH ranks flag the step, both restore and redo on the exact-f32 kernels — same number of all-reduces on both (a
    divergence here would be an RCCL deadlock, not a wrong number), bitwise-identical replicas, and the result of ONE handle over the union of the shards"""
    from test_gpu_dataparallel import _ranks, _union_perm, _same_stats
    capi = pkg._capi
    world, El = 2, E // 2
    common = dict(n_steps=T, batch_size=B, epochs=2, episode_len=11, hidden1=H, hidden2=H)
    one = pkg.Handle(_cfg(pkg, kind, n_envs=E, **common))
    hs = _ranks(pkg, kind, world, E, **common)
    flat = (np.random.default_rng(77).standard_normal(one.P) * (0.3 if H == 64 else 0.08)).astype(np.float32)
    one.set_params(flat); one.env_reset(5); one.collect_rollout()

    def roll(r, h):
        h.set_params(flat); h.env_reset(5); h.collect_rollout()
    _each(hs, roll)
    ret = one.buffer(capi.BUF_RETURNS).reshape(T, E).copy()
    ret[:, El:] = np.random.default_rng(3).uniform(-1, 1, (T, E - El)).astype(np.float32) * 3e5        # rank 1's envs only
    one.set_buffer(capi.BUF_RETURNS, ret.reshape(-1))
    hs[1].set_buffer(capi.BUF_RETURNS, np.ascontiguousarray(ret[:, El:]).reshape(-1))
    Nl = El * T
    local = [np.stack([np.random.default_rng(100 * r + ep).permutation(Nl) for ep in range(2)]).astype(np.int64) for r in range(world)]
    one.set_permutation(_union_perm(local, E, world, T, B))
    s1 = one.ppo_update()
    calls0 = [h.comm_allreduce_calls() for h in hs]

    def upd(r, h):
        h.set_permutation(local[r]); return h.ppo_update()
    st = _each(hs, upd)
    assert s1.f32_path == 1
    assert [s.f32_path for s in st] == [1, 1] and [h.f32_retries() for h in hs] == [1, 1]
    calls = [h.comm_allreduce_calls() - c for h, c in zip(hs, calls0)]
    assert calls[0] == calls[1]
    assert _same_stats(st[0], st[1])
    np.testing.assert_array_equal(hs[0].get_params(), hs[1].get_params())
    assert (st[0].n_updates, st[0].early_stopped) == (s1.n_updates, s1.early_stopped)
    assert st[0].loss == pytest.approx(s1.loss, rel=1e-4)
    np.testing.assert_allclose(hs[0].get_params(), one.get_params(), rtol=2e-4, atol=3e-6)
    # and the next update of the group (in range again) runs f16 on both ranks
    hs[1].set_buffer(capi.BUF_RETURNS, np.ascontiguousarray(one.buffer(capi.BUF_ADVANTAGES).reshape(T, E)[:, El:]).reshape(-1))
    st2 = _each(hs, lambda r, h: h.ppo_update())
    assert [s.f32_path for s in st2] == [0, 0]
    np.testing.assert_array_equal(hs[0].get_params(), hs[1].get_params())
    for x in hs + [one]:
        x.close()

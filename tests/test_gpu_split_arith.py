"""GPU (-m gpu): the operand split of the update kernels — since the end of round 3 TWO f16 pieces and three products per k16 step (dril_device.h; rounds 2 - 3:
three bf16 pieces, six products) — at the minibatch sizes where the size rule selects those kernels,
(1) directly against the CPU oracle (ppo.jl:365-407 + hand-written backward), (2) against a float64 gradient with an error budget relative to
the exact-f32 kernel, (3) with a negative control: the same library built with the `lo` products dropped (an 11-bit product; 16-bit in the bf16 form)
must BREAK the budget — which is what shows that (2) can tell a 24-bit product from a shorter one, (4) at the full minibatch size of the headline
config through additivity.

Tolerances: loss 1e-4 rel (BASELINE.json north_star), gradient within 2e-4 of its norm, as for the f32 kernel in tests/test_gpu_parity.py.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import split_budget

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
BUFS = ("BUF_OBSERVATIONS", "BUF_ACTIONS", "BUF_ADVANTAGES", "BUF_RETURNS", "BUF_LOGPROBS", "BUF_VALUES")


def _cfg(pkg, kind, **kw):
    c = pkg._capi.default_config(kind)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _batch(oracle, cfg, B, seed):
    rng = np.random.default_rng(seed)
    obs = rng.uniform(-1, 1, (B, oracle.D)).astype(np.float32)
    act = (rng.integers(0, oracle.A, B) + cfg.action_start).astype(np.int32) if oracle.discrete else rng.normal(0, 1, (B, oracle.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    _, lp, _ = oracle.evaluate_actions(obs, act)
    return obs, act, adv, ret, (lp + rng.normal(0, 0.1, B)).astype(np.float32), ov


EXPECTED = {64: "ppo_grad_pair_kernel", 128: "ppo_grad_wide_split_kernel", 256: "ppo_grad_wide_split_kernel"}


@pytest.mark.parametrize("kind,H,B,variant", [
    (0, 64, 65536, "default"),             # 2 048 tiles = 8 tiles per CU: the smallest minibatch the size rule gives to the pair kernel (one tile per pair)
    (1, 64, 65536 + 33, "ent_vfclip"),
    (0, 64, 131072, "default"),
    (1, 64, 131072, "ent_vfclip"),
    (0, 64, 262144 + 45, "ent_vfclip"),    # multi-trip loop with unequal actor / critic pair counts and a ragged last tile
    (1, 64, 262144, "default"),
    (3, 64, 131072 + 1, "default"),        # MountainCar: Categorical(3), D = 2
    (6, 64, 131072 + 77, "ent_vfclip"),    # Acrobot: Categorical(3), D = 6 — four first-layer k-steps, three-quad records, dW1 / db1 through the dz1 image on the matrix cores
    (6, 64, 65536, "default"),
    (1, 256, 32768, "default"),            # configs[2] shape: 8 tiles per workgroup of the chip-filling grid
    (0, 256, 16384 + 19, "ent_vfclip"),
    (0, 128, 32768, "default"),
    (1, 128, 32768 + 7, "ent_vfclip"),
])
def test_selected_kernel_loss_and_gradient_vs_oracle(pkg, oracle_mod, kind, H, B, variant):
    """dril_ppo_loss_grad with the library's own kernel selection (asserted through dril_grad_kernel_info) against orc_ppo_loss_grad"""
    kw = dict(n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H)
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    cfg = _cfg(pkg, kind, **kw)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(50 + kind).standard_normal(h.P) * (0.25 if H == 64 else 0.08)).astype(np.float32)
    h.set_params(flat); o.set_params(flat)
    batch = _batch(o, cfg, B, 3)
    lh, sh, gh = h.ppo_loss_grad(*batch); lo, so, go = o.ppo_loss_grad(*batch)
    assert h.grad_kernel_info().split(":")[0] == EXPECTED[H]
    assert lh == pytest.approx(lo, rel=1e-4)
    np.testing.assert_allclose(sh, so, rtol=2e-4, atol=2e-6)
    rel = np.linalg.norm(gh - go) / np.linalg.norm(go)
    print(f"[split vs oracle] kind {kind} H {H} B {B}: loss rel {abs(lh - lo) / abs(lo):.2e}, |dg|/|g| {rel:.2e}")
    assert rel <= 2e-4
    lh2, _, gh2 = h.ppo_loss_grad(*batch)
    assert lh2 == lh and np.array_equal(gh, gh2)                     # deterministic slabs


@pytest.mark.parametrize("kind,H,scale", [(0, 64, 3e-3), (1, 64, 3e-3), (0, 64, 1.0), (1, 64, 1.0), (1, 256, 1e-3), (0, 128, 1.0)])
def test_f16_pieces_across_the_operand_range(pkg, oracle_mod, kind, H, scale):
    """the f16 pieces have a RANGE (dril_device.h): operands are scaled so that weights of O(0.1) and activations / gradient tiles of O(1) sit where hi AND lo are
    normal f16 numbers.  Away from that — weights 100 x smaller (every `lo` piece subnormal: the representation error becomes an absolute 2^-25 of the scaled value)
    or 4 x larger (saturated tanh units, pre-activations of tens; beyond that the Categorical head itself produces 0 log 0) — the gradient must still agree with the
    oracle to the tolerance of the other parity tests"""
    B = 131072 if H == 64 else 32768
    cfg = _cfg(pkg, kind, n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(77 + kind).standard_normal(h.P) * scale * (1.0 if H == 64 else 0.3)).astype(np.float32)
    h.set_params(flat); o.set_params(flat)
    batch = _batch(o, cfg, B, 9)
    lh, sh, gh = h.ppo_loss_grad(*batch); lo, so, go = o.ppo_loss_grad(*batch)
    assert h.grad_kernel_info().split(":")[0] == EXPECTED[H]
    rel = np.linalg.norm(gh - go) / np.linalg.norm(go)
    print(f"[range] kind {kind} H {H} weight scale {scale}: loss rel {abs(lh - lo) / abs(lo):.2e}, |dg|/|g| {rel:.2e}")
    assert lh == pytest.approx(lo, rel=1e-4) and rel <= 2e-4


def test_an_update_outside_the_f16_range_is_redone_on_the_exact_f32_kernels(pkg, oracle_mod, monkeypatch):
    """what f32 has and the f16 pieces do not is range: a staged weight beyond ~ 350 overflows its hi piece, and the step that meets it reports a non-finite
    gradient without touching the parameters.  dril_ppo_update then takes the update back (parameters, Adam moments, beta powers, counters) and redoes it on the
    exact-f32 kernels (dril_api.hip ppo_update): the result is the one DRIL_GRAD_VARIANT=0 gives from the same state, bit for bit, and equals the oracle's; with
    DRIL_NO_F32_RETRY=1 the non-finite gradient surfaces as the reference's error (ppo.jl:213-214) — never a silently wrong update"""
    capi = pkg._capi
    cfg = _cfg(pkg, 0, n_envs=4096, n_steps=64, batch_size=4096 * 64 // 2, epochs=2, episode_len=25)
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(5).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(3); o.collect_rollout()
    big = flat.copy(); big[4 * 64 + 64 + 5] = 1000.0                                  # one W2 entry of the actor
    perm = np.stack([np.random.default_rng(e).permutation(4096 * 64) for e in range(2)]).astype(np.int64)

    def run(variant, no_retry=False):
        if no_retry:
            monkeypatch.setenv("DRIL_NO_F32_RETRY", "1")
        with split_budget.grad_variant(variant):
            h = pkg.Handle(cfg)
        monkeypatch.delenv("DRIL_NO_F32_RETRY", raising=False)
        h.set_params(big)
        for which in BUFS:
            h.set_buffer(getattr(capi, which), o.buffer(getattr(capi, which)))
        h.set_permutation(perm)
        return h

    h = run(None, no_retry=True)
    with pytest.raises(Exception) as ei:
        h.ppo_update()
    assert "nan" in str(ei.value).lower() and h.grad_kernel_info().split(":")[0] == "ppo_grad_pair_kernel"
    np.testing.assert_array_equal(h.get_params(), big)                                # the step that met the overflow did not touch the parameters
    h.close()
    ha, hb = run(None), run("0")
    sa, sb = ha.ppo_update(), hb.ppo_update()
    assert ha.grad_kernel_info().split(":")[0] == hb.grad_kernel_info().split(":")[0] == "ppo_grad_kernel"
    assert (sa.n_updates, sa.loss, sa.grad_norm) == (sb.n_updates, sb.loss, sb.grad_norm) and sa.n_updates == 4 and np.isfinite(sa.loss)
    np.testing.assert_array_equal(ha.get_params(), hb.get_params())
    o.set_params(big); o.set_permutation(perm); so = o.ppo_update()
    assert sa.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(ha.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)
    # the next update of the same handle starts from the redone state: optimiser state and counters were restored consistently
    sa2, sb2 = ha.ppo_update(), hb.ppo_update()
    assert np.isfinite(sa2.loss) and sa2.loss == sb2.loss
    np.testing.assert_array_equal(ha.get_params(), hb.get_params())
    ha.close(); hb.close()


@pytest.mark.parametrize("kind,H,variant", [(0, 64, "ent_vfclip"), (1, 64, "default"), (1, 256, "default")])
def test_full_size_minibatch_is_the_weighted_mean_of_its_halves(pkg, kind, H, variant):
    """the headline minibatch size of BASELINE configs[1] / configs[2] (B = 65 536 x 2048 / 32 = 4 194 304 samples per launch: 256 tiles per pair of ppo_grad_pair_kernel,
    the multi-trip loop with the unequal actor / critic division of the chip) is too large for the oracle to be the checker in a test, but the loss is a MEAN over the
    minibatch (ppo.jl:382-386), so with normalize_advantage off loss / statistics / gradient of the whole minibatch must be the sample-weighted mean of those of two
    unequal parts — each of which is the size the direct oracle tests above cover per tile, run through different tile -> pair assignments and slab reductions.
    f32 sums in different orders: loss and gradient to 2e-6 of their size (measured: loss 1e-9 ... 6e-8, gradient 1.1e-7 ... 1.4e-7 of its norm; tests/diag/full_size_additivity.py)"""
    B = 4194304
    kw = dict(n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H, normalize_advantage=0)
    if variant == "ent_vfclip":
        kw.update(ent_coef=0.01, has_clip_range_vf=1, clip_range_vf=0.3, clip_range=0.1)
    cfg = _cfg(pkg, kind, **kw)
    h = pkg.Handle(cfg)
    rng = np.random.default_rng(5)
    flat = rng.uniform(-0.3, 0.3, h.P).astype(np.float32) if H == 64 else rng.uniform(-0.08, 0.08, h.P).astype(np.float32)
    h.set_params(flat)
    D, A, disc = h.D, h.A, h.discrete
    obs = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    act = (rng.integers(0, A, B) + cfg.action_start).astype(np.int32) if disc else rng.normal(0, 1, (B, A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
    olp = rng.normal(-0.7, 0.1, B).astype(np.float32)                                  # old log-probabilities near those of a fresh policy: ratios on both sides of the clip
    n1 = 1572864 + 37                                                                 # unequal parts, the first one with a ragged last tile
    full = h.ppo_loss_grad(obs, act, adv, ret, olp, ov)
    assert h.grad_kernel_info().split(":")[0] == EXPECTED[H]
    parts = [h.ppo_loss_grad(obs[a:b], act[a:b], adv[a:b], ret[a:b], olp[a:b], ov[a:b]) for a, b in ((0, n1), (n1, B))]
    w = np.array([n1, B - n1], np.float64) / B
    loss = w[0] * parts[0][0] + w[1] * parts[1][0]
    stats = w[0] * parts[0][1].astype(np.float64) + w[1] * parts[1][1].astype(np.float64)
    grad = w[0] * parts[0][2].astype(np.float64) + w[1] * parts[1][2].astype(np.float64)
    assert np.isfinite(full[0]) and abs(full[0] - loss) <= 2e-6 * max(1.0, abs(loss))
    np.testing.assert_allclose(full[1], stats, rtol=2e-5, atol=2e-6)
    gn = np.linalg.norm(grad)
    assert gn > 0 and np.linalg.norm(full[2].astype(np.float64) - grad) <= 2e-6 * gn
    h.close()


@pytest.mark.parametrize("kind,H,E,T", [(0, 64, 2048, 128), (1, 64, 2048, 128), (1, 256, 512, 64), (0, 128, 512, 64), (6, 256, 512, 64), (6, 128, 512, 64)])
def test_selected_kernel_update_vs_oracle(pkg, oracle_mod, kind, H, E, T):
    """one dril_ppo_update (2 epochs x 2 minibatches of N/2 samples: 131 072 at hidden 64, i.e. the pair kernel by the size rule) on the oracle's rollout and an
    injected DataLoader order: statistics and parameters after four Adam steps against orc_ppo_update"""
    capi = pkg._capi
    N = E * T
    cfg = _cfg(pkg, kind, n_envs=E, n_steps=T, batch_size=N // 2, epochs=2, episode_len=25, hidden1=H, hidden2=H, ent_coef=0.01)
    h, o = pkg.Handle(cfg), oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(60 + kind).standard_normal(h.P) * (0.3 if H == 64 else 0.08)).astype(np.float32)
    h.set_params(flat); o.set_params(flat)
    o.env_reset(5); o.collect_rollout()
    for name in BUFS:
        h.set_buffer(getattr(capi, name), o.buffer(getattr(capi, name)))
    perm = np.stack([np.random.default_rng(e).permutation(N) for e in range(2)]).astype(np.int64)
    h.set_permutation(perm); o.set_permutation(perm)
    sh, so = h.ppo_update(), o.ppo_update()
    assert h.grad_kernel_info().split(":")[0] == EXPECTED[H]
    assert (sh.n_updates, sh.early_stopped) == (so.n_updates, so.early_stopped) == (4, 0)
    for f in ("policy_loss", "value_loss", "entropy_loss", "approx_kl_div", "clip_fraction", "loss", "grad_norm", "explained_variance", "ratio_first"):
        assert getattr(sh, f) == pytest.approx(getattr(so, f), rel=5e-4, abs=2e-6), f
    assert sh.loss == pytest.approx(so.loss, rel=1e-4)
    np.testing.assert_allclose(h.get_params(), o.get_params(), rtol=2e-4, atol=3e-6)


BUDGET_CASES = [(0, 64, 4096, 2), (1, 64, 4096, 2), (0, 64, 131072, None), (1, 64, 131072, None), (1, 256, 8192, None), (0, 128, 8192, None)]


@pytest.mark.parametrize("kind,H,B,forced", BUDGET_CASES)
def test_split_error_budget_vs_float64(pkg, oracle_mod, kind, H, B, forced):
    """‖g_split − g_f64‖ <= 2 ‖g_f32kernel − g_f64‖ and the same for the loss; the dW2 error has no bias beyond its spread and no shrinkage (split_budget.within_budget)"""
    m = split_budget.measure(pkg, oracle_mod, kind, H, B, forced)
    print("[budget]", json.dumps(m))
    assert m["kernel_split"] == EXPECTED[H] and m["kernel_f32"] in ("ppo_grad_kernel", "ppo_grad_wide_kernel")
    ok_grad, ok_loss, ok_bias = split_budget.within_budget(m)
    assert ok_grad, (m["grad_err_split"], m["grad_err_f32"])
    assert ok_loss, (m["loss_err_split"], m["loss_err_f32"])
    assert ok_bias, (m["dW2_mean_err_split"], m["dW2_std_err_split"], m["dW2_shrink_split"], m["dW2_shrink_f32"])
    assert m["grad_err_split"] <= 5e-7                                    # absolute: fp32-level agreement with float64 (measured 1.0 - 1.5e-7, profiles/r03_split_arith.md section 6)


@pytest.mark.parametrize("kind,H,B,scale", [(0, 64, 131072, 1e3), (1, 64, 131072, 1e3), (0, 64, 131072, 1e-3), (1, 64, 131072, 1e-3), (1, 256, 8192, 1e3), (0, 128, 8192, 1e-3)])
def test_split_error_budget_away_from_benign_inputs(pkg, oracle_mod, kind, H, B, scale):
    """the same measurement with advantages / returns / old values of N(0, scale^2), scale 1e3 (un-normalised rewards: the value head's gradient tiles are 1e3 x the benign
    case, still inside f16's range) and 1e-3 (tiny returns: gradient tiles whose `lo` pieces approach f16's subnormals).  Per-minibatch advantage normalisation cancels the
    scale of the advantages, not that of the returns.  No redo may be involved: dril_ppo_loss_grad runs the selected kernel once.
    At 1e-3 the full budget of the benign case holds.  At 1e3 the gradient is dominated by the critic's sums of +-1e3 terms that cancel to ~1 / 360 of their absolute sum,
    and EVERY f32 kernel loses digits there (the exact-f32 kernel is 5 x further from float64 than at scale 1: 6.2e-7).  What is asked of the f16 pieces is asked where
    they act — the sample contraction dW2 = dz2' h1: its error spread must match the exact-f32 kernel's (measured 4.4e-8 vs 4.1e-8) — while the whole gradient, whose other
    blocks are f32 sums in another order in the pair kernel (per-lane accumulators over a workgroup's tiles instead of MFMA accumulation), must stay at fp32 level: within
    4 x the exact-f32 kernel's distance and 5e-6 of the gradient norm (measured 1.0e-6 - 1.6e-6)."""
    m = split_budget.measure(pkg, oracle_mod, kind, H, B, None, scale=scale)
    print("[budget, scaled]", json.dumps(m))
    assert m["kernel_split"] == EXPECTED[H] and np.isfinite(m["grad_err_split"])
    ok_grad, ok_loss, _ = split_budget.within_budget(m)
    assert ok_loss, (m["loss_err_split"], m["loss_err_f32"])
    if scale < 1.0:
        assert ok_grad and m["grad_err_split"] <= 5e-7, (m["grad_err_split"], m["grad_err_f32"])
    else:
        assert m["dW2_std_err_split"] <= 1.25 * m["dW2_std_err_f32"], (m["dW2_std_err_split"], m["dW2_std_err_f32"])
        assert m["grad_err_split"] <= 4.0 * m["grad_err_f32"] and m["grad_err_split"] <= 5e-6, (m["grad_err_split"], m["grad_err_f32"])


def test_negative_control_two_piece_split_breaks_the_budget(pkg):
    """the same measurement on libdril_hip_droplo.so (mfma_split3 without its two `lo` products = hi.hi only, an 11-bit product): the gradient criterion of the
    budget test must FAIL for every case — a test that a short product passes would prove nothing about 2^-24"""
    so = ROOT / "dril.jl_amd" / "csrc" / "libdril_hip_droplo.so"
    assert so.exists(), f"{so} missing: __graft_entry__.build() compiles it"
    env = dict(os.environ, DRIL_HIP_LIBRARY=str(so))
    env.pop("DRIL_GRAD_VARIANT", None)
    cases = [c for c in BUDGET_CASES if c[2] <= 8192]
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "split_budget.py"), json.dumps(cases)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ms = json.loads(r.stdout.strip().splitlines()[-1])
    for m in ms:
        print("[negative control]", json.dumps(m))
        ok_grad, _, _ = split_budget.within_budget(m)
        assert not ok_grad, m
        assert m["grad_err_split"] > 5.0 * m["grad_err_f32"]              # and not marginally

"""First-contact diagnostics on the GPU box: stage-by-stage parity prints (not a test)."""
import sys, time, traceback
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g
pkg = g.load_package()
import oracle_lib
capi = pkg._capi

def stage(name, fn):
    try:
        t = time.perf_counter(); fn(); print(f"[ok] {name} ({time.perf_counter()-t:.2f}s)", flush=True)
    except Exception:
        print(f"[FAIL] {name}", flush=True); traceback.print_exc(); sys.stdout.flush()

def d(a, b): return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())

for kind in (0, 1):
    cfg = capi.default_config(kind); cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.episode_len = 64, 16, 256, 2, 7
    h = pkg.Handle(cfg); o = oracle_lib.Oracle(cfg)
    flat = (np.random.default_rng(0).standard_normal(h.P) * 0.4).astype(np.float32)
    h.set_params(flat); o.set_params(flat)
    rng = np.random.default_rng(1)
    obs = rng.uniform(-2, 2, (100, h.D)).astype(np.float32)
    def fwd():
        v = h.predict_values(obs); vo = o.predict_values(obs); print("  predict_values maxdiff", d(v, vo), v[:3], vo[:3])
        noise = rng.random(100) if kind == 0 else rng.standard_normal((100, h.A)).astype(np.float32)
        a, v, lp = h.policy_forward(obs, noise); ao, vo, lo = o.policy_forward(obs, noise)
        print("  forward: values", d(v, vo), "logp", d(lp, lo), "actions equal", float((a == ao).mean()) if kind == 0 else d(a, ao))
        ve, le, ee = h.evaluate_actions(obs, ao); vo2, lo2, eo2 = o.evaluate_actions(obs, ao)
        print("  evaluate: values", d(ve, vo2), "logp", d(le, lo2), "entropy", d(ee, eo2))
    stage(f"kind{kind} forward", fwd)
    def lossgrad():
        B = 200
        ob = rng.uniform(-1, 1, (B, h.D)).astype(np.float32)
        act = (rng.integers(0, h.A, B) + cfg.action_start).astype(np.int32) if kind == 0 else rng.normal(0, 1, (B, h.A)).astype(np.float32)
        adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
        _, lp, _ = o.evaluate_actions(ob, act); olp = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
        lh, sh, gh = h.ppo_loss_grad(ob, act, adv, ret, olp, ov); lo, so, go = o.ppo_loss_grad(ob, act, adv, ret, olp, ov)
        print("  loss", lh, lo, "stats diff", d(sh, so)); print("  stats hip", sh); print("  stats orc", so)
        n = h.P
        A = h.A; D = h.D; H = 64
        offs = [0, H*D, H*D+H, H*D+H+H*H, H*D+2*H+H*H, H*D+2*H+H*H+A*H, H*D+2*H+H*H+A*H+A]
        names = ["aW1", "ab1", "aW2", "ab2", "aW3", "ab3"]
        for i, nm in enumerate(names):
            s = slice(offs[i], offs[i+1]); print(f"   {nm}: maxdiff {d(gh[s], go[s]):.3e}  ref max {np.abs(go[s]).max():.3e}")
        base = offs[-1]
        offs2 = [0, H*D, H*D+H, H*D+H+H*H, H*D+2*H+H*H, H*D+2*H+H*H+H, H*D+2*H+H*H+H+1]
        for i, nm in enumerate(["cW1", "cb1", "cW2", "cb2", "cW3", "cb3"]):
            s = slice(base+offs2[i], base+offs2[i+1]); print(f"   {nm}: maxdiff {d(gh[s], go[s]):.3e}  ref max {np.abs(go[s]).max():.3e}")
        if kind == 1: print("   log_std grad", gh[-1], go[-1])
        print("  rel grad err", np.linalg.norm(gh - go) / np.linalg.norm(go))
    stage(f"kind{kind} loss_grad", lossgrad)
    def roll():
        h.env_reset(3); o.env_reset(3)
        noise = rng.random(h.N) if kind == 0 else rng.standard_normal((h.N, h.A)).astype(np.float32)
        h.set_noise(noise); o.set_noise(noise)
        print("  fps", h.collect_rollout()); o.collect_rollout()
        for nm, w in (("obs", 0), ("act", 1), ("rew", 2), ("adv", 3), ("ret", 4), ("logp", 5), ("val", 6), ("flags", 7), ("boot", 8), ("lastv", 9)):
            print(f"   {nm}: maxdiff {d(h.buffer(w), o.buffer(w)):.3e}")
    stage(f"kind{kind} rollout", roll)
    def upd():
        perm = np.stack([np.random.default_rng(e).permutation(h.N) for e in range(cfg.epochs)]).astype(np.int64)
        for w in (0, 1, 3, 4, 5, 6): h.set_buffer(w, o.buffer(w))
        h.set_permutation(perm); o.set_permutation(perm)
        a, b = h.ppo_update(), o.ppo_update()
        for f in ("loss", "policy_loss", "value_loss", "approx_kl_div", "clip_fraction", "grad_norm", "explained_variance", "n_updates"):
            print(f"   {f}: {getattr(a, f)} vs {getattr(b, f)}")
        print("   params maxdiff", d(h.get_params(), o.get_params()))
    stage(f"kind{kind} update", upd)
    h.close()

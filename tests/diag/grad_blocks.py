"""diagnostic (GPU): per-parameter-block distance of dril_ppo_loss_grad from the oracle (which block of the gradient is off?)  usage: grad_blocks.py [kind] [B]"""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g
import oracle_lib
pkg = g.load_package()
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
c = pkg._capi.default_config(kind)
for k, v in dict(n_envs=2, n_steps=2, batch_size=2).items(): setattr(c, k, v)
h, o = pkg.Handle(c), oracle_lib.Oracle(c)
rng = np.random.default_rng(3)
flat = rng.uniform(-0.3, 0.3, o.P).astype(np.float32); h.set_params(flat); o.set_params(flat)
obs = rng.uniform(-1, 1, (B, o.D)).astype(np.float32)
act = (rng.integers(0, o.A, B) + c.action_start).astype(np.int32) if o.discrete else rng.normal(0, 1, (B, o.A)).astype(np.float32)
adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3))
_, lp, _ = o.evaluate_actions(obs, act); olp = (lp + rng.normal(0, 0.1, B)).astype(np.float32)
lh, sh, gh = h.ppo_loss_grad(obs, act, adv, ret, olp, ov)
lo, so, go = o.ppo_loss_grad(obs, act, adv, ret, olp, ov)
print(h.grad_kernel_info().split(":")[0], "loss", lh, lo, "total", np.linalg.norm(gh - go) / np.linalg.norm(go))
D, H, A = o.D, 64, o.A
off = 0
for net, O in (("actor", A), ("critic", 1)):
    for name, n in (("W1", D * H), ("b1", H), ("W2", H * H), ("b2", H), ("W3", O * H), ("b3", O)):
        a, b = gh[off:off + n], go[off:off + n]
        print(f"  {net} {name}: rel {np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30):.3e}  ratio {float(np.dot(a, b) / max(np.dot(b, b), 1e-30)):.6f}")
        off += n
if off < len(gh): print("  log_std rel", np.linalg.norm(gh[off:] - go[off:]) / np.linalg.norm(go[off:]))

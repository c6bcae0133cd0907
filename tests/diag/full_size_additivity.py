"""diagnostic (GPU): measured size of the differences that tests/test_gpu_split_arith.py::test_full_size_minibatch_is_the_weighted_mean_of_its_halves bounds"""
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg = g.load_package()
for kind, H in ((0, 64), (1, 64), (1, 256)):
    c = pkg._capi.default_config(kind)
    for k, v in dict(n_envs=2, n_steps=2, batch_size=2, hidden1=H, hidden2=H, normalize_advantage=0).items(): setattr(c, k, v)
    h = pkg.Handle(c); rng = np.random.default_rng(5)
    h.set_params(rng.uniform(-0.3, 0.3, h.P).astype(np.float32) if H == 64 else rng.uniform(-0.08, 0.08, h.P).astype(np.float32))
    B = 4194304
    obs = rng.uniform(-1, 1, (B, h.D)).astype(np.float32)
    act = (rng.integers(0, h.A, B) + c.action_start).astype(np.int32) if h.discrete else rng.normal(0, 1, (B, h.A)).astype(np.float32)
    adv, ret, ov = (rng.standard_normal(B).astype(np.float32) for _ in range(3)); olp = rng.normal(-0.7, 0.1, B).astype(np.float32)
    n1 = 1572864 + 37
    full = h.ppo_loss_grad(obs, act, adv, ret, olp, ov)
    parts = [h.ppo_loss_grad(obs[a:b], act[a:b], adv[a:b], ret[a:b], olp[a:b], ov[a:b]) for a, b in ((0, n1), (n1, B))]
    w = np.array([n1, B - n1], np.float64) / B
    loss = w[0] * parts[0][0] + w[1] * parts[1][0]; grad = w[0] * parts[0][2].astype(np.float64) + w[1] * parts[1][2].astype(np.float64)
    print(kind, H, h.grad_kernel_info().split(":")[0], "loss", full[0], "rel", abs(full[0] - loss) / abs(loss), "grad rel", np.linalg.norm(full[2] - grad) / np.linalg.norm(grad), "clipfrac", full[1][3])
    h.close()

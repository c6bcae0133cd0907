"""Diagnostic (GPU): one PPO update at a large minibatch on the [64,64] gradient-kernel variants (DRIL_GRAD_VARIANT 0 f32, 1 split, 2 two-waves-per-tile experiment):
parameters after the update and the learn statistics must agree to f32 rounding."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); pkg._capi.load_library()
res = {}
for env_name in ("cartpole", "pendulum"):
    for v in ("0", "1", "2"):
        os.environ["DRIL_GRAD_VARIANT"] = v
        env = pkg.CartPoleEnv(max_steps=500) if env_name == "cartpole" else pkg.PendulumEnv(max_steps=200)
        E, T = (int(os.environ.get("DIAG_E", "2048")), int(os.environ.get("DIAG_T", "256")))
        alg = pkg.PPO(n_steps=T, batch_size=E * T // int(os.environ.get("DIAG_MB", "2")), epochs=2)
        layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space())
        cfg = pkg.make_config(env, E, alg, layer, seed=3, fixed_length_episodes=True)
        h = pkg.Handle(cfg)
        h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(5))))
        h.env_reset(3); h.collect_rollout(); st = h.ppo_update()
        res[(env_name, v)] = (h.get_params().copy(), st.loss, st.grad_norm, h.grad_kernel_info().split(":")[0])
        h.close()
    p0 = res[(env_name, "0")][0]
    for v in ("1", "2"):
        p, loss, gn, k = res[(env_name, v)]
        print(env_name, v, k, "loss", loss, res[(env_name, "0")][1], "grad_norm", gn, res[(env_name, "0")][2], "max |dp|", float(np.abs(p - p0).max()), "rel", float(np.linalg.norm(p - p0) / np.linalg.norm(p0)))
        assert np.isfinite(p).all()

"""Diagnostic: what a launch of ppo_grad_pair_kernel costs besides its tiles.  Times the kernel (HIP events around every launch) on minibatches of 1 024 pairs x {1, 2, 4, 8, 16}
tiles per pair: the intercept of launch time against tiles per pair = staging + first record request + epilogue + launch floor."""
import sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
os.environ["DRIL_GRAD_VARIANT"] = "2"
import __graft_entry__ as g
pkg = g.load_package()
env = pkg.CartPoleEnv(max_steps=500)
rows = []
for tiles_per_pair in (1, 2, 4, 8, 16, 64):
    E = 65536; B = 1024 * 32 * tiles_per_pair; T = max(B // E, 1) * 1
    if B < E: E = B; T = 1
    alg = pkg.PPO(n_steps=T, batch_size=B, epochs=8)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(64, 64))
    cfg = pkg.make_config(env, E, alg, layer, seed=1, fixed_length_episodes=True, profile_events=True)
    h = pkg.Handle(cfg)
    h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(1))))
    h.env_reset(1); h.collect_rollout(); h.ppo_update(); h.profile_reset(); h.ppo_update()
    p = h.profile()
    k = p["ppo_grad_kernel"]
    rows.append((tiles_per_pair, (k["timed_ms"] / max(k["timed_launches"], 1)) * 1e3, k["launches"], h.grad_kernel_info().split(":")[0]))
    print(f"tiles per pair {tiles_per_pair:3d}: {rows[-1][1]:8.1f} us per launch  ({k['launches']} launches, {rows[-1][3]})", flush=True)
x = np.array([r[0] for r in rows[:5]], float); y = np.array([r[1] for r in rows[:5]], float)
b, a = np.polyfit(x, y, 1)
print(f"fit over 1..16 tiles: {a:.1f} us + {b:.2f} us per tile and pair")

"""diagnostic: host enqueue time vs drained time of SAC updates (DRIL_SAC_TRACE_ENQUEUE=1)"""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["DRIL_SAC_TRACE_ENQUEUE"] = "1"
import __graft_entry__ as g
pkg = g.load_package()
env = pkg.PendulumEnv(max_steps=200); alg = pkg.SAC(); layer = pkg.SACLayer(env.observation_space(), env.action_space())
h = pkg.SacHandle(pkg.make_sac_config(env, 4096, alg, layer, seed=1))
h.set_params(pkg.sac_flatten_params(layer.initialparameters(np.random.default_rng(0)))); h.env_reset(1)
h.collect_rollout(4, True)
for n in (1, 1, 1, 10, 10, 100):
    h.update(n)

"""Diagnostic: run one PPO update with the -DDRIL_STAMPS build of the library (per-phase s_memtime shares on stderr)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
capi = pkg._capi
import subprocess
import os
_so = ROOT / "dril.jl_amd" / "csrc" / os.environ.get("STAMPS_LIB", "libdril_hip_stamps.so")   # STAMPS_LIB=libdril_stamps_hi.so (tools/build_variant.sh, -DDRIL_STAMPS_HI): the younger wave of every SIMD
import os
if not os.environ.get("STAMPS_NO_MAKE"):      # (a library prebuilt in the container travels with the gpurun snapshot: STAMPS_NO_MAKE=1 skips the rebuild on the box)
    subprocess.run(["make", "-C", str(ROOT / "dril.jl_amd" / "csrc"), "-j8", "stamps"], check=True, stdout=sys.stderr)   # diagnostic build (-DDRIL_STAMPS); never used for timing claims
lib = capi.load_library(_so)
wide = len(sys.argv) > 1 and sys.argv[1] == "wide"       # config 3 shape: Pendulum, [256,256]
small = len(sys.argv) > 1 and sys.argv[1] == "small"     # configs[0] shape: 4 envs, PPO() defaults (the persistent small-minibatch kernel)
env = pkg.PendulumEnv(max_steps=200) if wide else pkg.CartPoleEnv(max_steps=500)
E, T = (4 if small else 65536), (256 if wide else 2048)
alg = pkg.PPO() if small else pkg.PPO(n_steps=T, batch_size=E * T // (4 if wide else 32), epochs=1)
import os
HW = int(os.environ.get("STAMPS_HIDDEN", "256"))
layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(HW, HW) if wide else (64, 64))
cfg = pkg.make_config(env, E, alg, layer, seed=42, fixed_length_episodes=True)
h = pkg.Handle(cfg, lib)
h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(42))))
h.env_reset(42); h.collect_rollout(); st = h.ppo_update(); print("loss", st.loss)

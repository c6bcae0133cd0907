"""one-off stress of gae_scan_kernel (not part of the suite): random shapes against the CPU oracle, and the configs[1] shape launched repeatedly on one handle —
every launch must give the same bits (the chunk-to-chunk carries travel through L2 with a per-launch tag; a stale or torn carry would show as a run-to-run difference)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g   # noqa: E402
import oracle_lib             # noqa: E402

pkg = g.load_package(); lib = pkg._capi.load_library(); orc = oracle_lib.lib()
p = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    E, T = int(rng.integers(1, 5000)), int(rng.integers(1, 600))
    gamma, lam = float(rng.choice([0.99, 0.9, 1.0])), float(rng.choice([0.95, 0.8, 1.0]))
    r = rng.standard_normal(E * T).astype(np.float32); v = rng.standard_normal(E * T).astype(np.float32)
    fl = rng.choice([0, 0, 0, 0, 0, 0, 1, 2, 3], E * T, p=None).astype(np.uint8) if rng.random() < 0.8 else np.zeros(E * T, np.uint8)
    if not fl.any(): r *= np.float32(0.01)
    b = rng.standard_normal(E * T).astype(np.float32); lv = rng.standard_normal(E).astype(np.float32)
    out = [np.full(E * T, np.nan, np.float32) for _ in range(4)]
    assert lib.dril_gae(E, T, gamma, lam, p(r), p(v), p(fl), p(b), p(lv), p(out[0]), p(out[1])) == 0
    assert orc.orc_gae(E, T, gamma, lam, p(r), p(v), p(fl), p(b), p(lv), p(out[2]), p(out[3])) == 0
    scale = max(1.0, float(np.abs(out[2]).max()))
    err = float(np.abs(out[0] - out[2]).max()) / scale
    worst = max(worst, err)
    assert np.isfinite(out[0]).all() and err < 5e-6, (E, T, gamma, lam, err)
print(f"random shapes ok, worst relative deviation from the serial oracle {worst:.2e}")
# repeated launches at the bench shape: bitwise stable
capi = pkg._capi
cfg = capi.default_config(capi.ENV_CARTPOLE)
cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.episode_len, cfg.fixed_length_episodes = 65536, 2048, 65536 * 2048 // 32, 500, 1
h = pkg.Handle(cfg)
layer = pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space())
h.set_params(pkg.flatten_params(layer.initialparameters(np.random.default_rng(0)))); h.env_reset(1); h.collect_rollout()
ref = h.buffer(capi.BUF_ADVANTAGES).copy()
import hashlib
hd = hashlib.sha256(ref.tobytes()).hexdigest()
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    h.compute_gae()
    assert hashlib.sha256(h.buffer(capi.BUF_ADVANTAGES).tobytes()).hexdigest() == hd, k
print("repeated launches at 65536 x 2048: bitwise stable", hd[:16])

"""ctypes wrapper of the CPU oracle (oracle/libdril_oracle.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as _g  # noqa: E402

_pkg = _g.load_package()
capi = _pkg._capi
DrilConfig, DrilPPOStats = capi.DrilConfig, capi.DrilPPOStats
SO = ROOT / "oracle" / "libdril_oracle.so"
_lib = None
_P = C.c_void_p


def host_threads() -> int:
    """threads the oracle may use: the CPU share of the box (cgroup quota / affinity), capped at 16 — a GPU box
    reports 256 logical CPUs but grants one GPU's job about 16 of them; ORACLE_THREADS overrides."""
    import os
    if os.environ.get("ORACLE_THREADS"):
        return max(1, int(os.environ["ORACLE_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, min(n, 16))


def ensure_built():
    srcs = [ROOT / "oracle" / "dril_oracle.c", ROOT / "oracle" / "dril_sac_oracle.c", ROOT / "include" / "dril_hip.h", ROOT / "include" / "dril_sac.h"]
    if not SO.exists() or SO.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, stdout=sys.stderr)   # never on stdout: bench.py prints ONE JSON line there


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        ensure_built()
        L = C.CDLL(str(SO))
        L.orc_create.argtypes = [C.POINTER(DrilConfig), C.POINTER(_P)]
        L.orc_param_count.restype = C.c_int64
        L.orc_param_count.argtypes = [_P]
        L.orc_perm_index.restype = C.c_int64
        L.orc_perm_index.argtypes = [C.c_int64, C.c_int64, C.c_uint64]
        L.orc_perm_key.restype = C.c_uint64
        L.orc_perm_key.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P]
        for n in ("orc_gauss_logpdf", "orc_gauss_entropy", "orc_categorical_logpdf", "orc_categorical_entropy"):
            getattr(L, n).restype = C.c_float
        L.orc_gauss_logpdf.argtypes = [_P, _P, _P, C.c_int]
        L.orc_gauss_entropy.argtypes = [_P, C.c_int]
        L.orc_categorical_logpdf.argtypes = [_P, C.c_int, C.c_int, C.c_int]
        L.orc_categorical_entropy.argtypes = [_P, C.c_int]
        L.orc_categorical_sample.argtypes = [_P, C.c_int, C.c_double, C.c_int]
        L.orc_compute_advantages.restype = None
        L.orc_compute_advantages.argtypes = [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
        L.orc_gae.argtypes = [C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _P]
        L.orc_rms_update.restype = None
        L.orc_rms_update.argtypes = [_P, _P, C.POINTER(C.c_int64), C.c_int, _P, C.c_int64]
        L.orc_set_params.argtypes = [_P, _P, C.c_size_t]
        L.orc_get_params.argtypes = [_P, _P, C.c_size_t]
        L.orc_set_learning_rate.argtypes = [_P, C.c_float]
        L.orc_env_reset.argtypes = [_P, C.c_uint64]
        L.orc_env_observe.argtypes = [_P, _P, C.c_int32]
        L.orc_env_step.argtypes = [_P, _P, _P, _P, _P, _P]
        L.orc_env_get_state.argtypes = [_P, _P, _P]
        L.orc_env_set_state.argtypes = [_P, _P, _P]
        L.orc_policy_forward.argtypes = [_P, _P, C.c_int64, _P, _P, _P, _P]
        L.orc_evaluate_actions.argtypes = [_P, _P, _P, C.c_int64, _P, _P, _P]
        L.orc_predict_values.argtypes = [_P, _P, C.c_int64, _P]
        L.orc_predict_actions.argtypes = [_P, _P, C.c_int64, C.c_int32, _P, _P]
        L.orc_collect_rollout.argtypes = [_P, C.POINTER(C.c_double)]
        L.orc_debug_set_noise.argtypes = [_P, _P, C.c_size_t]
        L.orc_buffer_copy_out.argtypes = [_P, C.c_int32, _P, C.c_size_t]
        L.orc_buffer_copy_in.argtypes = [_P, C.c_int32, _P, C.c_size_t]
        L.orc_ref_order.argtypes = [_P, _P]
        L.orc_compute_gae.argtypes = [_P]
        L.orc_ppo_update.argtypes = [_P, C.POINTER(DrilPPOStats)]
        L.orc_debug_set_permutation.argtypes = [_P, _P, C.c_size_t]
        L.orc_ppo_loss_grad.argtypes = [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_float), _P, _P]
        L.orc_apply_gradients.argtypes = [_P, _P, C.c_size_t, C.POINTER(C.c_float)]
        L.orc_train.argtypes = [_P, C.c_int64, _P, _P, C.POINTER(C.c_int32)]
        L.orc_norm_get_stats.argtypes = [_P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        L.orc_monitor_get_stats.argtypes = [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.orc_evaluate_agent.argtypes = [_P, C.c_int32, C.c_int32, C.POINTER(capi.DrilEvalStats), _P, _P]
        L.orc_ext_act.argtypes = [_P, _P, _P, _P, _P]
        L.orc_ext_record.argtypes = [_P, _P, _P, _P, _P]
        L.orc_ext_finish.argtypes = [_P, _P]
        L.orc_ext_steps.argtypes = [_P]
        L.orc_destroy.argtypes = [_P]
        L.orc_reset_optimizer.argtypes = [_P]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(host_threads())
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Same surface as dril_jl_amd.Handle so parity tests read symmetrically."""

    def __init__(self, cfg: DrilConfig):
        self.L = lib()
        self.cfg = cfg
        self._h = _P()
        rc = self.L.orc_create(C.byref(cfg), C.byref(self._h))
        assert rc == 0, rc
        self.discrete = cfg.env_kind in (capi.ENV_CARTPOLE, capi.ENV_MOUNTAINCAR, capi.ENV_ACROBOT)
        self.D, self.A, self.S = {capi.ENV_CARTPOLE: (4, 2, 4), capi.ENV_MOUNTAINCAR: (2, 3, 2), capi.ENV_MOUNTAINCAR_CONTINUOUS: (2, 1, 2), capi.ENV_MOUNTAINCAR_CONTINUOUS_SCALED: (2, 1, 2), capi.ENV_ACROBOT: (6, 3, 4)}.get(cfg.env_kind, (3, 1, 2))
        if cfg.env_kind == capi.ENV_EXTERNAL:
            self.discrete, self.D, self.A, self.S = bool(cfg.ext_discrete), cfg.ext_obs_dim, cfg.ext_action_dim, 0
        self.P = int(self.L.orc_param_count(self._h))
        self.E, self.T = cfg.n_envs, cfg.n_steps
        self.N = self.E * self.T
        self._keep = []

    def __del__(self):
        try:
            self.L.orc_destroy(self._h)
        except Exception:
            pass

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, np.float32)
        assert self.L.orc_set_params(self._h, _p(flat), flat.size) == 0

    def get_params(self):
        out = np.empty(self.P, np.float32)
        assert self.L.orc_get_params(self._h, _p(out), out.size) == 0
        return out

    def reset_optimizer(self):
        self.L.orc_reset_optimizer(self._h)

    def set_learning_rate(self, lr):
        self.L.orc_set_learning_rate(self._h, lr)

    def env_reset(self, seed):
        self.L.orc_env_reset(self._h, seed)

    def env_observe(self, update_stats=True):
        obs = np.empty((self.E, self.D), np.float32)
        self.L.orc_env_observe(self._h, _p(obs), int(update_stats))
        return obs

    def env_step(self, actions):
        actions = np.ascontiguousarray(actions, np.int32 if self.discrete else np.float32)
        rew = np.empty(self.E, np.float32); term = np.empty(self.E, np.uint8); trunc = np.empty(self.E, np.uint8)
        tobs = np.zeros((self.E, self.D), np.float32)
        self.L.orc_env_step(self._h, _p(actions), _p(rew), _p(term), _p(trunc), _p(tobs))
        return rew, term.astype(bool), trunc.astype(bool), tobs

    def env_get_state(self):
        st = np.empty((self.E, self.S), np.float32); sc = np.empty(self.E, np.int32)
        self.L.orc_env_get_state(self._h, _p(st), _p(sc))
        return st, sc

    def env_set_state(self, st, sc=None):
        st = np.ascontiguousarray(st, np.float32)
        sc = None if sc is None else np.ascontiguousarray(sc, np.int32)
        self.L.orc_env_set_state(self._h, _p(st), _p(sc))

    def policy_forward(self, obs, noise):
        obs = np.ascontiguousarray(obs, np.float32); B = obs.shape[0]
        noise = np.ascontiguousarray(noise, np.float64 if self.discrete else np.float32)
        act = np.empty(B, np.int32) if self.discrete else np.empty((B, self.A), np.float32)
        val = np.empty(B, np.float32); lp = np.empty(B, np.float32)
        self.L.orc_policy_forward(self._h, _p(obs), B, _p(noise), _p(act), _p(val), _p(lp))
        return act, val, lp

    def evaluate_actions(self, obs, actions):
        obs = np.ascontiguousarray(obs, np.float32); B = obs.shape[0]
        actions = np.ascontiguousarray(actions, np.int32 if self.discrete else np.float32)
        val, lp, ent = (np.empty(B, np.float32) for _ in range(3))
        self.L.orc_evaluate_actions(self._h, _p(obs), _p(actions), B, _p(val), _p(lp), _p(ent))
        return val, lp, ent

    def predict_actions(self, obs, deterministic=False, noise=None):
        obs = np.ascontiguousarray(obs, np.float32)
        B = obs.shape[0]
        act = np.empty(B, np.int32) if self.discrete else np.empty((B, self.A), np.float32)
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float64 if self.discrete else np.float32)
        assert self.L.orc_predict_actions(self._h, _p(obs), B, int(deterministic), _p(noise), _p(act)) == 0
        return act

    def predict_values(self, obs):
        obs = np.ascontiguousarray(obs, np.float32)
        val = np.empty(obs.shape[0], np.float32)
        self.L.orc_predict_values(self._h, _p(obs), obs.shape[0], _p(val))
        return val

    def set_noise(self, noise):
        if noise is None:
            self.L.orc_debug_set_noise(self._h, None, 0); return
        noise = np.ascontiguousarray(noise, np.float64 if self.discrete else np.float32)
        self._keep.append(noise)  # the oracle keeps the pointer until the next rollout
        self.L.orc_debug_set_noise(self._h, _p(noise), noise.size)

    # rollout over host envs (DRIL_ENV_EXTERNAL); `noise` is the step's sampling noise (f64[E] | f32[E, A])
    def ext_act(self, obs, noise):
        obs = np.ascontiguousarray(obs, np.float32).reshape(self.E, self.D)
        noise = np.ascontiguousarray(noise, np.float64 if self.discrete else np.float32)
        raw = np.empty(self.E, np.int32) if self.discrete else np.empty((self.E, self.A), np.float32)
        env_a = np.empty_like(raw)
        assert self.L.orc_ext_act(self._h, _p(obs), _p(noise), _p(raw), _p(env_a)) == 0
        return raw, env_a

    def ext_record(self, rewards, terminated, truncated, terminal_obs=None):
        r = np.ascontiguousarray(rewards, np.float32); te = np.ascontiguousarray(terminated, np.uint8); tr = np.ascontiguousarray(truncated, np.uint8)
        to = None if terminal_obs is None else np.ascontiguousarray(terminal_obs, np.float32).reshape(self.E, self.D)
        assert self.L.orc_ext_record(self._h, _p(r), _p(te), _p(tr), _p(to)) == 0

    def ext_finish(self, last_obs):
        o = np.ascontiguousarray(last_obs, np.float32).reshape(self.E, self.D)
        assert self.L.orc_ext_finish(self._h, _p(o)) == 0

    def collect_rollout(self):
        fps = C.c_double()
        assert self.L.orc_collect_rollout(self._h, C.byref(fps)) == 0
        return fps.value

    def _buf_like(self, which):
        if which == capi.BUF_ACTIONS:
            return np.empty(self.N, np.int32) if self.discrete else np.empty((self.N, self.A), np.float32)
        if which == capi.BUF_OBSERVATIONS:
            return np.empty((self.N, self.D), np.float32)
        if which == capi.BUF_FLAGS:
            return np.empty(self.N, np.uint8)
        if which == capi.BUF_LAST_VALUES:
            return np.empty(self.E, np.float32)
        return np.empty(self.N, np.float32)

    def buffer(self, which):
        out = self._buf_like(which)
        assert self.L.orc_buffer_copy_out(self._h, which, _p(out), out.nbytes) == 0
        return out

    def set_buffer(self, which, arr):
        like = self._buf_like(which)
        arr = np.ascontiguousarray(arr, like.dtype).reshape(like.shape)
        assert self.L.orc_buffer_copy_in(self._h, which, _p(arr), arr.nbytes) == 0

    def ref_order(self):
        out = np.empty(self.N, np.int64)
        self.L.orc_ref_order(self._h, _p(out))
        return out

    def compute_gae(self):
        self.L.orc_compute_gae(self._h)

    def set_permutation(self, perm):
        if perm is None:
            self.L.orc_debug_set_permutation(self._h, None, 0); return
        perm = np.ascontiguousarray(perm, np.int64)
        self._keep.append(perm)
        self.L.orc_debug_set_permutation(self._h, _p(perm), perm.size)

    def ppo_update(self):
        st = DrilPPOStats()
        self.last_rc = self.L.orc_ppo_update(self._h, C.byref(st))
        return st

    def ppo_loss_grad(self, obs, actions, adv, ret, old_logp, old_val):
        obs = np.ascontiguousarray(obs, np.float32)
        actions = np.ascontiguousarray(actions, np.int32 if self.discrete else np.float32)
        adv = np.array(adv, np.float32, copy=True)  # normalised in place by the oracle, like the reference
        ret, old_logp, old_val = (np.ascontiguousarray(x, np.float32) for x in (ret, old_logp, old_val))
        loss = C.c_float(); stats = np.empty(7, np.float32); grads = np.empty(self.P, np.float32)
        self.L.orc_ppo_loss_grad(self._h, _p(obs), _p(actions), _p(adv), _p(ret), _p(old_logp), _p(old_val), obs.shape[0],
                                 C.byref(loss), _p(stats), _p(grads))
        return loss.value, stats, grads

    def apply_gradients(self, grads):
        grads = np.array(grads, np.float32, copy=True)
        norm = C.c_float()
        self.last_rc = self.L.orc_apply_gradients(self._h, _p(grads), grads.size, C.byref(norm))
        return norm.value

    def train(self, max_steps):
        iters = max_steps // (self.N * max(1, self.cfg.world_size))
        stats = (DrilPPOStats * max(iters, 1))(); fps = (C.c_double * max(iters, 1))(); done = C.c_int32()
        self.L.orc_train(self._h, max_steps, stats, fps, C.byref(done))
        return [stats[i] for i in range(done.value)], [fps[i] for i in range(done.value)]

    def evaluate_agent(self, n_eval_episodes=10, deterministic=True):
        st = capi.DrilEvalStats()
        er = np.empty(n_eval_episodes, np.float32); el = np.empty(n_eval_episodes, np.int32)
        assert self.L.orc_evaluate_agent(self._h, n_eval_episodes, int(deterministic), C.byref(st), _p(er), _p(el)) == 0
        return dict(mean_reward=st.mean_reward, std_reward=st.std_reward, mean_length=st.mean_length, std_length=st.std_length, n_steps=st.n_steps), er, el

    def monitor_stats(self):
        r, l, n = C.c_float(), C.c_float(), C.c_int32()
        self.L.orc_monitor_get_stats(self._h, C.byref(r), C.byref(l), C.byref(n))
        return r.value, l.value, n.value

    def norm_stats(self):
        om = np.empty(self.D, np.float32); ov = np.empty(self.D, np.float32)
        oc, rc = C.c_int64(), C.c_int64(); rm, rv = C.c_float(), C.c_float()
        self.L.orc_norm_get_stats(self._h, _p(om), _p(ov), C.byref(oc), C.byref(rm), C.byref(rv), C.byref(rc))
        return om, ov, oc.value, rm.value, rv.value, rc.value


def sac_oracle(cfg):
    """the SAC oracle behind the product's own typed wrapper (dril.jl_amd/sac.py): same signatures, prefix orc_sac_"""
    L = lib()
    L.orc_squashed_logpdf.restype = C.c_float
    L.orc_squashed_logpdf.argtypes = [_P, _P, _P, C.c_int]
    L.orc_polyak_update.restype = None
    L.orc_polyak_update.argtypes = [_P, _P, C.c_size_t, C.c_float]
    L.orc_sac_schedule.restype = None
    L.orc_sac_schedule.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [C.POINTER(C.c_int64)] * 4
    for name, (res, args) in _pkg._capi._SAC_SIG.items():
        fn = getattr(L, "orc_sac_" + name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args

    class OracleSacHandle(_pkg.SacHandle):
        """the product's typed wrapper pointed at the CPU oracle's symbols — test infrastructure, which is why the switch lives here and not in the package"""
        _PREFIX = "orc_sac_"

        @classmethod
        def _load(cls):
            return L
    return OracleSacHandle(cfg)

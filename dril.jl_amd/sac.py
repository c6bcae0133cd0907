"""Host-side mirror of the reference's SAC interface (src/algorithms/sac.jl) over the C ABI of include/dril_sac.h.

    SAC, AutoEntropyCoefficient, FixedEntropyCoefficient        sac.jl:25-36, src/interfaces/entropy.jl
    SACLayer(obs_space, act_space; hidden_dims=[512,512], ...)  sac.jl:72-85  (ContinuousActorCriticLayer{QCritic})
    SACAgent(layer, alg)                                        sac.jl:160-188
    ReplayBuffer(obs_space, act_space, capacity)                src/buffers/replay_buffer.jl
    sac_train_(agent, env, alg, max_steps)                      sac.jl:406-549  ->  (agent, replay_buffer, training_stats, timer)

`SacHandle` types one dril_sac_handle* of libdril_hip.so — there is no fallback and no way to point it elsewhere; the parity tests drive the
CPU oracle ("orc_sac_" symbols, same signatures) through a subclass that lives in tests/oracle_lib.py.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _capi as capi
from ._capi import DrilSacConfig, DrilSacStats
from .host import Box, DrilError, PendulumEnv, ScalingWrapperEnv, _orthogonal


# --------------------------------------------------------------------------------------------
# algorithm + layer (host-side data only)
# --------------------------------------------------------------------------------------------
@dataclass
class AutoEntropyCoefficient:
    """entropy.jl:21-24; target None = AutoEntropyTarget (-dim(action_space), sac.jl:47-57)"""
    target: Optional[float] = None
    initial_value: float = 1.0


@dataclass
class FixedEntropyCoefficient:
    coef: float = 0.2


@dataclass
class SAC:
    """SAC(; ...) sac.jl:25-36"""
    learning_rate: float = 3e-4
    buffer_capacity: int = 1_000_000
    start_steps: int = 100
    batch_size: int = 256
    tau: float = 0.005
    gamma: float = 0.99
    train_freq: int = 1
    gradient_steps: int = 1
    ent_coef: object = field(default_factory=AutoEntropyCoefficient)
    target_update_interval: int = 1


def get_gradient_steps(alg: SAC, train_freq: Optional[int] = None, n_envs: int = 1) -> int:
    """sac.jl:59-65"""
    tf = alg.train_freq if train_freq is None else train_freq
    return tf * n_envs if alg.gradient_steps == -1 else alg.gradient_steps


@dataclass
class SACLayer:
    """SACLayer(observation_space, action_space; log_std_init=-3, hidden_dims=[512,512], activation=relu) sac.jl:72-85:
    ContinuousActorCriticLayer with critic_type = QCritic() (n_critics = 2) and separate features."""
    observation_space: Box
    action_space: Box
    hidden_dims: Sequence[int] = (512, 512)
    log_std_init: float = -3.0
    activation: str = "relu"

    @property
    def obs_dim(self) -> int:
        return len(self.observation_space.low)

    @property
    def act_dim(self) -> int:
        return len(self.action_space.low)

    def q_parameterlength(self) -> int:
        d, a, (h1, h2) = self.obs_dim, self.act_dim, self.hidden_dims
        return (d + a) * h1 + h1 + h1 * h2 + h2 + h2 + 1

    def parameterlength(self) -> int:
        d, a, (h1, h2) = self.obs_dim, self.act_dim, self.hidden_dims
        return d * h1 + h1 + h1 * h2 + h2 + h2 * a + a + 2 * self.q_parameterlength() + a

    def initialparameters(self, rng: np.random.Generator) -> dict:
        """orthogonal gains sqrt(2) hidden / 0.01 actor output / 1.0 Q output, zero bias (layer_constructors.jl:16-20);
        critic_head = Lux.Parallel(vcat, mlp, mlp) -> layer_1 / layer_2 (layer_helpers.jl:100-112)"""
        d, a, (h1, h2) = self.obs_dim, self.act_dim, self.hidden_dims

        def mlp(inp, out, gain):
            return {"layer_1": {"weight": _orthogonal(rng, h1, inp, math.sqrt(2.0)), "bias": np.zeros(h1, np.float32)},
                    "layer_2": {"weight": _orthogonal(rng, h2, h1, math.sqrt(2.0)), "bias": np.zeros(h2, np.float32)},
                    "layer_3": {"weight": _orthogonal(rng, out, h2, gain), "bias": np.zeros(out, np.float32)}}

        return {"actor_head": mlp(d, a, 0.01), "critic_head": {"layer_1": mlp(d + a, 1, 1.0), "layer_2": mlp(d + a, 1, 1.0)},
                "log_std": np.full(a, self.log_std_init, np.float32)}


def _mlp_flat(m: dict):
    for l in ("layer_1", "layer_2", "layer_3"):
        yield np.asarray(m[l]["weight"], np.float32).ravel(order="F")
        yield np.asarray(m[l]["bias"], np.float32).ravel()


def sac_flatten_params(ps: dict) -> np.ndarray:
    """Lux NamedTuple -> the flat layout of dril_sac_set_params"""
    parts = list(_mlp_flat(ps["actor_head"])) + list(_mlp_flat(ps["critic_head"]["layer_1"])) + list(_mlp_flat(ps["critic_head"]["layer_2"]))
    parts.append(np.asarray(ps["log_std"], np.float32).ravel())
    return np.concatenate(parts)


def sac_unflatten_params(flat: np.ndarray, like: dict) -> dict:
    off = 0

    def mlp(m):
        nonlocal off
        out = {}
        for l in ("layer_1", "layer_2", "layer_3"):
            w, b = m[l]["weight"], m[l]["bias"]
            out[l] = {"weight": flat[off:off + w.size].reshape(w.shape, order="F").copy()}
            off += w.size
            out[l]["bias"] = flat[off:off + b.size].copy()
            off += b.size
        return out

    ps = {"actor_head": mlp(like["actor_head"])}
    ps["critic_head"] = {"layer_1": mlp(like["critic_head"]["layer_1"]), "layer_2": mlp(like["critic_head"]["layer_2"])}
    ps["log_std"] = flat[off:off + like["log_std"].size].copy()
    return ps


def make_sac_config(env, n_envs: int, alg: SAC, layer: SACLayer, *, seed: int = 42, device: int = 0,
                    profile_events: bool = False) -> DrilSacConfig:
    external = getattr(env, "kind", None) == capi.ENV_EXTERNAL
    if not external and getattr(env, "kind", None) not in (capi.ENV_PENDULUM, capi.ENV_PENDULUM_SCALED, capi.ENV_MOUNTAINCAR_CONTINUOUS, capi.ENV_MOUNTAINCAR_CONTINUOUS_SCALED):
        raise NotImplementedError("SAC needs a Box action space (sac.jl:74); the device envs with one are Pendulum-v1 (optionally under ScalingWrapperEnv) and MountainCarContinuous-v0")
    c = DrilSacConfig()
    c.abi_version = capi.SAC_ABI_VERSION
    c.env_kind, c.n_envs, c.episode_len = env.kind, n_envs, getattr(env, "max_steps", 0)
    if external:     # host envs: the spaces travel in the config (include/dril_sac.h)
        osp, asp = env.observation_space(), env.action_space()
        lo, hi = np.unique(np.asarray(asp.low, np.float32)), np.unique(np.asarray(asp.high, np.float32))
        if lo.size != 1 or hi.size != 1:
            raise NotImplementedError("DRIL_ENV_EXTERNAL SAC: one (low, high) pair for all action dimensions (wrap the env in a ScalingWrapperEnv-style Box(-1, 1))")
        c.ext_obs_dim, c.ext_action_dim, c.ext_action_low, c.ext_action_high = len(osp.low), len(asp.low), float(lo[0]), float(hi[0])
    c.hidden1, c.hidden2 = layer.hidden_dims
    c.activation = {"tanh": 0, "relu": 1}[layer.activation]
    c.buffer_capacity, c.start_steps, c.batch_size = alg.buffer_capacity, alg.start_steps, alg.batch_size
    c.tau, c.gamma = alg.tau, alg.gamma
    c.train_freq, c.gradient_steps, c.target_update_interval = alg.train_freq, alg.gradient_steps, alg.target_update_interval
    if isinstance(alg.ent_coef, AutoEntropyCoefficient):
        c.auto_ent_coef, c.ent_coef_init = 1, alg.ent_coef.initial_value
        c.auto_target_entropy = int(alg.ent_coef.target is None)
        c.target_entropy = 0.0 if alg.ent_coef.target is None else alg.ent_coef.target
    else:
        c.auto_ent_coef, c.ent_coef_init, c.auto_target_entropy, c.target_entropy = 0, alg.ent_coef.coef, 1, 0.0
    c.learning_rate, c.adam_beta1, c.adam_beta2, c.adam_eps = alg.learning_rate, 0.9, 0.999, 1e-8
    c.seed, c.device, c.profile_events = seed, device, int(profile_events)
    return c


# --------------------------------------------------------------------------------------------
# typed wrapper of one dril_sac_handle*
# --------------------------------------------------------------------------------------------
class SacHandle:
    """typed wrapper of one dril_sac_handle* of libdril_hip.so (there is no other backend: the CPU oracle is driven by a subclass that lives under tests/)"""
    _PREFIX = "dril_sac_"

    @classmethod
    def _load(cls) -> C.CDLL:
        return capi.load_library()

    def __init__(self, cfg: DrilSacConfig):
        self.lib = self._load()
        self.prefix = self._PREFIX
        prefix = self.prefix
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self._f("create")(C.byref(cfg), C.byref(self._h))
        if rc != capi.OK:
            le = getattr(self.lib, prefix + "last_error", None)
            raise DrilError(rc, (le(None) or b"").decode() if le else "create failed")
        self.D, self.A = self._f("obs_dim")(self._h), self._f("action_dim")(self._h)
        self.P, self.Pq = int(self._f("param_count")(self._h)), int(self._f("q_param_count")(self._h))
        self.E, self.B = cfg.n_envs, cfg.batch_size

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def close(self):
        if self._h:
            self._f("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != capi.OK:
            le = getattr(self.lib, self.prefix + "last_error", None)
            raise DrilError(rc, (le(self._h) or b"").decode() if le else f"status {rc}")

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _f32(a):
        return None if a is None else np.ascontiguousarray(a, np.float32)

    # parameters
    def set_params(self, flat):
        flat = self._f32(flat)
        self._chk(self._f("set_params")(self._h, self._p(flat), flat.size))

    def get_params(self) -> np.ndarray:
        out = np.empty(self.P, np.float32)
        self._chk(self._f("get_params")(self._h, self._p(out), out.size))
        return out

    def get_target_params(self) -> np.ndarray:
        out = np.empty(2 * self.Pq, np.float32)
        self._chk(self._f("get_target_params")(self._h, self._p(out), out.size))
        return out

    def set_target_params(self, flat):
        flat = self._f32(flat)
        self._chk(self._f("set_target_params")(self._h, self._p(flat), flat.size))

    def get_log_ent_coef(self) -> float:
        v = C.c_float()
        self._chk(self._f("get_log_ent_coef")(self._h, C.byref(v)))
        return v.value

    def set_log_ent_coef(self, v: float):
        self._chk(self._f("set_log_ent_coef")(self._h, v))

    def reset_optimizer(self):
        self._chk(self._f("reset_optimizer")(self._h))

    # env
    def env_reset(self, seed: int):
        self._chk(self._f("env_reset")(self._h, seed))

    def env_observe(self) -> np.ndarray:
        out = np.empty((self.E, self.D), np.float32)
        self._chk(self._f("env_observe")(self._h, self._p(out)))
        return out

    # layer calls; arrays are (batch, features) row-major == (features x batch) column-major
    def action_log_prob(self, obs, noise=None):
        obs, noise = self._f32(obs), self._f32(noise)
        B = obs.shape[0]
        a, lp = np.empty((B, self.A), np.float32), np.empty(B, np.float32)
        self._chk(self._f("action_log_prob")(self._h, self._p(obs), B, self._p(noise), self._p(a), self._p(lp)))
        return a, lp

    def predict_actions(self, obs, deterministic=False, noise=None):
        obs, noise = self._f32(obs), self._f32(noise)
        B = obs.shape[0]
        raw, env = np.empty((B, self.A), np.float32), np.empty((B, self.A), np.float32)
        self._chk(self._f("predict_actions")(self._h, self._p(obs), B, int(deterministic), self._p(noise), self._p(raw), self._p(env)))
        return raw, env

    def predict_q(self, obs, actions, use_target=False):
        obs, actions = self._f32(obs), self._f32(actions)
        B = obs.shape[0]
        q = np.empty((B, 2), np.float32)
        self._chk(self._f("predict_q")(self._h, self._p(obs), self._p(actions), B, int(use_target), self._p(q)))
        return q

    # collection + replay
    def collect_rollout(self, n_steps: int, use_random_actions: bool = False) -> float:
        fps = C.c_double()
        self._chk(self._f("collect_rollout")(self._h, n_steps, int(use_random_actions), C.byref(fps)))
        return fps.value

    def ext_push(self, obs, stored_actions, rewards, terminated, truncated, next_obs, terminal_obs=None):
        """one env step of the caller's host envs into the replay ring (DRIL_ENV_EXTERNAL)"""
        o, a, r, n = self._f32(obs), self._f32(stored_actions), self._f32(rewards), self._f32(next_obs)
        te, tr = np.ascontiguousarray(terminated, np.uint8), np.ascontiguousarray(truncated, np.uint8)
        to = None if terminal_obs is None else self._f32(terminal_obs)
        self._chk(self._f("ext_push")(self._h, self._p(o), self._p(a), self._p(r), self._p(te), self._p(tr), self._p(n), self._p(to)))

    def set_collect_noise(self, noise):
        self._noise = self._f32(noise)       # the oracle keeps the pointer until the next collect call
        self._chk(self._f("debug_set_collect_noise")(self._h, self._p(self._noise), 0 if noise is None else self._noise.size))

    def replay_size(self) -> int:
        return int(self._f("replay_size")(self._h))

    def replay_capacity(self) -> int:
        return int(self._f("replay_capacity")(self._h))

    def replay(self, which: int) -> np.ndarray:
        n = self.replay_size()
        shape, dt = {capi.RB_OBSERVATIONS: ((n, self.D), np.float32), capi.RB_NEXT_OBSERVATIONS: ((n, self.D), np.float32),
                     capi.RB_ACTIONS: ((n, self.A), np.float32), capi.RB_REWARDS: ((n,), np.float32),
                     capi.RB_TERMINATED: ((n,), np.uint8), capi.RB_TRUNCATED: ((n,), np.uint8)}[which]
        out = np.empty(shape, dt)
        self._chk(self._f("replay_copy_out")(self._h, which, self._p(out), out.nbytes))
        return out

    def replay_fill(self, obs, actions, rewards, terminated, truncated, next_obs):
        obs, actions, rewards, next_obs = self._f32(obs), self._f32(actions), self._f32(rewards), self._f32(next_obs)
        term = np.ascontiguousarray(terminated, np.uint8)
        trunc = np.ascontiguousarray(truncated, np.uint8)
        self._chk(self._f("replay_fill")(self._h, rewards.size, self._p(obs), self._p(actions), self._p(rewards), self._p(term), self._p(trunc), self._p(next_obs)))

    # gradient steps
    def set_batches(self, n_updates: int, idx=None, noise_ent=None, noise_next=None, noise_pi=None):
        self._inj = (None if idx is None else np.ascontiguousarray(idx, np.int64), self._f32(noise_ent), self._f32(noise_next), self._f32(noise_pi))
        self._chk(self._f("debug_set_batches")(self._h, n_updates, *[self._p(a) for a in self._inj]))

    def update(self, n_updates: int = 1):
        out = (DrilSacStats * n_updates)()
        self._chk(self._f("update")(self._h, n_updates, C.cast(out, C.c_void_p)))
        return list(out)

    def last_grads(self):
        gc, ga = np.empty(self.P, np.float32), np.empty(self.P, np.float32)
        self._chk(self._f("get_last_grads")(self._h, self._p(gc), self._p(ga), self.P))
        return gc, ga

    def train(self, max_steps: int, stats_capacity: int = 1 << 16, fps_capacity: int = 1 << 16):
        stats = (DrilSacStats * stats_capacity)()
        fps = np.zeros(fps_capacity, np.float64)
        n_upd, iters, total = C.c_int64(), C.c_int32(), C.c_int64()
        self._chk(self._f("train")(self._h, max_steps, C.cast(stats, C.c_void_p), stats_capacity, C.byref(n_upd),
                                   fps.ctypes.data_as(C.POINTER(C.c_double)), fps_capacity, C.byref(iters), C.byref(total)))
        return list(stats[:min(n_upd.value, stats_capacity)]), fps[:min(iters.value, fps_capacity)], n_upd.value, iters.value, total.value

    def iterate(self, iterations: int, want_stats: bool = True):
        """`iterations` rounds of train!'s loop body (sac.jl:464-535: collect train_freq env steps, then the gradient steps) without a host sync between them
        (dril_sac_iterate) -> (stats of every gradient step, fps of every iteration)"""
        n_upd = self.cfg.train_freq * self.E if self.cfg.gradient_steps == -1 else self.cfg.gradient_steps
        cap = iterations * max(n_upd, 0) if want_stats else 0
        stats = (DrilSacStats * max(cap, 1))()
        fps = np.zeros(iterations if want_stats else 1, np.float64)
        self._chk(self._f("iterate")(self._h, C.c_int32(iterations), C.cast(stats, C.c_void_p) if want_stats else None, C.c_int64(cap),
                                     fps.ctypes.data_as(C.POINTER(C.c_double)) if want_stats else None, C.c_int64(iterations if want_stats else 0)))
        return list(stats[:cap]), fps[:iterations] if want_stats else fps[:0]

    def profile(self) -> dict:
        cm, um, cs, us = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
        self._chk(self._f("profile_get")(self._h, C.byref(cm), C.byref(cs), C.byref(um), C.byref(us)))
        return {"collect_ms": cm.value, "collect_steps": cs.value, "update_ms": um.value, "updates": us.value}

    def profile_reset(self):
        self._chk(self._f("profile_reset")(self._h))


# --------------------------------------------------------------------------------------------
# Agent / ReplayBuffer / train!
# --------------------------------------------------------------------------------------------
@dataclass
class SACAgent:
    """Agent(layer, alg::SAC; rng) sac.jl:160-188: train_state, Q_target_parameters (copy of the critics), ent_train_state"""
    layer: SACLayer
    alg: SAC
    seed: int = 0
    parameters: dict = field(init=False)
    q_target_parameters: np.ndarray = field(init=False)
    log_ent_coef: float = field(init=False)
    steps_taken: int = 0
    gradient_updates: int = 0

    def __post_init__(self):
        rng = np.random.default_rng(self.seed)
        self.parameters = self.layer.initialparameters(rng)
        flat = sac_flatten_params(self.parameters)
        a = self.layer
        n_actor = a.obs_dim * a.hidden_dims[0] + a.hidden_dims[0] + a.hidden_dims[0] * a.hidden_dims[1] + a.hidden_dims[1] + a.hidden_dims[1] * a.act_dim + a.act_dim
        self.q_target_parameters = flat[n_actor:n_actor + 2 * a.q_parameterlength()].copy()           # copy_critic_parameters sac.jl:191-197
        ec = self.alg.ent_coef
        self.log_ent_coef = math.log(ec.initial_value if isinstance(ec, AutoEntropyCoefficient) else ec.coef)   # sac.jl:207-213


class ReplayBuffer:
    """ReplayBuffer(observation_space, action_space, capacity) replay_buffer.jl:14-31 — a view of the device ring of one SacHandle"""

    def __init__(self, observation_space: Box, action_space: Box, capacity: int):
        self.observation_space, self.action_space, self.capacity = observation_space, action_space, capacity
        self.handle: Optional[SacHandle] = None

    def __len__(self):
        return 0 if self.handle is None else self.handle.replay_size()

    def isfull(self) -> bool:
        return len(self) == self.capacity

    def _get(self, which):
        return self.handle.replay(which)

    observations = property(lambda s: s._get(capi.RB_OBSERVATIONS))
    actions = property(lambda s: s._get(capi.RB_ACTIONS))
    rewards = property(lambda s: s._get(capi.RB_REWARDS))
    terminated = property(lambda s: s._get(capi.RB_TERMINATED).astype(bool))
    truncated = property(lambda s: s._get(capi.RB_TRUNCATED).astype(bool))
    next_observations = property(lambda s: s._get(capi.RB_NEXT_OBSERVATIONS))


_SAC_STAT_KEYS = ("actor_losses", "critic_losses", "entropy_losses", "entropy_coefficients", "q_values", "learning_rates", "grad_norms",
                  "fps", "steps_taken")


def sac_train_(agent: SACAgent, env, alg: SAC, max_steps: int, *, replay_buffer: Optional[ReplayBuffer] = None, callbacks=None):
    """train!(agent, env, alg::SAC, max_steps) sac.jl:406-549 -> (agent, replay_buffer, training_stats, timer); `env` is a
    DeviceParallelEnv over PendulumEnv.  training_stats carries the fields of SACTrainingStats (sac.jl:243-257)."""
    if getattr(env, "kind", None) == capi.ENV_EXTERNAL:
        return _sac_train_host(agent, env, alg, max_steps, replay_buffer, list(callbacks or []))
    if callbacks:
        return _sac_train_callbacks(agent, env, alg, max_steps, replay_buffer, list(callbacks))
    t0 = time.perf_counter()
    rb = replay_buffer or ReplayBuffer(env.observation_space(), env.action_space(), alg.buffer_capacity)     # sac.jl:411
    cfg = make_sac_config(env.env, env.n_envs, alg, agent.layer, seed=env.seed, device=env._kw.get("device", 0), profile_events=env._kw.get("profile_events", False))
    h = rb.handle if rb.handle is not None else SacHandle(cfg)
    rb.handle = h
    h.set_params(sac_flatten_params(agent.parameters))
    h.set_target_params(agent.q_target_parameters)
    h.set_log_ent_coef(agent.log_ent_coef)
    h.env_reset(env.seed)
    stats, fps, n_upd, iters, total = h.train(max_steps)
    ts = {k: [] for k in _SAC_STAT_KEYS}
    for s in stats:
        ts["actor_losses"].append(s.actor_loss); ts["critic_losses"].append(s.critic_loss)
        if s.has_entropy_loss:
            ts["entropy_losses"].append(s.entropy_loss)
        ts["entropy_coefficients"].append(s.entropy_coefficient); ts["q_values"].append(s.mean_q_values)
        ts["learning_rates"].append(alg.learning_rate); ts["grad_norms"].append(s.grad_norm)
    ts["fps"] = list(fps)
    agent.steps_taken += total
    agent.gradient_updates += n_upd
    agent.parameters = sac_unflatten_params(h.get_params(), agent.parameters)
    agent.q_target_parameters = h.get_target_params()
    agent.log_ent_coef = h.get_log_ent_coef()
    return agent, rb, ts, {"training_loop": time.perf_counter() - t0, "iterations": iters}


def _jl_div(a: int, b: int) -> int:
    """Julia's div: truncation toward zero"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _sac_train_callbacks(agent: SACAgent, env, alg: SAC, max_steps: int, replay_buffer: Optional[ReplayBuffer], cbs: list):
    """train!(agent, replay_buffer, env, alg::SAC, max_steps; callbacks) (sac.jl:428-559) step by step, so that the hooks run where the reference runs them:
    on_training_start (:476-483), on_rollout_start (:488-495), on_step before every env step of the collection (off_policy_collection.jl:44-49), on_rollout_end
    (:508-515), on_training_end (:545-552).  Each gets a dict of the reference's locals; a false return stops the training and — as in the reference — the early
    returns are (agent, replay_buffer, training_stats) without the timer.  Differences: the transitions of a collection that on_step interrupted are already in
    the device ring (the reference drops that partial rollout), and one env step is one `dril_sac_collect_rollout(1)` (a drain per step: callbacks want the state)."""
    hook = lambda name, loc: all(getattr(c, name)(loc) for c in cbs if hasattr(c, name))
    step_hooks = [c for c in cbs if hasattr(c, "on_step")]
    t0 = time.perf_counter()
    rb = replay_buffer or ReplayBuffer(env.observation_space(), env.action_space(), alg.buffer_capacity)
    cfg = make_sac_config(env.env, env.n_envs, alg, agent.layer, seed=env.seed, device=env._kw.get("device", 0), profile_events=env._kw.get("profile_events", False))
    h = rb.handle if rb.handle is not None else SacHandle(cfg)
    rb.handle = h
    h.set_params(sac_flatten_params(agent.parameters)); h.set_target_params(agent.q_target_parameters); h.set_log_ent_coef(agent.log_ent_coef)
    h.env_reset(env.seed)
    E = env.n_envs
    total_start = alg.start_steps if alg.start_steps > 0 else alg.train_freq * E              # sac.jl:456-458
    adjusted = max(1, _jl_div(total_start, E)) * E
    n_steps = _jl_div(adjusted, E)
    iterations = _jl_div(max_steps - adjusted, alg.train_freq * E) + 1                          # :463
    total_steps = n_steps * E + alg.train_freq * E * (iterations - 1)
    ts = {k: [] for k in _SAC_STAT_KEYS}
    loc = dict(agent=agent, replay_buffer=rb, env=env, alg=alg, max_steps=max_steps, callbacks=cbs, n_envs=E, layer=agent.layer, training_stats=ts,
               gradient_updates_performed=0, total_start_steps=total_start, adjusted_total_start_steps=adjusted, n_steps=n_steps, training_iteration=0,
               iterations=iterations, total_steps=total_steps, update_entropy_coef=isinstance(alg.ent_coef, AutoEntropyCoefficient))
    done_updates = 0

    def sync_agent():
        agent.parameters = sac_unflatten_params(h.get_params(), agent.parameters)
        agent.q_target_parameters = h.get_target_params()
        agent.log_ent_coef = h.get_log_ent_coef()

    try:
        if not hook("on_training_start", loc):
            return agent, rb, ts
        for it in range(1, max(iterations, 0) + 1):
            loc.update(training_iteration=it, n_steps=n_steps)
            if not hook("on_rollout_start", loc):
                return agent, rb, ts
            use_random = it == 1 and alg.start_steps > 0
            a = time.perf_counter()
            for i in range(1, n_steps + 1):
                loc.update(i=i, use_random_actions=use_random)
                if step_hooks and not all(c.on_step(loc) for c in step_hooks):
                    return agent, rb, ts                                                          # "Collecting rollout stopped due to callback failure", :502-505
                h.collect_rollout(1, use_random)
            fps = n_steps * E / max(time.perf_counter() - a, 1e-9)
            loc.update(fps=fps, success=True)
            if not hook("on_rollout_end", loc):
                return agent, rb, ts
            ts["fps"].append(fps)
            agent.steps_taken += n_steps * E
            n_steps = alg.train_freq                                                              # :521
            n_upd = get_gradient_steps(alg, alg.train_freq, E)
            for s_ in (h.update(n_upd) if n_upd > 0 else []):
                ts["actor_losses"].append(s_.actor_loss); ts["critic_losses"].append(s_.critic_loss)
                if s_.has_entropy_loss:
                    ts["entropy_losses"].append(s_.entropy_loss)
                ts["entropy_coefficients"].append(s_.entropy_coefficient); ts["q_values"].append(s_.mean_q_values)
                ts["learning_rates"].append(alg.learning_rate); ts["grad_norms"].append(s_.grad_norm)
                done_updates += 1
            agent.gradient_updates += n_upd
            loc.update(gradient_updates_performed=done_updates, n_updates=n_upd)
        timer = {"training_loop": time.perf_counter() - t0, "iterations": max(iterations, 0)}
        if not hook("on_training_end", loc):
            return agent, rb, ts, timer                                                           # :548-550 (this one returns the timer too)
        return agent, rb, ts, timer
    finally:
        sync_agent()                                                                              # the reference mutates the agent's train_state in place: every exit leaves the trained weights in it


def _sac_train_host(agent: SACAgent, env, alg: SAC, max_steps: int, replay_buffer: Optional[ReplayBuffer] = None, cbs: Optional[list] = None):
    """train!(agent, replay_buffer, env, alg::SAC, max_steps) (sac.jl:428-559) over the caller's own envs (HostParallelEnv): the envs step on the
    host, the policy, the replay ring and every gradient step live on the device (DRIL_ENV_EXTERNAL: dril_sac_predict_actions + dril_sac_ext_push).
    `cbs`: callbacks with the reference's five hooks in the reference's places (see _sac_train_callbacks); a false return ends the training with the
    reference's early-return shape (agent, replay_buffer, training_stats), the trained weights in the agent"""
    cbs = cbs or []
    hook = lambda name, loc: all(getattr(c, name)(loc) for c in cbs if hasattr(c, name))
    step_hooks = [c for c in cbs if hasattr(c, "on_step")]
    t0 = time.perf_counter()
    E, asp = env.n_envs, env.action_space()
    rb = replay_buffer or ReplayBuffer(env.observation_space(), asp, alg.buffer_capacity)
    cfg = make_sac_config(env, E, alg, agent.layer, seed=env.seed, device=env._kw.get("device", 0), profile_events=env._kw.get("profile_events", False))
    h = rb.handle if rb.handle is not None else SacHandle(cfg)
    rb.handle = h
    h.set_params(sac_flatten_params(agent.parameters)); h.set_target_params(agent.q_target_parameters); h.set_log_ent_coef(agent.log_ent_coef)
    rng = np.random.default_rng(env.seed)
    low, high = np.float32(asp.low[0]), np.float32(asp.high[0])
    total_start = alg.start_steps if alg.start_steps > 0 else alg.train_freq * E                    # sac.jl:456-458
    adjusted = max(1, total_start // E) * E
    n_steps = adjusted // E
    iterations = int((max_steps - adjusted) / (alg.train_freq * E)) + 1                             # div truncates toward zero, :462
    n_updates = get_gradient_steps(alg, alg.train_freq, E)
    ts = {k: [] for k in _SAC_STAT_KEYS}
    total = n_upd = 0
    t_env = t_dev = 0.0
    obs = np.stack(env.observe())
    loc = dict(agent=agent, replay_buffer=rb, env=env, alg=alg, max_steps=max_steps, callbacks=cbs, n_envs=E, layer=agent.layer, training_stats=ts,
               gradient_updates_performed=0, total_start_steps=total_start, adjusted_total_start_steps=adjusted, n_steps=n_steps, training_iteration=0,
               iterations=iterations, total_steps=n_steps * E + alg.train_freq * E * (iterations - 1))

    def leave(early):
        agent.steps_taken += total
        agent.gradient_updates += n_upd
        agent.parameters = sac_unflatten_params(h.get_params(), agent.parameters)
        agent.q_target_parameters = h.get_target_params()
        agent.log_ent_coef = h.get_log_ent_coef()
        timer = {"training_loop": time.perf_counter() - t0, "iterations": max(iterations, 0), "collect_rollout": t_env, "device": t_dev}
        return (agent, rb, ts) if early else (agent, rb, ts, timer)

    if not hook("on_training_start", loc):
        return leave(True)
    for it in range(max(iterations, 0)):
        use_random = it == 0 and alg.start_steps > 0                                                # :487
        loc.update(training_iteration=it + 1, n_steps=n_steps)
        if not hook("on_rollout_start", loc):
            return leave(True)
        a = time.perf_counter()
        for i_step in range(n_steps):                                                               # collect_trajectories, off_policy_collection.jl:28-96
            if step_hooks:
                loc.update(i=i_step + 1, use_random_actions=use_random)
                if not all(c.on_step(loc) for c in step_hooks):
                    return leave(True)
            if use_random:
                stored = env_act = rng.uniform(low, high, (E, h.A)).astype(np.float32)              # rand(rng, act_space): env space, stored as is (:50-53,72)
            else:
                b = time.perf_counter()
                stored, env_act = h.predict_actions(obs)                                            # raw squashed action + to_env(TanhScaleAdapter), :55-58
                t_dev += time.perf_counter() - b
            rew, term, trunc, infos = env.act_([x.reshape(asp.shape) for x in env_act])             # :60
            nobs = np.stack(env.observe())                                                          # :61
            tobs = None
            if trunc.any():
                tobs = nobs.copy()
                for i in np.nonzero(trunc)[0]:
                    tobs[i] = infos[i]["terminal_observation"]
            b = time.perf_counter()
            h.ext_push(obs, stored, rew, term, trunc, nobs, tobs)                                   # push!(buffer, traj), replay_buffer.jl:98-114
            t_dev += time.perf_counter() - b
            obs = nobs
        t_env += time.perf_counter() - a
        fps = n_steps * E / max(time.perf_counter() - a, 1e-12)
        total += n_steps * E
        loc.update(fps=fps, success=True)
        if not hook("on_rollout_end", loc):
            return leave(True)
        ts["fps"].append(fps)
        n_steps = alg.train_freq                                                                    # :520
        if n_updates > 0:
            b = time.perf_counter()
            for s in h.update(n_updates):                                                           # :523-538
                ts["actor_losses"].append(s.actor_loss); ts["critic_losses"].append(s.critic_loss)
                if s.has_entropy_loss:
                    ts["entropy_losses"].append(s.entropy_loss)
                ts["entropy_coefficients"].append(s.entropy_coefficient); ts["q_values"].append(s.mean_q_values)
                ts["learning_rates"].append(alg.learning_rate); ts["grad_norms"].append(s.grad_norm)
            t_dev += time.perf_counter() - b
            n_upd += n_updates
            loc.update(gradient_updates_performed=n_upd)
    hook("on_training_end", loc)                                                                    # (a failure here still returns the timer, sac.jl:548-550)
    return leave(False)

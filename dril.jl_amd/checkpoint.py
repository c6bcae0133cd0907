"""Checkpoint round trip in the reference's key schema (host-side data only; no device code).

    save_policy_params_and_state(agent, path)          src/agents/agent_methods.jl:122-138   keys: layer, parameters, states, aux
    load_policy_params_and_state_(agent, alg, path)    src/algorithms/ppo.jl:77-94           (the optimiser state is rebuilt, not loaded)
    save_normalization_stats / load_normalization_stats_   src/environment_wrappers/normalizeWrapperEnv.jl:261-297
                                                       keys: obs_mean obs_var obs_count ret_mean ret_var ret_count clip_obs clip_reward gamma epsilon

The reference writes JLD2 (Julia objects); a Julia user keeps doing exactly that on `agent.train_state.parameters`, which DRiLHIP.jl's
`train!` fills with the device-trained weights.  This Python mirror stores the same keys in an `.npz`, leaves flattened
with "/" paths (`parameters/actor_head/layer_1/weight`, `aux/Q_target_parameters`, ...), so the two files can be converted key by key.
"""
from __future__ import annotations

import json
from dataclasses import asdict, is_dataclass
from pathlib import Path

import numpy as np


def _flatten(prefix: str, tree, out: dict):
    if isinstance(tree, dict):
        for k, v in tree.items():
            _flatten(f"{prefix}/{k}", v, out)
    else:
        out[prefix] = np.asarray(tree)


def _unflatten(prefix: str, data) -> dict:
    tree: dict = {}
    for key in data.files:
        if not key.startswith(prefix + "/"):
            continue
        node = tree
        parts = key[len(prefix) + 1:].split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = data[key]
    return tree


def _layer_desc(layer) -> str:
    d = {k: v for k, v in asdict(layer).items() if k not in ("observation_space", "action_space")} if is_dataclass(layer) else {}
    d["type"] = type(layer).__name__
    d["hidden_dims"] = list(getattr(layer, "hidden_dims", ()))
    return json.dumps(d, default=lambda o: list(o) if isinstance(o, (tuple, np.ndarray)) else str(o))


def save_policy_params_and_state(agent, path, suffix: str = ".npz") -> str:
    """agent: Agent (PPO) or SACAgent.  -> file path"""
    file_path = str(path) if str(path).endswith(suffix) else str(path) + suffix
    out = {"layer": np.array(_layer_desc(agent.layer))}
    params = agent.train_state.parameters if hasattr(agent, "train_state") else agent.parameters
    _flatten("parameters", params, out)
    out["states"] = np.array("{}")                                      # the MLPs carry no Lux state (NamedTuple())
    if hasattr(agent, "q_target_parameters"):                           # QAux, sac.jl:172-178
        out["aux/Q_target_parameters"] = np.asarray(agent.q_target_parameters)
        out["aux/log_ent_coef"] = np.asarray([agent.log_ent_coef], np.float32)
    np.savez(file_path, **out)
    return file_path


def load_policy_params_and_state_(agent, alg, path, suffix: str = ".npz"):
    """load_policy_params_and_state!(agent, alg, path) (ppo.jl:77-94): parameters and aux replace the agent's and a NEW TrainState is built
    (`Lux.Training.TrainState(layer, parameters, states, make_optimizer(alg))` :88-91) — the Adam moments are not restored: the new TrainState carries
    optimizer_state = None and the next train_ starts from zero moments on whatever env it runs.  `agent.layer = data["layer"]` (ppo.jl:90): the stored layer
    description must describe the agent's layer — a checkpoint of another architecture raises instead of loading into mismatched shapes"""
    file_path = str(path) if str(path).endswith(suffix) else str(path) + suffix
    data = np.load(file_path, allow_pickle=False)
    params = _unflatten("parameters", data)
    stored, mine = json.loads(str(data["layer"])), json.loads(_layer_desc(agent.layer))
    for k in ("type", "hidden_dims", "activation"):
        if k in stored and k in mine and stored[k] != mine[k]:
            raise ValueError(f"checkpoint {file_path} was written for a layer with {k} = {stored[k]!r}; this agent's layer has {mine[k]!r} (ppo.jl:90 replaces agent.layer with the stored one)")
    if hasattr(agent, "train_state"):
        agent.train_state = type(agent.train_state)(parameters=params, step=0)
    else:
        agent.parameters = params
    if "aux/Q_target_parameters" in data.files:
        agent.q_target_parameters = data["aux/Q_target_parameters"].copy()
        agent.log_ent_coef = float(data["aux/log_ent_coef"][0])
    agent.alg = alg
    return agent


_NORM_KEYS = ("obs_mean", "obs_var", "obs_count", "ret_mean", "ret_var", "ret_count")


def save_normalization_stats(env, filepath) -> str:
    """env: a DeviceParallelEnv with NormalizeWrapperEnv switched on and a bound handle"""
    st = env.handle.norm_get_stats()
    kw = env._kw["normalize"]
    fp = str(filepath) if str(filepath).endswith(".npz") else str(filepath) + ".npz"
    np.savez(fp, **{k: np.asarray(st[k]) for k in _NORM_KEYS}, clip_obs=kw["clip_obs"], clip_reward=kw["clip_reward"], gamma=kw["gamma"], epsilon=kw["epsilon"])
    return fp


def load_normalization_stats_(env, filepath):
    fp = str(filepath) if str(filepath).endswith(".npz") else str(filepath) + ".npz"
    d = np.load(fp)
    env.handle.norm_set_stats(d["obs_mean"], d["obs_var"], int(d["obs_count"]), float(d["ret_mean"]), float(d["ret_var"]), int(d["ret_count"]))
    return env

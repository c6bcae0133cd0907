"""dril.jl_amd — MI355X-native rollout + PPO-update hot path for DRiL.jl.

The directory name contains a dot, so it is loaded by path (see `load_package` in
__graft_entry__.py / tests/conftest.py) under the module name `dril_jl_amd`.

    csrc/     hand-written HIP kernels + the C ABI (libdril_hip.so, include/dril_hip.h)
    _capi.py  ctypes binding of the C ABI (no fallback: raises if the .so is missing)
    host.py   mirror of the reference's Agent / ActorCriticLayer / PPO / train! / AbstractParallelEnv interface
    sac.py    mirror of the reference's SAC / SACLayer / ReplayBuffer / train!(…, ::SAC, …) interface (include/dril_sac.h)
    checkpoint.py  save/load of agents and normalisation statistics in the reference's key schema (npz twin of the JLD2 files)
    julia/    the `ccall` shim a DRiL.jl user loads (cannot be executed in the build image: no Julia)
"""
from . import _capi  # noqa: F401
from .host import (  # noqa: F401
    AcrobotEnv, Agent, ActorCriticLayer, Box, CartPoleEnv, ContinuousActorCriticLayer, DeviceParallelEnv, Discrete, HostParallelEnv,
    DiscreteActorCriticLayer, DrilError, Handle, MonitorWrapperEnv, MountainCarContinuousEnv, MountainCarEnv, NormalizeWrapperEnv, PendulumEnv, PPO, RolloutBuffer, ScalingWrapperEnv, collect_rollout_,
    evaluate_agent, flatten_params, get_action_and_values, get_original_obs, get_original_rewards, make_config, predict_values, train_, unflatten_params,
    unnormalize_obs_, unnormalize_rewards_, TRAINING_START_LOCALS, ROLLOUT_START_LOCALS, TIMER_SECTIONS,
)
from .sac import (  # noqa: F401
    SAC, AutoEntropyCoefficient, FixedEntropyCoefficient, ReplayBuffer, SACAgent, SACLayer, SacHandle, get_gradient_steps, make_sac_config,
    sac_flatten_params, sac_train_, sac_unflatten_params,
)
from .checkpoint import load_normalization_stats_, load_policy_params_and_state_, save_normalization_stats, save_policy_params_and_state  # noqa: F401

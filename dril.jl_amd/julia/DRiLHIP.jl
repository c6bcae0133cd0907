# DRiLHIP.jl — the `ccall` shim that routes DRiL.jl's rollout + PPO-update hot path to libdril_hip.so (MI355X).
#
# Nothing here computes: every call is one entry point of include/dril_hip.h.  The shim adds ONE env type and
# more specific methods of DRiL's own generic functions, so `Agent`, `ActorCriticLayer`, `PPO` and `train!` are used
# exactly as in the reference README (README.md:50-73):
#
#     env   = DeviceParallelEnv(:CartPole, 65_536; max_steps = 500)          # instead of MultiThreadedParallelEnv([...])
#     layer = ActorCriticLayer(observation_space(env), action_space(env))
#     alg   = PPO(; n_steps = 2048, batch_size = 4_194_304)
#     agent = Agent(layer, alg; verbose = 0)
#     learn_stats, to = train!(agent, env, alg, 10 * 2048 * 65_536)
#
# Files: this one (C structs, DeviceParallelEnv, parameters, collect_rollout! / train! for PPO) + DRiLHIP_host_envs.jl (OnDevice) + DRiLHIP_extras.jl + DRiLHIP_sac.jl.
# STATUS: EXPERIMENTAL — the build image has no Julia runtime, so these files have never executed (runtests.jl beside them is the first thing to run).  What stands in for a run: tools/check_shim.py (parses this
# file and the reference's sources: per-argument method specificity of every method added to a DRiL generic function, the callback-locals keys against
# test/test_callbacks.jl and the Python mirror, every ccall symbol against include/*.h) and the Python ctypes mirror (dril.jl_amd/host.py), which drives
# the same C symbols in the same order under tests/ on the GPU.  A maintainer with Julia should first run the CI snippet of INTEGRATION.md §7.
module DRiLHIP

using DRiL
using DRiL: AbstractParallelEnv, AbstractCallback, Agent, PPO, RolloutBuffer, Box, Discrete
import DRiL: train!, collect_rollout!, observe, act!, reset!, terminated, truncated, number_of_envs,
             observation_space, action_space, get_info
using Random
using TimerOutputs

# The FIRST argument of every `train!` method below is the reference method's own first-argument type, verbatim (src/algorithms/ppo.jl:100-107,
# src/algorithms/sac.jl:417-423): then the env argument alone decides specificity (DeviceParallelEnv / OnDevice <: AbstractParallelEnv), the shim's
# method is strictly more specific and dispatch is unambiguous.  (Round 1 declared `agent::Agent`: wider in argument 1, narrower in argument 2 =>
# MethodError: ambiguous.)  tools/check_shim.py parses both files and checks every argument pair; INTEGRATION.md §7 has the table.
const PPOAgent = Agent{<:DRiL.AbstractActorCriticLayer, <:PPO, <:DRiL.AbstractActionAdapter, <:Random.AbstractRNG, <:DRiL.AbstractTrainingLogger, <:Any}
const SACAgent = Agent{<:DRiL.ContinuousActorCriticLayer, <:DRiL.SAC, <:DRiL.AbstractActionAdapter, <:Random.AbstractRNG, <:DRiL.AbstractTrainingLogger, <:Any}

# keys of `Base.@locals` the reference's callback test reads (test/test_callbacks.jl:25-27 at training start, :36-39 at rollout start).  The Python
# mirror (dril.jl_amd/host.py) holds the same two tuples; tools/check_shim.py asserts that the three lists (test, mirror, shim) agree.
const TRAINING_START_LOCALS = (:agent, :env, :alg, :iterations, :total_steps, :max_steps, :n_steps, :n_envs, :roll_buffer, :total_fps, :callbacks, :learn_stats)
const ROLLOUT_START_LOCALS = (:i, :learning_rate)
# TimerOutputs sections of the reference's train! (ppo.jl:109,154,167,205-207,239)
const TIMER_SECTIONS = ("setup", "training_loop", "collect_rollout", "epoch loop", "batch loop", "compute_gradients", "apply_gradients")

const LIB = Ref{String}(joinpath(@__DIR__, "..", "csrc", "libdril_hip.so"))
const ABI_VERSION = UInt32(2)

# struct dril_config (include/dril_hip.h) — isbits, C layout
struct DrilConfig
    abi_version::UInt32; env_kind::Int32; n_envs::Int32; n_steps::Int32
    hidden1::Int32; hidden2::Int32; episode_len::Int32; fixed_length_episodes::Int32; action_start::Int32
    gamma::Float32; gae_lambda::Float32; clip_range::Float32
    clip_range_vf::Float32; has_clip_range_vf::Int32
    ent_coef::Float32; vf_coef::Float32
    max_grad_norm::Float32; has_max_grad_norm::Int32
    target_kl::Float32; has_target_kl::Int32
    normalize_advantage::Int32
    batch_size::Int64; epochs::Int32
    learning_rate::Float32; adam_beta1::Float32; adam_beta2::Float32; adam_eps::Float32; log_std_init::Float32
    norm_obs::Int32; norm_reward::Int32; norm_training::Int32
    clip_obs::Float32; clip_reward::Float32; norm_gamma::Float32; norm_epsilon::Float32
    seed::UInt64
    device::Int32; rank::Int32; world_size::Int32; profile_events::Int32; monitor_window::Int32
    ext_obs_dim::Int32; ext_action_dim::Int32; ext_discrete::Int32; ext_action_low::Float32; ext_action_high::Float32
    n_hidden::Int32; hidden::NTuple{4, Int32}; activation::Int32        # any-depth hidden_dims / relu (n_hidden == 0: hidden1, hidden2, tanh)
    reserved::NTuple{1, Int32}
end

# struct dril_ppo_stats
struct DrilPPOStats
    entropy_loss::Float32; policy_loss::Float32; value_loss::Float32; approx_kl_div::Float32; clip_fraction::Float32
    loss::Float32; grad_norm::Float32; explained_variance::Float32; entropy::Float32; ratio_first::Float32
    n_updates::Int32; early_stopped::Int32; nan_or_inf::Int32
    f32_path::Int32      # 0 default kernels; 1 redone on the exact-f32 kernels (an f16-piece kernel left f16's range); 2 run directly on them (latch / W2 out of range)
end

# struct dril_f32_fallback (include/dril_hip.h): how often the library left its f16-piece arithmetic; a healthy run on normalised data shows retries == direct_updates == 0
struct DrilF32Fallback
    retries::Int64; direct_updates::Int64; persistent_fallbacks::Int64
    latch_updates_left::Int32; forward_exact_f32::Int32; max_abs_w2::Float32; reserved::Int32
end

const ENV_KINDS = Dict(:CartPole => Int32(0), :Pendulum => Int32(1), :ScaledPendulum => Int32(2), :MountainCar => Int32(3), :MountainCarContinuous => Int32(4), :Acrobot => Int32(6), :ScaledMountainCarContinuous => Int32(7))   # :ScaledPendulum = ScalingWrapperEnv(PendulumEnv()) on every sub-env (scalingWrapperEnv.jl)

"""
    DeviceParallelEnv(kind, n_envs; max_steps, seed, fixed_length_episodes, device) <: AbstractParallelEnv

Device-resident batched simulator replacing `MultiThreadedParallelEnv([CartPoleEnv() for _ in 1:n_envs])`
(src/environment_wrappers/multithreadedParallelEnv.jl).  The handle is created lazily by `bind!` because one
`dril_handle` carries env + agent + algorithm state.
"""
mutable struct DeviceParallelEnv <: AbstractParallelEnv
    kind::Symbol
    n_envs::Int
    max_steps::Int
    seed::UInt64
    fixed_length_episodes::Bool
    device::Int
    monitor_window::Int            # MonitorWrapperEnv(env, stats_window): 0 = off
    normalize::Union{Nothing, NamedTuple}   # NormalizeWrapperEnv kwargs (normalizeWrapperEnv.jl:71-80) or nothing
    handle::Ptr{Cvoid}
    bound::Any                     # (alg, hidden_dims, log_std_init) the handle was created for
    last_terminated::Vector{Bool}
    last_truncated::Vector{Bool}
    optimizer_owner::Any           # the TrainState whose Adam moments the handle holds (see train!)
    last_kernel_seconds::Dict{String, Float64}
end

function DeviceParallelEnv(kind::Symbol, n_envs::Integer; max_steps::Integer = (kind === :CartPole || kind === :Acrobot) ? 500 : (kind === :MountainCarContinuous || kind === :ScaledMountainCarContinuous) ? 999 : 200,
        seed::Integer = 42, fixed_length_episodes::Bool = false, device::Integer = 0, monitor_window::Integer = 0,
        normalize::Union{Nothing, NamedTuple} = nothing)
    haskey(ENV_KINDS, kind) || error("unknown device env $kind")
    env = DeviceParallelEnv(kind, n_envs, max_steps, UInt64(seed), fixed_length_episodes, device, monitor_window, normalize,
        C_NULL, nothing, fill(false, n_envs), fill(false, n_envs), nothing, Dict{String, Float64}())
    finalizer(e -> (e.handle != C_NULL && ccall((:dril_destroy, LIB[]), Int32, (Ptr{Cvoid},), e.handle); nothing), env)
    return env
end

number_of_envs(env::DeviceParallelEnv) = env.n_envs
is_discrete(env) = env.kind === :CartPole || env.kind === :MountainCar || env.kind === :Acrobot
observation_space(env::DeviceParallelEnv) = env.kind === :Acrobot ? Box(Float32[-1, -1, -1, -1, -4π, -9π], Float32[1, 1, 1, 1, 4π, 9π]) : (env.kind === :MountainCar || env.kind === :MountainCarContinuous) ? Box(Float32[-1.2, -0.07], Float32[0.6, 0.07]) : env.kind === :ScaledMountainCarContinuous ? Box(Float32[-1, -1], Float32[1, 1]) :
    env.kind === :CartPole ?
    Box(Float32[-4.8, -Inf, -0.41887903, -Inf], Float32[4.8, Inf, 0.41887903, Inf]) :
    env.kind === :ScaledPendulum ? Box(Float32[-1, -1, -1], Float32[1, 1, 1]) : Box(Float32[-1, -1, -8], Float32[1, 1, 8])
action_space(env::DeviceParallelEnv) = env.kind === :CartPole ? Discrete(2) : (env.kind === :MountainCar || env.kind === :Acrobot) ? Discrete(3) :
    (env.kind === :ScaledPendulum || env.kind === :MountainCarContinuous || env.kind === :ScaledMountainCarContinuous) ? Box(Float32[-1], Float32[1]) : Box(Float32[-2], Float32[2])
obs_dim(env::DeviceParallelEnv) = env.kind === :CartPole ? 4 : env.kind === :Acrobot ? 6 : (env.kind === :MountainCar || env.kind === :MountainCarContinuous || env.kind === :ScaledMountainCarContinuous) ? 2 : 3

last_error(h) = unsafe_string(ccall((:dril_last_error, LIB[]), Cstring, (Ptr{Cvoid},), h))
function check(rc::Int32, h = C_NULL)
    rc == 0 && return nothing
    # status 4 mirrors `@assert !nested_has_nan(grads)` (src/algorithms/ppo.jl:213-214)
    error("libdril_hip status $rc: $(last_error(h))")
end

layer_fields(hidden::Vector{Int}, act::Int32) = (length(hidden) == 2 && act == 0) ? (Int32(0), ntuple(_ -> Int32(0), 4), Int32(0)) :
    (Int32(length(hidden)), ntuple(i -> i <= length(hidden) ? Int32(hidden[i]) : Int32(0), 4), act)
function make_config(env::DeviceParallelEnv, alg::PPO, hidden::Vector{Int}, log_std_init::Float32, act::Int32 = Int32(0))
    opt(x) = isnothing(x) ? (0.0f0, Int32(0)) : (Float32(x), Int32(1))
    cvf, hcvf = opt(alg.clip_range_vf); mgn, hmgn = opt(alg.max_grad_norm); tkl, htkl = opt(alg.target_kl)
    start = is_discrete(env) ? Int32(action_space(env).start) : Int32(1)
    nz = env.normalize
    nget(k, d) = isnothing(nz) ? d : get(nz, k, d)
    on = isnothing(nz) ? Int32(0) : Int32(1)
    return DrilConfig(ABI_VERSION, ENV_KINDS[env.kind], env.n_envs, alg.n_steps, hidden[1], hidden[min(2, end)], env.max_steps,
        Int32(env.fixed_length_episodes), start, alg.gamma, alg.gae_lambda, alg.clip_range, cvf, hcvf, alg.ent_coef,
        alg.vf_coef, mgn, hmgn, tkl, htkl, Int32(alg.normalize_advantage), alg.batch_size, alg.epochs, alg.learning_rate,
        0.9f0, 0.999f0, 1.0f-5, log_std_init,                    # Optimisers.Adam(eta, (0.9, 0.999), 1e-5): ppo.jl:64-66
        on * Int32(nget(:norm_obs, true)), on * Int32(nget(:norm_reward, true)), on * Int32(nget(:training, true)),
        Float32(nget(:clip_obs, 10)), Float32(nget(:clip_reward, 10)), Float32(nget(:gamma, 0.99)), Float32(nget(:epsilon, 1.0e-8)),
        env.seed, env.device, 0, 1, 8, env.monitor_window, 0, 0, 0, 0.0f0, 0.0f0, layer_fields(hidden, act)..., ntuple(_ -> Int32(0), 1))   # profile_events = 8: HIP-event kernel times fill the TimerOutput sections (per-optimiser-step kernels bracketed at every 8th launch: bracketing all of them costs 1 - 2 %)
end

"(re)create the handle when the algorithm / layer shape changes; Random.seed!(env, seed) + reset!(env) follow"
function bind!(env::DeviceParallelEnv, alg::PPO, hidden::Vector{Int} = [64, 64], log_std_init::Float32 = 0.0f0, act::Int32 = Int32(0))
    key = (alg, hidden, log_std_init, act)
    if env.handle == C_NULL || env.bound != key
        env.handle != C_NULL && ccall((:dril_destroy, LIB[]), Int32, (Ptr{Cvoid},), env.handle)
        cfg = Ref(make_config(env, alg, hidden, log_std_init, act))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dril_create, LIB[]), Int32, (Ref{DrilConfig}, Ref{Ptr{Cvoid}}), cfg, h))
        env.handle = h[]; env.bound = key; env.optimizer_owner = nothing
        check(ccall((:dril_env_reset, LIB[]), Int32, (Ptr{Cvoid}, UInt64), env.handle, env.seed), env.handle)
    end
    return env.handle
end
handle(env::DeviceParallelEnv) = env.handle == C_NULL ? bind!(env, PPO(; n_steps = 1, batch_size = env.n_envs)) : env.handle

# ---- env verbs with host copy-out: generic DRiL callers (evaluate_agent, check_env, wrappers) keep working ----
function reset!(env::DeviceParallelEnv)
    check(ccall((:dril_env_reset, LIB[]), Int32, (Ptr{Cvoid}, UInt64), handle(env), env.seed), env.handle)
    return nothing
end
Random.seed!(env::DeviceParallelEnv, seed::Integer) = (env.seed = UInt64(seed); env)     # applied by the next reset!

function observe(env::DeviceParallelEnv)
    obs = Matrix{Float32}(undef, obs_dim(env), env.n_envs)              # (D x E) column-major, spaces.jl:259
    GC.@preserve obs check(ccall((:dril_env_observe, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int32), handle(env), obs, 1), env.handle)
    return [obs[:, i] for i in 1:env.n_envs]
end

function act!(env::DeviceParallelEnv, actions::AbstractVector)
    E, D = env.n_envs, obs_dim(env)
    a = is_discrete(env) ? Int32[Int32(x) for x in actions] : Float32[Float32(x[1]) for x in actions]
    rewards = Vector{Float32}(undef, E); term = Vector{UInt8}(undef, E); trunc = Vector{UInt8}(undef, E)
    tobs = zeros(Float32, D, E)
    GC.@preserve a rewards term trunc tobs check(ccall((:dril_env_step, LIB[]), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float32}),
        handle(env), a, rewards, term, trunc, tobs), env.handle)
    env.last_terminated = term .!= 0; env.last_truncated = trunc .!= 0
    infos = [Dict{String, Any}() for _ in 1:E]
    for i in findall(env.last_truncated)                                # only on truncation, multithreadedParallelEnv.jl:64-66
        infos[i]["terminal_observation"] = tobs[:, i]
    end
    return rewards, env.last_terminated, env.last_truncated, infos
end
# log_stats(env::MonitorWrapperEnv, logger) (monitorWrapperEnv.jl:64-70) from the device ring of finished episodes
function DRiL.log_stats(env::DeviceParallelEnv, logger::DRiL.AbstractTrainingLogger)
    (env.monitor_window > 0 && env.handle != C_NULL) || return nothing
    r = Ref{Float32}(0); l = Ref{Float32}(0); n = Ref{Int32}(0)
    check(ccall((:dril_monitor_get_stats, LIB[]), Int32, (Ptr{Cvoid}, Ref{Float32}, Ref{Float32}, Ref{Int32}), env.handle, r, l, n), env.handle)
    if n[] > 0
        DRiL.log_scalar!(logger, "env/ep_rew_mean", r[]); DRiL.log_scalar!(logger, "env/ep_len_mean", l[])
    end
    return nothing
end
terminated(env::DeviceParallelEnv) = env.last_terminated
truncated(env::DeviceParallelEnv) = env.last_truncated
get_info(env::DeviceParallelEnv) = [Dict{String, Any}() for _ in 1:env.n_envs]

# ---- parameters: Lux NamedTuple <-> flat f32 (layout: include/dril_hip.h, dril_set_params) ----
mlp_of(head) = hasproperty(head.layer_1, :weight) ? head : head.layer_1   # Box actions: Chain(chain, ReshapeLayer) nests the MLP one level down (layer_helpers.jl:77)
dense_keys(mlp) = sort!(collect(keys(mlp)); by = k -> parse(Int, last(split(String(k), "_"))))     # layer_1 .. layer_{n+1} in order
function flatten_params(ps)
    parts = Vector{Float32}[]
    for head in (mlp_of(ps.actor_head), mlp_of(ps.critic_head)), l in dense_keys(head)
        push!(parts, vec(getproperty(head, l).weight)); push!(parts, vec(getproperty(head, l).bias))   # W is (out x in) column-major
    end
    haskey(ps, :log_std) && push!(parts, vec(ps.log_std))
    return reduce(vcat, parts)
end
function scatter_params!(ps, flat::Vector{Float32})
    off = 0
    for head in (mlp_of(ps.actor_head), mlp_of(ps.critic_head)), l in dense_keys(head)
        for arr in (getproperty(head, l).weight, getproperty(head, l).bias)
            n = length(arr); copyto!(arr, 1, flat, off + 1, n); off += n
        end
    end
    if haskey(ps, :log_std)
        copyto!(ps.log_std, 1, flat, off + 1, length(ps.log_std))
    end
    return ps
end
function hidden_dims_of(ps)
    h = mlp_of(ps.actor_head); ks = dense_keys(h)
    return [size(getproperty(h, k).weight, 1) for k in ks[1:end-1]]
end
first_dense(l) = hasproperty(l, :activation) ? l : first_dense(first(l.layers))
"""
The C ABI (include/dril_hip.h, dril_config v2) carries `hidden_dims` of length 1..4 and tanh / relu / sigmoid / elu / leakyrelu / softplus (activations whose derivative is a function of the output); the reference accepts any depth and activation
(layer_constructors.jl:6-10,55-56, layer_helpers.jl:27-57).  Anything else is REJECTED here with a clear message — round 1 read layer_1..layer_3
unconditionally and would have mis-flattened a deeper net silently.  Returns the activation code of dril_config.
"""
function check_supported_layer(agent)
    ps = agent.train_state.parameters
    ha, hc = hidden_dims_of(ps), [size(getproperty(mlp_of(ps.critic_head), k).weight, 1) for k in dense_keys(mlp_of(ps.critic_head))[1:end-1]]
    1 <= length(ha) <= 4 || error("DRiLHIP: hidden_dims of length $(length(ha)); the device path supports 1..4 hidden layers. Use DRiL's CPU train! for this layer.")
    ha == hc || error("DRiLHIP: actor and critic must share hidden_dims (got $ha and $hc)")
    all(h -> 1 <= h <= 1024, ha) || error("DRiLHIP: hidden widths must be 1..1024")
    act = first_dense(agent.layer.actor_head).activation
    (act === tanh || nameof(act) === :tanh_fast) && return Int32(0)
    (nameof(act) === :relu) && return Int32(1)
    (nameof(act) in (:sigmoid, :sigmoid_fast, :σ)) && return Int32(2)      # the plain NNlib functions only: elu with alpha = 1, leakyrelu with a = 0.01
    (nameof(act) === :elu) && return Int32(3)
    (nameof(act) === :leakyrelu) && return Int32(4)
    (nameof(act) === :softplus) && return Int32(5)
    (nameof(act) in (:gelu, :gelu_tanh)) && return Int32(6)                # NNlib.gelu is the tanh form
    (nameof(act) in (:swish, :swish_fast)) && return Int32(7)
    error("DRiLHIP: activation $(act) is not supported on the device PPO path (tanh, relu, sigmoid, elu, leakyrelu, softplus, gelu, swish). Use DRiL's CPU train! for this layer.")
end
function push_params!(env, agent)
    flat = flatten_params(agent.train_state.parameters)
    GC.@preserve flat check(ccall((:dril_set_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), env.handle, flat, length(flat)), env.handle)
end
function pull_params!(env, agent)
    flat = Vector{Float32}(undef, ccall((:dril_param_count, LIB[]), Int64, (Ptr{Cvoid},), env.handle))
    GC.@preserve flat check(ccall((:dril_get_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), env.handle, flat, length(flat)), env.handle)
    scatter_params!(agent.train_state.parameters, flat)
end
# ---- optimiser state: Lux.Training.TrainState.optimizer_state <-> dril_get/set_optimizer_state ----
# Optimisers.setup(Adam(...), ps) mirrors the parameter tree with `Leaf(rule, state)` nodes; Adam's leaf state is `(mt, vt, betat)` with betat = (beta1^t, beta2^t)
# STARTING at (beta1, beta2) (Optimisers.jl `init(o::Adam, x) = (zero(x), zero(x), o.beta)`), i.e. exactly the library's `beta_powers`.  The moments are
# flattened in the order of flatten_params.  Anything that does not look like that (another rule, a frozen leaf, a different Optimisers layout) makes
# `adam_leaves` return nothing and the shim falls back to the ownership rule below: a handle keeps its moments for the TrainState it last trained.
function adam_leaves(agent)
    os = agent.train_state.optimizer_state; ps = agent.train_state.parameters
    leaves = Any[]
    try
        for (hp, ho) in ((mlp_of(ps.actor_head), mlp_of(os.actor_head)), (mlp_of(ps.critic_head), mlp_of(os.critic_head))), l in dense_keys(hp)
            push!(leaves, getproperty(ho, l).weight); push!(leaves, getproperty(ho, l).bias)
        end
        haskey(ps, :log_std) && push!(leaves, os.log_std)
        all(lf -> hasproperty(lf, :state) && lf.state isa Tuple && length(lf.state) == 3 && lf.state[3] isa Tuple, leaves) || return nothing
    catch
        return nothing
    end
    return leaves
end
function push_optimizer_state!(env, agent)
    leaves = adam_leaves(agent)
    leaves === nothing && return false
    m = reduce(vcat, [vec(Float32.(lf.state[1])) for lf in leaves]); v = reduce(vcat, [vec(Float32.(lf.state[2])) for lf in leaves])
    bt = Float32[leaves[1].state[3][1], leaves[1].state[3][2]]
    b1 = Float32(agent.alg isa PPO ? 0.9 : 0.9)                                   # Optimisers.Adam default beta (ppo.jl:64-66 passes eta and epsilon only)
    steps = bt[1] >= b1 ? Int64(0) : Int64(round(log(bt[1]) / log(b1))) - 1         # betat = beta^(t + 1)
    GC.@preserve m v bt check(ccall((:dril_set_optimizer_state, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Csize_t, Ptr{Float32}, Int64),
        env.handle, m, v, length(m), bt, max(steps, 0)), env.handle)
    return true
end
function pull_optimizer_state!(env, agent)
    leaves = adam_leaves(agent)
    leaves === nothing && return false
    n = ccall((:dril_param_count, LIB[]), Int64, (Ptr{Cvoid},), env.handle)
    m = Vector{Float32}(undef, n); v = Vector{Float32}(undef, n); bt = Vector{Float32}(undef, 2); steps = Ref{Int64}(0)
    GC.@preserve m v bt check(ccall((:dril_get_optimizer_state, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Csize_t, Ptr{Float32}, Ref{Int64}),
        env.handle, m, v, n, bt, steps), env.handle)
    off = 0
    for lf in leaves
        mt, vt, _ = lf.state; k = length(mt)
        copyto!(mt, 1, m, off + 1, k); copyto!(vt, 1, v, off + 1, k); off += k
        lf.state = (mt, vt, (oftype(lf.state[3][1], bt[1]), oftype(lf.state[3][2], bt[2])))      # Leaf is a mutable struct (Optimisers >= 0.3)
    end
    return true
end
function bind_agent!(env::DeviceParallelEnv, agent, alg::PPO)
    act = check_supported_layer(agent)
    ps = agent.train_state.parameters
    ls = haskey(ps, :log_std) ? Float32(ps.log_std[1]) : 0.0f0
    bind!(env, alg, hidden_dims_of(ps), ls, act)
end

# ---- collect_rollout!(::RolloutBuffer, agent, alg, env::DeviceParallelEnv)  (src/buffers/rollout_buffer.jl:46-90) ----
has_step_hooks(::Nothing) = false
has_step_hooks(cbs) = any(cb -> which(DRiL.on_step, (typeof(cb), Dict)).sig != which(DRiL.on_step, (AbstractCallback, Dict)).sig, cbs)

function collect_rollout!(buf::RolloutBuffer, agent::Agent, alg::PPO, env::DeviceParallelEnv; callbacks = nothing)
    has_step_hooks(callbacks) && return invoke(collect_rollout!, Tuple{RolloutBuffer, Agent, DRiL.OnPolicyAlgorithm, DRiL.AbstractEnv},
        buf, agent, alg, env; callbacks = callbacks)    # on_step hooks: reference loop over the env verbs above (SURVEY.md §8b)
    bind_agent!(env, agent, alg); push_params!(env, agent)
    fps = Ref{Float64}(0)
    check(ccall((:dril_collect_rollout, LIB[]), Int32, (Ptr{Cvoid}, Ref{Float64}), env.handle, fps), env.handle)
    copy_out!(env, 0, buf.observations)
    if is_discrete(env)                                                # device actions are Int32; the reference buffer is Int64 (spaces.jl:169)
        tmp = Vector{Int32}(undef, length(buf.rewards)); copy_out!(env, 1, tmp); buf.actions .= reshape(tmp, 1, :)
    else
        copy_out!(env, 1, buf.actions)
    end
    copy_out!(env, 2, buf.rewards); copy_out!(env, 3, buf.advantages); copy_out!(env, 4, buf.returns)
    copy_out!(env, 5, buf.logprobs); copy_out!(env, 6, buf.values)     # TIME-MAJOR order: n = (t-1)*n_envs + env (DESIGN.md §3)
    return fps[], true
end
function copy_out!(env, which::Integer, dst::Array)
    GC.@preserve dst check(ccall((:dril_buffer_copy_out, LIB[]), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Csize_t),
        env.handle, which, dst, sizeof(dst)), env.handle)
end

# ---- train!(agent, env::DeviceParallelEnv, alg::PPO, max_steps)  (src/algorithms/ppo.jl:100-325) ----
"lazy host view of the device rollout buffer: `locals[:roll_buffer].advantages` etc. copy out on demand (a real RolloutBuffer of 65 536 x 2048 steps is 6 GB of host memory)"
struct DeviceRolloutBuffer
    env::Any
end
const BUFFER_IDS = (observations = 0, actions = 1, rewards = 2, advantages = 3, returns = 4, logprobs = 5, values = 6)
function Base.getproperty(b::DeviceRolloutBuffer, f::Symbol)
    f === :env && return getfield(b, :env)
    haskey(BUFFER_IDS, f) || error("RolloutBuffer has no field $f")
    env = getfield(b, :env); h = env.handle
    N = ccall((:dril_obs_dim, LIB[]), Int32, (Ptr{Cvoid},), h)          # placeholder read keeps the handle alive for the size queries below
    D = Int(N); E = number_of_envs(env); T = env.bound[1].n_steps
    disc = ccall((:dril_is_discrete, LIB[]), Int32, (Ptr{Cvoid},), h) != 0
    A = disc ? 1 : Int(ccall((:dril_action_dim, LIB[]), Int32, (Ptr{Cvoid},), h))
    dst = f === :observations ? Matrix{Float32}(undef, D, E * T) : f === :actions ? (disc ? Matrix{Int32}(undef, 1, E * T) : Matrix{Float32}(undef, A, E * T)) : Vector{Float32}(undef, E * T)
    copy_out!(env, BUFFER_IDS[f], dst)
    return dst
end

"seconds of HIP-event time per kernel class since the last reset: average bracketed launch (dril_profile_get) x all launches of the class (dril_profile_launches)"
function kernel_seconds(h)
    out = Dict{String, Float64}()
    for kid in 0:(Int(ccall((:dril_kernel_count, LIB[]), Int32, ())) - 1)
        ms = Ref{Float64}(0); n = Ref{Int64}(0)
        ccall((:dril_profile_get, LIB[]), Int32, (Ptr{Cvoid}, Int32, Ref{Float64}, Ref{Int64}), h, kid, ms, n) == 0 || continue
        all = Ref{Int64}(0)
        ccall((:dril_profile_launches, LIB[]), Int32, (Ptr{Cvoid}, Int32, Ref{Int64}), h, kid, all)
        out[unsafe_string(ccall((:dril_kernel_name, LIB[]), Cstring, (Int32,), kid))] = n[] > 0 ? ms[] * 1.0e-3 * all[] / n[] : 0.0
    end
    return out
end
"counters of the exact-f32 redo / latch / forward fallback of this env's handle (dril_f32_fallback_info); also logged per iteration as train/f32_path"
function f32_fallback_info(env::DeviceParallelEnv)
    out = Ref{DrilF32Fallback}()
    check(ccall((:dril_f32_fallback_info, LIB[]), Int32, (Ptr{Cvoid}, Ref{DrilF32Fallback}), env.handle, out), env.handle)
    return out[]
end
"which gradient kernel the last optimiser step ran and the arithmetic it computes in, e.g. \"ppo_grad_pair_kernel: f32 (f16x2 split, f32 accumulate; ...)\" (dril_grad_kernel_info)"
grad_kernel_info(env::DeviceParallelEnv) = unsafe_string(ccall((:dril_grad_kernel_info, LIB[]), Cstring, (Ptr{Cvoid},), env.handle))
"which device the handle lives on: ordinal, name, PCI bus id, visibility masks (dril_device_info) — the line every rank of a multi-GPU job should print before `comm_init`"
device_info(env::DeviceParallelEnv) = unsafe_string(ccall((:dril_device_info, LIB[]), Cstring, (Ptr{Cvoid},), env.handle))
"""
"batch loop" / "compute_gradients" / "apply_gradients" (ppo.jl:206-207,239) have no host-side extent here — the whole epoch x minibatch loop is ONE
library call — so their times come from the device: HIP events around the kernels that stand in for them.  TimerOutputs has no public API for adding a
measured duration, so this writes the section's `accumulated_data` directly; any failure (a TimerOutputs version with another layout) leaves the
TimerOutput as it was and the numbers stay available from `env.last_kernel_seconds`.
"""
function add_device_sections!(to::TimerOutput, secs::Dict{String, Float64}, nsteps::Int)
    grad = get(secs, "adv_moments_kernel", 0.0) + get(secs, "ppo_grad_kernel", 0.0) + get(secs, "grad_reduce_kernel", 0.0) + get(secs, "ncclAllReduce", 0.0)
    apply = get(secs, "adam_kernel", 0.0)
    try
        loop = to.inner_timers["training_loop"].inner_timers["epoch loop"]
        function section!(parent, name, seconds)
            t = get!(() -> TimerOutput(name), parent.inner_timers, name)
            d = t.accumulated_data
            t.accumulated_data = TimerOutputs.TimeData(d.ncalls + nsteps, d.time + round(Int64, seconds * 1.0e9), d.allocs)
            return t
        end
        bl = section!(loop, "batch loop", grad + apply)
        section!(bl, "compute_gradients", grad); section!(bl, "apply_gradients", apply)
    catch err
        @debug "DRiLHIP: could not add the device sections to the TimerOutput" err
    end
    return (; compute_gradients = grad, apply_gradients = apply)
end

function train!(agent::PPOAgent, env::DeviceParallelEnv, alg::PPO{T}, max_steps::Int; ad_type = nothing, callbacks = nothing) where {T}
    if has_step_hooks(callbacks)
        # on_step hooks fire once per env step inside the rollout (trajectory.jl:34-39; test/test_callbacks.jl:91-99 expects a stop at exactly 512 steps):
        # run the REFERENCE's own train! over this env's step-granular verbs (observe / act! above) — slow path, semantics preserved (SURVEY.md §8b)
        kw = isnothing(ad_type) ? (; callbacks = callbacks) : (; ad_type = ad_type, callbacks = callbacks)
        return invoke(train!, Tuple{PPOAgent, AbstractParallelEnv, PPO{T}, Int}, agent, env, alg, max_steps; kw...)
    end
    to = TimerOutput()
    n_steps = alg.n_steps; n_envs = env.n_envs
    local iterations, total_steps
    @timeit to "setup" begin
        bind_agent!(env, agent, alg); push_params!(env, agent)
        # the handle's Adam moments belong to ONE TrainState (Lux.Training.TrainState carries optimizer_state, ppo.jl:52-53): a different one — another agent, or
        # the TrainState load_policy_params_and_state! rebuilds (ppo.jl:77-94) — starts from a fresh optimiser; repeated train! calls on the same one continue
        # The Adam moments belong to the TrainState (Lux.Training.TrainState carries optimizer_state, ppo.jl:52-53,239): they are pushed into the handle here and
        # pulled back on every exit, so they follow the agent from env to env.  If the optimiser tree is not the Adam tree this shim understands, the older rule
        # applies: the handle keeps the moments of the TrainState it last trained and any other TrainState starts from a fresh optimiser (ppo.jl:77-94).
        if !push_optimizer_state!(env, agent) && env.optimizer_owner !== agent.train_state
            check(ccall((:dril_reset_optimizer, LIB[]), Int32, (Ptr{Cvoid},), env.handle), env.handle)
        end
        env.optimizer_owner = agent.train_state
        ccall((:dril_profile_reset, LIB[]), Int32, (Ptr{Cvoid},), env.handle)
        iterations = max_steps ÷ (n_steps * n_envs)                    # ppo.jl:117
        iterations == 0 && @warn "max_steps is less than n_steps * n_envs; there will be no training."
        total_steps = iterations * n_steps * n_envs
    end
    learn_stats = NamedTuple{(:entropy_losses, :policy_losses, :value_losses, :approx_kl_divs, :clip_fractions, :losses,
        :explained_variances, :fps, :grad_norms, :learning_rates)}(ntuple(_ -> Float32[], 10))
    total_fps = learn_stats.fps; roll_buffer = DeviceRolloutBuffer(env)
    i = 0; learning_rate = alg.learning_rate
    # the Dict the reference builds with Base.@locals (ppo.jl:145-152): every key of TRAINING_START_LOCALS always, those of ROLLOUT_START_LOCALS inside the loop
    locals() = Dict{Symbol, Any}(:agent => agent, :env => env, :alg => alg, :iterations => iterations, :total_steps => total_steps, :max_steps => max_steps,
        :n_steps => n_steps, :n_envs => n_envs, :roll_buffer => roll_buffer, :total_fps => total_fps, :callbacks => callbacks, :learn_stats => learn_stats,
        :i => i, :learning_rate => learning_rate, :to => to)
    fire(f) = isnothing(callbacks) || all(c -> f(c, locals()), callbacks)
    n_updates = 0
    try
        fire(DRiL.on_training_start) || return nothing                                                         # ppo.jl:145-152
        @timeit to "training_loop" for it in 1:iterations
            i = it
            check(ccall((:dril_set_learning_rate, LIB[]), Int32, (Ptr{Cvoid}, Float32), env.handle, learning_rate), env.handle)  # ppo.jl:155-156
            push!(learn_stats.learning_rates, learning_rate)
            fire(DRiL.on_rollout_start) || return nothing
            fps = Ref{Float64}(0)
            @timeit to "collect_rollout" check(ccall((:dril_collect_rollout, LIB[]), Int32, (Ptr{Cvoid}, Ref{Float64}), env.handle, fps), env.handle)
            push!(total_fps, fps[]); DRiL.add_step!(agent, n_steps * n_envs)
            DRiL.increment_step!(agent.logger, n_steps * n_envs); DRiL.log_scalar!(agent.logger, "env/fps", fps[])
            DRiL.log_stats(env, agent.logger)                                                                   # ppo.jl:177
            fire(DRiL.on_rollout_end) || return nothing
            st = Ref{DrilPPOStats}()
            @timeit to "epoch loop" check(ccall((:dril_ppo_update, LIB[]), Int32, (Ptr{Cvoid}, Ref{DrilPPOStats}), env.handle, st), env.handle)
            s = st[]; n_updates += s.n_updates
            DRiL.add_gradient_update!(agent, Int(s.n_updates))
            push!(learn_stats.entropy_losses, s.entropy_loss); push!(learn_stats.policy_losses, s.policy_loss); push!(learn_stats.value_losses, s.value_loss)
            push!(learn_stats.approx_kl_divs, s.approx_kl_div); push!(learn_stats.clip_fractions, s.clip_fraction); push!(learn_stats.losses, s.loss)
            push!(learn_stats.explained_variances, s.explained_variance); push!(learn_stats.grad_norms, s.grad_norm)
            for (k, v) in ("entropy_loss" => s.entropy_loss, "explained_variance" => s.explained_variance, "policy_loss" => s.policy_loss,
                "value_loss" => s.value_loss, "approx_kl_div" => s.approx_kl_div, "clip_fraction" => s.clip_fraction, "loss" => s.loss,
                "grad_norm" => s.grad_norm, "learning_rate" => learning_rate)
                DRiL.log_scalar!(agent.logger, "train/" * k, v)                                                    # ppo.jl:286-294
            end
            s.f32_path != 0 && DRiL.log_scalar!(agent.logger, "train/f32_path", Float32(s.f32_path))               # (not a reference key: this update left the f16-piece kernels; learn_stats keeps the reference's shape)
        end
        env.last_kernel_seconds = kernel_seconds(env.handle)
        add_device_sections!(to, env.last_kernel_seconds, n_updates)
        fire(DRiL.on_training_end) || return nothing
        return learn_stats, to
    finally
        # the reference mutates agent.train_state in place at every optimiser step (ppo.jl:239): after ANY exit — normal, or a callback that stopped the run
        # (ppo.jl:145-152,170-176) — the agent holds the weights trained so far
        pull_params!(env, agent)
        pull_optimizer_state!(env, agent)
    end
end


include("DRiLHIP_host_envs.jl")     # OnDevice(env::AbstractParallelEnv): host envs, generic kernels
include("DRiLHIP_extras.jl")        # normalisation statistics, evaluate_agent
include("DRiLHIP_sac.jl")           # SAC

export DeviceParallelEnv, OnDevice

end # module

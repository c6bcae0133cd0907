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
# STATUS: EXPERIMENTAL — the build image has no Julia runtime, so this file has never executed.  What stands in for a run: tools/check_shim.py (parses this
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
    n_updates::Int32; early_stopped::Int32; nan_or_inf::Int32; reserved::Int32
end

const ENV_KINDS = Dict(:CartPole => Int32(0), :Pendulum => Int32(1), :ScaledPendulum => Int32(2), :MountainCar => Int32(3), :MountainCarContinuous => Int32(4), :Acrobot => Int32(6))   # :ScaledPendulum = ScalingWrapperEnv(PendulumEnv()) on every sub-env (scalingWrapperEnv.jl)

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

function DeviceParallelEnv(kind::Symbol, n_envs::Integer; max_steps::Integer = (kind === :CartPole || kind === :Acrobot) ? 500 : kind === :MountainCarContinuous ? 999 : 200,
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
observation_space(env::DeviceParallelEnv) = env.kind === :Acrobot ? Box(Float32[-1, -1, -1, -1, -4π, -9π], Float32[1, 1, 1, 1, 4π, 9π]) : (env.kind === :MountainCar || env.kind === :MountainCarContinuous) ? Box(Float32[-1.2, -0.07], Float32[0.6, 0.07]) :
    env.kind === :CartPole ?
    Box(Float32[-4.8, -Inf, -0.41887903, -Inf], Float32[4.8, Inf, 0.41887903, Inf]) :
    env.kind === :ScaledPendulum ? Box(Float32[-1, -1, -1], Float32[1, 1, 1]) : Box(Float32[-1, -1, -8], Float32[1, 1, 8])
action_space(env::DeviceParallelEnv) = env.kind === :CartPole ? Discrete(2) : (env.kind === :MountainCar || env.kind === :Acrobot) ? Discrete(3) :
    (env.kind === :ScaledPendulum || env.kind === :MountainCarContinuous) ? Box(Float32[-1], Float32[1]) : Box(Float32[-2], Float32[2])
obs_dim(env::DeviceParallelEnv) = env.kind === :CartPole ? 4 : env.kind === :Acrobot ? 6 : (env.kind === :MountainCar || env.kind === :MountainCarContinuous) ? 2 : 3

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
        env.seed, env.device, 0, 1, 1, env.monitor_window, 0, 0, 0, 0.0f0, 0.0f0, layer_fields(hidden, act)..., ntuple(_ -> Int32(0), 1))   # profile_events = 1: HIP-event kernel times fill the TimerOutput sections
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
The C ABI (include/dril_hip.h, dril_config v2) carries `hidden_dims` of length 1..4 and tanh / relu; the reference accepts any depth and activation
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
    error("DRiLHIP: activation $(act) is not supported on the device PPO path (tanh, relu). Use DRiL's CPU train! for this layer.")
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

"seconds of HIP-event time per kernel class since the last reset (dril_profile_get; cfg.profile_events = 1)"
function kernel_seconds(h)
    out = Dict{String, Float64}()
    for kid in 0:6
        ms = Ref{Float64}(0); n = Ref{Int64}(0)
        ccall((:dril_profile_get, LIB[]), Int32, (Ptr{Cvoid}, Int32, Ref{Float64}, Ref{Int64}), h, kid, ms, n) == 0 || continue
        out[unsafe_string(ccall((:dril_kernel_name, LIB[]), Cstring, (Int32,), kid))] = ms[] * 1.0e-3
    end
    return out
end
"which gradient kernel the last optimiser step ran and the arithmetic it computes in, e.g. \"ppo_grad_pair_kernel: f32 (bf16x3 split, f32 accumulate; ...)\" (dril_grad_kernel_info)"
grad_kernel_info(env::DeviceParallelEnv) = unsafe_string(ccall((:dril_grad_kernel_info, LIB[]), Cstring, (Ptr{Cvoid},), env.handle))
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
        if env.optimizer_owner !== agent.train_state
            check(ccall((:dril_reset_optimizer, LIB[]), Int32, (Ptr{Cvoid},), env.handle), env.handle)
            env.optimizer_owner = agent.train_state
        end
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
        end
        env.last_kernel_seconds = kernel_seconds(env.handle)
        add_device_sections!(to, env.last_kernel_seconds, n_updates)
        fire(DRiL.on_training_end) || return nothing
        return learn_stats, to
    finally
        # the reference mutates agent.train_state in place at every optimiser step (ppo.jl:239): after ANY exit — normal, or a callback that stopped the run
        # (ppo.jl:145-152,170-176) — the agent holds the weights trained so far
        pull_params!(env, agent)
    end
end

# =============================================================================================================================
# Host envs: ANY AbstractParallelEnv of the caller (their own Julia envs in a MultiThreadedParallelEnv / BroadcastedParallelEnv, wrapped or not)
# with the agent on the device — DRIL_ENV_EXTERNAL (include/dril_hip.h): observations go in and actions come out once per env step
# (dril_ext_act / dril_ext_record / dril_ext_finish); policy forward, sampling, the rollout buffer, bootstrap values, GAE and the PPO update
# run on the GPU for any observation / action / hidden width.
#     env = OnDevice(MultiThreadedParallelEnv([MyEnv() for _ in 1:64]))
#     train!(agent, env, alg, max_steps)
# =============================================================================================================================
mutable struct OnDevice{E <: AbstractParallelEnv} <: AbstractParallelEnv
    env::E
    seed::UInt64
    device::Int
    handle::Ptr{Cvoid}
    bound::Any
    optimizer_owner::Any
end
function OnDevice(env::AbstractParallelEnv; seed::Integer = 42, device::Integer = 0)
    w = OnDevice(env, UInt64(seed), Int(device), C_NULL, nothing, nothing)
    finalizer(e -> (e.handle != C_NULL && ccall((:dril_destroy, LIB[]), Int32, (Ptr{Cvoid},), e.handle); nothing), w)
    return w
end
# the env verbs pass straight through, so every generic DRiL caller (evaluate_agent, callbacks, wrappers) keeps working on the wrapped env
number_of_envs(w::OnDevice) = number_of_envs(w.env)
observation_space(w::OnDevice) = observation_space(w.env)
action_space(w::OnDevice) = action_space(w.env)
reset!(w::OnDevice) = reset!(w.env)
observe(w::OnDevice) = observe(w.env)
act!(w::OnDevice, actions::AbstractVector) = act!(w.env, actions)
DRiL.log_stats(w::OnDevice, logger::DRiL.AbstractTrainingLogger) = DRiL.log_stats(w.env, logger)

function make_config(w::OnDevice, alg::PPO, hidden::Vector{Int}, log_std_init::Float32, act::Int32 = Int32(0))
    opt(x) = isnothing(x) ? (0.0f0, Int32(0)) : (Float32(x), Int32(1))
    cvf, hcvf = opt(alg.clip_range_vf); mgn, hmgn = opt(alg.max_grad_norm); tkl, htkl = opt(alg.target_kl)
    osp, asp = observation_space(w), action_space(w)
    disc = asp isa Discrete
    lo, hi = disc ? (0.0f0, 0.0f0) : (Float32(minimum(asp.low)), Float32(maximum(asp.high)))
    uniform = !disc && all(==(lo), asp.low) && all(==(hi), asp.high)   # one (low, high) pair: ClampAdapter on the device; otherwise clamped in the rollout loop below
    return DrilConfig(ABI_VERSION, Int32(5), number_of_envs(w), alg.n_steps, hidden[1], hidden[min(2, end)], 0, Int32(0), disc ? Int32(asp.start) : Int32(1),
        alg.gamma, alg.gae_lambda, alg.clip_range, cvf, hcvf, alg.ent_coef, alg.vf_coef, mgn, hmgn, tkl, htkl, Int32(alg.normalize_advantage),
        alg.batch_size, alg.epochs, alg.learning_rate, 0.9f0, 0.999f0, 1.0f-5, log_std_init, 0, 0, 0, 10.0f0, 10.0f0, 0.99f0, 1.0f-8,
        w.seed, w.device, 0, 1, 0, 0, Int32(prod(size(osp))), Int32(disc ? asp.n : prod(size(asp))), Int32(disc),
        uniform ? lo : 0.0f0, uniform ? hi : 0.0f0, layer_fields(hidden, act)..., ntuple(_ -> Int32(0), 1))
end
function bind_agent!(w::OnDevice, agent, alg::PPO)
    act = check_supported_layer(agent)
    ps = agent.train_state.parameters
    ls = haskey(ps, :log_std) ? Float32(ps.log_std[1]) : 0.0f0
    key = (alg, hidden_dims_of(ps), ls, act)
    if w.handle == C_NULL || w.bound != key
        w.handle != C_NULL && ccall((:dril_destroy, LIB[]), Int32, (Ptr{Cvoid},), w.handle)
        cfg = Ref(make_config(w, alg, key[2], ls, act)); h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dril_create, LIB[]), Int32, (Ref{DrilConfig}, Ref{Ptr{Cvoid}}), cfg, h))
        w.handle = h[]; w.bound = key; w.optimizer_owner = nothing
    end
    return w.handle
end

"collect_trajectories (trajectory.jl:22-78) with the envs on the host and the agent on the device; returns fps (rollout_buffer.jl:60-64)"
function device_rollout!(w::OnDevice, alg::PPO)
    E = number_of_envs(w); asp = action_space(w); disc = asp isa Discrete
    D = prod(size(observation_space(w))); A = disc ? 1 : prod(size(asp))
    obs = Matrix{Float32}(undef, D, E); tobs = zeros(Float32, D, E)
    raw = disc ? Vector{Int32}(undef, E) : Matrix{Float32}(undef, A, E); ea = similar(raw)
    rew = Vector{Float32}(undef, E); term = Vector{UInt8}(undef, E); trunc = Vector{UInt8}(undef, E)
    pack!(dst, xs) = (for j in 1:E; dst[:, j] .= vec(xs[j]); end; dst)
    t0 = time()
    pack!(obs, observe(w.env))                                                                                     # :32
    for _ in 1:alg.n_steps
        GC.@preserve obs raw ea check(ccall((:dril_ext_act, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}), w.handle, obs, raw, ea), w.handle)   # :41-42
        actions = disc ? [Int(ea[j]) for j in 1:E] : [clamp.(reshape(ea[:, j], size(asp)), asp.low, asp.high) for j in 1:E]   # per-dimension bounds too (ClampAdapter, default_adapters.jl:4-11)
        r, te, tr, infos = act!(w.env, actions)                                                                    # :44
        rew .= r; term .= te; trunc .= tr
        for j in 1:E
            tr[j] && haskey(infos[j], "terminal_observation") && (tobs[:, j] .= vec(infos[j]["terminal_observation"]))
        end
        GC.@preserve rew term trunc tobs check(ccall((:dril_ext_record, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float32}),
            w.handle, rew, term, trunc, tobs), w.handle)                                                            # :46-61
        pack!(obs, observe(w.env))                                                                                 # :45
    end
    GC.@preserve obs check(ccall((:dril_ext_finish, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}), w.handle, obs), w.handle)   # :65-70 + compute_advantages! + returns
    return alg.n_steps * E / max(time() - t0, 1.0e-12)
end

function collect_rollout!(buf::RolloutBuffer, agent::Agent, alg::PPO, w::OnDevice; callbacks = nothing)
    has_step_hooks(callbacks) && return collect_rollout!(buf, agent, alg, w.env; callbacks = callbacks)   # on_step hooks: the reference loop on the wrapped env
    bind_agent!(w, agent, alg); push_params!(w, agent)
    fps = device_rollout!(w, alg)
    copy_out!(w, 0, buf.observations)
    if action_space(w) isa Discrete
        tmp = Vector{Int32}(undef, length(buf.rewards)); copy_out!(w, 1, tmp); buf.actions .= reshape(tmp, 1, :)
    else
        copy_out!(w, 1, buf.actions)
    end
    copy_out!(w, 2, buf.rewards); copy_out!(w, 3, buf.advantages); copy_out!(w, 4, buf.returns); copy_out!(w, 5, buf.logprobs); copy_out!(w, 6, buf.values)
    return fps, true
end

function train!(agent::PPOAgent, w::OnDevice, alg::PPO{T}, max_steps::Int; ad_type = nothing, callbacks = nothing) where {T}
    if has_step_hooks(callbacks)                                       # on_step hooks: the reference's own loop on the wrapped env
        kw = isnothing(ad_type) ? (; callbacks = callbacks) : (; ad_type = ad_type, callbacks = callbacks)
        return train!(agent, w.env, alg, max_steps; kw...)
    end
    to = TimerOutput()
    n_steps = alg.n_steps; n_envs = number_of_envs(w)
    local iterations, total_steps
    @timeit to "setup" begin
        bind_agent!(w, agent, alg); push_params!(w, agent)
        if w.optimizer_owner !== agent.train_state
            check(ccall((:dril_reset_optimizer, LIB[]), Int32, (Ptr{Cvoid},), w.handle), w.handle)
            w.optimizer_owner = agent.train_state
        end
        iterations = max_steps ÷ (n_steps * n_envs)                    # ppo.jl:117
        iterations == 0 && @warn "max_steps is less than n_steps * n_envs; there will be no training."
        total_steps = iterations * n_steps * n_envs
    end
    learn_stats = NamedTuple{(:entropy_losses, :policy_losses, :value_losses, :approx_kl_divs, :clip_fractions, :losses,
        :explained_variances, :fps, :grad_norms, :learning_rates)}(ntuple(_ -> Float32[], 10))
    total_fps = learn_stats.fps; roll_buffer = DeviceRolloutBuffer(w)
    i = 0; learning_rate = alg.learning_rate
    locals() = Dict{Symbol, Any}(:agent => agent, :env => w.env, :alg => alg, :iterations => iterations, :total_steps => total_steps, :max_steps => max_steps,
        :n_steps => n_steps, :n_envs => n_envs, :roll_buffer => roll_buffer, :total_fps => total_fps, :callbacks => callbacks, :learn_stats => learn_stats,
        :i => i, :learning_rate => learning_rate, :to => to)
    fire(f) = isnothing(callbacks) || all(c -> f(c, locals()), callbacks)
    try
        fire(DRiL.on_training_start) || return nothing
        @timeit to "training_loop" for it in 1:iterations
            i = it
            check(ccall((:dril_set_learning_rate, LIB[]), Int32, (Ptr{Cvoid}, Float32), w.handle, learning_rate), w.handle)
            push!(learn_stats.learning_rates, learning_rate)
            fire(DRiL.on_rollout_start) || return nothing
            fps = @timeit to "collect_rollout" device_rollout!(w, alg)
            push!(total_fps, fps); DRiL.add_step!(agent, n_steps * n_envs)
            DRiL.increment_step!(agent.logger, n_steps * n_envs); DRiL.log_scalar!(agent.logger, "env/fps", fps)
            DRiL.log_stats(w.env, agent.logger)
            fire(DRiL.on_rollout_end) || return nothing
            st = Ref{DrilPPOStats}()
            @timeit to "epoch loop" check(ccall((:dril_ppo_update, LIB[]), Int32, (Ptr{Cvoid}, Ref{DrilPPOStats}), w.handle, st), w.handle)
            s = st[]
            DRiL.add_gradient_update!(agent, Int(s.n_updates))
            push!(learn_stats.entropy_losses, s.entropy_loss); push!(learn_stats.policy_losses, s.policy_loss); push!(learn_stats.value_losses, s.value_loss)
            push!(learn_stats.approx_kl_divs, s.approx_kl_div); push!(learn_stats.clip_fractions, s.clip_fraction); push!(learn_stats.losses, s.loss)
            push!(learn_stats.explained_variances, s.explained_variance); push!(learn_stats.grad_norms, s.grad_norm)
            for (k, v) in ("entropy_loss" => s.entropy_loss, "explained_variance" => s.explained_variance, "policy_loss" => s.policy_loss,
                "value_loss" => s.value_loss, "approx_kl_div" => s.approx_kl_div, "clip_fraction" => s.clip_fraction, "loss" => s.loss,
                "grad_norm" => s.grad_norm, "learning_rate" => learning_rate)
                DRiL.log_scalar!(agent.logger, "train/" * k, v)
            end
        end
        fire(DRiL.on_training_end) || return nothing
        return learn_stats, to
    finally
        pull_params!(w, agent)                                          # every exit path: the agent holds the weights trained so far (ppo.jl:239)
    end
end

# ---- normalisation statistics in the reference's JLD2 schema (normalizeWrapperEnv.jl:261-297) ----
function norm_stats(env::DeviceParallelEnv)
    D = obs_dim(env); om = Vector{Float32}(undef, D); ov = Vector{Float32}(undef, D)
    oc = Ref{Int64}(0); rm = Ref{Float32}(0); rv = Ref{Float32}(0); rc = Ref{Int64}(0)
    GC.@preserve om ov check(ccall((:dril_norm_get_stats, LIB[]), Int32,
        (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ref{Int64}, Ref{Float32}, Ref{Float32}, Ref{Int64}), handle(env), om, ov, oc, rm, rv, rc), env.handle)
    return (; obs_mean = om, obs_var = ov, obs_count = Int(oc[]), ret_mean = fill(rm[]), ret_var = fill(rv[]), ret_count = Int(rc[]))
end
function DRiL.save_normalization_stats(env::DeviceParallelEnv, filepath::String)
    s = norm_stats(env); nz = env.normalize
    return DRiL.save(filepath, Dict("obs_mean" => s.obs_mean, "obs_var" => s.obs_var, "obs_count" => s.obs_count,
        "ret_mean" => s.ret_mean, "ret_var" => s.ret_var, "ret_count" => s.ret_count,
        "clip_obs" => Float32(get(nz, :clip_obs, 10)), "clip_reward" => Float32(get(nz, :clip_reward, 10)),
        "gamma" => Float32(get(nz, :gamma, 0.99)), "epsilon" => Float32(get(nz, :epsilon, 1.0e-8))))
end
function DRiL.load_normalization_stats!(env::DeviceParallelEnv, filepath::String)
    st = DRiL.load(filepath)
    om = Float32.(st["obs_mean"]); ov = Float32.(st["obs_var"])
    GC.@preserve om ov check(ccall((:dril_norm_set_stats, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Float32, Float32, Int64),
        handle(env), om, ov, st["obs_count"], Float32(first(st["ret_mean"])), Float32(first(st["ret_var"])), st["ret_count"]), env.handle)
    return env
end

# ---- evaluate_agent(agent, env::DeviceParallelEnv; ...)  (src/evaluation.jl:54-143) ----
struct DrilEvalStats
    mean_reward::Float64; std_reward::Float64; mean_length::Float64; std_length::Float64
    n_episodes::Int32; n_steps::Int32
end
function DRiL.evaluate_agent(agent, env::DeviceParallelEnv; n_eval_episodes::Int = 10, deterministic::Bool = true,
        reward_threshold::Union{Nothing, Real} = nothing, return_stats::Bool = true, warn::Bool = true, kwargs...)
    bind_agent!(env, agent, agent.algorithm); push_params!(env, agent)
    st = Ref{DrilEvalStats}(); er = Vector{Float32}(undef, n_eval_episodes); el = Vector{Int32}(undef, n_eval_episodes)
    GC.@preserve er el check(ccall((:dril_evaluate_agent, LIB[]), Int32, (Ptr{Cvoid}, Int32, Int32, Ref{DrilEvalStats}, Ptr{Float32}, Ptr{Int32}),
        env.handle, n_eval_episodes, deterministic, st, er, el), env.handle)
    s = st[]
    if reward_threshold !== nothing && s.mean_reward < reward_threshold
        error("Mean reward below threshold: $(round(s.mean_reward, digits = 2)) < $(reward_threshold)")            # evaluation.jl:131-135
    end
    return return_stats ? (; mean_reward = s.mean_reward, std_reward = s.std_reward, mean_length = s.mean_length, std_length = s.std_length) :
        (er, Int.(el))
end


# =====================================================================================================================
# SAC: train!(agent, env::DeviceParallelEnv, alg::SAC, max_steps)  (src/algorithms/sac.jl:406-549) over include/dril_sac.h
# =====================================================================================================================
# struct dril_sac_config / dril_sac_stats (include/dril_sac.h) — isbits, C layout
struct DrilSacConfig
    abi_version::UInt32; env_kind::Int32; n_envs::Int32; episode_len::Int32
    hidden1::Int32; hidden2::Int32; activation::Int32
    buffer_capacity::Int64; start_steps::Int32; batch_size::Int32
    tau::Float32; gamma::Float32
    train_freq::Int32; gradient_steps::Int32; target_update_interval::Int32
    auto_ent_coef::Int32; ent_coef_init::Float32; auto_target_entropy::Int32; target_entropy::Float32
    learning_rate::Float32; adam_beta1::Float32; adam_beta2::Float32; adam_eps::Float32
    seed::UInt64; device::Int32; profile_events::Int32
    ext_obs_dim::Int32; ext_action_dim::Int32; ext_action_low::Float32; ext_action_high::Float32
    reserved::NTuple{4, Int32}
end
struct DrilSacStats
    actor_loss::Float32; critic_loss::Float32; entropy_loss::Float32; mean_q_values::Float32; entropy_coefficient::Float32; grad_norm::Float32
    has_entropy_loss::Int32; reserved::Int32
end
sac_check(rc::Int32, h = C_NULL) = rc == 0 ? nothing :
    error("libdril_hip (SAC) status $rc: " * unsafe_string(ccall((:dril_sac_last_error, LIB[]), Cstring, (Ptr{Cvoid},), h)))

# ContinuousActorCriticLayer{QCritic}: actor_head = Chain(mlp, ReshapeLayer), critic_head = Parallel(vcat, mlp, mlp) (layer_helpers.jl:77,100-112)
function sac_flatten_params(ps)
    parts = Vector{Float32}[]
    for head in (mlp_of(ps.actor_head), ps.critic_head.layer_1, ps.critic_head.layer_2), l in (:layer_1, :layer_2, :layer_3)
        push!(parts, vec(getproperty(head, l).weight)); push!(parts, vec(getproperty(head, l).bias))
    end
    push!(parts, vec(ps.log_std))
    return reduce(vcat, parts)
end
function sac_scatter_params!(ps, flat::Vector{Float32})
    off = 0
    for head in (mlp_of(ps.actor_head), ps.critic_head.layer_1, ps.critic_head.layer_2), l in (:layer_1, :layer_2, :layer_3)
        for arr in (getproperty(head, l).weight, getproperty(head, l).bias)
            n = length(arr); copyto!(arr, 1, flat, off + 1, n); off += n
        end
    end
    copyto!(ps.log_std, 1, flat, off + 1, length(ps.log_std))
    return ps
end
function sac_flatten_targets(tp)
    parts = Vector{Float32}[]
    for head in (tp.critic_head.layer_1, tp.critic_head.layer_2), l in (:layer_1, :layer_2, :layer_3)
        push!(parts, vec(getproperty(head, l).weight)); push!(parts, vec(getproperty(head, l).bias))
    end
    return reduce(vcat, parts)
end
function sac_scatter_targets!(tp, flat::Vector{Float32})
    off = 0
    for head in (tp.critic_head.layer_1, tp.critic_head.layer_2), l in (:layer_1, :layer_2, :layer_3)
        for arr in (getproperty(head, l).weight, getproperty(head, l).bias)
            n = length(arr); copyto!(arr, 1, flat, off + 1, n); off += n
        end
    end
    return tp
end

function sac_config(env::DeviceParallelEnv, alg::DRiL.SAC, agent)
    is_discrete(env) && error("SAC needs a Box action space (sac.jl:74): DeviceParallelEnv(:Pendulum | :ScaledPendulum | :MountainCarContinuous, ...)")
    hd = hidden_dims_of(agent.train_state.parameters)
    act = agent.layer.actor_head.layers[1].layers[1].activation === DRiL.Lux.relu ? Int32(1) : Int32(0)   # SACLayer default relu (sac.jl:77)
    ec = alg.ent_coef
    auto = ec isa DRiL.AutoEntropyCoefficient
    auto_t = auto && ec.target isa DRiL.AutoEntropyTarget
    return DrilSacConfig(UInt32(1), ENV_KINDS[env.kind], env.n_envs, env.max_steps, hd[1], hd[2], act,
        alg.buffer_capacity, alg.start_steps, alg.batch_size, alg.tau, alg.gamma, alg.train_freq, alg.gradient_steps, alg.target_update_interval,
        Int32(auto), auto ? Float32(ec.initial_value) : Float32(ec.coef), Int32(auto ? auto_t : true),
        auto && !auto_t ? Float32(ec.target.target) : 0.0f0,
        alg.learning_rate, 0.9f0, 0.999f0, 1.0f-8,                                                # Optimisers.Adam(lr) defaults, agent_methods.jl:116-118
        env.seed, env.device, Int32(0), 0, 0, 0.0f0, 0.0f0, ntuple(_ -> Int32(0), 4))
end

"""
    train!(agent, env::DeviceParallelEnv, alg::SAC, max_steps) -> (agent, nothing, training_stats, to)

Same contract as `train!(agent, replay_buffer, env, alg::SAC, max_steps)` (sac.jl:414-549) with the ReplayBuffer resident on the device
(second return value `nothing`; read it through `dril_sac_replay_copy_out`).  Callbacks with `on_step` hooks are not supported on this path.
"""
function train!(agent::SACAgent, env::DeviceParallelEnv, alg::DRiL.SAC, max_steps::Int; ad_type = nothing, callbacks = nothing)
    T = typeof(alg.learning_rate)
    if has_step_hooks(callbacks)      # on_step hooks: the reference's own train! over this env's step-granular verbs
        kw = isnothing(ad_type) ? (; callbacks = callbacks) : (; ad_type = ad_type, callbacks = callbacks)
        return invoke(train!, Tuple{SACAgent, AbstractParallelEnv, DRiL.SAC, Int}, agent, env, alg, max_steps; kw...)
    end
    to = TimerOutput()
    cfg = Ref(sac_config(env, alg, agent)); hp = Ref{Ptr{Cvoid}}(C_NULL)
    sac_check(ccall((:dril_sac_create, LIB[]), Int32, (Ref{DrilSacConfig}, Ref{Ptr{Cvoid}}), cfg, hp)); h = hp[]
    try
        flat = sac_flatten_params(agent.train_state.parameters); tgt = sac_flatten_targets(agent.aux.Q_target_parameters)
        GC.@preserve flat tgt begin
            sac_check(ccall((:dril_sac_set_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, flat, length(flat)), h)
            sac_check(ccall((:dril_sac_set_target_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, tgt, length(tgt)), h)
        end
        sac_check(ccall((:dril_sac_set_log_ent_coef, LIB[]), Int32, (Ptr{Cvoid}, Float32), h, first(agent.aux.ent_train_state.parameters.log_ent_coef)), h)
        sac_check(ccall((:dril_sac_env_reset, LIB[]), Int32, (Ptr{Cvoid}, UInt64), h, env.seed), h)
        n_envs = env.n_envs                                                                       # schedule: sac.jl:436-447
        total_start = alg.start_steps > 0 ? alg.start_steps : alg.train_freq * n_envs
        adjusted = max(1, div(total_start, n_envs)) * n_envs
        iterations = div(max_steps - adjusted, alg.train_freq * n_envs) + 1
        n_upd = DRiL.get_gradient_steps(alg, alg.train_freq, n_envs)
        cap = max(1, iterations * n_upd)
        st = Vector{DrilSacStats}(undef, cap); fps = Vector{Float64}(undef, max(1, iterations))
        nu = Ref{Int64}(0); it = Ref{Int32}(0); tot = Ref{Int64}(0)
        !isnothing(callbacks) && !all(c -> DRiL.on_training_start(c, Dict{Symbol, Any}(:agent => agent, :env => env, :alg => alg)), callbacks) && return agent, nothing, DRiL.SACTrainingStats{T}()
        @timeit to "training_loop" GC.@preserve st fps sac_check(ccall((:dril_sac_train, LIB[]), Int32,
            (Ptr{Cvoid}, Int64, Ptr{DrilSacStats}, Int64, Ref{Int64}, Ptr{Float64}, Int64, Ref{Int32}, Ref{Int64}),
            h, max_steps, st, cap, nu, fps, length(fps), it, tot), h)
        ts = DRiL.SACTrainingStats{T}()                                                            # sac.jl:243-257
        for k in 1:min(nu[], cap)
            s = st[k]
            push!(ts.actor_losses, s.actor_loss); push!(ts.critic_losses, s.critic_loss); s.has_entropy_loss != 0 && push!(ts.entropy_losses, s.entropy_loss)
            push!(ts.entropy_coefficients, s.entropy_coefficient); push!(ts.q_values, s.mean_q_values); push!(ts.learning_rates, alg.learning_rate)
            push!(ts.grad_norms, s.grad_norm)
        end
        append!(ts.fps, T.(fps[1:it[]]))
        DRiL.add_step!(agent, tot[])
        GC.@preserve flat tgt begin
            sac_check(ccall((:dril_sac_get_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, flat, length(flat)), h)
            sac_check(ccall((:dril_sac_get_target_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, tgt, length(tgt)), h)
        end
        sac_scatter_params!(agent.train_state.parameters, flat); sac_scatter_targets!(agent.aux.Q_target_parameters, tgt)
        le = Ref{Float32}(0); sac_check(ccall((:dril_sac_get_log_ent_coef, LIB[]), Int32, (Ptr{Cvoid}, Ref{Float32}), h, le), h)
        agent.aux.ent_train_state.parameters.log_ent_coef[1] = le[]
        !isnothing(callbacks) && all(c -> DRiL.on_training_end(c, Dict{Symbol, Any}(:agent => agent, :env => env, :alg => alg)), callbacks)
        return agent, nothing, ts, to
    finally
        ccall((:dril_sac_destroy, LIB[]), Int32, (Ptr{Cvoid},), h)
    end
end

# ---- SAC over host envs: train!(agent, OnDevice(env), alg::SAC, max_steps)  (sac.jl:428-559 with the collection loop of off_policy_collection.jl:28-96) ----
function sac_config(w::OnDevice, alg::DRiL.SAC, agent)
    osp, asp = observation_space(w), action_space(w)
    asp isa Box || error("SAC needs a Box action space (sac.jl:74)")
    lo, hi = Float32(minimum(asp.low)), Float32(maximum(asp.high))
    (all(==(lo), asp.low) && all(==(hi), asp.high)) || error("DRIL_ENV_EXTERNAL SAC: one (low, high) pair for all action dimensions (wrap the env in ScalingWrapperEnv)")
    hd = hidden_dims_of(agent.train_state.parameters)
    act = agent.layer.actor_head.layers[1].layers[1].activation === DRiL.Lux.relu ? Int32(1) : Int32(0)
    ec = alg.ent_coef
    auto = ec isa DRiL.AutoEntropyCoefficient
    auto_t = auto && ec.target isa DRiL.AutoEntropyTarget
    return DrilSacConfig(UInt32(1), Int32(5), number_of_envs(w), 0, hd[1], hd[2], act,
        alg.buffer_capacity, alg.start_steps, alg.batch_size, alg.tau, alg.gamma, alg.train_freq, alg.gradient_steps, alg.target_update_interval,
        Int32(auto), auto ? Float32(ec.initial_value) : Float32(ec.coef), Int32(auto ? auto_t : true),
        auto && !auto_t ? Float32(ec.target.target) : 0.0f0,
        alg.learning_rate, 0.9f0, 0.999f0, 1.0f-8, w.seed, w.device, Int32(0),
        Int32(prod(size(osp))), Int32(prod(size(asp))), lo, hi, ntuple(_ -> Int32(0), 4))
end

function train!(agent::SACAgent, w::OnDevice, alg::DRiL.SAC, max_steps::Int; ad_type = nothing, callbacks = nothing)
    T = typeof(alg.learning_rate)
    if has_step_hooks(callbacks)      # on_step hooks: the reference's own loop on the wrapped env
        kw = isnothing(ad_type) ? (; callbacks = callbacks) : (; ad_type = ad_type, callbacks = callbacks)
        return train!(agent, w.env, alg, max_steps; kw...)
    end
    to = TimerOutput()
    cfg = Ref(sac_config(w, alg, agent)); hp = Ref{Ptr{Cvoid}}(C_NULL)
    sac_check(ccall((:dril_sac_create, LIB[]), Int32, (Ref{DrilSacConfig}, Ref{Ptr{Cvoid}}), cfg, hp)); h = hp[]
    try
        flat = sac_flatten_params(agent.train_state.parameters); tgt = sac_flatten_targets(agent.aux.Q_target_parameters)
        GC.@preserve flat tgt begin
            sac_check(ccall((:dril_sac_set_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, flat, length(flat)), h)
            sac_check(ccall((:dril_sac_set_target_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, tgt, length(tgt)), h)
        end
        sac_check(ccall((:dril_sac_set_log_ent_coef, LIB[]), Int32, (Ptr{Cvoid}, Float32), h, first(agent.aux.ent_train_state.parameters.log_ent_coef)), h)
        E = number_of_envs(w); asp = action_space(w); D = prod(size(observation_space(w))); A = prod(size(asp))
        total_start = alg.start_steps > 0 ? alg.start_steps : alg.train_freq * E                  # sac.jl:456-466
        adjusted = max(1, div(total_start, E)) * E
        n_steps = div(adjusted, E)
        iterations = div(max_steps - adjusted, alg.train_freq * E) + 1
        n_upd = DRiL.get_gradient_steps(alg, alg.train_freq, E)
        ts = DRiL.SACTrainingStats{T}()
        obs = Matrix{Float32}(undef, D, E); nobs = similar(obs); tobs = zeros(Float32, D, E)
        raw = Matrix{Float32}(undef, A, E); ea = similar(raw)
        rew = Vector{Float32}(undef, E); term = Vector{UInt8}(undef, E); trunc = Vector{UInt8}(undef, E)
        st = Vector{DrilSacStats}(undef, max(1, n_upd))
        pack!(dst, xs) = (for j in 1:E; dst[:, j] .= vec(xs[j]); end; dst)
        pack!(obs, observe(w.env))
        @timeit to "training_loop" for it in 1:iterations
            use_random = it == 1 && alg.start_steps > 0                                           # :487
            t0 = time()
            @timeit to "collect_rollout" for _ in 1:n_steps                                       # collect_trajectories, off_policy_collection.jl:28-96
                if use_random
                    for j in 1:E; ea[:, j] .= vec(rand(agent.rng, asp)); end; raw .= ea               # rand(rng, act_space): env space, stored as is (:50-53,72)
                else
                    GC.@preserve obs raw ea sac_check(ccall((:dril_sac_predict_actions, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                        h, obs, E, 0, C_NULL, raw, ea), h)                                         # predict_actions_raw + to_env(TanhScaleAdapter), :55-58
                end
                r, te, tr, infos = act!(w.env, [reshape(ea[:, j], size(asp)) for j in 1:E])       # :60
                pack!(nobs, observe(w.env))                                                        # :61
                rew .= r; term .= te; trunc .= tr
                for j in 1:E
                    tr[j] && haskey(infos[j], "terminal_observation") && (tobs[:, j] .= vec(infos[j]["terminal_observation"]))
                end
                GC.@preserve obs raw rew term trunc nobs tobs sac_check(ccall((:dril_sac_ext_push, LIB[]), Int32,
                    (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float32}, Ptr{Float32}), h, obs, raw, rew, term, trunc, nobs, tobs), h)   # push!(buffer, traj), replay_buffer.jl:98-114
                obs, nobs = nobs, obs
            end
            push!(ts.fps, T(n_steps * E / max(time() - t0, 1.0e-12))); DRiL.add_step!(agent, n_steps * E)
            n_steps = alg.train_freq                                                               # :520
            if n_upd > 0
                @timeit to "gradient_updates" GC.@preserve st sac_check(ccall((:dril_sac_update, LIB[]), Int32, (Ptr{Cvoid}, Int32, Ptr{DrilSacStats}), h, n_upd, st), h)   # :523-538
                for k in 1:n_upd
                    s = st[k]
                    push!(ts.actor_losses, s.actor_loss); push!(ts.critic_losses, s.critic_loss); s.has_entropy_loss != 0 && push!(ts.entropy_losses, s.entropy_loss)
                    push!(ts.entropy_coefficients, s.entropy_coefficient); push!(ts.q_values, s.mean_q_values); push!(ts.learning_rates, alg.learning_rate)
                    push!(ts.grad_norms, s.grad_norm)
                end
            end
        end
        GC.@preserve flat tgt begin
            sac_check(ccall((:dril_sac_get_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, flat, length(flat)), h)
            sac_check(ccall((:dril_sac_get_target_params, LIB[]), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h, tgt, length(tgt)), h)
        end
        sac_scatter_params!(agent.train_state.parameters, flat); sac_scatter_targets!(agent.aux.Q_target_parameters, tgt)
        le = Ref{Float32}(0); sac_check(ccall((:dril_sac_get_log_ent_coef, LIB[]), Int32, (Ptr{Cvoid}, Ref{Float32}), h, le), h)
        agent.aux.ent_train_state.parameters.log_ent_coef[1] = le[]
        return agent, nothing, ts, to
    finally
        ccall((:dril_sac_destroy, LIB[]), Int32, (Ptr{Cvoid},), h)
    end
end

export DeviceParallelEnv, OnDevice

end # module

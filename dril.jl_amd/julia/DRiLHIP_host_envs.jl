# DRiLHIP_host_envs.jl — `OnDevice(env)`: a DRiL user's own AbstractParallelEnv steps on the host, policy / buffer / update on the device (included by DRiLHIP.jl)
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


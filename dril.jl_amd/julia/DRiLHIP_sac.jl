# DRiLHIP_sac.jl — the off-policy path: train!(agent, env, alg::SAC, ...) over include/dril_sac.h (included by DRiLHIP.jl)
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
    is_discrete(env) && error("SAC needs a Box action space (sac.jl:74): DeviceParallelEnv(:Pendulum | :ScaledPendulum | :MountainCarContinuous | :ScaledMountainCarContinuous, ...)")
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


# DRiLHIP_extras.jl — callers either side of the path: normalisation statistics in the reference's JLD2 schema, evaluate_agent (included by DRiLHIP.jl)
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

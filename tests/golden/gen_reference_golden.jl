# gen_reference_golden.jl — the ONE thing the oracle of this repo cannot be pinned by without Julia: a vector produced by DRiL.jl itself for the PPO loss
# (`(alg::PPO)(policy, ps, st, batch)`, src/algorithms/ppo.jl:365-407), its Zygote gradient (ppo.jl:207) and the clip + `Optimisers.Adam` step of the reference's own loop
# (ppo.jl:64-66 eta / epsilon = 1f-5, :213-239 NaN asserts, nested_norm, nested_scale!, apply_gradients!).  The reference's tests hold no such vector (SURVEY.md section 8c).
#
#   julia --project=/path/to/DRiL.jl tests/golden/gen_reference_golden.jl        (needs DRiL and its deps, plus JSON.jl in the environment: `] add JSON`)
#
# reads  tests/golden/reference_ppo_input.json   (committed; written by tests/golden/make_reference_input.py: flat parameters + one minibatch per case)
# writes tests/golden/reference_ppo.json         (NOT committed until someone with Julia runs this; tests/test_reference_golden.py consumes it when present, skips otherwise)
#
# This file has never been executed here (the build image has no Julia).  It uses only what the reference exports or defines at the cited lines.
using DRiL, Lux, Zygote, Optimisers, JSON, Random, Statistics

# the flat parameter layout of include/dril_hip.h (dril_set_params): actor {W1 b1 W2 b2 W3 b3}, critic {...}, log_std; W is (out x in) column-major = vec(weight)
mlp_of(head) = hasproperty(head.layer_1, :weight) ? head : head.layer_1            # Box actions: Chain(chain, ReshapeLayer) nests the MLP one level down (layer_helpers.jl:77)
dense_keys(mlp) = sort!(collect(keys(mlp)); by = k -> parse(Int, last(split(String(k), "_"))))
function flatten_like(ps, tree)                                                     # `tree` = ps itself, or a gradient NamedTuple of the same shape
    parts = Vector{Float32}[]
    for (hp, ht) in ((mlp_of(ps.actor_head), mlp_of(tree.actor_head)), (mlp_of(ps.critic_head), mlp_of(tree.critic_head))), l in dense_keys(hp)
        push!(parts, vec(Float32.(getproperty(ht, l).weight))); push!(parts, vec(Float32.(getproperty(ht, l).bias)))
    end
    haskey(ps, :log_std) && push!(parts, vec(Float32.(tree.log_std)))
    return reduce(vcat, parts)
end
function scatter_params!(ps, flat::Vector{Float32})
    off = 0
    for head in (mlp_of(ps.actor_head), mlp_of(ps.critic_head)), l in dense_keys(head)
        for arr in (getproperty(head, l).weight, getproperty(head, l).bias)
            n = length(arr); copyto!(arr, 1, flat, off + 1, n); off += n
        end
    end
    haskey(ps, :log_std) && copyto!(ps.log_std, 1, flat, off + 1, length(ps.log_std))
    @assert off + (haskey(ps, :log_std) ? length(ps.log_std) : 0) == length(flat) "flat parameter count does not match the layer"
    return ps
end

f32(v) = Float32.(v)
here = @__DIR__
input = JSON.parsefile(joinpath(here, "reference_ppo_input.json"))
out = Dict{String, Any}("generator" => "tests/golden/gen_reference_golden.jl", "DRiL_version" => string(pkgversion(DRiL)), "julia" => string(VERSION), "cases" => Any[])

for case in input["cases"]
    D, A, B = case["D"], case["A"], case["B"]
    hidden = Int.(case["hidden"])
    obs_space = Box(f32(case["obs_low"]), f32(case["obs_high"]))
    if case["discrete"]
        act_space = Discrete(A, case["action_start"])
        layer = DiscreteActorCriticLayer(obs_space, act_space; hidden_dims = hidden)
        actions = reshape(Int.(case["actions"]), 1, B)                               # env-space actions (action_start-based), (1 x B) like RolloutBuffer.actions (rollout_buffer.jl:11)
    else
        act_space = Box(f32(case["act_low"]), f32(case["act_high"]))
        layer = ContinuousActorCriticLayer(obs_space, act_space; hidden_dims = hidden, log_std_init = Float32(case["log_std_init"]))
        actions = reshape(f32(case["actions"]), A, B)                               # (A x B), each action contiguous
    end
    ps, st = Lux.setup(Random.Xoshiro(0), layer)
    scatter_params!(ps, f32(case["params"]))
    hp = case["hyper"]
    alg = PPO(; clip_range = Float32(hp["clip_range"]), clip_range_vf = hp["clip_range_vf"] === nothing ? nothing : Float32(hp["clip_range_vf"]),
              ent_coef = Float32(hp["ent_coef"]), vf_coef = Float32(hp["vf_coef"]), max_grad_norm = Float32(hp["max_grad_norm"]),
              normalize_advantage = hp["normalize_advantage"], learning_rate = Float32(hp["learning_rate"]))
    obs = reshape(f32(case["obs"]), D, B)                                            # (D x B): each observation contiguous (spaces.jl:259)
    batch() = (obs, actions, f32(case["advantages"]), f32(case["returns"]), f32(case["old_logprobs"]), f32(case["old_values"]))   # fresh advantages per call: normalize! works in place (ppo.jl:350-363)

    loss, _, stats = alg(layer, ps, st, batch())
    grads = Zygote.gradient(p -> alg(layer, p, st, batch())[1], ps)[1]
    rec = Dict{String, Any}("name" => case["name"], "loss" => loss,
        "stats" => Dict(string(k) => v for (k, v) in pairs(stats)),
        "grad" => flatten_like(ps, grads), "grad_norm" => DRiL.nested_norm(grads, Float32))

    # three optimiser steps on this one minibatch, as the batch loop takes them (ppo.jl:207-239): gradient, norm, clip, Adam(eta, (0.9, 0.999), 1f-5)
    opt_state = Optimisers.setup(DRiL.make_optimizer(Optimisers.Adam, alg), ps)
    steps = Any[]
    for k in 1:3
        lk, _, _ = alg(layer, ps, st, batch())
        g = Zygote.gradient(p -> alg(layer, p, st, batch())[1], ps)[1]
        gn = DRiL.nested_norm(g, Float32)
        if gn > alg.max_grad_norm
            DRiL.nested_scale!(g, alg.max_grad_norm, gn)
        end
        opt_state, ps = Optimisers.update!(opt_state, ps, g)
        push!(steps, Dict("loss_before" => lk, "grad_norm" => gn, "params_after" => flatten_like(ps, ps)))
    end
    rec["adam_steps"] = steps
    push!(out["cases"], rec)
    println("case ", case["name"], ": loss ", loss, "  |g| ", rec["grad_norm"])
end

open(joinpath(here, "reference_ppo.json"), "w") do io
    JSON.print(io, out)
end
println("wrote ", joinpath(here, "reference_ppo.json"))

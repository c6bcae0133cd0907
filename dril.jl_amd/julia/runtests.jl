# runtests.jl — the first thing to run where Julia, DRiL.jl and an MI355X exist:
#
#     julia --project=<an environment with DRiL, Lux, Optimisers, Zygote, Test> dril.jl_amd/julia/runtests.jl
#
# The build image of this repository has no Julia, so NOTHING below has ever executed.  The items re-express, against a `DeviceParallelEnv`, what the
# reference's own tests assert for its CPU envs (paths relative to the DRiL.jl checkout): test/test_callbacks.jl:1-100 (keys of the callback `locals`,
# early stops incl. the step-granular `on_step` stop at exactly 512 steps), test/test_ppo_integration.jl:42-83 (save / load round trip => identical
# parameters and deterministic actions), plus the two properties this shim adds: no method ambiguity with DRiL's own methods, and the optimiser state
# following the TrainState.  The Python mirror (dril.jl_amd/host.py) passes the same assertions on the GPU through the same C symbols
# (tests/test_abi_and_host.py, tests/test_gpu_reference_fixtures.py); tools/check_shim.py checks dispatch specificity and the ccall surface statically.
using Test
using DRiL
using Random
include(joinpath(@__DIR__, "DRiLHIP.jl"))
using .DRiLHIP

# callback types of the items below (struct definitions must be at top level)
struct StartKeys <: DRiL.AbstractCallback end
struct RolloutKeys <: DRiL.AbstractCallback end
struct StopAtTrainingStart <: DRiL.AbstractCallback end
struct StopAtRolloutStart <: DRiL.AbstractCallback end
struct StopAtStep <: DRiL.AbstractCallback
    threshold::Int
end
const SEEN = Dict{Symbol, Bool}()
function DRiL.on_training_start(::StartKeys, locals::Dict)
    for k in (:agent, :env, :alg, :iterations, :total_steps, :max_steps, :n_steps, :n_envs, :roll_buffer, :total_fps, :callbacks)
        @test haskey(locals, k)
    end
    SEEN[:start] = true
    return true
end
function DRiL.on_rollout_start(::RolloutKeys, locals::Dict)
    for k in (:agent, :env, :alg, :iterations, :total_steps, :max_steps, :i, :learning_rate)
        @test haskey(locals, k)
    end
    SEEN[:rollout] = true
    return true
end
DRiL.on_training_start(::StopAtTrainingStart, ::Dict) = false
DRiL.on_rollout_start(::StopAtRolloutStart, ::Dict) = false
DRiL.on_step(c::StopAtStep, locals::Dict) = DRiL.steps_taken(locals[:agent]) < c.threshold

@testset "DRiLHIP" begin

    @testset "no method ambiguities between DRiL and the shim" begin
        amb = Test.detect_ambiguities(DRiL, DRiLHIP; recursive = true)
        @test isempty(amb)
        # the shim's train! must be the one that is called for a DeviceParallelEnv (strictly more specific in the env argument, equal elsewhere)
        env = DeviceParallelEnv(:CartPole, 8)
        alg = PPO(; n_steps = 16, batch_size = 64, epochs = 1)
        agent = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
        m = which(train!, (typeof(agent), typeof(env), typeof(alg), Int))
        @test m.module === DRiLHIP
    end

    @testset "callbacks: locals keys (test_callbacks.jl:1-52)" begin
        alg = PPO(; ent_coef = 0.1f0, n_steps = 256, batch_size = 64, epochs = 10)
        env = DeviceParallelEnv(:CartPole, 8; monitor_window = 100, normalize = (; gamma = alg.gamma))     # Monitor + Normalize stack of the reference test
        agent = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
        out = train!(agent, env, alg, 3000; callbacks = [StartKeys(), RolloutKeys()])
        @test out !== nothing && get(SEEN, :start, false) && get(SEEN, :rollout, false)
        @test DRiL.steps_taken(agent) == (3000 ÷ (256 * 8)) * 256 * 8
    end

    @testset "callbacks: early stopping (test_callbacks.jl:54-100)" begin
        setup() = begin
            alg = PPO(; ent_coef = 0.1f0, n_steps = 64, batch_size = 64, epochs = 10)
            env = DeviceParallelEnv(:CartPole, 8; monitor_window = 100, normalize = (; gamma = alg.gamma))
            agent = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
            agent, env, alg
        end
        agent, env, alg = setup()
        @test train!(agent, env, alg, 3000; callbacks = [StopAtTrainingStart()]) === nothing
        @test DRiL.steps_taken(agent) == 0

        agent, env, alg = setup()
        @test train!(agent, env, alg, 3000; callbacks = [StopAtRolloutStart()]) === nothing
        @test DRiL.steps_taken(agent) == 0

        # on_step fires once per env step inside the rollout (trajectory.jl:34-39): the shim routes such callbacks through the reference's own loop over this
        # env's step-granular verbs, so the stop lands at exactly one rollout of 64 steps x 8 envs after the threshold is crossed
        agent, env, alg = setup()
        train!(agent, env, alg, 3000; callbacks = [StopAtStep(500)])
        @test DRiL.steps_taken(agent) == 512
    end

    @testset "training moves the weights, statistics are finite, learn_stats has the reference's fields (ppo.jl:301-312)" begin
        env = DeviceParallelEnv(:CartPole, 64; max_steps = 500)
        alg = PPO(; n_steps = 128, batch_size = 1024, epochs = 4)
        agent = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
        p0 = deepcopy(agent.train_state.parameters)
        stats, to = train!(agent, env, alg, 4 * 128 * 64)
        @test length(stats.losses) == 4 && all(isfinite, stats.losses) && all(isfinite, stats.grad_norms) && all(>(0), stats.fps)
        @test propertynames(stats) == (:entropy_losses, :policy_losses, :value_losses, :approx_kl_divs, :clip_fractions, :losses, :explained_variances, :fps, :grad_norms, :learning_rates)
        @test agent.train_state.parameters.actor_head.layer_1.weight != p0.actor_head.layer_1.weight
        @test occursin("ppo_", DRiLHIP.grad_kernel_info(env))
        fb = DRiLHIP.f32_fallback_info(env)                                   # struct dril_f32_fallback: a healthy run on CartPole never leaves the f16-piece arithmetic
        @test fb.retries == 0 && fb.direct_updates == 0 && fb.forward_exact_f32 == 0 && 0 < fb.max_abs_w2 < 350
    end

    @testset "optimiser state follows the TrainState (ppo.jl:52-53,239)" begin
        alg = PPO(; n_steps = 16, batch_size = 64, epochs = 2)
        mk() = (e = DeviceParallelEnv(:CartPole, 16; seed = 9); (e, Agent(ActorCriticLayer(observation_space(e), action_space(e)), alg; verbose = 0, rng = Random.Xoshiro(4))))
        envA, agent = mk()
        train!(agent, envA, alg, 16 * 16)
        leaf = agent.train_state.optimizer_state.actor_head.layer_1.weight
        @test any(!iszero, leaf.state[1])                                                   # Adam's first moment came back from the device
        @test isapprox(leaf.state[3][1], 0.9f0^9; rtol = 1.0f-4)                            # betat = beta^(t + 1) after 8 steps
        envB, _ = mk()
        train!(agent, envB, alg, 16 * 16)                                                   # a NEW handle: the moments arrive with the TrainState
        @test isapprox(agent.train_state.optimizer_state.actor_head.layer_1.weight.state[3][1], 0.9f0^17; rtol = 1.0f-4)
    end

    @testset "save / load round trip (test_ppo_integration.jl:42-83)" begin
        env = DeviceParallelEnv(:CartPole, 16)
        alg = PPO(; n_steps = 32, batch_size = 128, epochs = 2)
        agent = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
        train!(agent, env, alg, 2 * 32 * 16)
        dir = mktempdir(); path = joinpath(dir, "agent")
        DRiL.save_policy_params_and_state(agent, path)
        fresh = Agent(ActorCriticLayer(observation_space(env), action_space(env)), alg; verbose = 0)
        DRiL.load_policy_params_and_state!(fresh, alg, path)
        @test fresh.train_state.parameters.actor_head.layer_1.weight == agent.train_state.parameters.actor_head.layer_1.weight
        obs = observe(env)
        @test DRiL.predict_actions(agent, obs; deterministic = true) == DRiL.predict_actions(fresh, obs; deterministic = true)
    end
end

/*
 * dril_hip.h — C ABI of libdril_hip.so: the MI355X (gfx950) implementation of
 * DRiL.jl's vectorised rollout-collection + PPO-update hot path.
 *
 * The reference (KristianHolme/DRiL.jl) is pure Julia and has no FFI; the seam
 * this library plugs into is Julia dispatch.  Every entry point
 * below names the reference function it stands in for (paths relative to the
 * reference checkout).  A Julia `ccall` shim (dril.jl_amd/julia/DRiLHIP.jl) and
 * a Python ctypes mirror (dril.jl_amd/host.py) both bind exactly these symbols.
 *
 * Conventions
 *   - every function returns int32_t status, 0 == DRIL_OK; no C++ exception
 *     crosses the ABI; dril_last_error() returns the message of the last failure
 *   - the library owns all device memory and the handle; the CALLER owns every
 *     host pointer and must keep it alive for the duration of the call only
 *   - a handle is not thread-safe; different handles may be used concurrently
 *   - all calls are synchronous at return (the handle's HIP stream is drained)
 *     unless documented otherwise
 *   - arrays use the reference's memory layout: observations (D x n) column-major
 *     (each observation's D floats contiguous, src/spaces.jl:259), weights
 *     (out x in) column-major exactly as Lux.Dense stores them
 *   - the device rollout buffer is TIME-MAJOR: flat index n = t * n_envs + env
 *     (the reference buffer is trajectory-major in completion order,
 *     src/buffers/rollout_buffer.jl:70-80; DESIGN.md §3 gives the index map)
 */
#ifndef DRIL_HIP_H
#define DRIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRIL_ABI_VERSION 2u   /* 2: n_hidden / hidden[] / activation (any-depth MLPs, relu) */

typedef struct dril_handle dril_handle;

enum dril_status {
    DRIL_OK = 0,
    DRIL_ERR_INVALID_ARG = 1,
    DRIL_ERR_HIP = 2,
    DRIL_ERR_RCCL = 3,
    DRIL_ERR_NAN_IN_GRADS = 4, /* mirrors `@assert !nested_has_nan(grads)` src/algorithms/ppo.jl:213-214 */
    DRIL_ERR_NOT_INITIALISED = 5,
    DRIL_ERR_UNSUPPORTED = 6
};

enum dril_env_kind {
    DRIL_ENV_CARTPOLE = 0, /* CartPole-v1: D=4, Discrete(2)            */
    DRIL_ENV_PENDULUM = 1, /* Pendulum-v1: D=3, Box(-2,2) 1-dim action */
    /* ScalingWrapperEnv(PendulumEnv()) (src/environment_wrappers/scalingWrapperEnv.jl:15-49): every sub-env is wrapped, so the agent sees
     * observation_space = action_space = Box(-1, 1): observe returns (obs - low) * 2/(high - low) - 1 (:71-74,93-98) and act! maps the
     * action back with (a + 1) / (2/(high - low)) + low (:76-79,110-113) before the physics; the affine maps are fused into the env kernels */
    DRIL_ENV_PENDULUM_SCALED = 2,
    DRIL_ENV_MOUNTAINCAR = 3,            /* MountainCar-v0: D=2 (position, velocity), Discrete(3), reward -1/step, goal at 0.5, limit 200 */
    DRIL_ENV_MOUNTAINCAR_CONTINUOUS = 4, /* MountainCarContinuous-v0: D=2, Box(-1,1), reward 100 at the goal (0.45) - 0.1 a^2, limit 999 */
    /* any AbstractParallelEnv that lives on the HOST (the caller's own Julia envs, interfaces/environments.jl:21-39): observations come in and
     * actions go out once per env step (dril_ext_act / dril_ext_record / dril_ext_finish below), everything else of the path — policy forward,
     * sampling, rollout buffer, bootstrap values, GAE, the PPO update — runs on the device.  Spaces are given by ext_obs_dim / ext_action_dim /
     * ext_discrete; any hidden_dims = [h1, h2] up to 1024.  The env verbs (dril_env_*), dril_collect_rollout, dril_train, dril_evaluate_agent
     * and the wrappers fused into the env kernels (norm_*, monitor_window) belong to the device envs and return DRIL_ERR_UNSUPPORTED here */
    DRIL_ENV_EXTERNAL = 5,
    /* Acrobot-v1 (Gymnasium "book" dynamics, one RK4 step of 0.2 s per env step): D=6 (cos t1, sin t1, cos t2, sin t2, w1, w2), Discrete(3) torques
     * -1/0/+1, reward -1 per step (0 on reaching the height), limit 500.  A device env like the others; hidden_dims [64,64] run the fused kernels (four
     * first-layer k-steps for its six observation dims; the update on the pair / persistent f16-piece kernels like the other envs, dW1 through a third piece image),
     * hidden 128 / 256 the wide fused kernels, any other hidden_dims the generic kernels */
    DRIL_ENV_ACROBOT = 6,
    /* ScalingWrapperEnv(MountainCarContinuousEnv()): observations Box((-1.2, -0.07), (0.6, 0.07)) scaled to Box(-1, 1), actions Box(-1, 1) mapped back by the same
     * affine formulas (scalingWrapperEnv.jl:71-79); every kernel that does not touch the simulator is shared with DRIL_ENV_MOUNTAINCAR_CONTINUOUS */
    DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED = 7
};

/* ids for dril_buffer_copy_out / dril_buffer_copy_in (fields of RolloutBuffer,
 * src/buffers/buffer_types.jl:3-15, plus the per-step flags the reference keeps
 * in Trajectory, buffer_types.jl:17-26) */
enum dril_buffer_id {
    DRIL_BUF_OBSERVATIONS = 0, /* f32  (D, N)                                  */
    DRIL_BUF_ACTIONS = 1,      /* i32 (1, N) discrete | f32 (A, N) continuous   */
    DRIL_BUF_REWARDS = 2,      /* f32 (N)                                       */
    DRIL_BUF_ADVANTAGES = 3,   /* f32 (N)                                       */
    DRIL_BUF_RETURNS = 4,      /* f32 (N)                                       */
    DRIL_BUF_LOGPROBS = 5,     /* f32 (N)                                       */
    DRIL_BUF_VALUES = 6,       /* f32 (N)                                       */
    DRIL_BUF_FLAGS = 7,        /* u8 (N): bit0 terminated, bit1 truncated       */
    DRIL_BUF_BOOTSTRAP = 8,    /* f32 (N): V(terminal_observation), valid where truncated */
    DRIL_BUF_LAST_VALUES = 9   /* f32 (n_envs): V(obs after the last step)      */
};

/* kernels whose HIP-event timings dril_profile_get reports */
enum dril_kernel_id {
    DRIL_K_ROLLOUT = 0,
    DRIL_K_GAE = 1,
    DRIL_K_ADV_MOMENTS = 2,
    DRIL_K_PPO_GRAD = 3,
    DRIL_K_GRAD_REDUCE = 4,
    DRIL_K_ADAM = 5,
    DRIL_K_ALLREDUCE = 6,
    DRIL_K_PACK_RECORDS = 7,   /* pack_records_kernel: the rollout buffer's SoA fields -> one 32-byte record per sample and net, once per update */
    DRIL_K_EXPLAINED_VAR = 8,  /* explained_var_kernel: the four sums of ppo.jl:256 over the whole buffer */
    DRIL_K_COUNT = 9           /* grows with the library: loop to dril_kernel_count() */
};

/* Plain-C mirror of `PPO` (src/algorithms/ppo.jl:25-40), the layer kwargs
 * (src/layers/layer_constructors.jl:3-11,55-56), `NormalizeWrapperEnv` kwargs
 * (src/environment_wrappers/normalizeWrapperEnv.jl:71-80) and the env ctor
 * kwargs used by the reference's benchmarks (benchmark/bench_utils.jl:14,20). */
typedef struct dril_config {
    uint32_t abi_version;      /* must be DRIL_ABI_VERSION */
    int32_t env_kind;          /* enum dril_env_kind */
    int32_t n_envs;            /* E on THIS rank */
    int32_t n_steps;           /* T (PPO.n_steps) */
    int32_t hidden1, hidden2;  /* hidden_dims = [hidden1, hidden2], 1..1024 each, when n_hidden == 0 (below): [64,64], [128,128], [256,256] with tanh run the
                                * fused kernels, every other shape (and DRIL_ENV_EXTERNAL) the generic layer-by-layer kernels */
    int32_t episode_len;       /* max_steps kwarg: 500 CartPole-v1, 200 Pendulum-v1 */
    int32_t fixed_length_episodes; /* 1: termination disabled (synthetic bench episodes) */
    int32_t action_start;      /* Discrete(n, start): src/spaces.jl:157-164 */

    float gamma, gae_lambda, clip_range;
    float clip_range_vf;  int32_t has_clip_range_vf;   /* Union{T,Nothing} */
    float ent_coef, vf_coef;
    float max_grad_norm;  int32_t has_max_grad_norm;
    float target_kl;      int32_t has_target_kl;
    int32_t normalize_advantage;
    int64_t batch_size;        /* GLOBAL minibatch size B (over all ranks) */
    int32_t epochs;
    float learning_rate;
    float adam_beta1, adam_beta2, adam_eps; /* Optimisers.Adam defaults + eps=1e-5, ppo.jl:64-66 */
    float log_std_init;

    int32_t norm_obs, norm_reward, norm_training; /* NormalizeWrapperEnv; all 0 = no wrapper */
    float clip_obs, clip_reward, norm_gamma, norm_epsilon;

    uint64_t seed;             /* env i (0-based, global index) is seeded seed + i, wrapper_utils.jl:39-44 */
    int32_t device;            /* HIP device ordinal */
    int32_t rank, world_size;  /* data-parallel position; global env index = rank*n_envs + local */
    int32_t profile_events;    /* k >= 1: bracket hot kernels with HIP events (dril_profile_get); the per-optimiser-step kernels at every k-th launch */
    int32_t monitor_window;    /* MonitorWrapperEnv(env, stats_window): > 0 tracks episode returns/lengths (monitorWrapperEnv.jl:15-24); 0 = no wrapper */
    /* DRIL_ENV_EXTERNAL only (ignored otherwise): observation_space = Box of ext_obs_dim floats (<= 1024); action_space =
     * Discrete(ext_action_dim, action_start) when ext_discrete, else Box(ext_action_low, ext_action_high) of ext_action_dim floats (<= 64);
     * the bounds feed ClampAdapter (default_adapters.jl:4-11); low >= high = no clamp */
    int32_t ext_obs_dim, ext_action_dim, ext_discrete;
    float ext_action_low, ext_action_high;
    /* ActorCriticLayer(...; hidden_dims, activation) in full (src/layers/layer_constructors.jl:6-10,55-56; get_mlp layer_helpers.jl:27-57 builds
     * Dense(in => h_1, act), ..., Dense(h_{n-1} => h_n, act), Dense(h_n => out)): n_hidden = length(hidden_dims) in 1..4 with hidden[0..n_hidden-1]
     * (1..1024 each), or 0 = the two-layer form hidden1 / hidden2 above.  activation: 0 tanh (the reference's default), 1 relu, 2 sigmoid, 3 elu (alpha 1), 4 leakyrelu (0.01), 5 softplus, 6 gelu (the tanh form), 7 swish (NNlib's definitions; the generic kernels).  Every shape other
     * than two equal tanh layers of 64 / 128 / 256 runs the generic kernels.  Parameter layout per net: {W_1 b_1 ... W_{n+1} b_{n+1}} */
    int32_t n_hidden, hidden[4], activation;
    int32_t reserved[1];
} dril_config;

/* per-iteration means returned by dril_ppo_update; field names follow the
 * `learn_stats` NamedTuple, src/algorithms/ppo.jl:301-312 */
typedef struct dril_ppo_stats {
    float entropy_loss, policy_loss, value_loss, approx_kl_div, clip_fraction;
    float loss, grad_norm, explained_variance, entropy, ratio_first; /* ratio of epoch1/batch1, ppo.jl:209-212 */
    int32_t n_updates;         /* optimiser steps actually applied */
    int32_t early_stopped;     /* 1 if target_kl stopped the loops, ppo.jl:235-238 */
    int32_t nan_or_inf;        /* 1 if a gradient contained NaN/Inf (status is DRIL_ERR_NAN_IN_GRADS too) */
    int32_t f32_path;          /* which kernels produced THIS update (was `reserved`): 0 the default ones; 1 redone on the exact-f32 kernels after an f16-piece kernel
                                * left f16's range; 2 run directly on the exact-f32 kernels (latched after repeated redos, or a W2 entry out of range) */
} dril_ppo_stats;

/* The fused kernels' default arithmetic (fp32-equivalent products from two f16 pieces per operand) holds fp32's PRECISION but f16's RANGE.  The reference asks no range of
 * its user (ppo.jl:213-214 only asserts finiteness), so leaving it is handled inside the library and merely COUNTED here:
 *   retries              updates taken back and redone on the exact-f32 kernels because an f16-piece step met a non-finite gradient (2x the update's time, each)
 *   direct_updates       updates run on the exact-f32 kernels at once: after 2 consecutive retries the next 16 updates (then one update probes f16 again), or max|W2| >= 350
 *   persistent_fallbacks updates of the two-workgroup persistent kernel (batch_size <= 64) redone on the per-step kernels because its workgroups were not co-resident
 *   latch_updates_left   updates the latch still covers; forward_exact_f32: 1 while rollout / policy forwards run the f32-MFMA kernels (max_abs_w2 >= 350 or DRIL_GRAD_VARIANT=0) */
typedef struct dril_f32_fallback {
    int64_t retries, direct_updates, persistent_fallbacks;
    int32_t latch_updates_left, forward_exact_f32;
    float max_abs_w2; int32_t reserved;
} dril_f32_fallback;

/* fill cfg with the reference defaults: PPO() ppo.jl:26-39, hidden_dims [64,64]
 * layer_constructors.jl:55, NormalizeWrapperEnv kwargs normalizeWrapperEnv.jl:71-80 (disabled) */
int32_t dril_config_default(dril_config* cfg, int32_t env_kind);

/* ---- lifetime ------------------------------------------------------------ */
/* RolloutBuffer(...) ppo.jl:112-115 + Agent(layer, alg) ppo.jl:42-62 (device state only) */
int32_t dril_create(const dril_config* cfg, dril_handle** out);
int32_t dril_destroy(dril_handle* h);
/* message of the last failing call on h (or of the last failing create when h == NULL) */
const char* dril_last_error(const dril_handle* h);
/* drains the handle's stream */
int32_t dril_synchronize(dril_handle* h);

/* ---- shapes ---------------------------------------------------------------*/
int32_t dril_obs_dim(const dril_handle* h);     /* D  */
int32_t dril_action_dim(const dril_handle* h);  /* A: n for Discrete, dims for Box */
int32_t dril_is_discrete(const dril_handle* h);
int64_t dril_param_count(const dril_handle* h); /* Lux.parameterlength, layer_lux.jl */

/* ---- parameters: agent.train_state.parameters <-> flat f32 -----------------
 * layout: actor_head {W1(H1xD) b1 W2(H2xH1) b2 W3(AoutxH2) b3}, critic_head {W1 b1 W2 b2 W3(1xH2) b3},
 * then log_std(A) for Box actions (layer_lux.jl:4-39); every W column-major (out x in).  With n_hidden layers: {W_1 b_1 ... W_{n+1} b_{n+1}} per head */
int32_t dril_set_params(dril_handle* h, const float* flat, size_t n);
int32_t dril_get_params(dril_handle* h, float* flat, size_t n);
/* fresh optimiser state (load_policy_params_and_state! rebuilds Adam, ppo.jl:77-94) */
int32_t dril_reset_optimizer(dril_handle* h);
/* the optimiser state itself — what Lux.Training.TrainState carries as optimizer_state between train! calls (ppo.jl:52-53,239; Optimisers.Adam leaf state
 * (mt, vt, betat)): first and second moments in the parameter layout of dril_get_params, the running products (beta1^t, beta2^t) and the number of applied
 * steps.  The handle keeps this state between dril_train / dril_ppo_update calls; get / set move it with a TrainState from one handle to another (another
 * env, a re-created handle).  n = dril_param_count; beta_powers = 2 floats. */
int32_t dril_get_optimizer_state(dril_handle* h, float* m, float* v, size_t n, float* beta_powers, int64_t* steps);
int32_t dril_set_optimizer_state(dril_handle* h, const float* m, const float* v, size_t n, const float* beta_powers, int64_t steps);
/* Optimisers.adjust!(train_state, lr) ppo.jl:155-156 */
int32_t dril_set_learning_rate(dril_handle* h, float lr);

/* ---- env verbs (MultiThreadedParallelEnv, src/environment_wrappers/multithreadedParallelEnv.jl) */
/* Random.seed!(env, seed) wrapper_utils.jl:39-44 followed by reset!(env) :12-17 */
int32_t dril_env_reset(dril_handle* h, uint64_t seed);
/* observe(env) :19-25 (NormalizeWrapperEnv.observe normalizeWrapperEnv.jl:123-137 when enabled:
 * updates obs statistics when update_stats != 0); host_obs is (D x E) column-major */
int32_t dril_env_observe(dril_handle* h, float* host_obs, int32_t update_stats);
/* act!(env, actions) :47-74 with auto-reset; actions i32(E) | f32(A x E) are ENV-space
 * (already passed through to_env, src/adapters/default_adapters.jl:4-11,34-38);
 * terminal_obs (D x E) is written only for truncated envs; any out pointer may be NULL */
int32_t dril_env_step(dril_handle* h, const void* host_actions, float* rewards, uint8_t* terminated,
                      uint8_t* truncated, float* terminal_obs);
/* raw simulator state: CartPole (x, x_dot, theta, theta_dot), Pendulum (theta, theta_dot), + step counter */
int32_t dril_env_get_state(dril_handle* h, float* state /* S x E */, int32_t* step_count /* E */);
int32_t dril_env_set_state(dril_handle* h, const float* state, const int32_t* step_count);
/* RunningMeanStd fields, normalizeWrapperEnv.jl:8-19 (save/load :261-297) */
int32_t dril_norm_get_stats(dril_handle* h, float* obs_mean, float* obs_var, int64_t* obs_count,
                            float* ret_mean, float* ret_var, int64_t* ret_count);
int32_t dril_norm_set_stats(dril_handle* h, const float* obs_mean, const float* obs_var, int64_t obs_count,
                            float ret_mean, float ret_var, int64_t ret_count);
/* get_original_obs(env) / get_original_rewards(env) (normalizeWrapperEnv.jl:220-222): the un-normalised observations (D x E) of the last
 * observe and the un-normalised rewards (E) of the last act! through the step-granular verbs; either pointer may be NULL */
int32_t dril_norm_get_original(dril_handle* h, float* obs, float* rewards);

/* MonitorWrapperEnv: mean return / length over the last `monitor_window` finished episodes (log_stats, monitorWrapperEnv.jl:64-70)
 * and the number of episodes currently in the window; rewards are the RAW env rewards (the monitor sits inside the normaliser) */
int32_t dril_monitor_get_stats(dril_handle* h, float* ep_rew_mean, float* ep_len_mean, int32_t* n_episodes);

/* ---- policy (src/layers/layer_forward.jl, layer_methods.jl) on host batches -- */
/* layer(obs, ps, st) -> (actions, values, logprobs): layer_forward.jl:3-13 / :30-39.
 * noise: f64(B) uniforms for Categorical.rand (categorical.jl:47-52) or f32(A x B) normals for
 * DiagGaussian.rand (diagGaussian.jl:13-17); NULL = draw from the handle's Philox stream.
 * actions are the RAW policy actions (pre-adapter, trajectory.jl:48) */
int32_t dril_policy_forward(dril_handle* h, const float* obs, int64_t batch, const void* noise,
                            void* actions, float* values, float* logprobs);
/* evaluate_actions(layer, obs, actions, ps, st): layer_methods.jl:28-55 */
int32_t dril_evaluate_actions(dril_handle* h, const float* obs, const void* actions, int64_t batch,
                              float* values, float* logprobs, float* entropy);
/* predict_actions(layer, obs, ps, st; deterministic, rng): layer_methods.jl:3-26 — mode(d) when deterministic (argmax of the Categorical,
 * categorical.jl:42-44; the mean of the DiagGaussian, diagGaussian.jl:45-47), else rand(d) with `noise` as in dril_policy_forward.
 * actions are RAW policy actions (evaluate_agent passes them through to_env, evaluation.jl:92-93) */
int32_t dril_predict_actions(dril_handle* h, const float* obs, int64_t batch, int32_t deterministic, const void* noise, void* actions);
/* predict_values(layer, obs, ps, st): layer_methods.jl:57-61 */
int32_t dril_predict_values(dril_handle* h, const float* obs, int64_t batch, float* values);

/* ---- rollout over HOST envs (DRIL_ENV_EXTERNAL): collect_trajectories, trajectory.jl:22-78, one call pair per env step ------------
 * for step in 1:n_steps
 *     obs = observe(env)                                   # caller
 *     dril_ext_act(h, obs, raw, env_actions)               # get_action_and_values :41 + to_env :42; obs/actions/values/logprobs -> buffer :46-51
 *     rewards, terminateds, truncateds, infos = act!(env, env_actions)     # caller
 *     dril_ext_record(h, rewards, terminateds, truncateds, terminal_obs)   # rewards/flags -> buffer; V(terminal_observation) for truncated envs :57-61
 * end
 * dril_ext_finish(h, observe(env))                         # V(new_obs) for unfinished trajectories :65-70, compute_advantages!, returns
 * obs / terminal_obs / last_obs are (D x E) column-major host arrays; raw_actions (stored, pre-adapter) and env_actions (ClampAdapter /
 * DiscreteAdapter applied) are i32(E) | f32(A x E), either may be NULL; terminal_obs may be NULL when no env was truncated; only the
 * columns of truncated envs are read.  dril_debug_set_noise injects the sampling noise of the next n_steps dril_ext_act calls. */
int32_t dril_ext_act(dril_handle* h, const float* obs, void* raw_actions, void* env_actions);
int32_t dril_ext_record(dril_handle* h, const float* rewards, const uint8_t* terminated, const uint8_t* truncated, const float* terminal_obs);
int32_t dril_ext_finish(dril_handle* h, const float* last_obs);
/* env steps recorded since the last dril_ext_finish (0 .. n_steps) */
int32_t dril_ext_steps(const dril_handle* h);

/* ---- rollout --------------------------------------------------------------- */
/* collect_rollout!(buffer, agent, alg, env): rollout_buffer.jl:46-90 =
 * collect_trajectories trajectory.jl:22-78 + compute_advantages! :80-102 + returns :87.
 * fps = steps / wall time of the collection part, rollout_buffer.jl:60-64 */
int32_t dril_collect_rollout(dril_handle* h, double* fps);
/* injected sampling noise for the NEXT dril_collect_rollout call only: f64 (E x T) uniforms
 * [t*E+e] (discrete) or f32 (A x E x T) normals; NULL clears */
int32_t dril_debug_set_noise(dril_handle* h, const void* noise, size_t count);
int32_t dril_buffer_copy_out(dril_handle* h, int32_t which, void* host, size_t bytes);
int32_t dril_buffer_copy_in(dril_handle* h, int32_t which, const void* host, size_t bytes);
/* compute_advantages! over the device buffer as it stands (rewards/values/flags/bootstrap/last_values) */
int32_t dril_compute_gae(dril_handle* h);
/* stand-alone GAE on caller arrays, time-major [t*E+e]; no handle state is used or changed */
int32_t dril_gae(int32_t n_envs, int32_t n_steps, float gamma, float gae_lambda, const float* rewards,
                 const float* values, const uint8_t* flags, const float* bootstrap, const float* last_values,
                 float* advantages, float* returns);

/* ---- PPO update -------------------------------------------------------------- */
/* the epoch x minibatch loop of train!, ppo.jl:188-264 (DataLoader shuffle, loss :365-407 + gradient,
 * NaN asserts :213-214, nested_norm/nested_scale! :216-232, target_kl :235-238, Adam :239,
 * explained_variance :256, per-iteration means :257-264) */
int32_t dril_ppo_update(dril_handle* h, dril_ppo_stats* out);
/* see dril_f32_fallback above; dril_f32_retries = its `retries` alone (-1 for a null handle).  A healthy run on normalised data shows 0 / 0 */
int64_t dril_f32_retries(const dril_handle* h);
int32_t dril_f32_fallback_info(const dril_handle* h, dril_f32_fallback* out);
/* injected DataLoader order: perm[e*N + p] = 0-based buffer index at position p of epoch e
 * (ppo.jl:188-195); NULL = device-generated pseudo-random bijection per epoch */
int32_t dril_debug_set_permutation(dril_handle* h, const int64_t* perm, size_t count);
/* (alg::PPO)(layer, ps, st, batch) ppo.jl:365-407 and its gradient (Lux.Training.compute_gradients,
 * ppo.jl:207) on a caller minibatch; stats7 = policy_loss, value_loss, entropy_loss, clip_fraction,
 * approx_kl_div, entropy, ratio; grads has dril_param_count entries, same layout as the params.
 * normalises advantages per ppo.jl:350-363 when cfg.normalize_advantage */
int32_t dril_ppo_loss_grad(dril_handle* h, const float* obs, const void* actions, const float* advantages,
                           const float* returns, const float* old_logprobs, const float* old_values,
                           int64_t batch, float* loss, float* stats7, float* grads);
/* one optimiser step from caller gradients: nested_norm, nested_scale!, Adam (ppo.jl:216-239);
 * returns the pre-clip norm */
int32_t dril_apply_gradients(dril_handle* h, const float* grads, size_t n, float* grad_norm);

/* ---- evaluate_agent (src/evaluation.jl:54-143) --------------------------------------------- */
typedef struct dril_eval_stats {
    double mean_reward, std_reward, mean_length, std_length;   /* Julia mean / std (corrected) over the collected episodes */
    int32_t n_episodes, n_steps;                               /* episodes collected, env steps taken */
} dril_eval_stats;
/* reset!(env); then predict_actions(agent, obs; deterministic) -> act! -> observe until the first n_eval_episodes episodes have
 * finished, taken in (step, env) order (evaluation.jl:90-124).  deterministic: mode(d) — argmax for Categorical (categorical.jl:42-44),
 * the mean for DiagGaussian (diagGaussian.jl:45-47).  With MonitorWrapperEnv on (cfg.monitor_window > 0) episode returns use the RAW
 * rewards, like infos[i]["episode"]["r"]; otherwise the rewards as the wrappers deliver them.  episode_rewards / episode_lengths
 * (n_eval_episodes entries each) may be NULL. */
int32_t dril_evaluate_agent(dril_handle* h, int32_t n_eval_episodes, int32_t deterministic, dril_eval_stats* out,
                            float* episode_rewards, int32_t* episode_lengths);

/* ---- train! ------------------------------------------------------------------ */
/* iterations = max_steps / (T*E*world) of {set lr, collect_rollout!, ppo update}: ppo.jl:154-298.
 * stats / fps arrays need `iterations` entries (may be NULL) */
int32_t dril_train(dril_handle* h, int64_t max_steps, dril_ppo_stats* stats, double* fps, int32_t* iterations_done);

/* ---- multi-GPU (new; the reference is single-process) ------------------------- */
/* 128-byte ncclUniqueId from rank 0, distributed to the other ranks by the host */
int32_t dril_comm_unique_id(uint8_t id[128]);
/* RCCL communicator over cfg.world_size ranks; gradients and loss statistics are summed with one
 * ncclAllReduce per optimiser step on the handle's stream */
int32_t dril_comm_init(dril_handle* h, const uint8_t id[128]);
/* ranks the handle's communicator spans: ncclCommCount of the RCCL communicator (what RCCL itself saw, not cfg.world_size), the group
 * size of a loopback communicator, 1 without a communicator, -1 on error */
int32_t dril_comm_ranks(dril_handle* h);
/* all-reduces this handle has issued since dril_create (every call site counts: advantage moments, [grads || 8 sums], the per-epoch
 * moment table, NormalizeWrapperEnv's batch moments, the explained-variance sums) */
int64_t dril_comm_allreduce_calls(const dril_handle* h);
/* which device the handle lives on, for the banner every rank of a multi-GPU job prints before dril_comm_init and for RCCL failure messages:
 * "device <ordinal> of <visible count> visible: <name> <arch>, <CUs> CUs, PCI <bus id>, HIP_VISIBLE_DEVICES=... ROCR_VISIBLE_DEVICES=..." (new; no reference counterpart).
 * A failing ncclCommInitRank / ncclAllReduce puts ncclGetErrorString, ncclGetLastError and this line into dril_last_error. */
const char* dril_device_info(const dril_handle* h);
/* DEBUG / TEST: join n handles of THIS process, all on ONE device, with cfg.world_size == n and ranks 0..n-1, into a loopback
 * communicator (RCCL refuses two ranks per device).  Every all-reduce call site, count and dtype of the data-parallel path is unchanged;
 * the transport is an in-process rendezvous plus one kernel that sums the ranks' device buffers in rank order and writes the sum back to
 * all of them.  Each handle must then be driven from its own host thread, all making the same sequence of calls; a rank that waits
 * 120 s for the others fails with DRIL_ERR_RCCL.  (new: the reference has no distributed code) */
int32_t dril_debug_comm_loopback(dril_handle** handles, int32_t n);

/* ---- measurement ---------------------------------------------------------------- */
/* accumulated HIP-event time and count of the BRACKETED launches of one kernel class since the last reset (total_ms / launches = average launch);
 * dril_profile_launches: all launches of the class in the same window (= launches unless cfg.profile_events > 1 thinned the per-step classes) */
int32_t dril_profile_get(dril_handle* h, int32_t kernel_id, double* total_ms, int64_t* launches);
int32_t dril_profile_launches(dril_handle* h, int32_t kernel_id, int64_t* all_launches);
int32_t dril_profile_reset(dril_handle* h);
const char* dril_kernel_name(int32_t kernel_id);
int32_t dril_kernel_count(void);   /* DRIL_K_COUNT of the loaded library */
/* which gradient kernel the handle's LAST optimiser step ran and the arithmetic it computes in ("<kernel>: <arithmetic>"; "none yet" before the
 * first step): hidden [64,64] runs ppo_grad_pair_kernel (f16 matrix cores, fp32-equivalent two-piece operand split) on large minibatches and the f32-MFMA
 * ppo_grad_kernel on small ones, [128,128] and [256,256] ppo_grad_wide_split_kernel, everything else the generic path (docs/kernels/ppo_kernels.md;
 * DRIL_GRAD_VARIANT overrides the choice) */
const char* dril_grad_kernel_info(const dril_handle* h);
const char* dril_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DRIL_HIP_H */

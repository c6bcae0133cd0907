/*
 * dril_sac.h — C ABI of the off-policy (SAC) path of libdril_hip.so (BASELINE.json configs[4]:
 * "SAC on Pendulum-v1, n_envs=4096, device ReplayBuffer + twin-Q/actor HIP kernels").
 *
 * Stands in for, in the reference checkout (KristianHolme/DRiL.jl):
 *   src/algorithms/sac.jl                       SAC, SACLayer, losses, update!, train!
 *   src/buffers/replay_buffer.jl                ReplayBuffer, get_data_loader
 *   src/buffers/off_policy_collection.jl        collect_trajectories / collect_rollout! (off-policy)
 *   src/DRiLDistributions/squashedDiagGaussian.jl
 *   src/layers/layer_forward.jl:15-28,75-87,118-125   ContinuousActorCriticLayer{QCritic}
 *   src/utils/optimization_utils.jl:3-58        polyak_update!, merge_params
 *
 * Conventions are those of dril_hip.h: int32 status (enum dril_status), the library owns device memory and the
 * handle, the caller owns host pointers for the duration of the call, calls are synchronous at return, arrays are
 * (features x batch) column-major, weights (out x in) column-major as Lux.Dense stores them.
 *
 * The SAC handle is its own object (the on-policy handle of dril_hip.h holds a RolloutBuffer and PPO state that
 * SAC has no use for); both live in the same shared library and share the device env kernels.
 */
#ifndef DRIL_SAC_H
#define DRIL_SAC_H

#include "dril_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DRIL_SAC_ABI_VERSION 1u

typedef struct dril_sac_handle dril_sac_handle;

/* fields of ReplayBuffer (src/buffers/buffer_types.jl:28-41) for dril_sac_replay_copy_out; logical index 0 = oldest
 * element; within one collect call the device order is time-major (step, env) where the reference pushes whole
 * trajectories in completion order (replay_buffer.jl:98-114) — irrelevant to uniform sampling, documented for
 * anyone comparing flat arrays */
enum dril_replay_id {
    DRIL_RB_OBSERVATIONS = 0,      /* f32 (D, n)                                                          */
    DRIL_RB_ACTIONS = 1,           /* f32 (A, n): the UNPROCESSED policy action (off_policy_collection.jl:72) */
    DRIL_RB_REWARDS = 2,           /* f32 (n)                                                             */
    DRIL_RB_TERMINATED = 3,        /* u8  (n)                                                             */
    DRIL_RB_TRUNCATED = 4,         /* u8  (n): trajectory cut here by the env's time limit                */
    DRIL_RB_NEXT_OBSERVATIONS = 5  /* f32 (D, n): what get_data_loader resolves per sample, replay_buffer.jl:127-146
                                      (next step's observation | truncated_observation); rows of TERMINATED steps hold the
                                      terminal observation where the reference substitutes NaN — the loss never reads them */
};

/* Plain-C mirror of `SAC` (src/algorithms/sac.jl:25-36), the SACLayer kwargs (:72-85), the entropy-coefficient
 * types (src/interfaces/entropy.jl) and the env ctor kwargs */
typedef struct dril_sac_config {
    uint32_t abi_version;       /* DRIL_SAC_ABI_VERSION */
    int32_t env_kind;           /* Box action space required (sac.jl:74): DRIL_ENV_PENDULUM[_SCALED], DRIL_ENV_MOUNTAINCAR_CONTINUOUS, or DRIL_ENV_EXTERNAL (host envs: ext_* below) */
    int32_t n_envs;
    int32_t episode_len;        /* max_steps kwarg: 200 Pendulum-v1 */
    int32_t hidden1, hidden2;   /* SACLayer hidden_dims, default [512, 512] (sac.jl:76); multiples of 32 */
    int32_t activation;         /* 0 tanh, 1 relu (SACLayer default, sac.jl:77) */
    int64_t buffer_capacity;    /* :27 */
    int32_t start_steps;        /* :28 */
    int32_t batch_size;         /* :29 */
    float tau, gamma;           /* :30-31 */
    int32_t train_freq;         /* :32 */
    int32_t gradient_steps;     /* :33, -1 = train_freq * n_envs (get_gradient_steps :59-65) */
    int32_t target_update_interval; /* :35 */
    int32_t auto_ent_coef;      /* 1 AutoEntropyCoefficient, 0 FixedEntropyCoefficient (entropy.jl:17-24) */
    float ent_coef_init;        /* initial_value (1.0) | coef */
    int32_t auto_target_entropy;/* 1: -prod(size(action_space)) (sac.jl:47-57) */
    float target_entropy;       /* FixedEntropyTarget value */
    float learning_rate;        /* :26; the optimiser is Optimisers.Adam(lr) with library defaults (agent_methods.jl:116-118) */
    float adam_beta1, adam_beta2, adam_eps;   /* 0.9, 0.999, 1e-8 */
    uint64_t seed;              /* env i is seeded seed + i (wrapper_utils.jl:39-44) */
    int32_t device;
    int32_t profile_events;
    /* DRIL_ENV_EXTERNAL only: observation_space = Box of ext_obs_dim floats (<= 1024), action_space = Box(ext_action_low, ext_action_high) of
     * ext_action_dim floats (<= 16; the bounds feed TanhScaleAdapter, default_adapters.jl:13-21, and rand(action_space) of the start phase) */
    int32_t ext_obs_dim, ext_action_dim;
    float ext_action_low, ext_action_high;
    int32_t reserved[4];
} dril_sac_config;

/* NamedTuple returned by update!(agent, alg::SAC, batch), sac.jl:395-403; one per gradient step */
typedef struct dril_sac_stats {
    float actor_loss, critic_loss, entropy_loss, mean_q_values, entropy_coefficient, grad_norm;
    int32_t has_entropy_loss;   /* entropy_loss === nothing for FixedEntropyCoefficient */
    int32_t reserved;
} dril_sac_stats;

/* SAC() sac.jl:25-36, SACLayer kwargs :72-85, AutoEntropyCoefficient() entropy.jl:21-24 */
int32_t dril_sac_config_default(dril_sac_config* cfg, int32_t env_kind);

/* ---- lifetime: ReplayBuffer(obs_space, act_space, capacity) sac.jl:411 + Agent(layer, alg::SAC) sac.jl:160-188 */
int32_t dril_sac_create(const dril_sac_config* cfg, dril_sac_handle** out);
int32_t dril_sac_destroy(dril_sac_handle* h);
const char* dril_sac_last_error(const dril_sac_handle* h);

int32_t dril_sac_obs_dim(const dril_sac_handle* h);
int32_t dril_sac_action_dim(const dril_sac_handle* h);
/* Lux.parameterlength of ContinuousActorCriticLayer{QCritic}: actor + n_critics(2) Q nets + log_std */
int64_t dril_sac_param_count(const dril_sac_handle* h);
/* parameters of ONE Q network: (D+A)*H1+H1 + H1*H2+H2 + H2+1 */
int64_t dril_sac_q_param_count(const dril_sac_handle* h);

/* ---- parameters -----------------------------------------------------------------------------------------------
 * flat layout: actor_head {W1(H1xD) b1 W2(H2xH1) b2 W3(AxH2) b3}, critic_head.layer_1 {W1(H1x(D+A)) b1 W2 b2 W3(1xH2) b3},
 * critic_head.layer_2 {...} (Lux.Parallel(vcat, mlp, mlp), layer_helpers.jl:100-112), log_std(A).
 * dril_sac_set_params also re-initialises the target networks from the critics, as Agent(layer, alg::SAC) does
 * (copy_critic_parameters, sac.jl:172,191-197) */
int32_t dril_sac_set_params(dril_sac_handle* h, const float* flat, size_t n);
int32_t dril_sac_get_params(dril_sac_handle* h, float* flat, size_t n);
/* agent.aux.Q_target_parameters: {layer_1, layer_2}, 2 * q_param_count floats */
int32_t dril_sac_get_target_params(dril_sac_handle* h, float* flat, size_t n);
int32_t dril_sac_set_target_params(dril_sac_handle* h, const float* flat, size_t n);
/* agent.aux.ent_train_state.parameters.log_ent_coef[1] (init_entropy_coefficient, sac.jl:207-213) */
int32_t dril_sac_get_log_ent_coef(dril_sac_handle* h, float* value);
int32_t dril_sac_set_log_ent_coef(dril_sac_handle* h, float value);
/* fresh Adam state for the layer and for the entropy coefficient; gradient-update counter back to 0 */
int32_t dril_sac_reset_optimizer(dril_sac_handle* h);

/* ---- env ---------------------------------------------------------------------------------------------------- */
int32_t dril_sac_env_reset(dril_sac_handle* h, uint64_t seed);
int32_t dril_sac_env_observe(dril_sac_handle* h, float* host_obs /* D x E */);

/* ---- layer calls on host batches ------------------------------------------------------------------------------ */
/* action_log_prob(layer, obs, ps, st; rng): layer_methods.jl:66-76 with SquashedDiagGaussian (squashedDiagGaussian.jl:24-46).
 * noise f32 (A x B) standard normals, NULL = draw from the handle's Philox stream; actions are the squashed samples
 * tanh(mean + exp(log_std) * noise) */
int32_t dril_sac_action_log_prob(dril_sac_handle* h, const float* obs, int64_t batch, const float* noise,
                                 float* actions, float* logprobs);
/* predict_actions(agent, obs; deterministic, raw): sac.jl:215-240.  raw actions = rand / mode of the squashed
 * distribution; env actions = to_env(TanhScaleAdapter(), raw, space) = scale_to_space(tanh.(raw), space)
 * (default_adapters.jl:13-21 — the adapter applies tanh again; kept as the reference has it).  Either out pointer may be NULL */
int32_t dril_sac_predict_actions(dril_sac_handle* h, const float* obs, int64_t batch, int32_t deterministic,
                                 const float* noise, float* raw_actions, float* env_actions);
/* predict_values(layer, obs, actions, ps, st): layer_methods.jl:63-67 — q is (2 x B) column-major (vcat of the critics);
 * use_target != 0 evaluates the target networks (merge_params(ps, target_ps), sac.jl:124-127) */
int32_t dril_sac_predict_q(dril_sac_handle* h, const float* obs, const float* actions, int64_t batch,
                           int32_t use_target, float* q);

/* ---- collection: collect_rollout!(buffer, agent, alg, env, n_steps; use_random_actions) off_policy_collection.jl:117-136
 * = collect_trajectories :28-96 + push!(buffer, traj) replay_buffer.jl:98-114.  fps = steps / wall time (:126-128) */
int32_t dril_sac_collect_rollout(dril_sac_handle* h, int32_t n_steps, int32_t use_random_actions, double* fps);
/* injected noise for the NEXT collect call only, f32 [step][env][A]: standard normals for policy actions, uniforms in
 * [0,1) for random actions (rand(rng, act_space) = low + u * (high - low)); NULL clears */
int32_t dril_sac_debug_set_collect_noise(dril_sac_handle* h, const float* noise, size_t count);

/* ---- collection over HOST envs (DRIL_ENV_EXTERNAL): one env step of collect_trajectories (off_policy_collection.jl:28-96) + push! (replay_buffer.jl:98-114)
 *   obs = observe(env);  dril_sac_predict_actions(h, obs, E, 0, noise, raw, env_actions)   (or rand(action_space) during the start phase, :50-53)
 *   rewards, terminateds, truncateds, infos = act!(env, env_actions);  next_obs = observe(env)
 *   dril_sac_ext_push(h, obs, stored_actions, rewards, terminateds, truncateds, next_obs, terminal_obs)
 * stored_actions (A x E) are what the reference stores: the raw squashed policy action, or the env-space random action (:72); the next observation of a
 * truncated env is its terminal_obs column (infos[i]["terminal_observation"], :75-79; may be NULL when no env was truncated).  n_envs transitions per call. */
int32_t dril_sac_ext_push(dril_sac_handle* h, const float* obs, const float* stored_actions, const float* rewards, const uint8_t* terminated,
                          const uint8_t* truncated, const float* next_obs, const float* terminal_obs);

/* ---- replay buffer ---------------------------------------------------------------------------------------------- */
int64_t dril_sac_replay_size(const dril_sac_handle* h);       /* length(buffer) */
int64_t dril_sac_replay_capacity(const dril_sac_handle* h);
int32_t dril_sac_replay_copy_out(dril_sac_handle* h, int32_t which, void* host, size_t bytes);
/* empty!(buffer) followed by `count` pushes of caller transitions (tests) */
int32_t dril_sac_replay_fill(dril_sac_handle* h, int64_t count, const float* obs, const float* actions,
                             const float* rewards, const uint8_t* terminated, const uint8_t* truncated,
                             const float* next_obs);

/* ---- gradient steps ------------------------------------------------------------------------------------------------
 * n_updates x update!(agent, alg, batch) sac.jl:299-404 over batches drawn like get_data_loader (replay_buffer.jl:116-157:
 * batch_size * n_updates indices uniform with replacement).  Per step, in the reference's order: entropy-coefficient
 * step (:313-343), critic step (:345-363), actor step with zero_critic_grads! (:365-383), polyak target update
 * (:385-389).  out has n_updates entries (may be NULL) */
int32_t dril_sac_update(dril_sac_handle* h, int32_t n_updates, dril_sac_stats* out);
/* injected batches for the NEXT dril_sac_update call only: idx i64 [n_updates][B] 0-based logical replay indices;
 * noise_ent / noise_next / noise_pi f32 [n_updates][B][A] standard normals for the three action_log_prob draws of one
 * update! (entropy constant :318-325, next actions in the critic target :120, actor loss :104).  Any pointer may be NULL
 * (= device Philox stream for that input) */
int32_t dril_sac_debug_set_batches(dril_sac_handle* h, int32_t n_updates, const int64_t* idx, const float* noise_ent,
                                   const float* noise_next, const float* noise_pi);
/* gradients of the LAST gradient step in the parameter layout (n = param_count): critic_grad = d critic_loss (non-zero
 * only in the two Q nets), actor_grad = d actor_loss after zero_critic_grads! (non-zero in actor_head and log_std) */
int32_t dril_sac_get_last_grads(dril_sac_handle* h, float* critic_grad, float* actor_grad, size_t n);

/* ---- train!(agent, env, alg::SAC, max_steps) sac.jl:406-549 ----------------------------------------------------------
 * first collection of max(1, start_steps / E) steps with random actions (:438-440,485-489), then `iterations` rounds of
 * {collect train_freq steps, get_gradient_steps updates}.  stats: up to stats_capacity entries, one per gradient step;
 * fps: up to fps_capacity entries, one per iteration.  Any out pointer may be NULL */
int32_t dril_sac_train(dril_sac_handle* h, int64_t max_steps, dril_sac_stats* stats, int64_t stats_capacity,
                       int64_t* n_updates_done, double* fps, int64_t fps_capacity, int32_t* iterations_done,
                       int64_t* total_steps);

/* the loop body of train! (sac.jl:464-535) for `iterations` iterations on a handle whose env has been reset: {collect train_freq env steps with the policy,
 * get_gradient_steps updates} enqueued back to back, the stream drained once per 64 iterations (dril_sac_train runs its iterations after the first through the same
 * loop).  Bit-identical to calling dril_sac_collect_rollout(train_freq, 0) + dril_sac_update(n) per iteration.  stats: one entry per gradient step; fps: one entry per
 * iteration = env steps / HIP-event time of its collection */
int32_t dril_sac_iterate(dril_sac_handle* h, int32_t iterations, dril_sac_stats* stats, int64_t stats_capacity, double* fps,
                         int64_t fps_capacity);

/* ---- measurement: accumulated HIP-event milliseconds since the last reset --------------------------------------------- */
int32_t dril_sac_profile_get(dril_sac_handle* h, double* collect_ms, int64_t* collect_steps, double* update_ms,
                             int64_t* updates);
int32_t dril_sac_profile_reset(dril_sac_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* DRIL_SAC_H */

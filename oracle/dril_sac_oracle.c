/*
 * dril_sac_oracle.c — CPU ORACLE of the off-policy (SAC) path (test infrastructure, NOT product code).
 * Textually included at the end of dril_oracle.c (one translation unit: it reuses the Philox, env-physics and
 * distribution helpers above).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PINNING.  polyak_update!            test/test_utils.jl:4-25                  -> PINNED (tests/golden/sac_kats.json)
 *           SquashedDiagGaussian      squashedDiagGaussian.jl:36-46 vs the closed form
 *                                     log N(atanh x) - sum log(1 - x^2)        -> PINNED (closed form, f64)
 *           train! schedule           sac.jl:436-447                           -> PINNED (arithmetic fixture)
 *           SAC losses / gradients / Adam sequencing                          -> PARITY UNPINNED by the reference's tests
 *             (test/test_sac.jl:187-348 checks only finiteness and which sub-trees are zero); cross-checked against
 *             torch-CPU autograd in tests/test_sac_oracle.py, which also asserts the properties test_sac.jl does.
 */
#include "../include/dril_sac.h"

/* SACLayer nets are always two hidden layers here (sac.jl:76-77: hidden_dims = [512, 512]); the on-policy oracle's layout is any-depth */
typedef struct { int D, H1, H2, O; size_t w1, b1, w2, b2, w3, b3, end; } sac_net;
static sac_net sac_net_at(size_t base, int D, int H1, int H2, int O) {
    sac_net n; n.D = D; n.H1 = H1; n.H2 = H2; n.O = O;
    n.w1 = base; n.b1 = n.w1 + (size_t)H1 * D; n.w2 = n.b1 + H1; n.b2 = n.w2 + (size_t)H2 * H1;
    n.w3 = n.b2 + H2; n.b3 = n.w3 + (size_t)O * H2; n.end = n.b3 + O;
    return n;
}

typedef struct orc_sac {
    dril_sac_config cfg; env_spec es; int D, A, H1, H2; float act_lo, act_hi;   /* bounds of the agent-facing Box action space */
    sac_net actor, q[2]; size_t log_std_off, P, Pq;
    float *params, *adam_m, *adam_v, *target;
    float bt_actor[2], bt_critic[2];          /* running beta powers of the two groups of leaves (Optimisers keeps them per leaf) */
    float log_ent, ent_m, ent_v, ent_bt[2];   /* ent_train_state, sac.jl:176-178 */
    int64_t grad_updates;                     /* agent.stats.gradient_updates */
    float* state; int32_t* step_count; uint32_t* episode; uint32_t* gstep; uint64_t env_seed0;
    int64_t cap, size, head;                  /* CircularBuffer: head = physical slot of logical index 0 */
    float *rb_obs, *rb_act, *rb_rew, *rb_next; uint8_t *rb_term, *rb_trunc;
    const float* collect_noise; size_t collect_noise_count;
    int inj_updates; const int64_t* inj_idx; const float *inj_ne, *inj_nn, *inj_np;
    uint64_t update_counter;
    float *g_critic, *g_actor;
    float target_entropy;
} orc_sac;

/* ---- generic MLP with the SACLayer activation (relu default, sac.jl:77); act: 0 tanh, 1 relu -------------------- */
static void dense_a(const float* W, const float* b, int out, int in, const float* x, float* y, int act) {
    for (int o = 0; o < out; ++o) y[o] = b[o];
    for (int i = 0; i < in; ++i) { const float xi = x[i]; const float* w = W + (size_t)i * out; for (int o = 0; o < out; ++o) y[o] += w[o] * xi; }
    if (act == 0) for (int o = 0; o < out; ++o) y[o] = tanhf(y[o]);
    else if (act == 1) for (int o = 0; o < out; ++o) y[o] = (y[o] > 0.0f || y[o] != y[o]) ? y[o] : 0.0f;   /* NNlib.relu = max(0, x): Julia's max propagates NaN */
}
static void mlp_fwd_a(const float* P, const sac_net* n, const float* x, float* h1, float* h2, float* out, int act) {
    dense_a(P + n->w1, P + n->b1, n->H1, n->D, x, h1, act);
    dense_a(P + n->w2, P + n->b2, n->H2, n->H1, h1, h2, act);
    dense_a(P + n->w3, P + n->b3, n->O, n->H2, h2, out, -1);
}
static inline float dact(float h, int act) { return act == 0 ? 1.0f - h * h : (h > 0.0f ? 1.0f : 0.0f); }
/* reverse pass of one sample given dL/dout; accumulates parameter gradients into G (f64, flat layout of P; may be NULL) and
 * writes dL/dx into dx (may be NULL) */
static void mlp_bwd_a(const float* P, const sac_net* n, const float* x, const float* h1, const float* h2, const float* dout,
                      double* G, float* dx, float* dz2, float* dz1, int act) {
    const int H1 = n->H1, H2 = n->H2, O = n->O, D = n->D;
    for (int j = 0; j < H2; ++j) {
        float s = 0; for (int o = 0; o < O; ++o) s += P[n->w3 + (size_t)j * O + o] * dout[o];
        dz2[j] = s * dact(h2[j], act);
        if (G) for (int o = 0; o < O; ++o) G[n->w3 + (size_t)j * O + o] += (double)dout[o] * h2[j];
    }
    if (G) for (int o = 0; o < O; ++o) G[n->b3 + o] += dout[o];
    for (int i = 0; i < H1; ++i) {
        const float* w = P + n->w2 + (size_t)i * H2; const float h = h1[i]; float s = 0;
        if (G) { double* g = G + n->w2 + (size_t)i * H2; for (int j = 0; j < H2; ++j) g[j] += (double)dz2[j] * h; }
        for (int j = 0; j < H2; ++j) s += w[j] * dz2[j];
        dz1[i] = s * dact(h, act);
    }
    if (G) for (int j = 0; j < H2; ++j) G[n->b2 + j] += dz2[j];
    for (int d = 0; d < D; ++d) {
        const float* w = P + n->w1 + (size_t)d * H1; float s = 0;
        if (G) { double* g = G + n->w1 + (size_t)d * H1; const float xd = x[d]; for (int i = 0; i < H1; ++i) g[i] += (double)dz1[i] * xd; }
        if (dx) { for (int i = 0; i < H1; ++i) s += w[i] * dz1[i]; dx[d] = s; }
    }
    if (G) for (int i = 0; i < H1; ++i) G[n->b1 + i] += dz1[i];
}

/* ---- SquashedDiagGaussian, squashedDiagGaussian.jl:24-46 (epsilon = 1e-6 :21) ------------------------------------ */
static inline float softplusf(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.0f ? x : 0.0f); }   /* Lux.softplus = NNlib.softplus */
/* logpdf(d, x) :36-46; g receives atanh(clamp(x)) */
static float squashed_logpdf(const float* x, const float* mu, const float* ls, int k, float* g) {
    const float eps = 1.0e-6f, lo = -1.0f + eps, hi = 1.0f - eps;
    float corr = 0;
    for (int i = 0; i < k; ++i) {
        float xc = x[i] < lo ? lo : (x[i] > hi ? hi : x[i]);
        g[i] = atanhf(xc);
        corr += 2.0f * (logf(2.0f) - g[i] - softplusf(-2.0f * g[i]));
    }
    return gauss_logpdf(g, mu, ls, k) - corr;
}
ORC_API float orc_squashed_logpdf(const float* x, const float* mu, const float* ls, int k) { float g[ORC_MAX_OUT]; return squashed_logpdf(x, mu, ls, k, g); }
/* rand(rng, d) :24-27 followed by logpdf(d, action): a = tanh(mu + exp(ls) * noise) */
static float squashed_sample_logp(const float* mu, const float* ls, const float* noise, int k, float* a, float* g) {
    for (int i = 0; i < k; ++i) a[i] = tanhf(mu[i] + expf(ls[i]) * noise[i]);      /* diagGaussian.jl:13-17 then tanh */
    return squashed_logpdf(a, mu, ls, k, g);
}
/* reverse of squashed_sample_logp for one sample: given dL/dlogp and dL/da, accumulate dL/dmu and dL/dlog_std.
 * Chain (Zygote through tanh -> clamp -> atanh -> logpdf): dg/du = 1 inside the clamp, 0 outside. */
static void squashed_backward(const float* mu, const float* ls, const float* noise, const float* a, const float* g, int k,
                              float dlogp, const float* da, float* dmu, double* dls) {
    const float eps = 1.0e-6f, lo = -1.0f + eps, hi = 1.0f - eps;
    for (int i = 0; i < k; ++i) {
        const float sig = expf(ls[i]), e2 = expf(-2.0f * ls[i]), d = g[i] - mu[i];
        const float inside = (a[i] >= lo && a[i] <= hi) ? 1.0f : 0.0f;
        const float dlp_dg = -d * e2 + 2.0f * tanhf(g[i]);                 /* gaussian part + d/dg of -correction */
        const float du = dlogp * dlp_dg * inside + da[i] * (1.0f - a[i] * a[i]);
        dmu[i] = dlogp * (d * e2) + du;
        dls[i] += (double)(dlogp * (-1.0f + d * d * e2) + du * sig * noise[i]);
    }
}

/* ---- lifetime ------------------------------------------------------------------------------------------------------ */
#define ORC_SAC_MAX_X 1048   /* Q-net input: obs (<= 1024, DRIL_ENV_EXTERNAL) ++ action (<= 16) */
ORC_API int32_t orc_sac_config_default(dril_sac_config* c, int32_t env_kind) {
    memset(c, 0, sizeof(*c));
    c->abi_version = DRIL_SAC_ABI_VERSION; c->env_kind = env_kind; c->n_envs = 1; c->episode_len = env_kind == DRIL_ENV_PENDULUM ? 200 : 500;
    c->hidden1 = 512; c->hidden2 = 512; c->activation = 1;                                   /* sac.jl:76-77 */
    c->buffer_capacity = 1000000; c->start_steps = 100; c->batch_size = 256; c->tau = 0.005f; c->gamma = 0.99f;   /* :26-31 */
    c->train_freq = 1; c->gradient_steps = 1; c->target_update_interval = 1;                 /* :32-35 */
    c->auto_ent_coef = 1; c->ent_coef_init = 1.0f; c->auto_target_entropy = 1; c->target_entropy = 0.0f;   /* entropy.jl:21-24 */
    c->learning_rate = 3.0e-4f; c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1.0e-8f;
    c->seed = 42; return DRIL_OK;
}
ORC_API int32_t orc_sac_reset_optimizer(orc_sac* c) {
    memset(c->adam_m, 0, c->P * 4); memset(c->adam_v, 0, c->P * 4);
    c->bt_actor[0] = c->bt_critic[0] = c->ent_bt[0] = c->cfg.adam_beta1; c->bt_actor[1] = c->bt_critic[1] = c->ent_bt[1] = c->cfg.adam_beta2;
    c->ent_m = c->ent_v = 0; c->grad_updates = 0; return DRIL_OK;
}
ORC_API int32_t orc_sac_create(const dril_sac_config* cfg, orc_sac** out) {
    const int ext = cfg && cfg->env_kind == DRIL_ENV_EXTERNAL;
    if (!cfg || cfg->abi_version != DRIL_SAC_ABI_VERSION || (!ext && cfg->env_kind != DRIL_ENV_PENDULUM && cfg->env_kind != DRIL_ENV_PENDULUM_SCALED && cfg->env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS && cfg->env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED)) return DRIL_ERR_INVALID_ARG;
    if (ext && (cfg->ext_obs_dim < 1 || cfg->ext_obs_dim > 1024 || cfg->ext_action_dim < 1 || cfg->ext_action_dim > 16 || !(cfg->ext_action_low < cfg->ext_action_high))) return DRIL_ERR_INVALID_ARG;
    orc_sac* c = (orc_sac*)calloc(1, sizeof(orc_sac)); c->cfg = *cfg; c->es = spec_of(cfg->env_kind);
    c->act_hi = act_bound(cfg->env_kind); c->act_lo = -c->act_hi;
    if (ext) { c->es.kind = DRIL_ENV_EXTERNAL; c->es.D = cfg->ext_obs_dim; c->es.S = 0; c->es.A = cfg->ext_action_dim; c->es.discrete = 0; c->act_lo = cfg->ext_action_low; c->act_hi = cfg->ext_action_high; }
    const int D = c->D = c->es.D, A = c->A = c->es.A, H1 = c->H1 = cfg->hidden1, H2 = c->H2 = cfg->hidden2, E = cfg->n_envs;
    c->actor = sac_net_at(0, D, H1, H2, A); c->q[0] = sac_net_at(c->actor.end, D + A, H1, H2, 1); c->q[1] = sac_net_at(c->q[0].end, D + A, H1, H2, 1);
    c->Pq = c->q[0].end - c->q[0].w1; c->log_std_off = c->q[1].end; c->P = c->log_std_off + A;
    c->params = (float*)calloc(c->P, 4); c->adam_m = (float*)calloc(c->P, 4); c->adam_v = (float*)calloc(c->P, 4);
    c->target = (float*)calloc(2 * c->Pq, 4); c->g_critic = (float*)calloc(c->P, 4); c->g_actor = (float*)calloc(c->P, 4);
    c->state = (float*)calloc((size_t)E * c->es.S, 4); c->step_count = (int32_t*)calloc(E, 4); c->episode = (uint32_t*)calloc(E, 4); c->gstep = (uint32_t*)calloc(E, 4);
    c->cap = cfg->buffer_capacity;
    c->rb_obs = (float*)calloc((size_t)c->cap * D, 4); c->rb_next = (float*)calloc((size_t)c->cap * D, 4); c->rb_act = (float*)calloc((size_t)c->cap * A, 4);
    c->rb_rew = (float*)calloc(c->cap, 4); c->rb_term = (uint8_t*)calloc(c->cap, 1); c->rb_trunc = (uint8_t*)calloc(c->cap, 1);
    c->log_ent = logf(cfg->ent_coef_init);                                                   /* init_entropy_coefficient, sac.jl:207-213 */
    c->target_entropy = cfg->auto_target_entropy ? -(float)A : cfg->target_entropy;          /* get_target_entropy, sac.jl:47-57 */
    orc_sac_reset_optimizer(c);
    *out = c; return DRIL_OK;
}
ORC_API int32_t orc_sac_destroy(orc_sac* c) {
    if (!c) return DRIL_OK;
    free(c->params); free(c->adam_m); free(c->adam_v); free(c->target); free(c->g_critic); free(c->g_actor); free(c->state); free(c->step_count);
    free(c->episode); free(c->gstep); free(c->rb_obs); free(c->rb_next); free(c->rb_act); free(c->rb_rew); free(c->rb_term); free(c->rb_trunc); free(c);
    return DRIL_OK;
}
ORC_API int32_t orc_sac_obs_dim(const orc_sac* c) { return c->D; }
ORC_API int32_t orc_sac_action_dim(const orc_sac* c) { return c->A; }
ORC_API int64_t orc_sac_param_count(const orc_sac* c) { return (int64_t)c->P; }
ORC_API int64_t orc_sac_q_param_count(const orc_sac* c) { return (int64_t)c->Pq; }
/* Agent(layer, alg::SAC): Q_target_parameters = copy_critic_parameters(layer, ps), sac.jl:172,191-197 */
ORC_API int32_t orc_sac_set_params(orc_sac* c, const float* flat, size_t n) {
    if (n != c->P) return DRIL_ERR_INVALID_ARG;
    memcpy(c->params, flat, n * 4); memcpy(c->target, c->params + c->q[0].w1, 2 * c->Pq * 4); return DRIL_OK;
}
ORC_API int32_t orc_sac_get_params(orc_sac* c, float* flat, size_t n) { if (n != c->P) return DRIL_ERR_INVALID_ARG; memcpy(flat, c->params, n * 4); return DRIL_OK; }
ORC_API int32_t orc_sac_get_target_params(orc_sac* c, float* flat, size_t n) { if (n != 2 * c->Pq) return DRIL_ERR_INVALID_ARG; memcpy(flat, c->target, n * 4); return DRIL_OK; }
ORC_API int32_t orc_sac_set_target_params(orc_sac* c, const float* flat, size_t n) { if (n != 2 * c->Pq) return DRIL_ERR_INVALID_ARG; memcpy(c->target, flat, n * 4); return DRIL_OK; }
ORC_API int32_t orc_sac_get_log_ent_coef(orc_sac* c, float* v) { *v = c->log_ent; return DRIL_OK; }
ORC_API int32_t orc_sac_set_log_ent_coef(orc_sac* c, float v) { c->log_ent = v; return DRIL_OK; }

/* ---- env (MultiThreadedParallelEnv verbs, multithreadedParallelEnv.jl:12-74) ----------------------------------------- */
ORC_API int32_t orc_sac_env_reset(orc_sac* c, uint64_t seed) {
    c->env_seed0 = seed;
    for (int e = 0; e < c->cfg.n_envs; ++e) { c->episode[e] = 0; c->step_count[e] = 0; c->gstep[e] = 0; env_reset_one(c->es.kind, seed + e, 0, c->state + (size_t)e * c->es.S); }
    return DRIL_OK;
}
ORC_API int32_t orc_sac_env_observe(orc_sac* c, float* obs) {
    for (int e = 0; e < c->cfg.n_envs; ++e) env_obs_one(c->es.kind, c->state + (size_t)e * c->es.S, obs + (size_t)e * c->D);
    return DRIL_OK;
}

/* ---- layer calls -------------------------------------------------------------------------------------------------- */
/* action_log_prob: layer_methods.jl:66-76 */
ORC_API int32_t orc_sac_action_log_prob(orc_sac* c, const float* obs, int64_t B, const float* noise, float* actions, float* logp) {
    const int D = c->D, A = c->A, act = c->cfg.activation;
#pragma omp parallel if (B >= 256)
    {
        float* h1 = (float*)malloc(4 * c->H1); float* h2 = (float*)malloc(4 * c->H2); float mu[ORC_MAX_OUT], g[ORC_MAX_OUT];
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            mlp_fwd_a(c->params, &c->actor, obs + b * D, h1, h2, mu, act);
            logp[b] = squashed_sample_logp(mu, c->params + c->log_std_off, noise + b * A, A, actions + b * A, g);
        }
        free(h1); free(h2);
    }
    return DRIL_OK;
}
/* predict_actions(agent, obs; deterministic, raw): sac.jl:215-240; mode :48-50; to_env(TanhScaleAdapter) default_adapters.jl:13-21,
 * scale_to_space spaces.jl:134-139 with Box(-2, 2) */
ORC_API int32_t orc_sac_predict_actions(orc_sac* c, const float* obs, int64_t B, int32_t deterministic, const float* noise, float* raw, float* env) {
    const int D = c->D, A = c->A, act = c->cfg.activation;
    const float high = c->act_hi, low = c->act_lo;
#pragma omp parallel if (B >= 256)
    {
        float* h1 = (float*)malloc(4 * c->H1); float* h2 = (float*)malloc(4 * c->H2); float mu[ORC_MAX_OUT], a[ORC_MAX_OUT];
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            mlp_fwd_a(c->params, &c->actor, obs + b * D, h1, h2, mu, act);
            for (int i = 0; i < A; ++i) {
                a[i] = deterministic ? tanhf(mu[i]) : tanhf(mu[i] + expf(c->params[c->log_std_off + i]) * noise[b * A + i]);
                if (raw) raw[b * A + i] = a[i];
                if (env) env[b * A + i] = tanhf(a[i]) * (high - low) / 2.0f + (low + high) / 2.0f;
            }
        }
        free(h1); free(h2);
    }
    return DRIL_OK;
}
/* predict_values(layer, obs, actions, ps, st): layer_methods.jl:63-67, get_values_from_features layer_forward.jl:75-87 (vcat(feats, actions)) */
ORC_API int32_t orc_sac_predict_q(orc_sac* c, const float* obs, const float* actions, int64_t B, int32_t use_target, float* q) {
    const int D = c->D, A = c->A, act = c->cfg.activation;
#pragma omp parallel if (B >= 256)
    {
        float* h1 = (float*)malloc(4 * c->H1); float* h2 = (float*)malloc(4 * c->H2); float x[ORC_SAC_MAX_X];
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            memcpy(x, obs + b * D, D * 4); memcpy(x + D, actions + b * A, A * 4);
            for (int k = 0; k < 2; ++k) {
                if (use_target) { sac_net n = sac_net_at((size_t)k * c->Pq, D + A, c->H1, c->H2, 1); mlp_fwd_a(c->target, &n, x, h1, h2, q + b * 2 + k, act); }
                else mlp_fwd_a(c->params, &c->q[k], x, h1, h2, q + b * 2 + k, act);
            }
        }
        free(h1); free(h2);
    }
    return DRIL_OK;
}

/* ---- replay buffer: CircularBuffer semantics, replay_buffer.jl:14-31,98-114 -------------------------------------------- */
static void rb_push(orc_sac* c, const float* obs, const float* act, float rew, uint8_t term, uint8_t trunc, const float* next) {
    int64_t slot;
    if (c->size < c->cap) { slot = (c->head + c->size) % c->cap; c->size += 1; }
    else { slot = c->head; c->head = (c->head + 1) % c->cap; }                                /* overwrite the oldest */
    memcpy(c->rb_obs + slot * c->D, obs, c->D * 4); memcpy(c->rb_next + slot * c->D, next, c->D * 4); memcpy(c->rb_act + slot * c->A, act, c->A * 4);
    c->rb_rew[slot] = rew; c->rb_term[slot] = term; c->rb_trunc[slot] = trunc;
}
ORC_API int64_t orc_sac_replay_size(const orc_sac* c) { return c->size; }
ORC_API int64_t orc_sac_replay_capacity(const orc_sac* c) { return c->cap; }
ORC_API int32_t orc_sac_replay_copy_out(orc_sac* c, int32_t which, void* host, size_t bytes) {
    const int D = c->D, A = c->A;
    size_t w = which == DRIL_RB_OBSERVATIONS || which == DRIL_RB_NEXT_OBSERVATIONS ? (size_t)D * 4 : which == DRIL_RB_ACTIONS ? (size_t)A * 4
             : which == DRIL_RB_REWARDS ? 4 : 1;
    if (bytes != w * (size_t)c->size) return DRIL_ERR_INVALID_ARG;
    const char* src = which == DRIL_RB_OBSERVATIONS ? (char*)c->rb_obs : which == DRIL_RB_NEXT_OBSERVATIONS ? (char*)c->rb_next : which == DRIL_RB_ACTIONS ? (char*)c->rb_act
                    : which == DRIL_RB_REWARDS ? (char*)c->rb_rew : which == DRIL_RB_TERMINATED ? (char*)c->rb_term : which == DRIL_RB_TRUNCATED ? (char*)c->rb_trunc : NULL;
    if (!src) return DRIL_ERR_INVALID_ARG;
    for (int64_t i = 0; i < c->size; ++i) memcpy((char*)host + i * w, src + ((c->head + i) % c->cap) * w, w);
    return DRIL_OK;
}
ORC_API int32_t orc_sac_replay_fill(orc_sac* c, int64_t count, const float* obs, const float* act, const float* rew, const uint8_t* term,
                                    const uint8_t* trunc, const float* next) {
    c->size = 0; c->head = 0;                                                                 /* empty!(buffer) :62-70 */
    for (int64_t i = 0; i < count; ++i) rb_push(c, obs + i * c->D, act + i * c->A, rew[i], term[i], trunc ? trunc[i] : 0, next + i * c->D);
    return DRIL_OK;
}

/* ---- collection: off_policy_collection.jl:28-96,117-136 ----------------------------------------------------------------
 * Noise per (step, env): injected, else Philox stream 1 keyed by seed + env at the env's global step count (same stream the
 * on-policy rollout uses): standard normals for policy actions, uniforms for random actions. */
ORC_API int32_t orc_sac_debug_set_collect_noise(orc_sac* c, const float* noise, size_t count) { c->collect_noise = noise; c->collect_noise_count = count; return DRIL_OK; }
ORC_API int32_t orc_sac_collect_rollout(orc_sac* c, int32_t n_steps, int32_t use_random, double* fps) {
    const int E = c->cfg.n_envs, D = c->D, A = c->A, S = c->es.S;
    if (c->collect_noise && c->collect_noise_count != (size_t)n_steps * E * A) return DRIL_ERR_INVALID_ARG;
    float* obs = (float*)malloc((size_t)E * D * 4); float* nobs = (float*)malloc((size_t)E * D * 4); float* nz = (float*)malloc((size_t)E * A * 4);
    float* raw = (float*)malloc((size_t)E * A * 4); float* envact = (float*)malloc((size_t)E * A * 4);
    const float high = c->act_hi, low = c->act_lo;
    orc_sac_env_observe(c, obs);                                                              /* :41 */
    for (int t = 0; t < n_steps; ++t) {
        if (c->collect_noise) memcpy(nz, c->collect_noise + (size_t)t * E * A, (size_t)E * A * 4);
        else for (int e = 0; e < E; ++e) for (int a0 = 0; a0 < A; a0 += 2) {
            uint32_t r[4]; uint64_t k = c->env_seed0 + e; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), c->gstep[e], 0, 1, (uint32_t)(a0 / 2), r);
            if (use_random) { nz[e * A + a0] = u01_f32(r[0]); if (a0 + 1 < A) nz[e * A + a0 + 1] = u01_f32(r[2]); }
            else { nz[e * A + a0] = randn_f32(r[0], r[1]); if (a0 + 1 < A) nz[e * A + a0 + 1] = randn_f32(r[2], r[3]); }
        }
        if (use_random) {                                                                     /* :50-53: rand(rng, act_space), already in env space */
            for (int i = 0; i < E * A; ++i) { raw[i] = low + nz[i] * (high - low); envact[i] = raw[i]; }
        } else orc_sac_predict_actions(c, obs, E, 0, nz, raw, envact);                        /* :55-58 */
        for (int e = 0; e < E; ++e) {                                                         /* act!(env, processed_actions) :60, multithreadedParallelEnv.jl:47-74 */
            float* st = c->state + (size_t)e * S; int term = 0; float tobs[ORC_MAX_OBS];
            float r = env_step_one(c->es.kind, st, envact + (size_t)e * A, 0, 0, &term);
            c->step_count[e] += 1; c->gstep[e] += 1;
            int trunc = c->step_count[e] >= c->cfg.episode_len;
            if (trunc) env_obs_one(c->es.kind, st, tobs);
            if (term || trunc) { c->episode[e] += 1; c->step_count[e] = 0; env_reset_one(c->es.kind, c->env_seed0 + e, c->episode[e], st); }
            env_obs_one(c->es.kind, st, nobs + (size_t)e * D);                                /* new_obs = observe(env) :61 */
            /* :63-92 + replay_buffer.jl:127-146: the next observation of this transition is the truncated_observation when the
             * trajectory ends here by truncation, else the following observation of the same env */
            rb_push(c, obs + (size_t)e * D, raw + (size_t)e * A, r, (uint8_t)term, (uint8_t)trunc, trunc ? tobs : nobs + (size_t)e * D);
        }
        memcpy(obs, nobs, (size_t)E * D * 4);
    }
    c->collect_noise = NULL; c->collect_noise_count = 0;
    if (fps) *fps = 0.0;
    free(obs); free(nobs); free(nz); free(raw); free(envact);
    return DRIL_OK;
}

/* one env step of the caller's host envs into the ring (DRIL_ENV_EXTERNAL): off_policy_collection.jl:63-92 + push! replay_buffer.jl:98-114 */
ORC_API int32_t orc_sac_ext_push(orc_sac* c, const float* obs, const float* stored_actions, const float* rewards, const uint8_t* terminated,
                                 const uint8_t* truncated, const float* next_obs, const float* terminal_obs) {
    const int E = c->cfg.n_envs, D = c->D, A = c->A;
    if (c->es.kind != DRIL_ENV_EXTERNAL) return DRIL_ERR_UNSUPPORTED;
    for (int e = 0; e < E; ++e) {
        if (truncated[e] && !terminal_obs) return DRIL_ERR_INVALID_ARG;
        rb_push(c, obs + (size_t)e * D, stored_actions + (size_t)e * A, rewards[e], terminated[e], truncated[e],
                truncated[e] ? terminal_obs + (size_t)e * D : next_obs + (size_t)e * D);
    }
    return DRIL_OK;
}

/* ---- one gradient step: update!(agent, alg::SAC, batch) sac.jl:299-404 ------------------------------------------------- */
static void adam_range(orc_sac* c, size_t lo, size_t hi, const float* g /* NULL = zero gradients */, const float* bt) {
    const float b1 = c->cfg.adam_beta1, b2 = c->cfg.adam_beta2, eps = c->cfg.adam_eps, eta = c->cfg.learning_rate;
    for (size_t i = lo; i < hi; ++i) {
        const float gi = g ? g[i] : 0.0f;
        float m = b1 * c->adam_m[i] + (1.0f - b1) * gi, v = b2 * c->adam_v[i] + (1.0f - b2) * gi * gi;
        c->adam_m[i] = m; c->adam_v[i] = v;
        c->params[i] -= m / (1.0f - bt[0]) / (sqrtf(v / (1.0f - bt[1])) + eps) * eta;
    }
}
ORC_API int32_t orc_sac_debug_set_batches(orc_sac* c, int32_t n_updates, const int64_t* idx, const float* ne, const float* nn, const float* np) {
    c->inj_updates = n_updates; c->inj_idx = idx; c->inj_ne = ne; c->inj_nn = nn; c->inj_np = np; return DRIL_OK;
}
/* index / noise streams of gradient step `u` (global counter): Philox keyed by cfg.seed ^ 0x5ac5ac5ac5ac5ac5, counter
 * (u lo, u hi, stream, sample * 4 + sub): stream 4 = replay indices (53-bit uniform * size), 5/6/7 = the three normal draws */
static int64_t sac_sample_index(const orc_sac* c, uint64_t u, int64_t i) {
    uint32_t r[4]; uint64_t k = c->cfg.seed ^ 0x5ac5ac5ac5ac5ac5ull;
    philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)u, (uint32_t)(u >> 32), 4, (uint32_t)i, r);
    int64_t j = (int64_t)(u01_f64(r[0], r[1]) * (double)c->size); return j < c->size ? j : c->size - 1;
}
static float sac_noise(const orc_sac* c, uint64_t u, int stream, int64_t i, int a) {
    uint32_t r[4]; uint64_t k = c->cfg.seed ^ 0x5ac5ac5ac5ac5ac5ull;
    philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)u, (uint32_t)(u >> 32), (uint32_t)stream, (uint32_t)(i * 4 + a / 2), r);
    return (a & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]);
}
static void sac_one_update(orc_sac* c, int inj_slot, dril_sac_stats* out) {
    const int D = c->D, A = c->A, B = c->cfg.batch_size, H1 = c->H1, H2 = c->H2, act = c->cfg.activation;
    const uint64_t u = c->update_counter;
    float* obs = (float*)malloc((size_t)B * D * 4); float* nobs = (float*)malloc((size_t)B * D * 4); float* actn = (float*)malloc((size_t)B * A * 4);
    float* rew = (float*)malloc(B * 4); uint8_t* term = (uint8_t*)malloc(B);
    float* ne = (float*)malloc((size_t)B * A * 4); float* nn = (float*)malloc((size_t)B * A * 4); float* np = (float*)malloc((size_t)B * A * 4);
    float* y = (float*)malloc(B * 4); float* q = (float*)malloc((size_t)B * 2 * 4);
    /* get_data_loader, replay_buffer.jl:116-157 */
    for (int i = 0; i < B; ++i) {
        int64_t j = (c->inj_idx && inj_slot >= 0) ? c->inj_idx[(size_t)inj_slot * B + i] : sac_sample_index(c, u, i);
        int64_t slot = (c->head + j) % c->cap;
        memcpy(obs + (size_t)i * D, c->rb_obs + slot * D, D * 4); memcpy(nobs + (size_t)i * D, c->rb_next + slot * D, D * 4);
        memcpy(actn + (size_t)i * A, c->rb_act + slot * A, A * 4); rew[i] = c->rb_rew[slot]; term[i] = c->rb_term[slot];
        for (int a = 0; a < A; ++a) {
            ne[i * A + a] = (c->inj_ne && inj_slot >= 0) ? c->inj_ne[((size_t)inj_slot * B + i) * A + a] : sac_noise(c, u, 5, i, a);
            nn[i * A + a] = (c->inj_nn && inj_slot >= 0) ? c->inj_nn[((size_t)inj_slot * B + i) * A + a] : sac_noise(c, u, 6, i, a);
            np[i * A + a] = (c->inj_np && inj_slot >= 0) ? c->inj_np[((size_t)inj_slot * B + i) * A + a] : sac_noise(c, u, 7, i, a);
        }
    }
    const float* ls = c->params + c->log_std_off;
    float ent_loss = 0; int has_ent = 0;
    /* ---- entropy coefficient, :313-343: c = mean(log_probs_pi .+ target_entropy); loss = -(log_ent_coef * c) ---- */
    if (c->cfg.auto_ent_coef) {
        float* a_ = (float*)malloc((size_t)B * A * 4); float* lp = (float*)malloc(B * 4);
        orc_sac_action_log_prob(c, obs, B, ne, a_, lp);
        double s = 0; for (int i = 0; i < B; ++i) s += (double)(lp[i] + c->target_entropy);
        const float cc = (float)(s / B);
        ent_loss = -(c->log_ent * cc); has_ent = 1;
        const float g = -cc, b1 = c->cfg.adam_beta1, b2 = c->cfg.adam_beta2;
        c->ent_m = b1 * c->ent_m + (1.0f - b1) * g; c->ent_v = b2 * c->ent_v + (1.0f - b2) * g * g;
        c->log_ent -= c->ent_m / (1.0f - c->ent_bt[0]) / (sqrtf(c->ent_v / (1.0f - c->ent_bt[1])) + c->cfg.adam_eps) * c->cfg.learning_rate;
        c->ent_bt[0] *= b1; c->ent_bt[1] *= b2;
        free(a_); free(lp);
    }
    const float alpha = expf(c->log_ent);                                                    /* :100,114 */
    /* ---- critic loss, sac_critic_loss :107-150 ---- */
    {
        float* na = (float*)malloc((size_t)B * A * 4); float* nlp = (float*)malloc(B * 4); float* nq = (float*)malloc((size_t)B * 2 * 4);
        orc_sac_action_log_prob(c, nobs, B, nn, na, nlp);                                     /* :131 (terminated rows are dropped by the reference :127; unused here) */
        orc_sac_predict_q(c, nobs, na, B, 1, nq);                                             /* :133-135 with the target parameters */
        for (int i = 0; i < B; ++i) {
            const float mn = nq[2 * i] < nq[2 * i + 1] ? nq[2 * i] : nq[2 * i + 1];
            y[i] = term[i] ? rew[i] : rew[i] + c->cfg.gamma * (mn - alpha * nlp[i]);          /* :136-142 */
        }
        free(na); free(nlp); free(nq);
    }
    double* G = (double*)calloc(c->P, 8);
    double closs = 0, qsum = 0;
#pragma omp parallel if (B >= 64)
    {
        double* Gl = (double*)calloc(c->P, 8); double cl = 0, qs = 0;
        float* h1 = (float*)malloc(4 * H1); float* h2 = (float*)malloc(4 * H2); float* dz2 = (float*)malloc(4 * H2); float* dz1 = (float*)malloc(4 * H1);
        float x[ORC_SAC_MAX_X];
#pragma omp for schedule(static)
        for (int i = 0; i < B; ++i) {
            memcpy(x, obs + (size_t)i * D, D * 4); memcpy(x + D, actn + (size_t)i * A, A * 4);
            for (int k = 0; k < 2; ++k) {
                float qv; mlp_fwd_a(c->params, &c->q[k], x, h1, h2, &qv, act);               /* :117 */
                const float d = qv - y[i]; cl += 0.5 * (double)d * d / B; qs += qv;           /* :146 */
                const float dout = d / (float)B;
                mlp_bwd_a(c->params, &c->q[k], x, h1, h2, &dout, Gl, NULL, dz2, dz1, act);
            }
        }
#pragma omp critical
        { for (size_t p = c->q[0].w1; p < c->q[1].end; ++p) G[p] += Gl[p]; closs += cl; qsum += qs; }
        free(Gl); free(h1); free(h2); free(dz2); free(dz1);
    }
    memset(c->g_critic, 0, c->P * 4);
    double n2c = 0; for (size_t p = c->q[0].w1; p < c->q[1].end; ++p) { c->g_critic[p] = (float)G[p]; n2c += (double)c->g_critic[p] * c->g_critic[p]; }
    /* apply_gradients(train_state, critic_grad) :362 — actor_head / log_std gradients are `nothing` (everything that touches them is
     * inside @ignore_derivatives :129-143), so Optimisers leaves those leaves and their Adam state alone */
    adam_range(c, c->q[0].w1, c->q[1].end, c->g_critic, c->bt_critic);
    c->bt_critic[0] *= c->cfg.adam_beta1; c->bt_critic[1] *= c->cfg.adam_beta2;
    /* ---- actor loss, sac_actor_loss :93-105, with the UPDATED critics ---- */
    memset(G, 0, c->P * 8);
    double aloss = 0;
#pragma omp parallel if (B >= 64)
    {
        double* Gl = (double*)calloc(c->P, 8); double al = 0;
        float* h1 = (float*)malloc(4 * H1); float* h2 = (float*)malloc(4 * H2); float* dz2 = (float*)malloc(4 * H2); float* dz1 = (float*)malloc(4 * H1);
        float* qh1 = (float*)malloc(4 * H1 * 2); float* qh2 = (float*)malloc(4 * H2 * 2);
        float x[ORC_SAC_MAX_X], mu[ORC_MAX_OUT], a[ORC_MAX_OUT], g[ORC_MAX_OUT], dx[ORC_SAC_MAX_X], da[ORC_MAX_OUT], dmu[ORC_MAX_OUT];
#pragma omp for schedule(static)
        for (int i = 0; i < B; ++i) {
            const float* o = obs + (size_t)i * D;
            mlp_fwd_a(c->params, &c->actor, o, h1, h2, mu, act);
            const float lp = squashed_sample_logp(mu, ls, np + (size_t)i * A, A, a, g);      /* :101 */
            memcpy(x, o, D * 4); memcpy(x + D, a, A * 4);
            float qv[2];
            for (int k = 0; k < 2; ++k) mlp_fwd_a(c->params, &c->q[k], x, qh1 + k * H1, qh2 + k * H2, &qv[k], act);   /* :102 */
            const int km = qv[1] < qv[0] ? 1 : 0;                                             /* minimum(q_values, dims = 1) :103 */
            al += ((double)alpha * lp - qv[km]) / B;                                          /* :104 */
            const float dq = -1.0f / (float)B;
            mlp_bwd_a(c->params, &c->q[km], x, qh1 + km * H1, qh2 + km * H2, &dq, NULL, dx, dz2, dz1, act);
            for (int j = 0; j < A; ++j) da[j] = dx[D + j];
            squashed_backward(mu, ls, np + (size_t)i * A, a, g, A, alpha / (float)B, da, dmu, Gl + c->log_std_off);
            mlp_bwd_a(c->params, &c->actor, o, h1, h2, dmu, Gl, NULL, dz2, dz1, act);
        }
#pragma omp critical
        { for (size_t p = 0; p < c->actor.end; ++p) G[p] += Gl[p]; for (int j = 0; j < A; ++j) G[c->log_std_off + j] += Gl[c->log_std_off + j]; aloss += al; }
        free(Gl); free(h1); free(h2); free(dz2); free(dz1); free(qh1); free(qh2);
    }
    memset(c->g_actor, 0, c->P * 4);                                                          /* zero_critic_grads! :381, layer_helpers.jl:114-144 */
    double n2a = 0;
    for (size_t p = 0; p < c->actor.end; ++p) { c->g_actor[p] = (float)G[p]; n2a += (double)c->g_actor[p] * c->g_actor[p]; }
    for (int j = 0; j < A; ++j) { c->g_actor[c->log_std_off + j] = (float)G[c->log_std_off + j]; n2a += (double)c->g_actor[c->log_std_off + j] * c->g_actor[c->log_std_off + j]; }
    /* apply_gradients(train_state, actor_loss_grad) :382 — the critic leaves carry ZERO arrays (not `nothing`), so Adam still
     * decays their moments and moves them along the remaining momentum */
    adam_range(c, 0, c->actor.end, c->g_actor, c->bt_actor);
    adam_range(c, c->log_std_off, c->P, c->g_actor, c->bt_actor);
    adam_range(c, c->q[0].w1, c->q[1].end, NULL, c->bt_critic);
    c->bt_actor[0] *= c->cfg.adam_beta1; c->bt_actor[1] *= c->cfg.adam_beta2;
    c->bt_critic[0] *= c->cfg.adam_beta1; c->bt_critic[1] *= c->cfg.adam_beta2;
    /* ---- target networks :385-389, polyak_update! optimization_utils.jl:3-6 ---- */
    if (c->grad_updates % c->cfg.target_update_interval == 0) {
        const float tau = c->cfg.tau; const float* src = c->params + c->q[0].w1;
        for (size_t p = 0; p < 2 * c->Pq; ++p) c->target[p] = tau * src[p] + (1.0f - tau) * c->target[p];
    }
    c->grad_updates += 1; c->update_counter += 1;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->actor_loss = (float)aloss; out->critic_loss = (float)closs; out->entropy_loss = ent_loss; out->has_entropy_loss = has_ent;
        out->mean_q_values = (float)(qsum / (2.0 * B)); out->entropy_coefficient = expf(c->log_ent);      /* :391 */
        out->grad_norm = (float)sqrt(n2c + n2a);                                              /* :393 */
    }
    free(G); free(obs); free(nobs); free(actn); free(rew); free(term); free(ne); free(nn); free(np); free(y); free(q);
}
ORC_API int32_t orc_sac_update(orc_sac* c, int32_t n_updates, dril_sac_stats* out) {
    if (c->size <= 0) return DRIL_ERR_NOT_INITIALISED;
    const int inj = c->inj_updates;
    if (inj && inj != n_updates) return DRIL_ERR_INVALID_ARG;
    for (int k = 0; k < n_updates; ++k) sac_one_update(c, inj ? k : -1, out ? out + k : NULL);
    c->inj_updates = 0; c->inj_idx = NULL; c->inj_ne = c->inj_nn = c->inj_np = NULL;
    return DRIL_OK;
}
ORC_API int32_t orc_sac_get_last_grads(orc_sac* c, float* gc, float* ga, size_t n) {
    if (n != c->P) return DRIL_ERR_INVALID_ARG;
    if (gc) memcpy(gc, c->g_critic, n * 4); if (ga) memcpy(ga, c->g_actor, n * 4); return DRIL_OK;
}
/* polyak_update!(target, source, tau) on caller arrays, optimization_utils.jl:3-6 (known answers test/test_utils.jl:4-25) */
ORC_API void orc_polyak_update(float* target, const float* source, size_t n, float tau) {
    for (size_t i = 0; i < n; ++i) target[i] = tau * source[i] + (1.0f - tau) * target[i];
}

/* ---- train!(agent, replay_buffer, env, alg::SAC, max_steps) sac.jl:414-549 -------------------------------------------- */
/* the schedule arithmetic :436-447 on its own (Julia div truncates toward zero like C) */
ORC_API void orc_sac_schedule(int64_t max_steps, int32_t n_envs, int32_t start_steps, int32_t train_freq, int32_t gradient_steps,
                              int64_t* first_steps, int64_t* iterations, int64_t* total_steps, int64_t* updates_per_iteration) {
    const int64_t E = n_envs;
    const int64_t total_start = start_steps > 0 ? start_steps : (int64_t)train_freq * E;     /* :436 */
    const int64_t q = total_start / E; const int64_t adjusted = (q > 1 ? q : 1) * E;          /* :437 */
    const int64_t n_steps = adjusted / E;                                                     /* :438 */
    const int64_t it = (max_steps - adjusted) / ((int64_t)train_freq * E) + 1;                /* :443 */
    *first_steps = n_steps; *iterations = it; *total_steps = n_steps * E + (int64_t)train_freq * E * (it - 1);   /* :445 */
    *updates_per_iteration = gradient_steps == -1 ? (int64_t)train_freq * E : gradient_steps; /* get_gradient_steps :59-65 */
}
/* the loop body of train! (sac.jl:464-535) for `iterations` iterations: collect train_freq steps with the policy, then the gradient steps (include/dril_sac.h dril_sac_iterate) */
ORC_API int32_t orc_sac_iterate(orc_sac* c, int32_t iterations, dril_sac_stats* stats, int64_t stats_capacity, double* fps, int64_t fps_capacity) {
    const int64_t n_upd = c->cfg.gradient_steps == -1 ? (int64_t)c->cfg.train_freq * c->cfg.n_envs : c->cfg.gradient_steps;
    int64_t done = 0;
    for (int it = 0; it < iterations; ++it) {
        double f = 0;
        int32_t rc = orc_sac_collect_rollout(c, c->cfg.train_freq, 0, &f);
        if (rc) return rc;
        if (fps && it < fps_capacity) fps[it] = f;
        for (int64_t k = 0; k < n_upd; ++k) {
            dril_sac_stats s; sac_one_update(c, -1, &s);
            if (stats && done < stats_capacity) stats[done] = s;
            ++done;
        }
    }
    return DRIL_OK;
}
ORC_API int32_t orc_sac_train(orc_sac* c, int64_t max_steps, dril_sac_stats* stats, int64_t stats_capacity, int64_t* n_updates_done,
                              double* fps, int64_t fps_capacity, int32_t* iterations_done, int64_t* total_steps) {
    int64_t n_steps, iterations, total, n_upd;
    orc_sac_schedule(max_steps, c->cfg.n_envs, c->cfg.start_steps, c->cfg.train_freq, c->cfg.gradient_steps, &n_steps, &iterations, &total, &n_upd);
    int64_t done = 0; int it = 0;
    for (; it < iterations; ++it) {
        double f = 0;
        int32_t rc = orc_sac_collect_rollout(c, (int32_t)n_steps, it == 0 && c->cfg.start_steps > 0, &f);   /* :485-489 */
        if (rc) return rc;
        if (fps && it < fps_capacity) fps[it] = f;
        n_steps = c->cfg.train_freq;                                                          /* :511 */
        for (int64_t k = 0; k < n_upd; ++k) {                                                 /* :514-531 */
            dril_sac_stats s; sac_one_update(c, -1, &s);
            if (stats && done < stats_capacity) stats[done] = s;
            ++done;
        }
    }
    if (n_updates_done) *n_updates_done = done; if (iterations_done) *iterations_done = it; if (total_steps) *total_steps = it > 0 ? total : 0;
    return DRIL_OK;
}

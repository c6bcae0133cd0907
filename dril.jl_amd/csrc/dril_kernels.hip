// dril_kernels.hip — the HIP kernels of libdril_hip.so (gfx950 / CDNA4 only) other than the PPO update kernels, and their launchers.
//
// Kernel map (reference function each one replaces — paths relative to the reference root):
//   env_reset_kernel / env_observe_kernel / env_step_kernel   MultiThreadedParallelEnv reset!/observe/act!
//                                                            src/environment_wrappers/multithreadedParallelEnv.jl:12-74
//   policy_kernel        layer(obs,ps,st) / evaluate_actions / predict_values
//                        src/layers/layer_forward.jl:3-39, src/layers/layer_methods.jl:28-61
//   rollout_kernel       collect_trajectories, src/buffers/trajectory.jl:22-78 (persistent: one wave owns 32 envs for all T steps)
//   gae_kernel           compute_advantages! trajectory.jl:80-102 + returns rollout_buffer.jl:87
//   adv_moments_kernel   normalize! statistics, src/algorithms/ppo.jl:350-356
//   ppo_grad_*_kernel    (alg::PPO)(layer,ps,st,batch) ppo.jl:365-407 + its reverse pass (Zygote in the reference, ppo.jl:207): dril_grad_f32.hip (exact f32, small
//                        minibatches), dril_grad_pair.hip (hidden [64,64], bf16 matrix cores: the headline kernel), dril_grad_wide.hip (hidden 128 / 256)
//   grad_reduce_kernel / grad_norm_kernel / adam_kernel   nested_norm, nested_scale!, target_kl check, Adam — ppo.jl:213-239
//   explained_var_kernel ppo.jl:256
#include <cstdlib>
#include <utility>
#include <algorithm>
#include <mutex>
#include <vector>

#include "dril_internal.h"
#include "dril_grad_common.h"
#include "dril_split_pieces.h"

namespace dril {

// =============================================================================================
// env verbs (step-granular path)
// =============================================================================================
template <int KIND>
__global__ void env_reset_kernel(int E, uint64_t seed0, float* state, int32_t* step_count, uint32_t* episode,
                                 uint32_t* gstep, float* disc_returns) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float st[EnvSpec<KIND>::S];
    env_reset<KIND>(seed0 + (uint64_t)e, 0u, st);
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::S; ++i) state[(size_t)e * EnvSpec<KIND>::S + i] = st[i];
    step_count[e] = 0; episode[e] = 0; gstep[e] = 0; disc_returns[e] = 0.f;
}

template <int KIND>
__global__ void env_observe_kernel(int E, const float* state, float* obs) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float st[EnvSpec<KIND>::S], o[EnvSpec<KIND>::D];
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::S; ++i) st[i] = state[(size_t)e * EnvSpec<KIND>::S + i];
    env_obs<KIND>(st, o);
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::D; ++i) obs[(size_t)e * EnvSpec<KIND>::D + i] = o[i];
}

template <int KIND>
__global__ void env_step_kernel(int E, uint64_t seed0, int episode_len, int fixed_len, int action_start,
                                const void* actions, float* state, int32_t* step_count, uint32_t* episode,
                                uint32_t* gstep, float* rewards, uint8_t* term, uint8_t* trunc, float* terminal_obs, MonitorArgs mon) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    float st[S];
#pragma unroll
    for (int i = 0; i < S; ++i) st[i] = state[(size_t)e * S + i];
    int ai = 0; float af = 0.f;
    if (EnvSpec<KIND>::discrete) ai = ((const int32_t*)actions)[e] - action_start; else af = ((const float*)actions)[e];
    bool t;
    const float r = env_step<KIND>(st, af, ai, fixed_len != 0, &t);
    const int sc = step_count[e] + 1;
    const bool tr = sc >= episode_len;
    rewards[e] = r; term[e] = t; trunc[e] = tr; gstep[e] += 1;
    if (mon.cur_ret) {                                             // MonitorWrapperEnv.act! (monitorWrapperEnv.jl:46-60), raw reward
        const float cr = mon.cur_ret[e] + r; const int cl = mon.cur_len[e] + 1;
        if (t || tr) { mon.ep_ret[e] = cr; mon.ep_len[e] = cl; mon.cur_ret[e] = 0.f; mon.cur_len[e] = 0; } else { mon.cur_ret[e] = cr; mon.cur_len[e] = cl; }
        mon.flags_out[e] = (uint8_t)((t ? 1 : 0) | (tr ? 2 : 0));
    }
    if (tr) { float o[D]; env_obs<KIND>(st, o);
#pragma unroll
        for (int i = 0; i < D; ++i) terminal_obs[(size_t)e * D + i] = o[i]; }
    if (t || tr) { const uint32_t ep = episode[e] + 1; episode[e] = ep; step_count[e] = 0; env_reset<KIND>(seed0 + (uint64_t)e, ep, st); }
    else step_count[e] = sc;
#pragma unroll
    for (int i = 0; i < S; ++i) state[(size_t)e * S + i] = st[i];
}

// =============================================================================================
// NormalizeWrapperEnv on device (src/environment_wrappers/normalizeWrapperEnv.jl): batch moments over the env axis,
// parallel-Welford merge (update_from_moments! :28-50), normalise + clip (:174-197).  Two launches per statistic:
// *_partials (per-block f64 sums) and *_apply (every block folds the partials in the same order and merges; block 0
// persists the new RunningMeanStd into the other half of the ping-pong state).
// =============================================================================================

template <int KIND>
__global__ void obs_partials_kernel(int E, const float* __restrict__ state, float* __restrict__ raw, double* __restrict__ partials) {
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    __shared__ double sh[16];
    double s[D], q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = 0; q[d] = 0; }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        float st[S], o[D];
#pragma unroll
        for (int i = 0; i < S; ++i) st[i] = state[(size_t)e * S + i];
        env_obs<KIND>(st, o);
#pragma unroll
        for (int d = 0; d < D; ++d) { raw[(size_t)e * D + d] = o[d]; s[d] += o[d]; q[d] += (double)o[d] * o[d]; }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double ss = block_sum_f64(s[d], sh), qq = block_sum_f64(q[d], sh);
        if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * 16 + 2 * d] = ss; partials[(size_t)blockIdx.x * 16 + 2 * d + 1] = qq; }
    }
}

// update_from_moments! (normalizeWrapperEnv.jl:28-50) in the reference's f32 arithmetic
__device__ __forceinline__ void rms_merge(float& mean, float& var, long long count, float bmean, float bvar, long long bcount) {
    if (count == 0) { mean = bmean; var = bvar; }
    else {
        const long long tot = count + bcount;
        const float delta = bmean - mean;
        const float new_mean = mean + delta * (float)bcount / (float)tot;
        const float m_a = var * (float)count, m_b = bvar * (float)bcount;
        const float M2 = m_a + m_b + delta * delta * (float)count * (float)bcount / (float)tot;
        mean = new_mean; var = M2 / (float)tot;
    }
}

// column sums of the [nblocks][16] partial table with all 256 threads (16 columns x 16 block groups, independent loads in flight, fixed
// summation order): the serial `for b < nblocks` fold by D + 1 threads cost one dependent global round trip per block — 57 us per env step
__device__ __forceinline__ void fold_partials16(const double* __restrict__ partials, int nblocks, double (&s_part)[16][17], double (&s_col)[16]) {
    const int col = threadIdx.x & 15, seg = threadIdx.x >> 4;
    double t = 0;
#pragma unroll 4
    for (int b = seg; b < nblocks; b += 16) t += partials[(size_t)b * 16 + col];
    s_part[seg][col] = t;
    __syncthreads();
    if (threadIdx.x < 16) {
        double u = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) u += s_part[g][threadIdx.x];
        s_col[threadIdx.x] = u;
    }
    __syncthreads();
}

// data-parallel runs: the [nblocks][16] table of one rank's partial sums folded to ONE row, which RCCL then sums over ranks — the batch moments of
// NormalizeWrapperEnv cover every env of the job, as in the reference's single vector env (normalizeWrapperEnv.jl:21-26,139-171)
__global__ void fold_partials_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out16) {
    __shared__ double s_part[16][17], s_col[16];
    fold_partials16(partials, nblocks, s_part, s_col);
    if (threadIdx.x < 16) out16[threadIdx.x] = s_col[threadIdx.x];
}

__global__ void norm_obs_apply_kernel(NormObsArgs a) {
    __shared__ float s_mean[8], s_var[8];
    __shared__ double s_part[16][17], s_col[16];
    if (a.update) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;          // envs behind the batch moments: all ranks' when the partials were all-reduced
    if (threadIdx.x < a.D) {
        const int d = threadIdx.x;
        float mean = a.in->mean[d], var = a.in->var[d];
        if (a.update) {
            const double s = s_col[2 * d], q = s_col[2 * d + 1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;      // mean / var(corrected=false), :21-26
            rms_merge(mean, var, a.in->count, (float)bm, (float)bv, nb_);
        }
        s_mean[d] = mean; s_var[d] = var;
        if (blockIdx.x == 0) { a.out->mean[d] = mean; a.out->var[d] = var; if (d == 0) a.out->count = a.in->count + (a.update ? nb_ : 0); }
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.E * a.D; i += gridDim.x * blockDim.x) {
        const int d = i % a.D;
        float v = a.raw[i];
        if (a.norm_obs) { v = (v - s_mean[d]) / sqrtf(s_var[d] + a.eps); v = fminf(fmaxf(v, -a.clip), a.clip); }   // normalize_obs! :174-179
        a.obs_n[i] = v;
    }
}

__global__ void rew_partials_kernel(int E, const float* __restrict__ rew_raw, float* __restrict__ disc, float gamma, int update,
                                    double* __restrict__ partials) {
    __shared__ double sh[16];
    double s = 0, q = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        float r = disc[e];
        if (update) { r = r * gamma + rew_raw[e]; disc[e] = r; }                        // update_reward_stats! :167-171
        s += r; q += (double)r * r;
    }
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * 16] = s; partials[(size_t)blockIdx.x * 16 + 1] = q; }
}

__global__ void norm_rew_apply_kernel(NormRewArgs a) {
    __shared__ float s_var;
    __shared__ double s_part[16][17], s_col[16];
    if (a.update) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;
    if (threadIdx.x == 0) {
        float mean = a.in->mean[0], var = a.in->var[0];
        if (a.update) {
            const double s = s_col[0], q = s_col[1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;
            rms_merge(mean, var, a.in->count, (float)bm, (float)bv, nb_);
        }
        s_var = var;
        if (blockIdx.x == 0) { a.out->mean[0] = mean; a.out->var[0] = var; a.out->count = a.in->count + (a.update ? nb_ : 0); }
    }
    __syncthreads();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float r = a.rew_raw[e];
        if (a.norm_reward) { r = r / sqrtf(s_var + a.eps); r = fminf(fmaxf(r, -a.clip_reward), a.clip_reward); }   // normalize_rewards! :188-197 (no mean subtraction)
        a.rew_out[e] = r;
        if (a.flags_out) a.flags_out[e] = (uint8_t)((a.term[e] ? 1 : 0) | (a.trunc[e] ? 2 : 0));
        if (a.term[e] || a.trunc[e]) a.disc_returns[e] = 0.f;                                                    // :152-155
        if (a.norm_obs && a.trunc[e]) {                                                                          // terminal_observation, :157-163
            for (int d = 0; d < a.D; ++d) {
                float v = (a.tobs[(size_t)e * a.D + d] - a.obs_stats->mean[d]) / sqrtf(a.obs_stats->var[d] + a.eps);
                a.tobs[(size_t)e * a.D + d] = fminf(fmaxf(v, -a.clip_obs), a.clip_obs);
            }
        }
    }
}

// fused step of the normalised rollout: one launch does act! for every env (physics, flags, terminal_observation, auto-reset,
// discounted-return update) AND the per-block partial moments of the new observations and of the discounted returns;
// norm_apply_kernel then merges both RunningMeanStd states and normalises rewards, terminal observations (OLD obs statistics,
// normalizeWrapperEnv.jl:157-163) and the next observations (NEW statistics, :123-137).  7 launches per env step -> 3.
template <int KIND>
__global__ void norm_step_kernel(NormStepArgs a) {
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    __shared__ double sh[16];
    double acc[2 + 2 * D];
#pragma unroll
    for (int i = 0; i < 2 + 2 * D; ++i) acc[i] = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float st[S];
#pragma unroll
        for (int i = 0; i < S; ++i) st[i] = a.state[(size_t)e * S + i];
        int ai = 0; float af = 0.f;
        if (EnvSpec<KIND>::discrete) ai = ((const int32_t*)a.actions)[e] - a.action_start; else af = ((const float*)a.actions)[e];
        bool t;
        const float r = env_step<KIND>(st, af, ai, a.fixed_len != 0, &t);
        const int sc = a.step_count[e] + 1;
        const bool tr = sc >= a.episode_len;
        a.rew_raw[e] = r; a.term[e] = t; a.trunc[e] = tr; a.gstep[e] += 1;
        if (a.flags_out) a.flags_out[e] = (uint8_t)((t ? 1 : 0) | (tr ? 2 : 0));
        if (a.mon_cur_ret) {                                       // MonitorWrapperEnv sits inside the normaliser: raw reward
            const float cr = a.mon_cur_ret[e] + r; const int cl = a.mon_cur_len[e] + 1;
            if (t || tr) { a.ep_ret[e] = cr; a.ep_len[e] = cl; a.mon_cur_ret[e] = 0.f; a.mon_cur_len[e] = 0; } else { a.mon_cur_ret[e] = cr; a.mon_cur_len[e] = cl; }
        }
        float disc = a.disc_returns[e];
        if (a.update_ret) { disc = disc * a.gamma + r; a.disc_returns[e] = disc; }       // update_reward_stats! :167-171 (reset of done envs happens in norm_apply_kernel)
        acc[0] += disc; acc[1] += (double)disc * disc;
        if (tr) { float o[D]; env_obs<KIND>(st, o);
#pragma unroll
            for (int i = 0; i < D; ++i) a.tobs_raw[(size_t)e * D + i] = o[i]; }
        if (t || tr) { const uint32_t ep = a.episode[e] + 1; a.episode[e] = ep; a.step_count[e] = 0; env_reset<KIND>(a.seed0 + (uint64_t)e, ep, st); }
        else a.step_count[e] = sc;
#pragma unroll
        for (int i = 0; i < S; ++i) a.state[(size_t)e * S + i] = st[i];
        float o[D]; env_obs<KIND>(st, o);
#pragma unroll
        for (int d = 0; d < D; ++d) { a.obs_raw[(size_t)e * D + d] = o[d]; acc[2 + 2 * d] += o[d]; acc[3 + 2 * d] += (double)o[d] * o[d]; }
    }
#pragma unroll
    for (int i = 0; i < 2 + 2 * D; ++i) {
        const double v = block_sum_f64(acc[i], sh);
        if (threadIdx.x == 0) a.partials[(size_t)blockIdx.x * 16 + i] = v;
    }
}

__global__ void norm_apply_kernel(NormApplyArgs a) {
    __shared__ float s_mean_new[8], s_var_new[8], s_mean_old[8], s_var_old[8], s_rvar;
    __shared__ double s_part[16][17], s_col[16];
    if (a.update_ret || a.update_obs) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;
    if (threadIdx.x <= a.D) {
        const int i = threadIdx.x;                       // 0: discounted returns, 1..D: observation dims
        const RmsState* in = i == 0 ? a.ret_in : a.obs_in;
        const int d = i == 0 ? 0 : i - 1;
        float mean = in->mean[d], var = in->var[d];
        const float mean_old = mean, var_old = var;
        const int upd = i == 0 ? a.update_ret : a.update_obs;
        if (upd) {
            const int col = i == 0 ? 0 : 2 * i;
            const double s = s_col[col], q = s_col[col + 1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;
            rms_merge(mean, var, in->count, (float)bm, (float)bv, nb_);
        }
        if (i == 0) { s_rvar = var; if (blockIdx.x == 0) { a.ret_out->mean[0] = mean; a.ret_out->var[0] = var; a.ret_out->count = in->count + (upd ? nb_ : 0); } }
        else {
            s_mean_new[d] = mean; s_var_new[d] = var; s_mean_old[d] = mean_old; s_var_old[d] = var_old;
            if (blockIdx.x == 0) { a.obs_out->mean[d] = mean; a.obs_out->var[d] = var; if (d == 0) a.obs_out->count = in->count + (upd ? nb_ : 0); }
        }
    }
    __syncthreads();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float r = a.rew_raw[e];
        if (a.norm_reward) { r = r / sqrtf(s_rvar + a.eps); r = fminf(fmaxf(r, -a.clip_reward), a.clip_reward); }
        a.rew_out[e] = r;
        const bool tr = a.trunc[e] != 0;
        if (a.term[e] || tr) a.disc_returns[e] = 0.f;
        for (int d = 0; d < a.D; ++d) {
            float v = a.obs_raw[(size_t)e * a.D + d];
            if (a.norm_obs) { v = (v - s_mean_new[d]) / sqrtf(s_var_new[d] + a.eps); v = fminf(fmaxf(v, -a.clip_obs), a.clip_obs); }
            a.obs_n[(size_t)e * a.D + d] = v;
            if (a.norm_obs && tr) {
                float tv = (a.tobs[(size_t)e * a.D + d] - s_mean_old[d]) / sqrtf(s_var_old[d] + a.eps);
                a.tobs[(size_t)e * a.D + d] = fminf(fmaxf(tv, -a.clip_obs), a.clip_obs);
            }
        }
    }
}

// MonitorWrapperEnv's CircularBuffer of the last W finished episodes (monitorWrapperEnv.jl:1-7,53-58), kept on device.
// Episodes finish in (step, env) order; after a rollout the collector pushes this rollout's events in that order, which
// only requires the LAST min(n_events, W) of them: count per step, suffix-sum to the first contributing step, then a
// block-wide prefix scan over the few rows that matter.
__global__ void monitor_count_kernel(const uint8_t* __restrict__ flags, int E, int T, int* __restrict__ cnt) {
    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        int c = 0;
        for (int e = threadIdx.x; e < E; e += blockDim.x) c += flags[(size_t)t * E + e] != 0;
        __shared__ int sh[16];
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w]; cnt[t] = s; }
        __syncthreads();
    }
}
__global__ void monitor_collect_kernel(const uint8_t* __restrict__ flags, const float* __restrict__ ep_ret, const int32_t* __restrict__ ep_len,
                                       int E, int T, int W, const int* __restrict__ cnt, float* ring_ret, int32_t* ring_len, int* meta) {
    __shared__ int s_t0, s_skip, s_total, s_base, wsum[16];
    if (threadIdx.x == 0) {
        int total = 0, t0 = T;
        for (int t = T - 1; t >= 0; --t) { if (total >= W) break; total += cnt[t]; t0 = t; }
        int all = 0; for (int t = 0; t < T; ++t) all += cnt[t];
        s_t0 = t0; s_total = all; s_skip = total > W ? total - W : 0; s_base = 0;
    }
    __syncthreads();
    if (s_total == 0) return;
    const int head0 = meta[1];
    for (int t = s_t0; t < T; ++t) {
        for (int e0 = 0; e0 < E; e0 += blockDim.x) {
            const int e = e0 + threadIdx.x;
            const int f = (e < E && flags[(size_t)t * E + e] != 0) ? 1 : 0;
            int incl = f;                                                            // inclusive scan within the wave
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            int off = 0; for (int w = 0; w < wv; ++w) off += wsum[w];
            int chunk = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) chunk += wsum[w];
            const int gi = s_base + off + incl - f;                                  // index of this event among the rows t0..T-1
            if (f && gi >= s_skip) { const int pos = (head0 + gi - s_skip) % W; ring_ret[pos] = ep_ret[(size_t)t * E + e]; ring_len[pos] = ep_len[(size_t)t * E + e]; }
            __syncthreads();
            if (threadIdx.x == 0) s_base += chunk;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const int pushed = s_base - s_skip;
        meta[1] = (head0 + pushed) % W;
        const int c = meta[0] + pushed; meta[0] = c > W ? W : c;
    }
}

// (distribution heads: dril_heads.h)
// first-layer B operand from an observation held in registers: xk[s] = obs[2s + h] (static register indices only)
template <int D> __device__ __forceinline__ void pair_obs(const float (&obs)[D], int h, float (&xk)[FirstLayer<D>::KS]) {
#pragma unroll
    for (int s = 0; s < FirstLayer<D>::KS; ++s) {
        const float v0 = (2 * s < D) ? obs[(2 * s < D) ? 2 * s : 0] : 0.f;
        const float v1 = (2 * s + 1 < D) ? obs[(2 * s + 1 < D) ? 2 * s + 1 : 0] : 0.f;
        xk[s] = h ? v1 : v0;
    }
}

template <int D, int H, int O> struct NetLdsSplit {
    static_assert(H == 64, "the split kernel is laid out for hidden_dims [64,64]");
    static constexpr int DP = FirstLayer<D>::DP, OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, B3 = W3S + O * H, SMALL_END = (B3 + OP + 3) / 4 * 4;
    static constexpr int W2P = SMALL_END;                    // two f16 pieces x [64][64] = 2 x 8192 bytes
    static constexpr int END = W2P + 2 * H * H / 2;          // in floats
};
// 16-byte slot swizzle of the forward's W2 piece image (128-byte rows; lane c reads slot 2 j + h of row c with one ds_read_b128: the row-read pattern of the update kernels'
// images, dril_split_pieces.h wimg_g<64>).  A slot holds the EIGHT k values half-wave h contracts in k16 step j — k = 16 j + 4 h + {0..3} and 16 j + 8 + 4 h + {0..3}, the k order
// of a register B operand — so the A fragment is one 16-byte read per piece.  (Until round 4 the row was in plain k order: two 8-byte reads per piece, regrouped into the
// operand's four consecutive registers by 4 - 6 v_mov per MFMA group: 160 of the ~1 200 vector instructions of a rollout step.)
__device__ __forceinline__ int w2img_gw(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int timg_gs(int r) { return (((r >> 1) & 1) << 3) | (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | (r & 1); }

template <int D, int H, int O>
__device__ inline void stage_net_split(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    using L = NetLdsSplit<D, H, O>;
    for (int i = tid; i < L::DP * H; i += nthreads) { const int o = i % H, k = i / H; lds[L::W1T + k * H + o] = k < D ? kTanhScale * P[n.w1 + o + k * H] : 0.0f; }
    for (int i = tid; i < H; i += nthreads) { lds[L::B1 + i] = kTanhScale * P[n.b1 + i]; lds[L::B2 + i] = (kTanhScale * kWScale * kActScale) * P[n.b2 + i]; }   // b2 starts the SCALED accumulator of L2 (f16 pieces, dril_device.h)
    for (int i = tid; i < O * H; i += nthreads) { const int o = i % O, k = i / O; lds[L::W3S + o * H + k] = P[n.w3 + i]; }
    for (int i = tid; i < L::OP; i += nthreads) lds[L::B3 + i] = i < O ? P[n.b3 + i] : 0.0f;
    char* img = reinterpret_cast<char*>(lds + L::W2P);
    for (int i = tid; i < H * H / 2; i += nthreads) {         // pair (k, k+1) of row o: W2 is column-major (out x in), so consecutive threads read consecutive o
        const int o = i % H, kp = i / H;
        unsigned hi, lo;
        split2_pair((kTanhScale * kWScale) * P[n.w2 + o + H * (2 * kp)], (kTanhScale * kWScale) * P[n.w2 + o + H * (2 * kp + 1)], hi, lo);
        const int j = kp >> 3, x = (kp >> 1) & 3;                  // k16 step; quarter of the step (k = 16 j + 4 x + 2 (kp & 1) + {0, 1})
        const int byte = o * 128 + (((2 * j + (x & 1)) ^ w2img_gw(o)) << 4) + ((x >> 1) << 3) + ((kp & 1) << 2);
        *reinterpret_cast<unsigned*>(img + byte) = hi; *reinterpret_cast<unsigned*>(img + 8192 + byte) = lo;
    }
}

__device__ __forceinline__ f16x8 chunk_frag(const unsigned (&pc)[2][4], int p) { return __builtin_bit_cast(f16x8, u32x4{pc[p][0], pc[p][1], pc[p][2], pc[p][3]}); }
// ---- forward of one [64,64] net on the f16 matrix cores (fp32-equivalent two-piece split, as the gradient kernels; dril_device.h): rollout_kernel / policy_kernel -----
// L1 on the f32 MFMA (K = 4), tanh and split of kActScale h1 a k16 step at a time, L2 as three v_mfma_f32_32x32x16_f16 per step against the swizzled W2 piece image
// (kTanhScale kWScale W2), the scales undone before the second tanh, L3 on the VALU.  Against the f32-MFMA forward (64 x 64 cycles on the VALU's lanes per net and
// tile) this is 24 x 32 cycles of matrix pipe + ~150 VALU instructions (round 2 - 3: three bf16 pieces, 48 MFMAs).
template <int D, int H, int O>
__device__ __forceinline__ void net_forward_split(const float* __restrict__ lds, const float (&xk)[FirstLayer<D>::KS], float (&out)[O], int lane) {
    using L = NetLdsSplit<D, H, O>;
    constexpr int MT = H / 32;
    const int c = lane & 31, h = lane >> 5;
    const char* Wimg = reinterpret_cast<const char*>(lds + L::W2P);
    const int wf_base = c * 128 + ((h ^ w2img_gw(c)) << 4);
    f32x16 h1[MT], acc[MT];
    dense_first<H, MT, FirstLayer<D>::KS>(lds + L::W1T, lds + L::B1, xk, h1, lane);
#pragma unroll
    for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B2 + 32 * mo + 8 * q + 4 * h);
            acc[mo][4 * q + 0] = b[0]; acc[mo][4 * q + 1] = b[1]; acc[mo][4 * q + 2] = b[2]; acc[mo][4 * q + 3] = b[3];
        }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float ex[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ex[i] = __builtin_amdgcn_exp2f(h1[mi][8 * s + i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ex[i] = fmaf(-2.0f * kActScale, __builtin_amdgcn_rcpf(ex[i] + 1.0f), kActScale);   // kActScale h1
            unsigned pc[2][4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) split2_pair(ex[2 * tt], ex[2 * tt + 1], pc[0][tt], pc[1][tt]);
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                const int a0 = (wf_base ^ (64 * mi + 32 * s)) + 4096 * mo;
                f16x8 A[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    A[p] = *reinterpret_cast<const f16x8*>(Wimg + 8192 * p + a0);
                acc[mo] = mfma_split3(A[0], A[1], chunk_frag(pc, 0), chunk_frag(pc, 1), acc[mo]);
            }
        }
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mo][i] *= 1.0f / (kWScale * kActScale);       // the operand scales of L2 (powers of two: exact)
        tanh16(acc[mo]);
    }
    dense_out<MT, O, H>(lds + L::W3S, lds + L::B3, acc, out, lane);
}
// ---- forward of a wide net (hidden 128 / 256) for one 32-sample tile on the f16 matrix cores: kActScale h1 as two f16 pieces in registers (the same 16 registers per
// m-tile the f32 form holds), W2 from L2 as a pre-split fragment stream (build_wimg_split_kernel: [(mo MT + mi) 2 + s][piece][lane], kTanhScale kWScale W2, 4 bytes per
// weight like the f32 stream; in the k-order of a REGISTER B operand: accumulator registers 8 s .. 8 s + 7 of half-wave h are units 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3
// of the k16 step, not 8 h .. 8 h + 7 as in the LDS-image operand of the update kernel — hence its own stream, w2pf), three MFMAs per k16 step instead of sixteen v_mfma_f32_32x32x2_f32; each fragment register is refilled
// right after the MFMAs that consumed it, across m-tile boundaries too
template <int D, int H, int O>
__device__ __forceinline__ void net_forward_wide_split(const float* __restrict__ lds, const u32x4* __restrict__ w2p, const float (&xk)[FirstLayer<D>::KS],
                                                       float (&out)[O], int lane) {
    using L = NetLdsSmall<D, H, O>;
    constexpr int MT = H / 32;
    const int h = lane >> 5;
    u32x4 pcs[MT][2][2];                                              // [input m-tile][k16 step][piece]: the B operands of L2
    {
        f32x16 h1[MT];
        dense_first<H, MT, FirstLayer<D>::KS>(lds + L::W1T, lds + L::B1, xk, h1, lane);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            tanh16_scaled<false>(h1[mi], 1.0f);                       // kActScale h1
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) split2_pair(h1[mi][8 * s + 2 * tt], h1[mi][8 * s + 2 * tt + 1], hi[tt], lo[tt]);
                pcs[mi][s][0] = u32x4{hi[0], hi[1], hi[2], hi[3]}; pcs[mi][s][1] = u32x4{lo[0], lo[1], lo[2], lo[3]};
            }
        }
    }
    float part[O];
#pragma unroll
    for (int o = 0; o < O; ++o) part[o] = 0.f;
    const u32x4* fbase = w2p + lane;
    constexpr int F = MT * MT * 4;
    u32x4 af[4];                                                      // (s, piece) = (0, hi) (0, lo) (1, hi) (1, lo) of the current (mo, mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q] = fbase[(size_t)q * 64];
#pragma unroll 1
    for (int mo = 0; mo < MT; ++mo) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B2 + 32 * mo + 8 * q + 4 * h);   // staged as kTanhScale b2
            acc[4 * q + 0] = b[0] * (kWScale * kActScale); acc[4 * q + 1] = b[1] * (kWScale * kActScale); acc[4 * q + 2] = b[2] * (kWScale * kActScale); acc[4 * q + 3] = b[3] * (kWScale * kActScale);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                acc = mfma_split3(__builtin_bit_cast(f16x8, af[2 * s]), __builtin_bit_cast(f16x8, af[2 * s + 1]), __builtin_bit_cast(f16x8, pcs[mi][s][0]), __builtin_bit_cast(f16x8, pcs[mi][s][1]), acc);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    int f = (mo * MT + mi) * 4 + 2 * s + p + 4; f = f < F ? f : F - 4 + 2 * s + p;   // the tail re-reads the last fragments (in bounds, unused)
                    af[2 * s + p] = fbase[(size_t)f * 64];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] *= 1.0f / (kWScale * kActScale);       // the operand scales of L2 (powers of two: exact)
        tanh16(acc);
#pragma unroll
        for (int o = 0; o < O; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(lds + L::W3S + o * H + 32 * mo + 8 * q + 4 * h);
                part[o] = fmaf(w[0], acc[4 * q + 0], part[o]); part[o] = fmaf(w[1], acc[4 * q + 1], part[o]);
                part[o] = fmaf(w[2], acc[4 * q + 2], part[o]); part[o] = fmaf(w[3], acc[4 * q + 3], part[o]);
            }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) out[o] = part[o] + __shfl_xor(part[o], 32) + lds[L::B3 + o];
}
// forward of one net for a 32-sample tile: LDS-resident weights (H = 64) or the wide path (W2 streamed from L2)
// SPLIT: the f16 two-piece forward (the default; W2 inside f16's range, |W2| < kFwdSplitMaxW); !SPLIT: the f32-MFMA forward, exact for any finite weight — the host
// picks per launch (PolicyArgs / RolloutArgs::exact_f32: a W2 entry out of range, or DRIL_GRAD_VARIANT=0) and hands the matching W2 operand of wide nets in w2a
template <int D, int H, int O, bool WIDE, bool SPLIT>
__device__ __forceinline__ void eval_net(const float* __restrict__ lds, const float* __restrict__ w2a, const float (&xk)[FirstLayer<D>::KS], float (&out)[O], int lane) {
    if constexpr (WIDE && SPLIT) net_forward_wide_split<D, H, O>(lds, reinterpret_cast<const u32x4*>(w2a), xk, out, lane);
    else if constexpr (WIDE) net_forward_wide<D, H, O>(lds, w2a, xk, out, lane);
    else if constexpr (SPLIT) net_forward_split<D, H, O>(lds, xk, out, lane);
    else { f32x16 h1[H / 32], h2[H / 32]; net_forward<D, H, H, O>(lds, xk, h1, h2, out, lane); }
}
template <int D, int H, int O, bool WIDE, bool SPLIT> struct FwdLds { static constexpr int SIZE = WIDE ? NetLdsSmall<D, H, O>::END : SPLIT ? (NetLdsSplit<D, 64, O>::END + 3) / 4 * 4 : NetLds<D, H, H, O>::FWD_END; };
template <int D, int H, int O, bool WIDE, bool SPLIT>
__device__ __forceinline__ void stage_fwd(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    if constexpr (WIDE) stage_net_small<D, H, O>(lds, P, n, tid, nthreads);
    else if constexpr (SPLIT) stage_net_split<D, H, O>(lds, P, n, tid, nthreads);
    else stage_net<D, H, H, O, false>(lds, P, n, tid, nthreads);
}
// max |W2| over both nets as the bits of a non-negative float (atomicMax on the unsigned image orders them; a NaN's bits lie above +inf's, so it reads as "out of range"):
// what decides between the f16 two-piece forward and the f32-MFMA forward (kFwdSplitMaxW).  Enqueued behind every optimiser run, read back with its statistics.
__global__ void w2_absmax_kernel(const float* __restrict__ P, int w2_actor, int w2_critic, int HH, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * HH; i += gridDim.x * blockDim.x) {
        const float w = i < HH ? P[w2_actor + i] : P[w2_critic + i - HH];
        const unsigned b = __float_as_uint(w) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned q = __shfl_xor(m, o); m = q > m ? q : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// pre-tile W2 and W2' of one net for the wide path (see dril_device.h "wide nets")
__global__ void build_wimg_kernel(const float* __restrict__ P, NetOff off, int H, float* __restrict__ w2a, float* __restrict__ w2ta) {
    const int MT = H / 32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * H; idx += gridDim.x * blockDim.x) {
        const int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, mi = (idx >> 10) % MT, mo = (idx >> 10) / MT;
        const int o = 32 * mo + (lane & 31), k = 32 * mi + 8 * q + 4 * (lane >> 5) + c;
        w2a[idx] = kTanhScale * P[off.w2 + o + k * H];      // W2[o][k]  (column-major out x in), pre-scaled for tanh16
        w2ta[idx] = P[off.w2 + k + o * H];     // W2'[o][k] = W2[k][o]
    }
}

// =============================================================================================
// policy_kernel: host-batch / step-granular forward.  One wave = 32 samples.
// mode 0: sample + logprob + value; 1: evaluate given actions (+entropy); 2: critic only
// =============================================================================================
template <int KIND, int H, bool WIDE, bool SPLIT>
__global__ __launch_bounds__(256, WIDE ? 1 : 2) void policy_kernel(PolicyArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* la = smem; float* lc = smem + FwdLds<D, H, A, WIDE, SPLIT>::SIZE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.mode != 2) stage_fwd<D, H, A, WIDE, SPLIT>(la, a.params, a.actor, tid, 256);
    stage_fwd<D, H, 1, WIDE, SPLIT>(lc, a.params, a.critic, tid, 256);
    __syncthreads();
    const int64_t ntiles = (a.B + kTile - 1) / kTile;
    const int c = lane & 31, h = lane >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t b = tile * kTile + c;
        const bool valid = b < a.B;
        const int64_t bb = valid ? b : a.B - 1;
        if (a.only_where && !__any(valid && a.only_where[bb] != 0)) continue;     // wave-uniform skip
        if (a.boot_where) {                                                        // V(terminal_observation) of the previous env step
            const bool tr = valid && a.boot_where[bb] != 0;
            if (__any(tr)) {
                float tk[FirstLayer<D>::KS];
#pragma unroll
                for (int s = 0; s < FirstLayer<D>::KS; ++s) { const int d = 2 * s + h; tk[s] = d < D ? a.boot_obs[bb * D + d] : 0.f; }
                float bv[1];
                eval_net<D, H, 1, WIDE, SPLIT>(lc, a.w2a_critic, tk, bv, lane);
                if (tr && h == 0) a.boot_out[b] = bv[0];
            }
        }
        float xk[FirstLayer<D>::KS];
#pragma unroll
        for (int s = 0; s < FirstLayer<D>::KS; ++s) { const int d = 2 * s + h; xk[s] = d < D ? a.obs[bb * D + d] : 0.f; }
        if (a.obs_out && valid) {
#pragma unroll
            for (int s = 0; s < FirstLayer<D>::KS; ++s) { const int d = 2 * s + h; if (d < D) a.obs_out[b * D + d] = xk[s]; }
        }
        float v[1];
        eval_net<D, H, 1, WIDE, SPLIT>(lc, a.w2a_critic, xk, v, lane);
        if (valid && h == 0 && a.values) a.values[b] = v[0];
        if (a.mode == 2) continue;
        float out[A];
        eval_net<D, H, A, WIDE, SPLIT>(la, a.w2a_actor, xk, out, lane);
        if (DISC) {
            float p[A]; softmax_n<A>(out, p);
            int act;
            if (a.mode == 0) {
                double u;
                if (a.noise) u = ((const double*)a.noise)[bb];
                else if (a.gstep) { const uint64_t k = a.env_seed0 + (uint64_t)bb; uint32_t r[4]; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), a.gstep[bb], 0, 1, 0, r); u = u01_f64(r[0], r[1]); }
                else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)bb, (uint32_t)(bb >> 32), 3, a.call_counter, r); u = u01_f64(r[0], r[1]); }
                act = categorical_sample<A>(p, u);
                if (a.deterministic) {                                   // mode(d) = argmax(p) (first maximum), categorical.jl:42-44
                    act = 0; float best = p[0];
#pragma unroll
                    for (int i = 1; i < A; ++i) if (p[i] > best) { best = p[i]; act = i; }
                }
                if (valid && h == 0) ((int32_t*)a.actions)[b] = act + a.action_start;
            } else act = ((const int32_t*)a.actions)[bb] - a.action_start;
            if (valid && h == 0) {
                a.logp[b] = flog(pick<A>(p, act));
                if (a.mode == 1 && a.entropy) a.entropy[b] = categorical_entropy<A>(p);
            }
        } else {
            const float* ls = a.params + a.log_std_off;
            float x[A];
            if (a.mode == 0) {
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    float z;
                    if (a.noise) z = ((const float*)a.noise)[bb * A + i];
                    else if (a.gstep) { const uint64_t k = a.env_seed0 + (uint64_t)bb; uint32_t r[4]; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), a.gstep[bb], 0, 1, (uint32_t)(i / 2), r); z = (i & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                    else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)bb, (uint32_t)(bb >> 32), 3 + 16 * (uint32_t)i, a.call_counter, r); z = randn_f32(r[0], r[1]); }
                    x[i] = a.deterministic ? out[i] : out[i] + fexp(ls[i]) * z;            // mode(d) = mean, diagGaussian.jl:45-47
                    if (valid && h == 0) ((float*)a.actions)[b * A + i] = x[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) x[i] = ((const float*)a.actions)[bb * A + i];
            }
            if (valid && h == 0) {
                a.logp[b] = gauss_logpdf<A>(x, out, ls);
                if (a.mode == 1 && a.entropy) a.entropy[b] = gauss_entropy<A>(ls);
            }
        }
    }
}

// =============================================================================================
// rollout_kernel: persistent collect_trajectories (trajectory.jl:22-78).  A wave owns 32 envs for all
// T steps: env state lives in registers, both nets' weights in LDS, only buffer writes touch HBM.
// Buffer layout: time-major, index k = t*E + e; every store of a wave is one full 128-byte line
// (or 512 B for the float4 observation rows).
// =============================================================================================
template <int KIND, int H, bool WIDE, bool SPLIT>
__global__ __launch_bounds__(256, WIDE ? 1 : 2) void rollout_kernel(RolloutArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A, S = EnvSpec<KIND>::S;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* la = smem; float* lc = smem + FwdLds<D, H, A, WIDE, SPLIT>::SIZE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_fwd<D, H, A, WIDE, SPLIT>(la, a.params, a.actor, tid, 256);
    stage_fwd<D, H, 1, WIDE, SPLIT>(lc, a.params, a.critic, tid, 256);
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int e_raw = (blockIdx.x * 4 + wave) * kTile + c;
    if ((blockIdx.x * 4 + wave) * kTile >= a.E) return;  // whole wave out of range (wave-uniform)
    const bool valid = e_raw < a.E;
    const int e = valid ? e_raw : a.E - 1;
    const bool writer = valid && h == 0;
    const uint64_t env_seed = a.env_seed0 + (uint64_t)e;

    float st[S];
#pragma unroll
    for (int i = 0; i < S; ++i) st[i] = a.state[(size_t)e * S + i];
    int sc = a.step_count[e];
    uint32_t ep = a.episode[e], gs = a.gstep[e];
    float obs[D];
    env_obs<KIND>(st, obs);
    float lsr[4] = {0.f, 0.f, 0.f, 0.f};                               // log_std in registers for the whole rollout (the parameters are read-only during it)
    if (!DISC) {
#pragma unroll
        for (int i = 0; i < (A < 4 ? A : 4); ++i) lsr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + i])));
    }
    const float* ls = lsr;
    float mon_ret = a.mon_cur_ret ? a.mon_cur_ret[e] : 0.f;           // MonitorWrapperEnv running episode return / length
    int mon_len = a.mon_cur_len ? a.mon_cur_len[e] : 0;

    for (int t = 0; t < a.T; ++t) {
        const size_t k = (size_t)t * a.E + e;
        // keep the weights in LDS: an opaque zero offset stops hipcc from hoisting ~150 loop-invariant
        // ds_reads into VGPRs (which spilled at the 256-register budget of 2 waves/SIMD)
        int zoff = 0; asm volatile("" : "+v"(zoff));
        const float* la_t = la + zoff; const float* lc_t = lc + zoff;
        float xk[FirstLayer<D>::KS];
        pair_obs<D>(obs, h, xk);
        float v[1], out[A];
        eval_net<D, H, 1, WIDE, SPLIT>(lc_t, a.w2a_critic, xk, v, lane);
        __builtin_amdgcn_sched_barrier(0);   // do not interleave the two nets: that doubles the live weight fragments
        eval_net<D, H, A, WIDE, SPLIT>(la_t, a.w2a_actor, xk, out, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sample (layer_forward.jl:10-11 / :36-37) ----
        int act_env = 0; float actf_env = 0.f; float logp;
        if (DISC) {
            float p[A]; softmax_n<A>(out, p);
            double u;
            if (a.noise) u = ((const double*)a.noise)[k];
            else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, 0, r); u = u01_f64(r[0], r[1]); }
            const int act = categorical_sample<A>(p, u);
            logp = flog(pick<A>(p, act));
            act_env = act;                                                   // DiscreteAdapter: identity (default_adapters.jl:34-38)
            if (writer) ((int32_t*)a.act)[k] = act + a.action_start;         // raw action stored (trajectory.jl:48)
        } else {
            float x[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float z;
                if (a.noise) z = ((const float*)a.noise)[k * A + i];
                else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, (uint32_t)(i / 2), r); z = (i & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                x[i] = out[i] + fexp(ls[i]) * z;
                if (writer) ((float*)a.act)[k * A + i] = x[i];
            }
            logp = gauss_logpdf<A>(x, out, ls);
            actf_env = fminf(fmaxf(x[0], -act_bound<KIND>()), act_bound<KIND>());   // ClampAdapter on action_space(env) (default_adapters.jl:4-11)
        }
        if (writer) {
            if (D == 4) *reinterpret_cast<float4*>(a.obs + k * 4) = make_float4(obs[0], obs[1], obs[2], obs[3]);
            else {
#pragma unroll
                for (int i = 0; i < D; ++i) a.obs[k * D + i] = obs[i];
            }
            a.val[k] = v[0]; a.logp[k] = logp;
        }
        // ---- act! with auto-reset (multithreadedParallelEnv.jl:56-71) ----
        bool term;
        const float rew = env_step<KIND>(st, actf_env, act_env, a.fixed_len != 0, &term);
        sc += 1; gs += 1;
        const bool trunc = sc >= a.episode_len;
        if (__any(trunc && valid)) {                                         // V(terminal_observation), trajectory.jl:57-61
            float tobs[D]; env_obs<KIND>(st, tobs);
            float tk[FirstLayer<D>::KS];
            pair_obs<D>(tobs, h, tk);
            float bv[1];
            eval_net<D, H, 1, WIDE, SPLIT>(lc_t, a.w2a_critic, tk, bv, lane);
            if (writer && trunc) a.boot[k] = bv[0];
        }
        mon_ret += rew; mon_len += 1;
        if (term || trunc) {
            ep += 1; sc = 0; env_reset<KIND>(env_seed, ep, st);
            if (writer && a.ep_ret) { a.ep_ret[k] = mon_ret; a.ep_len[k] = mon_len; }     // finished episode (monitorWrapperEnv.jl:53-58)
            mon_ret = 0.f; mon_len = 0;
        }
        if (writer) { a.rew[k] = rew; a.flags[k] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0)); }
        env_obs<KIND>(st, obs);                                              // observe(env), trajectory.jl:45
    }
    {   // V(new_obs) for rollout-limited trajectories, trajectory.jl:65-70 (computed for every env; GAE uses it when needed)
        float xk[FirstLayer<D>::KS];
        pair_obs<D>(obs, h, xk);
        float v[1];
        eval_net<D, H, 1, WIDE, SPLIT>(lc, a.w2a_critic, xk, v, lane);
        if (writer) a.last_values[e] = v[0];
    }
    if (writer) {
#pragma unroll
        for (int i = 0; i < S; ++i) a.state[(size_t)e * S + i] = st[i];
        a.step_count[e] = sc; a.episode[e] = ep; a.gstep[e] = gs;
        if (a.mon_cur_ret) { a.mon_cur_ret[e] = mon_ret; a.mon_cur_len[e] = mon_len; }
    }
}

// =============================================================================================
// rollout_duo_kernel — collect_trajectories (trajectory.jl:22-78) for env counts that do not fill the chip (E <= 16 384: the reference's own scale is 4 envs).  In
// rollout_kernel ONE wave walks both nets for its 32 envs, step after step: with a handful of envs the rollout is a single dependent chain of ~10 k cycles per env step
// (critic forward -> actor forward -> sample -> physics).  Here a tile of 32 envs has TWO waves: wave 0 runs the actor, the sampling and the simulator, wave 1 the critic
// (value of every observation, V(terminal_observation) of truncated steps, the last values), one step behind through a double-buffered observation slot in LDS and ONE
// barrier per env step.  Same device functions as rollout_kernel (eval_net, env_step, heads), so every stored number is bit-identical to the one-wave kernel.
// =============================================================================================
template <int KIND, int H, bool SPLIT>
__global__ __launch_bounds__(128, 2) void rollout_duo_kernel(RolloutArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A, S = EnvSpec<KIND>::S;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    constexpr int LA = FwdLds<D, H, A, false, SPLIT>::SIZE, LC = FwdLds<D, H, 1, false, SPLIT>::SIZE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* la = smem; float* lc = smem + LA;
    float* slot = smem + LA + LC;                                       // [2 parities][obs D x 32 | terminal obs D x 32 | truncated flag 32]
    constexpr int SLOT = (2 * D + 1) * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_fwd<D, H, A, false, SPLIT>(la, a.params, a.actor, tid, 128);
    stage_fwd<D, H, 1, false, SPLIT>(lc, a.params, a.critic, tid, 128);
    const int c = lane & 31, h = lane >> 5;
    const int e_raw = blockIdx.x * kTile + c;
    const bool valid = e_raw < a.E;
    const int e = valid ? e_raw : a.E - 1;
    const bool writer = valid && h == 0;
    if (wave == 0) {
        // ================= actor + simulator =================
        const uint64_t env_seed = a.env_seed0 + (uint64_t)e;
        float st[S];
#pragma unroll
        for (int i = 0; i < S; ++i) st[i] = a.state[(size_t)e * S + i];
        int sc = a.step_count[e];
        uint32_t ep = a.episode[e], gs = a.gstep[e];
        float obs[D];
        env_obs<KIND>(st, obs);
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) slot[i * 32 + c] = obs[i];     // parity 0: the observation of step 0
        }
        float lsr[4] = {0.f, 0.f, 0.f, 0.f};
        if (!DISC) {
#pragma unroll
            for (int i = 0; i < (A < 4 ? A : 4); ++i) lsr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + i])));
        }
        const float* ls = lsr;
        float mon_ret = a.mon_cur_ret ? a.mon_cur_ret[e] : 0.f;
        int mon_len = a.mon_cur_len ? a.mon_cur_len[e] : 0;
        __syncthreads();                                                // weights staged, slot 0 published
        for (int t = 0; t < a.T; ++t) {
            const size_t k = (size_t)t * a.E + e;
            int zoff = 0; asm volatile("" : "+v"(zoff));
            const float* la_t = la + zoff;
            float xk[FirstLayer<D>::KS];
            pair_obs<D>(obs, h, xk);
            float out[A];
            eval_net<D, H, A, false, SPLIT>(la_t, a.w2a_actor, xk, out, lane);
            int act_env = 0; float actf_env = 0.f; float logp;
            if (DISC) {
                float p[A]; softmax_n<A>(out, p);
                double u;
                if (a.noise) u = ((const double*)a.noise)[k];
                else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, 0, r); u = u01_f64(r[0], r[1]); }
                const int act = categorical_sample<A>(p, u);
                logp = flog(pick<A>(p, act));
                act_env = act;
                if (writer) ((int32_t*)a.act)[k] = act + a.action_start;
            } else {
                float x[A];
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    float z;
                    if (a.noise) z = ((const float*)a.noise)[k * A + i];
                    else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, (uint32_t)(i / 2), r); z = (i & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                    x[i] = out[i] + fexp(ls[i]) * z;
                    if (writer) ((float*)a.act)[k * A + i] = x[i];
                }
                logp = gauss_logpdf<A>(x, out, ls);
                actf_env = fminf(fmaxf(x[0], -act_bound<KIND>()), act_bound<KIND>());
            }
            if (writer) {
                if (D == 4) *reinterpret_cast<float4*>(a.obs + k * 4) = make_float4(obs[0], obs[1], obs[2], obs[3]);
                else {
#pragma unroll
                    for (int i = 0; i < D; ++i) a.obs[k * D + i] = obs[i];
                }
                a.logp[k] = logp;
            }
            bool term;
            const float rew = env_step<KIND>(st, actf_env, act_env, a.fixed_len != 0, &term);
            sc += 1; gs += 1;
            const bool trunc = sc >= a.episode_len;
            float* sl = slot + ((t + 1) & 1) * SLOT;                    // what the critic reads during step t + 1
            if (h == 0) {
                float tobs[D]; env_obs<KIND>(st, tobs);                 // terminal_observation (only read where truncated)
#pragma unroll
                for (int i = 0; i < D; ++i) sl[(D + i) * 32 + c] = tobs[i];
                sl[2 * D * 32 + c] = (trunc && valid) ? 1.0f : 0.0f;
            }
            mon_ret += rew; mon_len += 1;
            if (term || trunc) {
                ep += 1; sc = 0; env_reset<KIND>(env_seed, ep, st);
                if (writer && a.ep_ret) { a.ep_ret[k] = mon_ret; a.ep_len[k] = mon_len; }
                mon_ret = 0.f; mon_len = 0;
            }
            if (writer) { a.rew[k] = rew; a.flags[k] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0)); }
            env_obs<KIND>(st, obs);
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < D; ++i) sl[i * 32 + c] = obs[i];
            }
            lds_barrier();                                              // step t published; the critic is done with the other slot (LDS only: the buffer stores stay in flight)
        }
        if (writer) {
#pragma unroll
            for (int i = 0; i < S; ++i) a.state[(size_t)e * S + i] = st[i];
            a.step_count[e] = sc; a.episode[e] = ep; a.gstep[e] = gs;
            if (a.mon_cur_ret) { a.mon_cur_ret[e] = mon_ret; a.mon_cur_len[e] = mon_len; }
        }
    } else {
        // ================= critic: V(obs_t), V(terminal_observation) of step t - 1, the last values =================
        __syncthreads();
        for (int t = 0; t <= a.T; ++t) {
            const float* sl = slot + (t & 1) * SLOT;
            int zoff = 0; asm volatile("" : "+v"(zoff));
            const float* lc_t = lc + zoff;
            float obs[D];
#pragma unroll
            for (int i = 0; i < D; ++i) obs[i] = sl[i * 32 + c];
            float xk[FirstLayer<D>::KS], v[1];
            pair_obs<D>(obs, h, xk);
            eval_net<D, H, 1, false, SPLIT>(lc_t, a.w2a_critic, xk, v, lane);
            if (t < a.T) { if (writer) a.val[(size_t)t * a.E + e] = v[0]; }
            else if (writer) a.last_values[e] = v[0];                  // V(new_obs) for rollout-limited trajectories, trajectory.jl:65-70
            if (t > 0) {
                const bool trunc = sl[2 * D * 32 + c] != 0.0f;          // of step t - 1
                if (__any(trunc)) {                                     // V(terminal_observation), trajectory.jl:57-61
                    float tobs[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) tobs[i] = sl[(D + i) * 32 + c];
                    float tk[FirstLayer<D>::KS], bv[1];
                    pair_obs<D>(tobs, h, tk);
                    eval_net<D, H, 1, false, SPLIT>(lc_t, a.w2a_critic, tk, bv, lane);
                    if (writer && trunc) a.boot[(size_t)(t - 1) * a.E + e] = bv[0];
                }
            }
            if (t < a.T) lds_barrier();
        }
    }
}

// =============================================================================================
// gae_scan_kernel — compute_advantages! (trajectory.jl:80-102) + returns = advantages + values (rollout_buffer.jl:87) over the time-major buffer.
//
// The recurrence A_t = delta_t + gamma lambda A_{t+1} (cut where a trajectory ends) is serial in t, and one thread per env walking all T rows (the round 1 - 3 kernel:
// 1 024 waves on 256 CUs, a dependent chain of 2 048 small loads each) left HBM at 25 % of its peak.  Here the time axis is cut into chunks of kGaeRows rows:
// a workgroup owns ONE chunk of 256 envs, issues all of its loads at once (kGaeRows x {r, V, flags}: 18 KB in flight per wave), forms every delta_t — which needs
// nothing but memory: V_{t+1} is a buffer row — and only then needs ONE number per env from the chunk that follows it in time: A at that chunk's first row.
// That carry travels through L2 as one 8-byte word per (chunk, env), {launch tag | f32 value}, written and polled with agent-scope atomics — value and validity in
// one word: no flag, no fence (the message scheme of ppo_update_small_kernel).  With the carry in hand the chunk runs the SAME fma chain the serial kernel ran, so
// every advantage is bit-identical to the serial scan — the chunking changes the schedule, not the arithmetic.
// Order / progress: later chunks get lower block indices, so a workgroup only ever waits for one that was dispatched before it; the chain of a 256-env column is
// C = T / kGaeRows hops long, but columns are independent and a hop (poll, 32 fmas, 64 stores) is short against a chunk's load time, so the machine streams.  The wait
// is bounded (kGaeSpinLimit polls); on expiry the workgroup writes NaN advantages, raises *err and leaves — a status code for the host, never a hang.
// =============================================================================================
constexpr int kGaeRows = 32;
constexpr unsigned kGaeSpinLimit = 1u << 22;
__global__ __launch_bounds__(256) void gae_scan_kernel(int E, int T, int C, int GB, float gamma, float lam, const float* __restrict__ rew,
                                                       const float* __restrict__ val, const uint8_t* __restrict__ flags,
                                                       const float* __restrict__ boot, const float* __restrict__ last_values,
                                                       float* __restrict__ adv, float* __restrict__ ret, unsigned long long* __restrict__ carry, unsigned tag, int* __restrict__ err) {
    const int n = blockIdx.x;
    const int j = C - 1 - n / GB;                               // chunk (time-later chunks first)
    const int e_raw = (n % GB) * 256 + (int)threadIdx.x;
    const bool live = e_raw < E;
    const int e = live ? e_raw : E - 1;                         // dead lanes read in bounds and store nothing
    const int t0 = j * kGaeRows;
    const int rows = (T - t0) < kGaeRows ? (T - t0) : kGaeRows;
    float r[kGaeRows], v[kGaeRows]; unsigned f[kGaeRows];
#pragma unroll
    for (int i = 0; i < kGaeRows; ++i) {                        // every load of the chunk in flight before the first use
        const int t = t0 + i < T ? t0 + i : T - 1;
        const size_t k = (size_t)t * E + e;
        r[i] = rew[k]; v[i] = val[k]; f[i] = flags[k];
    }
    const bool last_chunk = t0 + rows == T;
    const float v_after = last_chunk ? 0.f : val[(size_t)(t0 + rows) * E + e];        // V of the row that follows the chunk
    const float lv = last_values[e];
    const float gl = gamma * lam;
    unsigned done_mask = 0;
#pragma unroll
    for (int i = 0; i < kGaeRows; ++i) {                        // delta_t and the cut flag of every row (independent of each other; trajectory.jl:85-99)
        const int t = t0 + i;
        const bool term = f[i] & 1u, trunc = f[i] & 2u;
        const float vn = (i + 1 < kGaeRows) ? ((i + 1 < rows) ? v[i + 1] : v_after) : v_after;
        float delta;
        if (term) delta = r[i] - v[i];
        else if (trunc) delta = r[i] + gamma * boot[(size_t)(t < T ? t : T - 1) * E + e] - v[i];
        else if (t == T - 1) delta = r[i] + gamma * lv - v[i];
        else delta = r[i] + gamma * vn - v[i];
        r[i] = delta;
        if (term || trunc || t >= T - 1) done_mask |= 1u << i;
    }
    float a = 0.f;
    bool ok = true;
    if (!last_chunk) {                                           // A at the first row of chunk j + 1, published by its workgroup
        const unsigned long long* src = carry + (size_t)(j + 1) * E + e;
        unsigned long long w = 0; unsigned spins = 0;
        for (;;) {
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(w >> 32) == tag)) break;       // wave-uniform
            if (++spins > kGaeSpinLimit) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        a = ok ? __uint_as_float((unsigned)w) : __builtin_nanf("");
        if (!ok && (threadIdx.x & 63) == 0) atomicExch(err, 1);
    }
#pragma unroll
    for (int i = kGaeRows - 1; i >= 0; --i) {
        if (i < rows) {
            a = ((done_mask >> i) & 1u) ? r[i] : r[i] + gl * a;  // the serial kernel's expression (contracted to one fma by the compiler, as there)
            if (!ok) a = __builtin_nanf("");
            if (live) { const size_t k = (size_t)(t0 + i) * E + e; adv[k] = a; ret[k] = a + v[i]; }   // rollout_buffer.jl:87
        }
    }
    if (j > 0 && live) __hip_atomic_store(carry + (size_t)j * E + e, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// =============================================================================================
// advantage moments of one minibatch (normalize!, ppo.jl:350-356): per-block f64 partials, fixed-order finalize
// =============================================================================================
__global__ void adv_moments_kernel(MomentsArgs a) {
    __shared__ double sh[16];
    if (*a.stop_flag) return;
    double s = 0, q = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = a.pos0 + i;
        const int64_t idx = a.perm32 ? (int64_t)a.perm32[p] : a.perm ? a.perm[p] : (a.perm_bits ? perm_index(p, a.N, a.perm_key, a.perm_bits) : p);
        const int64_t li = idx - a.idx_lo;                    // this rank's shard of the buffer
        if (li >= 0 && li < a.n_local) { const double x = a.adv[li]; s += x; q += x * x; }
    }
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { a.partials[2 * blockIdx.x] = s; a.partials[2 * blockIdx.x + 1] = q; }
}
// epoch_moments_kernel: advantage sums of ALL minibatches of one epoch in a single sequential pass over the buffer
// (v1-v4 gathered adv[perm(p)] per optimiser step: 32 x 85 us of random 4-byte reads per epoch).  Each index is mapped back
// to its position in the epoch order by the inverse bijection, binned by minibatch in LDS (ds_add_f64), one partial
// table per block; epoch_moments_finalize_kernel folds the blocks in a fixed order.
// Round 3, end: per sample the pass costs the inverse bijection (eight v_mul_lo_u32), the minibatch of its position and two LDS adds.  Two things were slow (806 us
// per epoch at configs[1], 134 M samples): the 64-bit `p / B` (an emulated division, more instructions than the bijection) and two waves per SIMD for a loop whose
// iterations are one dependent chain.  Now: every size the 32-bit bijection covers (bits <= 31) divides by a reciprocal (double, one correction step: exact), the
// xorshift inverses are two shifts without a loop, and the grid is 2 048 blocks; the bins are replicated R times (copy = lane % R; a copy is nb + 1 slots of 16
// bytes) — measured to matter little (equal-address adds are not what the loop waits for) and kept for small nb.
template <bool FAST32>
__global__ void epoch_moments_kernel(const float* __restrict__ adv, int64_t N, int64_t B, double inv_b, int nb, int R, uint64_t key, int bits,
                                     double* __restrict__ block_tables, const int* stop_flag) {
    extern __shared__ double bins[];     // [R][nb + 1][2]
    if (*stop_flag) return;
    const int pitch = 2 * (nb + 1);
    for (int i = threadIdx.x; i < R * pitch; i += blockDim.x) bins[i] = 0.0;
    __syncthreads();
    double* mine = bins + (threadIdx.x % R) * pitch;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < N ? lo + per : N;
    if (FAST32) {                                                    // bits <= 31: positions, indices and B fit 31 bits
        const uint32_t mask = (1u << bits) - 1u, n32 = (uint32_t)N, b32 = (uint32_t)B;
        const int s = bits / 2 > 0 ? bits / 2 : 1, s2 = s + 1 < bits ? s + 1 : s;
        const int ss = 2 * s < 31 ? 2 * s : 31, ss2 = 2 * s2 < 31 ? 2 * s2 : 31;     // x < 2^31: a shift by 31 gives the 0 the loop of unxorshift32 stops at
        constexpr uint32_t I1 = (uint32_t)mul_inverse(0x9E3779B97F4A7C15ull), I2 = (uint32_t)mul_inverse(0xBF58476D1CE4E5B9ull);
        uint32_t kr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) kr[r] = (uint32_t)(key >> (r * 13)) & mask;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const double x = adv[i];
            uint32_t p = (uint32_t)i;
            do {                                                     // mix_bij_inv32 (dril_device.h) with the shift loops written out; cycle walking backwards
#pragma unroll
                for (int r = 3; r >= 0; --r) {
                    p = p ^ (p >> s2) ^ (p >> ss2);
                    p = (p * I2) & mask;
                    p = p ^ (p >> s) ^ (p >> ss);
                    p = ((p - 0xD192ED03u) * I1) & mask;
                    p ^= kr[r];
                }
            } while (p >= n32);
            uint32_t k = (uint32_t)((double)p * inv_b);              // p / B: off by at most one
            const uint32_t kb = k * b32;
            if (kb > p) --k; else if (p - kb >= b32) ++k;
            atomicAdd(&mine[2 * k], x); atomicAdd(&mine[2 * k + 1], x * x);
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const double x = adv[i];
            const int64_t p = perm_position(i, N, key, bits);
            const int k = (int)((uint64_t)p / (uint64_t)B);
            atomicAdd(&mine[2 * k], x); atomicAdd(&mine[2 * k + 1], x * x);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nb; i += blockDim.x) {
        double v = 0.0;
        for (int r = 0; r < R; ++r) v += bins[r * pitch + i];
        block_tables[(size_t)blockIdx.x * 2 * nb + i] = v;
    }
}
// one wave per minibatch: lane l folds blocks l, l + 64, ... in order, then the 64 lanes in a fixed tree
__global__ void epoch_moments_finalize_kernel(const double* __restrict__ block_tables, int nblocks, int nb, int64_t N, int64_t B,
                                              double* __restrict__ table3, const int* stop_flag) {
    if (*stop_flag) return;
    const int i = blockIdx.x, lane = threadIdx.x;
    double s = 0, q = 0;
    for (int b = lane; b < nblocks; b += 64) { s += block_tables[(size_t)b * 2 * nb + 2 * i]; q += block_tables[(size_t)b * 2 * nb + 2 * i + 1]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m); q += __shfl_xor(q, m); }
    if (lane == 0) {
        const int64_t pos0 = (int64_t)i * B;
        table3[3 * i] = s; table3[3 * i + 1] = q; table3[3 * i + 2] = (double)(pos0 + B <= N ? B : N - pos0);
    }
}

__global__ void moments_finalize_kernel(const double* partials, int nblocks, double* out3, double n_local, const int* stop_flag) {
    __shared__ double sh[16];
    if (*stop_flag) return;
    double s = 0, q = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s += partials[2 * i]; q += partials[2 * i + 1]; }   // fixed order per thread
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { out3[0] = s; out3[1] = q; out3[2] = n_local; }
}

// pack_records_kernel: the six per-sample fields the loss reads (ppo.jl:366-371) as one 32-byte record, so that a
// minibatch gather is ONE 16-byte load per lane (the two half-waves of a sample fetch the two halves of its record = one
// 32-B sector) instead of five scattered 4-byte loads (v1-v4: FETCH 1.6-3.1 GB per launch vs 0.22 GB algorithmic).
template <int KIND>
__global__ void pack_records_kernel(int64_t N, const float* __restrict__ obs, const void* __restrict__ act, const float* __restrict__ adv,
                                    const float* __restrict__ logp, const float* __restrict__ ret, float4* __restrict__ rec) {
    constexpr int D = EnvSpec<KIND>::D, RS = RecLayout<D>::RS;          // D <= 4: {obs0..3}{scalars}; D <= 8: {obs0..3}{obs4..7}{scalars}
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = obs[n * D + d];
        const float a = EnvSpec<KIND>::discrete ? __int_as_float(((const int32_t*)act)[n]) : ((const float*)act)[n];
        rec[RS * n] = make_float4(o[0], o[1], o[2], o[3]);
        if (RS == 3) rec[RS * n + 1] = make_float4(o[4], o[5], o[6], o[7]);
        rec[RS * n + RS - 1] = make_float4(a, adv[n], logp[n], ret[n]);
    }
}

// pre-split fragment streams of ppo_grad_wide_split_kernel: two f16 pieces per weight (dril_device.h), forward kTanhScale kWScale W2, reverse kWScale W2'
__global__ void build_wimg_split_kernel(const float* __restrict__ P, NetOff off, int H, u32x4* __restrict__ w2p, u32x4* __restrict__ w2tp, u32x4* __restrict__ w2pf) {
    const int MT = H / 32, total = MT * MT * 2 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, s = (idx >> 6) & 1, mi = (idx >> 7) % MT, mo = (idx >> 7) / MT;
        const int i = 32 * mo + (lane & 31), k0 = 32 * mi + 16 * s + 8 * (lane >> 5);
        unsigned f[2][4], b[2][4], g[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 2 * t;
            const int kf = 32 * mi + 16 * s + 8 * (t >> 1) + 4 * (lane >> 5) + 2 * (t & 1);                                                                             // register-B order of the forward kernels (net_forward_wide_split)
            split2_pair((kTanhScale * kWScale) * P[off.w2 + i + (size_t)kf * H], (kTanhScale * kWScale) * P[off.w2 + i + (size_t)(kf + 1) * H], g[0][t], g[1][t]);
            split2_pair((kTanhScale * kWScale) * P[off.w2 + i + (size_t)k * H], (kTanhScale * kWScale) * P[off.w2 + i + (size_t)(k + 1) * H], f[0][t], f[1][t]);   // W2[o][k] (column-major out x in)
            split2_pair(kWScale * P[off.w2 + k + (size_t)i * H], kWScale * P[off.w2 + k + 1 + (size_t)i * H], b[0][t], b[1][t]);                                   // W2'[i][k] = W2[k][i]
        }
        const size_t base = ((size_t)((mo * MT + mi) * 2 + s) * 2) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p) { w2p[base + (size_t)p * 64] = u32x4{f[p][0], f[p][1], f[p][2], f[p][3]}; w2tp[base + (size_t)p * 64] = u32x4{b[p][0], b[p][1], b[p][2], b[p][3]}; w2pf[base + (size_t)p * 64] = u32x4{g[p][0], g[p][1], g[p][2], g[p][3]}; }
    }
}

// =============================================================================================
// slab reduction -> flat [grads | stats] buffer; norm; clip + KL check + Adam
// flat layout: params order (actor net, critic net, log_std) then 8 stats:
//   0 sum(-min term)  1 sum(entropy)  2 sum(clipped)  3 sum(kl)  4 sum(ratio)  5 sum((V-R)^2)  6 n_samples  7 unused
// =============================================================================================
__global__ __launch_bounds__(1024) void grad_reduce_kernel(ReduceArgs a) {
    // block = 32 parameters x 32 slab groups: every thread has all its (<= 8) slab loads in flight at once; the slabs were
    // written by other XCDs, so each load is a full memory round trip and serial loops cost ~0.5 us per iteration (v2: 65 us)
    __shared__ double sh[16];
    __shared__ float part[32][33];
    if (*a.stop_flag) return;
    const int pl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + pl;
    float acc = 0.f;
    if (p < a.P) {
        const float* base; int stride, offp;
        int Gn = a.G;
        if (p < a.Pa) { base = a.slabs_actor; stride = a.slab_a; offp = p; }
        else if (p < a.Pa + a.Pc) { base = a.slabs_critic; stride = a.slab_c; offp = p - a.Pa; Gn = a.Gc; }
        else { base = a.slabs_actor; stride = a.slab_a; offp = a.Pa + (p - a.Pa - a.Pc); }   // log_std grads sit after the actor net
#pragma unroll 8
        for (int g = grp; g < Gn; g += 32) acc += base[(size_t)g * stride + offp];
    }
    part[grp][pl] = acc;
    __syncthreads();
    float gsum = 0.f;
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < 32; ++k) gsum += part[k][pl];                      // fixed order => deterministic
        if (p < a.P) a.flat[p] = gsum;
    }
    const double q = block_sum_f64(grp == 0 ? (double)gsum * (double)gsum : 0.0, sh);
    if (threadIdx.x == 0) a.norm_partials[blockIdx.x] = q;
    if (blockIdx.x == 0) {                                                     // loss/statistics sums: 8 stats x 32 slab groups
        __syncthreads();
        const int k = threadIdx.x & 7, sg = (threadIdx.x >> 3) & 31;
        float sacc = 0.f;
        if (threadIdx.x < 256) {
            if (k < 5) { for (int g = sg; g < a.G; g += 32) sacc += a.slabs_actor[(size_t)g * a.slab_a + a.slab_a - 8 + k]; }
            else if (k == 5) { for (int g = sg; g < a.Gc; g += 32) sacc += a.slabs_critic[(size_t)g * a.slab_c + a.slab_c - 8]; }
            part[sg][k] = sacc;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            double t = 0;
            for (int j = 0; j < 32; ++j) t += (double)part[j][threadIdx.x];
            if (threadIdx.x == 6) t = a.n_samples_local;
            a.flat[a.P + threadIdx.x] = (float)t;
        }
    }
}
__global__ void grad_norm_kernel(const float* flat, int P, double* norm_partials, const int* stop_flag) {
    __shared__ double sh[16];
    if (*stop_flag) return;
    const int p = blockIdx.x * 32 + (threadIdx.x & 31);          // same 32-parameter blocking as grad_reduce_kernel
    const float g = (p < P && threadIdx.x < 32) ? flat[p] : 0.f;
    const double q = block_sum_f64((double)g * (double)g, sh);
    if (threadIdx.x == 0) norm_partials[blockIdx.x] = q;
}
__global__ void adam_kernel(AdamArgs a) {
    if (*a.stop_flag) return;
    __shared__ double shn[16];
    double ps = 0;
    if (a.norm_from_flat) { for (int i = threadIdx.x; i < a.P; i += blockDim.x) { const double g = a.flat[i]; ps += g * g; } }   // 9 k values per block: cheaper than a launch
    else for (int i = threadIdx.x; i < a.n_partials; i += blockDim.x) ps += a.norm_partials[i];    // same order in every block
    const float norm = sqrtf((float)block_sum_f64(ps, shn));
    const float* stf = a.flat + a.P;
    const float n = a.use_stats ? stf[6] : 1.f;
    const float kl = a.use_stats ? stf[3] / n : 0.f;
    const bool bad = !(norm == norm) || isinf(norm);                              // NaN/Inf anywhere poisons the norm (ppo.jl:213-214)
    const bool kl_stop = a.use_stats && a.has_target_kl && kl > 1.5f * a.target_kl;   // ppo.jl:235-238: skip this apply, stop
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float* bt_out = a.bt + 2 * ((a.step_parity + 1) & 1);
    const float bt1 = bt_in[0], bt2 = bt_in[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.step_stats) {
            float* o = a.step_stats;
            const float pl = stf[0] / n, ent = stf[1] / n, vl = stf[5] / n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = stf[2] / n; o[4] = kl; o[5] = ent; o[6] = stf[4] / n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;                       // loss, ppo.jl:386
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
        }
        if (a.norm_out) *a.norm_out = norm;
        if (bad) { *a.nan_flag = 1; *a.stop_flag_w = 1; }
        if (kl_stop) *a.stop_flag_w = 1;
        if (bad || kl_stop) { bt_out[0] = bt1; bt_out[1] = bt2; } else { bt_out[0] = bt1 * a.beta1; bt_out[1] = bt2 * a.beta2; }
    }
    if (bad || kl_stop) return;
    const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm / norm : 1.0f;   // optimization_utils.jl:98-107
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < a.P) {
        float g = a.flat[p];
        if (scale != 1.0f) g = g * scale;
        const float m = a.beta1 * a.m[p] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[p] + (1.0f - a.beta2) * g * g;
        a.m[p] = m; a.v[p] = v;
        a.params[p] -= m / (1.0f - bt1) / (sqrtf(v / (1.0f - bt2)) + a.eps) * a.lr;  // Optimisers.Adam, eps=1e-5 (ppo.jl:64-66)
    }
}

// Small minibatches (few slabs): grad_reduce_kernel + norm + adam_kernel in ONE workgroup — at B = 64 the optimiser step is bound by its
// dependent launches (~6 x 5-8 us), not by work.  Same arithmetic as the two kernels; the squared norm is summed in a different (fixed) order.
__global__ __launch_bounds__(1024) void ppo_finish_small_kernel(ReduceArgs r, AdamArgs a) {
    __shared__ double sh[16];
    __shared__ float stf[8];
    if (*a.stop_flag) return;
    constexpr int KMAX = 16;                                                            // P <= 16 384 (checked by the host)
    const int tid = threadIdx.x;
    float gacc[KMAX], mo[KMAX], vo[KMAX], po[KMAX];
    double ss = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + 1024 * k;
        gacc[k] = 0.f; mo[k] = 0.f; vo[k] = 0.f; po[k] = 0.f;
        if (p < r.P) {
            const float* base; int stride, offp;
            int Gn = r.G;
            if (p < r.Pa) { base = r.slabs_actor; stride = r.slab_a; offp = p; }
            else if (p < r.Pa + r.Pc) { base = r.slabs_critic; stride = r.slab_c; offp = p - r.Pa; Gn = r.Gc; }
            else { base = r.slabs_actor; stride = r.slab_a; offp = r.Pa + (p - r.Pa - r.Pc); }
            float acc = 0.f;
            for (int g = 0; g < Gn; ++g) acc += base[(size_t)g * stride + offp];      // fixed order
            gacc[k] = acc; r.flat[p] = acc; ss += (double)acc * (double)acc;
            mo[k] = a.m[p]; vo[k] = a.v[p]; po[k] = a.params[p];                        // optimiser state in flight under the norm reduction
        }
    }
    if (tid < 8) {
        double t = 0;
        if (tid < 5) { for (int g = 0; g < r.G; ++g) t += (double)r.slabs_actor[(size_t)g * r.slab_a + r.slab_a - 8 + tid]; }
        else if (tid == 5) { for (int g = 0; g < r.Gc; ++g) t += (double)r.slabs_critic[(size_t)g * r.slab_c + r.slab_c - 8]; }
        else if (tid == 6) t = r.n_samples_local;
        stf[tid] = (float)t; r.flat[r.P + tid] = (float)t;
    }
    const float norm = sqrtf((float)block_sum_f64(ss, sh));
    __syncthreads();
    const float n = a.use_stats ? stf[6] : 1.f;
    const float kl = a.use_stats ? stf[3] / n : 0.f;
    const bool bad = !(norm == norm) || isinf(norm);
    const bool kl_stop = a.use_stats && a.has_target_kl && kl > 1.5f * a.target_kl;
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float* bt_out = a.bt + 2 * ((a.step_parity + 1) & 1);
    const float bt1 = bt_in[0], bt2 = bt_in[1];
    if (tid == 0) {
        if (a.step_stats) {
            float* o = a.step_stats;
            const float pl = stf[0] / n, ent = stf[1] / n, vl = stf[5] / n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = stf[2] / n; o[4] = kl; o[5] = ent; o[6] = stf[4] / n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
        }
        if (a.norm_out) *a.norm_out = norm;
        if (bad) { *a.nan_flag = 1; *a.stop_flag_w = 1; }
        if (kl_stop) *a.stop_flag_w = 1;
        if (bad || kl_stop) { bt_out[0] = bt1; bt_out[1] = bt2; } else { bt_out[0] = bt1 * a.beta1; bt_out[1] = bt2 * a.beta2; }
    }
    if (bad || kl_stop) return;
    const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm / norm : 1.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + 1024 * k;
        if (p < r.P) {
            float g = gacc[k];
            if (scale != 1.0f) g = g * scale;
            const float m = a.beta1 * mo[k] + (1.0f - a.beta1) * g;
            const float v = a.beta2 * vo[k] + (1.0f - a.beta2) * g * g;
            a.m[p] = m; a.v[p] = v;
            a.params[p] = po[k] - m / (1.0f - bt1) / (sqrtf(v / (1.0f - bt2)) + a.eps) * a.lr;
        }
    }
}

// explained_variance sums over the whole buffer (ppo.jl:256): partials of (v-r), (v-r)^2, r, r^2.  A pure stream of 8 bytes per env-step: 16-byte loads, four of them
// per array in flight per thread before the first is used (4-byte loads in a plain grid-stride loop reached 3.6 TB/s, 0.45 of HBM: round 4)
__global__ __launch_bounds__(256) void explained_var_kernel(const float* __restrict__ val, const float* __restrict__ ret, int64_t N, double* partials) {
    __shared__ double sh[16];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const int64_t nq = N >> 2;                                        // float4 quads (both arrays come from hipMalloc: 256-byte aligned)
    const f32x4* __restrict__ v4 = reinterpret_cast<const f32x4*>(val); const f32x4* __restrict__ r4 = reinterpret_cast<const f32x4*>(ret);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    constexpr int U = 4;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < nq; i += U * stride) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = __builtin_nontemporal_load(v4 + i + u * stride); b[u] = __builtin_nontemporal_load(r4 + i + u * stride); }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double d = (double)a[u][k] - (double)b[u][k], r = b[u][k]; s0 += d; s1 += d * d; s2 += r; s3 += r * r; }
    }
    for (; i < nq; i += stride) {
        const f32x4 a = v4[i], b = r4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = (double)a[k] - (double)b[k], r = b[k]; s0 += d; s1 += d * d; s2 += r; s3 += r * r; }
    }
    for (int64_t j = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += stride) {      // N % 4 tail
        const double d = (double)val[j] - (double)ret[j], r = ret[j];
        s0 += d; s1 += d * d; s2 += r; s3 += r * r;
    }
    s0 = block_sum_f64(s0, sh); s1 = block_sum_f64(s1, sh); s2 = block_sum_f64(s2, sh); s3 = block_sum_f64(s3, sh);
    if (threadIdx.x == 0) { double* o = partials + 4 * blockIdx.x; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; }
}

// =============================================================================================
// launchers
// =============================================================================================
template <int KIND, int H, bool WIDE, bool SPLIT> static size_t fwd_lds_bytes() {
    return sizeof(float) * (FwdLds<EnvSpec<KIND>::D, H, EnvSpec<KIND>::A, WIDE, SPLIT>::SIZE + FwdLds<EnvSpec<KIND>::D, H, 1, WIDE, SPLIT>::SIZE);
}
// kind 2 (ScalingWrapperEnv(Pendulum)) shares every kernel that never touches the simulator with kind 1
#define DRIL_DISPATCH(kind, hidden, CALL)                                            \
    do {                                                                             \
        if ((kind) == 0 && (hidden) == 64) { CALL(0, 64); }                          \
        else if (((kind) == 1 || (kind) == 2) && (hidden) == 64) { CALL(1, 64); }    \
        else if ((kind) == 3 && (hidden) == 64) { CALL(3, 64); }                     \
        else if (((kind) == 4 || (kind) == 7) && (hidden) == 64) { CALL(4, 64); }    \
        else return hipErrorInvalidValue;                                            \
    } while (0)

// every kernel below is built for hidden widths 64 (one wave per net), 128 and 256 (wide path)
#define DRIL_DISPATCH_H(K, hidden, CALL)                                             \
    { if ((hidden) == 64) { CALL(K, 64); } else if ((hidden) == 128) { CALL(K, 128); } else if ((hidden) == 256) { CALL(K, 256); } else if ((hidden) == 32) { CALL(K, 32); } else return hipErrorInvalidValue; }
#define DRIL_DISPATCH_FWD(kind, hidden, CALL)                                        \
    do {                                                                             \
        if ((kind) == 0) DRIL_DISPATCH_H(0, hidden, CALL)                            \
        else if ((kind) == 1 || (kind) == 2) DRIL_DISPATCH_H(1, hidden, CALL)        \
        else if ((kind) == 3) DRIL_DISPATCH_H(3, hidden, CALL)                       \
        else if ((kind) == 4 || (kind) == 7) DRIL_DISPATCH_H(4, hidden, CALL)        \
        else if ((kind) == 6) DRIL_DISPATCH_H(6, hidden, CALL)                       \
        else return hipErrorInvalidValue;                                            \
    } while (0)

// kernels that step / observe the simulator: one instantiation per env kind
#define DRIL_DISPATCH_ENV(kind, hidden, CALL)                                        \
    do {                                                                             \
        if ((kind) == 0) DRIL_DISPATCH_H(0, hidden, CALL)                            \
        else if ((kind) == 1) DRIL_DISPATCH_H(1, hidden, CALL)                       \
        else if ((kind) == 2) DRIL_DISPATCH_H(2, hidden, CALL)                       \
        else if ((kind) == 3) DRIL_DISPATCH_H(3, hidden, CALL)                       \
        else if ((kind) == 4) DRIL_DISPATCH_H(4, hidden, CALL)                       \
        else if ((kind) == 7) DRIL_DISPATCH_H(7, hidden, CALL)                       \
        else if ((kind) == 6) DRIL_DISPATCH_H(6, hidden, CALL)                       \
        else return hipErrorInvalidValue;                                            \
    } while (0)

hipError_t launch_fold_partials(const double* partials, int nblocks, double* out16, hipStream_t s) {
    fold_partials_kernel<<<1, 256, 0, s>>>(partials, nblocks, out16);
    return hipGetLastError();
}
hipError_t set_max_dynamic_lds(const void* fn, size_t bytes) {
    static std::mutex mu; static std::vector<std::pair<const void*, int>> done;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), std::make_pair(fn, dev)) != done.end()) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.push_back(std::make_pair(fn, dev));
    return e;
}
hipError_t launch_build_wimg_split(const float* params, NetOff off, int H, void* w2p, void* w2tp, void* w2pf, hipStream_t s) {
    const int total = (H / 32) * (H / 32) * 2 * 64;
    build_wimg_split_kernel<<<(total + 255) / 256, 256, 0, s>>>(params, off, H, (u32x4*)w2p, (u32x4*)w2tp, (u32x4*)w2pf);
    return hipGetLastError();
}
hipError_t launch_w2_absmax(const float* params, NetOff actor, NetOff critic, int H, unsigned* out_bits, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out_bits, 0, sizeof(unsigned), s); if (e != hipSuccess) return e;
    const int total = 2 * H * H;
    w2_absmax_kernel<<<(total + 4095) / 4096, 256, 0, s>>>(params, actor.w2, critic.w2, H * H, out_bits);
    return hipGetLastError();
}
hipError_t launch_build_wimg(const float* params, NetOff off, int H, float* w2a, float* w2ta, hipStream_t s) {
    build_wimg_kernel<<<(H * H + 255) / 256, 256, 0, s>>>(params, off, H, w2a, w2ta);
    return hipGetLastError();
}

hipError_t launch_env_reset(int kind, int E, uint64_t seed0, float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* dr, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_reset_kernel<0><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else if (kind == 3 || kind == 4 || kind == 7) env_reset_kernel<3><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else if (kind == 6) env_reset_kernel<6><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else env_reset_kernel<1><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    return hipGetLastError();
}
hipError_t launch_env_observe(int kind, int E, const float* state, float* obs, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_observe_kernel<0><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 1) env_observe_kernel<1><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 2) env_observe_kernel<2><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 6) env_observe_kernel<6><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 7) env_observe_kernel<7><<<blocks, 256, 0, s>>>(E, state, obs);
    else env_observe_kernel<3><<<blocks, 256, 0, s>>>(E, state, obs);
    return hipGetLastError();
}
hipError_t launch_env_step(int kind, int E, uint64_t seed0, int episode_len, int fixed_len, int action_start, const void* actions,
                           float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* rew, uint8_t* term, uint8_t* trunc,
                           float* tobs, MonitorArgs mon, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_step_kernel<0><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 1) env_step_kernel<1><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 2) env_step_kernel<2><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 3) env_step_kernel<3><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 6) env_step_kernel<6><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 7) env_step_kernel<7><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else env_step_kernel<4><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    return hipGetLastError();
}
hipError_t launch_monitor_collect(const uint8_t* flags, const float* ep_ret, const int32_t* ep_len, int E, int T, int W, int* cnt,
                                  float* ring_ret, int32_t* ring_len, int* meta, hipStream_t s) {
    monitor_count_kernel<<<T < 1024 ? T : 1024, 256, 0, s>>>(flags, E, T, cnt);
    monitor_collect_kernel<<<1, 1024, 0, s>>>(flags, ep_ret, ep_len, E, T, W, cnt, ring_ret, ring_len, meta);
    return hipGetLastError();
}

hipError_t launch_norm_step(int kind, const NormStepArgs& a, int nblocks, hipStream_t s) {
    if (kind == 0) norm_step_kernel<0><<<nblocks, 256, 0, s>>>(a); else if (kind == 1) norm_step_kernel<1><<<nblocks, 256, 0, s>>>(a); else if (kind == 2) norm_step_kernel<2><<<nblocks, 256, 0, s>>>(a);
    else if (kind == 3) norm_step_kernel<3><<<nblocks, 256, 0, s>>>(a); else if (kind == 6) norm_step_kernel<6><<<nblocks, 256, 0, s>>>(a); else if (kind == 7) norm_step_kernel<7><<<nblocks, 256, 0, s>>>(a); else norm_step_kernel<4><<<nblocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_norm_apply(const NormApplyArgs& a, hipStream_t s) {
    int blocks = (a.E + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_obs_partials(int kind, int E, const float* state, float* raw, double* partials, int nblocks, hipStream_t s) {
    if (kind == 0) obs_partials_kernel<0><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 1) obs_partials_kernel<1><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 2) obs_partials_kernel<2><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 6) obs_partials_kernel<6><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 7) obs_partials_kernel<7><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else obs_partials_kernel<3><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    return hipGetLastError();
}
hipError_t launch_norm_obs_apply(const NormObsArgs& a, hipStream_t s) {
    int blocks = (a.E * a.D + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_obs_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_rew_partials(int E, const float* rew_raw, float* disc_returns, float gamma, int update, double* partials, int nblocks, hipStream_t s) {
    rew_partials_kernel<<<nblocks, 256, 0, s>>>(E, rew_raw, disc_returns, gamma, update, partials);
    return hipGetLastError();
}
hipError_t launch_norm_rew_apply(const NormRewArgs& a, hipStream_t s) {
    int blocks = (a.E + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_rew_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_policy(int kind, int hidden, const PolicyArgs& a, int max_blocks, hipStream_t s) {
    const int64_t ntiles = (a.B + kTile - 1) / kTile;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
#define CALLS(K, HH, SP)                                                                                      \
    {                                                                                                         \
        const size_t lds = fwd_lds_bytes<K, HH, (HH > 64), SP>();                                             \
        { hipError_t e = set_max_dynamic_lds((const void*)policy_kernel<K, HH, (HH > 64), SP>, lds); if (e != hipSuccess) return e; } \
        policy_kernel<K, HH, (HH > 64), SP><<<blocks, 256, lds, s>>>(a);                                      \
    }
// (hidden 32 — the reference's benchmark-suite shape — exists on the f32-MFMA forward only: the f16-piece images are laid out for 64-wide layers)
#define CALL(K, HH) { if constexpr (HH == 32) { CALLS(K, HH, false) } else { if (a.exact_f32) CALLS(K, HH, false) else CALLS(K, HH, true) } }
    DRIL_DISPATCH_FWD(kind, hidden, CALL);
#undef CALL
#undef CALLS
    return hipGetLastError();
}

template <int KIND, int H, bool SPLIT> static size_t duo_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    return sizeof(float) * (FwdLds<D, H, A, false, SPLIT>::SIZE + FwdLds<D, H, 1, false, SPLIT>::SIZE + 2 * (2 * D + 1) * 32);
}
hipError_t launch_rollout(int kind, int hidden, const RolloutArgs& a, hipStream_t s) {
    static const bool no_duo = debug_env("DRIL_NO_ROLLOUT_DUO") != nullptr;             // A/B (DRIL_DEBUG=1)
    if ((hidden == 64 || hidden == 32) && a.E <= 16384 && !no_duo && ((kind >= 0 && kind <= 4) || kind == 6 || kind == 7)) {                                      // env counts that leave SIMDs idle: two waves per tile of 32 envs
        const int blocks = (a.E + kTile - 1) / kTile;
#define CALLDS(K, HH, SP) { const size_t lds = duo_lds_bytes<K, HH, SP>(); \
            { hipError_t e = set_max_dynamic_lds((const void*)rollout_duo_kernel<K, HH, SP>, lds); if (e != hipSuccess) return e; } \
            rollout_duo_kernel<K, HH, SP><<<blocks, 128, lds, s>>>(a); }
#define CALLD(K) { if (hidden == 32) CALLDS(K, 32, false) else if (a.exact_f32) CALLDS(K, 64, false) else CALLDS(K, 64, true) }
        if (kind == 0) CALLD(0) else if (kind == 1) CALLD(1) else if (kind == 2) CALLD(2) else if (kind == 3) CALLD(3) else if (kind == 6) CALLD(6) else if (kind == 7) CALLD(7) else CALLD(4)
#undef CALLD
#undef CALLDS
        return hipGetLastError();
    }
    const int blocks = (a.E + 4 * kTile - 1) / (4 * kTile);
#define CALLS(K, HH, SP)                                                                                      \
    {                                                                                                         \
        const size_t lds = fwd_lds_bytes<K, HH, (HH > 64), SP>();                                             \
        { hipError_t e = set_max_dynamic_lds((const void*)rollout_kernel<K, HH, (HH > 64), SP>, lds); if (e != hipSuccess) return e; } \
        rollout_kernel<K, HH, (HH > 64), SP><<<blocks, 256, lds, s>>>(a);                                     \
    }
#define CALL(K, HH) { if constexpr (HH == 32) { CALLS(K, HH, false) } else { if (a.exact_f32) CALLS(K, HH, false) else CALLS(K, HH, true) } }
    DRIL_DISPATCH_ENV(kind, hidden, CALL);
#undef CALL
#undef CALLS
    return hipGetLastError();
}

int gae_chunks(int T) { return (T + kGaeRows - 1) / kGaeRows; }
hipError_t launch_gae(int E, int T, float gamma, float lam, const float* rew, const float* val, const uint8_t* flags,
                      const float* boot, const float* last_values, float* adv, float* ret, unsigned long long* carry, unsigned tag, int* err, hipStream_t s) {
    const int C = gae_chunks(T), GB = (E + 255) / 256;
    gae_scan_kernel<<<(unsigned)((size_t)C * GB), 256, 0, s>>>(E, T, C, GB, gamma, lam, rew, val, flags, boot, last_values, adv, ret, carry, tag, err);
    return hipGetLastError();
}

hipError_t launch_adv_moments(const MomentsArgs& a, int nblocks, hipStream_t s) {
    adv_moments_kernel<<<nblocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_epoch_moments(const float* adv, int64_t N, int64_t B, int nb, uint64_t key, int bits, double* block_tables, int nblocks,
                                double* table3, const int* stop_flag, hipStream_t s) {
    int R = 16;
    while (R > 1 && (size_t)R * 2 * (nb + 1) * sizeof(double) > 32768) R >>= 1;          // nb <= 2048 (dril_api.hip): one copy is at most 32 KB
    const size_t lds = (size_t)R * 2 * (nb + 1) * sizeof(double);
    if (bits <= 31) epoch_moments_kernel<true><<<nblocks, 256, lds, s>>>(adv, N, B, 1.0 / (double)B, nb, R, key, bits, block_tables, stop_flag);
    else epoch_moments_kernel<false><<<nblocks, 256, lds, s>>>(adv, N, B, 0.0, nb, R, key, bits, block_tables, stop_flag);
    epoch_moments_finalize_kernel<<<nb, 64, 0, s>>>(block_tables, nblocks, nb, N, B, table3, stop_flag);
    return hipGetLastError();
}
hipError_t launch_moments_finalize(const double* partials, int nblocks, double* out3, double n_local, const int* stop_flag, hipStream_t s) {
    moments_finalize_kernel<<<1, 256, 0, s>>>(partials, nblocks, out3, n_local, stop_flag);
    return hipGetLastError();
}

// the update kernels live in their own translation units (dril_grad_f32.hip, dril_grad_pair.hip, dril_grad_wide.hip)
hipError_t launch_ppo_grad(int kind, int hidden, const GradArgs& a, hipStream_t s) {
    if (hidden > 64) return launch_ppo_grad_wide(kind, hidden, a, s);
    if (hidden == 64 && a.variant == 2 && a.rec) return launch_ppo_grad_pair(kind, a, s);   // two waves per tile, two waves per SIMD
    return launch_ppo_grad_f32(kind, hidden, a, s);
}

// the DataLoader order of one epoch written out once: position -> buffer index (perm_index evaluated N times here instead of once per lane, wave, net and tile inside the
// update kernels, where it costs ~35 VALU of an issue-bound loop).  The pass is bound by its WRITES (134 M x 8 bytes in 0.195 ms = 5.5 TB/s at configs[1]), so the
// indices are 32-bit (N < 2^31, host-checked): half the bytes here and in every read-back
__global__ void epoch_index_kernel(int64_t N, uint64_t key, int bits, int32_t* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < N; p += (int64_t)gridDim.x * blockDim.x) out[p] = (int32_t)perm_index(p, N, key, bits);
}
hipError_t launch_epoch_index(int64_t N, uint64_t key, int bits, int32_t* out, hipStream_t s) {
    int blocks = (int)((N + 255) / 256); if (blocks > 8192) blocks = 8192;
    epoch_index_kernel<<<blocks, 256, 0, s>>>(N, key, bits, out);
    return hipGetLastError();
}
hipError_t launch_pack_records(int kind, int64_t N, const float* obs, const void* act, const float* adv, const float* logp, const float* ret, float4* rec, hipStream_t s) {
    int blocks = (int)((N + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (kind == 0) pack_records_kernel<0><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else if (kind == 3) pack_records_kernel<3><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else if (kind == 4 || kind == 7) pack_records_kernel<4><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else if (kind == 6) pack_records_kernel<6><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else pack_records_kernel<1><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    return hipGetLastError();
}
hipError_t launch_grad_reduce(const ReduceArgs& a, hipStream_t s) {
    grad_reduce_kernel<<<(a.P + 31) / 32, 1024, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_finish_small(const ReduceArgs& r, const AdamArgs& a, hipStream_t s) {
    ppo_finish_small_kernel<<<1, 1024, 0, s>>>(r, a);
    return hipGetLastError();
}
hipError_t launch_grad_norm(const float* flat, int P, double* norm_partials, const int* stop_flag, hipStream_t s) {
    grad_norm_kernel<<<(P + 31) / 32, 256, 0, s>>>(flat, P, norm_partials, stop_flag);
    return hipGetLastError();
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t s) {
    adam_kernel<<<(a.P + 255) / 256, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_explained_var(const float* val, const float* ret, int64_t N, double* partials, int nblocks, hipStream_t s) {
    explained_var_kernel<<<nblocks, 256, 0, s>>>(val, ret, N, partials);
    return hipGetLastError();
}

static void kind_dims(int kind, int& D, int& A, bool& disc) {
    switch (kind) {
        case 0: D = 4; A = 2; disc = true; break;
        case 3: D = 2; A = 3; disc = true; break;
        case 4: case 7: D = 2; A = 1; disc = false; break;
        case 6: D = 6; A = 3; disc = true; break;
        default: D = 3; A = 1; disc = false; break;
    }
}
int slab_size_actor(int kind, int hidden) {
    int D, A; bool disc; kind_dims(kind, D, A, disc);
    const NetOff n = net_off(0, D, hidden, hidden, A);
    return (n.end + (disc ? 0 : A) + 8 + 3) / 4 * 4 + 0;
}
int slab_size_critic(int kind, int hidden) {
    int D, A_; bool disc_; kind_dims(kind, D, A_, disc_);
    const NetOff n = net_off(0, D, hidden, hidden, 1);
    return (n.end + 8 + 3) / 4 * 4;
}

}  // namespace dril

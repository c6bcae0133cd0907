// dril_api.hip — the C ABI of libdril_hip.so (include/dril_hip.h): handle, device memory, launch sequencing.
// One HIP stream per handle; kernels of one PPO iteration are enqueued back-to-back with no host sync
// (KL early-stop and NaN detection are device flags the later kernels test), RCCL all-reduces ride the
// same stream.  RCCL is dlopen'ed lazily so single-GPU users never load it.
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dril_hip.h"
#include "dril_internal.h"

using namespace dril;

#define DRIL_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_create_error;

// ---- minimal RCCL surface, resolved at run time ------------------------------------------------
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, const void*, int) = nullptr;   // ncclUniqueId passed by value = 128-byte struct
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* (*GetLastError)(void*) = nullptr;       // ncclGetLastError(comm): the human-readable cause behind a status code (which rank / device / transport)
};
struct NcclId { char bytes[128]; };
typedef int (*nccl_init_rank_fn)(void**, int, NcclId, int);
RcclApi g_rccl;
bool load_rccl(std::string& err) {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, const void*, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommCount");
    g_rccl.GetLastError = (const char* (*)(void*))dlsym(g_rccl.lib, "ncclGetLastError");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) { err = "librccl lacks ncclGetUniqueId/ncclCommInitRank/ncclAllReduce"; return false; }
    return true;
}
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;
// "<call>: <ncclGetErrorString(rc)> [<ncclGetLastError(comm)>]": a failing first contact with a multi-GPU node must name its cause in dril_last_error
std::string rccl_error(const char* call, int rc, void* comm) {
    std::string m = std::string(call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error") + " (status " + std::to_string(rc) + ")";
    if (g_rccl.GetLastError) { const char* le = g_rccl.GetLastError(comm); if (le && *le) m += std::string(" [ncclGetLastError: ") + le + "]"; }
    return m;
}

// ---- debug loopback communicator -----------------------------------------------------------------
// RCCL refuses two ranks on one device, so the data-parallel code of this file (every `reduce` branch below) could only be
// executed on a multi-GPU node.  A loopback group joins n handles of ONE process on ONE device as ranks 0..n-1: the all-reduce
// call sites, counts and dtypes are unchanged, only the transport differs — the ranks' host threads rendezvous, the last arriver
// makes its stream wait for every rank's buffer, one kernel sums the n device buffers element-wise in rank order and writes the
// sum back to all of them (every rank ends with bit-identical values, as after a ring all-reduce), and the other ranks' streams
// wait for that kernel.  Each handle must be driven from its own host thread (dril_debug_comm_loopback, include/dril_hip.h).
constexpr int kLoopMax = 8;
struct LoopPtrs { void* p[kLoopMax]; };
struct LoopGroup {
    std::mutex mu; std::condition_variable cv;
    int n = 0, arrived = 0, refs = 0; uint64_t gen = 0; int64_t calls = 0;
    LoopPtrs bufs{}; hipEvent_t ready[kLoopMax] = {nullptr}; hipEvent_t done = nullptr;
    size_t count = 0; int dtype = 0; bool mismatch = false, failed = false;
};
template <typename T> __global__ void loop_allreduce_kernel(LoopPtrs bufs, int n, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = ((const T*)bufs.p[0])[i];
    for (int r = 1; r < n; ++r) s += ((const T*)bufs.p[r])[i];      // fixed rank order: deterministic
    for (int r = 0; r < n; ++r) ((T*)bufs.p[r])[i] = s;             // this thread has read element i of every rank before it writes any
}

struct ProfEvent { int kid; hipEvent_t a, b; };

}  // namespace

struct dril_handle {
    dril_config cfg;
    int D = 0, A = 0, S = 0, P = 0, Pa = 0, Pc = 0, log_std_off = 0;
    bool discrete = true;
    NetOff actor{}, critic{};
    int64_t N = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    std::string device_info;   // "device <ordinal> of <visible>: <name> <arch>, PCI <bus id>, HIP_VISIBLE_DEVICES=.. ROCR_VISIBLE_DEVICES=.." (dril_device_info; part of every RCCL failure message)
    float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr, *bt = nullptr, *flat = nullptr, *norm_out = nullptr;
    double* norm_partials = nullptr; int n_norm_partials = 0;
    float *slabs_a = nullptr, *slabs_c = nullptr; int slab_a = 0, slab_c = 0, Gmax = 0;
    float* state = nullptr; int32_t* step_count = nullptr; uint32_t *episode = nullptr, *gstep = nullptr; float* disc_returns = nullptr;
    float *obs = nullptr, *rew = nullptr, *adv = nullptr, *ret = nullptr, *logp = nullptr, *val = nullptr, *boot = nullptr, *last_values = nullptr;
    void* act = nullptr; uint8_t* flags = nullptr;
    void* noise_dev = nullptr; bool noise_set = false;
    int64_t* perm_dev = nullptr; size_t perm_count = 0;
    int32_t* epoch_index = nullptr; bool no_epoch_index = false;   // the device DataLoader order of the current epoch, written out (large minibatches; DRIL_NO_EPOCH_INDEX)
    double *adv_partials = nullptr, *adv_stats = nullptr, *ev_partials = nullptr; int adv_blocks = 0, ev_blocks = 0;
    float* step_stats = nullptr; int step_stats_cap = 0;
    int *stop_flag = nullptr, *nan_flag = nullptr;
    float *e_obs = nullptr, *e_rew = nullptr, *e_tobs = nullptr; uint8_t *e_term = nullptr, *e_trunc = nullptr; void* e_act = nullptr;
    uint64_t env_seed0 = 0, adam_steps = 0, update_counter = 0; uint32_t policy_calls = 0;
    float lr = 0;
    bool env_ready = false;
    unsigned long long* dbg = nullptr;
    bool force_allreduce = false, force_stepwise = false;
    double *epoch_tables = nullptr, *epoch_stats = nullptr; int epoch_blocks = 2048, epoch_nb_cap = 0;   // per-epoch advantage moments
    float *w2a_actor = nullptr, *w2ta_actor = nullptr, *w2a_critic = nullptr, *w2ta_critic = nullptr; bool wide = false, wimg_dirty = true;   // wide nets (H > 64)
    void *w2pf_actor = nullptr, *w2pf_critic = nullptr;   // wide nets: the forward stream in the k-order of a register B operand (net_forward_wide_split)
    void *w2p_actor = nullptr, *w2tp_actor = nullptr, *w2p_critic = nullptr, *w2tp_critic = nullptr;   // wide nets: pre-split bf16 fragment streams of W2 / W2' (ppo_grad_wide_split_kernel)
    // MonitorWrapperEnv (cfg.monitor_window > 0)
    float *mon_cur_ret = nullptr, *ep_ret = nullptr, *mon_ring_ret = nullptr, *e_ep_ret = nullptr; int32_t *mon_cur_len = nullptr, *ep_len = nullptr, *mon_ring_len = nullptr, *e_ep_len = nullptr;
    int *mon_cnt = nullptr, *mon_meta = nullptr; uint8_t* e_flags = nullptr;
    float4* rec = nullptr;   // packed minibatch records (see pack_records_kernel)
    RmsState *obs_rms = nullptr, *ret_rms = nullptr; int obs_par = 0, ret_par = 0;   // ping-pong RunningMeanStd pairs
    double* rms_partials = nullptr; int rms_blocks = 256; float* e_obs_raw = nullptr; float* e_rew_n = nullptr;   // e_obs_raw / e_rew: get_original_obs / get_original_rewards; e_rew_n: rewards as the wrapper delivers them
    double* rms_red = nullptr;   // data-parallel: this step's partial sums folded to one row and summed over ranks
    int grad_actor_pct = 0;   // ppo_grad_pair_kernel: share (per mille) of the pairs given to the actor; 0 = by head (env DRIL_GRAD_ACTOR_PERMILLE)
    int last_variant = -1;  // which gradient kernel the last optimiser step ran: 0 ppo_grad_kernel, 2 ppo_grad_wide_kernel, 3 generic path, 4 ppo_grad_wide_split_kernel, 5 ppo_grad_pair_kernel (dril_grad_kernel_info)
    int grad_variant = -1;  // env DRIL_GRAD_VARIANT: 0 = the exact-f32 kernels (ppo_grad_kernel / ppo_grad_wide_kernel), 1 = the bf16-split kernels (ppo_grad_pair_kernel at hidden 64, ppo_grad_wide_split_kernel at 128 / 256) for every minibatch size, -1 = default: wide nets split, hidden 64 by minibatch size
    bool external = false; bool generic = false; float* gen_tmp = nullptr;   // generic: layer-by-layer kernels (host envs, or a device env whose hidden_dims the fused kernels are not built for)
    GenericDims gd{}; GenericWs gws; int ext_t = 0; bool ext_acted = false;   // DRIL_ENV_EXTERNAL: host envs, generic kernels
    float* ext_stage_rew = nullptr; uint8_t* ext_stage_flags = nullptr;   // pinned [T][E] staging: dril_ext_record returns without draining the stream
    void* comm = nullptr;
    LoopGroup* loop = nullptr;   // debug loopback communicator (dril_debug_comm_loopback)
    int64_t allreduce_calls = 0;
    float* retry_snap = nullptr; bool no_f32_retry = false; int64_t f32_retries = 0;   // ppo_update: state before the update (for the exact-f32 redo of an update that left f16's range); DRIL_NO_F32_RETRY; how often it happened
    // the redo's bookkeeping (dril_f32_fallback_info): used_f16 = an f16-piece kernel ran in the current update (sticky over its optimiser steps — the LAST step's kernel says nothing
    // about a ragged tail); streak = consecutive redone updates; after kRetryLatchAfter of them the next kRetryLatchUpdates updates run the exact-f32 kernels directly (no wasted f16
    // pass, no snapshot), then ONE update probes f16 again; direct = updates run that way (latched, or max |W2| out of f16's range)
    bool used_f16 = false, spin_timeout = false; int f32_streak = 0, f32_latch_left = 0; int64_t f32_direct_updates = 0, persistent_fallbacks = 0;
    unsigned long long* gae_carry = nullptr; unsigned gae_tag = 0; int* gae_err = nullptr;   // gae_scan_kernel: the chunk-to-chunk carry words, the launch tag that validates them, its give-up flag
    unsigned* w2max_dev = nullptr; float w2max = 0.f;   // max |W2| over both nets (fused kernels): host copy refreshed by dril_set_params and with every optimiser run's statistics
    bool no_small_path = false;   // DRIL_NO_SMALL_PATH (with DRIL_DEBUG=1), latched in dril_create
    bool no_persistent = false; unsigned long long* small_xchg = nullptr; uint64_t* epoch_keys = nullptr; int epoch_keys_cap = 0; int64_t small_chunk = 16384;   // ppo_update_small_kernel (batch_size <= 64): DRIL_NO_PERSISTENT_UPDATE; per-epoch DataLoader keys on the device
    std::vector<ProfEvent> prof_pending; std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
    double prof_ms[DRIL_K_COUNT] = {0}; int64_t prof_n[DRIL_K_COUNT] = {0}, prof_all[DRIL_K_COUNT] = {0}; bool prof_open = false;
    std::string err;
};

namespace {

int fail(dril_handle* h, int code, const std::string& msg) { if (h) h->err = msg; else g_create_error = msg; return code; }
#define HIPCHK(h, expr)                                                                                      \
    do { hipError_t _e = (expr); if (_e != hipSuccess)                                                       \
        return fail(h, DRIL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
// every entry point makes the handle's device current first: a process may hold handles on several devices
#define NEED(h) do { if (!(h)) return fail(nullptr, DRIL_ERR_NOT_INITIALISED, "null handle"); (void)hipSetDevice((h)->cfg.device); } while (0)

template <typename T> hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, (n ? n : 1) * sizeof(T)); }

// cfg.profile_events = k >= 1: the per-iteration classes (rollout, GAE, pack, moments, explained variance) are bracketed at every launch; the classes launched once per
// OPTIMISER STEP (gradient, reduce, Adam, all-reduce) at every k-th launch — a hipEventRecord costs the stream ~3.5 us and an iteration of configs[1] holds 960 such launches
// (measured round 4, same box: 350.4 / 346.9 ms with all of them bracketed, 343.7 / 343.3 with none / every 8th; hipEventDisableSystemFence changes nothing)
bool prof_per_step(int kid) { return kid == DRIL_K_PPO_GRAD || kid == DRIL_K_GRAD_REDUCE || kid == DRIL_K_ADAM || kid == DRIL_K_ALLREDUCE; }
void prof_begin(dril_handle* h, int kid) {
    h->prof_open = false;
    if (!h->cfg.profile_events) return;
    const int64_t i = h->prof_all[kid]++;
    if (prof_per_step(kid) && h->cfg.profile_events > 1 && i % h->cfg.profile_events) return;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!h->prof_pool.empty()) { ev = h->prof_pool.back(); h->prof_pool.pop_back(); }
    else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
    hipEventRecord(ev.first, h->stream);
    h->prof_pending.push_back({kid, ev.first, ev.second});
    h->prof_open = true;
}
void prof_end(dril_handle* h) {
    if (!h->prof_open) return;
    hipEventRecord(h->prof_pending.back().b, h->stream);
    h->prof_open = false;
}
void prof_resolve(dril_handle* h) {   // stream must be drained
    for (auto& p : h->prof_pending) {
        float ms = 0; if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { h->prof_ms[p.kid] += ms; h->prof_n[p.kid] += 1; }
        h->prof_pool.push_back({p.a, p.b});
    }
    h->prof_pending.clear();
}
int sync(dril_handle* h) { HIPCHK(h, hipStreamSynchronize(h->stream)); prof_resolve(h); return DRIL_OK; }

bool comm_ready(const dril_handle* h) { return h->comm != nullptr || h->loop != nullptr; }

// the loopback transport of one all-reduce (see LoopGroup): called by every rank's host thread with the same count / dtype
int loop_allreduce(dril_handle* h, void* buf, size_t count, int dtype) {
    LoopGroup* g = h->loop; const int r = h->cfg.rank;
    HIPCHK(h, hipEventRecord(g->ready[r], h->stream));                 // this rank's contribution is complete behind this event
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->failed) return fail(h, DRIL_ERR_RCCL, "loopback communicator: an earlier all-reduce failed");
    const uint64_t my_gen = g->gen;
    if (g->arrived == 0) { g->count = count; g->dtype = dtype; g->mismatch = false; }
    else if (g->count != count || g->dtype != dtype) g->mismatch = true;
    g->bufs.p[r] = buf;
    if (++g->arrived == g->n) {
        hipError_t e = hipSuccess;
        for (int q = 0; q < g->n && e == hipSuccess; ++q) e = hipStreamWaitEvent(h->stream, g->ready[q], 0);
        if (e == hipSuccess && !g->mismatch) {
            const unsigned nb = (unsigned)((count + 255) / 256);
            if (dtype == kNcclFloat64) loop_allreduce_kernel<double><<<nb, 256, 0, h->stream>>>(g->bufs, g->n, count);
            else loop_allreduce_kernel<float><<<nb, 256, 0, h->stream>>>(g->bufs, g->n, count);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(g->done, h->stream);
        if (e != hipSuccess || g->mismatch) g->failed = true;
        g->arrived = 0; g->gen += 1; g->calls += 1;
        g->cv.notify_all();
        if (e != hipSuccess) return fail(h, DRIL_ERR_HIP, std::string("loopback all-reduce: ") + hipGetErrorString(e));
    } else {
        if (!g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->gen != my_gen; })) {
            g->failed = true; g->arrived = 0; g->gen += 1; g->cv.notify_all();
            return fail(h, DRIL_ERR_RCCL, "loopback all-reduce: the other ranks did not arrive within 120 s (every handle of the group needs its own host thread making the same calls)");
        }
        if (!g->failed) HIPCHK(h, hipStreamWaitEvent(h->stream, g->done, 0));
    }
    if (g->failed) return fail(h, DRIL_ERR_RCCL, g->mismatch ? "loopback all-reduce: the ranks disagree on count / dtype (they are not making the same calls)" : "loopback all-reduce failed on another rank");
    return DRIL_OK;
}

int rccl_allreduce(dril_handle* h, void* buf, size_t count, int dtype) {
    if (!comm_ready(h)) return DRIL_OK;
    h->allreduce_calls += 1;
    prof_begin(h, DRIL_K_ALLREDUCE);
    int rc = 0;
    if (h->loop) { const int lrc = loop_allreduce(h, buf, count, dtype); prof_end(h); return lrc; }
    rc = g_rccl.AllReduce(buf, buf, count, dtype, kNcclSum, h->comm, h->stream);
    prof_end(h);
    if (rc != 0) return fail(h, DRIL_ERR_RCCL, rccl_error("ncclAllReduce", rc, h->comm) + "; " + h->device_info);
    return DRIL_OK;
}

bool normalizing(const dril_handle* h) { return h->cfg.norm_obs || h->cfg.norm_reward; }
size_t act_bytes_per(const dril_handle* h) { return h->discrete ? 4 : 4 * (size_t)h->A; }

// wide nets: (re)build the pre-tiled W2 / W2' images after every parameter change (enqueued on the handle's stream)
// wide nets: 1 = ppo_grad_wide_split_kernel (bf16 matrix cores), 0 = the f32-MFMA ppo_grad_wide_kernel (DRIL_GRAD_VARIANT; records are needed by the split form)
// default by measurement (profiles/r02_wide_split.md): the split form for both wide widths (hidden 256: 176-183 vs 118 TFLOP/s; hidden 128: 151 vs 113.5)
int wide_variant(const dril_handle* h) { return (h->wide && h->rec && (h->grad_variant < 0 ? 1 : h->grad_variant)) ? 1 : 0; }
// hidden [64,64]: may ppo_grad_pair_kernel run at all (it reads packed records; DRIL_GRAD_VARIANT=0 pins the exact-f32 kernel)
// minibatches of at least this many 32-sample tiles per CU run ppo_grad_pair_kernel (measured on the f16 arithmetic, 256 CUs: 65 536 samples 50.7 us against 64.8 us on the
// exact-f32 kernel, 16 384 samples 42.2 against 39.0; with the bf16 x 3 arithmetic of rounds 2 - 3 the crossover was at 16 tiles per CU)
constexpr int kPairTilesPerCu = 8;
bool pair_variant(const dril_handle* h) { return !h->wide && !h->generic && h->cfg.hidden1 == 64 && h->grad_variant != 0 && h->rec && h->D <= 8; }   // (D > 4, Acrobot: dW1 / db1 through a third piece image on the matrix cores — 16 (D + 1) per-lane accumulators would not fit)
// the forward of the fused rollout / policy kernels: f16 two-piece W2 while every W2 entry is inside f16's range, else (or with DRIL_GRAD_VARIANT=0: "exact f32 everywhere") the f32-MFMA instantiations
bool fwd_exact(const dril_handle* h) { return h->grad_variant == 0 || h->cfg.hidden1 == 32 || !(h->w2max < kFwdSplitMaxW); }   // (hidden 32 has no f16-piece instantiation)
constexpr int kRetryLatchAfter = 2, kRetryLatchUpdates = 16;
int ensure_wimg(dril_handle* h) {
    if (!h->wide || !h->wimg_dirty) return DRIL_OK;
    HIPCHK(h, launch_build_wimg(h->params, h->actor, h->cfg.hidden1, h->w2a_actor, h->w2ta_actor, h->stream));
    HIPCHK(h, launch_build_wimg(h->params, h->critic, h->cfg.hidden1, h->w2a_critic, h->w2ta_critic, h->stream));
    {                                             // the pre-split f16 fragment streams: ppo_grad_wide_split_kernel, and the forward of rollout / policy kernels
        HIPCHK(h, launch_build_wimg_split(h->params, h->actor, h->cfg.hidden1, h->w2p_actor, h->w2tp_actor, h->w2pf_actor, h->stream));
        HIPCHK(h, launch_build_wimg_split(h->params, h->critic, h->cfg.hidden1, h->w2p_critic, h->w2tp_critic, h->w2pf_critic, h->stream));
    }
    h->wimg_dirty = false;
    return DRIL_OK;
}

// fused per-kind kernels for the device envs, the layer-by-layer generic path for DRIL_ENV_EXTERNAL
hipError_t run_policy(dril_handle* h, const PolicyArgs& a) {
    if (!h->generic) return launch_policy(h->cfg.env_kind, h->cfg.hidden1, a, 8 * h->num_cus, h->stream);
    if (a.boot_where) {                       // V(terminal_observation) of the previous env step (fused into policy_kernel on the fused path): critic over all rows, kept where truncated
        PolicyArgs b = a; b.obs = a.boot_obs; b.noise = nullptr; b.actions = nullptr; b.values = h->gen_tmp; b.logp = nullptr; b.entropy = nullptr; b.mode = 2;
        b.obs_out = nullptr; b.boot_obs = nullptr; b.boot_where = nullptr; b.boot_out = nullptr; b.gstep = nullptr;
        hipError_t e = generic_policy(h->gd, b, h->gws, h->stream); if (e != hipSuccess) return e;
        e = generic_select(a.B, a.boot_where, h->gen_tmp, a.boot_out, h->stream); if (e != hipSuccess) return e;
    }
    PolicyArgs q = a; q.boot_obs = nullptr; q.boot_where = nullptr; q.boot_out = nullptr;
    return generic_policy(h->gd, q, h->gws, h->stream);
}
#define NOT_EXTERNAL(h, what) do { if ((h)->external) return fail(h, DRIL_ERR_UNSUPPORTED, what ": the envs of DRIL_ENV_EXTERNAL live on the host (use dril_ext_act / dril_ext_record / dril_ext_finish)"); } while (0)

PolicyArgs policy_args(dril_handle* h, const float* obs, int64_t B, const void* noise, void* actions, float* values, float* logp,
                       float* entropy, int mode) {
    PolicyArgs a{};
    a.params = h->params; a.obs = obs; a.B = B; a.noise = noise; a.actions = actions; a.values = values; a.logp = logp; a.entropy = entropy;
    a.mode = mode; a.action_start = h->cfg.action_start; a.log_std_off = h->log_std_off; a.seed = h->cfg.seed; a.call_counter = h->policy_calls;
    a.actor = h->actor; a.critic = h->critic;
    a.exact_f32 = fwd_exact(h) ? 1 : 0;
    a.w2a_actor = a.exact_f32 ? h->w2a_actor : (const float*)h->w2pf_actor; a.w2a_critic = a.exact_f32 ? h->w2a_critic : (const float*)h->w2pf_critic;   // wide nets: W2 operand of the forward
    return a;
}

// ---- MonitorWrapperEnv helpers ----
MonitorArgs monitor_step_args(dril_handle* h) {   // one env step: finished episodes land in the E-sized step arrays
    MonitorArgs m{}; if (h->mon_cur_ret) { m.cur_ret = h->mon_cur_ret; m.cur_len = h->mon_cur_len; m.ep_ret = h->e_ep_ret; m.ep_len = h->e_ep_len; m.flags_out = h->e_flags; }
    return m;
}
int monitor_collect_step(dril_handle* h) {
    if (!h->mon_cur_ret) return DRIL_OK;
    HIPCHK(h, launch_monitor_collect(h->e_flags, h->e_ep_ret, h->e_ep_len, h->cfg.n_envs, 1, h->cfg.monitor_window, h->mon_cnt, h->mon_ring_ret, h->mon_ring_len, h->mon_meta, h->stream));
    return DRIL_OK;
}
int monitor_collect_rollout(dril_handle* h) {
    if (!h->mon_cur_ret) return DRIL_OK;
    HIPCHK(h, launch_monitor_collect(h->flags, h->ep_ret, h->ep_len, h->cfg.n_envs, h->cfg.n_steps, h->cfg.monitor_window, h->mon_cnt, h->mon_ring_ret, h->mon_ring_len, h->mon_meta, h->stream));
    return DRIL_OK;
}

// data-parallel runs: NormalizeWrapperEnv's batch moments cover every env of the job (the reference has ONE vector env, normalizeWrapperEnv.jl:21-26):
// this rank's partial table is folded to one row, RCCL sums the rows, and the apply kernels merge that row with n_stats = world * E.  One 128-byte
// all-reduce per env step; every rank applies the identical update, so the running statistics stay bit-identical across ranks
int global_partials(dril_handle* h, bool update, const double*& partials, int& nb, long long& n_stats) {
    partials = h->rms_partials; n_stats = 0;
    const int world = comm_ready(h) ? h->cfg.world_size : 1;
    if (!update || !(world > 1 || (comm_ready(h) && h->force_allreduce))) return DRIL_OK;
    HIPCHK(h, launch_fold_partials(h->rms_partials, nb, h->rms_red, h->stream));
    int rc = rccl_allreduce(h, h->rms_red, 16, kNcclFloat64); if (rc) return rc;
    partials = h->rms_red; nb = 1; n_stats = (long long)h->cfg.n_envs * world;
    return DRIL_OK;
}

// ---- step-granular env verbs on device (NormalizeWrapperEnv.observe / act!, normalizeWrapperEnv.jl:123-165) ----
int observe_dev(dril_handle* h, bool update_stats) {
    const int E = h->cfg.n_envs;
    int nb = (E + 255) / 256; if (nb > h->rms_blocks) nb = h->rms_blocks;
    HIPCHK(h, launch_obs_partials(h->cfg.env_kind, E, h->state, h->e_obs_raw, h->rms_partials, nb, h->stream));
    NormObsArgs a{};
    a.E = E; a.D = h->D; a.update = (update_stats && h->cfg.norm_training && h->cfg.norm_obs) ? 1 : 0;
    { int rcg = global_partials(h, a.update != 0, a.partials, nb, a.n_stats); if (rcg) return rcg; }
    a.nblocks = nb;
    a.raw = h->e_obs_raw; a.in = h->obs_rms + h->obs_par; a.out = h->obs_rms + (h->obs_par ^ 1);
    a.obs_n = h->e_obs; a.clip = h->cfg.clip_obs; a.eps = h->cfg.norm_epsilon; a.norm_obs = h->cfg.norm_obs;
    HIPCHK(h, launch_norm_obs_apply(a, h->stream));
    h->obs_par ^= 1;
    return DRIL_OK;
}
// actions: device pointer (stored/raw policy actions; the kernels apply the adapters); rew_out/flags_out: device destinations
int step_dev(dril_handle* h, const void* actions, float* rew_out, uint8_t* flags_out) {
    const int E = h->cfg.n_envs;
    HIPCHK(h, launch_env_step(h->cfg.env_kind, E, h->env_seed0, h->cfg.episode_len, h->cfg.fixed_length_episodes, h->cfg.action_start, actions,
                              h->state, h->step_count, h->episode, h->gstep, h->e_rew, h->e_term, h->e_trunc, h->e_tobs, monitor_step_args(h), h->stream));
    { int rcm = monitor_collect_step(h); if (rcm) return rcm; }
    int nb = (E + 255) / 256; if (nb > h->rms_blocks) nb = h->rms_blocks;
    const int upd = (h->cfg.norm_reward && h->cfg.norm_training) ? 1 : 0;
    HIPCHK(h, launch_rew_partials(E, h->e_rew, h->disc_returns, h->cfg.norm_gamma, upd, h->rms_partials, nb, h->stream));
    NormRewArgs a{};
    a.E = E; a.D = h->D; a.update = upd; a.norm_obs = h->cfg.norm_obs; a.norm_reward = h->cfg.norm_reward;
    { int rcg = global_partials(h, upd != 0, a.partials, nb, a.n_stats); if (rcg) return rcg; }
    a.nblocks = nb;
    a.rew_raw = h->e_rew; a.in = h->ret_rms + h->ret_par; a.out = h->ret_rms + (h->ret_par ^ 1);
    a.obs_stats = h->obs_rms + h->obs_par; a.rew_out = rew_out; a.disc_returns = h->disc_returns; a.term = h->e_term; a.trunc = h->e_trunc;
    a.tobs = h->e_tobs; a.clip_obs = h->cfg.clip_obs; a.clip_reward = h->cfg.clip_reward; a.eps = h->cfg.norm_epsilon; a.flags_out = flags_out;
    HIPCHK(h, launch_norm_rew_apply(a, h->stream));
    h->ret_par ^= 1;
    return DRIL_OK;
}

// one optimiser step on [pos0, pos0+count) of the current epoch order; all launches asynchronous
int ppo_step(dril_handle* h, const float* obs, const void* actions, const float* adv, const float* ret, const float* logp_old,
             const float* val_old, const int64_t* perm, int64_t pos0, int64_t count, int64_t N, uint64_t key, int bits,
             float* step_stats, bool apply, const float4* rec = nullptr, const double* pre_stats = nullptr, const int32_t* perm32 = nullptr) {
    const int world = comm_ready(h) ? h->cfg.world_size : 1;
    const bool reduce = world > 1 || (comm_ready(h) && h->force_allreduce);   // force: exercise the RCCL path on one rank (tests)
    const int64_t tiles = (count + kTile - 1) / kTile;
    int G = h->wide ? (int)(tiles < h->Gmax ? tiles : h->Gmax) : (int)((tiles + 3) / 4); if (G > h->Gmax) G = h->Gmax; if (G < 1) G = 1;
    // hidden [64,64]: large minibatches run ppo_grad_pair_kernel (bf16 matrix cores, two waves per tile, two workgroups of two pairs per CU); small ones keep
    // the f32 kernel, whose weight staging is cheaper (no operand split per workgroup) and which the launch-bound small path is tuned for
    int variant = 0;
    if (h->wide) variant = (wide_variant(h) && rec) ? 1 : 0;
    if (h->wide && variant == 1) { const int64_t passes = (tiles + kWideSplitNT - 1) / kWideSplitNT; G = (int)(passes < h->Gmax ? passes : h->Gmax); if (G < 1) G = 1; }   // ppo_grad_wide_split_kernel takes kWideSplitNT sample tiles per pass
    if (!h->wide && !h->generic) {
        variant = h->grad_variant == 0 ? 0 : h->grad_variant > 0 ? 2 : (tiles >= kPairTilesPerCu * (int64_t)h->num_cus ? 2 : 0);   // large minibatches: the pair kernel
        if (variant == 2 && !(pair_variant(h) && rec && h->Gmax >= 2)) variant = 0;   // the pair kernel reads packed records and needs two slabs per workgroup (DRIL_GRAD_GMAX=1: not the pair kernel)
        if (variant == 2) { G = (int)(tiles < h->Gmax ? tiles : h->Gmax) & ~1; if (G < 2) G = 2; }   // pairs per net (even: two pairs per workgroup)
        if (variant == 0 && G > h->num_cus) G = h->num_cus;                                   // Gmax is sized for the pair kernel's slabs; the f32 kernel runs two workgroups per CU
    }
    int Gc = G;
    if (variant == 2 && 2 * G >= 4 * h->num_cus) {     // pair kernel filling the chip (two workgroups of two pairs per CU): the pairs are divided between the nets by their cost per tile
        const int pml = h->grad_actor_pct ? h->grad_actor_pct : (h->discrete ? 500 : 450), total = 4 * h->num_cus;   // measured optima (re-swept on the f16 arithmetic at the end of round 3): the older (actor) workgroups win the issue arbitration on a shared SIMD
        int ga = ((total * pml + 500) / 1000) & ~1; if (ga < 2) ga = 2; if (ga > total - 2) ga = total - 2;
        G = ga; Gc = total - ga;
        if (G > h->Gmax) G = h->Gmax & ~1; if (Gc > h->Gmax) Gc = h->Gmax & ~1;
    }
    if (h->generic) { G = generic_pick_slabs(h->gd, count, h->Gmax); if (G < 1) return fail(h, DRIL_ERR_UNSUPPORTED, "minibatch too large for the generic path's workspace"); Gc = G; }
    { int rcw = ensure_wimg(h); if (rcw) return rcw; }
    h->last_variant = h->generic ? 3 : h->wide ? (variant ? 4 : 2) : (variant == 2 ? 5 : variant);
    if (h->last_variant == 4 || h->last_variant == 5) h->used_f16 = true;
    const double* adv_stats = h->adv_stats;
    // launch-bound regime (the reference's default batch_size = 64): the advantage moments are computed inside the grad kernel and
    // reduce + norm + Adam run as one workgroup: 2 dependent launches per optimiser step instead of 6
    const bool small = !reduce && !h->wide && !h->generic && count <= 4096 && !h->no_small_path && variant != 2;   // (the pair kernel has no in-kernel moments: forced onto a small minibatch it takes the general path)
    if (h->cfg.normalize_advantage && pre_stats) adv_stats = pre_stats;   // per-epoch table (already all-reduced in data-parallel runs)
    else if (h->cfg.normalize_advantage && small) adv_stats = nullptr;
    else if (h->cfg.normalize_advantage) {
        MomentsArgs m{}; m.adv = adv; m.perm = perm; m.perm32 = perm32; m.pos0 = pos0; m.count = count; m.N = N; m.idx_lo = 0; m.n_local = N;
        m.perm_key = key; m.perm_bits = bits; m.partials = h->adv_partials; m.stop_flag = h->stop_flag;
        int nb = (int)((count + 255) / 256); if (nb > h->adv_blocks) nb = h->adv_blocks; if (nb < 1) nb = 1;
        prof_begin(h, DRIL_K_ADV_MOMENTS);
        HIPCHK(h, launch_adv_moments(m, nb, h->stream));
        HIPCHK(h, launch_moments_finalize(h->adv_partials, nb, h->adv_stats, (double)count, h->stop_flag, h->stream));
        prof_end(h);
        if (reduce) { int rc = rccl_allreduce(h, h->adv_stats, 3, kNcclFloat64); if (rc) return rc; }
    }
    GradArgs g{};
    g.params = h->params; g.obs = obs; g.actions = actions; g.adv = adv; g.ret = ret; g.logp_old = logp_old; g.val_old = val_old;
    g.perm = perm; g.perm32 = perm32; g.pos0 = pos0; g.count = count; g.N = N; g.idx_lo = 0; g.n_local = N; g.perm_key = key; g.perm_bits = bits;
    g.w2p_actor = (const u32x4*)h->w2p_actor; g.w2tp_actor = (const u32x4*)h->w2tp_actor; g.w2p_critic = (const u32x4*)h->w2p_critic; g.w2tp_critic = (const u32x4*)h->w2tp_critic;
    g.rec = rec; g.w2a_actor = h->w2a_actor; g.w2ta_actor = h->w2ta_actor; g.w2a_critic = h->w2a_critic; g.w2ta_critic = h->w2ta_critic;
    g.adv_stats = adv_stats; g.inline_moments = (h->cfg.normalize_advantage && adv_stats == nullptr) ? 1 : 0; g.invB = 1.0f / (float)(count * world);
    g.clip_range = h->cfg.clip_range; g.ent_coef = h->cfg.ent_coef; g.vf_coef = h->cfg.vf_coef; g.clip_range_vf = h->cfg.clip_range_vf;
    g.has_clip_vf = h->cfg.has_clip_range_vf; g.normalize_adv = h->cfg.normalize_advantage; g.action_start = h->cfg.action_start;
    g.log_std_off = h->log_std_off; g.slabs_actor = h->slabs_a; g.slabs_critic = h->slabs_c; g.slab_a = h->slab_a; g.slab_c = h->slab_c;
    g.G = G; g.Gc = Gc; g.dbg = h->dbg; g.variant = variant; g.stop_flag = h->stop_flag; g.actor = h->actor; g.critic = h->critic;
    // host-side preconditions of the gradient kernels: a violated one would be a device fault (null advantage moments in a kernel without in-kernel
    // moments — the SIGABRT of profiles/r02_split_kernel.md "the abort of 09:41" —, a slab index past the Gmax slabs that were allocated); only a status code
    // may cross the ABI
    const bool kernel_has_inline_moments = !h->wide && !h->generic && variant != 2;
    if (g.normalize_adv && !adv_stats && !kernel_has_inline_moments) return fail(h, DRIL_ERR_INVALID_ARG, "ppo_step: advantage moments missing for a gradient kernel without in-kernel moments");
    if (G < 1 || Gc < 1 || G > h->Gmax || Gc > h->Gmax || (variant == 2 && !h->wide && !h->generic && ((G | Gc) & 1))) return fail(h, DRIL_ERR_INVALID_ARG, "ppo_step: gradient grid does not fit the slab buffers");
    prof_begin(h, DRIL_K_PPO_GRAD);
    if (h->generic) HIPCHK(h, generic_ppo_grad(h->gd, g, h->gws, h->stream));
    else HIPCHK(h, launch_ppo_grad(h->cfg.env_kind, h->cfg.hidden1, g, h->stream));
    prof_end(h);
    ReduceArgs r{};
    r.slabs_actor = h->slabs_a; r.slabs_critic = h->slabs_c; r.slab_a = h->slab_a; r.slab_c = h->slab_c; r.G = G; r.Gc = Gc;
    if (h->last_variant == 5) { r.G = G / 2; r.Gc = Gc / 2; }                       // ppo_grad_pair_kernel: G / Gc count pairs, its workgroups (two pairs each) write one slab
    r.P = h->P; r.Pa = h->Pa; r.Pc = h->Pc; r.flat = h->flat; r.norm_partials = h->norm_partials; r.n_samples_local = (double)count;
    r.stop_flag = h->stop_flag;
    if (small && apply && h->P <= 16384 && G <= 32) {
        AdamArgs ad{};
        ad.params = h->params; ad.m = h->adam_m; ad.v = h->adam_v; ad.flat = h->flat; ad.P = h->P;
        ad.norm_partials = h->norm_partials; ad.n_partials = h->n_norm_partials; ad.bt = h->bt; ad.step_parity = (int)(h->adam_steps & 1);
        ad.beta1 = h->cfg.adam_beta1; ad.beta2 = h->cfg.adam_beta2; ad.eps = h->cfg.adam_eps; ad.lr = h->lr;
        ad.max_grad_norm = h->cfg.max_grad_norm; ad.target_kl = h->cfg.target_kl; ad.ent_coef = h->cfg.ent_coef; ad.vf_coef = h->cfg.vf_coef;
        ad.has_max_grad_norm = h->cfg.has_max_grad_norm; ad.has_target_kl = h->cfg.has_target_kl; ad.use_stats = 1;
        ad.step_stats = step_stats; ad.norm_out = h->norm_out; ad.nan_flag = h->nan_flag; ad.stop_flag = h->stop_flag; ad.stop_flag_w = h->stop_flag;
        prof_begin(h, DRIL_K_ADAM);
        HIPCHK(h, launch_finish_small(r, ad, h->stream));
        prof_end(h);
        h->adam_steps += 1; h->wimg_dirty = true;
        return DRIL_OK;
    }
    prof_begin(h, DRIL_K_GRAD_REDUCE);
    HIPCHK(h, launch_grad_reduce(r, h->stream));
    prof_end(h);
    const bool norm_in_adam = reduce && apply && h->P <= 65536;                        // data-parallel: adam_kernel sums |g|^2 from the all-reduced gradient itself (one launch less per step)
    if (reduce) {
        int rc = rccl_allreduce(h, h->flat, (size_t)h->P + 8, kNcclFloat32); if (rc) return rc;
        if (!norm_in_adam) HIPCHK(h, launch_grad_norm(h->flat, h->P, h->norm_partials, h->stop_flag, h->stream));
    }
    if (!apply) return DRIL_OK;
    AdamArgs ad{}; ad.norm_from_flat = norm_in_adam ? 1 : 0;
    ad.params = h->params; ad.m = h->adam_m; ad.v = h->adam_v; ad.flat = h->flat; ad.P = h->P;
    ad.norm_partials = h->norm_partials; ad.n_partials = h->n_norm_partials; ad.bt = h->bt; ad.step_parity = (int)(h->adam_steps & 1);
    ad.beta1 = h->cfg.adam_beta1; ad.beta2 = h->cfg.adam_beta2; ad.eps = h->cfg.adam_eps; ad.lr = h->lr;
    ad.max_grad_norm = h->cfg.max_grad_norm; ad.target_kl = h->cfg.target_kl; ad.ent_coef = h->cfg.ent_coef; ad.vf_coef = h->cfg.vf_coef;
    ad.has_max_grad_norm = h->cfg.has_max_grad_norm; ad.has_target_kl = h->cfg.has_target_kl; ad.use_stats = 1;
    ad.step_stats = step_stats; ad.norm_out = h->norm_out; ad.nan_flag = h->nan_flag; ad.stop_flag = h->stop_flag; ad.stop_flag_w = h->stop_flag;
    prof_begin(h, DRIL_K_ADAM);
    HIPCHK(h, launch_adam(ad, h->stream));
    prof_end(h);
    h->adam_steps += 1; h->wimg_dirty = true;
    return DRIL_OK;
}

uint64_t perm_key(uint64_t seed, uint64_t counter, int epoch) {
    uint32_t r[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), 2, (uint32_t)epoch, r);
    return ((uint64_t)r[0] << 32) | r[1];
}

int reset_optimizer(dril_handle* h) {
    HIPCHK(h, hipMemsetAsync(h->adam_m, 0, sizeof(float) * h->P, h->stream));
    HIPCHK(h, hipMemsetAsync(h->adam_v, 0, sizeof(float) * h->P, h->stream));
    const float bt[4] = {h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_beta1, h->cfg.adam_beta2};
    HIPCHK(h, hipMemcpyAsync(h->bt, bt, sizeof(bt), hipMemcpyHostToDevice, h->stream));
    h->adam_steps = 0;
    return sync(h);
}

void* buf_ptr(dril_handle* h, int which, size_t* bytes) {
    const size_t N = (size_t)h->N;
    switch (which) {
    case DRIL_BUF_OBSERVATIONS: *bytes = N * h->D * 4; return h->obs;
    case DRIL_BUF_ACTIONS: *bytes = N * act_bytes_per(h); return h->act;
    case DRIL_BUF_REWARDS: *bytes = N * 4; return h->rew;
    case DRIL_BUF_ADVANTAGES: *bytes = N * 4; return h->adv;
    case DRIL_BUF_RETURNS: *bytes = N * 4; return h->ret;
    case DRIL_BUF_LOGPROBS: *bytes = N * 4; return h->logp;
    case DRIL_BUF_VALUES: *bytes = N * 4; return h->val;
    case DRIL_BUF_FLAGS: *bytes = N; return h->flags;
    case DRIL_BUF_BOOTSTRAP: *bytes = N * 4; return h->boot;
    case DRIL_BUF_LAST_VALUES: *bytes = (size_t)h->cfg.n_envs * 4; return h->last_values;
    }
    *bytes = 0; return nullptr;
}

}  // namespace

// ================================================================================================
DRIL_EXPORT int32_t dril_config_default(dril_config* c, int32_t env_kind) {
    if (!c || env_kind < DRIL_ENV_CARTPOLE || env_kind > DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) return fail(nullptr, DRIL_ERR_INVALID_ARG, "bad cfg/env_kind");
    std::memset(c, 0, sizeof(*c));
    c->abi_version = DRIL_ABI_VERSION; c->env_kind = env_kind; c->n_envs = 4; c->n_steps = 2048; c->hidden1 = c->hidden2 = 64;
    c->episode_len = (env_kind == DRIL_ENV_CARTPOLE || env_kind == DRIL_ENV_ACROBOT) ? 500 : (env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS || env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) ? 999 : 200; c->action_start = 1;   // the Gymnasium time limits
    c->gamma = 0.99f; c->gae_lambda = 0.95f; c->clip_range = 0.2f; c->ent_coef = 0.0f; c->vf_coef = 0.5f;
    c->max_grad_norm = 0.5f; c->has_max_grad_norm = 1; c->normalize_advantage = 1; c->batch_size = 64; c->epochs = 10;
    c->learning_rate = 3.0e-4f; c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1.0e-5f;
    c->clip_obs = c->clip_reward = 10.0f; c->norm_gamma = 0.99f; c->norm_epsilon = 1.0e-8f; c->seed = 42; c->world_size = 1;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_create(const dril_config* cfg, dril_handle** out) {
    if (!cfg || !out) return fail(nullptr, DRIL_ERR_INVALID_ARG, "null cfg/out");
    if (cfg->abi_version != DRIL_ABI_VERSION) return fail(nullptr, DRIL_ERR_INVALID_ARG, "abi_version mismatch");
    if (cfg->env_kind < DRIL_ENV_CARTPOLE || cfg->env_kind > DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) return fail(nullptr, DRIL_ERR_INVALID_ARG, "unknown env_kind");
    const bool ext = cfg->env_kind == DRIL_ENV_EXTERNAL;
    if (ext && (cfg->ext_obs_dim < 1 || cfg->ext_obs_dim > 1024 || cfg->ext_action_dim < 1 || cfg->ext_action_dim > 64)) return fail(nullptr, DRIL_ERR_INVALID_ARG, "DRIL_ENV_EXTERNAL: ext_obs_dim must be 1..1024 and ext_action_dim 1..64");
    // hidden_dims / activation: n_hidden == 0 is the two-layer form (hidden1, hidden2); otherwise hidden[0 .. n_hidden-1]
    if (cfg->n_hidden < 0 || cfg->n_hidden > kMaxHidden) return fail(nullptr, DRIL_ERR_INVALID_ARG, "n_hidden must be 0 (hidden1 / hidden2) or 1..4");
    if (cfg->activation < 0 || cfg->activation > 7) return fail(nullptr, DRIL_ERR_UNSUPPORTED, "activation must be 0 (tanh), 1 (relu), 2 (sigmoid), 3 (elu), 4 (leakyrelu), 5 (softplus), 6 (gelu) or 7 (swish)");
    int nh = cfg->n_hidden ? cfg->n_hidden : 2, hd[kMaxHidden] = {cfg->hidden1, cfg->hidden2, 0, 0};
    if (cfg->n_hidden) for (int l = 0; l < nh; ++l) hd[l] = cfg->hidden[l];
    for (int l = 0; l < nh; ++l) if (hd[l] < 1 || hd[l] > 1024) return fail(nullptr, DRIL_ERR_INVALID_ARG, "hidden widths must be 1..1024");
    if (ext && (cfg->norm_obs || cfg->norm_reward || cfg->monitor_window)) return fail(nullptr, DRIL_ERR_UNSUPPORTED, "DRIL_ENV_EXTERNAL: NormalizeWrapperEnv / MonitorWrapperEnv wrap the host env on the host");
    if (cfg->n_envs < 1 || cfg->n_steps < 1 || cfg->epochs < 0 || cfg->batch_size < 1) return fail(nullptr, DRIL_ERR_INVALID_ARG, "n_envs/n_steps/batch_size must be positive");
    const bool fused_shape = nh == 2 && cfg->activation == 0 && hd[0] == hd[1] && (hd[0] == 32 || hd[0] == 64 || hd[0] == 128 || hd[0] == 256);   // 32: the reference's benchmark-suite shape (f32-MFMA kernels only)   // everything else: the generic kernels (any depth <= 4, any width <= 1024, tanh / relu)
    if (cfg->monitor_window < 0) return fail(nullptr, DRIL_ERR_INVALID_ARG, "monitor_window must be >= 0");
    if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail(nullptr, DRIL_ERR_INVALID_ARG, "bad rank/world_size");
    if (cfg->batch_size % cfg->world_size != 0) return fail(nullptr, DRIL_ERR_INVALID_ARG, "batch_size must be divisible by world_size");
    dril_handle* h = nullptr;
    try { h = new dril_handle(); } catch (...) { return fail(nullptr, DRIL_ERR_INVALID_ARG, "out of host memory"); }
    h->cfg = *cfg;
    h->cfg.hidden1 = hd[0]; h->cfg.hidden2 = nh > 1 ? hd[1] : hd[0];   // the fused kernels read hidden1 (only reached with two equal layers)
    switch (cfg->env_kind) {
        case DRIL_ENV_CARTPOLE: h->discrete = true; h->D = 4; h->A = 2; h->S = 4; break;
        case DRIL_ENV_MOUNTAINCAR: h->discrete = true; h->D = 2; h->A = 3; h->S = 2; break;
        case DRIL_ENV_MOUNTAINCAR_CONTINUOUS: case DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED: h->discrete = false; h->D = 2; h->A = 1; h->S = 2; break;
        case DRIL_ENV_ACROBOT: h->discrete = true; h->D = 6; h->A = 3; h->S = 4; h->generic = !fused_shape; break;   // six observation dims: four first-layer k-steps and three-quad records in every fused kernel (round 3)
        case DRIL_ENV_EXTERNAL: h->discrete = cfg->ext_discrete != 0; h->D = cfg->ext_obs_dim; h->A = cfg->ext_action_dim; h->S = 0; h->external = true; h->generic = true; break;
        default: h->discrete = false; h->D = 3; h->A = 1; h->S = 2; break;                    // Pendulum, ScalingWrapperEnv(Pendulum)
    }
    h->gd = GenericDims{h->D, h->A, nh, {hd[0], hd[1], hd[2], hd[3]}, h->discrete ? 1 : 0, cfg->activation};
    if (nh == 2) { h->actor = net_off(0, h->D, hd[0], hd[1], h->A); h->critic = net_off(h->actor.end, h->D, hd[0], hd[1], 1); }
    else {                                                                            // other depths exist on the generic path only, which reads just the first offset and the end of a net
        std::memset(&h->actor, 0, sizeof(h->actor)); std::memset(&h->critic, 0, sizeof(h->critic));
        h->actor.w1 = 0; h->actor.end = generic_net_size(h->gd, h->A);
        h->critic.w1 = h->actor.end; h->critic.end = h->actor.end + generic_net_size(h->gd, 1);
    }
    h->Pa = h->actor.end; h->Pc = h->critic.end - h->actor.end; h->log_std_off = h->critic.end;
    h->P = h->critic.end + (h->discrete ? 0 : h->A);
    h->N = (int64_t)cfg->n_envs * cfg->n_steps; h->lr = cfg->learning_rate;
    if (!fused_shape || std::getenv("DRIL_FORCE_GENERIC")) h->generic = true;            // DRIL_FORCE_GENERIC: run a fused-shape handle on the generic kernels (A/B and parity tests)
    if (const char* e = std::getenv("DRIL_FORCE_ALLREDUCE")) h->force_allreduce = std::atoi(e) != 0;
    if (const char* e = std::getenv("DRIL_FORCE_STEPWISE")) h->force_stepwise = std::atoi(e) != 0;
    if (const char* e = std::getenv("DRIL_GRAD_ACTOR_PERMILLE")) { h->grad_actor_pct = std::atoi(e); if (h->grad_actor_pct < 100 || h->grad_actor_pct > 900) h->grad_actor_pct = 0; }
    if (const char* e = std::getenv("DRIL_GRAD_VARIANT")) { h->grad_variant = std::atoi(e); if (h->grad_variant < -1 || h->grad_variant > 2) h->grad_variant = -1; if (h->grad_variant == 2) h->grad_variant = 1; }
    h->no_persistent = std::getenv("DRIL_NO_PERSISTENT_UPDATE") != nullptr; { const char* e = std::getenv("DRIL_NO_EPOCH_INDEX"); h->no_epoch_index = e && std::atoi(e) != 0; }
    if (const char* e = debug_env("DRIL_SMALL_CHUNK")) { const long c = std::atol(e); if (c > 0) h->small_chunk = c; }   // optimiser steps per launch of ppo_update_small_kernel (tests: launch boundaries)
    h->no_small_path = debug_env("DRIL_NO_SMALL_PATH") != nullptr;
    h->no_f32_retry = std::getenv("DRIL_NO_F32_RETRY") != nullptr;
    // Multi-process RCCL on this platform needs dmabuf IPC: with the legacy IPC mode (the ROCr default) `hipIpcGetMemHandle` fails with "invalid argument" on a
    // host driver that only supports dmabuf, and ncclCommInitRank / the first collective across processes dies with it.  The ROCr runtime reads the variable
    // when it initialises, i.e. at this process's first HIP call — which for a DRiL user is normally the hipSetDevice below.  It is only set if the caller left it
    // unset (bench.py and the tests export it themselves; a process that already touched HIP before dril_create must export it on its own: README "Multi-GPU").
    if (cfg->world_size > 1) setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", /*overwrite=*/0);
#define CCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::string m = std::string(#expr) + ": " + hipGetErrorString(_e); dril_destroy(h); return fail(nullptr, DRIL_ERR_HIP, m); } } while (0)
    CCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop; CCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { std::string m = std::string("libdril_hip targets gfx950 (MI355X) only; device is ") + prop.gcnArchName; dril_destroy(h); return fail(nullptr, DRIL_ERR_UNSUPPORTED, m); }
    h->num_cus = prop.multiProcessorCount;
    {
        char bus[64] = "?"; int ndev = -1; (void)hipDeviceGetPCIBusId(bus, (int)sizeof(bus), cfg->device); (void)hipGetDeviceCount(&ndev);
        const char* hv = std::getenv("HIP_VISIBLE_DEVICES"); const char* rv = std::getenv("ROCR_VISIBLE_DEVICES");
        h->device_info = "device " + std::to_string(cfg->device) + " of " + std::to_string(ndev) + " visible: " + prop.name + " " + prop.gcnArchName + ", " + std::to_string(prop.multiProcessorCount) + " CUs, PCI " + bus +
                         ", HIP_VISIBLE_DEVICES=" + (hv ? hv : "(unset)") + " ROCR_VISIBLE_DEVICES=" + (rv ? rv : "(unset)");
    }
    CCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    const size_t E = cfg->n_envs, N = (size_t)h->N, P = h->P;
    CCHK(dmalloc(&h->params, P)); CCHK(dmalloc(&h->adam_m, P)); CCHK(dmalloc(&h->adam_v, P)); CCHK(dmalloc(&h->bt, 4));
    CCHK(dmalloc(&h->flat, P + 8)); CCHK(dmalloc(&h->norm_out, 1)); CCHK(dmalloc(&h->w2max_dev, 1));
    { const size_t words = (size_t)gae_chunks(cfg->n_steps) * cfg->n_envs; CCHK(dmalloc(&h->gae_carry, words)); CCHK(hipMemsetAsync(h->gae_carry, 0, words * 8, h->stream));
      CCHK(dmalloc(&h->gae_err, 1)); CCHK(hipMemsetAsync(h->gae_err, 0, 4, h->stream)); }
    h->n_norm_partials = (int)((P + 31) / 32); CCHK(dmalloc(&h->norm_partials, h->n_norm_partials));
    if (h->generic) { h->slab_a = generic_slab_size(h->gd, true); h->slab_c = generic_slab_size(h->gd, false); }
    else { h->slab_a = slab_size_actor(cfg->env_kind, hd[0]); h->slab_c = slab_size_critic(cfg->env_kind, hd[0]); }
    h->wide = !h->generic && hd[0] > 64;
    h->Gmax = (h->wide && hd[0] > 128) ? (h->num_cus / 2 > 0 ? h->num_cus / 2 : 1) : (!h->wide && h->grad_variant != 0) ? 3 * h->num_cus : h->num_cus;   // H = 128: 4 waves and 77 KB LDS per workgroup, two workgroups per CU   // [64,64]: 2 workgroups per CU (actor + critic), 4 waves each; wide: 1 workgroup of H/32 waves per CU
    if (h->generic) h->Gmax = 64;                                                                                                                                                                                             // generic path: one slab per row chunk of the minibatch
    if (const char* e = debug_env("DRIL_GRAD_GMAX")) { const int g = std::atoi(e); if (g > 0 && g < h->Gmax) h->Gmax = g; }   // diagnostic: fewer workgroups per net
    if (h->wide) { const size_t pb = (size_t)hd[0] * hd[0] * 6;    // three bf16 pieces per element
        CCHK(hipMalloc(&h->w2p_actor, pb)); CCHK(hipMalloc(&h->w2tp_actor, pb)); CCHK(hipMalloc(&h->w2p_critic, pb)); CCHK(hipMalloc(&h->w2tp_critic, pb));
        CCHK(hipMalloc(&h->w2pf_actor, pb)); CCHK(hipMalloc(&h->w2pf_critic, pb)); }
    if (h->wide) { const size_t hh = (size_t)hd[0] * hd[0]; CCHK(dmalloc(&h->w2a_actor, hh)); CCHK(dmalloc(&h->w2ta_actor, hh)); CCHK(dmalloc(&h->w2a_critic, hh)); CCHK(dmalloc(&h->w2ta_critic, hh)); }
    CCHK(dmalloc(&h->slabs_a, (size_t)h->Gmax * h->slab_a)); CCHK(dmalloc(&h->slabs_c, (size_t)h->Gmax * h->slab_c));
    CCHK(dmalloc(&h->state, E * h->S)); CCHK(dmalloc(&h->step_count, E)); CCHK(dmalloc(&h->episode, E)); CCHK(dmalloc(&h->gstep, E));
    CCHK(dmalloc(&h->disc_returns, E));
    CCHK(dmalloc(&h->obs, N * h->D)); CCHK(hipMalloc(&h->act, N * act_bytes_per(h))); CCHK(dmalloc(&h->rew, N)); CCHK(dmalloc(&h->adv, N));
    CCHK(dmalloc(&h->ret, N)); CCHK(dmalloc(&h->logp, N)); CCHK(dmalloc(&h->val, N)); CCHK(dmalloc(&h->boot, N)); CCHK(dmalloc(&h->flags, N));
    CCHK(dmalloc(&h->last_values, E));
    if (!h->generic) CCHK(dmalloc(&h->rec, (size_t)(h->D <= 4 ? 2 : 3) * N));   // packed minibatch records: 2 (D <= 4) or 3 (D <= 8) float4 per sample
    if (h->generic) CCHK(dmalloc(&h->gen_tmp, E));
    if (ext) { CCHK(hipHostMalloc((void**)&h->ext_stage_rew, N * 4)); CCHK(hipHostMalloc((void**)&h->ext_stage_flags, N)); }
    if (cfg->monitor_window > 0) {
        const size_t W = cfg->monitor_window;
        CCHK(dmalloc(&h->mon_cur_ret, E)); CCHK(dmalloc(&h->mon_cur_len, E)); CCHK(dmalloc(&h->ep_ret, N)); CCHK(dmalloc(&h->ep_len, N));
        CCHK(dmalloc(&h->mon_ring_ret, W)); CCHK(dmalloc(&h->mon_ring_len, W)); CCHK(dmalloc(&h->e_ep_ret, E)); CCHK(dmalloc(&h->e_ep_len, E));
        CCHK(dmalloc(&h->mon_cnt, (size_t)cfg->n_steps)); CCHK(dmalloc(&h->mon_meta, 2)); CCHK(dmalloc(&h->e_flags, E));
        CCHK(hipMemset(h->mon_cur_ret, 0, E * 4)); CCHK(hipMemset(h->mon_cur_len, 0, E * 4)); CCHK(hipMemset(h->mon_meta, 0, 8));
    }
    h->adv_blocks = 1024; CCHK(dmalloc(&h->adv_partials, 2 * (size_t)h->adv_blocks)); CCHK(dmalloc(&h->adv_stats, 4));
    h->ev_blocks = 1024; CCHK(dmalloc(&h->ev_partials, 4 * (size_t)h->ev_blocks));
    CCHK(dmalloc(&h->stop_flag, 1)); CCHK(dmalloc(&h->nan_flag, 1));
#ifdef DRIL_STAMPS
    CCHK(dmalloc(&h->dbg, (size_t)2 * h->Gmax * 4 * 12)); CCHK(hipMemset(h->dbg, 0, (size_t)2 * h->Gmax * 4 * 12 * 8));
#endif
    CCHK(dmalloc(&h->e_obs, E * h->D)); CCHK(dmalloc(&h->e_rew, E)); CCHK(dmalloc(&h->e_tobs, E * h->D)); CCHK(dmalloc(&h->e_term, E));
    CCHK(dmalloc(&h->e_trunc, E)); CCHK(hipMalloc(&h->e_act, E * act_bytes_per(h)));
    CCHK(dmalloc(&h->e_obs_raw, E * h->D)); CCHK(dmalloc(&h->e_rew_n, E)); CCHK(dmalloc(&h->obs_rms, 2)); CCHK(dmalloc(&h->ret_rms, 2)); CCHK(dmalloc(&h->rms_partials, (size_t)h->rms_blocks * 16)); CCHK(dmalloc(&h->rms_red, 16));
    { RmsState init[2]; for (auto& r : init) { for (int d = 0; d < 8; ++d) { r.mean[d] = 0.f; r.var[d] = 1.f; } r.count = 0; }   // RunningMeanStd{T}(shape): zeros, ones, 0 (normalizeWrapperEnv.jl:14-16)
      CCHK(hipMemcpy(h->obs_rms, init, sizeof(init), hipMemcpyHostToDevice)); CCHK(hipMemcpy(h->ret_rms, init, sizeof(init), hipMemcpyHostToDevice)); }
    CCHK(hipMemsetAsync(h->params, 0, P * 4, h->stream)); CCHK(hipMemsetAsync(h->boot, 0, N * 4, h->stream));
    CCHK(hipMemsetAsync(h->flags, 0, N, h->stream)); CCHK(hipMemsetAsync(h->last_values, 0, E * 4, h->stream));
    CCHK(hipMemsetAsync(h->stop_flag, 0, 4, h->stream)); CCHK(hipMemsetAsync(h->nan_flag, 0, 4, h->stream));
    CCHK(hipMemsetAsync(h->e_tobs, 0, E * h->D * 4, h->stream));
    if (!h->discrete) { std::vector<float> ls(h->A, cfg->log_std_init); CCHK(hipMemcpyAsync(h->params + h->log_std_off, ls.data(), 4 * h->A, hipMemcpyHostToDevice, h->stream)); CCHK(hipStreamSynchronize(h->stream)); }
#undef CCHK
    int rc = reset_optimizer(h);
    if (rc) { std::string m = h->err; dril_destroy(h); return fail(nullptr, rc, m); }
    *out = h;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_destroy(dril_handle* h) {
    if (h) (void)hipSetDevice(h->cfg.device);
    if (!h) return DRIL_OK;
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    if (h->loop) {                                                   // the last handle to leave frees the group
        LoopGroup* g = h->loop; bool last;
        { std::lock_guard<std::mutex> lk(g->mu); last = --g->refs == 0; }
        if (last) { for (int q = 0; q < g->n; ++q) if (g->ready[q]) hipEventDestroy(g->ready[q]); if (g->done) hipEventDestroy(g->done); delete g; }
        h->loop = nullptr;
    }
    generic_ws_free(h->gws);
    if (h->ext_stage_rew) (void)hipHostFree(h->ext_stage_rew); if (h->ext_stage_flags) (void)hipHostFree(h->ext_stage_flags);
    void* ptrs[] = {h->params, h->adam_m, h->adam_v, h->bt, h->flat, h->norm_out, h->norm_partials, h->retry_snap, h->slabs_a, h->slabs_c, h->state,
                    h->step_count, h->episode, h->gstep, h->disc_returns, h->obs, h->act, h->rew, h->adv, h->ret, h->logp, h->val, h->boot,
                    h->flags, h->last_values, h->noise_dev, h->perm_dev, h->epoch_index, h->epoch_keys, h->small_xchg, h->w2max_dev, h->gae_carry, h->gae_err, h->w2pf_actor, h->w2pf_critic, h->adv_partials, h->adv_stats, h->ev_partials, h->step_stats,
                    h->stop_flag, h->nan_flag, h->e_obs, h->e_rew, h->e_tobs, h->e_term, h->e_trunc, h->e_act, h->e_obs_raw, h->e_rew_n, h->obs_rms, h->ret_rms, h->rms_partials, h->rms_red, h->gen_tmp, h->dbg, h->rec, h->epoch_tables, h->epoch_stats, h->w2a_actor, h->w2ta_actor, h->w2a_critic, h->w2ta_critic, h->w2p_actor, h->w2tp_actor, h->w2p_critic, h->w2tp_critic, h->mon_cur_ret, h->ep_ret, h->mon_ring_ret, h->e_ep_ret, h->mon_cur_len, h->ep_len,
                    h->mon_ring_len, h->e_ep_len, h->mon_cnt, h->mon_meta, h->e_flags};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& p : h->prof_pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto& p : h->prof_pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return DRIL_OK;
}
DRIL_EXPORT const char* dril_last_error(const dril_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }
DRIL_EXPORT int32_t dril_synchronize(dril_handle* h) { NEED(h); return sync(h); }
DRIL_EXPORT int32_t dril_obs_dim(const dril_handle* h) { return h ? h->D : -1; }
DRIL_EXPORT int32_t dril_action_dim(const dril_handle* h) { return h ? h->A : -1; }
DRIL_EXPORT int32_t dril_is_discrete(const dril_handle* h) { return h ? (h->discrete ? 1 : 0) : -1; }
DRIL_EXPORT int64_t dril_param_count(const dril_handle* h) { return h ? h->P : -1; }

DRIL_EXPORT int32_t dril_set_params(dril_handle* h, const float* flat, size_t n) {
    NEED(h); if (!flat || n != (size_t)h->P) return fail(h, DRIL_ERR_INVALID_ARG, "dril_set_params: n != dril_param_count");
    HIPCHK(h, hipMemcpyAsync(h->params, flat, n * 4, hipMemcpyHostToDevice, h->stream)); h->wimg_dirty = true;
    if (!h->generic) {                                                                // max |W2| of the new parameters decides the forward's arithmetic (fwd_exact)
        const size_t HH = (size_t)h->cfg.hidden1 * h->cfg.hidden1; float m = 0.f;
        for (const float* w : {flat + h->actor.w2, flat + h->critic.w2}) for (size_t i = 0; i < HH; ++i) { const float x = std::fabs(w[i]); m = (x > m || x != x) ? (x != x ? INFINITY : x) : m; }
        h->w2max = m;
    }
    return sync(h);
}
DRIL_EXPORT int32_t dril_get_params(dril_handle* h, float* flat, size_t n) {
    NEED(h); if (!flat || n != (size_t)h->P) return fail(h, DRIL_ERR_INVALID_ARG, "dril_get_params: n != dril_param_count");
    HIPCHK(h, hipMemcpyAsync(flat, h->params, n * 4, hipMemcpyDeviceToHost, h->stream)); return sync(h);
}
DRIL_EXPORT int32_t dril_reset_optimizer(dril_handle* h) { NEED(h); return reset_optimizer(h); }
DRIL_EXPORT int32_t dril_set_learning_rate(dril_handle* h, float lr) { NEED(h); h->lr = lr; return DRIL_OK; }
DRIL_EXPORT int32_t dril_get_optimizer_state(dril_handle* h, float* m, float* v, size_t n, float* beta_powers, int64_t* steps) {
    NEED(h);
    if (!m || !v || !beta_powers || n != (size_t)h->P) return fail(h, DRIL_ERR_INVALID_ARG, "dril_get_optimizer_state: n != dril_param_count or null pointer");
    float bt[4];
    HIPCHK(h, hipMemcpyAsync(m, h->adam_m, n * 4, hipMemcpyDeviceToHost, h->stream)); HIPCHK(h, hipMemcpyAsync(v, h->adam_v, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(bt, h->bt, sizeof(bt), hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    const int slot = 2 * (int)(h->adam_steps & 1);                                     // the slot the NEXT step reads (ping-pong)
    beta_powers[0] = bt[slot]; beta_powers[1] = bt[slot + 1];
    if (steps) *steps = h->adam_steps;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_set_optimizer_state(dril_handle* h, const float* m, const float* v, size_t n, const float* beta_powers, int64_t steps) {
    NEED(h);
    if (!m || !v || !beta_powers || n != (size_t)h->P || steps < 0) return fail(h, DRIL_ERR_INVALID_ARG, "dril_set_optimizer_state: n != dril_param_count, null pointer or negative step count");
    const float bt[4] = {beta_powers[0], beta_powers[1], beta_powers[0], beta_powers[1]};
    HIPCHK(h, hipMemcpyAsync(h->adam_m, m, n * 4, hipMemcpyHostToDevice, h->stream)); HIPCHK(h, hipMemcpyAsync(h->adam_v, v, n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->bt, bt, sizeof(bt), hipMemcpyHostToDevice, h->stream));
    h->adam_steps = steps;
    return sync(h);
}

// ---- env verbs -----------------------------------------------------------------------------------
DRIL_EXPORT int32_t dril_env_reset(dril_handle* h, uint64_t seed) {
    NEED(h); NOT_EXTERNAL(h, "dril_env_reset");
    h->env_seed0 = seed + (uint64_t)h->cfg.rank * (uint64_t)h->cfg.n_envs;
    HIPCHK(h, launch_env_reset(h->cfg.env_kind, h->cfg.n_envs, h->env_seed0, h->state, h->step_count, h->episode, h->gstep, h->disc_returns, h->stream));
    if (h->mon_cur_ret) { HIPCHK(h, hipMemsetAsync(h->mon_cur_ret, 0, (size_t)h->cfg.n_envs * 4, h->stream)); HIPCHK(h, hipMemsetAsync(h->mon_cur_len, 0, (size_t)h->cfg.n_envs * 4, h->stream)); }   // MonitorWrapperEnv.reset! :38-44
    h->env_ready = true;
    return sync(h);
}
DRIL_EXPORT int32_t dril_env_observe(dril_handle* h, float* host_obs, int32_t update_stats) {
    NEED(h); NOT_EXTERNAL(h, "dril_env_observe");
    if (!h->env_ready) return fail(h, DRIL_ERR_NOT_INITIALISED, "dril_env_observe before dril_env_reset");
    if (!host_obs) return fail(h, DRIL_ERR_INVALID_ARG, "null host_obs");
    if (normalizing(h)) { int rc = observe_dev(h, update_stats != 0); if (rc) return rc; }
    else HIPCHK(h, launch_env_observe(h->cfg.env_kind, h->cfg.n_envs, h->state, h->e_obs, h->stream));
    HIPCHK(h, hipMemcpyAsync(host_obs, h->e_obs, (size_t)h->cfg.n_envs * h->D * 4, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}
DRIL_EXPORT int32_t dril_env_step(dril_handle* h, const void* actions, float* rewards, uint8_t* terminated, uint8_t* truncated, float* terminal_obs) {
    NEED(h); NOT_EXTERNAL(h, "dril_env_step");
    if (!h->env_ready) return fail(h, DRIL_ERR_NOT_INITIALISED, "dril_env_step before dril_env_reset");
    if (!actions) return fail(h, DRIL_ERR_INVALID_ARG, "null actions");
    const size_t E = h->cfg.n_envs;
    HIPCHK(h, hipMemcpyAsync(h->e_act, actions, E * act_bytes_per(h), hipMemcpyHostToDevice, h->stream));
    float* rew_dev = h->e_rew;
    if (normalizing(h)) { rew_dev = h->e_rew_n; int rc = step_dev(h, h->e_act, rew_dev, nullptr); if (rc) return rc; }
    else {
        HIPCHK(h, launch_env_step(h->cfg.env_kind, (int)E, h->env_seed0, h->cfg.episode_len, h->cfg.fixed_length_episodes, h->cfg.action_start,
                                  h->e_act, h->state, h->step_count, h->episode, h->gstep, h->e_rew, h->e_term, h->e_trunc, h->e_tobs, monitor_step_args(h), h->stream));
        int rcm = monitor_collect_step(h); if (rcm) return rcm;
    }
    if (rewards) HIPCHK(h, hipMemcpyAsync(rewards, rew_dev, E * 4, hipMemcpyDeviceToHost, h->stream));
    if (terminated) HIPCHK(h, hipMemcpyAsync(terminated, h->e_term, E, hipMemcpyDeviceToHost, h->stream));
    if (truncated) HIPCHK(h, hipMemcpyAsync(truncated, h->e_trunc, E, hipMemcpyDeviceToHost, h->stream));
    if (terminal_obs) HIPCHK(h, hipMemcpyAsync(terminal_obs, h->e_tobs, E * h->D * 4, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}
DRIL_EXPORT int32_t dril_env_get_state(dril_handle* h, float* state, int32_t* step_count) {
    NEED(h); NOT_EXTERNAL(h, "dril_env_get_state"); if (!state) return fail(h, DRIL_ERR_INVALID_ARG, "null state");
    HIPCHK(h, hipMemcpyAsync(state, h->state, (size_t)h->cfg.n_envs * h->S * 4, hipMemcpyDeviceToHost, h->stream));
    if (step_count) HIPCHK(h, hipMemcpyAsync(step_count, h->step_count, (size_t)h->cfg.n_envs * 4, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}
DRIL_EXPORT int32_t dril_env_set_state(dril_handle* h, const float* state, const int32_t* step_count) {
    NEED(h); NOT_EXTERNAL(h, "dril_env_set_state"); if (!state) return fail(h, DRIL_ERR_INVALID_ARG, "null state");
    HIPCHK(h, hipMemcpyAsync(h->state, state, (size_t)h->cfg.n_envs * h->S * 4, hipMemcpyHostToDevice, h->stream));
    if (step_count) HIPCHK(h, hipMemcpyAsync(h->step_count, step_count, (size_t)h->cfg.n_envs * 4, hipMemcpyHostToDevice, h->stream));
    return sync(h);
}
DRIL_EXPORT int32_t dril_norm_get_stats(dril_handle* h, float* obs_mean, float* obs_var, int64_t* obs_count, float* ret_mean, float* ret_var, int64_t* ret_count) {
    NEED(h); NOT_EXTERNAL(h, "dril_norm_get_stats");
    RmsState o, r;
    HIPCHK(h, hipMemcpyAsync(&o, h->obs_rms + h->obs_par, sizeof(o), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&r, h->ret_rms + h->ret_par, sizeof(r), hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    if (obs_mean) std::memcpy(obs_mean, o.mean, 4 * h->D); if (obs_var) std::memcpy(obs_var, o.var, 4 * h->D); if (obs_count) *obs_count = o.count;
    if (ret_mean) *ret_mean = r.mean[0]; if (ret_var) *ret_var = r.var[0]; if (ret_count) *ret_count = r.count;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_norm_set_stats(dril_handle* h, const float* obs_mean, const float* obs_var, int64_t obs_count, float ret_mean, float ret_var, int64_t ret_count) {
    NEED(h); NOT_EXTERNAL(h, "dril_norm_set_stats");
    if (!obs_mean || !obs_var) return fail(h, DRIL_ERR_INVALID_ARG, "null statistics");
    RmsState o{}, r{};
    for (int d = 0; d < 8; ++d) { o.mean[d] = d < h->D ? obs_mean[d] : 0.f; o.var[d] = d < h->D ? obs_var[d] : 1.f; r.mean[d] = 0.f; r.var[d] = 1.f; }
    o.count = obs_count; r.mean[0] = ret_mean; r.var[0] = ret_var; r.count = ret_count;
    HIPCHK(h, hipMemcpyAsync(h->obs_rms + h->obs_par, &o, sizeof(o), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->ret_rms + h->ret_par, &r, sizeof(r), hipMemcpyHostToDevice, h->stream));
    return sync(h);
}

DRIL_EXPORT int32_t dril_norm_get_original(dril_handle* h, float* obs, float* rewards) {
    NEED(h); NOT_EXTERNAL(h, "dril_norm_get_original");
    if (!normalizing(h)) return fail(h, DRIL_ERR_NOT_INITIALISED, "NormalizeWrapperEnv is off (cfg.norm_obs == cfg.norm_reward == 0)");
    if (obs) HIPCHK(h, hipMemcpyAsync(obs, h->e_obs_raw, (size_t)h->cfg.n_envs * h->D * 4, hipMemcpyDeviceToHost, h->stream));
    if (rewards) HIPCHK(h, hipMemcpyAsync(rewards, h->e_rew, (size_t)h->cfg.n_envs * 4, hipMemcpyDeviceToHost, h->stream));
    return sync(h);
}

DRIL_EXPORT int32_t dril_monitor_get_stats(dril_handle* h, float* ep_rew_mean, float* ep_len_mean, int32_t* n_episodes) {
    NEED(h);
    if (!h->mon_cur_ret) return fail(h, DRIL_ERR_NOT_INITIALISED, "MonitorWrapperEnv is off (cfg.monitor_window == 0)");
    const int W = h->cfg.monitor_window;
    std::vector<float> r(W); std::vector<int32_t> l(W); int meta[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(r.data(), h->mon_ring_ret, (size_t)W * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(l.data(), h->mon_ring_len, (size_t)W * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(meta, h->mon_meta, 8, hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    double sr = 0, sl = 0;
    for (int i = 0; i < meta[0]; ++i) { sr += r[i]; sl += l[i]; }
    if (n_episodes) *n_episodes = meta[0];
    if (ep_rew_mean) *ep_rew_mean = meta[0] ? (float)(sr / meta[0]) : 0.f;      // log_stats: mean over the CircularBuffer, monitorWrapperEnv.jl:64-70
    if (ep_len_mean) *ep_len_mean = meta[0] ? (float)(sl / meta[0]) : 0.f;
    return DRIL_OK;
}

// ---- policy on host batches ----------------------------------------------------------------------
namespace {
int policy_host(dril_handle* h, const float* obs, int64_t B, const void* noise, void* actions, bool actions_in, float* values, float* logp,
                float* entropy, int mode, int deterministic = 0) {
    if (!obs || B < 1) return fail(h, DRIL_ERR_INVALID_ARG, "policy: null obs or batch < 1");
    float *d_obs = nullptr, *d_val = nullptr, *d_lp = nullptr, *d_ent = nullptr; void *d_noise = nullptr, *d_act = nullptr;
    const size_t ab = (size_t)B * act_bytes_per(h), nb = (size_t)B * (h->discrete ? 8 : 4 * (size_t)h->A);
    int rc = DRIL_OK;
    auto cleanup = [&]() { hipFree(d_obs); hipFree(d_val); hipFree(d_lp); hipFree(d_ent); hipFree(d_noise); hipFree(d_act); };
#define PCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(h, DRIL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    PCHK(dmalloc(&d_obs, (size_t)B * h->D)); PCHK(dmalloc(&d_val, (size_t)B)); PCHK(dmalloc(&d_lp, (size_t)B)); PCHK(dmalloc(&d_ent, (size_t)B));
    PCHK(hipMalloc(&d_act, ab));
    PCHK(hipMemcpyAsync(d_obs, obs, (size_t)B * h->D * 4, hipMemcpyHostToDevice, h->stream));
    if (noise) { PCHK(hipMalloc(&d_noise, nb)); PCHK(hipMemcpyAsync(d_noise, noise, nb, hipMemcpyHostToDevice, h->stream)); }
    if (actions_in) PCHK(hipMemcpyAsync(d_act, actions, ab, hipMemcpyHostToDevice, h->stream));
    { int rcw = ensure_wimg(h); if (rcw) { cleanup(); return rcw; } }
    PolicyArgs a = policy_args(h, d_obs, B, d_noise, d_act, d_val, d_lp, d_ent, mode);
    a.deterministic = deterministic;
    PCHK(run_policy(h, a));
    h->policy_calls += 1;
    if (values) PCHK(hipMemcpyAsync(values, d_val, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (logp && mode != 2) PCHK(hipMemcpyAsync(logp, d_lp, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (entropy && mode == 1) PCHK(hipMemcpyAsync(entropy, d_ent, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (actions && mode == 0) PCHK(hipMemcpyAsync(actions, d_act, ab, hipMemcpyDeviceToHost, h->stream));
    rc = sync(h);
#undef PCHK
    cleanup();
    return rc;
}
}  // namespace
DRIL_EXPORT int32_t dril_policy_forward(dril_handle* h, const float* obs, int64_t batch, const void* noise, void* actions, float* values, float* logprobs) {
    NEED(h); return policy_host(h, obs, batch, noise, actions, false, values, logprobs, nullptr, 0);
}
DRIL_EXPORT int32_t dril_evaluate_actions(dril_handle* h, const float* obs, const void* actions, int64_t batch, float* values, float* logprobs, float* entropy) {
    NEED(h); if (!actions) return fail(h, DRIL_ERR_INVALID_ARG, "null actions");
    return policy_host(h, obs, batch, nullptr, const_cast<void*>(actions), true, values, logprobs, entropy, 1);
}
DRIL_EXPORT int32_t dril_predict_actions(dril_handle* h, const float* obs, int64_t batch, int32_t deterministic, const void* noise, void* actions) {
    NEED(h); if (!actions) return fail(h, DRIL_ERR_INVALID_ARG, "null actions");
    return policy_host(h, obs, batch, deterministic ? nullptr : noise, actions, false, nullptr, nullptr, nullptr, 0, deterministic ? 1 : 0);
}
DRIL_EXPORT int32_t dril_predict_values(dril_handle* h, const float* obs, int64_t batch, float* values) {
    NEED(h); return policy_host(h, obs, batch, nullptr, nullptr, false, values, nullptr, nullptr, 2);
}

// ---- rollout ---------------------------------------------------------------------------------------
namespace {
int compute_gae(dril_handle* h) {
    if (++h->gae_tag == 0) h->gae_tag = 1;                                            // 0 = "never written" (the carry words are zero at allocation)
    prof_begin(h, DRIL_K_GAE);
    HIPCHK(h, launch_gae(h->cfg.n_envs, h->cfg.n_steps, h->cfg.gamma, h->cfg.gae_lambda, h->rew, h->val, h->flags, h->boot, h->last_values,
                         h->adv, h->ret, h->gae_carry, h->gae_tag, h->gae_err, h->stream));
    prof_end(h);
    return DRIL_OK;
}
// step-granular collect_trajectories (trajectory.jl:22-78) for wrapped envs: three launches per env step on the handle's stream
//   policy_kernel (+ V(terminal_observation) of the previous step) -> norm_step_kernel -> norm_apply_kernel
int collect_rollout_stepwise(dril_handle* h) {
    const int E = h->cfg.n_envs, T = h->cfg.n_steps, D = h->D, A = h->A;
    const size_t ab = act_bytes_per(h);
    int nb = (E + 255) / 256; if (nb > h->rms_blocks) nb = h->rms_blocks;
    int rc = observe_dev(h, true); if (rc) return rc;                                      // new_obs = observe(env), trajectory.jl:32
    for (int t = 0; t < T; ++t) {
        const size_t k = (size_t)t * E;
        const void* nz = h->noise_set ? (const void*)((const char*)h->noise_dev + k * (h->discrete ? 8 : 4 * (size_t)A)) : nullptr;
        PolicyArgs p = policy_args(h, h->e_obs, E, nz, (char*)h->act + k * ab, h->val + k, h->logp + k, nullptr, 0);
        p.gstep = h->gstep; p.env_seed0 = h->env_seed0; p.obs_out = h->obs + k * D;
        if (t > 0) { p.boot_obs = h->e_tobs; p.boot_where = h->e_trunc; p.boot_out = h->boot + (k - E); }   // :57-61 for step t-1
        HIPCHK(h, run_policy(h, p));   // get_action_and_values, :41
        NormStepArgs s{};
        s.E = E; s.episode_len = h->cfg.episode_len; s.fixed_len = h->cfg.fixed_length_episodes; s.action_start = h->cfg.action_start;
        s.seed0 = h->env_seed0; s.gamma = h->cfg.norm_gamma; s.update_ret = (h->cfg.norm_reward && h->cfg.norm_training) ? 1 : 0;
        s.actions = (const char*)h->act + k * ab; s.state = h->state; s.step_count = h->step_count; s.episode = h->episode; s.gstep = h->gstep;
        s.disc_returns = h->disc_returns; s.rew_raw = h->e_rew; s.term = h->e_term; s.trunc = h->e_trunc; s.flags_out = h->flags + k;
        s.tobs_raw = h->e_tobs; s.obs_raw = h->e_obs_raw; s.partials = h->rms_partials;
        if (h->mon_cur_ret) { s.mon_cur_ret = h->mon_cur_ret; s.mon_cur_len = h->mon_cur_len; s.ep_ret = h->ep_ret + k; s.ep_len = h->ep_len + k; }
        HIPCHK(h, launch_norm_step(h->cfg.env_kind, s, nb, h->stream));                              // to_env + act!, :43-44
        NormApplyArgs ap{};
        ap.E = E; ap.D = D; ap.update_obs = (h->cfg.norm_obs && h->cfg.norm_training) ? 1 : 0; ap.update_ret = s.update_ret;
        ap.norm_obs = h->cfg.norm_obs; ap.norm_reward = h->cfg.norm_reward;
        int nba = nb;
        { int rcg = global_partials(h, ap.update_obs || ap.update_ret, ap.partials, nba, ap.n_stats); if (rcg) return rcg; }
        ap.nblocks = nba;
        ap.obs_in = h->obs_rms + h->obs_par; ap.obs_out = h->obs_rms + (h->obs_par ^ 1); ap.ret_in = h->ret_rms + h->ret_par; ap.ret_out = h->ret_rms + (h->ret_par ^ 1);
        ap.rew_raw = h->e_rew; ap.rew_out = h->rew + k; ap.disc_returns = h->disc_returns; ap.term = h->e_term; ap.trunc = h->e_trunc; ap.tobs = h->e_tobs;
        ap.obs_raw = h->e_obs_raw; ap.obs_n = h->e_obs; ap.clip_obs = h->cfg.clip_obs; ap.clip_reward = h->cfg.clip_reward; ap.eps = h->cfg.norm_epsilon;
        HIPCHK(h, launch_norm_apply(ap, h->stream));                                                 // new_obs = observe(env), :45
        h->obs_par ^= 1; h->ret_par ^= 1;
    }
    PolicyArgs l = policy_args(h, h->e_obs, E, nullptr, nullptr, h->last_values, nullptr, nullptr, 2);
    l.boot_obs = h->e_tobs; l.boot_where = h->e_trunc; l.boot_out = h->boot + (size_t)(T - 1) * E;
    HIPCHK(h, run_policy(h, l));       // V(new_obs) for rollout-limited tails, :65-70
    return monitor_collect_rollout(h);
}

int collect_rollout(dril_handle* h, double* fps, bool do_sync) {
    if (!h->env_ready) return fail(h, DRIL_ERR_NOT_INITIALISED, "dril_collect_rollout before dril_env_reset");
    { int rcw = ensure_wimg(h); if (rcw) return rcw; }
    if (normalizing(h) || h->force_stepwise || h->generic) {            // generic nets have no fused rollout kernel: step-granular launches
        const auto t0s = std::chrono::steady_clock::now();
        if (fps) HIPCHK(h, hipStreamSynchronize(h->stream));
        prof_begin(h, DRIL_K_ROLLOUT);
        int rcs = collect_rollout_stepwise(h);
        prof_end(h);
        if (rcs) return rcs;
        if (fps) { HIPCHK(h, hipStreamSynchronize(h->stream)); const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0s).count(); *fps = (double)h->N / (dt > 0 ? dt : 1e-12); }
        h->noise_set = false;
        rcs = compute_gae(h); if (rcs) return rcs;
        return do_sync ? sync(h) : DRIL_OK;
    }
    RolloutArgs a{};
    a.params = h->params; a.state = h->state; a.step_count = h->step_count; a.episode = h->episode; a.gstep = h->gstep;
    a.obs = h->obs; a.act = h->act; a.rew = h->rew; a.logp = h->logp; a.val = h->val; a.boot = h->boot; a.flags = h->flags; a.last_values = h->last_values;
    a.noise = h->noise_set ? h->noise_dev : nullptr;
    a.E = h->cfg.n_envs; a.T = h->cfg.n_steps; a.episode_len = h->cfg.episode_len; a.fixed_len = h->cfg.fixed_length_episodes;
    a.action_start = h->cfg.action_start; a.log_std_off = h->log_std_off; a.env_seed0 = h->env_seed0; a.actor = h->actor; a.critic = h->critic;
    a.exact_f32 = fwd_exact(h) ? 1 : 0;
    a.w2a_actor = a.exact_f32 ? h->w2a_actor : (const float*)h->w2pf_actor; a.w2a_critic = a.exact_f32 ? h->w2a_critic : (const float*)h->w2pf_critic;
    a.mon_cur_ret = h->mon_cur_ret; a.mon_cur_len = h->mon_cur_len; a.ep_ret = h->ep_ret; a.ep_len = h->ep_len;
    const auto t0 = std::chrono::steady_clock::now();
    if (fps) HIPCHK(h, hipStreamSynchronize(h->stream));
    prof_begin(h, DRIL_K_ROLLOUT);
    HIPCHK(h, launch_rollout(h->cfg.env_kind, h->cfg.hidden1, a, h->stream));
    prof_end(h);
    if (fps) {   // fps = steps / wall time of collect_trajectories, rollout_buffer.jl:60-64
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        *fps = (double)h->N / (dt > 0 ? dt : 1e-12);
    }
    h->noise_set = false;
    { int rcm = monitor_collect_rollout(h); if (rcm) return rcm; }
    int rc = compute_gae(h); if (rc) return rc;
    return do_sync ? sync(h) : DRIL_OK;
}
}  // namespace
// ---- collect_trajectories over HOST envs (DRIL_ENV_EXTERNAL), trajectory.jl:22-78: the caller steps its envs, the device does the rest ----
DRIL_EXPORT int32_t dril_ext_act(dril_handle* h, const float* obs, void* raw_actions, void* env_actions) {
    NEED(h);
    if (!h->external) return fail(h, DRIL_ERR_UNSUPPORTED, "dril_ext_act: the handle was not created with DRIL_ENV_EXTERNAL");
    if (!obs) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_act: null obs");
    if (h->ext_acted) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_act: the previous step has no dril_ext_record yet");
    if (h->ext_t >= h->cfg.n_steps) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_act: n_steps env steps are recorded; call dril_ext_finish");
    const size_t E = h->cfg.n_envs, D = h->D, A = h->A, ab = act_bytes_per(h), k = (size_t)h->ext_t * E;
    HIPCHK(h, hipMemcpyAsync(h->obs + k * D, obs, E * D * 4, hipMemcpyHostToDevice, h->stream));                          // observation -> buffer, :46-47
    const void* nz = h->noise_set ? (const void*)((const char*)h->noise_dev + k * (h->discrete ? 8 : 4 * A)) : nullptr;
    PolicyArgs p = policy_args(h, h->obs + k * D, (int64_t)E, nz, (char*)h->act + k * ab, h->val + k, h->logp + k, nullptr, 0);     // get_action_and_values :41; raw action stored :48
    HIPCHK(h, run_policy(h, p));
    h->policy_calls += 1;
    void* first = raw_actions ? raw_actions : env_actions;                             // one device-to-host copy; the second output is a host copy of it
    if (first) HIPCHK(h, hipMemcpyAsync(first, (char*)h->act + k * ab, E * ab, hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    if (raw_actions && env_actions) std::memcpy(env_actions, raw_actions, E * ab);
    if (env_actions && !h->discrete && h->cfg.ext_action_low < h->cfg.ext_action_high) {                                  // to_env(ClampAdapter) :42, default_adapters.jl:4-11; E * A floats, on the host
        float* a = (float*)env_actions; const float lo = h->cfg.ext_action_low, hi = h->cfg.ext_action_high;
        for (size_t i = 0; i < E * A; ++i) a[i] = a[i] < lo ? lo : (a[i] > hi ? hi : a[i]);
    }
    h->ext_acted = true;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_ext_record(dril_handle* h, const float* rewards, const uint8_t* terminated, const uint8_t* truncated, const float* terminal_obs) {
    NEED(h);
    if (!h->external) return fail(h, DRIL_ERR_UNSUPPORTED, "dril_ext_record: the handle was not created with DRIL_ENV_EXTERNAL");
    if (!rewards || !terminated || !truncated) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_record: null rewards / terminated / truncated");
    if (!h->ext_acted) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_record without a preceding dril_ext_act");
    const size_t E = h->cfg.n_envs, D = h->D, k = (size_t)h->ext_t * E;
    float* srew = h->ext_stage_rew + k; uint8_t* fl = h->ext_stage_flags + k;     // this step's slot of the pinned staging area: not reused before the next rollout
    std::vector<int> tr;
    for (size_t e = 0; e < E; ++e) { fl[e] = (uint8_t)((terminated[e] ? 1 : 0) | (truncated[e] ? 2 : 0)); if (truncated[e]) tr.push_back((int)e); }
    if (!tr.empty() && !terminal_obs) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_record: truncated envs need terminal_obs (infos[i][\"terminal_observation\"], multithreadedParallelEnv.jl:64-66)");
    std::memcpy(srew, rewards, E * 4);
    HIPCHK(h, hipMemcpyAsync(h->rew + k, srew, E * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->flags + k, fl, E, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->boot + k, 0, E * 4, h->stream));
    if (!tr.empty()) {                                                                // V(terminal_observation) of the truncated envs only, trajectory.jl:57-61
        const size_t n = tr.size();
        std::vector<float> tobs(n * D), bv(n), row(E, 0.f);
        for (size_t j = 0; j < n; ++j) std::memcpy(&tobs[j * D], terminal_obs + (size_t)tr[j] * D, D * 4);
        HIPCHK(h, hipMemcpyAsync(h->e_tobs, tobs.data(), n * D * 4, hipMemcpyHostToDevice, h->stream));
        PolicyArgs p = policy_args(h, h->e_tobs, (int64_t)n, nullptr, nullptr, h->e_rew, nullptr, nullptr, 2);
        HIPCHK(h, run_policy(h, p));
        HIPCHK(h, hipMemcpyAsync(bv.data(), h->e_rew, n * 4, hipMemcpyDeviceToHost, h->stream));
        int rc = sync(h); if (rc) return rc;                                          // host temporaries: drain before they go out of scope
        for (size_t j = 0; j < n; ++j) row[tr[j]] = bv[j];
        HIPCHK(h, hipMemcpyAsync(h->boot + k, row.data(), E * 4, hipMemcpyHostToDevice, h->stream));
        rc = sync(h); if (rc) return rc;
    }
    // no truncation: nothing to wait for — the copies read the pinned slot and complete behind the next dril_ext_act
    h->ext_t += 1; h->ext_acted = false;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_ext_finish(dril_handle* h, const float* last_obs) {
    NEED(h);
    if (!h->external) return fail(h, DRIL_ERR_UNSUPPORTED, "dril_ext_finish: the handle was not created with DRIL_ENV_EXTERNAL");
    if (!last_obs) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_finish: null last_obs");
    if (h->ext_acted || h->ext_t != h->cfg.n_steps) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ext_finish: the rollout needs exactly n_steps act/record pairs");
    const size_t E = h->cfg.n_envs, D = h->D;
    HIPCHK(h, hipMemcpyAsync(h->e_obs, last_obs, E * D * 4, hipMemcpyHostToDevice, h->stream));
    PolicyArgs p = policy_args(h, h->e_obs, (int64_t)E, nullptr, nullptr, h->last_values, nullptr, nullptr, 2);          // V(new_obs) where the last step left the trajectory open, :65-70
    HIPCHK(h, run_policy(h, p));
    h->ext_t = 0; h->noise_set = false;
    int rc = compute_gae(h); if (rc) return rc;                                       // compute_advantages! + returns, rollout_buffer.jl:83-87
    return sync(h);
}
DRIL_EXPORT int32_t dril_ext_steps(const dril_handle* h) { return h ? h->ext_t : -1; }

DRIL_EXPORT int32_t dril_collect_rollout(dril_handle* h, double* fps) { NEED(h); NOT_EXTERNAL(h, "dril_collect_rollout"); return collect_rollout(h, fps, true); }
DRIL_EXPORT int32_t dril_debug_set_noise(dril_handle* h, const void* noise, size_t count) {
    NEED(h);
    if (!noise) { h->noise_set = false; return DRIL_OK; }
    const size_t need = (size_t)h->N * (h->discrete ? 1 : (size_t)h->A);
    if (count != need) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_set_noise: count must be n_envs*n_steps (*action_dim)");
    const size_t bytes = need * (h->discrete ? 8 : 4);
    if (!h->noise_dev) HIPCHK(h, hipMalloc(&h->noise_dev, bytes));
    HIPCHK(h, hipMemcpyAsync(h->noise_dev, noise, bytes, hipMemcpyHostToDevice, h->stream));
    h->noise_set = true;
    return sync(h);
}
DRIL_EXPORT int32_t dril_buffer_copy_out(dril_handle* h, int32_t which, void* host, size_t bytes) {
    NEED(h); size_t b; void* p = buf_ptr(h, which, &b);
    if (!p || !host || b != bytes) return fail(h, DRIL_ERR_INVALID_ARG, "dril_buffer_copy_out: bad id or byte count");
    HIPCHK(h, hipMemcpyAsync(host, p, b, hipMemcpyDeviceToHost, h->stream)); return sync(h);
}
DRIL_EXPORT int32_t dril_buffer_copy_in(dril_handle* h, int32_t which, const void* host, size_t bytes) {
    NEED(h); size_t b; void* p = buf_ptr(h, which, &b);
    if (!p || !host || b != bytes) return fail(h, DRIL_ERR_INVALID_ARG, "dril_buffer_copy_in: bad id or byte count");
    HIPCHK(h, hipMemcpyAsync(p, host, b, hipMemcpyHostToDevice, h->stream)); return sync(h);
}
DRIL_EXPORT int32_t dril_compute_gae(dril_handle* h) {
    NEED(h); int rc = compute_gae(h); if (rc) return rc;
    int gave_up = 0; HIPCHK(h, hipMemcpyAsync(&gave_up, h->gae_err, 4, hipMemcpyDeviceToHost, h->stream));
    rc = sync(h); if (rc) return rc;
    if (gave_up) { (void)hipMemsetAsync(h->gae_err, 0, 4, h->stream); return fail(h, DRIL_ERR_HIP, "gae_scan_kernel: a chunk's predecessor did not publish its carry within the spin limit"); }
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_gae(int32_t E, int32_t T, float gamma, float lam, const float* rewards, const float* values, const uint8_t* flags,
                             const float* bootstrap, const float* last_values, float* advantages, float* returns) {
    if (E < 1 || T < 1 || !rewards || !values || !flags || !bootstrap || !last_values || !advantages || !returns)
        return fail(nullptr, DRIL_ERR_INVALID_ARG, "dril_gae: bad argument");
    const size_t N = (size_t)E * T;
    float *d_r = nullptr, *d_v = nullptr, *d_b = nullptr, *d_l = nullptr, *d_a = nullptr, *d_ret = nullptr; uint8_t* d_f = nullptr; unsigned long long* d_c = nullptr; int* d_e = nullptr;
    auto cleanup = [&]() { hipFree(d_r); hipFree(d_v); hipFree(d_b); hipFree(d_l); hipFree(d_a); hipFree(d_ret); hipFree(d_f); hipFree(d_c); hipFree(d_e); };
#define GCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(nullptr, DRIL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    GCHK(dmalloc(&d_r, N)); GCHK(dmalloc(&d_v, N)); GCHK(dmalloc(&d_b, N)); GCHK(dmalloc(&d_l, (size_t)E)); GCHK(dmalloc(&d_a, N)); GCHK(dmalloc(&d_ret, N)); GCHK(dmalloc(&d_f, N));
    GCHK(hipMemcpy(d_r, rewards, N * 4, hipMemcpyHostToDevice)); GCHK(hipMemcpy(d_v, values, N * 4, hipMemcpyHostToDevice));
    GCHK(hipMemcpy(d_b, bootstrap, N * 4, hipMemcpyHostToDevice)); GCHK(hipMemcpy(d_l, last_values, (size_t)E * 4, hipMemcpyHostToDevice));
    GCHK(hipMemcpy(d_f, flags, N, hipMemcpyHostToDevice));
    const size_t words = (size_t)gae_chunks(T) * E; int gave_up = 0;
    GCHK(dmalloc(&d_c, words)); GCHK(hipMemset(d_c, 0, words * 8)); GCHK(dmalloc(&d_e, 1)); GCHK(hipMemset(d_e, 0, 4));
    GCHK(launch_gae(E, T, gamma, lam, d_r, d_v, d_f, d_b, d_l, d_a, d_ret, d_c, 1u, d_e, nullptr));
    GCHK(hipDeviceSynchronize());
    GCHK(hipMemcpy(&gave_up, d_e, 4, hipMemcpyDeviceToHost));
    if (gave_up) { cleanup(); return fail(nullptr, DRIL_ERR_HIP, "gae_scan_kernel: a chunk's predecessor did not publish its carry within the spin limit"); }
    GCHK(hipMemcpy(advantages, d_a, N * 4, hipMemcpyDeviceToHost)); GCHK(hipMemcpy(returns, d_ret, N * 4, hipMemcpyDeviceToHost));
#undef GCHK
    cleanup();
    return DRIL_OK;
}

// ---- PPO update ------------------------------------------------------------------------------------
DRIL_EXPORT int32_t dril_debug_set_permutation(dril_handle* h, const int64_t* perm, size_t count) {
    NEED(h);
    if (!perm) { h->perm_count = 0; return DRIL_OK; }
    const size_t need = (size_t)h->cfg.epochs * (size_t)h->N;
    if (count != need) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_set_permutation: count must be epochs * n_envs * n_steps");
    for (size_t i = 0; i < count; ++i) if (perm[i] < 0 || perm[i] >= h->N) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_set_permutation: index out of range");
    if (!h->perm_dev) HIPCHK(h, dmalloc(&h->perm_dev, need));
    HIPCHK(h, hipMemcpyAsync(h->perm_dev, perm, need * 8, hipMemcpyHostToDevice, h->stream));
    h->perm_count = need;
    return sync(h);
}

namespace {
int ppo_update_once(dril_handle* h, dril_ppo_stats* out) {
    const int world = comm_ready(h) ? h->cfg.world_size : 1;
    if (h->cfg.world_size > 1 && !comm_ready(h)) return fail(h, DRIL_ERR_NOT_INITIALISED, "world_size > 1 but dril_comm_init was not called");
    const int64_t N = h->N, B = h->cfg.batch_size / h->cfg.world_size;
    const int64_t nb = (N + B - 1) / B;                       // partial last batch kept (MLUtils partial=true)
    const int64_t total_steps = nb * h->cfg.epochs;
    const uint64_t adam_steps0 = h->adam_steps;
    h->used_f16 = false; h->spin_timeout = false;
    if (total_steps > h->step_stats_cap) {
        if (h->step_stats) hipFree(h->step_stats);
        h->step_stats = nullptr; HIPCHK(h, dmalloc(&h->step_stats, (size_t)total_steps * 16)); h->step_stats_cap = (int)total_steps;
    }
    if (total_steps > 0) HIPCHK(h, hipMemsetAsync(h->step_stats, 0, (size_t)total_steps * 16 * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->stop_flag, 0, 4, h->stream)); HIPCHK(h, hipMemsetAsync(h->nan_flag, 0, 4, h->stream));
    const int bits = perm_bits(N);
    if (h->rec) { prof_begin(h, DRIL_K_PACK_RECORDS); HIPCHK(h, launch_pack_records(h->cfg.env_kind, N, h->obs, h->act, h->adv, h->logp, h->ret, h->rec, h->stream)); prof_end(h); }
    int64_t step = 0;
    // the reference's default PPO() (batch_size = 64) on hidden [64,64]: every optimiser step of the iteration inside ONE launch of two persistent workgroups, one per net (dril_update_small.hip);
    // single-rank only (a data-parallel run all-reduces between the gradient and the step)
    const bool persistent = !h->wide && !h->generic && h->cfg.hidden1 == 64 && h->D <= 8 && world == 1 && !(comm_ready(h) && h->force_allreduce) && h->rec && B >= 2 && B <= 64 && h->P <= 2 * 256 * 19 &&
                            !h->no_persistent && !h->no_small_path && h->grad_variant < 0 && total_steps > 0;   // (DRIL_GRAD_VARIANT pins one of the per-step kernels)
    if (persistent) {
        if (h->cfg.epochs > h->epoch_keys_cap) {
            if (h->epoch_keys) hipFree(h->epoch_keys);
            h->epoch_keys = nullptr; HIPCHK(h, dmalloc(&h->epoch_keys, (size_t)h->cfg.epochs)); h->epoch_keys_cap = h->cfg.epochs;
        }
        std::vector<uint64_t> keys((size_t)h->cfg.epochs);
        for (int ep = 0; ep < h->cfg.epochs; ++ep) keys[ep] = perm_key(h->cfg.seed + (uint64_t)h->cfg.rank, h->update_counter, ep);
        HIPCHK(h, hipMemcpyAsync(h->epoch_keys, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));                                    // (keys is a stack-lifetime host buffer)
        SmallUpdateArgs u{};
        u.params = h->params; u.adam_m = h->adam_m; u.adam_v = h->adam_v; u.bt = h->bt; u.step_parity = (int)(h->adam_steps & 1);
        u.rec = h->rec; u.val_old = h->val; u.perm = h->perm_count ? h->perm_dev : nullptr; u.keys = h->epoch_keys; u.perm_bits = bits;
        u.N = N; u.B = B; u.nb = (int)nb; u.step_stats = h->step_stats; u.norm_out = h->norm_out; u.nan_flag = h->nan_flag; u.stop_flag = h->stop_flag;
        u.lr = h->lr; u.beta1 = h->cfg.adam_beta1; u.beta2 = h->cfg.adam_beta2; u.eps = h->cfg.adam_eps; u.max_grad_norm = h->cfg.max_grad_norm;
        u.target_kl = h->cfg.target_kl; u.ent_coef = h->cfg.ent_coef; u.vf_coef = h->cfg.vf_coef; u.clip_range = h->cfg.clip_range; u.clip_range_vf = h->cfg.clip_range_vf;
        u.has_max_grad_norm = h->cfg.has_max_grad_norm; u.has_target_kl = h->cfg.has_target_kl; u.has_clip_vf = h->cfg.has_clip_range_vf;
        u.normalize_adv = h->cfg.normalize_advantage; u.action_start = h->cfg.action_start; u.P = h->P; u.Pa = h->Pa; u.Pc = h->Pc; u.dbg = h->dbg;
        if (!h->small_xchg) HIPCHK(h, dmalloc(&h->small_xchg, (size_t)kSmallXchgWords));
        u.xchg = h->small_xchg; u.debug_solo = std::getenv("DRIL_SMALL_DEBUG_SOLO") != nullptr;
        const int64_t chunk = h->small_chunk;                                          // optimiser steps per launch (16 384: a bound on one kernel's run time, ~0.2 s)
        for (int64_t s0 = 0; s0 < total_steps; s0 += chunk) {
            u.step0 = (int)s0; u.nsteps = (int)(total_steps - s0 < chunk ? total_steps - s0 : chunk);
            u.step_parity = s0 == 0 ? (int)(h->adam_steps & 1) : 0;                  // (the kernel leaves both ping-pong slots of the beta powers equal)
            prof_begin(h, DRIL_K_PPO_GRAD);
            HIPCHK(h, launch_ppo_update_small(h->cfg.env_kind, u, h->stream));
            prof_end(h);
        }
        h->adam_steps += total_steps; h->wimg_dirty = true; h->last_variant = 6; h->used_f16 = true;
#ifdef DRIL_STAMPS
        {   // per-phase s_memtime ticks of the last launch, per wave
            hipStreamSynchronize(h->stream);
            std::vector<unsigned long long> d(8 * 16);
            hipMemcpy(d.data(), h->dbg, d.size() * 8, hipMemcpyDeviceToHost);
            const char* nm[16] = {"top: moments + barrier", "L1 + tanh + h1 pieces", "wait B1", "L2 + tanh + out partial", "wait B2", "head + dW3 + dz2 + pieces", "wait B3", "dh1 + dW1 + dW2", "wait B4",
                                  "dW2 overlay", "wait B5", "g + norm", "stats + Adam + images", "-", "-", "-"};
            for (int wv = 0; wv < 8; wv += 4) {
                double tot = 0; for (int k = 0; k < 15; ++k) tot += (double)d[wv * 16 + k];
                fprintf(stderr, "[small stamps] wave %d: %.0f ticks per step (100 MHz: %.2f us)\n", wv, tot / (double)total_steps, tot / (double)total_steps / 100.0);
                for (int k = 0; k < 15; ++k) fprintf(stderr, "   %-28s %8.0f ticks/step  %5.1f %%\n", nm[k], (double)d[wv * 16 + k] / (double)total_steps, 100.0 * (double)d[wv * 16 + k] / tot);
            }
        }
#endif
        step = total_steps;
    }
    for (int ep = 0; ep < h->cfg.epochs && !persistent; ++ep) {
        const uint64_t key = perm_key(h->cfg.seed + (uint64_t)h->cfg.rank, h->update_counter, ep);
        const int64_t* perm = h->perm_count ? h->perm_dev + (size_t)ep * N : nullptr;
        const bool epoch_moments = h->cfg.normalize_advantage && !perm && nb >= 2 && nb <= 2048;
        // chip-filling minibatches of the fused kernels: the epoch's order as an index array (the update kernels then read 8 bytes per sample instead of evaluating the
        // keyed bijection per lane, wave, net and tile)
        const bool index_array = !perm && !h->generic && !h->no_epoch_index && N < (1ll << 31) && (B + kTile - 1) / kTile >= kPairTilesPerCu * (int64_t)h->num_cus;
        if (index_array) {
            if (!h->epoch_index) HIPCHK(h, dmalloc(&h->epoch_index, (size_t)N));
            prof_begin(h, DRIL_K_ADV_MOMENTS);
            HIPCHK(h, launch_epoch_index(N, key, bits, h->epoch_index, h->stream));
            prof_end(h);
        }
        if (epoch_moments) {
            if (nb > h->epoch_nb_cap) {
                if (h->epoch_tables) hipFree(h->epoch_tables); if (h->epoch_stats) hipFree(h->epoch_stats);
                h->epoch_tables = nullptr; h->epoch_stats = nullptr;
                HIPCHK(h, dmalloc(&h->epoch_tables, (size_t)h->epoch_blocks * 2 * nb)); HIPCHK(h, dmalloc(&h->epoch_stats, (size_t)3 * nb));
                h->epoch_nb_cap = (int)nb;
            }
            prof_begin(h, DRIL_K_ADV_MOMENTS);
            const int eb = (int)std::min<int64_t>(h->epoch_blocks, std::max<int64_t>(64, N / 8192));       // eight waves per SIMD at chip-filling sizes: the pass is one dependent chain per sample
            HIPCHK(h, launch_epoch_moments(h->adv, N, B, (int)nb, key, bits, h->epoch_tables, eb, h->epoch_stats, h->stop_flag, h->stream));
            prof_end(h);
            if (world > 1 || (comm_ready(h) && h->force_allreduce)) {     // ONE all-reduce per epoch for the advantage moments of all its minibatches
                int rca = rccl_allreduce(h, h->epoch_stats, (size_t)3 * nb, kNcclFloat64); if (rca) return rca;
            }
        }
        for (int64_t k = 0; k < nb; ++k, ++step) {
            const int64_t pos0 = k * B, count = (pos0 + B <= N) ? B : N - pos0;
            int rc = ppo_step(h, h->obs, h->act, h->adv, h->ret, h->logp, h->val, perm, pos0, count, N, key, bits, h->step_stats + step * 16, true, h->rec,
                              epoch_moments ? h->epoch_stats + 3 * k : nullptr, index_array ? h->epoch_index : nullptr);
            if (rc) return rc;
        }
    }
    h->update_counter += 1;
    prof_begin(h, DRIL_K_EXPLAINED_VAR);
    HIPCHK(h, launch_explained_var(h->val, h->ret, N, h->ev_partials, h->ev_blocks, h->stream));
    prof_end(h);
    std::vector<float> st((size_t)total_steps * 16); std::vector<double> ev(4 * (size_t)h->ev_blocks); int nan = 0; unsigned w2bits = 0;
    if (!h->generic) {    // max |W2| of the parameters this update leaves behind rides home with its statistics (fwd_exact: the next rollout's arithmetic)
        HIPCHK(h, launch_w2_absmax(h->params, h->actor, h->critic, h->cfg.hidden1, h->w2max_dev, h->stream));
        HIPCHK(h, hipMemcpyAsync(&w2bits, h->w2max_dev, 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (total_steps > 0) HIPCHK(h, hipMemcpyAsync(st.data(), h->step_stats, st.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(ev.data(), h->ev_partials, ev.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&nan, h->nan_flag, 4, hipMemcpyDeviceToHost, h->stream));
    int gae_gave_up = 0; HIPCHK(h, hipMemcpyAsync(&gae_gave_up, h->gae_err, 4, hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    // (a rank-local early return here would leave the other ranks of a data-parallel job alone in the collectives that follow — they saw this rank's NaN advantages in the
    // all-reduced gradient and start the exact-f32 redo: the flag is folded into the explained-variance all-reduce below, and every rank returns the same code at the same point)
    if (gae_gave_up && world == 1) { (void)hipMemsetAsync(h->gae_err, 0, 4, h->stream); return fail(h, DRIL_ERR_HIP, "gae_scan_kernel: a chunk's predecessor did not publish its carry within the spin limit (advantages of this rollout are NaN)"); }
    if (!h->generic) { float m; std::memcpy(&m, &w2bits, 4); h->w2max = (m == m) ? m : INFINITY; }
#ifdef DRIL_STAMPS
    {   // shares of the LAST grad launch, averaged over waves, per head
        std::vector<unsigned long long> d((size_t)2 * h->Gmax * 4 * 12);
        hipMemcpy(d.data(), h->dbg, d.size() * 8, hipMemcpyDeviceToHost);
        const char* names_f32[] = {"gather/loop", "L1+tanh", "L2+tanh", "out+head", "dW3 block", "dz2", "h1img+dh1+dz1", "dW2 block", "dW1 block", "-"};
        const char* names_split[] = {"S1 unpack+L1", "S2 tanh+split+L2+tanh", "S3 L3+head+dW3", "S4 dz2+split+dh1+mask", "S5 dW2", "S6 dz1 img+dW1", "-", "-", "-", "-"};
        // ppo_grad_wide_split_kernel (round 5; per pass of kWideSplitNT x 32 samples)
        const char* names_wide[] = {"L1+tanh+h1 pieces", "wait B1", "L2 chain+tanh", "out partials+wait B2", "head+dW3+dz2+pieces", "wait B3", "dh1 chain", "loader+dz1+dW1", "dW2 chain", "wait B4 (+DMA)"};
        const char** names = std::getenv("DRIL_GRAD_VARIANT") && std::atoi(std::getenv("DRIL_GRAD_VARIANT")) == 0 ? names_f32 : h->wide ? names_wide : names_split;
        for (int head = 0; head < 3; ++head) {
            double acc[10] = {0}; double tiles = 0; int nw = 0;
            for (size_t w = 0; w < d.size() / 12; ++w) if (d[w * 12 + 10] > 0 && (int)d[w * 12 + 11] == head) { for (int k = 0; k < 10; ++k) acc[k] += (double)d[w * 12 + k]; tiles += (double)d[w * 12 + 10]; ++nw; }
            if (!nw) continue;
            double tot = 0; for (int k = 0; k < 10; ++k) tot += acc[k];
            fprintf(stderr, "[stamps] head %d: %d waves, %.0f tiles/wave, %.0f ticks/tile (s_memtime ticks)\n", head, nw, tiles / nw, tot / tiles);
            for (int k = 0; k < 10; ++k) fprintf(stderr, "   %-22s %8.0f ticks/tile  %5.1f %%\n", names[k], acc[k] / tiles, 100.0 * acc[k] / tot);
        }
    }
#endif
    // per-iteration means over the applied steps (ppo.jl:242-264); grad_norm also counts the KL-stopped step (:223)
    double acc[8] = {0}; int n_upd = 0, n_gn = 0, stopped = 0; float ratio_first = 0;
    for (int64_t s = 0; s < total_steps; ++s) {
        const float* o = &st[(size_t)s * 16];
        if (s == 0) ratio_first = o[6];
        if (o[9] == 1.0f) { acc[0] += o[2]; acc[1] += o[0]; acc[2] += o[1]; acc[3] += o[4]; acc[4] += o[3]; acc[5] += o[7]; acc[6] += o[8]; acc[7] += o[5]; ++n_upd; ++n_gn; }
        else if (o[11] == 1.0f) { acc[6] += o[8]; ++n_gn; stopped = 1; break; }
        else if (o[10] == 1.0f) break;
    }
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < h->ev_blocks; ++i) { s0 += ev[4 * i]; s1 += ev[4 * i + 1]; s2 += ev[4 * i + 2]; s3 += ev[4 * i + 3]; }
    double nn = (double)N;
    if (world > 1) {   // explained variance over all shards: tiny host-free all-reduce of the four sums
        double v[6] = {s0, s1, s2, s3, nn, gae_gave_up ? 1.0 : 0.0}; double* d = h->adv_stats;   // reuse the 4-double scratch + norm scratch is too small: use ev_partials
        HIPCHK(h, hipMemcpyAsync(h->ev_partials, v, sizeof(v), hipMemcpyHostToDevice, h->stream)); (void)d;
        rc = rccl_allreduce(h, h->ev_partials, 6, kNcclFloat64); if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(v, h->ev_partials, sizeof(v), hipMemcpyDeviceToHost, h->stream));
        rc = sync(h); if (rc) return rc;
        s0 = v[0]; s1 = v[1]; s2 = v[2]; s3 = v[3]; nn = v[4];
        if (v[5] > 0.0) {                                                  // some rank's GAE scan gave up: every rank reports it, at this same point
            if (gae_gave_up) (void)hipMemsetAsync(h->gae_err, 0, 4, h->stream);
            return fail(h, DRIL_ERR_HIP, gae_gave_up ? "gae_scan_kernel: a chunk's predecessor did not publish its carry within the spin limit (advantages of this rollout are NaN)"
                                                     : "gae_scan_kernel gave up on another rank of the data-parallel job (its advantages are NaN): the update is invalid on every rank");
        }
    }
    const double var_d = (s1 - s0 * s0 / nn) / (nn - 1.0), var_r = (s3 - s2 * s2 / nn) / (nn - 1.0);
    h->adam_steps = adam_steps0 + (uint64_t)n_upd;     // the steps the device APPLIED (a KL stop or a non-finite gradient skips the rest; the stopping step leaves both slots of the beta powers equal, so the parity is free)
    if (out) {
        std::memset(out, 0, sizeof(*out));
        const double den = n_upd > 0 ? n_upd : 1, gden = n_gn > 0 ? n_gn : 1;
        out->entropy_loss = (float)(acc[0] / den); out->policy_loss = (float)(acc[1] / den); out->value_loss = (float)(acc[2] / den);
        out->approx_kl_div = (float)(acc[3] / den); out->clip_fraction = (float)(acc[4] / den); out->loss = (float)(acc[5] / den);
        out->grad_norm = (float)(acc[6] / gden); out->entropy = (float)(acc[7] / den);
        out->explained_variance = (float)(1.0 - var_d / var_r); out->ratio_first = ratio_first;
        out->n_updates = n_upd; out->early_stopped = stopped; out->nan_or_inf = nan;
    }
    if (nan == 2) { h->spin_timeout = true; return fail(h, DRIL_ERR_HIP, "ppo_update_small_kernel: the partner workgroup did not answer within the spin limit (its two workgroups must be resident together); DRIL_NO_PERSISTENT_UPDATE=1 selects the per-step kernels"); }
    if (nan) return fail(h, DRIL_ERR_NAN_IN_GRADS, "gradient contains nan or is not finite (ppo.jl:213-214)");
    return DRIL_OK;
}
// The f16-piece kernels (the default of the fused path) compute fp32-equivalent products inside f16's RANGE: a staged weight beyond ~ 350 or a gradient tile beyond
// 65 504 / SG overflows a hi piece, and the step that meets it reports a non-finite gradient without touching the parameters.  Range is not something the reference
// asks of its user, so such an update is taken back (parameters, Adam moments, beta powers and the counters as they were before it) and redone on the exact-f32
// kernels; only a gradient that is non-finite there as well is an error (ppo.jl:213-214).  Costs three small device copies per update.
int ppo_update(dril_handle* h, dril_ppo_stats* out) {
    const bool may_retry = !h->generic && h->grad_variant < 0 && !h->no_f32_retry;
    // (a) known in advance that f16 cannot hold this update: a W2 entry out of range, or the latch (kRetryLatchAfter consecutive redone updates: the workload lives outside
    //     f16's range, so the next kRetryLatchUpdates run the exact-f32 kernels at once instead of paying an f16 pass + snapshot + redo each) — no wasted pass, no snapshot
    if (may_retry && (h->f32_latch_left > 0 || !(h->w2max < kFwdSplitMaxW))) {
        if (h->f32_latch_left > 0) h->f32_latch_left -= 1;
        const int gv = h->grad_variant; h->grad_variant = 0;
        const int rc = ppo_update_once(h, out);
        h->grad_variant = gv; h->f32_direct_updates += 1;
        if (out) out->f32_path = 2;
        return rc;
    }
    const uint64_t adam_steps0 = h->adam_steps; const uint64_t counter0 = h->update_counter;
    const size_t P = (size_t)h->P;
    if (may_retry) {
        if (!h->retry_snap) HIPCHK(h, dmalloc(&h->retry_snap, 3 * P + 4));
        HIPCHK(h, hipMemcpyAsync(h->retry_snap, h->params, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->retry_snap + P, h->adam_m, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->retry_snap + 2 * P, h->adam_v, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->retry_snap + 3 * P, h->bt, 4 * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    auto restore = [&]() -> int {
        HIPCHK(h, hipMemcpyAsync(h->params, h->retry_snap, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->adam_m, h->retry_snap + P, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->adam_v, h->retry_snap + 2 * P, P * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->bt, h->retry_snap + 3 * P, 4 * 4, hipMemcpyDeviceToDevice, h->stream));
        h->adam_steps = adam_steps0; h->update_counter = counter0; h->wimg_dirty = true;
        return DRIL_OK;
    };
    int rc = ppo_update_once(h, out);
    // (b) the persistent two-workgroup kernel gave up waiting for its partner (the GPU is shared with another stream / process and the two workgroups were not resident
    //     together): nothing is lost — the state of before the update is in the snapshot and the per-step kernels need no co-residency.  Redone once; the handle stays on them.
    if (rc == DRIL_ERR_HIP && h->spin_timeout && may_retry) {
        { const int rr = restore(); if (rr) return rr; }
        h->no_persistent = true; h->persistent_fallbacks += 1;
        rc = ppo_update_once(h, out);
        if (rc == DRIL_OK) h->err.clear();
    }
    // (c) an f16-piece kernel ran somewhere in this update (sticky flag, not the last step's kernel) and a step met a non-finite gradient: redo on the exact-f32 kernels
    if (rc == DRIL_ERR_NAN_IN_GRADS && may_retry && h->used_f16) {
        { const int rr = restore(); if (rr) return rr; }
        const int gv = h->grad_variant; h->grad_variant = 0;                            // the exact-f32 kernels for this update (the persistent small kernel is f16 as well: per-step path)
        rc = ppo_update_once(h, out);
        h->grad_variant = gv; h->f32_retries += 1;
        if (out) out->f32_path = 1;
        if (rc == DRIL_OK) h->err.clear();                                              // the first pass's message is not this call's outcome
        if (rc == DRIL_OK && ++h->f32_streak >= kRetryLatchAfter) h->f32_latch_left = kRetryLatchUpdates;   // (a gradient that is non-finite on exact f32 too is not a range problem: it arms nothing)
    } else if (rc == DRIL_OK && h->used_f16) h->f32_streak = 0;                         // an update inside f16's range: the streak (and with it the latch's re-arming) ends
    return rc;
}
}  // namespace
DRIL_EXPORT int32_t dril_ppo_update(dril_handle* h, dril_ppo_stats* out) { NEED(h); return ppo_update(h, out); }
// how often the f16-piece arithmetic had to be left (include/dril_hip.h): nothing here is an error — every number delivered came from kernels that could hold it
DRIL_EXPORT int64_t dril_f32_retries(const dril_handle* h) { return h ? h->f32_retries : -1; }
DRIL_EXPORT int32_t dril_f32_fallback_info(const dril_handle* h, dril_f32_fallback* out) {
    if (!h || !out) return fail(nullptr, DRIL_ERR_INVALID_ARG, "dril_f32_fallback_info: null argument");
    out->retries = h->f32_retries; out->direct_updates = h->f32_direct_updates; out->persistent_fallbacks = h->persistent_fallbacks;
    out->latch_updates_left = h->f32_latch_left; out->forward_exact_f32 = (!h->generic && fwd_exact(h)) ? 1 : 0; out->max_abs_w2 = h->w2max;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_ppo_loss_grad(dril_handle* h, const float* obs, const void* actions, const float* advantages, const float* returns,
                                       const float* old_logprobs, const float* old_values, int64_t batch, float* loss, float* stats7, float* grads) {
    NEED(h);
    if (!obs || !actions || !advantages || !returns || !old_logprobs || !old_values || batch < 1) return fail(h, DRIL_ERR_INVALID_ARG, "dril_ppo_loss_grad: bad argument");
    if (h->cfg.world_size > 1) return fail(h, DRIL_ERR_UNSUPPORTED, "dril_ppo_loss_grad is a single-rank parity entry point");
    const size_t B = (size_t)batch;
    float *d_obs = nullptr, *d_adv = nullptr, *d_ret = nullptr, *d_lp = nullptr, *d_val = nullptr; void* d_act = nullptr; float4* d_rec = nullptr;
    auto cleanup = [&]() { hipFree(d_obs); hipFree(d_adv); hipFree(d_ret); hipFree(d_lp); hipFree(d_val); hipFree(d_act); hipFree(d_rec); };
#define LCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail(h, DRIL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } } while (0)
    LCHK(dmalloc(&d_obs, B * h->D)); LCHK(dmalloc(&d_adv, B)); LCHK(dmalloc(&d_ret, B)); LCHK(dmalloc(&d_lp, B)); LCHK(dmalloc(&d_val, B)); LCHK(hipMalloc(&d_act, B * act_bytes_per(h)));
    LCHK(hipMemcpyAsync(d_obs, obs, B * h->D * 4, hipMemcpyHostToDevice, h->stream)); LCHK(hipMemcpyAsync(d_adv, advantages, B * 4, hipMemcpyHostToDevice, h->stream));
    LCHK(hipMemcpyAsync(d_ret, returns, B * 4, hipMemcpyHostToDevice, h->stream)); LCHK(hipMemcpyAsync(d_lp, old_logprobs, B * 4, hipMemcpyHostToDevice, h->stream));
    LCHK(hipMemcpyAsync(d_val, old_values, B * 4, hipMemcpyHostToDevice, h->stream)); LCHK(hipMemcpyAsync(d_act, actions, B * act_bytes_per(h), hipMemcpyHostToDevice, h->stream));
    LCHK(hipMemsetAsync(h->stop_flag, 0, 4, h->stream));
    // the same packed records dril_ppo_update builds, so that this parity entry point runs the kernel the size rule picks for `batch` in production
    // (the pair / wide split kernels read records only)
    if (h->rec) { LCHK(dmalloc(&d_rec, (size_t)(h->D <= 4 ? 2 : 3) * B)); LCHK(launch_pack_records(h->cfg.env_kind, batch, d_obs, d_act, d_adv, d_lp, d_ret, d_rec, h->stream)); }
#undef LCHK
    int rc = ppo_step(h, d_obs, d_act, d_adv, d_ret, d_lp, d_val, nullptr, 0, batch, batch, 0, /*bits=0: identity order*/ 0, nullptr, false, d_rec);
    std::vector<float> flat((size_t)h->P + 8);
    if (!rc) { hipError_t e = hipMemcpyAsync(flat.data(), h->flat, flat.size() * 4, hipMemcpyDeviceToHost, h->stream); if (e != hipSuccess) rc = fail(h, DRIL_ERR_HIP, hipGetErrorString(e)); }
    if (!rc) rc = sync(h);
    cleanup();
    if (rc) return rc;
    const float* s = flat.data() + h->P; const float n = s[6];
    const float pl = s[0] / n, ent = s[1] / n, vl = s[5] / n;
    if (grads) std::memcpy(grads, flat.data(), (size_t)h->P * 4);
    if (loss) *loss = pl + h->cfg.ent_coef * (-ent) + h->cfg.vf_coef * vl;
    if (stats7) { stats7[0] = pl; stats7[1] = vl; stats7[2] = -ent; stats7[3] = s[2] / n; stats7[4] = s[3] / n; stats7[5] = ent; stats7[6] = s[4] / n; }
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_apply_gradients(dril_handle* h, const float* grads, size_t n, float* grad_norm) {
    NEED(h);
    if (!grads || n != (size_t)h->P) return fail(h, DRIL_ERR_INVALID_ARG, "dril_apply_gradients: n != dril_param_count");
    HIPCHK(h, hipMemcpyAsync(h->flat, grads, n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->stop_flag, 0, 4, h->stream)); HIPCHK(h, hipMemsetAsync(h->nan_flag, 0, 4, h->stream));
    HIPCHK(h, launch_grad_norm(h->flat, h->P, h->norm_partials, h->stop_flag, h->stream));
    AdamArgs ad{};
    ad.params = h->params; ad.m = h->adam_m; ad.v = h->adam_v; ad.flat = h->flat; ad.P = h->P; ad.norm_partials = h->norm_partials;
    ad.n_partials = h->n_norm_partials; ad.bt = h->bt; ad.step_parity = (int)(h->adam_steps & 1);
    ad.beta1 = h->cfg.adam_beta1; ad.beta2 = h->cfg.adam_beta2; ad.eps = h->cfg.adam_eps; ad.lr = h->lr; ad.max_grad_norm = h->cfg.max_grad_norm;
    ad.has_max_grad_norm = h->cfg.has_max_grad_norm; ad.has_target_kl = 0; ad.use_stats = 0; ad.step_stats = nullptr; ad.norm_out = h->norm_out;
    ad.nan_flag = h->nan_flag; ad.stop_flag = h->stop_flag; ad.stop_flag_w = h->stop_flag;
    HIPCHK(h, launch_adam(ad, h->stream));
    h->adam_steps += 1; h->wimg_dirty = true;
    float norm = 0; int nan = 0; unsigned w2bits = 0;
    if (!h->generic) { HIPCHK(h, launch_w2_absmax(h->params, h->actor, h->critic, h->cfg.hidden1, h->w2max_dev, h->stream)); HIPCHK(h, hipMemcpyAsync(&w2bits, h->w2max_dev, 4, hipMemcpyDeviceToHost, h->stream)); }
    HIPCHK(h, hipMemcpyAsync(&norm, h->norm_out, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&nan, h->nan_flag, 4, hipMemcpyDeviceToHost, h->stream));
    int rc = sync(h); if (rc) return rc;
    if (!h->generic) { float m; std::memcpy(&m, &w2bits, 4); h->w2max = (m == m) ? m : INFINITY; }
    if (grad_norm) *grad_norm = norm;
    if (nan == 2) return fail(h, DRIL_ERR_HIP, "ppo_update_small_kernel: the partner workgroup did not answer within the spin limit (its two workgroups must be resident together); DRIL_NO_PERSISTENT_UPDATE=1 selects the per-step kernels");
    if (nan) return fail(h, DRIL_ERR_NAN_IN_GRADS, "gradient contains nan or is not finite (ppo.jl:213-214)");
    return DRIL_OK;
}

// ---- evaluate_agent (src/evaluation.jl:54-143) ------------------------------------------------------------
DRIL_EXPORT int32_t dril_evaluate_agent(dril_handle* h, int32_t n_eval, int32_t deterministic, dril_eval_stats* out, float* ep_rewards, int32_t* ep_lengths) {
    NEED(h); NOT_EXTERNAL(h, "dril_evaluate_agent");
    if (n_eval < 1 || !out) return fail(h, DRIL_ERR_INVALID_ARG, "dril_evaluate_agent: n_eval_episodes >= 1 and out != NULL");
    const int E = h->cfg.n_envs;
    int rc = ensure_wimg(h); if (rc) return rc;
    HIPCHK(h, launch_env_reset(h->cfg.env_kind, E, h->env_seed0, h->state, h->step_count, h->episode, h->gstep, h->disc_returns, h->stream));   // reset!(env), :87
    if (h->mon_cur_ret) { HIPCHK(h, hipMemsetAsync(h->mon_cur_ret, 0, (size_t)E * 4, h->stream)); HIPCHK(h, hipMemsetAsync(h->mon_cur_len, 0, (size_t)E * 4, h->stream)); }
    h->env_ready = true;
    const bool raw = h->mon_cur_ret != nullptr;                                  // monitored: infos[i]["episode"]["r"] is the raw return
    std::vector<float> rew(E), cur_r(E, 0.f), er; std::vector<uint8_t> term(E), trunc(E); std::vector<int32_t> cur_l(E, 0), el;
    float* rew_n = h->e_rew_n;                                                   // wrapper-delivered rewards (see dril_env_step)
    rc = observe_dev(h, true); if (rc) return rc;                                // observations = observe(env), :88
    int steps = 0;
    while ((int)er.size() < n_eval) {
        PolicyArgs p = policy_args(h, h->e_obs, E, nullptr, h->e_act, nullptr, h->logp /*scratch*/, nullptr, 0);
        p.gstep = h->gstep; p.env_seed0 = h->env_seed0; p.deterministic = deterministic ? 1 : 0;
        p.logp = h->e_rew;                                                       // logprobs are not needed: park them in a scratch array
        HIPCHK(h, run_policy(h, p));   // predict_actions(agent, observations; deterministic), :92
        rc = step_dev(h, h->e_act, rew_n, nullptr); if (rc) return rc;           // act!(env, actions), :94
        HIPCHK(h, hipMemcpyAsync(rew.data(), raw ? h->e_rew : rew_n, (size_t)E * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(term.data(), h->e_term, E, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(trunc.data(), h->e_trunc, E, hipMemcpyDeviceToHost, h->stream));
        rc = observe_dev(h, true); if (rc) return rc;                            // observations = observe(env), :97
        rc = sync(h); if (rc) return rc;
        ++steps;
        for (int i = 0; i < E; ++i) {
            cur_r[i] += rew[i]; cur_l[i] += 1;                                   // :95-96
            if ((term[i] || trunc[i]) && (int)er.size() < n_eval) { er.push_back(cur_r[i]); el.push_back(cur_l[i]); cur_r[i] = 0.f; cur_l[i] = 0; }   // :100-121
        }
        if (steps > 100000000 / (E > 0 ? E : 1) + 100000) return fail(h, DRIL_ERR_UNSUPPORTED, "dril_evaluate_agent: no episode finishes");
    }
    double mr = 0, ml = 0; for (int i = 0; i < n_eval; ++i) { mr += er[i]; ml += el[i]; }
    mr /= n_eval; ml /= n_eval;
    double vr = 0, vl = 0; for (int i = 0; i < n_eval; ++i) { vr += (er[i] - mr) * (er[i] - mr); vl += (el[i] - ml) * (el[i] - ml); }
    out->mean_reward = mr; out->mean_length = ml;
    out->std_reward = std::sqrt(vr / (n_eval - 1)); out->std_length = std::sqrt(vl / (n_eval - 1));     // Julia std: corrected; NaN for one episode
    out->n_episodes = n_eval; out->n_steps = steps;
    if (ep_rewards) std::memcpy(ep_rewards, er.data(), (size_t)n_eval * 4);
    if (ep_lengths) std::memcpy(ep_lengths, el.data(), (size_t)n_eval * 4);
    return DRIL_OK;
}

// ---- train! ------------------------------------------------------------------------------------------
DRIL_EXPORT int32_t dril_train(dril_handle* h, int64_t max_steps, dril_ppo_stats* stats, double* fps, int32_t* iterations_done) {
    NEED(h); NOT_EXTERNAL(h, "dril_train");
    const int64_t per_iter = (int64_t)h->cfg.n_steps * h->cfg.n_envs * h->cfg.world_size;
    const int64_t iterations = max_steps / per_iter;                                   // ppo.jl:117
    int32_t done = 0;
    for (int64_t i = 0; i < iterations; ++i) {
        h->lr = h->cfg.learning_rate;                                                  // ppo.jl:155-156
        double f = 0; int rc = collect_rollout(h, &f, false);                          // ppo.jl:167
        if (rc) { if (iterations_done) *iterations_done = done; return rc; }
        if (fps) fps[i] = f;
        rc = ppo_update(h, stats ? &stats[i] : nullptr);                               // ppo.jl:188-264
        if (rc) { if (iterations_done) *iterations_done = done; return rc; }
        ++done;
    }
    if (iterations_done) *iterations_done = done;
    return DRIL_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------
DRIL_EXPORT int32_t dril_comm_unique_id(uint8_t id[128]) {
    std::string err;
    if (!id) return fail(nullptr, DRIL_ERR_INVALID_ARG, "null id");
    if (!load_rccl(err)) return fail(nullptr, DRIL_ERR_RCCL, err);
    NcclId nid; std::memset(&nid, 0, sizeof(nid));
    const int rc = g_rccl.GetUniqueId(&nid);
    if (rc != 0) return fail(nullptr, DRIL_ERR_RCCL, rccl_error("ncclGetUniqueId", rc, nullptr));
    std::memcpy(id, nid.bytes, 128);
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_comm_init(dril_handle* h, const uint8_t id[128]) {
    NEED(h); std::string err;
    if (!id) return fail(h, DRIL_ERR_INVALID_ARG, "null id");
    if (!load_rccl(err)) return fail(h, DRIL_ERR_RCCL, err);
    NcclId nid; std::memcpy(nid.bytes, id, 128);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int rc = ((nccl_init_rank_fn)(void*)g_rccl.CommInitRank)(&h->comm, h->cfg.world_size, nid, h->cfg.rank);
    if (rc != 0) { const std::string m = rccl_error("ncclCommInitRank", rc, nullptr) + "; rank " + std::to_string(h->cfg.rank) + " of " + std::to_string(h->cfg.world_size) + " on " + h->device_info; h->comm = nullptr; return fail(h, DRIL_ERR_RCCL, m); }
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_debug_comm_loopback(dril_handle** hs, int32_t n) {
    if (!hs || n < 1 || n > kLoopMax) return fail(nullptr, DRIL_ERR_INVALID_ARG, "dril_debug_comm_loopback: 1..8 handles");
    std::vector<bool> seen((size_t)n, false);
    for (int i = 0; i < n; ++i) {
        dril_handle* h = hs[i];
        if (!h) return fail(nullptr, DRIL_ERR_INVALID_ARG, "dril_debug_comm_loopback: null handle");
        if (comm_ready(h)) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_comm_loopback: the handle already has a communicator");
        if (h->cfg.world_size != n || h->cfg.rank < 0 || h->cfg.rank >= n || seen[h->cfg.rank]) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_comm_loopback: the handles must be ranks 0..n-1 of world_size n, one each");
        if (h->cfg.device != hs[0]->cfg.device) return fail(h, DRIL_ERR_INVALID_ARG, "dril_debug_comm_loopback: all handles must live on one device");
        seen[h->cfg.rank] = true;
    }
    (void)hipSetDevice(hs[0]->cfg.device);
    LoopGroup* g = new LoopGroup(); g->n = n; g->refs = n;
    hipError_t e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    for (int q = 0; q < n && e == hipSuccess; ++q) e = hipEventCreateWithFlags(&g->ready[q], hipEventDisableTiming);
    if (e != hipSuccess) { for (int q = 0; q < n; ++q) if (g->ready[q]) hipEventDestroy(g->ready[q]); if (g->done) hipEventDestroy(g->done); delete g; return fail(hs[0], DRIL_ERR_HIP, hipGetErrorString(e)); }
    for (int i = 0; i < n; ++i) hs[i]->loop = g;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_comm_ranks(dril_handle* h) {
    if (!h) return -1;
    if (h->loop) return h->loop->n;
    if (h->comm) { int c = -1; if (g_rccl.CommCount && g_rccl.CommCount(h->comm, &c) == 0) return c; return -1; }
    return 1;
}
DRIL_EXPORT int64_t dril_comm_allreduce_calls(const dril_handle* h) { return h ? h->allreduce_calls : -1; }
DRIL_EXPORT const char* dril_device_info(const dril_handle* h) { return h ? h->device_info.c_str() : ""; }

// ---- measurement ----------------------------------------------------------------------------------------
DRIL_EXPORT int32_t dril_profile_get(dril_handle* h, int32_t kid, double* total_ms, int64_t* launches) {
    NEED(h); if (kid < 0 || kid >= DRIL_K_COUNT) return fail(h, DRIL_ERR_INVALID_ARG, "bad kernel id");
    int rc = sync(h); if (rc) return rc;
    if (total_ms) *total_ms = h->prof_ms[kid]; if (launches) *launches = h->prof_n[kid];
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_profile_launches(dril_handle* h, int32_t kid, int64_t* all_launches) {
    NEED(h); if (kid < 0 || kid >= DRIL_K_COUNT || !all_launches) return fail(h, DRIL_ERR_INVALID_ARG, "bad kernel id");
    *all_launches = h->prof_all[kid];
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_profile_reset(dril_handle* h) {
    NEED(h); int rc = sync(h); if (rc) return rc;
    for (int i = 0; i < DRIL_K_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; h->prof_all[i] = 0; }
    return DRIL_OK;
}
DRIL_EXPORT const char* dril_kernel_name(int32_t kid) {
    static const char* names[] = {"rollout_kernel", "gae_kernel", "adv_moments_kernel", "ppo_grad_kernel", "grad_reduce_kernel", "adam_kernel", "ncclAllReduce", "pack_records_kernel", "explained_var_kernel"};
    static_assert(sizeof(names) / sizeof(names[0]) == DRIL_K_COUNT, "one name per kernel class");
    return (kid >= 0 && kid < DRIL_K_COUNT) ? names[kid] : "?";
}
DRIL_EXPORT int32_t dril_kernel_count(void) { return DRIL_K_COUNT; }
DRIL_EXPORT const char* dril_grad_kernel_info(const dril_handle* h) {
    if (!h) return "?";
    switch (h->last_variant) {
        case 0: return "ppo_grad_kernel: f32 (v_mfma_f32_32x32x2_f32)";
        case 2: return "ppo_grad_wide_kernel: f32 (v_mfma_f32_32x32x2_f32)";
        case 6: return "ppo_update_small_kernel: f32 (f16x2 split, f32 accumulate; v_mfma_f32_32x32x16_f16 x 3 per k16 step on two-piece f16 operands = 2^-24 relative, input layer on v_mfma_f32_32x32x2_f32; two persistent workgroups (actor | critic), all optimiser steps of the iteration in one launch)";
        case 5: return "ppo_grad_pair_kernel: f32 (f16x2 split, f32 accumulate; v_mfma_f32_32x32x16_f16 x 3 per k16 step on two-piece f16 operands = 2^-24 relative, input layer on v_mfma_f32_32x32x2_f32)";
        case 4: return "ppo_grad_wide_split_kernel: f32 (f16x2 split, f32 accumulate; v_mfma_f32_32x32x16_f16 x 3 per k16 step on two-piece f16 operands = 2^-24 relative, input layer on f32 MFMAs, dW1 / dW3 in f32 on the vector ALU)";
        case 3: return "generic path: f32 contractions (sac_gemm_*; large ones bf16x3 split, f32 accumulate)";
        default: return "none yet";
    }
}
DRIL_EXPORT const char* dril_version(void) { return "dril_hip 0.2 (gfx950, abi 2)"; }

// dril_grad_f32.hip — ppo_grad_kernel: the exact-f32 update kernel of hidden [64,64] (v_mfma_f32_32x32x2_f32), small and medium minibatches; (alg::PPO)(layer,ps,st,batch) ppo.jl:365-407 + its reverse pass (Zygote in the reference, ppo.jl:207)
#include <utility>

#include "dril_grad_common.h"

namespace dril {

template <int D, int H, int O> struct GradScratch {
    static constexpr int T = 0;
    static constexpr int XI = T + H * kTS;
    static constexpr int ZI = XI + (D + 2) * kTS;
    static constexpr int SIZE = ZI + O * kTS;
};

template <int KIND, int H, int O, int HEAD, bool REC>
__device__ __forceinline__ void grad_body(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32;
    using L = NetLds<D, H, H, O>;
    using SC = GradScratch<D, H, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    // threadIdx.x / 64 IS wave-uniform but hipcc cannot prove it: readfirstlane moves the wave id - and every tile index,
    // LDS base and loop bound derived from it - into SGPRs (v3 spilled those to scratch, and each scratch reload's
    // s_waitcnt vmcnt(0) drained the prefetched gathers: profiles/r01 stamps, "out+head" 7.0k cycles)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    float* wl = smem;
    float* T = smem + L::BWD_END + wave * SC::SIZE + SC::T;
    float* XI = smem + L::BWD_END + wave * SC::SIZE + SC::XI;
    float* ZI = smem + L::BWD_END + wave * SC::SIZE + SC::ZI;
    stage_net<D, H, H, O, true>(wl, a.params, off, tid, blockDim.x);
    for (int i = lane; i < (D + 2) * kTS; i += 64) XI[i] = (i / kTS == D) ? 1.0f : 0.0f;
    __syncthreads();

    // advantage normalisation constants (ppo.jl:350-356): mean, corrected std, eps added to the std
    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        double s, q, n;
        if (a.inline_moments) {
            // small minibatch: sum A and A^2 of the whole minibatch here (same index map as load_tile), fixed-order tree => every workgroup gets the same bits
            double* shd = reinterpret_cast<double*>(smem + ((L::BWD_END + 1) & ~1));      // per-wave scratch, not yet in use
            double ls_ = 0, lq_ = 0;
            for (int64_t i2 = tid; i2 < a.count; i2 += blockDim.x) {
                const int64_t p2 = a.pos0 + i2;
                const int64_t gi = a.perm32 ? (int64_t)a.perm32[p2] : a.perm ? a.perm[p2] : (a.perm_bits ? perm_index(p2, a.N, a.perm_key, a.perm_bits) : p2);
                const int64_t li2 = gi - a.idx_lo;
                if (li2 >= 0 && li2 < a.n_local) { const float v = REC ? a.rec[RecLayout<D>::RS * li2 + RecLayout<D>::RS - 1].y : a.adv[li2]; ls_ += v; lq_ += (double)v * v; }
            }
            shd[tid] = ls_; shd[256 + tid] = lq_;
            __syncthreads();
            for (int st_ = 128; st_ > 0; st_ >>= 1) { if (tid < st_) { shd[tid] += shd[tid + st_]; shd[256 + tid] += shd[256 + tid + st_]; } __syncthreads(); }
            s = shd[0]; q = shd[256]; n = (double)a.count;
            __syncthreads();
        } else { s = a.adv_stats[0]; q = a.adv_stats[1]; n = a.adv_stats[2]; }
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    const float adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    constexpr bool LS_GAUSS = HEAD == HEAD_GAUSSIAN; constexpr int LS_N = O;
    // log_std hoisted into scalar registers: a per-tile global load would sit in the in-order vmcnt queue between the prefetched
    // gathers and their first use and drain them every tile (Pendulum [64,64]: 91 -> TFLOP/s below)
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (LS_GAUSS) {
#pragma unroll
        for (int o = 0; o < LS_N; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;

    f32x16 dW2[MT][MT];
    f32x4 dW1[H / 16];                                             // 16x16x4 tiles: rows = hidden, cols = [x | 1 | 0...]
    float dW3a[O][MT], db2p[MT], db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int i = 0; i < H / 16; ++i) dW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        db2p[i] = 0.f;
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        db3p[o] = 0.f; dlsp[o] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) dW3a[o][m] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = (int)(blockIdx.x % a.G);                          // the first G workgroups run the actor, the next G the critic
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    const int64_t tstride = (int64_t)a.G * 4, first = (int64_t)g * 4 + wave;
    constexpr int KS = FirstLayer<D>::KS;
    TileIn<O, KS> cur, nxt;
    int64_t tile = first;
    if (tile < ntiles) load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += tstride) {
        load_tile<KIND, O, HEAD, REC>(a, tile + tstride, ntiles, c, h, nxt);      // prefetch the next tile's gathers
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        float xk[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xk[s] = cur.xk[s];
        STAMP(0);
        // ---- forward ----
        f32x16 h1[MT], h2[MT];
        float out[O], dz[O];
        dense_first<H, MT, KS>(wl + L::W1T, wl + L::B1, xk, h1, lane);
        tanh_tiles(h1);
        STAMP(1);
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) {
            h2[mo] = dense_mfma_tile<MT, true>(wl + L::W2S, L::WS1, wl + L::B2, h1, mo, lane);
            tanh16(h2[mo]);
        }
        store_image<MT>(T, h2, lane);          // early: the LDS write -> read round trip hides under the head below
        STAMP(2);
        dense_out<MT, O, H>(wl + L::W3S, wl + L::B3, h2, out, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- loss head (ppo.jl:377-404) and dLoss/dout ----
        loss_head<O, HEAD>(a, cur, out, valid, h == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
        STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- output layer backward: dW3 += dz * h2' over samples (h2 read back transposed: hidden on the lane) ----
#pragma unroll
        for (int o = 0; o < O; ++o) { if (h == 0) { db3p[o] += dz[o]; ZI[o * kTS + c] = dz[o]; } }
        {
            f32x16 Bh2[MT];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) Bh2[mj] = load_operand(T, mj, lane);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float acc[MT];
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) acc[mj] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(ZI + o * kTS + 16 * h + 4 * q);   // broadcast within the half-wave
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) {
                        acc[mj] = fmaf(Bh2[mj][4 * q + 0], z[0], acc[mj]); acc[mj] = fmaf(Bh2[mj][4 * q + 1], z[1], acc[mj]);
                        acc[mj] = fmaf(Bh2[mj][4 * q + 2], z[2], acc[mj]); acc[mj] = fmaf(Bh2[mj][4 * q + 3], z[3], acc[mj]);
                    }
                }
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) dW3a[o][mj] += acc[mj];
            }
        }
        STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dz2 = (W3' dz) .* (1 - h2^2), in h2's registers ----
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * m + 8 * q + 4 * h);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(w[cc], dz[o], dh[cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) { const float hv = h2[m][4 * q + cc]; h2[m][4 * q + cc] = dh[cc] * (1.0f - hv * hv); }
            }
        STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        // ---- h1 image (the LDS unit executes a wave's accesses in order, so the Bh2 reads above precede these writes) ----
        store_image<MT>(T, h1, lane);
        // ---- dh1 = W2' dz2 ; dz1 = dh1 .* (1 - h1^2) ----
        f32x16 g1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            g1[m] = dense_mfma_tile<MT, false>(wl + L::W2T, L::WS2, nullptr, h2, m, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[m][r] = g1[m][r] * (1.0f - h1[m][r] * h1[m][r]);
        }
        STAMP(6);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dW2 += dz2 * h1' ; db2 += rowsum(dz2) ----
        {
            f32x16 Bh[MT];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) Bh[mj] = load_operand(T, mj, lane);
            store_image<MT>(T, h2, lane);                                      // dz2 image
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const f32x16 Az = load_operand(T, mi, lane);
                db2p[mi] += sum16(Az);
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) dW2[mi][mj] = mfma_outer(Az, Bh[mj], dW2[mi][mj]);
            }
        }
        STAMP(7);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dW1 | db1 += dz1 * [x; 1]' ----
        store_image<MT>(T, g1, lane);
#pragma unroll
        for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; XI[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
        {   // v_mfma_f32_16x16x4_f32: M = 16 hidden rows, N = 16 columns [x_0..x_{D-1}, 1, 0...], K = 4 samples per step
            const int j = lane & 15;
            float bx[8];
            load_row8(XI, j <= D ? j : D + 1, lane, bx);
#pragma unroll
            for (int mt = 0; mt < H / 16; ++mt) {
                float az[8];
                load_row8(T, 16 * mt + j, lane, az);
#pragma unroll
                for (int k = 0; k < 8; ++k) dW1[mt] = mfma16(az[k], bx[k], dW1[mt]);
            }
        }
        STAMP(8);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int k = 0; k < 10; ++k) o[k] = stamp_acc[k];
        o[10] = (unsigned long long)((ntiles - first + tstride - 1) / tstride); o[11] = HEAD;
    }
#endif

    // ---- epilogue: 4 waves -> one slab (fixed wave order => deterministic) ----
    __syncthreads();
    float* red = smem + L::BWD_END;
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    for (int i = tid; i < SL; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mi + rowfn(r, h);
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) red[o_w2 + row + (32 * mj + c) * H] += dW2[mi][mj][r];
                }
                const float b2 = db2p[mi] + __shfl_xor(db2p[mi], 32);
                if (h == 0) red[o_b2 + 32 * mi + c] += b2;
            }
#pragma unroll
            for (int mt = 0; mt < H / 16; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + 4 * (lane >> 4) + r, col = lane & 15;
                    if (col < D) red[o_w1 + row + col * H] += dW1[mt][r];
                    else if (col == D) red[o_b1 + row] += dW1[mt][r];
                }
#pragma unroll
            for (int o = 0; o < O; ++o) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float v = dW3a[o][m] + __shfl_xor(dW3a[o][m], 32);      // the two halves hold different samples
                    if (h == 0) red[o_w3 + o + (32 * m + c) * O] += v;
                }
                const float b3 = half_sum(db3p[o]);
                if (lane == 0) red[o_b3 + o] += b3;
                if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (lane == 0) red[o_ls + o] += l; }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (lane == 0) red[o_st + k] += v; }
        }
        __syncthreads();
    }
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
    for (int i = tid; i < SL; i += blockDim.x) slab[i] = red[i];
}

template <int KIND, int H, bool REC>
__global__ __launch_bounds__(256, 2) void ppo_grad_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, REC>(a, smem);
    else grad_body<KIND, H, 1, HEAD_VALUE, REC>(a, smem);
}

template <int KIND, int H> static size_t grad_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = NetLds<D, H, H, A>::BWD_END + 4 * GradScratch<D, H, A>::SIZE;
    constexpr int wc = NetLds<D, H, H, 1>::BWD_END + 4 * GradScratch<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}

// kind 2 (ScalingWrapperEnv(Pendulum)) shares every kernel that never touches the simulator with kind 1
// hidden 64, and 32 (the reference's own benchmark suite shape, benchmark/bench_utils.jl:31,49: one m-tile per layer on the same templates)
#define DRIL_DISPATCH_HH(K, hidden, CALL) { if ((hidden) == 64) { CALL(K, 64); } else if ((hidden) == 32) { CALL(K, 32); } else return hipErrorInvalidValue; }
#define DRIL_DISPATCH(kind, hidden, CALL)                                            \
    do {                                                                             \
        if ((kind) == 0) DRIL_DISPATCH_HH(0, hidden, CALL)                           \
        else if ((kind) == 1 || (kind) == 2) DRIL_DISPATCH_HH(1, hidden, CALL)       \
        else if ((kind) == 3) DRIL_DISPATCH_HH(3, hidden, CALL)                      \
        else if ((kind) == 4 || (kind) == 7) DRIL_DISPATCH_HH(4, hidden, CALL)       \
        else if ((kind) == 6) DRIL_DISPATCH_HH(6, hidden, CALL)                      \
        else return hipErrorInvalidValue;                                            \
    } while (0)

hipError_t launch_ppo_grad_f32(int kind, int hidden, const GradArgs& a, hipStream_t s) {
#define CALLR(K, HH, R)                                                                                       \
    {                                                                                                         \
        const size_t lds = grad_lds_bytes<K, HH>();                                                           \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_grad_kernel<K, HH, R>, lds); if (e != hipSuccess) return e; } \
        ppo_grad_kernel<K, HH, R><<<2 * a.G, 256, lds, s>>>(a);                                               \
    }
#define CALL(K, HH) { if (a.rec) CALLR(K, HH, true) else CALLR(K, HH, false) }
    DRIL_DISPATCH(kind, hidden, CALL);
#undef CALL
#undef CALLR
    return hipGetLastError();
}

}  // namespace dril

// dril_grad_wide.hip — the update kernels of hidden [128,128] and [256,256]: ppo_grad_wide_kernel (exact f32 MFMA; here) and ppo_grad_wide_split_kernel (f16 matrix cores,
// fp32-equivalent two-piece operand split; the default; dril_grad_wide_split.h), and their launcher
#include <utility>

#include "dril_grad_common.h"
#include "dril_split_pieces.h"
#include "dril_grad_wide_split.h"   // ppo_grad_wide_split_kernel (the f16-piece form)

namespace dril {

// =============================================================================================
// ppo_grad_wide_kernel — the same fused forward + loss + backward for hidden widths that do not fit one wave
// (H = 256: W2 is 256 KB, dW2 is 64 K accumulators).  A workgroup of H/32 waves processes a 32-sample tile TOGETHER:
// wave w owns m-tile w (hidden rows 32w..32w+31) of every layer and the 32 x H slice of dW2 in registers (H/2 VGPRs);
// activations are exchanged through LDS (B-operand images [m][lane][16]), the two H x H operand streams W2 / W2' come
// pre-tiled from L2 (dril_device.h "wide nets").  Four workgroup barriers per tile.  Per tile and wave:
//   2 (L1) + 4*MT (L2) + 4*MT (dh1) + 4*MT... in 32x32x2 units: L2 16*MT, dh1 16*MT, dW2 16*MT, dW1 1.
// Every wave owns distinct rows of every gradient, so the slab is written straight from registers (no cross-wave sum).
// =============================================================================================
template <int D, int H, int O> struct WideScratch {
    static constexpr int MT = H / 32;
    static constexpr int SMALL = NetLdsSmall<D, H, O>::END;
    static constexpr int XA = (SMALL + 3) / 4 * 4;          // h1 as B-operand image [MT][64][16]
    static constexpr int XB = XA + MT * 1024;               // dz2 as B-operand image
    static constexpr int TA = XB + MT * 1024;               // h1 transposed [H][kTS]
    static constexpr int TB = TA + H * kTS;                 // per-wave rows: h2', then dz2', then dz1'
    static constexpr int XI = TB + H * kTS;                 // [D+2][kTS]
    static constexpr int ZI = XI + (D + 2) * kTS;           // [MT waves][O][kTS]
    static constexpr int PO = ZI + MT * O * kTS;            // [MT waves][O][32] output-layer partial sums
    static constexpr int SIZE = PO + MT * O * 32;
};

__device__ __forceinline__ void store_breg(float* img, int m, const f32x16& x, int lane) {
    float* p = img + ((size_t)m * 64 + lane) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
}
__device__ __forceinline__ f32x16 load_breg(const float* img, int m, int lane) {
    const float* p = img + ((size_t)m * 64 + lane) * 16;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q); v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3]; }
    return v;
}
__device__ __forceinline__ void store_image_tile(float* img, int m, const f32x16& x, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) img[(32 * m + rowfn(r, h)) * kTS + c] = x[r];
}
// Y tile mo = W * X with W from the pre-tiled global image and X read tile by tile from an LDS B-operand image
__device__ __forceinline__ void wide_preload(const float* __restrict__ wimg, int MTv, int mo, int lane, f32x4 (&af)[4]) {
    const float* base = wimg + ((size_t)mo * MTv * 4 * 64 + lane) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(base + (size_t)q * 256);
}
template <int MT, bool BIAS>
__device__ __forceinline__ f32x16 dense_tile_global_ldsB(const float* __restrict__ wimg, const float* __restrict__ bias, const float* __restrict__ ximg, int mo, int lane, f32x4 (&af)[4]) {
    const int h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
    const float* base = wimg + ((size_t)mo * MT * 4 * 64 + lane) * 4;
    // The A fragments stream from L2 (pre-tiled image, 1 KiB contiguous per wave-instruction).  Each of the four fragment registers is refilled
    // with the NEXT m-tile's fragment right after the four MFMAs that consumed it were issued, so every load has 12-16 MFMAs (~1 k cycles) of
    // cover with only 16 registers of buffering (the loop stays rolled: fully unrolled, hipcc hoists all 32 loads and spills 330 VGPRs)
    // af[] arrives preloaded with the first m-tile's fragments (wide_preload, issued before the workgroup barrier that precedes this chain)
#pragma unroll 1
    for (int mi = 0; mi < MT; ++mi) {
        const f32x16 X = load_breg(ximg, mi, lane);
        const float* nextp = base + (size_t)((mi + 1 < MT ? mi + 1 : mi) * 4) * 256;   // last iteration re-reads its own fragments (in bounds, unused)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = mfma32(af[q][0], X[4 * q + 0], acc); acc = mfma32(af[q][1], X[4 * q + 1], acc);
            acc = mfma32(af[q][2], X[4 * q + 2], acc); acc = mfma32(af[q][3], X[4 * q + 3], acc);
            af[q] = *reinterpret_cast<const f32x4*>(nextp + (size_t)q * 256);
        }
    }
    return acc;
}

template <int KIND, int H, int O, int HEAD, bool REC>
__device__ __forceinline__ void grad_body_wide(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideScratch<D, H, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's m-tile
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    const float* w2a = HEAD == HEAD_VALUE ? a.w2a_critic : a.w2a_actor;
    const float* w2ta = HEAD == HEAD_VALUE ? a.w2ta_critic : a.w2ta_actor;
    float* wl = smem;
    float* XA = smem + SC::XA; float* XB = smem + SC::XB; float* TA = smem + SC::TA; float* TB = smem + SC::TB;
    float* XI = smem + SC::XI; float* ZI = smem + SC::ZI + w * O * kTS; float* PO = smem + SC::PO;
    stage_net_small<D, H, O>(wl, a.params, off, tid, blockDim.x);
    for (int i = tid; i < (D + 2) * kTS; i += blockDim.x) XI[i] = (i / kTS == D) ? 1.0f : 0.0f;
    __syncthreads();

    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        const double s = a.adv_stats[0], q = a.adv_stats[1], n = a.adv_stats[2];
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

    f32x16 dW2[MT];                                                  // rows 32w.., all H columns
    f32x4 dW1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dW3a[O], db2p = 0.f, db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) { dW3a[o] = 0.f; db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = (int)(blockIdx.x % a.G);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    TileIn<O, FirstLayer<D>::KS> cur, nxt;
    int64_t tile = g;
    if (tile < ntiles) load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += a.G) {
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        constexpr int KS = FirstLayer<D>::KS;                                           // two first-layer k-steps for D <= 4, four for D <= 8 (Acrobot)
        float xk[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xk[s] = cur.xk[s];
        // ---- S2: h1 tile w ----
        f32x16 h1w;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16(h1w);
        }
        store_breg(XA, w, h1w, lane);
        store_image_tile(TA, w, h1w, lane);
        if (w == 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; XI[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
        }
        STAMP(0);
        f32x4 afw[4];
        wide_preload(w2a, MT, w, lane, afw);                                          // first W2 fragments in flight across the barrier
        __syncthreads();                                                              // B1: XA, TA, XI complete
        STAMP(1);
        // prefetch the next tile's record only now: issued before unpack_tile(cur) it sat behind cur's loads in the in-order vmcnt queue and the
        // spill reloads' s_waitcnt vmcnt(0) made every tile wait for a full gather latency (stamps: 10 k cycles in this phase)
        load_tile<KIND, O, HEAD, REC>(a, tile + a.G, ntiles, c, h, nxt);
        // ---- S3: h2 tile w ----
        f32x16 h2w = dense_tile_global_ldsB<MT, true>(w2a, wl + L::B2, XA, w, lane, afw);
        tanh16(h2w);
        STAMP(2);
        // ---- S4: output layer: partial over this wave's rows, summed across waves through LDS ----
        float out[O], dz[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
                p = fmaf(wv[0], h2w[4 * q + 0], p); p = fmaf(wv[1], h2w[4 * q + 1], p);
                p = fmaf(wv[2], h2w[4 * q + 2], p); p = fmaf(wv[3], h2w[4 * q + 3], p);
            }
            p += __shfl_xor(p, 32);
            if (h == 0) PO[(w * O + o) * 32 + c] = p;
        }
        store_image_tile(TB, w, h2w, lane);                                            // h2' (own rows; only this wave reads them)
        __syncthreads();                                                              // B2: PO complete
        STAMP(3);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = wl[L::B3 + o];
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) v += PO[(ww * O + o) * 32 + c];            // fixed order: every wave gets the same bits
            out[o] = v;
        }
        loss_head<O, HEAD>(a, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
        // ---- dW3 (own rows) ----
#pragma unroll
        for (int o = 0; o < O; ++o) { if (h == 0) { if (w == 0) db3p[o] += dz[o]; ZI[o * kTS + c] = dz[o]; } }
        {
            const f32x16 Bh2 = load_operand(TB, w, lane);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(ZI + o * kTS + 16 * h + 4 * q);
                    acc = fmaf(Bh2[4 * q + 0], z[0], acc); acc = fmaf(Bh2[4 * q + 1], z[1], acc);
                    acc = fmaf(Bh2[4 * q + 2], z[2], acc); acc = fmaf(Bh2[4 * q + 3], z[3], acc);
                }
                dW3a[o] += acc;
            }
        }
        // ---- dz2 tile w (in h2w's registers) ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * (1.0f - hv * hv); }
        }
        store_breg(XB, w, h2w, lane);
        store_image_tile(TB, w, h2w, lane);                                            // dz2' (after the Bh2 read: same wave, LDS in order)
        wide_preload(w2ta, MT, w, lane, afw);                                         // first W2' fragments in flight across the barrier
        STAMP(4);
        __syncthreads();                                                              // B3: XB complete
        STAMP(5);
        // ---- S6: dh1 tile w = W2' dz2 ; dz1 ----
        f32x16 g1 = dense_tile_global_ldsB<MT, false>(w2ta, nullptr, XB, w, lane, afw);
        {
            const f32x16 h1r = load_breg(XA, w, lane);                                 // h1 tile w re-read from its LDS image: 16 registers less across the two MFMA chains
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = g1[r] * (1.0f - h1r[r] * h1r[r]);
        }
        STAMP(6);
        // ---- S7: dW2[rows of w][:] += dz2 h1' ----
        {
            const f32x16 Az = load_operand(TB, w, lane);
            db2p += sum16(Az);
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                const f32x16 Bh = load_operand(TA, mj, lane);
                dW2[mj] = mfma_outer(Az, Bh, dW2[mj]);
            }
        }
        STAMP(7);
        // ---- S8: dW1 | db1 (own rows) ----
        store_image_tile(TB, w, g1, lane);
        {
            const int j = lane & 15;
            float bx[8];
            load_row8(XI, j <= D ? j : D + 1, lane, bx);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float az[8];
                load_row8(TB, 32 * w + 16 * t + j, lane, az);
#pragma unroll
                for (int k = 0; k < 8; ++k) dW1[t] = mfma16(az[k], bx[k], dW1[t]);
            }
        }
        __syncthreads();                                                              // B4: XA/TA/XB/PO/XI free for the next tile
        STAMP(8);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
        if (w < 4) { for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k]; o_[10] = (unsigned long long)((ntiles - g + a.G - 1) / a.G); o_[11] = HEAD; }
    }
#endif

    // ---- epilogue: every wave owns distinct gradient rows -> straight to the workgroup's slab ----
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 32 * w + 16 * t + 4 * (lane >> 4) + r, col = lane & 15;
            if (col < D) slab[o_w1 + row + col * H] = dW1[t][r];
            else if (col == D) slab[o_b1 + row] = dW1[t][r];
        }
    { const float b2 = db2p + __shfl_xor(db2p, 32); if (h == 0) slab[o_b2 + 32 * w + c] = b2; }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float v = dW3a[o] + __shfl_xor(dW3a[o], 32);
        if (h == 0) slab[o_w3 + o + (32 * w + c) * O] = v;
        const float b3 = half_sum(db3p[o]);
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
}

template <int KIND, int H, bool REC>
__global__ __launch_bounds__(H * 2, 2) void ppo_grad_wide_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body_wide<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, REC>(a, smem);
    else grad_body_wide<KIND, H, 1, HEAD_VALUE, REC>(a, smem);
}



template <int KIND, int H> static size_t grad_wide_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = WideScratch<D, H, A>::SIZE, wc = WideScratch<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}
template <int KIND, int H> static size_t grad_wide_split_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = WideSplitScratch<D, H, A, kWideSplitNT>::SIZE, wc = WideSplitScratch<D, H, 1, kWideSplitNT>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}

hipError_t launch_ppo_grad_wide(int kind, int hidden, const GradArgs& a, hipStream_t s) {
    if (kind == 7) kind = 4;                  // ScalingWrapperEnv(MountainCarContinuous): the update never touches the simulator
    if (a.variant && a.rec) {      // f16 matrix cores (two-piece split)
#define CALLWS(K, HH)                                                                                         \
    {                                                                                                         \
        const size_t lds = grad_wide_split_lds_bytes<K, HH>();                                                \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_grad_wide_split_kernel<K, HH, 1>, lds); if (e != hipSuccess) return e; } \
        ppo_grad_wide_split_kernel<K, HH, 1><<<2 * a.G, HH * 2, lds, s>>>(a);                                    \
    }
#define CALLWSH(K) { if (hidden == 256) CALLWS(K, 256) else if (hidden == 128) CALLWS(K, 128) else return hipErrorInvalidValue; }
        if (kind == 0) CALLWSH(0) else if (kind == 3) CALLWSH(3) else if (kind == 4) CALLWSH(4) else if (kind == 6) CALLWSH(6) else CALLWSH(1)
#undef CALLWSH
#undef CALLWS
        return hipGetLastError();
    }
#define CALLW(K, HH, R)                                                                                       \
    {                                                                                                         \
        const size_t lds = grad_wide_lds_bytes<K, HH>();                                                      \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_grad_wide_kernel<K, HH, R>, lds); if (e != hipSuccess) return e; } \
        ppo_grad_wide_kernel<K, HH, R><<<2 * a.G, HH * 2, lds, s>>>(a);                                       \
    }
#define CALLWK(K, HH) { if (a.rec) CALLW(K, HH, true) else CALLW(K, HH, false) }
#define CALLWH(K) { if (hidden == 256) CALLWK(K, 256) else if (hidden == 128) CALLWK(K, 128) else return hipErrorInvalidValue; }
    if (kind == 0) CALLWH(0) else if (kind == 3) CALLWH(3) else if (kind == 4) CALLWH(4) else if (kind == 6) CALLWH(6) else CALLWH(1)
#undef CALLWH
#undef CALLWK
#undef CALLW
    return hipGetLastError();
}

}  // namespace dril

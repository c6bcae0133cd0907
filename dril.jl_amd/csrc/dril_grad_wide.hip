// dril_grad_wide.hip — the update kernels of hidden [128,128] and [256,256]: ppo_grad_wide_kernel (exact f32 MFMA) and ppo_grad_wide_split_kernel (bf16 matrix cores, fp32-equivalent 3-piece operand split; the default)
#include <utility>

#include "dril_grad_common.h"
#include "dril_split_pieces.h"

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

// =============================================================================================
// ppo_grad_wide_split_kernel — ppo_grad_wide_kernel with its three H x H contractions on the f16 matrix cores (fp32-equivalent two-piece operand
// splitting, dril_device.h).  A workgroup of H/32 waves owns a tile of kWideSplitNT x 32 samples; wave w owns the m-tile w of every layer and the
// 32 x H slice of dW2 (H/2 VGPRs).  Round 5 form:
//   * W2 / W2' stream from L2 as PRE-SPLIT f16 fragments (build_wimg_split_kernel, once per optimiser step): [(mo*MT + mi)*2 + s][piece][lane][8 f16],
//     one 16-byte load per lane, piece and k16 step.  At one 32-sample tile per pass that stream was the bound of the two streaming stages (2 x 256 KB
//     per tile and CU at H = 256 = 42 B/clk/CU, 22 TB/s chip-wide against the L2's 34.5: profiles/r05_wide_split.md); every fragment now feeds the MFMAs
//     of NT = 2 sample tiles, so the stream per sample is halved and the workgroup crosses its four barriers once per 64 samples.
//   * activations: every wave splits its own 16 registers once and writes the packed pieces into ONE workgroup image per activation set and sample tile,
//     [piece][32 samples][H units] f16, 16-byte chunk ch of row n stored at ch ^ g(n), g(n) = ((n & 3) << 2) | ((n >> 2) & 3).  The same image gives the
//     operand of a product that sums over units (ds_read_b128 along the row: 8 consecutive units of one sample) and both operands of the product
//     that sums over samples (ds_read_b64_tr_b16: 4 samples x 16 units per 16-lane group).
//   * no f32 transpose images any more (36 KB at H = 256 — the room the second sample tile needed): the three small products that sum over SAMPLES get their operands another way.
//       dh1 is computed TRANSPOSED — the same two register operands in the other order give D' — so its accumulator holds (lane = unit, register = sample): dz1 needs h1 in that
//         layout (four transposed reads per piece from the h1 image), and dW1 | db1 become per-lane sums over the lane's 16 samples against x read as LDS broadcasts
//         (D + 1 accumulators instead of two 16x16x4 MFMA tiles and a 4.6 KB image per wave);
//       dW3: the products h2 dz are summed over the two sample tiles in registers and reduced across the lanes of a half-wave by the register-halving DPP
//         reduce-scatter (half_reduce16_lane: ~52 VALU per output, once per 64 samples);
//       db2: from the transposed f16 fragments of dz2 that the dW2 product loads anyway (v_dot2c_f32_f16 against ones).
//   * no AGPRs: at two waves per SIMD the allocator gives a function that uses ANY AGPR only 128 VGPRs; the 128 dW2 accumulators are VGPR-form MFMA results like the rest.
// LDS at H = 256: two piece images of 64 KB + ~12 KB of small parameters = 143 - 156 KB of the CU's 160 (one workgroup per CU, two waves per SIMD); H = 128: 72 KB, two workgroups per CU.
// =============================================================================================
template <int D, int H, int O, int NT> struct WideSplitScratch {
    static constexpr int MT = H / 32;
    static constexpr int SMALL = NetLdsSmall<D, H, O>::END;
    static constexpr int P1 = (SMALL + 3) / 4 * 4;          // h1 pieces: NT x 2 x 32 x H f16 = NT 32 H floats
    static constexpr int P2 = P1 + NT * 32 * H;             // dz2 pieces
    static constexpr int XI = P2 + NT * 32 * H;             // [NT][D+2][kTS]: the tile's observations, component-major (row D + 1 takes the padding components' writes)
    static constexpr int PO = XI + NT * (D + 2) * kTS;      // [NT][MT waves][O][32] output-layer partial sums
    static constexpr int W3B = PO + NT * MT * O * 32;       // W3 / kActScale^2 (dh); the staged W3S is W3 / kActScale (output layer on kActScale h2)
    static constexpr int RQ = RecLayout<D>::RS == 3 ? 4 : 2; // record quads kept per sample tile (three-quad records: the second DMA's upper half-wave lands in a fourth)
    static constexpr int REC = W3B + O * H;                 // [NT][RQ][32] float4: the pass's minibatch records, quad-major (wide_request_records)
    static constexpr int VO = REC + NT * RQ * 32 * 4;       // [NT][64] old values (critic with a value clip)
    static constexpr int VAL = VO + NT * 64;                // [NT][64] validity words
    static constexpr int SIZE = VAL + NT * 64;
    static_assert(REC % 4 == 0, "records: 16-byte aligned");
    static_assert(SIZE * 4 <= 160 * 1024, "ppo_grad_wide_split_kernel: LDS");
};
// LDS-DMA (global_load_lds_dwordx4 / _dword): lane l's 16 / 4 bytes land at the wave-uniform LDS base + l x size; no destination registers, counted by vmcnt
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// the minibatch records of ONE sample tile straight into the workgroup's LDS, by one wave: lane (c, h) fetches quad h of sample c's record -> rec_t[h][c] (a three-quad
// record's scalar quad with a second instruction -> rec_t[2][c]), the old value -> vo_t[c], and writes the validity word.  Until round 5 every one of the H/32 waves
// gathered the same records into its own registers (8 - 12 of them, live across the whole pass) in front of a streaming chain, whose first fragment wait then sat out the gather
template <int KIND, int HEAD>
__device__ __forceinline__ void wide_request_records(const GradArgs& a, const TileIdx& ti, int lane, float* rec_t, float* vo_t, int* val_t) {
    constexpr int RS = RecLayout<EnvSpec<KIND>::D>::RS;
    const int64_t li = ti.gidx - a.idx_lo;
    const bool valid = ti.inb && li >= 0 && li < a.n_local;
    const int64_t idx = valid ? li : 0;
    glds16(a.rec + RS * idx + (lane >> 5), rec_t);
    if (RS == 3) glds16(a.rec + RS * idx + 2, rec_t + 64 * 4);
    if (HEAD == HEAD_VALUE && a.has_clip_vf) glds4(a.val_old + idx, vo_t);
    val_t[lane] = valid ? 1 : 0;
}
__device__ __forceinline__ void wide_split_preload(const u32x4* __restrict__ wimg, int MTv, int mo, int lane, u32x4 (&af)[2][2]) {
    const u32x4* base = wimg + ((size_t)mo * MTv * 4) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[s][p] = base[(size_t)(s * 2 + p) * 64];
}
// output m-tile mo of Y = W X for NT sample tiles at once: W as pre-split fragments from L2 (af arrives preloaded with m-tile 0's, each refilled in place right after its MFMAs),
// X from the piece images (one per sample tile, NTS bytes apart).  TRANSPOSED: the two operands in the other order — the accumulator then holds Y' (lane = unit 32 mo + (lane & 31),
// register r = sample rowfn(r, lane >> 5))
template <int H, int NT, bool BIAS, bool TRANSPOSED>
__device__ __forceinline__ void dense_tile_split(const u32x4* __restrict__ wimg, const float* __restrict__ bias, const char* pimg, int mo, int lane, u32x4 (&af)[2][2], f32x16 (&acc)[NT]) {
    constexpr int MT = H / 32, RB = 2 * H, PS = 32 * RB, NTS = 2 * PS;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB, gsw = wimg_g<H>(c);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc[t][4 * q + 0] = b[0]; acc[t][4 * q + 1] = b[1]; acc[t][4 * q + 2] = b[2]; acc[t][4 * q + 3] = b[3]; }
    }
    const u32x4* base = wimg + ((size_t)mo * MT * 4) * 64 + lane;
    // the activation fragments of a k16 step are requested one step AHEAD of the MFMAs that use them (two named register sets, X0 for the even steps and X1 for the odd ones:
    // no copies): read-then-wait-then-MFMA left every step's six MFMAs behind an LDS round trip, with both waves of a SIMD in the same place
    f16x8 X0[NT][2], X1[NT][2];
    {
        const int a = rowb + (((0 + h) ^ gsw) << 4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) X0[t][p] = *reinterpret_cast<const f16x8*>(pimg + t * NTS + p * PS + a);
    }
#pragma unroll 1
    for (int mi = 0; mi < MT; ++mi) {
        const int mn = mi + 1 < MT ? mi + 1 : mi;                                      // the last iteration re-reads its own fragments (in bounds, unused)
        const u32x4* nextp = base + (size_t)(mn * 4) * 64;
        {
            const int a = rowb + (((4 * mi + 2 + h) ^ gsw) << 4);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) X1[t][p] = *reinterpret_cast<const f16x8*>(pimg + t * NTS + p * PS + a);
        }
        __builtin_amdgcn_sched_barrier(0);                                             // (the scheduler otherwise sinks the reads back to just in front of their MFMAs to save the registers)
        {
            const f16x8 W0 = __builtin_bit_cast(f16x8, af[0][0]), W1 = __builtin_bit_cast(f16x8, af[0][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = TRANSPOSED ? mfma_split3(X0[t][0], X0[t][1], W0, W1, acc[t]) : mfma_split3(W0, W1, X0[t][0], X0[t][1], acc[t]);
#pragma unroll
            for (int p = 0; p < 2; ++p) af[0][p] = nextp[(size_t)p * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int a = rowb + (((4 * mn + h) ^ gsw) << 4);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) X0[t][p] = *reinterpret_cast<const f16x8*>(pimg + t * NTS + p * PS + a);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const f16x8 W0 = __builtin_bit_cast(f16x8, af[1][0]), W1 = __builtin_bit_cast(f16x8, af[1][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = TRANSPOSED ? mfma_split3(X1[t][0], X1[t][1], W0, W1, acc[t]) : mfma_split3(W0, W1, X1[t][0], X1[t][1], acc[t]);
#pragma unroll
            for (int p = 0; p < 2; ++p) af[1][p] = nextp[(size_t)(2 + p) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// sum of the eight f16 values of a fragment register set, in f32 (v_dot2c_f32_f16 against {1, 1}: the products are exact, the sum is an f32 sum)
__device__ __forceinline__ float frag_sum8(f16x8 v, float acc) {
    const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_fdot2(f16x2_t{v[2 * k], v[2 * k + 1]}, one, acc, false);
    return acc;
}

template <int KIND, int H, int O, int HEAD>
__device__ __forceinline__ void grad_body_wide_split(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32, NT = kWideSplitNT;
    constexpr int RB = 2 * H, PS = 32 * RB, NTS = 2 * PS;            // bytes of an image row, of one piece of a sample tile, of a sample tile's image
    constexpr bool REC = true;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideSplitScratch<D, H, O, NT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's m-tile
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    const u32x4* w2p = HEAD == HEAD_VALUE ? a.w2p_critic : a.w2p_actor;
    const u32x4* w2tp = HEAD == HEAD_VALUE ? a.w2tp_critic : a.w2tp_actor;
    float* wl = smem;
    char* P1 = reinterpret_cast<char*>(smem + SC::P1); char* P2 = reinterpret_cast<char*>(smem + SC::P2);
    float* XI = smem + SC::XI; float* PO = smem + SC::PO;
    float* RECS = smem + SC::REC; float* VO = smem + SC::VO; int* VAL = reinterpret_cast<int*>(smem + SC::VAL);
    constexpr int RS = RecLayout<D>::RS, RECT = SC::RQ * 32 * 4;    // record quads per sample; floats of one sample tile's record block
    // staged small parameters in the scales of the f16-piece arithmetic (dril_device.h): b2 starts the SCALED accumulator of L2, W3S = W3 / kActScale for the output layer
    // (its operand is kActScale h2), W3B = W3 / kActScale^2 for dh
    {
        const float* __restrict__ P = a.params;
        for (int i = tid; i < L::DP * H; i += blockDim.x) { const int o = i % H, k = i / H; wl[L::W1T + k * H + o] = k < D ? kTanhScale * P[off.w1 + o + k * H] : 0.0f; }
        for (int i = tid; i < H; i += blockDim.x) { wl[L::B1 + i] = kTanhScale * P[off.b1 + i]; wl[L::B2 + i] = (kTanhScale * kWScale * kActScale) * P[off.b2 + i]; }
        for (int i = tid; i < O * H; i += blockDim.x) { const int o = i % O, k = i / O; const float w3 = P[off.w3 + i]; wl[L::W3S + o * H + k] = w3 * (1.0f / kActScale); smem[SC::W3B + o * H + k] = w3 * (1.0f / (kActScale * kActScale)); }
        for (int i = tid; i < L::OP; i += blockDim.x) wl[L::B3 + i] = i < O ? P[off.b3 + i] : 0.0f;
    }
    for (int i = tid; i < NT * (D + 2) * kTS; i += blockDim.x) XI[i] = 0.0f;
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
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (HEAD == HEAD_GAUSSIAN) {
#pragma unroll
        for (int o = 0; o < O; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;
    const int tbase = wide_tr_base<H>(lane), tmbase = wide_trm_base<H>(lane);
    // gradient tiles are split as dz2 SG with SG = 2^(exponent of 1 / invB + 3): 4 ... 8 / invB, a power of two; every scale is undone exactly in the epilogue
    const float sg = __uint_as_float((((__float_as_uint(1.0f / a.invB) >> 23) & 0xffu) + 3u) << 23);
    const float inv_sg = 1.0f / sg, inv_sa = inv_sg * (1.0f / kActScale);
    GradArgs as = a; as.invB = a.invB * sg;                                            // what loss_head multiplies dLoss/dout with
    const float* W3B = smem + SC::W3B;

    f32x16 dW2[MT];                                                  // rows 32w.., all H columns
    float dW1a[D], db1a = 0.f, dW3a[O], db2a = 0.f, db3p[O], dlsp[O], st[5];   // per-lane partial sums: dW1a / db1a / db2a for unit 32w + (lane & 31) over the samples of this half-wave, dW3a for unit 32w + rowfn(lane & 15, h)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dW1a[d] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) { dW3a[o] = 0.f; db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = (int)(blockIdx.x % a.G);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;            // sample tiles of 32; the workgroup takes NT consecutive ones per pass (a missing last one is all-invalid)
    constexpr int KS = FirstLayer<D>::KS;                            // two first-layer k-steps for D <= 4, four for D <= 8 (Acrobot)
    int64_t tile = (int64_t)g * NT;
    const int64_t stride = (int64_t)a.G * NT;
    // wave t (< NT) is the loader of sample tile t: it holds the epoch-order entries of the NEXT pass's samples and requests their records in front of the dW2 stage — the
    // one stage without vector-memory instructions, under which the gather (and the entry load for the pass after) completes unseen
    TileIdx nidx; nidx.gidx = 0; nidx.inb = false;
    if (w < NT && tile < ntiles) {
        wide_request_records<KIND, HEAD>(a, tile_index(a, tile + w, ntiles, c), lane, RECS + w * RECT, VO + w * 64, VAL + w * 64);
        nidx = tile_index(a, tile + stride + w, ntiles, c);
    }
    __syncthreads();                                                 // (drains the LDS-DMA: the first pass's records are in place)
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += stride) {
        // ---- h1 tile w of every sample tile; its pieces into the workgroup images ----
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;      // stage-local lane coordinates: every LDS address below is a lane constant, and derived from values the optimiser can see through they are all hoisted out of the pass loop and held in registers (or spilled) for the whole kernel
            float xk[KS];                                                             // xk[s] = component 2s + h of sample c (zero beyond D: the records are zero-padded)
#pragma unroll
            for (int s = 0; s < KS; ++s) xk[s] = RECS[t * RECT + ((((2 * s) >> 2) * 32 + c) << 2) + ((2 * s) & 3) + h];
            f32x16 h1w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16_scaled<false>(h1w, 1.0f);                                          // kActScale h1
            // `opaque(lane)`: the image addresses are lane constants, and hoisted out of the tile loop as loop invariants they hold ~60 registers for the whole kernel (they cost 2-3 VALU to rebuild)
            store_tile_pieces2<H>(P1 + t * NTS, w, h1w, ln_);
            if (w == t) {                                                             // wave t keeps sample tile t's observations for the dW1 sums
#pragma unroll
                for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; XI[(t * (D + 2) + (d < D ? d : D + 1)) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the spare row
            }
        }
        STAMP(0);
        u32x4 afw[2][2];
        wide_split_preload(w2p, MT, w, lane, afw);                                    // first W2 fragments in flight across the barrier
        __syncthreads();                                                              // B1: P1, XI complete
        STAMP(1);
        // ---- h2 tile w ----
        f32x16 h2w[NT];
        dense_tile_split<H, NT, true, false>(w2p, wl + L::B2, P1, w, opaque(lane), afw, h2w);
#pragma unroll
        for (int t = 0; t < NT; ++t) tanh16_scaled<true>(h2w[t], 1.0f / (kWScale * kActScale));   // kActScale h2
        STAMP(2);
        // ---- output layer: partial over this wave's rows, summed across waves through LDS ----
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
                    p = fmaf(wv[0], h2w[t][4 * q + 0], p); p = fmaf(wv[1], h2w[t][4 * q + 1], p);
                    p = fmaf(wv[2], h2w[t][4 * q + 2], p); p = fmaf(wv[3], h2w[t][4 * q + 3], p);
                }
                p += __shfl_xor(p, 32);
                if (h == 0) PO[((t * MT + w) * O + o) * 32 + c] = p;
            }
        __syncthreads();                                                              // B2: PO complete
        STAMP(3);
        float dz[NT][O];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
            float out[O];
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float v = wl[L::B3 + o];
#pragma unroll
                for (int ww = 0; ww < MT; ++ww) v += PO[((t * MT + ww) * O + o) * 32 + c];  // fixed order: every wave gets the same bits
                out[o] = v;
            }
            // this lane's sample: the record's scalar quad {action bits, adv, logp_old, ret}, the old value, the validity word
            const f32x4 sc = *reinterpret_cast<const f32x4*>(RECS + t * RECT + (((RS - 1) * 32 + c) << 2));
            TileIn<O, KS> cur; cur.act = 0; cur.s0 = 0.f; cur.s1 = 0.f;
            const bool valid = VAL[t * 64 + c] != 0;
            if (HEAD == HEAD_VALUE) { cur.s0 = sc[3]; cur.s1 = a.has_clip_vf ? VO[t * 64 + c] : 0.f; }
            else {
                cur.s0 = sc[1]; cur.s1 = sc[2];
                if (HEAD == HEAD_CATEGORICAL) cur.act = __float_as_int(sc[0]) - a.action_start; else cur.xa[0] = sc[0];
            }
            loss_head<O, HEAD>(as, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz[t], st, dlsp);     // dz = SG dLoss/dout
#pragma unroll
            for (int o = 0; o < O; ++o) db3p[o] += (h == 0 && w == 0) ? dz[t][o] : 0.f;
        }
        // ---- dW3 (own rows): sum over the sample tiles in registers, then over the lanes (= samples) of each half-wave; lane l ends with unit 32w + rowfn(l & 15, h) ----
#pragma unroll
        for (int o = 0; o < O; ++o) {
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = h2w[0][r] * dz[0][o];
#pragma unroll
            for (int t = 1; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaf(h2w[t][r], dz[t][o], v[r]);
            dW3a[o] += half_reduce16_lane(v, opaque(lane));
        }
        // ---- dz2 tile w (in h2w's registers); its pieces into the workgroup images ----
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ln_ = opaque(lane), h = ln_ >> 5;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(W3B + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[t][o], dh[cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[t][4 * q + cc]; h2w[t][4 * q + cc] = dh[cc] * fmaf(-hv, hv, kActScale * kActScale); }   // = SG dz2
            }
            store_tile_pieces2<H>(P2 + t * NTS, w, h2w[t], ln_);
        }
        wide_split_preload(w2tp, MT, w, lane, afw);                                   // first W2' fragments in flight across the barrier
        STAMP(4);
        __syncthreads();                                                              // B3: P2 complete
        STAMP(5);
        // ---- dh1' tile w = (W2' dz2)', transposed: lane = unit 32w + (lane & 31), register r = sample rowfn(r, h); dz1' ----
        f32x16 g1[NT];
        dense_tile_split<H, NT, false, true>(w2tp, nullptr, P2, w, opaque(lane), afw, g1);
        STAMP(6);
        // loader waves: next pass's records (LDS-DMA), the pass after next's epoch-order entries — requested in front of the two stages that issue no vector-memory
        // instruction (dz1 / dW1, then the dW2 product: ~11 k cycles): vmcnt retires in order, so a gather in front of a streaming chain holds that chain's first fragment wait
        // for the whole gather latency.  (NOT between the sched_barrier below and the dW2 stage: a branch there costs the register allocator ~200 spilled registers.)
        if (w < NT) {
            const int ln_ = opaque(lane);
            wide_request_records<KIND, HEAD>(a, nidx, ln_, RECS + w * RECT, VO + w * 64, VAL + w * 64);
            nidx = tile_index(a, tile + 2 * stride + w, ntiles, ln_ & 31);
        }
        // ---- dz1', then dW1 | db1 as per-lane sums over the lane's samples (before dW2, so that dz1 is dead while the 128 accumulators are being updated) ----
        // Eight groups (sample tile t, register group q = samples 8q + 4h + {0..3} of the lane's unit): two transposed reads (the h1 pieces) and D broadcast reads (x) each.
        // The loads of group g + 1 are requested before group g is computed (two named buffers) — left to itself the allocator, short of registers here, gave every read
        // the same four registers and a full LDS wait (round-5 stamps: 4.3 k cycles for ~300 vector instructions)
        {
            constexpr float c0 = 1.0f / kWScale, c1 = c0 / (kActScale * kActScale);             // g1 = (SG dz2 . kWScale W2) (1 - h1^2) / kWScale = SG dz1
            const int ln_ = opaque(lane), h = ln_ >> 5;
            const int tmw = opaque(tmbase) ^ (64 * w), tmw32 = tmw ^ 32;
            const float* xrow = XI + 4 * h;
            u32x2 hp[2][2]; f32x4 xq[2][D];
#define S3_LOAD(B, T, Q) { const int a_ = (((Q) & 1) ? tmw32 : tmw) + 8 * (Q) * RB; \
                           hp[B][0] = __builtin_bit_cast(u32x2, lds_read_tr16(P1 + (T) * NTS, a_)); hp[B][1] = __builtin_bit_cast(u32x2, lds_read_tr16(P1 + (T) * NTS, a_ + PS)); \
                           _Pragma("unroll") for (int d = 0; d < D; ++d) xq[B][d] = *reinterpret_cast<const f32x4*>(xrow + ((T) * (D + 2) + d) * kTS + 8 * (Q)); }
#define S3_COMP(B, T, Q) { float hv[4]; pieces_sum2(hp[B][0].x, hp[B][1].x, hv[0], hv[1]); pieces_sum2(hp[B][0].y, hp[B][1].y, hv[2], hv[3]); \
                           _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float t2 = hv[i] * hv[i]; const float gz = g1[T][4 * (Q) + i] * fmaf(-t2, c1, c0); db1a += gz; \
                               _Pragma("unroll") for (int d = 0; d < D; ++d) dW1a[d] = fmaf(gz, xq[B][d][i], dW1a[d]); } }
            S3_LOAD(0, 0, 0)
#pragma unroll
            for (int gq = 0; gq < 4 * NT; ++gq) {
                const int t = gq >> 2, q = gq & 3, tn = (gq + 1) >> 2, qn = (gq + 1) & 3;
                if (gq & 1) { if (gq + 1 < 4 * NT) S3_LOAD(0, tn, qn) } else { S3_LOAD(1, tn < NT ? tn : NT - 1, qn) }
                __builtin_amdgcn_sched_barrier(0);
                if (gq & 1) S3_COMP(1, t, q) else S3_COMP(0, t, q)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef S3_LOAD
#undef S3_COMP
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(7);
        // ---- dW2[rows of w][:] += dz2 h1' (both operands as transposed fragments of the piece images; k = the NT x 32 samples); db2 from the dz2 fragments ----
        {
            const int tb = opaque(tbase), tbw = tb ^ (64 * w), tbw16 = tbw ^ 16;
            f16x8 Az[2 * NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        Az[2 * t + s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P2 + t * NTS, tbw, tbw16, p, s));
                        db2a = frag_sum8(Az[2 * t + s][p], db2a);
                    }
            // the h1 fragments of a step (m-tile mj, sample tile t) are requested one step ahead of its MFMAs, in two named register sets (BhA: t = 0, BhB: t = 1)
            static_assert(NT == 2, "the dW2 stage alternates two fragment sets");
            f16x8 BhA[2][2], BhB[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) BhA[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1, tb, tb ^ 16, p, s));      // m-tile 0
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                const int tbj = tb ^ (64 * mj), tbj16 = tbj ^ 16;
                const int mn = mj + 1 < MT ? mj + 1 : mj, tbn = tb ^ (64 * mn), tbn16 = tbn ^ 16;       // (the last step requests its own fragments again: in bounds, unused)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhB[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1 + NTS, tbj, tbj16, p, s));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split3(Az[s][0], Az[s][1], BhA[s][0], BhA[s][1], dW2[mj]);            // (SG dz2)(kActScale h1)', sample tile 0
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhA[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1, tbn, tbn16, p, s));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split3(Az[2 + s][0], Az[2 + s][1], BhB[s][0], BhB[s][1], dW2[mj]);    // sample tile 1
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(8);
        __syncthreads();                                                              // B4: P1 / P2 / PO / XI free for the next pass; the next pass's records have landed (the barrier's fence drains the DMA)
        STAMP(9);
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
        const int64_t first = (int64_t)g * NT;
#ifdef DRIL_STAMPS_HI                                                                  // (the upper half of the workgroup's waves instead: the younger wave of every SIMD)
        if (w >= MT / 2)
#else
        if (w < 4)
#endif
        { for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k]; o_[10] = (unsigned long long)(first < ntiles ? (ntiles - first + stride - 1) / stride : 0); o_[11] = HEAD; }
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
        for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r] * inv_sa;
#pragma unroll
    for (int d = 0; d < D; ++d) { const float v = (dW1a[d] + __shfl_xor(dW1a[d], 32)) * inv_sg; if (h == 0) slab[o_w1 + 32 * w + c + d * H] = v; }
    { const float b1 = (db1a + __shfl_xor(db1a, 32)) * inv_sg; if (h == 0) slab[o_b1 + 32 * w + c] = b1; }
    { const float b2 = (db2a + __shfl_xor(db2a, 32)) * inv_sg; if (h == 0) slab[o_b2 + 32 * w + c] = b2; }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        if ((lane & 16) == 0) slab[o_w3 + o + (32 * w + rowfn(lane & 15, h)) * O] = dW3a[o] * inv_sa;
        const float b3 = half_sum(db3p[o]) * inv_sg;
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]) * inv_sg; if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
}

template <int KIND, int H>
__global__ __launch_bounds__(H * 2, 2) void ppo_grad_wide_split_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body_wide_split<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN>(a, smem);
    else grad_body_wide_split<KIND, H, 1, HEAD_VALUE>(a, smem);
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
    if (a.variant && a.rec) {      // bf16 matrix cores
#define CALLWS(K, HH)                                                                                         \
    {                                                                                                         \
        const size_t lds = grad_wide_split_lds_bytes<K, HH>();                                                \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_grad_wide_split_kernel<K, HH>, lds); if (e != hipSuccess) return e; } \
        ppo_grad_wide_split_kernel<K, HH><<<2 * a.G, HH * 2, lds, s>>>(a);                                    \
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

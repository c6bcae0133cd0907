// dril_grad_wide_split.h — ppo_grad_wide_split_kernel: the update kernel of hidden [128,128] / [256,256] on the f16 matrix cores (fp32-equivalent two-piece operand split).
// Included by dril_grad_wide.hip.  MW = m-tiles per wave: 1 in the product (H/32 waves of 256 registers, two per SIMD at H = 256); MW = 2 (four waves of 512 registers,
// one per SIMD, dW2 in AGPRs) was built and measured in round 5 — parity-green, the same speed to 0.5 % (profiles/r05_wide_split.md) — and is not instantiated.
#pragma once
#include "dril_grad_common.h"
#include "dril_split_pieces.h"

namespace dril {

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
//   * the dW2 product runs in the v_mfma_f32_16x16x32_f16 shape (end of round 5): same pipe cycles and LDS reads as 32x32x16, but the chip sustains a higher clock on it
//     (tools/micro/mfma_shape_f16.hip: 1.93 against 1.67 GHz in a bare loop) — this kernel is the clock-limited one (1.90 GHz): - 1 % per launch.
//   * no AGPRs: at two waves per SIMD the allocator gives a function that uses ANY AGPR only 128 VGPRs; the 128 dW2 accumulators are VGPR-form MFMA results like the rest.
// LDS at H = 256: two piece images of 64 KB + ~12 KB of small parameters = 143 - 156 KB of the CU's 160 (one workgroup per CU, two waves per SIMD); H = 128: 72 KB, two workgroups per CU.
// =============================================================================================
template <int D, int H, int O, int NT> struct WideSplitScratch {
    static constexpr int MT = H / 32;
    static constexpr int SMALL = NetLdsSmall<D, H, O>::END;
    static constexpr int P1 = (SMALL + 3) / 4 * 4;          // h1 pieces: NT x 2 x 32 x H f16 = NT 32 H floats
    static constexpr int P2 = P1 + NT * 32 * H;             // dz2 pieces
    static constexpr int XI = P2 + NT * 32 * H;             // [NT][D+2][kTS]: the tile's observations, component-major (row D + 1 takes the padding components' writes)
    static constexpr int PO = XI + NT * (D + 2) * kTS;      // [NT][waves <= MT][O][32] output-layer partial sums
    static constexpr int W3B = PO + NT * MT * O * 32;       // W3 / kActScale^2 (dh); the staged W3S is W3 / kActScale (output layer on kActScale h2)
    static constexpr int RQ = RecLayout<D>::RS == 3 ? 4 : 2; // record quads kept per sample tile (three-quad records: the second DMA's upper half-wave lands in a fourth)
    static constexpr int REC = W3B + O * H;                 // [NT][RQ][32] float4: the pass's minibatch records, quad-major (wide_request_records)
    static constexpr int VO = REC + NT * RQ * 32 * 4;       // [NT][64] old values (critic with a value clip)
    static constexpr int VAL = VO + NT * 64;                // [NT][64] validity words
    static constexpr int SIZE = VAL + NT * 64;
    static_assert(REC % 4 == 0, "records: 16-byte aligned");
    static_assert(SIZE * 4 <= 160 * 1024, "ppo_grad_wide_split_kernel: LDS");
};
__device__ __forceinline__ void wide_split_preload(const u32x4* __restrict__ wimg, int MTv, int mo, int lane, u32x4 (&af)[2][2]) {
    const u32x4* base = wimg + ((size_t)mo * MTv * 4) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[s][p] = base[(size_t)(s * 2 + p) * 64];
}
// output m-tiles mo0 .. mo0 + MW - 1 of Y = W X for NT sample tiles at once: W as pre-split fragments from L2 (af arrives preloaded with the first k-tile's, each refilled in place
// right after its MFMAs), X from the piece images (one per sample tile, NTS bytes apart; one read feeds the MW m-tiles).  TRANSPOSED: the two operands in the other order — the
// accumulator then holds Y' (lane = unit 32 mo + (lane & 31), register r = sample rowfn(r, lane >> 5))
template <int H, int NT, int MW, bool BIAS, bool TRANSPOSED>
__device__ __forceinline__ void dense_tile_split(const u32x4* __restrict__ wimg, const float* __restrict__ bias, const char* pimg, int mo0, int lane, u32x4 (&af)[MW][2][2], f32x16 (&acc)[MW][NT]) {
    constexpr int MT = H / 32, RB = 2 * H, PS = 32 * RB, NTS = 2 * PS;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB, gsw = wimg_g<H>(c);
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * (mo0 + m) + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[m][t][4 * q + 0] = b[0]; acc[m][t][4 * q + 1] = b[1]; acc[m][t][4 * q + 2] = b[2]; acc[m][t][4 * q + 3] = b[3]; }
        }
    const u32x4* base = wimg + ((size_t)mo0 * MT * 4) * 64 + lane;                   // m-tile mo0 + m: + m MT 4 64
    // the activation fragments of a k16 step are requested one step AHEAD of the MFMAs that use them (two named register sets, X0 for the even steps and X1 for the odd ones:
    // no copies): read-then-wait-then-MFMA left every step's MFMAs behind an LDS round trip
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
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const f16x8 W0 = __builtin_bit_cast(f16x8, af[m][0][0]), W1 = __builtin_bit_cast(f16x8, af[m][0][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = TRANSPOSED ? mfma_split3(X0[t][0], X0[t][1], W0, W1, acc[m][t]) : mfma_split3(W0, W1, X0[t][0], X0[t][1], acc[m][t]);
#pragma unroll
            for (int p = 0; p < 2; ++p) af[m][0][p] = nextp[(size_t)(m * MT * 4 + p) * 64];
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
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const f16x8 W0 = __builtin_bit_cast(f16x8, af[m][1][0]), W1 = __builtin_bit_cast(f16x8, af[m][1][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = TRANSPOSED ? mfma_split3(X1[t][0], X1[t][1], W0, W1, acc[m][t]) : mfma_split3(W0, W1, X1[t][0], X1[t][1], acc[m][t]);
#pragma unroll
            for (int p = 0; p < 2; ++p) af[m][1][p] = nextp[(size_t)(m * MT * 4 + 2 + p) * 64];
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

#ifndef DRIL_WIDE_DW2_16
#define DRIL_WIDE_DW2_16 1
#endif
// ---- the dW2 product in the v_mfma_f32_16x16x32_f16 shape (-DDRIL_WIDE_DW2_16=0: the 32x32x16 form it replaced; profiles/r05_mfma_shape_f16_microbench.md, r05_wide_split.md §9) ----
// one k32 step of the two-piece product on a 16 x 16 tile, small terms first (mfma_split3's order)
__device__ __forceinline__ f32x4 mfma16_split3(f16x8 Ah, f16x8 Al, f16x8 Bh, f16x8 Bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, acc, 0, 0, 0);
}
// operand of a 16x16x32 product that sums over the 32 SAMPLES of a tile: lane (unit 16 f + (lane & 15) of m-tile m, k-block kb = lane >> 4) gets samples 8 kb + j of its unit.
// The 16-lane group kb addresses rows (samples) 8 kb + q (+ 4 for the second read), q = e >> 2, and columns (units) 4 (e & 3) .. + 3 of the 16; address = base ^ (64 m) ^ (32 f),
// second read (base ^ 16) + 4 RB — g(n + 4) = g(n) ^ 1 for these rows.  A half-wave covers rows {0-3, 8-11} or {16-19, 24-27}: 16 distinct chunks x 16 bytes = all 64 banks once
template <int H>
__device__ __forceinline__ int wide_tr16_base(int lane) {
    static_assert(H >= 128, "wide_tr16_base: the four-bit chunk swizzle");
    constexpr int RB = 2 * H;
    const int kb = lane >> 4, e = lane & 15, q = e >> 2, p = e & 3, n = 8 * kb + q;
    return n * RB + ((((p >> 1) ^ wimg_g<H>(n)) & 15) << 4) + 8 * (p & 1);
}
template <int H>
__device__ __forceinline__ f16x8 load_frag16_T(const char* pimg, int t, int piece) {     // t = base ^ (64 m) ^ (32 f)
    constexpr int RB = 2 * H, PS = 32 * RB;
    return __builtin_bit_cast(f16x8, frag8(lds_read_tr16(pimg, t + piece * PS), lds_read_tr16(pimg, (t ^ 16) + piece * PS + 4 * RB)));
}

template <int KIND, int H, int O, int HEAD, int MW>
__device__ __forceinline__ void grad_body_wide_split(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32, NT = kWideSplitNT, NW = MT / MW;   // NW waves, each owning MW consecutive m-tiles of every layer
    static_assert(MT % MW == 0 && NW >= NT, "waves: one loader per sample tile");
    constexpr int RB = 2 * H, PS = 32 * RB, NTS = 2 * PS;            // bytes of an image row, of one piece of a sample tile, of a sample tile's image
    constexpr bool REC = true;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideSplitScratch<D, H, O, NT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave: m-tiles MW w .. MW w + MW - 1
    const int mw0 = MW * w;
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

    f32x16 dW2[MW][MT];                                              // rows of the own m-tiles, all H columns
    float db2a16[2] = {0.f, 0.f};                                    // (DRIL_WIDE_DW2_16: db2 partials for units 16 fa + (lane & 15) of the own m-tile, over the lane's k-block)
    float dW1a[MW][D], db1a[MW], dW3a[MW][O], db2a[MW], db3p[O], dlsp[O], st[5];   // per-lane partial sums: dW1a / db1a / db2a for unit 32 (mw0 + m) + (lane & 31) over the samples of this half-wave, dW3a for unit 32 (mw0 + m) + rowfn(lane & 15, h)
#pragma unroll
    for (int m = 0; m < MW; ++m) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2[m][j][r] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) dW1a[m][d] = 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) dW3a[m][o] = 0.f;
        db1a[m] = 0.f; db2a[m] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) { db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = (int)(blockIdx.x % a.G);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;            // sample tiles of 32; the workgroup takes NT consecutive ones per pass (a missing last one is all-invalid)
    constexpr int KS = FirstLayer<D>::KS;                            // two first-layer k-steps for D <= 4, four for D <= 8 (Acrobot)
    int64_t tile = (int64_t)g * NT;
    const int64_t stride = (int64_t)a.G * NT;
    // wave t (< NT) is the loader of sample tile t: it holds the epoch-order entries of the NEXT pass's samples and requests their records in front of the dW2 stage — the
    // one stage without vector-memory instructions, under which the gather (and the entry load for the pass after) completes unseen
    TileIdx nidx; nidx.gidx = 0; nidx.g32 = 0; nidx.is32 = false; nidx.inb = false;
    if (w < NT && tile < ntiles) {
        request_records_lds<KIND, HEAD>(a, tile_index(a, tile + w, ntiles, c), lane, RECS + w * RECT, VO + w * 64, VAL + w * 64);
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
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int mt = mw0 + m;
                f32x16 h1w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * mt + 8 * q + 4 * h);
                    h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * mt + c], xk[s], h1w);
                tanh16_scaled<false>(h1w, 1.0f);                                      // kActScale h1
                store_tile_pieces2<H>(P1 + t * NTS, mt, h1w, ln_);
            }
            if (w == t) {                                                             // wave t keeps sample tile t's observations for the dW1 sums
#pragma unroll
                for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; XI[(t * (D + 2) + (d < D ? d : D + 1)) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the spare row
            }
        }
        STAMP(0);
        u32x4 afw[MW][2][2];
#pragma unroll
        for (int m = 0; m < MW; ++m) wide_split_preload(w2p, MT, mw0 + m, lane, afw[m]);   // first W2 fragments in flight across the barrier
        __syncthreads();                                                              // B1: P1, XI complete
        STAMP(1);
        // ---- h2 tile w ----
        f32x16 h2w[MW][NT];
        dense_tile_split<H, NT, MW, true, false>(w2p, wl + L::B2, P1, mw0, opaque(lane), afw, h2w);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) tanh16_scaled<true>(h2w[m][t], 1.0f / (kWScale * kActScale));   // kActScale h2
        STAMP(2);
        // ---- output layer: partial over this wave's rows, summed across waves through LDS ----
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
                float p = 0.f;
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * (mw0 + m) + 8 * q + 4 * h);
                        p = fmaf(wv[0], h2w[m][t][4 * q + 0], p); p = fmaf(wv[1], h2w[m][t][4 * q + 1], p);
                        p = fmaf(wv[2], h2w[m][t][4 * q + 2], p); p = fmaf(wv[3], h2w[m][t][4 * q + 3], p);
                    }
                p += __shfl_xor(p, 32);
                if (h == 0) PO[((t * NW + w) * O + o) * 32 + c] = p;
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
                for (int ww = 0; ww < NW; ++ww) v += PO[((t * NW + ww) * O + o) * 32 + c];  // fixed order: every wave gets the same bits
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
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int o = 0; o < O; ++o) {
                f32x16 v;
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = h2w[m][0][r] * dz[0][o];
#pragma unroll
                for (int t = 1; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaf(h2w[m][t][r], dz[t][o], v[r]);
                dW3a[m][o] += half_reduce16_lane(v, opaque(lane));
            }
        // ---- dz2 tile w (in h2w's registers); its pieces into the workgroup images ----
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int ln_ = opaque(lane), h = ln_ >> 5, mt = mw0 + m;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int o = 0; o < O; ++o) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(W3B + o * H + 32 * mt + 8 * q + 4 * h);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[t][o], dh[cc]);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[m][t][4 * q + cc]; h2w[m][t][4 * q + cc] = dh[cc] * fmaf(-hv, hv, kActScale * kActScale); }   // = SG dz2
                }
                store_tile_pieces2<H>(P2 + t * NTS, mt, h2w[m][t], ln_);
            }
#pragma unroll
        for (int m = 0; m < MW; ++m) wide_split_preload(w2tp, MT, mw0 + m, lane, afw[m]);   // first W2' fragments in flight across the barrier
        STAMP(4);
        __syncthreads();                                                              // B3: P2 complete
        STAMP(5);
        // ---- dh1' tile w = (W2' dz2)', transposed: lane = unit 32w + (lane & 31), register r = sample rowfn(r, h); dz1' ----
        f32x16 g1[MW][NT];
        dense_tile_split<H, NT, MW, false, true>(w2tp, nullptr, P2, mw0, opaque(lane), afw, g1);
        STAMP(6);
        // loader waves: next pass's records (LDS-DMA), the pass after next's epoch-order entries — requested in front of the two stages that issue no vector-memory
        // instruction (dz1 / dW1, then the dW2 product: ~11 k cycles): vmcnt retires in order, so a gather in front of a streaming chain holds that chain's first fragment wait
        // for the whole gather latency.  (NOT between the sched_barrier below and the dW2 stage: a branch there costs the register allocator ~200 spilled registers.)
        if (w < NT) {
            const int ln_ = opaque(lane);
            request_records_lds<KIND, HEAD>(a, nidx, ln_, RECS + w * RECT, VO + w * 64, VAL + w * 64);
            nidx = tile_index(a, tile + 2 * stride + w, ntiles, ln_ & 31);
        }
        // ---- dz1', then dW1 | db1 as per-lane sums over the lane's samples (before dW2, so that dz1 is dead while the 128 accumulators are being updated) ----
        // Eight groups (sample tile t, register group q = samples 8q + 4h + {0..3} of the lane's unit): two transposed reads (the h1 pieces) and D broadcast reads (x) each.
        // The loads of group g + 1 are requested before group g is computed (two named buffers) — left to itself the allocator, short of registers here, gave every read
        // the same four registers and a full LDS wait (round-5 stamps: 4.3 k cycles for ~300 vector instructions)
        {
            constexpr float c0 = 1.0f / kWScale, c1 = c0 / (kActScale * kActScale);             // g1 = (SG dz2 . kWScale W2) (1 - h1^2) / kWScale = SG dz1
            const int ln_ = opaque(lane), h = ln_ >> 5;
            const int tm0 = opaque(tmbase) ^ (64 * mw0);                                      // m-tile mw0 + m: ^ 64 m (mw0 is a multiple of MW)
            const float* xrow = XI + 4 * h;
            u32x2 hp[2][MW][2]; f32x4 xq[2][D];                                               // one set of x reads feeds the MW m-tiles
#define S3_LOAD(B, T, Q) { _Pragma("unroll") for (int m = 0; m < MW; ++m) { const int a_ = (tm0 ^ (64 * m) ^ (((Q) & 1) ? 32 : 0)) + 8 * (Q) * RB; \
                               hp[B][m][0] = __builtin_bit_cast(u32x2, lds_read_tr16(P1 + (T) * NTS, a_)); hp[B][m][1] = __builtin_bit_cast(u32x2, lds_read_tr16(P1 + (T) * NTS, a_ + PS)); } \
                           _Pragma("unroll") for (int d = 0; d < D; ++d) xq[B][d] = *reinterpret_cast<const f32x4*>(xrow + ((T) * (D + 2) + d) * kTS + 8 * (Q)); }
#define S3_COMP(B, T, Q) { _Pragma("unroll") for (int m = 0; m < MW; ++m) { float hv[4]; pieces_sum2(hp[B][m][0].x, hp[B][m][1].x, hv[0], hv[1]); pieces_sum2(hp[B][m][0].y, hp[B][m][1].y, hv[2], hv[3]); \
                           _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float t2 = hv[i] * hv[i]; const float gz = g1[m][T][4 * (Q) + i] * fmaf(-t2, c1, c0); db1a[m] += gz; \
                               _Pragma("unroll") for (int d = 0; d < D; ++d) dW1a[m][d] = fmaf(gz, xq[B][d][i], dW1a[m][d]); } } }
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
#if DRIL_WIDE_DW2_16
        // ---- dW2 in the 16x16x32 shape: per (m-tile mj, sample tile t) four 16 x 16 tiles (fa, fb), one k32 step each; dW2[m][mj] holds them as registers 4 (2 fa + fb) .. + 3;
        // db2 from the dz2 fragments (per lane: unit 16 fa + (lane & 15), the 8 samples of its k-block) ----
        {
            static_assert(MW == 1 && NT == 2, "16x16x32 dW2: one m-tile per wave, two sample tiles");
            const int t16 = opaque(wide_tr16_base<H>(lane)), tw = t16 ^ (64 * mw0);
            f16x8 Az[NT][2][2];                                                        // [tile][fa][piece]
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        Az[t][fa][p] = load_frag16_T<H>(P2 + t * NTS, tw ^ (32 * fa), p);
                        db2a16[fa] = frag_sum8(Az[t][fa][p], db2a16[fa]);
                    }
            f16x8 BhA[2][2], BhB[2][2];                                                // [fb][piece]: sample tile 0 | 1 of the m-tile in flight
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int p = 0; p < 2; ++p) BhA[fb][p] = load_frag16_T<H>(P1, t16 ^ (32 * fb), p);
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                const int tj = t16 ^ (64 * mj), mn = mj + 1 < MT ? mj + 1 : mj, tn = t16 ^ (64 * mn);
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhB[fb][p] = load_frag16_T<H>(P1 + NTS, tj ^ (32 * fb), p);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        f32x4 acc = {dW2[0][mj][4 * (2 * fa + fb)], dW2[0][mj][4 * (2 * fa + fb) + 1], dW2[0][mj][4 * (2 * fa + fb) + 2], dW2[0][mj][4 * (2 * fa + fb) + 3]};
                        acc = mfma16_split3(Az[0][fa][0], Az[0][fa][1], BhA[fb][0], BhA[fb][1], acc);
                        dW2[0][mj][4 * (2 * fa + fb)] = acc[0]; dW2[0][mj][4 * (2 * fa + fb) + 1] = acc[1]; dW2[0][mj][4 * (2 * fa + fb) + 2] = acc[2]; dW2[0][mj][4 * (2 * fa + fb) + 3] = acc[3];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhA[fb][p] = load_frag16_T<H>(P1, tn ^ (32 * fb), p);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        f32x4 acc = {dW2[0][mj][4 * (2 * fa + fb)], dW2[0][mj][4 * (2 * fa + fb) + 1], dW2[0][mj][4 * (2 * fa + fb) + 2], dW2[0][mj][4 * (2 * fa + fb) + 3]};
                        acc = mfma16_split3(Az[1][fa][0], Az[1][fa][1], BhB[fb][0], BhB[fb][1], acc);
                        dW2[0][mj][4 * (2 * fa + fb)] = acc[0]; dW2[0][mj][4 * (2 * fa + fb) + 1] = acc[1]; dW2[0][mj][4 * (2 * fa + fb) + 2] = acc[2]; dW2[0][mj][4 * (2 * fa + fb) + 3] = acc[3];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        // ---- dW2[rows of w][:] += dz2 h1' (both operands as transposed fragments of the piece images; k = the NT x 32 samples); db2 from the dz2 fragments ----
        {
            const int tb = opaque(tbase);
            f16x8 Az[MW][2 * NT][2];                                                  // the own m-tiles' dz2 fragments: held for the whole stage
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int tbw = tb ^ (64 * (mw0 + m)), tbw16 = tbw ^ 16;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            Az[m][2 * t + s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P2 + t * NTS, tbw, tbw16, p, s));
                            db2a[m] = frag_sum8(Az[m][2 * t + s][p], db2a[m]);
                        }
            }
            // the h1 fragments of a step (m-tile mj, sample tile t) are requested one step ahead of its MFMAs, in two named register sets (BhA: t = 0, BhB: t = 1); one set
            // feeds the MW own m-tiles
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
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int s = 0; s < 2; ++s) dW2[m][mj] = mfma_split3(Az[m][s][0], Az[m][s][1], BhA[s][0], BhA[s][1], dW2[m][mj]);            // (SG dz2)(kActScale h1)', sample tile 0
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhA[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1, tbn, tbn16, p, s));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int s = 0; s < 2; ++s) dW2[m][mj] = mfma_split3(Az[m][2 + s][0], Az[m][2 + s][1], BhB[s][0], BhB[s][1], dW2[m][mj]);    // sample tile 1
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        STAMP(8);
        __syncthreads();                                                              // B4: P1 / P2 / PO / XI free for the next pass; the next pass's records have landed (the barrier's fence drains the DMA)
        STAMP(9);
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
        const int64_t first = (int64_t)g * NT;
#ifdef DRIL_STAMPS_HI                                                                  // (the upper half of the workgroup's waves instead: the younger wave of every SIMD)
        if (w >= NW / 2 && w < NW / 2 + 4)
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
    for (int m = 0; m < MW; ++m) {
        const int mt = mw0 + m;
#if DRIL_WIDE_DW2_16
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int fa = 0; fa < 2; ++fa)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)       // tile (fa, fb): lane = column 16 fb + (lane & 15), registers = rows 16 fa + 4 (lane >> 4) + r
                        slab[o_w2 + 32 * mt + 16 * fa + 4 * (lane >> 4) + r + (32 * mj + 16 * fb + (lane & 15)) * H] = dW2[m][mj][4 * (2 * fa + fb) + r] * inv_sa;
#else
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * mt + rowfn(r, h) + (32 * mj + c) * H] = dW2[m][mj][r] * inv_sa;
#endif
#pragma unroll
        for (int d = 0; d < D; ++d) { const float v = (dW1a[m][d] + __shfl_xor(dW1a[m][d], 32)) * inv_sg; if (h == 0) slab[o_w1 + 32 * mt + c + d * H] = v; }
        { const float b1 = (db1a[m] + __shfl_xor(db1a[m], 32)) * inv_sg; if (h == 0) slab[o_b1 + 32 * mt + c] = b1; }
#if DRIL_WIDE_DW2_16
#pragma unroll
        for (int fa = 0; fa < 2; ++fa) {                                               // the four k-blocks of a unit sit in lanes l, l ^ 16, l ^ 32, l ^ 48
            float v = db2a16[fa]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane < 16) slab[o_b2 + 32 * mt + 16 * fa + lane] = v * inv_sg;
        }
#else
        { const float b2 = (db2a[m] + __shfl_xor(db2a[m], 32)) * inv_sg; if (h == 0) slab[o_b2 + 32 * mt + c] = b2; }
#endif
#pragma unroll
        for (int o = 0; o < O; ++o) if ((lane & 16) == 0) slab[o_w3 + o + (32 * mt + rowfn(lane & 15, h)) * O] = dW3a[m][o] * inv_sa;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float b3 = half_sum(db3p[o]) * inv_sg;
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]) * inv_sg; if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
}

template <int KIND, int H, int MW>
__global__ __launch_bounds__(H * 2 / MW, 2 / MW) void ppo_grad_wide_split_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body_wide_split<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, MW>(a, smem);
    else grad_body_wide_split<KIND, H, 1, HEAD_VALUE, MW>(a, smem);
}

}  // namespace dril

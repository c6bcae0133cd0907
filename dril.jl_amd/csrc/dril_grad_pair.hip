// dril_grad_pair.hip — ppo_grad_pair_kernel: the update kernel of hidden [64,64] for large minibatches (THE HEADLINE KERNEL): f16 matrix cores (v_mfma_f32_32x32x16_f16),
// fp32-equivalent two-piece operand split (three MFMAs per k16 step; dril_device.h), two waves per tile
#include <utility>

#include "dril_grad_common.h"
#include "dril_split_pieces.h"

namespace dril {

// =============================================================================================
// ppo_grad_pair_kernel — hidden [64,64], large minibatches: TWO waves own one 32-sample tile (wave p the m-tile p of every layer and the 32 x 64 slice p of dW2: the
// decomposition of ppo_grad_wide_split_kernel at MT = 2), two pairs per workgroup, two workgroups per CU = TWO waves per SIMD at <= 256 registers: the second wave
// covers the first one's LDS round trips, barriers and dependency stalls.  (It does NOT hide the vector work behind the MFMAs, as round 2 believed: matrix-pipe time and
// vector-ALU time add on a SIMD, also across waves — profiles/r03_pair_kernel_notes.md.)  Arithmetic: every f32 operand of the H x H contractions as two f16 pieces
// (hi + lo, scaled by powers of two into f16's dense range), products hi.hi + hi.lo + lo.hi accumulated in f32 (2^-24 relative; profiles/r03_split_arith.md section 6).
//   * W2 lives in LDS once per workgroup as two f16 pieces in the piece-image layout of the wide split kernel (128-byte rows, 16-byte chunk ch of row r at ch ^ f(r)):
//     row reads (ds_read_b128) give the A operand of L2, ds_read_b64_tr_b16 the A operand of dh1 (W2'); pre-scaled by kTanhScale, dh1 folds 1 / kTanhScale into its mask.
//   * each pair has two 12 KB piece images (h1, dz2): every wave writes its own 32 columns once; row reads give the B operand of L2 / dh1 (both m-tiles), transposed
//     reads both operands of dW2.
//   * no f32 image at all: dW3, db2, dW1 and db1 are per-lane accumulations (the lane is the sample), reduced over the 32 lanes of a half once, in the epilogue.
//   * the minibatch records reach the pair by LDS-DMA (round 5: request_records_lds, dril_grad_common.h): wave 0 of the pair requests the NEXT tile's records after barrier B1
//     into the other of two buffers — one gather per pair instead of one per wave into registers (the traffic counters had the records read twice), no index arithmetic and no
//     half-wave exchange in the second wave; barriers B1 - B3 order LDS only (the DMA stays in flight), B4 drains it.  0.996 - 1.002 -> 0.982 ms per launch (same box).
//   * four workgroup barriers per tile; both pairs of a workgroup run the same number of tiles (the second pair's last tile may be an all-invalid one).
// Every wave owns distinct rows of every gradient; the two pairs of a workgroup are summed through LDS into ONE slab per workgroup (epilogue).  a.G / a.Gc = pairs of
// the actor / the critic (even); grid = (a.G + a.Gc) / 2 workgroups, the first a.G / 2 run the actor; slabs: a.G / 2 of the actor, a.Gc / 2 of the critic.
// =============================================================================================
// Every image starts at a multiple of 512 bytes of LDS (the dynamic segment itself is declared 1 KB-aligned).  Then bits 4-6 of a swizzled address ARE the chunk field
// chunk ^ g(row), and stepping the chunk by a constant is an XOR of the whole address with that constant — one VALU per access, the image and piece offsets going into
// the instruction's immediate, instead of xor / shift / add / add (round 3: - 60 VALU per wave and tile)
template <int D, int O> struct PairLds {
    static constexpr int H = 64, DP = FirstLayer<D>::DP, OP = (O + 3) / 4 * 4;
    static constexpr bool WIDE_IN = D > 4;                   // observations of 5 .. 8 components: dW1 / db1 on the matrix cores (below), four first-layer k-steps
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, W3B = W3S + O * H, B3 = W3B + O * H, SMALL_END = (B3 + OP + 127) / 128 * 128;   // W3S = W3 / kActScale (forward), W3B = W3 / kActScale^2 (dh)
    static constexpr int WIMG = SMALL_END;                    // two f16 pieces x [64 out][64 in] = 2 x 8192 bytes
    static constexpr int PAIR0 = WIMG + 2 * 2048;
    // per pair: two 8 KB piece images (h1, dz2), [2 waves][O][32] partial sums; D > 4: a third piece image (dz1, 8 KB) and the observation image (two f16 pieces x [32 samples][32 columns])
    static constexpr int P1 = 0, P2 = P1 + 2 * 1024, P3 = P2 + 2 * 1024, XI = P3 + (WIDE_IN ? 2 * 1024 : 0), PO = XI + (WIDE_IN ? 2 * 512 : 0);
    // the pair's minibatch records by LDS-DMA (request_records_lds, dril_grad_common.h): two buffers (the next tile's records land while this tile's are read), quad-major
    static constexpr int RQ = RecLayout<D>::RS == 3 ? 4 : 2, RECB = RQ * 32 * 4;            // floats of one buffer
    static constexpr int REC = (PO + 2 * O * 32 + 3) / 4 * 4, VO = REC + 2 * RECB, VAL = VO + 2 * 64, PAIR_SIZE = (VAL + 2 * 64 + 127) / 128 * 128;
    static constexpr int END = PAIR0 + 2 * PAIR_SIZE;
    static_assert((4 * WIMG) % 512 == 0 && (4 * PAIR0) % 512 == 0 && (4 * PAIR_SIZE) % 512 == 0 && (4 * P2) % 512 == 0, "image bases must be multiples of 512 bytes");
};
// A operand of dh1 = W2': lane (in-unit 32mk + (lane & 31), half kh) gets out-units 32mi + 16s + 8kh + j of the weight image (64 rows, piece stride 8192); tbase = wide_tr_base<64>
__device__ __forceinline__ f16x8 load_frag_W_T(const char* wimg, int tmk, int tmk16, int piece, int mi, int s) {   // tmk = tbase ^ (64 mk), tmk16 = tmk ^ 16 (see load_frag_wide_T)
    const int off = (32 * mi + 16 * s) * 128 + piece * 8192;
    return __builtin_bit_cast(f16x8, frag8(lds_read_tr16(wimg, tmk + off), lds_read_tr16(wimg, tmk16 + off + 4 * 128)));
}

template <int KIND, int O, int HEAD>
__device__ __forceinline__ void grad_body_pair(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, H = 64, MT = 2;
    constexpr bool REC = true, kKeepH1 = HEAD == HEAD_VALUE;
    constexpr float kInvTanhScale = 1.0f / kTanhScale;
    using L = PairLds<D, O>;
#ifdef DRIL_STAMPS_EDGES
    unsigned long long e_[6];
#endif
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[0] = t_; }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 1, pr = wave >> 1;                                           // this wave's m-tile; this wave's pair
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    float* wl = smem;
    char* Wimg = reinterpret_cast<char*>(smem + L::WIMG);
    float* pb = smem + L::PAIR0 + pr * L::PAIR_SIZE;
    char* P1 = reinterpret_cast<char*>(pb + L::P1); char* P2 = reinterpret_cast<char*>(pb + L::P2); float* PO = pb + L::PO;
    constexpr int kNT = 256;                                                          // the workgroup: two pairs (compile-time strides: the staging loops unroll and their global loads leave together — at a run-time blockDim.x every iteration was its own round trip)
    {   // stage the small parts (as stage_net_split) and the W2 piece image
        const float* __restrict__ P = a.params;
        for (int i = tid; i < L::DP * H; i += kNT) { const int o = i % H, k = i / H; wl[L::W1T + k * H + o] = k < D ? kTanhScale * P[off.w1 + o + k * H] : 0.0f; }
        for (int i = tid; i < H; i += kNT) { wl[L::B1 + i] = kTanhScale * P[off.b1 + i]; wl[L::B2 + i] = (kTanhScale * kWScale * kActScale) * P[off.b2 + i]; }   // b2 starts the SCALED accumulator of L2
        for (int i = tid; i < O * H; i += kNT) { const int o = i % O, k = i / O; const float w3 = P[off.w3 + i]; wl[L::W3S + o * H + k] = w3 * (1.0f / kActScale); wl[L::W3B + o * H + k] = w3 * (1.0f / (kActScale * kActScale)); }
        for (int i = tid; i < L::OP; i += kNT) wl[L::B3 + i] = i < O ? P[off.b3 + i] : 0.0f;
#pragma unroll
        for (int i = tid; i < H * H / 2; i += kNT) {       // pair (k, k+1) of row o: W2 is column-major (out x in), consecutive threads read consecutive o
            const int o = i % H, kp = i / H;
            unsigned hi, lo;
            split2_pair((kTanhScale * kWScale) * P[off.w2 + o + H * (2 * kp)], (kTanhScale * kWScale) * P[off.w2 + o + H * (2 * kp + 1)], hi, lo);
            const int byte = o * 128 + ((((kp >> 2) ^ wimg_g<64>(o)) & 7) << 4) + ((kp & 3) << 2);
            *reinterpret_cast<unsigned*>(Wimg + byte) = hi; *reinterpret_cast<unsigned*>(Wimg + 8192 + byte) = lo;
        }
    }
    if (L::WIDE_IN) { for (int i = tid; i < 2 * L::PAIR_SIZE; i += kNT) { const int q = i % L::PAIR_SIZE; if (q >= L::XI && q < L::PO) smem[L::PAIR0 + i] = 0.f; } }
    __syncthreads();
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[1] = t_; }
#endif

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
    const int tbase = wide_tr_base<64>(lane);
    // gradient tiles are split as dz2 SG with SG = 2^(exponent of 1 / invB + 3): 4 ... 8 / invB, a power of two (every scale is undone exactly in the epilogue)
    const float sg = __uint_as_float((((__float_as_uint(1.0f / a.invB) >> 23) & 0xffu) + 3u) << 23);
    const float inv_sg = 1.0f / sg;
    GradArgs as = a; as.invB = a.invB * sg;                                            // what loss_head multiplies dLoss/dout with
    lds_char* lds = (lds_char*)smem;
    constexpr int kWimgB = 4 * L::WIMG, kP1B = 4 * L::P1, kP2B = 4 * L::P2, kP3B = 4 * L::P3;
    constexpr int KS = FirstLayer<D>::KS;
    constexpr bool WIDE_IN = L::WIDE_IN;
    constexpr float kObsScale = 16.0f;                                                  // D > 4: the observation image holds 16 x (|x| < 4 000: Acrobot's largest component is 28.3)
    const int pairB = 4 * (L::PAIR0 + pr * L::PAIR_SIZE);                                                 // byte offset of this pair's block (wave-uniform)
    const int rowc = c * 128 + ((h ^ wimg_g<64>(c)) << 4);                                                // row c, chunk h of the row: row reads step the chunk by 2 ks
    const int rowP = pairB + rowc;                                                                        // ... in a pair image (B operand of L2 / dh1)
    const int rowW = rowc + 4096 * w;                                                                     // ... in the weight image, rows 32 w .. (A operand of L2)
    const int ownT = pairB + c * 128 + 8 * h + (((4 * w) ^ wimg_g<64>(c)) << 4);                          // the lane's own chunk 4 w + g of row c (pair_store_pieces)

    f32x16 dW2[MT], dW3acc[O], db2acc, dW1acc[WIDE_IN ? 1 : D], db1acc;   // dW2: rows 32w.., all 64 columns; the others: per-lane sums over this lane's samples (units rowfn(r, h) of m-tile w)
    // D > 4: dW1acc[0] is an MFMA accumulator — rows = this wave's units, column n = observation component n (n < D) | the bias (n = D): dz1' x [x | 1] on the matrix cores,
    // because 16 (D + 1) per-lane accumulators do not fit beside the others; db1acc is unused
    float db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int r = 0; r < 16; ++r) { db2acc[r] = 0.f; db1acc[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int d = 0; d < (WIDE_IN ? 1 : D); ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW1acc[d][r] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        db3p[o] = 0.f; dlsp[o] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dW3acc[o][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int nb = HEAD == HEAD_VALUE ? (int)blockIdx.x - a.G / 2 : (int)blockIdx.x;  // workgroup within its net
    const int GP = HEAD == HEAD_VALUE ? a.Gc : a.G;                                   // pairs of this net
    const int g = 2 * nb + pr;                                                        // pair within its net = slab index
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    const int64_t g0 = 2 * nb;
    const int64_t trips = g0 < ntiles ? (ntiles - g0 + GP - 1) / GP : 0;           // the same for both pairs of the workgroup (barriers inside the loop)
    int64_t tile = g;
    // wave 0 of each pair is its loader: the tile's records go straight into the pair's LDS block (one gather per pair instead of one per wave into registers)
    float* RECp = pb + L::REC; float* VOp = pb + L::VO; int* VALp = reinterpret_cast<int*>(pb + L::VAL);
    TileIdx nidx; nidx.gidx = 0; nidx.g32 = 0; nidx.is32 = false; nidx.inb = false;
    if (w == 0) {
        request_records_lds<KIND, HEAD>(a, tile_index(a, tile, ntiles, c), lane, RECp, VOp, VALp);          // a tile index past the end gives an all-invalid tile
        nidx = tile_index(a, tile + GP, ntiles, c);
    }
    __syncthreads();                                                                  // (drains the LDS-DMA)
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[2] = t_; }
#endif
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int64_t it = 0; it < trips; ++it, tile += GP) {
        const int buf = (int)(it & 1);
        const float* recb = RECp + buf * L::RECB;
        float xk[KS];                                                                 // xk[s] = component 2s + h of sample c (zero beyond D: the records are zero-padded)
#pragma unroll
        for (int s = 0; s < KS; ++s) xk[s] = recb[((((2 * s) >> 2) * 32 + c) << 2) + ((2 * s) & 3) + h];
        if (WIDE_IN && w == 0) {                                                        // the observation image of this tile: row c = 16 x [x_0 .. x_(D-1) | 1 | 0 ..] as two f16 pieces (columns 0-7 and 8-15)
            unsigned xh[2][4], xl[2][4];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const unsigned u = __float_as_uint(xk[s]);
                const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);       // {x[2 s], x[2 s + 1]} in every lane
                const float a0 = 2 * s < D ? kObsScale * __uint_as_float(r[0]) : (2 * s == D ? kObsScale : 0.f), a1 = 2 * s + 1 < D ? kObsScale * __uint_as_float(r[1]) : (2 * s + 1 == D ? kObsScale : 0.f);
                split2_pair(a0, a1, xh[0][s], xl[0][s]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) { const float a0 = 8 + 2 * s == D ? kObsScale : 0.f; split2_pair(a0, 0.f, xh[1][s], xl[1][s]); }   // column 8 = the bias column when D = 8
            if (h == 0) {
                const int xa = pairB + 4 * L::XI + c * 64;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) { pl_write(lds, xa + 16 * ch, u32x4{xh[ch][0], xh[ch][1], xh[ch][2], xh[ch][3]}); pl_write(lds, xa + 2048 + 16 * ch, u32x4{xl[ch][0], xl[ch][1], xl[ch][2], xl[ch][3]}); }
            }
        }
        // ---- h1 tile w; its pieces into the pair's image ----
        f32x16 h1k;                                                                   // kept across the tile where the registers allow it (the critic), rebuilt from the pieces elsewhere
        {
            f32x16 h1w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16_scaled<false>(h1w, 1.0f);                                              // kActScale h1
            pair_store_pieces2<kP1B>(lds, opaque(ownT), h1w);
            if (kKeepH1) h1k = h1w;
        }
        STAMP(0);
        lds_barrier();                                                                // B1: the pair's h1 image complete (LDS only: the next tile's DMA stays in flight across B1 - B3)
        STAMP(1);
        if (w == 0) {                                                                 // next tile's records into the other buffer (last read a tile ago), the tile after next's epoch-order entry
            request_records_lds<KIND, HEAD>(a, nidx, lane, RECp + (buf ^ 1) * L::RECB, VOp + (buf ^ 1) * 64, VALp + (buf ^ 1) * 64);
            nidx = tile_index(a, tile + 2 * GP, ntiles, c);
        }
        // ---- h2 tile w = tanh(W2[rows of w] h1 + b2): A from the weight image, B from the pair's h1 image (both row reads with the same chunk index) ----
        f32x16 h2w;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B2 + 32 * w + 8 * q + 4 * h);
                h2w[4 * q + 0] = b[0]; h2w[4 * q + 1] = b[1]; h2w[4 * q + 2] = b[2]; h2w[4 * q + 3] = b[3];
            }
            const int ra = opaque(rowW), rb = opaque(rowP);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                                          // ks = 2 mi + s: chunk 2 ks + h of the row
                const int ak = ra ^ (ks << 5), bk = rb ^ (ks << 5);
                f16x8 A[2], B[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) { A[p] = pl_read<f16x8>(lds, ak + kWimgB + p * 8192); B[p] = pl_read<f16x8>(lds, bk + kP1B + p * 4096); }
                h2w = mfma_split3(A[0], A[1], B[0], B[1], h2w);
            }
            tanh16_scaled<true>(h2w, 1.0f / (kWScale * kActScale));                        // kActScale h2
        }
        STAMP(2);
        // ---- output layer: partial over this wave's 32 units, summed across the pair through LDS ----
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
            PO[(w * O + o) * 32 + c] = both_halves_sum(p);                                // both half-waves write the same bits to the same word
        }
        lds_barrier();                                                                // B2: both partial sums
        STAMP(3);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(recb + (((RecLayout<D>::RS - 1) * 32 + c) << 2));   // {action bits, adv, logp_old, ret} of this lane's sample
        TileIn<O, KS> cur; cur.act = 0; cur.s0 = 0.f; cur.s1 = 0.f;
        const bool valid = VALp[buf * 64 + c] != 0;
        if (HEAD == HEAD_VALUE) { cur.s0 = sc[3]; cur.s1 = a.has_clip_vf ? VOp[buf * 64 + c] : 0.f; }
        else {
            cur.s0 = sc[1]; cur.s1 = sc[2];
            if (HEAD == HEAD_CATEGORICAL) cur.act = __float_as_int(sc[0]) - a.action_start; else cur.xa[0] = sc[0];
        }
#pragma unroll
        for (int o = 0; o < O; ++o) out[o] = (wl[L::B3 + o] + PO[o * 32 + c]) + PO[(O + o) * 32 + c];   // fixed order: both waves get the same bits
        loss_head<O, HEAD>(as, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);     // dz = SG dLoss/dout
#pragma unroll
        for (int o = 0; o < O; ++o) {
            if (h == 0 && w == 0) db3p[o] += dz[o];
            fma16(dW3acc[o], dz[o], h2w);                                                  // dW3[o][unit] += dz[o][sample] h2[unit][sample]: the lane IS the sample
        }
        // ---- dz2 tile w (in h2w's registers); db2; its pieces into the pair's image ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3B + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * fmaf(-hv, hv, kActScale * kActScale); }   // (W3 / S^2 . dz SG) (S^2 - (S h2)^2) = SG dz2
        }
        add16(db2acc, h2w);
        pair_store_pieces2<kP2B>(lds, opaque(ownT), h2w);
        STAMP(4);
        lds_barrier();                                                                // B3: the pair's dz2 image complete
        STAMP(5);
        // ---- dz1 tile w = (W2'[rows of w] dz2) .* (1 - h1^2): A = transposed reads of the weight image, B = row reads of the dz2 image ----
        f32x16 g1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = 0.f;
            const int rb = opaque(rowP), tbw = opaque(tbase) ^ (64 * w), tbw16 = tbw ^ 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int bk = rb ^ (ks << 5);
                f16x8 A[2], B[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) { A[p] = load_frag_W_T(Wimg, tbw, tbw16, p, ks >> 1, ks & 1); B[p] = pl_read<f16x8>(lds, bk + kP2B + p * 4096); }
                g1 = mfma_split3(A[0], A[1], B[0], B[1], g1);
            }
            f32x16 h1r;
            if (kKeepH1) h1r = h1k; else pair_load_pieces2<kP1B>(lds, opaque(ownT), h1r);  // kActScale h1 of tile w back from its own pieces (to 2^-24)
            constexpr float c0 = kInvTanhScale / kWScale, c1 = c0 / (kActScale * kActScale);   // g1 = (kTanhScale kWScale W2' . SG dz2) (1 - h1^2) / (kTanhScale kWScale) = SG dz1
            if constexpr (WIDE_IN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float t2 = h1r[r] * h1r[r]; g1[r] = g1[r] * fmaf(-t2, c1, c0); }
            } else {                                                                      // SG dz1 / c1: the constant waits for the epilogue (dW1 | db1 are its only readers) — one multiply per element less
#pragma unroll
                for (int r = 0; r < 16; ++r) g1[r] = g1[r] * fmaf(-h1r[r], h1r[r], kActScale * kActScale);
            }
        }
        STAMP(6);
        // ---- dW1 | db1: per-lane accumulation, dW1[unit][d] += dz1[unit][sample] x[sample][d] (D <= 4); D > 4: dz1' [x | 1] on the matrix cores ----
        if constexpr (!WIDE_IN) {
            float x4[4];                                                               // xk[s] = x[2 s + h]: the lower half's value is x[2 s], the upper half's x[2 s + 1]
            { const f32x4 xq = *reinterpret_cast<const f32x4*>(recb + (c << 2)); x4[0] = xq[0]; x4[1] = xq[1]; x4[2] = xq[2]; x4[3] = xq[3]; }   // the observation quad of this lane's sample
#pragma unroll
            for (int d = 0; d < D; ++d) fma16(dW1acc[d], x4[d], g1);
            add16(db1acc, g1);
        } else {
            // this wave's own 32 columns of the dz1 image (nobody else reads them: the LDS queue of the wave orders the reads behind the stores), transposed fragments of
            // them as the A operand, of the observation image (complete since B1) as the B operand: (SG dz1)' (16 [x | 1])
            pair_store_pieces2<kP3B>(lds, opaque(ownT), g1);
            const int tbw = opaque(tbase) ^ (64 * w), tbw16 = tbw ^ 16;
            const int kh = lane >> 5, gm = (lane >> 4) & 1, e = lane & 15, xq = e >> 2, xp = e & 3;
            const int tx = (8 * kh + xq) * 64 + ((2 * gm + (xp >> 1)) << 4) + 8 * (xp & 1);   // wide_tr_base for 64-byte rows without a swizzle (4 KB read eight times per tile)
            const char* P3 = reinterpret_cast<const char*>(pb + L::P3); const char* XI = reinterpret_cast<const char*>(pb + L::XI);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 Az[2], Bx[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) { Az[p] = __builtin_bit_cast(f16x8, load_frag_wide_T<64>(P3, tbw, tbw16, p, s)); Bx[p] = __builtin_bit_cast(f16x8, load_frag_wide_T<32>(XI, tx, tx, p, s)); }
                dW1acc[0] = mfma_split3(Az[0], Az[1], Bx[0], Bx[1], dW1acc[0]);
            }
        }
        // ---- dW2[rows of w][:] += dz2 h1' (both operands as transposed fragments of the pair's images) ----
        {
            const int tb = opaque(tbase), tbw = tb ^ (64 * w), tbw16 = tbw ^ 16;
            f16x8 Az[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) Az[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<64>(P2, tbw, tbw16, p, s));
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                f16x8 Bh[2][2];
                const int tbj = tb ^ (64 * mj), tbj16 = tbj ^ 16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) Bh[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<64>(P1, tbj, tbj16, p, s));
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split3(Az[s][0], Az[s][1], Bh[s][0], Bh[s][1], dW2[mj]);   // (SG dz2)(kActScale h1)'
            }
        }
        STAMP(7);
        __syncthreads();                                                              // B4: the pair's images and partial sums free for the next tile (and, with the records in LDS, the next tile's DMA drained)
        STAMP(8);
    }
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[3] = t_; }
#endif
#if defined(DRIL_STAMPS) && !defined(DRIL_STAMPS_EDGES)
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k];
        o_[10] = (unsigned long long)trips; o_[11] = HEAD;
    }
#endif

    // ---- epilogue: every wave owns distinct gradient rows; the TWO pairs of a workgroup are summed into ONE slab (round 4: grad_reduce_kernel then folds half as many
    // slabs — 512 instead of 1 024 at configs[1]).  Both pairs ran the same number of tiles, so every wave reaches both barriers. ----
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)nb * SL;
    const float inv_s1 = inv_sg * ((kInvTanhScale / kWScale) / (kActScale * kActScale));   // the per-lane dW1 | db1 sums carry SG / c1 (the dz1 mask above)
    const float inv_sa = inv_sg * (1.0f / kActScale);                                  // products with an activation operand carry SG kActScale, the others SG (powers of two: exact)
    auto emit = [&](auto&& put) {
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 16; ++r) put(o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H, dW2[mj][r] * inv_sa);
        // per-lane sums over samples -> sums over the 32 lanes of each half (the halves hold different units).  One register-halving DPP reduce-scatter per accumulator
        // (half_reduce16_lane, ~52 vector instructions: lane l ends with the half's sum of register l & 15) — until round 5 this was 16 x 5 `__shfl_xor` steps per accumulator,
        // ~690 ds_bpermute per wave with a wait behind each: 45 - 59 k cycles per launch (DRIL_STAMPS_EDGES), 2.5 % of the kernel
        {
            const int rl = lane & 15, unit = 32 * w + rowfn(rl, h);
            const bool wr = (lane & 16) == 0;                                             // lanes l and l ^ 16 hold the same sums
            { const float b2 = half_reduce16_lane(db2acc, lane) * inv_sg; if (wr) put(o_b2 + unit, b2); }
            if constexpr (!WIDE_IN) {
                { const float b1 = half_reduce16_lane(db1acc, lane) * inv_s1; if (wr) put(o_b1 + unit, b1); }
#pragma unroll
                for (int d = 0; d < D; ++d) { const float v = half_reduce16_lane(dW1acc[d], lane) * inv_s1; if (wr) put(o_w1 + unit + d * H, v); }
            } else {                                                                  // column c of the MFMA accumulator: observation component c, or the bias
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u_ = 32 * w + rowfn(r, h);
                    const float v = dW1acc[0][r] * (inv_sg * (1.0f / 16.0f));
                    if (c < D) put(o_w1 + u_ + c * H, v); else if (c == D) put(o_b1 + u_, v);
                }
            }
#pragma unroll
            for (int o = 0; o < O; ++o) { const float v = half_reduce16_lane(dW3acc[o], lane) * inv_sa; if (wr) put(o_w3 + o + unit * O, v); }
            // db3 | dlog_std | the five loss sums: accumulated by the lower half of wave 0 only — one more reduce-scatter, lane l < 16 of that half ends with the sum of value l
            static_assert(2 * O + 5 <= 16, "scalar sums: one f32x16");
            f32x16 sc16;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc16[r] = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) { sc16[o] = db3p[o]; sc16[O + o] = dlsp[o]; }
#pragma unroll
            for (int k = 0; k < 5; ++k) sc16[2 * O + k] = st[k];
            const float tot = half_reduce16_lane(sc16, lane);
            if (w == 0 && lane < 16) {
                if (lane < O) put(o_b3 + lane, tot * inv_sg);
                else if (lane < 2 * O) { if (HEAD == HEAD_GAUSSIAN) put(o_ls + lane - O, tot * inv_sg); }
                else if (lane < 2 * O + 5) put(o_st + lane - 2 * O, tot);
            }
        }
    };
    // both pairs park at the same time (pair 0 in the first SL floats of the workgroup's LDS, pair 1 behind it: everything staged there is dead now), then all 256
    // threads write slab[i] = pair0[i] + pair1[i] with consecutive lanes on consecutive floats — a coalesced store instead of the per-lane scatter of the round 3 epilogue
    const int SLr = (SL + 3) & ~3;
    float* park = smem + pr * SLr;
    __syncthreads();                                                                  // the other pair may still be reading its images
    emit([&](int i, float v) { park[i] = v; });
    __syncthreads();
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[4] = t_; }
#endif
    const int pad0 = HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls;
#pragma unroll 6
    for (int i = tid; i < SL; i += 256) {                                             // (unrolled: the LDS reads of six iterations leave together)
        const bool pad = (i >= pad0 && i < o_st) || i >= o_st + 5;                    // slab padding and the three unused statistics slots
        slab[i] = pad ? 0.f : smem[i] + smem[SLr + i];
    }
#ifdef DRIL_STAMPS_EDGES
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); e_[5] = t_; }
#endif
#ifdef DRIL_STAMPS_EDGES
    if (lane == 0 && a.dbg) {           // rows: staging | first records | tile loop | sums + park | slab write   (the reader divides by o_[10] = 1)
        unsigned long long* o_ = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int k = 0; k < 5; ++k) o_[k] = e_[k + 1] - e_[k];
        for (int k = 5; k < 10; ++k) o_[k] = 0;
        o_[10] = 1; o_[11] = HEAD;
    }
#endif
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void ppo_grad_pair_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];          // PairLds: XOR addressing needs 512-byte image bases
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)(a.G / 2);
    if (actor) grad_body_pair<KIND, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN>(a, smem);
    else grad_body_pair<KIND, 1, HEAD_VALUE>(a, smem);
}

template <int KIND> static size_t grad_pair_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = PairLds<D, A>::END, wc = PairLds<D, 1>::END;
    return sizeof(float) * (wa > wc ? wa : wc);
}

hipError_t launch_ppo_grad_pair(int kind, const GradArgs& a, hipStream_t s) {
    if (kind == 7) kind = 4;                  // ScalingWrapperEnv(MountainCarContinuous): the update never touches the simulator
#define CALLP(K) { const size_t lds = grad_pair_lds_bytes<K>(); \
        if (2 * ((a.slab_a + 3) & ~3) > PairLds<EnvSpec<K>::D, EnvSpec<K>::A>::END || 2 * ((a.slab_c + 3) & ~3) > PairLds<EnvSpec<K>::D, 1>::END) return hipErrorInvalidValue;   /* the epilogue parks both pairs' slabs in the workgroup's LDS */ \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_grad_pair_kernel<K>, lds); if (e != hipSuccess) return e; } \
        ppo_grad_pair_kernel<K><<<(a.G + a.Gc) / 2, 256, lds, s>>>(a); }
    if (kind == 0) CALLP(0) else if (kind == 3) CALLP(3) else if (kind == 4) CALLP(4) else if (kind == 6) CALLP(6) else CALLP(1)
#undef CALLP
    return hipGetLastError();
}

}  // namespace dril

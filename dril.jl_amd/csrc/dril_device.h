// dril_device.h — device-side building blocks shared by every kernel of libdril_hip.so (gfx950 only).
//
//   * Philox4x32-10 counter RNG + the keyed index bijection that stands in for MLUtils' shuffle
//   * env physics (Gymnasium CartPole-v1 / Pendulum-v1 — SURVEY.md §8c item 3)
//   * the wave-level MLP tile: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) with weights staged in LDS
//
// MLP tile geometry (docs/kernels/ppo_kernels.md).  One wave owns a tile of 32 samples.  Activations live in
// MFMA C/D layout: f32x16 X[H/32]; element (mt, r, lane) is X[row = 32*mt + rowfn(r, lane>>5)][col = lane&31]
// with rows = hidden units and cols = samples.  A layer Y = W * X takes X straight from the accumulator
// registers as the B operand (k-step (mi, r): lanes 0-31 contribute hidden row rowfn(r,0), lanes 32-63
// rowfn(r,1)); W comes from LDS as the A operand, 4 consecutive k per ds_read_b128 from a [out][in+4]
// image (the +4 pad makes the 16-lane b128 groups conflict-free).  No LDS round trip between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
namespace dril {

// Diagnostic / experiment switches (ablation bits, grid caps, the older form of a kernel for an A/B) are honoured only when DRIL_DEBUG=1 is set as well: a stray variable
// cannot change what a production run executes.  The documented switches (DRIL_GRAD_VARIANT, DRIL_NO_F32_RETRY, DRIL_NO_PERSISTENT_UPDATE, DRIL_FORCE_*, DRIL_NO_EPOCH_INDEX,
// DRIL_SAC_NO_FUSED_*, DRIL_SMALL_DEBUG_SOLO, DRIL_GRAD_ACTOR_PERMILLE) are read directly; DESIGN.md section 9 lists both groups.
// NNlib.relu(x) = max(zero(x), x), and Julia's max PROPAGATES NaN: a NaN pre-activation must stay NaN (x > 0 ? x : 0 would turn it into 0 and hide a broken parameter
// from the finiteness checks downstream)
__host__ __device__ inline float relu_nan(float x) { return (x > 0.f || x != x) ? x : 0.f; }
inline const char* debug_env(const char* name) {
    const char* e = std::getenv("DRIL_DEBUG");                  // read at every call (handle creation, first launch of a contraction shape): tests toggle it per handle
    return (e && std::atoi(e) != 0) ? std::getenv(name) : nullptr;
}



typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kTile = 32;   // samples per wave tile (MFMA N)
constexpr int kWPad = 4;    // pad of weight image rows (floats)
constexpr int kTS = 36;     // row stride of the transposed activation images (32 samples + 4)

// ---------------------------------------------------------------------------------------------
// RNG (spec shared with oracle/dril_oracle.c; both are restatements of the same published Philox)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                              uint32_t c3, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__host__ __device__ inline float u01_f32(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline double u01_f64(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ inline float randn_f32(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// keyed bijection on [0, n): position in the epoch -> buffer index (DataLoader(shuffle=true), ppo.jl:188-195)
__host__ __device__ inline uint64_t mix_bij(uint64_t x, uint64_t key, int bits) {
    const uint64_t mask = bits >= 64 ? ~0ull : (((uint64_t)1 << bits) - 1);
    const int s = bits / 2 > 0 ? bits / 2 : 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x ^= (key >> (r * 13)) & mask;
        x = (x * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) & mask;
        x ^= x >> s;
        x = (x * 0xBF58476D1CE4E5B9ull) & mask;
        x ^= x >> (s + 1 < bits ? s + 1 : s);
    }
    return x;
}
// the same map in 32-bit arithmetic for bits <= 32 (every buffer up to 4 G samples): the low `bits` bits of a product / sum depend only on the low bits of the
// operands, so truncating the constants and working modulo 2^32 gives the identical permutation — with one v_mul_lo_u32 per multiply instead of the four
// quarter-rate multiplies + carries of an emulated 64-bit product (the gather of every update kernel evaluates this once per lane and tile: ~1 000 -> ~250 cycles)
__host__ __device__ inline uint32_t mix_bij32(uint32_t x, uint64_t key, int bits) {
    const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    const int s = bits / 2 > 0 ? bits / 2 : 1, s2 = s + 1 < bits ? s + 1 : s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x ^= (uint32_t)(key >> (r * 13)) & mask;
        x = (x * 0x7F4A7C15u + 0xD192ED03u) & mask;
        x ^= x >> s;
        x = (x * 0x1CE4E5B9u) & mask;
        x ^= x >> s2;
    }
    return x;
}
__host__ __device__ inline int64_t perm_index(int64_t p, int64_t n, uint64_t key, int bits) {
    if (bits <= 32) {
        uint32_t x = (uint32_t)p;
        do { x = mix_bij32(x, key, bits); } while ((int64_t)x >= n);
        return (int64_t)x;
    }
    uint64_t x = (uint64_t)p;
    do { x = mix_bij(x, key, bits); } while ((int64_t)x >= n);
    return (int64_t)x;
}
__host__ inline int perm_bits(int64_t n) { int b = 1; while (((int64_t)1 << b) < n) ++b; return b; }

// inverse of mix_bij: every step is invertible on `bits`-bit words (xorshift-right: x = y ^ y>>s ^ y>>2s ...; odd multiply:
// modular inverse; xor key).  Used to bin buffer indices by minibatch in ONE sequential pass per epoch.
__host__ __device__ inline uint64_t unxorshift(uint64_t y, int s, int bits) {
    uint64_t x = y;
    for (int sh = s; sh < bits; sh += s) x ^= y >> sh;
    return x;
}
__host__ __device__ constexpr uint64_t mul_inverse(uint64_t a) {   // Newton iteration mod 2^64, a odd
    uint64_t x = a;
    for (int i = 0; i < 6; ++i) x *= 2 - a * x;
    return x;
}
__host__ __device__ inline uint64_t mix_bij_inv(uint64_t x, uint64_t key, int bits) {
    const uint64_t mask = bits >= 64 ? ~0ull : (((uint64_t)1 << bits) - 1);
    const int s = bits / 2 > 0 ? bits / 2 : 1;
    const int s2 = s + 1 < bits ? s + 1 : s;
    constexpr uint64_t I1 = mul_inverse(0x9E3779B97F4A7C15ull), I2 = mul_inverse(0xBF58476D1CE4E5B9ull);
#pragma unroll
    for (int r = 3; r >= 0; --r) {
        x = unxorshift(x, s2, bits);
        x = (x * I2) & mask;
        x = unxorshift(x, s, bits);
        x = ((x - 0xD1B54A32D192ED03ull) * I1) & mask;
        x ^= (key >> (r * 13)) & mask;
    }
    return x;
}
__host__ __device__ inline uint32_t unxorshift32(uint32_t y, int s, int bits) {
    uint32_t x = y;
    for (int sh = s; sh < bits; sh += s) x ^= y >> sh;
    return x;
}
__host__ __device__ inline uint32_t mix_bij_inv32(uint32_t x, uint64_t key, int bits) {   // inverse of mix_bij32 (the inverses modulo 2^64 reduce to those modulo 2^32)
    const uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    const int s = bits / 2 > 0 ? bits / 2 : 1;
    const int s2 = s + 1 < bits ? s + 1 : s;
    constexpr uint32_t I1 = (uint32_t)mul_inverse(0x9E3779B97F4A7C15ull), I2 = (uint32_t)mul_inverse(0xBF58476D1CE4E5B9ull);
#pragma unroll
    for (int r = 3; r >= 0; --r) {
        x = unxorshift32(x, s2, bits);
        x = (x * I2) & mask;
        x = unxorshift32(x, s, bits);
        x = ((x - 0xD192ED03u) * I1) & mask;
        x ^= (uint32_t)(key >> (r * 13)) & mask;
    }
    return x;
}
// position in the epoch order of buffer index idx (inverse of perm_index; cycle walking inverts by walking backwards)
__host__ __device__ inline int64_t perm_position(int64_t idx, int64_t n, uint64_t key, int bits) {
    if (bits <= 32) {
        uint32_t x32 = (uint32_t)idx;
        do { x32 = mix_bij_inv32(x32, key, bits); } while ((int64_t)x32 >= n);
        return (int64_t)x32;
    }
    uint64_t x = (uint64_t)idx;
    do { x = mix_bij_inv(x, key, bits); } while ((int64_t)x >= n);
    return (int64_t)x;
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
// tanh(x) = 1 - 2 / (e^{2x} + 1) on the 1-ulp hardware units (v_exp_f32, v_rcp_f32): 5 VALU instructions,
// absolute error <= ~2e-7 over the whole line (exp2 overflow -> rcp(inf) = 0 -> 1; underflow -> 1 - 2 = -1).
// The first version (odd polynomial near 0 + correctly rounded division) cost ~20 instructions and was 55 %
// of the VALU work of ppo_grad_kernel (profiles/r01_v1_*).  The reference applies Julia's tanh inside
// Lux.Dense (layer_helpers.jl:33-56); activations only enter the outputs through O(1) weights, so absolute
// error is what matters for the 1e-4 loss tolerance.
__device__ __forceinline__ float tanh_f32(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x} = 2^{2x log2(e)}
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// The forward weight images (W1, b1, W2, b2) are staged PRE-SCALED by 2*log2(e), so the MFMA accumulator already holds
// 2*log2(e)*x and tanh needs no multiply: tanh(x) = 1 - 2/(2^acc + 1).  The backward reads the unscaled W2' image.
constexpr float kTanhScale = 2.8853900817779268f;

// ---------------------------------------------------------------------------------------------
// (rounds 2 - 3; since the end of round 3 only the generic contractions of dril_gemm.hip, whose operands have any range — the fused kernels use the f16 form below)
// fp32-equivalent contraction on the bf16 matrix cores: every f32 operand is cut into three bf16 pieces x = hi + mid + lo (nearest bf16, exact
// remainder, twice: exact for a 24-bit mantissa) and a k16 step of a 32x32 tile is six v_mfma_f32_32x32x16_bf16 (hi.hi hi.mid mid.hi mid.mid hi.lo lo.hi;
// the three dropped partial products are <= 2^-23 relative; f32 accumulate).  profiles/r01_bf16_split_microbench.md: 2.0x the rate of
// v_mfma_f32_32x32x2_f32 with pre-split operands, max error 1.0e-7 vs 1.5e-7 for the f32 MFMA chain — not a precision reduction, and the matrix pipe
// runs BESIDE the VALU where the f32 MFMA runs ON its lanes (profiles/r01_mfma_valu_microbench.md)
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// pieces of two values packed for one 32-bit word, {piece of b, piece of a} in the high / low half.  ROUND-TO-NEAREST pieces (v_cvt_pk_bf16_f32, one instruction
// per pair and piece): hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid); both remainders are exact in f32 and lo is exact for a 24-bit mantissa, so
// x = hi + mid + lo.  Round 2 cut the pieces by TRUNCATION (mask + v_perm_b32, the same 11 VALU per pair): every remainder then has the sign of x, the three
// dropped partial products (mid.lo, lo.mid, lo.lo: up to 2 x 2^-24 relative) all pull a product toward zero, and the f64 error budget of
// tests/test_gpu_split_arith.py measured it — dW2 shrunk by 2.1e-7 of itself, 2.6x the f32 kernel's distance from the float64 gradient.  Nearest rounding makes
// the remainders half as large and signed at random: the dropped terms are <= 2^-26 relative and unbiased (profiles/r03_split_arith.md).
// (a vector cast, NOT inline asm: the hazard recogniser does not see through an asm statement, so the software wait states this part needs between an
// MFMA / trans result and a VALU consumer are not inserted around it — the first version of this function was `asm("v_cvt_pk_bf16_f32 ...")`, and one
// instantiation of rollout_kernel (ScalingWrapperEnv, last-value forward) then produced a NaN for one env, deterministically for that build:
// profiles/r03_split_arith.md section 5)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));     // v_cvt_pk_bf16_f32 (round to nearest even; a NaN stays a NaN)
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = cvt_pk_bf16(a, b);
    const float ar = a - __uint_as_float(hi << 16), br = b - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(ar, br);
    const float aq = ar - __uint_as_float(mid << 16), bq = br - __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(aq, bq);
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// one k16 step of the split product, small terms first
// -DDRIL_DEBUG_DROP_LO (libdril_hip_droplo.so, the NEGATIVE CONTROL of tests/test_gpu_split_arith.py, never the product): the two products with a `lo`
// piece are left out, i.e. a two-piece split (2^-16 relative) — the error-budget tests must FAIL on that build, which is what gives them power
__device__ __forceinline__ f32x16 mfma_split6(bf16x8 Ah, bf16x8 Am, bf16x8 Al, bf16x8 Bh, bf16x8 Bm, bf16x8 Bl, f32x16 acc) {
#ifndef DRIL_DEBUG_DROP_LO
    acc = mfma_bf16(Al, Bh, acc); acc = mfma_bf16(Ah, Bl, acc);
#endif
    acc = mfma_bf16(Am, Bm, acc);
    acc = mfma_bf16(Am, Bh, acc); acc = mfma_bf16(Ah, Bm, acc); acc = mfma_bf16(Ah, Bh, acc);
    return acc;
}
// ---------------------------------------------------------------------------------------------
// the same on f16 pieces (round 3, ppo_grad_pair_kernel): x S = hi + lo with hi = rn_f16(x S), lo = rn_f16(x S - hi) (both roundings to nearest: hi carries 11 bits, the
// exact remainder has at most 13 and lo keeps 11 of them, so |x S - hi - lo| <= 2^-24 |x S| — the relative precision of an f32 rounding — as long as lo stays a NORMAL
// f16, i.e. |x S| >= 2^-14 2^11 = 0.125; below that the error is an absolute 2^-25.  The power-of-two scale S moves the operand range there: activations (|h| < 1) use
// kActScale, the staged weights kWScale, the gradient tiles 4 / invB rounded up to a power of two.  A k16 step of a 32x32 tile is THREE v_mfma_f32_32x32x16_f16
// (lo.hi, hi.lo, hi.hi; the dropped lo.lo is <= 2^-24 relative; f32 accumulate) instead of six bf16 ones, and a pair of values splits in 6 VALU instead of 11.
// Against a float64 gradient this arithmetic is 0.97 - 1.16 x as far as the exact-f32 kernel on the device (the six-product bf16 form 0.85 - 1.04 x, a
// four-product bf16 form 12 - 39 x, hi.hi alone 1 000 - 5 700 x): tests/test_gpu_split_arith.py, profiles/r03_split_arith.md section 6.
// f16 has a RANGE: |x S| > 65 504 overflows.  The scales leave |w| < 350 for weights, a factor ~ 4 000 over a typical gradient tile; an overflow becomes Inf / NaN in
// the gradient and is reported like any non-finite gradient (ppo.jl:213-214), never silently wrong.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr float kActScale = 256.0f, kWScale = 64.0f;
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));       // v_cvt_pk_f16_f32 (round to nearest even)
}
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_f16(a, b);
    float ra, rb;                                                                                 // x - (float)hi.half in ONE instruction (left to itself the compiler picks the mixed form for about half of them and v_cvt_f32_f16 + v_sub for the rest)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = cvt_pk_f16(ra, rb);                                                                      // exact remainders (Sterbenz), rounded once
}
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// one k16 step of the two-piece product, small terms first.  -DDRIL_DEBUG_DROP_LO (negative control): hi.hi only, an 11-bit product
__device__ __forceinline__ f32x16 mfma_split3(f16x8 Ah, f16x8 Al, f16x8 Bh, f16x8 Bl, f32x16 acc) {
#ifndef DRIL_DEBUG_DROP_LO
    acc = mfma_f16(Al, Bh, acc); acc = mfma_f16(Ah, Bl, acc);
#endif
    return mfma_f16(Ah, Bh, acc);
}
// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements is delivered column-major — lane 4q+p of the group gives the
// address of row q, columns 4p..4p+3; lane i receives column i, row q in element q (cdna_hip_programming.md T10).  EXEC must be all ones.
__device__ __forceinline__ s16x4 lds_read_tr16(const char* lds_base, int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_base + byte_off));
}
__device__ __forceinline__ bf16x8 frag8(s16x4 a, s16x4 b) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7)); }
__device__ __forceinline__ bf16x8 frag8(u32x2 a, u32x2 b) { return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3)); }

// ---------------------------------------------------------------------------------------------
// environments
// ---------------------------------------------------------------------------------------------
template <int KIND> struct EnvSpec;
template <> struct EnvSpec<0> { static constexpr int D = 4, S = 4, A = 2; static constexpr bool discrete = true; };
template <> struct EnvSpec<1> { static constexpr int D = 3, S = 2, A = 1; static constexpr bool discrete = false; };
template <> struct EnvSpec<2> : EnvSpec<1> {};   // ScalingWrapperEnv(PendulumEnv()): same simulator, affine maps at the boundary
template <> struct EnvSpec<3> { static constexpr int D = 2, S = 2, A = 3; static constexpr bool discrete = true; };    // MountainCar-v0: (position, velocity), Discrete(3)
template <> struct EnvSpec<4> { static constexpr int D = 2, S = 2, A = 1; static constexpr bool discrete = false; };   // MountainCarContinuous-v0: Box(-1, 1)
template <> struct EnvSpec<7> : EnvSpec<4> {};   // ScalingWrapperEnv(MountainCarContinuousEnv()): same simulator, affine maps at the boundary (observations Box((-1.2, -0.07), (0.6, 0.07)) -> Box(-1, 1))
template <> struct EnvSpec<6> { static constexpr int D = 6, S = 4, A = 3; static constexpr bool discrete = true; };    // Acrobot-v1: (cos t1, sin t1, cos t2, sin t2, w1, w2), Discrete(3); fused at [64,64] / [128,128] / [256,256] through FirstLayer<6>, generic otherwise
// ScalingWrapperEnv (scalingWrapperEnv.jl): scale! :71-74 `(x - low) * sf - 1`, unscale! :76-79 `(x + 1) / sf + low`, sf = 2 / (high - low) :36-44
__host__ __device__ inline float scale_to_unit(float x, float low, float high) { const float sf = 2.0f / (high - low); return (x - low) * sf - 1.0f; }
__host__ __device__ inline float unscale_from_unit(float x, float low, float high) { const float sf = 2.0f / (high - low); return (x + 1.0f) / sf + low; }
// bound of the agent-facing action space (ClampAdapter / TanhScaleAdapter act on action_space(env)): Box(-2,2), Box(-1,1) under the wrapper
template <int KIND> __host__ __device__ constexpr float act_bound() { return (KIND == 2 || KIND == 4 || KIND == 7) ? 1.0f : 2.0f; }

// Acrobot-v1 (Gymnasium, "book" dynamics): d(theta1, theta2, dtheta1, dtheta2)/dt under torque a on the second joint
__host__ __device__ inline void acrobot_dsdt(const float* s, float a, float* ds) {
    const float m1 = 1.0f, m2 = 1.0f, l1 = 1.0f, lc1 = 0.5f, lc2 = 0.5f, I1 = 1.0f, I2 = 1.0f, g = 9.8f, hpi = 1.57079632679489661923f;
    const float t1 = s[0], t2 = s[1], w1 = s[2], w2 = s[3];
    const float c2 = cosf(t2), s2 = sinf(t2);
    const float d1 = m1 * lc1 * lc1 + m2 * (l1 * l1 + lc2 * lc2 + 2.0f * l1 * lc2 * c2) + I1 + I2;
    const float d2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2;
    const float phi2 = m2 * lc2 * g * cosf(t1 + t2 - hpi);
    const float phi1 = -m2 * l1 * lc2 * w2 * w2 * s2 - 2.0f * m2 * l1 * lc2 * w2 * w1 * s2 + (m1 * lc1 + m2 * l1) * g * cosf(t1 - hpi) + phi2;
    const float dd2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * w1 * w1 * s2 - phi2) / (m2 * lc2 * lc2 + I2 - d2 * d2 / d1);
    const float dd1 = -(d2 * dd2 + phi1) / d1;
    ds[0] = w1; ds[1] = w2; ds[2] = dd1; ds[3] = dd2;
}
__host__ __device__ inline float acrobot_wrap(float x) {            // wrap(x, -pi, pi): while loops of the Gymnasium helper
    const float pi = 3.14159265358979323846f;
    while (x > pi) x -= 2.0f * pi;
    while (x < -pi) x += 2.0f * pi;
    return x;
}
// one env step (dt = 0.2, one RK4 step); returns the reward, sets *terminated
__host__ __device__ inline float acrobot_step(float* st, int act_i, bool fixed_len, bool* terminated) {
    const float dt = 0.2f, a = (float)(act_i - 1), pi = 3.14159265358979323846f;
    float k1[4], k2[4], k3[4], k4[4], y[4];
    acrobot_dsdt(st, a, k1);
    for (int i = 0; i < 4; ++i) y[i] = st[i] + 0.5f * dt * k1[i];
    acrobot_dsdt(y, a, k2);
    for (int i = 0; i < 4; ++i) y[i] = st[i] + 0.5f * dt * k2[i];
    acrobot_dsdt(y, a, k3);
    for (int i = 0; i < 4; ++i) y[i] = st[i] + dt * k3[i];
    acrobot_dsdt(y, a, k4);
    for (int i = 0; i < 4; ++i) y[i] = st[i] + dt / 6.0f * (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]);
    st[0] = acrobot_wrap(y[0]); st[1] = acrobot_wrap(y[1]);
    st[2] = fminf(fmaxf(y[2], -4.0f * pi), 4.0f * pi); st[3] = fminf(fmaxf(y[3], -9.0f * pi), 9.0f * pi);
    const bool term = -cosf(st[0]) - cosf(st[1] + st[0]) > 1.0f;
    *terminated = fixed_len ? false : term;
    return term ? 0.0f : -1.0f;
}

template <int KIND> __device__ inline void env_reset(uint64_t env_seed, uint32_t episode, float* st) {
    uint32_t r[4];
    philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), episode, 0, 0, 0, r);
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = u01_f32(r[i]) * 0.1f - 0.05f;
    } else if (KIND == 6) {                                        // Acrobot: U(-0.1, 0.1)^4
#pragma unroll
        for (int i = 0; i < 4; ++i) st[i] = u01_f32(r[i]) * 0.2f - 0.1f;
    } else if (KIND == 3 || KIND == 4 || KIND == 7) {              // MountainCar: position ~ U(-0.6, -0.4), velocity 0
        st[0] = u01_f32(r[0]) * 0.2f - 0.6f; st[1] = 0.f;
    } else {
        st[0] = u01_f32(r[0]) * 6.28318530717958647692f - 3.14159265358979323846f;
        st[1] = u01_f32(r[1]) * 2.0f - 1.0f;
    }
}
template <int KIND> __device__ inline void env_obs(const float* st, float* obs) {
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) obs[i] = st[i];
    } else if (KIND == 3 || KIND == 4) { obs[0] = st[0]; obs[1] = st[1]; }
    else if (KIND == 7) { obs[0] = scale_to_unit(st[0], -1.2f, 0.6f); obs[1] = scale_to_unit(st[1], -0.07f, 0.07f); }   // observe(::ScalingWrapperEnv) :93-98
    else if (KIND == 6) { obs[0] = cosf(st[0]); obs[1] = sinf(st[0]); obs[2] = cosf(st[1]); obs[3] = sinf(st[1]); obs[4] = st[2]; obs[5] = st[3]; }
    else {
        obs[0] = cosf(st[0]); obs[1] = sinf(st[0]); obs[2] = st[1];
        if (KIND == 2) {                                           // observe(::ScalingWrapperEnv) :93-98 on Box((-1,-1,-8), (1,1,8))
            obs[0] = scale_to_unit(obs[0], -1.0f, 1.0f); obs[1] = scale_to_unit(obs[1], -1.0f, 1.0f); obs[2] = scale_to_unit(obs[2], -8.0f, 8.0f);
        }
    }
}
// one step; act_i is the env-space discrete action (0/1), act_f the env-space continuous action
template <int KIND> __device__ inline float env_step(float* st, float act_f, int act_i, bool fixed_len, bool* terminated) {
    if (KIND == 6) return acrobot_step(st, act_i, fixed_len, terminated);
    if (KIND == 0) {
        const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f;
        const float polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
        float x = st[0], x_dot = st[1], th = st[2], th_dot = st[3];
        const float force = act_i == 1 ? force_mag : -force_mag;
        const float c = cosf(th), s = sinf(th);
        const float temp = (force + polemass_length * th_dot * th_dot * s) / total_mass;
        const float thacc = (gravity * s - c * temp) / (length * (4.0f / 3.0f - masspole * c * c / total_mass));
        const float xacc = temp - polemass_length * thacc * c / total_mass;
        x = x + tau * x_dot; x_dot = x_dot + tau * xacc;
        th = th + tau * th_dot; th_dot = th_dot + tau * thacc;
        st[0] = x; st[1] = x_dot; st[2] = th; st[3] = th_dot;
        const bool term = (x < -2.4f) || (x > 2.4f) || (th < -0.20943951023931953f) || (th > 0.20943951023931953f);
        *terminated = fixed_len ? false : term;
        return 1.0f;
    } else if (KIND == 3 || KIND == 4 || KIND == 7) {
        // MountainCar-v0 (act_i in {0,1,2}) / MountainCarContinuous-v0 (act_f clipped to [-1,1]); Gymnasium equations
        if (KIND == 7) act_f = unscale_from_unit(act_f, -1.0f, 1.0f);   // act!(::ScalingWrapperEnv, action) :110-113 (the same Box: the affine map is still evaluated)
        const float min_position = -1.2f, max_position = 0.6f, max_speed = 0.07f;
        float position = st[0], velocity = st[1], reward;
        float force = 0.f;
        if (KIND == 3) velocity += (float)(act_i - 1) * 0.001f + cosf(3.0f * position) * (-0.0025f);
        else { force = fminf(fmaxf(act_f, -1.0f), 1.0f); velocity += force * 0.0015f - 0.0025f * cosf(3.0f * position); }
        velocity = fminf(fmaxf(velocity, -max_speed), max_speed);
        position += velocity;
        position = fminf(fmaxf(position, min_position), max_position);
        if (position == min_position && velocity < 0.f) velocity = 0.f;
        const bool goal = position >= (KIND == 3 ? 0.5f : 0.45f) && velocity >= 0.f;
        if (KIND == 3) reward = -1.0f; else reward = (goal ? 100.0f : 0.0f) - force * force * 0.1f;
        st[0] = position; st[1] = velocity;
        *terminated = fixed_len ? false : goal;
        return reward;
    } else {
        const float max_speed = 8.0f, max_torque = 2.0f, dt = 0.05f, g = 10.0f, m = 1.0f, l = 1.0f;
        const float pi = 3.14159265358979323846f;
        const float th = st[0], thdot = st[1];
        if (KIND == 2) act_f = unscale_from_unit(act_f, -2.0f, 2.0f);   // act!(::ScalingWrapperEnv, action) :110-113
        const float u = fminf(fmaxf(act_f, -max_torque), max_torque);
        float an = fmodf(th + pi, 2.0f * pi); if (an < 0) an += 2.0f * pi; an -= pi;
        const float cost = an * an + 0.1f * thdot * thdot + 0.001f * u * u;
        float nthdot = thdot + (3.0f * g / (2.0f * l) * sinf(th) + 3.0f / (m * l * l) * u) * dt;
        nthdot = fminf(fmaxf(nthdot, -max_speed), max_speed);
        st[0] = th + nthdot * dt; st[1] = nthdot;
        *terminated = false;
        return -cost;
    }
}

// ---------------------------------------------------------------------------------------------
// parameter layout (flat, include/dril_hip.h): net = {W1 b1 W2 b2 W3 b3}, W column-major (out x in)
// ---------------------------------------------------------------------------------------------
struct NetOff { int w1, b1, w2, b2, w3, b3, end; };
__host__ __device__ inline NetOff net_off(int base, int D, int H1, int H2, int O) {
    NetOff n; n.w1 = base; n.b1 = n.w1 + H1 * D; n.w2 = n.b1 + H1; n.b2 = n.w2 + H2 * H1;
    n.w3 = n.b2 + H2; n.b3 = n.w3 + O * H2; n.end = n.b3 + O; return n;
}

// first layer: the observation enters as KS k-steps of v_mfma_f32_32x32x2_f32, k-step s carrying components (2s, 2s + 1) on the two half-waves: two steps for D <= 4
// (every env kind of rounds 1-2), four for D <= 8 (round 3: Acrobot-v1, D = 6).  DP = padded observation width = rows of the W1T image.
template <int D> struct FirstLayer { static constexpr int KS = D <= 4 ? 2 : 4, DP = 2 * KS; static_assert(D <= 8, "observation dims > 8 run on the generic kernels"); };

// LDS image of one net (offsets in floats; every block starts on a 16-byte boundary)
template <int D, int H1, int H2, int O> struct NetLds {
    static constexpr int DP = FirstLayer<D>::DP;         // obs dim padded to the MFMA k-pairing
    static constexpr int WS1 = H1 + kWPad;              // row stride of W2S ([out=H2][in=H1])
    static constexpr int WS2 = H2 + kWPad;              // row stride of W2T ([in=H1][out=H2])
    static constexpr int OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0;                       // [DP][H1]   W1T[k][o] = W1[o][k]
    static constexpr int B1 = W1T + DP * H1;
    static constexpr int W2S = B1 + H1;                 // [H2][WS1]  W2S[o][k] = W2[o][k]
    static constexpr int B2 = W2S + H2 * WS1;
    static constexpr int W3S = B2 + H2;                 // [O][H2]
    static constexpr int B3 = W3S + O * H2;
    static constexpr int FWD_END = B3 + OP;
    static constexpr int W2T = FWD_END;                 // [H1][WS2]  W2T[k][o] = W2[o][k]   (backward only)
    static constexpr int BWD_END = W2T + H1 * WS2;
    static_assert(D <= DP, "obs dim beyond the first-layer pairing");
    static_assert(H1 % 32 == 0 && H2 % 32 == 0, "hidden widths must be multiples of 32");
};

// cooperative global -> LDS staging of one net's weights (all threads of the workgroup)
template <int D, int H1, int H2, int O, bool BWD>
__device__ inline void stage_net(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    using L = NetLds<D, H1, H2, O>;
    for (int i = tid; i < L::DP * H1; i += nthreads) {
        const int o = i % H1, k = i / H1;
        lds[L::W1T + k * H1 + o] = k < D ? kTanhScale * P[n.w1 + o + k * H1] : 0.0f;
    }
    for (int i = tid; i < H1; i += nthreads) lds[L::B1 + i] = kTanhScale * P[n.b1 + i];
    for (int i = tid; i < H2 * H1; i += nthreads) {
        const int o = i % H2, k = i / H2;
        const float w = P[n.w2 + i];
        lds[L::W2S + o * L::WS1 + k] = kTanhScale * w;
        if (BWD) lds[L::W2T + k * L::WS2 + o] = w;
    }
    for (int i = tid; i < H2; i += nthreads) lds[L::B2 + i] = kTanhScale * P[n.b2 + i];
    for (int i = tid; i < O * H2; i += nthreads) {
        const int o = i % O, k = i / O;
        lds[L::W3S + o * H2 + k] = P[n.w3 + i];
    }
    for (int i = tid; i < L::OP; i += nthreads) lds[L::B3 + i] = i < O ? P[n.b3 + i] : 0.0f;
}

__device__ __forceinline__ int rowfn(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Y[MO] (+bias) = W * X[KI]; W image [32*MO rows][WS], A operand by ds_read_b128
template <int KI, int MO, bool BIAS>
__device__ __forceinline__ void dense_mfma(const float* __restrict__ Wimg, int WS, const float* __restrict__ bias,
                                           const f32x16 (&X)[KI], f32x16 (&Y)[MO], int lane) {
    const int o = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
            acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
        }
        const float* wrow = Wimg + (32 * mo + o) * WS + 4 * h;
#pragma unroll
        for (int mi = 0; mi < KI; ++mi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + 32 * mi + 8 * q);
                acc = mfma32(a[0], X[mi][4 * q + 0], acc);
                acc = mfma32(a[1], X[mi][4 * q + 1], acc);
                acc = mfma32(a[2], X[mi][4 * q + 2], acc);
                acc = mfma32(a[3], X[mi][4 * q + 3], acc);
            }
        }
        Y[mo] = acc;
    }
}

// one output m-tile of Y = W * X (+bias): lets callers consume/retire tiles one at a time (shorter live ranges)
template <int KI, bool BIAS>
__device__ __forceinline__ f32x16 dense_mfma_tile(const float* __restrict__ Wimg, int WS, const float* __restrict__ bias,
                                                  const f32x16 (&X)[KI], int mo, int lane) {
    const int o = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
    const float* wrow = Wimg + (32 * mo + o) * WS + 4 * h;
#pragma unroll
    for (int mi = 0; mi < KI; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + 32 * mi + 8 * q);
            acc = mfma32(a[0], X[mi][4 * q + 0], acc);
            acc = mfma32(a[1], X[mi][4 * q + 1], acc);
            acc = mfma32(a[2], X[mi][4 * q + 2], acc);
            acc = mfma32(a[3], X[mi][4 * q + 3], acc);
        }
    }
    return acc;
}

// v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], C/D col = lane&15, row = 4*(lane>>4) + reg
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 8 values of row `row` of a [rows][kTS] image for the 16x16x4 contraction over the 32 samples of a tile:
// lane group g = lane>>4 owns samples 8g..8g+7, k-step s pairs samples {s, 8+s, 16+s, 24+s}
__device__ __forceinline__ void load_row8(const float* img, int row, int lane, float (&v)[8]) {
    const float* p = img + row * kTS + 8 * (lane >> 4);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(p), t1 = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3]; v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
}

// first layer: K = DP -> KS k-steps; xk[s] = obs[2s + (lane>>5)] of sample (lane&31)
template <int H1, int MO, int KS>
__device__ __forceinline__ void dense_first(const float* __restrict__ W1T, const float* __restrict__ bias,
                                            const float (&xk)[KS], f32x16 (&Y)[MO], int lane) {
    const int o = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
            acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma32(W1T[(2 * s + h) * H1 + 32 * mo + o], xk[s], acc);
        Y[mo] = acc;
    }
}

// stage-wise over the 16 registers of a tile so the five dependent ops of one element interleave with the other
// fifteen (the element-by-element form left s_nop bubbles after every v_exp/v_rcp: stamps, profiles/r01_v3).
// Element-wise loops, not f32x16 arithmetic: a vector expression is legalised into v_pk_add_f32 / v_pk_fma_f32 whatever -fno-slp-vectorize says, and beside MFMAs
// the packed forms are the slower ones (MI355X_MICROARCH.md; ppo_grad_pair_kernel, same box, alternating runs: 149.6 / 150.2 packed vs 151.1 / 152.8 TFLOP/s scalar)
__device__ __forceinline__ void tanh16(f32x16& x) {
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = __builtin_amdgcn_exp2f(x[i]);        // x is pre-scaled by kTanhScale
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = t[i] + 1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = __builtin_amdgcn_rcpf(t[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(-2.0f, t[i], 1.0f);
}
// tanh16 for the f16-piece kernels: the argument is x CIN (CIN = 1: none; undoes the operand scales of the product that made x), the result kActScale tanh —
// the scale rides in the constants of the last fma
template <bool SCALE_IN> __device__ __forceinline__ void tanh16_scaled(f32x16& x, float cin) {
    f32x16 t;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = __builtin_amdgcn_exp2f(SCALE_IN ? x[i] * cin : x[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = t[i] + 1.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = __builtin_amdgcn_rcpf(t[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(-2.0f * kActScale, t[i], kActScale);
}
// acc += s * x and acc += x on sixteen registers, as scalar v_fma_f32 / v_add_f32 (see tanh16)
__device__ __forceinline__ void fma16(f32x16& acc, float s, const f32x16& x) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(s, x[i], acc[i]);
}
__device__ __forceinline__ void add16(f32x16& acc, const f32x16& x) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc[i] + x[i];
}
template <int M> __device__ __forceinline__ void tanh_tiles(f32x16 (&X)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m) tanh16(X[m]);
}

// fast transcendental forms for the distribution heads: v_exp_f32 / v_log_f32 / v_rcp_f32 are 1-ulp units; the libm
// versions (and IEEE division) cost 10-30 dependent instructions each and made the actor head 312 VALU instructions
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// output layer on the VALU: out[o] = b3[o] + sum_j W3[o][j] * h2[j][sample]; each half-wave owns half the rows
template <int M, int O, int H2>
__device__ __forceinline__ void dense_out(const float* __restrict__ W3S, const float* __restrict__ b3,
                                          const f32x16 (&X)[M], float (&out)[O], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float p = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(W3S + o * H2 + 32 * m + 8 * q + 4 * h);
                p = fmaf(w[0], X[m][4 * q + 0], p); p = fmaf(w[1], X[m][4 * q + 1], p);
                p = fmaf(w[2], X[m][4 * q + 2], p); p = fmaf(w[3], X[m][4 * q + 3], p);
            }
        out[o] = p + __shfl_xor(p, 32) + b3[o];
    }
}

// full forward of one net for a 32-sample tile
template <int D, int H1, int H2, int O>
__device__ __forceinline__ void net_forward(const float* __restrict__ lds, const float (&xk)[FirstLayer<D>::KS], f32x16 (&h1)[H1 / 32],
                                            f32x16 (&h2)[H2 / 32], float (&out)[O], int lane) {
    using L = NetLds<D, H1, H2, O>;
    dense_first<H1, H1 / 32, FirstLayer<D>::KS>(lds + L::W1T, lds + L::B1, xk, h1, lane);
    tanh_tiles(h1);
    dense_mfma<H1 / 32, H2 / 32, true>(lds + L::W2S, L::WS1, lds + L::B2, h1, h2, lane);
    tanh_tiles(h2);
    dense_out<H2 / 32, O, H2>(lds + L::W3S, lds + L::B3, h2, out, lane);
}

// ---- wide nets (H > 64): W2 (H*H*4 bytes = 256 KB at H = 256) does not fit LDS next to anything else, so the two
// H x H operand streams live in global memory PRE-TILED in MFMA A-operand order, [(mo*MT + mi)*4 + q][lane][4]: one
// wave-instruction reads 1 KiB contiguous and the image (0.5 MB per net) stays L2-resident.  Small parts stay in LDS.
template <int D, int H, int O> struct NetLdsSmall {
    static constexpr int DP = FirstLayer<D>::DP, OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, B3 = W3S + O * H, END = B3 + OP;
};
template <int D, int H, int O>
__device__ inline void stage_net_small(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    using L = NetLdsSmall<D, H, O>;
    for (int i = tid; i < L::DP * H; i += nthreads) { const int o = i % H, k = i / H; lds[L::W1T + k * H + o] = k < D ? kTanhScale * P[n.w1 + o + k * H] : 0.0f; }
    for (int i = tid; i < H; i += nthreads) { lds[L::B1 + i] = kTanhScale * P[n.b1 + i]; lds[L::B2 + i] = kTanhScale * P[n.b2 + i]; }
    for (int i = tid; i < O * H; i += nthreads) { const int o = i % O, k = i / O; lds[L::W3S + o * H + k] = P[n.w3 + i]; }
    for (int i = tid; i < L::OP; i += nthreads) lds[L::B3 + i] = i < O ? P[n.b3 + i] : 0.0f;
}
// one output m-tile of Y = W * X with W streamed from the pre-tiled global image
template <int MT, bool BIAS>
__device__ __forceinline__ f32x16 dense_tile_global(const float* __restrict__ wimg, const float* __restrict__ bias, const f32x16 (&X)[MT], int mo, int lane) {
    const int h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
    const float* base = wimg + ((size_t)mo * MT * 4 * 64 + lane) * 4;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + (size_t)(mi * 4 + q) * 256);
            acc = mfma32(a[0], X[mi][4 * q + 0], acc);
            acc = mfma32(a[1], X[mi][4 * q + 1], acc);
            acc = mfma32(a[2], X[mi][4 * q + 2], acc);
            acc = mfma32(a[3], X[mi][4 * q + 3], acc);
        }
    }
    return acc;
}
// forward of a wide net for one 32-sample tile: h1 stays in registers (H/2 VGPRs), h2 is consumed m-tile by m-tile
template <int D, int H, int O>
__device__ __forceinline__ void net_forward_wide(const float* __restrict__ lds, const float* __restrict__ w2a, const float (&xk)[FirstLayer<D>::KS],
                                                 float (&out)[O], int lane) {
    using L = NetLdsSmall<D, H, O>;
    constexpr int MT = H / 32;
    const int h = lane >> 5;
    f32x16 h1[MT];
    dense_first<H, MT, FirstLayer<D>::KS>(lds + L::W1T, lds + L::B1, xk, h1, lane);
    tanh_tiles(h1);
    float part[O];
#pragma unroll
    for (int o = 0; o < O; ++o) part[o] = 0.f;
    // W2 streams from L2 as ONE linear sequence of MT*MT*4 fragments (pre-tiled image): each of the four fragment registers is refilled with the
    // fragment four places ahead right after the MFMAs that consumed it, across m-tile boundaries too, so no output tile starts on a cold load
    const float* fbase = w2a + (size_t)lane * 4;
    constexpr int F = MT * MT * 4;
    f32x4 af[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(fbase + (size_t)q * 256);
#pragma unroll 1
    for (int mo = 0; mo < MT; ++mo) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B2 + 32 * mo + 8 * q + 4 * h);
            acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = mfma32(af[q][0], h1[mi][4 * q + 0], acc); acc = mfma32(af[q][1], h1[mi][4 * q + 1], acc);
                acc = mfma32(af[q][2], h1[mi][4 * q + 2], acc); acc = mfma32(af[q][3], h1[mi][4 * q + 3], acc);
                int f = (mo * MT + mi) * 4 + q + 4; f = f < F ? f : F - 4 + q;       // the tail re-reads the last fragments (in bounds, unused)
                af[q] = *reinterpret_cast<const f32x4*>(fbase + (size_t)f * 256);
            }
        }
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

// ---- transposes through a per-wave LDS image [H][kTS] ---------------------------------------------------
// store X (C/D layout) as img[hidden row][sample col]
template <int M> __device__ __forceinline__ void store_image(float* img, const f32x16 (&X)[M], int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) img[(32 * m + rowfn(r, h)) * kTS + c] = X[m][r];
}
// read m-tile m of an image as an MFMA A/B operand set for a contraction over samples:
// lane (i = lane&31, h) gets regs kk = 0..15 = img[32m + i][16h + kk]   (k-step kk pairs samples kk and 16+kk)
__device__ __forceinline__ f32x16 load_operand(const float* img, int m, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float* p = img + (32 * m + i) * kTS + 16 * h;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
        v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
    return v;
}
__device__ __forceinline__ f32x16 mfma_outer(const f32x16& a, const f32x16& b, f32x16 acc) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = mfma32(a[kk], b[kk], acc);
    return acc;
}
__device__ __forceinline__ float sum16(const f32x16& v) {
    float s0 = (v[0] + v[1]) + (v[2] + v[3]), s1 = (v[4] + v[5]) + (v[6] + v[7]);
    float s2 = (v[8] + v[9]) + (v[10] + v[11]), s3 = (v[12] + v[13]) + (v[14] + v[15]);
    return (s0 + s1) + (s2 + s3);
}
// sum over the 32 lanes of each half-wave (result valid in every lane of the half)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

// workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() also waits for every global store in flight (vmcnt(0)), which a
// kernel that exchanges through LDS but streams its results to global memory must not do once per step (rollout_duo_kernel, ppo_update_small_kernel)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DPP move: lanes disabled by the bank mask (banks = groups of 4 lanes within a row of 16) keep `old`
template <int CTRL, int BANK = 0xf> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
// sums of the 16 registers of x over the 32 lanes of each half-wave, as a complete reduce-scatter on the VALU: lane l returns the sum over its half of x[l & 15] (every
// value twice per half: lanes l and l ^ 16).  Each step halves the register count while it adds — lane bit b keeps the registers whose index has bit b set and receives
// its partner's copy of them (quad_perm for lane bits 0 and 1, row_shl / row_shr:4 under bank masks for bit 2, row_ror:8 for bit 3) — and the two rows of a half
// meet in one v_permlane16_swap: ~52 VALU and no LDS-crossbar permute (first form of round 3: scatter to four registers, then butterflies and four ds_bpermute:
// 60 VALU + 4 permutes with their waits).  tools/micro/dpp_reduce2.hip checks it against a plain sum.  EXEC must be all ones.
__device__ __forceinline__ float half_reduce16_lane(const f32x16& x, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float y[8], z[4], u[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // (the two elements as opaque scalars: otherwise LLVM folds `b0 ? x[2i+1] : x[2i]` into a vector element with a VARIABLE index = a 16-way compare / select chain)
        float e = x[2 * i], o = x[2 * i + 1];
        asm volatile("" : "+v"(e), "+v"(o));
        const float keep = b0 ? o : e, send = b0 ? e : o;
        y[i] = keep + dpp_mov<0xB1>(0.f, send);                                       // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float keep = b1 ? y[2 * i + 1] : y[2 * i], send = b1 ? y[2 * i] : y[2 * i + 1]; z[i] = keep + dpp_mov<0x4E>(0.f, send); }   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? z[2 * i + 1] : z[2 * i], send = b2 ? z[2 * i] : z[2 * i + 1];
        float t = dpp_mov<0x104, 0x5>(0.f, send);          // row_shl:4 -> the lanes of banks 0 and 2 receive lane + 4
        t = dpp_mov<0x114, 0xa>(t, send);                  // row_shr:4 -> the lanes of banks 1 and 3 receive lane - 4
        u[i] = keep + t;
    }
    const float keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
    const float v = keep + dpp_mov<0x128>(0.f, send);      // row_ror:8 = lane ^ 8
    const unsigned q = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(q, q, false, false);               // {even row's value in both rows of the half, odd row's value in both}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace dril

// dril_gemm.hip — generic strided fp32 contraction on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate); see dril_gemm.h.
// Written for the SAC update (docs/sac.md: ~25 small dense contractions per gradient step) and reused by the generic on-policy path.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <mutex>
#include <vector>
#include <utility>

#include "dril_device.h"
#include "dril_gemm.h"

namespace dril {

namespace {

// =================================================================================================================
// generic strided contraction  C[z](M x N) = epi(alpha * A[z](M x K) . B[z](K x N) + bias[z](M))
// =================================================================================================================

__device__ __forceinline__ void load_operand4(const float* __restrict__ P, int idx, int lim, int s_idx, int s_k, int k0, int K, int vec, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (idx >= lim || k0 >= K) return;
    if (vec) {
        const float4 t = *reinterpret_cast<const float4*>(P + (size_t)idx * s_idx + k0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) if (k0 + t < K) v[t] = P[(size_t)idx * s_idx + (size_t)(k0 + t) * s_k];
    }
}

// MFMA operand slots: A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; step t of chunk q contracts k = 8q + 4h + t
// for the half-wave h, so a k-major operand is one float4 per lane per chunk.  D: col = lane & 31, row = rowfn(r, lane >> 5).
// Two shapes of the same kernel:
//   SPLIT  (few output tiles: the 256-sample contractions of update!) — the 8 waves of a workgroup split the chunks of ONE 32x32 tile
//          and the partial tiles are summed through LDS in fixed wave order (deterministic);
//   !SPLIT (many tiles: the 4096-env actor forward of the collection) — the 8 waves take 8 neighbouring column tiles of the same
//          row tile (the weight operand is shared through L1) and each contracts the whole K.
// Every wave first issues ALL operand loads of up to kGemmDepth chunks and only then runs their MFMAs (one exposed L2 round trip per
// kGemmDepth chunks instead of one per chunk).  The epilogue goes through LDS so that global stores run along C's unit-stride axis
// (m): a wave writes 2 x 128 contiguous bytes per instruction instead of 64 scattered words.
constexpr int kGemmWaves = 8, kGemmDepth = 8, kRedStride = 65;
// the activations beyond tanh / relu (generic PPO path only) live in ONE out-of-line function: inlined into every epilogue of every instantiation their libm
// expansions grew this file's code by 45 % and cost the SAC collection forward 8 us per step (end of round 3, same-box A/B: 64.8 -> 72.8 us); SAC never calls it
__device__ __noinline__ float gemm_epilogue_rare(int epi, float v, float y) {
    if (epi == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    else if (epi == EPI_ELU) v = v > 0.f ? v : expm1f(v);
    else if (epi == EPI_LEAKY) v = v > 0.f ? v : 0.01f * v;
    else if (epi == EPI_SOFTPLUS) v = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
    else if (epi == EPI_MASK_SIGMOID) v *= y * (1.0f - y);
    else if (epi == EPI_MASK_ELU) v *= y > 0.f ? 1.0f : y + 1.0f;                 // alpha e^x = elu(x) + alpha
    else if (epi == EPI_MASK_LEAKY) v *= y > 0.f ? 1.0f : 0.01f;
    else if (epi == EPI_MASK_SOFTPLUS) v *= -expm1f(-y);                          // sigmoid(x) = 1 - e^(-softplus(x)); expm1: no cancellation for x << 0, where y = softplus(x) ~ e^x is tiny
    else if (epi == EPI_GELU) { const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v); v = 0.5f * v * (1.0f + tanhf(u)); }   // NNlib.gelu (tanh form)
    else if (epi == EPI_SWISH) v = v / (1.0f + expf(-v));
    else if (epi == EPI_MASK_GELU) {                                               // y = the pre-activation x
        const float u = 0.7978845608028654f * (y + 0.044715f * y * y * y), t = tanhf(u);
        v *= 0.5f * (1.0f + t) + 0.5f * y * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * y * y);
    } else if (epi == EPI_MASK_SWISH) { const float sg = 1.0f / (1.0f + expf(-y)); v *= sg * (1.0f + y * (1.0f - sg)); }
    return v;
}
__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, float b, float y) {
    v = v * g.alpha + b;
    if (g.epi == EPI_RELU) v = relu_nan(v);
    else if (g.epi == EPI_TANH) v = tanhf(v);
    else if (g.epi == EPI_MASK_RELU) v = y > 0.f ? v : 0.f;
    else if (g.epi == EPI_MASK_TANH) v *= 1.0f - y * y;
    else if (g.epi != EPI_NONE) v = gemm_epilogue_rare(g.epi, v, y);
    return v;
}
// one 32 x 32 output tile out of the LDS transposition buffer: lane -> 16 elements (row ml = lane & 31 fixed, 16 columns), stores along C's unit-stride axis.
// Two phases on purpose: ALL bias / aux operand loads of the 16 elements are issued before the first store.  Interleaved (load, compute, store per element)
// the compiler may not move element i+1's loads above element i's store, and every element then waits a full memory round trip: measured, that — not the
// contraction — set the 190 us of a 512 x 32768 x 512 reverse-pass contraction in every blocking and on both MFMA types (docs/external_envs.md)
template <class F>
__device__ __forceinline__ void store_tile(const GemmArgs& g, float* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ aux,
                                           int m_base, int n_base, int lane, F&& value) {
    const int ml = lane & 31, mm = m_base + ml;
    const bool m_ok = mm < g.M;
    const float b = (bias && m_ok) ? bias[mm] : 0.f;
    float* __restrict__ Zp = g.zout ? g.zout + (C - g.C) : nullptr;                        // same batch offset as C
    const bool need_aux = aux && epi_reads_aux(g.epi);
    float y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nn = n_base + (lane >> 5) + 2 * i;
        y[i] = (need_aux && m_ok && nn < g.N) ? aux[(size_t)mm * g.sCm + (size_t)nn * g.sCn] : 0.f;
    }
    const int rr = (ml & 3) + 4 * (ml >> 3), lb = 32 * ((ml >> 2) & 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nl = (lane >> 5) + 2 * i, nn = n_base + nl;
        if (!m_ok || nn >= g.N) continue;
        const size_t ci = (size_t)mm * g.sCm + (size_t)nn * g.sCn; const float raw = value(rr, nl + lb);
        if (Zp) Zp[ci] = raw * g.alpha + b;
        C[ci] = gemm_epilogue(g, raw, b, y[i]);
    }
}
template <bool SPLIT>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int z, float (&red)[kGemmWaves][16][kRedStride], int bx = (int)blockIdx.x, int by = (int)blockIdx.y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const float* __restrict__ A = g.A + (size_t)z * g.zA;
    const float* __restrict__ B = g.B + (size_t)(z / g.zdivB) * g.zB;
    const int tile_n = SPLIT ? by : by * kGemmWaves + wave;
    if (SPLIT && (bx * 32 >= g.M || tile_n * 32 >= g.N)) return;                               // pair launches: the grid covers the larger problem
    const int m = bx * 32 + c, n = tile_n * 32 + c;
    const int Q = (g.K + 7) >> 3, Qw = SPLIT ? (Q + kGemmWaves - 1) / kGemmWaves : Q, q0 = SPLIT ? wave * Qw : 0, q1 = min(Q, q0 + Qw);
    const bool ones = g.ones_n && n == g.N - 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q = q0; q < q1; q += kGemmDepth) {
        float a[kGemmDepth][4], b[kGemmDepth][4];
#pragma unroll
        for (int u = 0; u < kGemmDepth; ++u) {
            const int k0 = 8 * (q + u) + 4 * h;
            if (g.dbg & 16) { for (int t = 0; t < 4; ++t) { a[u][t] = 1.f; b[u][t] = 1.f; } }       // ablation (tools/micro/sac_gemm_shapes.hip): no operand loads
            else if (q + u < q1) {
                load_operand4(A, m, g.M, g.sAm, g.sAk, k0, g.K, g.vecA, a[u]);
                if (ones) { for (int t = 0; t < 4; ++t) b[u][t] = k0 + t < g.K ? 1.f : 0.f; }
                else load_operand4(B, n, g.N, g.sBn, g.sBk, k0, g.K, g.vecB, b[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kGemmDepth; ++u)
            if (q + u < q1) {
                if (g.dbg & 32) { acc[u] += a[u][0] + b[u][1]; continue; }                          // ablation: no MFMA
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = mfma32(a[u][t], b[u][t], acc);
            }
    }
    if (g.dbg & 64) { if (acc[0] == 123.456f) g.C[0] = 1.f; return; }                               // ablation: no reduction / epilogue
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float* __restrict__ C = g.C + (size_t)z * g.zC;
    const float* __restrict__ bias = g.bias ? g.bias + (size_t)z * g.zBias : nullptr;
    const float* __restrict__ aux = g.aux ? g.aux + (size_t)z * g.zAux : nullptr;
    // element e of a 32x32 tile: m_local = e & 31 (fastest, C's unit stride), n_local = e >> 5; it sits in register rr of lane ll
    if (SPLIT) {
        const bool need_aux = aux && epi_reads_aux(g.epi);
        const int ml = threadIdx.x & 31, mm = bx * 32 + ml, rr = (ml & 3) + 4 * (ml >> 3), lb = 32 * ((ml >> 2) & 1);
        const float b = (bias && mm < g.M) ? bias[mm] : 0.f;
        float y[2]; bool ok[2]; size_t ci[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                                          // both operand loads before the first store (see store_tile)
            const int nn = tile_n * 32 + (int)(threadIdx.x >> 5) + 16 * i;
            ok[i] = mm < g.M && nn < g.N; ci[i] = (size_t)mm * g.sCm + (size_t)nn * g.sCn;
            y[i] = (need_aux && ok[i]) ? aux[ci[i]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!ok[i]) continue;
            const int ll = (int)(threadIdx.x >> 5) + 16 * i + lb;
            float v = red[0][rr][ll];
#pragma unroll
            for (int w = 1; w < kGemmWaves; ++w) v += red[w][rr][ll];                          // fixed order
            if (g.zout) (g.zout + (C - g.C))[ci[i]] = v * g.alpha + b;
            C[ci[i]] = gemm_epilogue(g, v, b, y[i]);
        }
    } else store_tile(g, C, bias, aux, bx * 32, tile_n * 32, lane, [&](int rr, int ll) { return red[wave][rr][ll]; });
}
// ---- the split-K shape with COALESCED operand loads (round 4) -----------------------------------------------------------------------------------------
// gemm_body<true> lets every lane fetch its own MFMA operand straight from memory: 16 bytes out of 64 different cache lines per wave instruction (a k-contiguous
// operand: 32 rows 2 KB apart) or 4-byte words k-strided.  Ablated (tools/micro/sac_gemm_shapes.hip, DRIL_GEMM_DBG bits), those loads are HALF of a launch — the four-net
// 512 x 256 x 512 forward of update!: 17.4 us = 4.1 launch floor + 8.6 operand loads + 4.6 MFMA + 1.7 epilogue, purely additive because all waves of a workgroup
// (and, the launch being one round, of the chip) sit in the same phase.  Here the workgroup's 512 threads read the tile's operand rows the way memory stores them —
// consecutive lanes, consecutive 16 bytes: every wave instruction covers whole 128-byte lines — kLdsKc contraction steps at a time into LDS ([row][k], stride
// kLdsKc + 4 floats: the ds_read_b128 operand fetch of lane (row c, k-half h) is conflict-free), the NEXT pass's lines already in flight (registers) under this
// pass's MFMAs.  Same v_mfma_f32_32x32x2_f32 products; wave w contracts k in [32 w, 32 w + 32) of every pass, partial tiles summed through LDS in fixed wave
// order (deterministic).  The epilogue's buffer overlays the operand images.
constexpr int kLdsMinK = 64;                                                                  // shorter contractions (a first layer over 4 inputs) keep the direct-load body
constexpr int kLdsKc = 256, kLdsStride = kLdsKc + 4, kLdsTile = 32 * kLdsStride;              // floats
constexpr size_t kLdsBytes = sizeof(float) * 2 * kLdsTile;                                    // 66 560 B: two workgroups per CU
static_assert(2 * kLdsTile >= kGemmWaves * 16 * kRedStride, "the epilogue buffer overlays the operand images");
// one operand tile (32 rows x kLdsKc steps) of pass kc: global -> registers (issue) and registers -> LDS (commit).  Three access patterns, picked per operand
// (uniform over the launch): k-contiguous rows (a weight read transposed, an activation row per sample), row-contiguous (a column-major weight, the activation operand
// of a weight gradient: 32 consecutive rows of one k are one 128-byte line), anything else element by element.
struct LdsOperand { const float* P; int row0, lim, s_row, s_k, ones_row; };                    // ones_row: this row (global index) is the synthetic ones column, -1 none
// MODE 0 k-contiguous float4, 1 row-contiguous float4 (compile-time: one access pattern's addresses live at a time; an operand that fits neither keeps the direct-load body)
template <int MODE>
__device__ __forceinline__ void lds_issue(const LdsOperand& o, int kc, int K, int tid, float4 (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + 512 * j;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const int r = f >> 6, k = kc + ((f & 63) << 2), row = o.row0 + r;
            if (row < o.lim && k < K) t = row == o.ones_row ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(o.P + (size_t)row * o.s_row + k);     // K % 4 == 0 (mode 0)
        } else {
            const int k = kc + (f >> 3), row = o.row0 + ((f & 7) << 2);
            if (k < K) {
                const float* src = o.P + (size_t)k * o.s_k + row;
                if (row + 3 < o.lim && (o.ones_row < row || o.ones_row > row + 3)) t = *reinterpret_cast<const float4*>(src);
                else {
                    float e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = row + q == o.ones_row ? 1.f : (row + q < o.lim ? src[q] : 0.f);
                    t = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
        }
        v[j] = t;
    }
}
template <int MODE>
__device__ __forceinline__ void lds_commit(int tid, const float4 (&v)[4], float* __restrict__ img) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = tid + 512 * j;
        if (MODE == 0) *reinterpret_cast<float4*>(img + (f >> 6) * kLdsStride + ((f & 63) << 2)) = v[j];
        else {
            const int k = f >> 3, r = (f & 7) << 2;
            img[(r + 0) * kLdsStride + k] = v[j].x; img[(r + 1) * kLdsStride + k] = v[j].y; img[(r + 2) * kLdsStride + k] = v[j].z; img[(r + 3) * kLdsStride + k] = v[j].w;
        }
    }
}
template <int AM, int BM>
__device__ __forceinline__ void gemm_body_lds(const GemmArgs& g, int z, float* __restrict__ smem, int bx, int by) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    if (bx * 32 >= g.M || by * 32 >= g.N) return;                                               // pair / multi launches: the grid covers the larger problem (uniform per workgroup)
    float* __restrict__ As = smem; float* __restrict__ Bs = smem + kLdsTile;
    const int n_real = g.N - (g.ones_n ? 1 : 0);
    const LdsOperand oa{g.A + (size_t)z * g.zA, bx * 32, g.M, g.sAm, g.sAk, -1};
    const LdsOperand ob{g.B + (size_t)(z / g.zdivB) * g.zB, by * 32, g.N, g.sBn, g.sBk, g.ones_n ? n_real : -1};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 va[4], vb[4];
    if (!(g.dbg & 16)) { lds_issue<AM>(oa, 0, g.K, tid, va); lds_issue<BM>(ob, 0, g.K, tid, vb); }
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = vb[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    for (int kc = 0; kc < g.K; kc += kLdsKc) {
        lds_commit<AM>(tid, va, As); lds_commit<BM>(tid, vb, Bs);
        __syncthreads();
        if (kc + kLdsKc < g.K && !(g.dbg & 16)) { lds_issue<AM>(oa, kc + kLdsKc, g.K, tid, va); lds_issue<BM>(ob, kc + kLdsKc, g.K, tid, vb); }   // next pass in flight under this pass's MFMAs
        if (kc + 32 * wave < g.K) {                                                              // (a ragged last pass leaves the upper waves without work)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(As + c * kLdsStride + 32 * wave + 8 * q + 4 * h);
                const f32x4 b = *reinterpret_cast<const f32x4*>(Bs + c * kLdsStride + 32 * wave + 8 * q + 4 * h);
                if (g.dbg & 32) { acc[q] += a[0] + b[1]; continue; }
                acc = mfma32(a[0], b[0], acc); acc = mfma32(a[1], b[1], acc); acc = mfma32(a[2], b[2], acc); acc = mfma32(a[3], b[3], acc);
            }
        }
        __syncthreads();
    }
    if (g.dbg & 64) { if (acc[0] == 123.456f) g.C[0] = 1.f; return; }
    float (*red)[16][kRedStride] = reinterpret_cast<float (*)[16][kRedStride]>(smem);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float* __restrict__ C = g.C + (size_t)z * g.zC;
    const float* __restrict__ bias = g.bias ? g.bias + (size_t)z * g.zBias : nullptr;
    const float* __restrict__ aux = g.aux ? g.aux + (size_t)z * g.zAux : nullptr;
    const bool need_aux = aux && epi_reads_aux(g.epi);
    const int ml = threadIdx.x & 31, mm = bx * 32 + ml, rr = (ml & 3) + 4 * (ml >> 3), lb = 32 * ((ml >> 2) & 1);
    const float bv = (bias && mm < g.M) ? bias[mm] : 0.f;
    float y[2]; bool ok[2]; size_t ci[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                                              // both operand loads before the first store (see store_tile)
        const int nn = by * 32 + (int)(threadIdx.x >> 5) + 16 * i;
        ok[i] = mm < g.M && nn < g.N; ci[i] = (size_t)mm * g.sCm + (size_t)nn * g.sCn;
        y[i] = (need_aux && ok[i]) ? aux[ci[i]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!ok[i]) continue;
        const int ll = (int)(threadIdx.x >> 5) + 16 * i + lb;
        float v = red[0][rr][ll];
#pragma unroll
        for (int w = 1; w < kGemmWaves; ++w) v += red[w][rr][ll];                              // fixed order
        if (g.zout) (g.zout + (C - g.C))[ci[i]] = v * g.alpha + bv;
        C[ci[i]] = gemm_epilogue(g, v, bv, y[i]);
    }
}
__device__ __forceinline__ void gemm_body_lds_any(const GemmArgs& g, int z, float* smem, int bx, int by) {
    if (g.ldsA == 0) { if (g.ldsB == 0) gemm_body_lds<0, 0>(g, z, smem, bx, by); else gemm_body_lds<0, 1>(g, z, smem, bx, by); }
    else { if (g.ldsB == 0) gemm_body_lds<1, 0>(g, z, smem, bx, by); else gemm_body_lds<1, 1>(g, z, smem, bx, by); }
}
__global__ __launch_bounds__(64 * kGemmWaves, 4) void sac_gemm_lds_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds_smem[];
    gemm_body_lds_any(g, blockIdx.z, lds_smem, blockIdx.x, blockIdx.y);
}
template <bool SPLIT>
__global__ __launch_bounds__(64 * kGemmWaves) void sac_gemm_kernel(GemmArgs g) {
    __shared__ float red[kGemmWaves][16][kRedStride];
    gemm_body<SPLIT>(g, blockIdx.z, red);
}
// Throughput shape (the 4096-env actor forward of the collection): classic LDS-tiled contraction.  A workgroup owns a 32 (m) x 256 (n) output block,
// one 32 x 32 tile per wave; per 32-deep k-chunk it stages the weight chunk (m-major in memory, read coalesced along m) and the activation chunk (each
// sample row contiguous: 8 threads read one row's 128 B) into LDS with b128 stores and the waves read their MFMA operands back as one ds_read_b128 per
// lane per 8 k.  The plain !SPLIT shape read every activation row straight from L2, 16 B per lane = 32 cache lines per wave instruction (56 us for
// 512 x 512 x 4096); the next chunk's global loads are issued before the current chunk's MFMAs (register double buffer, one barrier pair per chunk).
constexpr int kBigKc = 32, kBigStride = kBigKc + 4, kBigNStride = 32 * kGemmWaves + 4;
// Operand layouts in memory (element strides, host-checked): A is m-contiguous (a column-major weight: forward layers) or, with AK, k-contiguous
// (the same weight read transposed: W' dz of the reverse pass); B is k-contiguous (one activation row per sample) or, with BN, n-contiguous (the
// activation operand of a weight gradient, contraction over samples; its last column may be the synthetic ones column, g.ones_n).  Whatever the
// layout, every global load instruction of a wave covers whole 64..128-byte runs and the chunk lands in LDS so that the MFMA operand reads are
// conflict-free: k-contiguous rows [n][k + 4 pad] read back as ds_read_b128, or — BN — the chunk stays [k][n + 4 pad] and is read as 4 x ds_read_b32.
template <bool AK, bool BN>
__global__ __launch_bounds__(64 * kGemmWaves) void sac_gemm_big_kernel(GemmArgs g) {
    __shared__ float red[kGemmWaves][16][kRedStride];
    __shared__ __attribute__((aligned(16))) float As[32][kBigStride];
    __shared__ __attribute__((aligned(16))) float Xs[32 * kGemmWaves * kBigStride];                  // [256][36] (k-contiguous B) or [32][260] (BN)
    static_assert(32 * kGemmWaves * kBigStride >= kBigKc * kBigNStride, "the n-major chunk image must fit the same LDS block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5, z = blockIdx.z;
    const float* __restrict__ A = g.A + (size_t)z * g.zA;
    const float* __restrict__ B = g.B + (size_t)(z / g.zdivB) * g.zB;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32 * kGemmWaves;
    // loader roles.  B k-contiguous: 256 rows x 8 float4 -> 4 float4 per thread (row i >> 3, k (i & 7) * 4).  BN: 32 k-rows x 64 float4 along n -> 4 per
    // thread (k = (tid >> 6) + 8 j, n = (tid & 63) * 4).  A m-contiguous: 32 k x 32 m floats, 2 per thread (m = tid & 31, k = tid >> 5 and + 16).
    // AK: 32 m-rows x 16 float2 along k, 1 float2 per thread (m = tid >> 4, k = (tid & 15) * 2).
    int xr[4], xk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = tid + 512 * j; if (BN) { xr[j] = (tid >> 6) + 8 * j; xk[j] = (tid & 63) * 4; } else { xr[j] = i >> 3; xk[j] = (i & 7) * 4; } }
    const int am = AK ? tid >> 4 : tid & 31, ak = AK ? (tid & 15) * 2 : tid >> 5;
    const int n_real = g.N - (g.ones_n ? 1 : 0);                                                       // columns that exist in memory
    float4 xv[4]; float av[2];
    auto gload = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BN) {
                const int n = n0 + xk[j]; const float* row = B + (size_t)(kc + xr[j]) * g.sBk;
                if (kc + xr[j] >= g.K) xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);                       // ragged last chunk
                else if (g.vecBn && n + 3 < n_real) xv[j] = *reinterpret_cast<const float4*>(row + n);
                else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (n + e < n_real) ? row[n + e] : ((g.ones_n && n + e == n_real) ? 1.0f : 0.0f);
                    xv[j] = make_float4(t[0], t[1], t[2], t[3]);
                }
            } else {
                const int n = n0 + xr[j];
                xv[j] = (n < g.N && kc + xk[j] < g.K) ? *reinterpret_cast<const float4*>(B + (size_t)n * g.sBn + kc + xk[j]) : make_float4(0.f, 0.f, 0.f, 0.f);   // K % 4 == 0 (vecB)
            }
        }
        const int mg = m0 + am;
        if (AK) { av[0] = (mg < g.M && kc + ak < g.K) ? A[(size_t)mg * g.sAm + kc + ak] : 0.f; av[1] = (mg < g.M && kc + ak + 1 < g.K) ? A[(size_t)mg * g.sAm + kc + ak + 1] : 0.f; }
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j) av[j] = (mg < g.M && kc + ak + 16 * j < g.K) ? A[(size_t)mg + (size_t)(kc + ak + 16 * j) * g.sAk] : 0.f;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    gload(0);
    for (int kc = 0; kc < g.K; kc += kBigKc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&Xs[BN ? xr[j] * kBigNStride + xk[j] : xr[j] * kBigStride + xk[j]]) = xv[j];
        if (AK) { As[am][ak] = av[0]; As[am][ak + 1] = av[1]; }
        else {
#pragma unroll
            for (int j = 0; j < 2; ++j) As[am][ak + 16 * j] = av[j];
        }
        __syncthreads();
        if (kc + kBigKc < g.K) gload(kc + kBigKc);                               // next chunk in flight under this chunk's MFMAs
#pragma unroll
        for (int q = 0; q < kBigKc / 8; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&As[c][8 * q + 4 * h]);
            f32x4 b;
            if (BN) {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t] = Xs[(8 * q + 4 * h + t) * kBigNStride + 32 * wave + c];
            } else b = *reinterpret_cast<const f32x4*>(&Xs[(32 * wave + c) * kBigStride + 8 * q + 4 * h]);
            acc = mfma32(a[0], b[0], acc); acc = mfma32(a[1], b[1], acc); acc = mfma32(a[2], b[2], acc); acc = mfma32(a[3], b[3], acc);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float* __restrict__ C = g.C + (size_t)z * g.zC;
    const float* __restrict__ bias = g.bias ? g.bias + (size_t)z * g.zBias : nullptr;
    const float* __restrict__ aux = g.aux ? g.aux + (size_t)z * g.zAux : nullptr;
    const int tile_n = (int)blockIdx.y * kGemmWaves + wave;
    store_tile(g, C, bias, aux, m0, tile_n * 32, lane, [&](int rr, int ll) { return red[wave][rr][ll]; });
}
// ---- the same LDS-tiled contraction on the bf16 matrix cores, fp32-equivalent by operand splitting ------------------------------------------------
// Every staged f32 operand is cut into three bf16 pieces x = hi + mid + lo (upper 16 bits, exact remainder, twice: exact for a 24-bit mantissa) when
// its chunk is written to LDS — once per element, whatever number of waves reads it — and a k16 step of a tile is six v_mfma_f32_32x32x16_bf16
// (hi.hi hi.mid mid.hi mid.mid hi.lo lo.hi; the dropped partial products are <= 2^-23 relative; f32 accumulate).  profiles/r01_bf16_split_microbench.md:
// 2.0x the rate of v_mfma_f32_32x32x2_f32 with pre-split operands, max error 1.0e-7 vs 1.5e-7 for the f32 MFMA chain — not a precision reduction.
// LDS image per piece and k16 step: [h = k-half][row][8 bf16] — lane (row c, half h) reads its MFMA operand as ONE ds_read_b128 and consecutive lanes
// read consecutive 16 bytes (conflict-free).  The epilogue's transposition buffer overlays the operand images.
// bf16x8 / u32x4 / split3_pair: dril_device.h
// MB = m-tiles per wave: the workgroup's output block is (32 MB) x 256.  One thin 32 x 256 block re-reads the activation chunk for every 32 output
// rows — at hidden 512 the LDS-tiled kernels moved 6 TB/s through L2 and the bf16 form was no faster than the f32 one (87 vs 90 TFLOP/s); with
// MB = 4 the same chunk feeds four tiles (43.7 FLOP per staged byte instead of 14.5) and the split is amortised over four times the MFMAs
// WM = waves along m (1 or 2): with WM = 2 the workgroup spans 64 MB output rows and 128 activation rows — at MB = 8 that is ALL 512 rows of a hidden-512
// layer, so every activation row is staged exactly once per contraction (weights are the re-read operand, and they live in L2)
template <bool AK, bool BN, int MB, int WM>
__global__ __launch_bounds__(64 * kGemmWaves, 4) void sac_gemm_split_kernel(GemmArgs g) {   // 4 waves per SIMD = two workgroups per CU (<= 128 VGPRs): they hide each other's staging / epilogue phases
    constexpr int WN = kGemmWaves / WM, kRowsA = 32 * MB * WM, kSplitRowsX = 32 * WN;
    constexpr int kSplitAImg = 2 * 2 * kRowsA * 8, kSplitXImg = 2 * 2 * kSplitRowsX * 8;           // bf16 elements per piece: [k16 step][half][row][8]
    constexpr int NGX = kSplitRowsX / 64;                                                         // (row, 4 k) groups per thread of the activation chunk
    constexpr int NPA = kRowsA / 32;                                                              // element pairs per thread of the weight chunk
    constexpr bool kRedOnA = 3 * kSplitAImg > 3 * kSplitXImg;                                     // the f32 epilogue buffer (33 KB) overlays the larger image block
    __shared__ __attribute__((aligned(16))) unsigned short Ap[3 * kSplitAImg];
    __shared__ __attribute__((aligned(16))) unsigned short Xp[3 * kSplitXImg];
    static_assert(sizeof(unsigned short) * 3 * (kRedOnA ? kSplitAImg : kSplitXImg) >= sizeof(float) * kGemmWaves * 16 * kRedStride, "epilogue buffer must fit the operand images");
    float (*red)[16][kRedStride] = reinterpret_cast<float (*)[16][kRedStride]>(kRedOnA ? Ap : Xp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5, z = blockIdx.z;
    const int wm = wave / WN, wn = wave % WN;
    const float* __restrict__ A = g.A + (size_t)z * g.zA;
    const float* __restrict__ B = g.B + (size_t)(z / g.zdivB) * g.zB;
    const int m0 = blockIdx.x * kRowsA, n0 = blockIdx.y * kSplitRowsX;
    // loader roles: every thread owns NGX groups of (one n-row, 4 consecutive k) = one 8-byte LDS store per piece and group.  B k-contiguous: one float4 per
    // group (8 lanes read one row's 128 B).  BN (n-contiguous): consecutive lanes take consecutive n and read the 4 k-rows as 4 coalesced scalar loads
    int xr[NGX], xk[NGX];
#pragma unroll
    for (int j = 0; j < NGX; ++j) { const int i = tid + 512 * j; if (BN) { xr[j] = tid % kSplitRowsX; xk[j] = (tid / kSplitRowsX + (512 / kSplitRowsX) * j) * 4; } else { xr[j] = i >> 3; xk[j] = (i & 7) * 4; } }
    // A chunk = (32 MB) m x 32 k.  m-contiguous: element e = tid + 512 j, m = e % rows, k = e / rows (consecutive lanes, consecutive m), two per store
    // pair (j, j + MB): k and k + 16 share an 8-k group?  no — stored as single bf16 each.  AK: pair e = tid + 512 j: m = e >> 4, k = (e & 15) * 2
    const int n_real = g.N - (g.ones_n ? 1 : 0);
    float4 xv[NGX]; float av[2 * NPA];
    auto gload = [&](int kc) {
#pragma unroll
        for (int j = 0; j < NGX; ++j) {
            if (BN) {
                const int n = n0 + xr[j];
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kc + xk[j] + e;
                    t[e] = (k < g.K && n < n_real) ? B[(size_t)k * g.sBk + n] : ((k < g.K && g.ones_n && n == n_real) ? 1.0f : 0.0f);
                }
                xv[j] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                const int n = n0 + xr[j];
                xv[j] = (n < g.N && kc + xk[j] < g.K) ? *reinterpret_cast<const float4*>(B + (size_t)n * g.sBn + kc + xk[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            if (AK) {
                const int e = tid + 512 * j, am = e >> 4, ak = (e & 15) * 2, mg = m0 + am;
                av[2 * j] = (mg < g.M && kc + ak < g.K) ? A[(size_t)mg * g.sAm + kc + ak] : 0.f;
                av[2 * j + 1] = (mg < g.M && kc + ak + 1 < g.K) ? A[(size_t)mg * g.sAm + kc + ak + 1] : 0.f;
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = tid + 512 * (2 * j + u), am = e % kRowsA, ak = e / kRowsA, mg = m0 + am;
                    av[2 * j + u] = (mg < g.M && kc + ak < g.K) ? A[(size_t)mg + (size_t)(kc + ak) * g.sAk] : 0.f;
                }
            }
        }
    };
    // element (row, k in 0..31) of a piece image: [k >> 4][(k >> 3) & 1][row][k & 7]
    auto xoff = [](int row, int k) { return (((k >> 4) * 2 + ((k >> 3) & 1)) * kSplitRowsX + row) * 8 + (k & 7); };
    auto aoff = [](int row, int k) { return (((k >> 4) * 2 + ((k >> 3) & 1)) * kRowsA + row) * 8 + (k & 7); };
    f32x16 acc[MB];
#pragma unroll
    for (int t = 0; t < MB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // chunk order staggered by workgroup: all workgroups walking k = 0, 32, 64, ... in step touch, at any moment, only the 128-byte column block
    // kc of every (power-of-two strided) operand row — a fraction of the memory channels; starting each workgroup at a different chunk spreads them
    const int nchunks = (g.K + kBigKc - 1) / kBigKc, c0 = (int)((blockIdx.y * 7 + blockIdx.x * 3) % nchunks);
    gload(c0 * kBigKc);
    for (int ci = 0; ci < nchunks; ++ci) {
        const int cn = c0 + ci + 1 >= nchunks ? c0 + ci + 1 - nchunks : c0 + ci + 1;   // next chunk (wraps)
        if (!(g.dbg & 8)) {
#pragma unroll
        for (int j = 0; j < NGX; ++j) {                                          // 4 consecutive k of one n-row (inside one 8-k group): one 8-byte store per piece
            unsigned h0, m0_, l0, h1, m1, l1;
            split3_pair(xv[j].x, xv[j].y, h0, m0_, l0); split3_pair(xv[j].z, xv[j].w, h1, m1, l1);
            const int o = xoff(xr[j], xk[j]);
            *reinterpret_cast<uint2*>(&Xp[o]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&Xp[kSplitXImg + o]) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(&Xp[2 * kSplitXImg + o]) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            unsigned ph, pm, pl;
            split3_pair(av[2 * j], av[2 * j + 1], ph, pm, pl);
            if (AK) {                                                            // two consecutive k of one m-row: one 4-byte store per piece
                const int e = tid + 512 * j, o = aoff(e >> 4, (e & 15) * 2);
                *reinterpret_cast<unsigned*>(&Ap[o]) = ph; *reinterpret_cast<unsigned*>(&Ap[kSplitAImg + o]) = pm; *reinterpret_cast<unsigned*>(&Ap[2 * kSplitAImg + o]) = pl;
            } else {                                                             // two unrelated elements: single bf16 stores
                const int e0 = tid + 512 * (2 * j), e1 = e0 + 512, o0 = aoff(e0 % kRowsA, e0 / kRowsA), o1 = aoff(e1 % kRowsA, e1 / kRowsA);
                Ap[o0] = (unsigned short)ph; Ap[kSplitAImg + o0] = (unsigned short)pm; Ap[2 * kSplitAImg + o0] = (unsigned short)pl;
                Ap[o1] = (unsigned short)(ph >> 16); Ap[kSplitAImg + o1] = (unsigned short)(pm >> 16); Ap[2 * kSplitAImg + o1] = (unsigned short)(pl >> 16);
            }
        }
        }
        __syncthreads();
        if (ci + 1 < nchunks && !(g.dbg & 1)) gload(cn * kBigKc);                // next chunk in flight under this chunk's MFMAs
        if (!(g.dbg & 2))
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int xo = ((ks * 2 + h) * kSplitRowsX + 32 * wn + c) * 8;
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Xp[xo]));
            const bf16x8 Bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Xp[kSplitXImg + xo]));
            const bf16x8 Bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Xp[2 * kSplitXImg + xo]));
#pragma unroll
            for (int t = 0; t < MB; ++t) {
                const int ao = ((ks * 2 + h) * kRowsA + 32 * (wm * MB + t) + c) * 8;
                const bf16x8 Ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Ap[ao]));
                const bf16x8 Am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Ap[kSplitAImg + ao]));
                const bf16x8 Al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Ap[2 * kSplitAImg + ao]));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[t], 0, 0, 0);   // small terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* __restrict__ C = g.C + (size_t)z * g.zC;
    const float* __restrict__ bias = g.bias ? g.bias + (size_t)z * g.zBias : nullptr;
    const float* __restrict__ aux = g.aux ? g.aux + (size_t)z * g.zAux : nullptr;
    const int tile_n = (int)blockIdx.y * WN + wn;
    if (g.dbg & 4) { if (acc[0][0] == 123.456f) C[0] = 1.f; return; }
#pragma unroll
    for (int t = 0; t < MB; ++t) {                                              // one m-tile at a time through the transposition buffer (the loop's last barrier freed the operand images)
        if (t) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[t][r];
        __syncthreads();
        store_tile(g, C, bias, aux, m0 + 32 * (wm * MB + t), tile_n * 32, lane, [&](int rr, int ll) { return red[wave][rr][ll]; });
    }
}
// two independent contractions in one launch (the weight-gradient and the data-gradient of one layer): blockIdx.z < za runs `a`
struct GemmPair { GemmArgs a, b; int za; };
// (both bodies: a contraction with K >= kLdsMinK runs the LDS-staged one, a short one the direct-load one; the dynamic LDS block serves either)
using RedBuf = float[kGemmWaves][16][kRedStride];
__device__ __forceinline__ void gemm_body_any(const GemmArgs& g, int z, float* smem, int bx, int by) {
    if (g.use_lds) gemm_body_lds_any(g, z, smem, bx, by); else gemm_body<true>(g, z, *reinterpret_cast<RedBuf*>(smem), bx, by);
}
__global__ __launch_bounds__(64 * kGemmWaves, 4) void sac_gemm_pair_kernel(GemmPair p) {
    extern __shared__ __attribute__((aligned(16))) float lds_smem[];
    if ((int)blockIdx.z < p.za) gemm_body_any(p.a, blockIdx.z, lds_smem, blockIdx.x, blockIdx.y); else gemm_body_any(p.b, (int)blockIdx.z - p.za, lds_smem, blockIdx.x, blockIdx.y);
}

// up to four independent contractions in one launch (a net's [dW3|db3], [dW2|db2] and dz1 of the reverse pass): blockIdx.z picks the contraction
struct GemmMulti { GemmArgs g[4]; int end[4], tm[4], tn[4]; int n; };   // flat grid: blocks [end[i-1], end[i]) run contraction i, tile (bx, by) of batch z
__global__ __launch_bounds__(64 * kGemmWaves, 4) void sac_gemm_multi_kernel(GemmMulti p) {
    extern __shared__ __attribute__((aligned(16))) float lds_smem[];
    const int b = blockIdx.x;
    const int i = b < p.end[0] ? 0 : b < p.end[1] ? 1 : b < p.end[2] ? 2 : 3;
    const int local = b - (i ? p.end[i - 1] : 0), per = p.tm[i] * p.tn[i];
    const int z = local / per, r = local - z * per, by = r / p.tm[i], bx = r - by * p.tm[i];
    gemm_body_any(p.g[i], z, lds_smem, bx, by);                                                 // (the argument block is read from the kernel-argument segment at a uniform runtime offset: scalar loads)
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
bool gemm_prepare(GemmArgs& g) {
    g.vecA = g.sAk == 1 && g.sAm % 4 == 0 && g.K % 4 == 0 && aligned16(g.A) && g.zA % 4 == 0;
    g.vecB = !g.ones_n && g.sBk == 1 && g.sBn % 4 == 0 && g.K % 4 == 0 && aligned16(g.B) && g.zB % 4 == 0;
    g.vecBn = g.sBn == 1 && g.sBk != 1 && g.sBk % 4 == 0 && aligned16(g.B) && g.zB % 4 == 0;        // n-contiguous operand: float4 runs along n (big kernel, BN)
    if (g.zdivB <= 0) g.zdivB = 1;
    // the LDS-staged split-K shape: access pattern per operand (rows = m for A, n for B)
    g.ldsA = g.vecA ? 0 : (g.sAm == 1 && g.sAk % 4 == 0 && aligned16(g.A) && g.zA % 4 == 0) ? 1 : 2;
    g.ldsB = g.vecB ? 0 : (g.sBn == 1 && g.sBk % 4 == 0 && aligned16(g.B) && g.zB % 4 == 0) ? 1 : 2;
    static const bool no_lds = debug_env("DRIL_GEMM_NO_LDS") != nullptr;                                // A/B knob: the direct-load split-K body for every contraction
    g.use_lds = (g.K >= kLdsMinK && g.ldsA < 2 && g.ldsB < 2 && !no_lds) ? 1 : 0;
    static const int dbg_bits = debug_env("DRIL_GEMM_DBG") ? std::atoi(debug_env("DRIL_GEMM_DBG")) : 0; g.dbg = dbg_bits;   // diagnostic ablations (results wrong on purpose)
    return g.M > 0 && g.N > 0 && g.K > 0;
}

}  // namespace

// kernels that take the 66 KB dynamic LDS block: raise the limit once per kernel and device context
static hipError_t lds_attr(const void* fn) {
    static std::mutex mu; static std::vector<std::pair<const void*, int>> done;            // (kernel, device): a process may hold handles on several devices
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (std::find(done.begin(), done.end(), std::make_pair(fn, dev)) != done.end()) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess) done.push_back(std::make_pair(fn, dev));
    return e;
}
GemmArgs gemm_args() { GemmArgs g; memset(&g, 0, sizeof(g)); g.alpha = 1.0f; return g; }

hipError_t launch_gemm_pair(GemmArgs a, int Za, GemmArgs b, int Zb, hipStream_t s) {
    if (!gemm_prepare(a) || !gemm_prepare(b)) return hipErrorInvalidValue;
    GemmPair p{a, b, Za};
    const int tm = std::max((a.M + 31) / 32, (b.M + 31) / 32), tn = std::max((a.N + 31) / 32, (b.N + 31) / 32);
    if (tn > 65535 || Za + Zb > 65535) return hipErrorInvalidValue;
    { hipError_t e = lds_attr((const void*)sac_gemm_pair_kernel); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(sac_gemm_pair_kernel, dim3(tm, tn, Za + Zb), dim3(64 * kGemmWaves), kLdsBytes, s, p);
    return hipGetLastError();
}
hipError_t launch_gemm_multi(const GemmArgs* gs, const int* Zs, int n, hipStream_t s) {
    if (n < 1 || n > 4) return hipErrorInvalidValue;
    GemmMulti p; p.n = n; long long total = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            p.g[i] = gs[i]; if (!gemm_prepare(p.g[i])) return hipErrorInvalidValue;
            p.tm[i] = (p.g[i].M + 31) / 32; p.tn[i] = (p.g[i].N + 31) / 32; total += (long long)p.tm[i] * p.tn[i] * Zs[i];
        } else { p.g[i] = gs[0]; p.tm[i] = p.tn[i] = 1; }
        p.end[i] = (int)total;
    }
    if (total > 0x7fffffff) return hipErrorInvalidValue;
    { hipError_t e = lds_attr((const void*)sac_gemm_multi_kernel); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(sac_gemm_multi_kernel, dim3((unsigned)total), dim3(64 * kGemmWaves), kLdsBytes, s, p);
    return hipGetLastError();
}
hipError_t launch_gemm(GemmArgs g, int Z, hipStream_t s) {
    if (!gemm_prepare(g)) return hipErrorInvalidValue;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    if ((tn + kGemmWaves - 1) / kGemmWaves > 65535 || Z > 65535) return hipErrorInvalidValue;
    constexpr long long many_tiles = 2048;                                                                 // tile count from which the LDS-tiled throughput shapes take over from split-K (measured both ways at 2 048 tiles: 31 vs 43 us)
    const bool many = (long long)tm * tn * Z >= many_tiles && g.K >= kBigKc;
    const bool a_m = g.sAm == 1, a_k = !a_m && g.sAk == 1;                                             // A m-contiguous / k-contiguous
    const bool b_k = g.sBk == 1 && g.vecB && !g.ones_n, b_n = !b_k && g.sBn == 1 && g.sBk != 1;        // B k-contiguous (float4 rows) / n-contiguous
    const dim3 bgrid(tm, (tn + kGemmWaves - 1) / kGemmWaves, Z), bblock(64 * kGemmWaves);
    static const bool split_all = debug_env("DRIL_GEMM_SPLIT") != nullptr;                                // A/B: also for callers that did not ask (SAC)
    const bool split = many && (g.allow_split || split_all) && g.K >= 64;
    if (split && (a_m || a_k) && (b_k || b_n)) {
        // output rows per workgroup: 64 (two m-tiles per wave) when that still leaves >= 2 workgroups per CU, else 32 (a 4096-row collection forward has 256
        // blocks of 32 x 256: it stays at MB = 1).  Taller blocks were built and measured (MB = 4, and full-height 256 / 512-row blocks that stage every
        // activation row once): SLOWER — 65 / 57 vs 69 TFLOP/s at hidden 512 — because their registers / LDS leave one workgroup per CU, and with one
        // workgroup the phases of a chunk do not overlap (ablation of a 512 x 32768 x 512 reverse-pass contraction, DRIL_GEMM_DBG: global loads 31 + staging
        // 37 + MFMA 57 + epilogue 41 = 166 us vs 160 us measured: purely additive).  Two co-resident workgroups hide each other's phases.  A two-slot LDS ring
        // inside one workgroup (staging of chunk i + 1 interleaved with the MFMAs of chunk i, loads two chunks ahead; built, parity-green) was slower still (55):
        // the split form reads 0.75 ds_read_b128 per MFMA (9 reads per 12 MFMAs of 32 cycles), ~2000 LDS cycles per chunk against 1536 MFMA cycles per SIMD —
        // the kernel is LDS-read-bound, and the next step is 2 x 2 register tiling per wave (0.5 reads per MFMA), not more overlap.
        constexpr int mb_cap = 2;                                                                           // (1, 2, 4 are built; 4 measured slower, comment above)
        int MB = 1;
        const long long tn256 = (g.N + 255) / 256;
        for (int cand = 4; cand > 1; cand >>= 1) if (cand <= mb_cap && g.M >= 32 * cand && (long long)((g.M + 32 * cand - 1) / (32 * cand)) * tn256 * Z >= 512) { MB = cand; break; }
        const dim3 sgrid((g.M + 32 * MB - 1) / (32 * MB), bgrid.y, Z);
#define DRIL_SPLIT_LAUNCH(AKv, BNv) { if (MB == 4) hipLaunchKernelGGL((sac_gemm_split_kernel<AKv, BNv, 4, 1>), sgrid, bblock, 0, s, g); \
                                      else if (MB == 2) hipLaunchKernelGGL((sac_gemm_split_kernel<AKv, BNv, 2, 1>), sgrid, bblock, 0, s, g); \
                                      else hipLaunchKernelGGL((sac_gemm_split_kernel<AKv, BNv, 1, 1>), sgrid, bblock, 0, s, g); }
        if (a_m && b_k) DRIL_SPLIT_LAUNCH(false, false) else if (a_k && b_k) DRIL_SPLIT_LAUNCH(true, false) else if (a_m && b_n) DRIL_SPLIT_LAUNCH(false, true) else DRIL_SPLIT_LAUNCH(true, true)
#undef DRIL_SPLIT_LAUNCH
    }
    else if (many && a_m && b_k) hipLaunchKernelGGL((sac_gemm_big_kernel<false, false>), bgrid, bblock, 0, s, g);
    else if (many && a_k && b_k) hipLaunchKernelGGL((sac_gemm_big_kernel<true, false>), bgrid, bblock, 0, s, g);
    else if (many && a_m && b_n) hipLaunchKernelGGL((sac_gemm_big_kernel<false, true>), bgrid, bblock, 0, s, g);
    else if (many && a_k && b_n) hipLaunchKernelGGL((sac_gemm_big_kernel<true, true>), bgrid, bblock, 0, s, g);
    else if ((long long)tm * tn * Z >= many_tiles) hipLaunchKernelGGL(sac_gemm_kernel<false>, dim3(tm, (tn + kGemmWaves - 1) / kGemmWaves, Z), dim3(64 * kGemmWaves), 0, s, g);
    else if (g.use_lds) {
        hipError_t e = lds_attr((const void*)sac_gemm_lds_kernel); if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sac_gemm_lds_kernel, dim3(tm, tn, Z), dim3(64 * kGemmWaves), kLdsBytes, s, g);
    }
    else hipLaunchKernelGGL(sac_gemm_kernel<true>, dim3(tm, tn, Z), dim3(64 * kGemmWaves), 0, s, g);
    return hipGetLastError();
}

}  // namespace dril

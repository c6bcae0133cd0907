// dril_gemm.hip — generic strided fp32 contraction on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate); see dril_gemm.h.
// Written for the SAC update (~25 small dense contractions per gradient step) and reused by the generic on-policy path.
#include <algorithm>
#include <cstring>

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
__device__ __forceinline__ float gemm_epilogue(const GemmArgs& g, float v, int mm, size_t ci, const float* __restrict__ bias, const float* __restrict__ aux) {
    v *= g.alpha;
    if (bias) v += bias[mm];
    if (g.epi == EPI_RELU) v = v > 0.f ? v : 0.f;
    else if (g.epi == EPI_TANH) v = tanhf(v);
    else if (g.epi == EPI_MASK_RELU) v = aux[ci] > 0.f ? v : 0.f;
    else if (g.epi == EPI_MASK_TANH) { const float y = aux[ci]; v *= 1.0f - y * y; }
    return v;
}
template <bool SPLIT>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int z, float (&red)[kGemmWaves][16][kRedStride]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const float* __restrict__ A = g.A + (size_t)z * g.zA;
    const float* __restrict__ B = g.B + (size_t)(z / g.zdivB) * g.zB;
    const int tile_n = SPLIT ? (int)blockIdx.y : (int)blockIdx.y * kGemmWaves + wave;
    if (SPLIT && ((int)blockIdx.x * 32 >= g.M || tile_n * 32 >= g.N)) return;                  // pair launches: the grid covers the larger problem
    const int m = blockIdx.x * 32 + c, n = tile_n * 32 + c;
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
            if (q + u < q1) {
                load_operand4(A, m, g.M, g.sAm, g.sAk, k0, g.K, g.vecA, a[u]);
                if (ones) { for (int t = 0; t < 4; ++t) b[u][t] = k0 + t < g.K ? 1.f : 0.f; }
                else load_operand4(B, n, g.N, g.sBn, g.sBk, k0, g.K, g.vecB, b[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kGemmDepth; ++u)
            if (q + u < q1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc = mfma32(a[u][t], b[u][t], acc);
            }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float* __restrict__ C = g.C + (size_t)z * g.zC;
    const float* __restrict__ bias = g.bias ? g.bias + (size_t)z * g.zBias : nullptr;
    const float* __restrict__ aux = g.aux ? g.aux + (size_t)z * g.zAux : nullptr;
    // element e of a 32x32 tile: m_local = e & 31 (fastest, C's unit stride), n_local = e >> 5; it sits in register rr of lane ll
    if (SPLIT) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = threadIdx.x + 512 * i, ml = e & 31, nl = e >> 5;
            const int rr = (ml & 3) + 4 * (ml >> 3), ll = nl + 32 * ((ml >> 2) & 1);
            const int mm = blockIdx.x * 32 + ml, nn = tile_n * 32 + nl;
            if (mm >= g.M || nn >= g.N) continue;
            float v = red[0][rr][ll];
#pragma unroll
            for (int w = 1; w < kGemmWaves; ++w) v += red[w][rr][ll];                          // fixed order
            const size_t ci = (size_t)mm * g.sCm + (size_t)nn * g.sCn;
            C[ci] = gemm_epilogue(g, v, mm, ci, bias, aux);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = lane + 64 * i, ml = e & 31, nl = e >> 5;
            const int rr = (ml & 3) + 4 * (ml >> 3), ll = nl + 32 * ((ml >> 2) & 1);
            const int mm = blockIdx.x * 32 + ml, nn = tile_n * 32 + nl;
            if (mm >= g.M || nn >= g.N) continue;
            const size_t ci = (size_t)mm * g.sCm + (size_t)nn * g.sCn;
            C[ci] = gemm_epilogue(g, red[wave][rr][ll], mm, ci, bias, aux);
        }
    }
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
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, ml = e & 31, nl = e >> 5;
        const int rr = (ml & 3) + 4 * (ml >> 3), ll = nl + 32 * ((ml >> 2) & 1);
        const int mm = m0 + ml, nn = tile_n * 32 + nl;
        if (mm >= g.M || nn >= g.N) continue;
        const size_t ci = (size_t)mm * g.sCm + (size_t)nn * g.sCn;
        C[ci] = gemm_epilogue(g, red[wave][rr][ll], mm, ci, bias, aux);
    }
}
// two independent contractions in one launch (the weight-gradient and the data-gradient of one layer): blockIdx.z < za runs `a`
struct GemmPair { GemmArgs a, b; int za; };
__global__ __launch_bounds__(64 * kGemmWaves) void sac_gemm_pair_kernel(GemmPair p) {
    __shared__ float red[kGemmWaves][16][kRedStride];
    if ((int)blockIdx.z < p.za) gemm_body<true>(p.a, blockIdx.z, red); else gemm_body<true>(p.b, (int)blockIdx.z - p.za, red);
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }
bool gemm_prepare(GemmArgs& g) {
    g.vecA = g.sAk == 1 && g.sAm % 4 == 0 && g.K % 4 == 0 && aligned16(g.A) && g.zA % 4 == 0;
    g.vecB = !g.ones_n && g.sBk == 1 && g.sBn % 4 == 0 && g.K % 4 == 0 && aligned16(g.B) && g.zB % 4 == 0;
    g.vecBn = g.sBn == 1 && g.sBk != 1 && g.sBk % 4 == 0 && aligned16(g.B) && g.zB % 4 == 0;        // n-contiguous operand: float4 runs along n (big kernel, BN)
    if (g.zdivB <= 0) g.zdivB = 1;
    return g.M > 0 && g.N > 0 && g.K > 0;
}

}  // namespace

GemmArgs gemm_args() { GemmArgs g; memset(&g, 0, sizeof(g)); g.alpha = 1.0f; return g; }

hipError_t launch_gemm_pair(GemmArgs a, int Za, GemmArgs b, int Zb, hipStream_t s) {
    if (!gemm_prepare(a) || !gemm_prepare(b)) return hipErrorInvalidValue;
    GemmPair p{a, b, Za};
    const int tm = std::max((a.M + 31) / 32, (b.M + 31) / 32), tn = std::max((a.N + 31) / 32, (b.N + 31) / 32);
    if (tn > 65535 || Za + Zb > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sac_gemm_pair_kernel, dim3(tm, tn, Za + Zb), dim3(64 * kGemmWaves), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_gemm(GemmArgs g, int Z, hipStream_t s) {
    if (!gemm_prepare(g)) return hipErrorInvalidValue;
    const int tm = (g.M + 31) / 32, tn = (g.N + 31) / 32;
    if ((tn + kGemmWaves - 1) / kGemmWaves > 65535 || Z > 65535) return hipErrorInvalidValue;
    const bool many = (long long)tm * tn * Z >= 2048 && g.K >= kBigKc;
    const bool a_m = g.sAm == 1, a_k = !a_m && g.sAk == 1;                                             // A m-contiguous / k-contiguous
    const bool b_k = g.sBk == 1 && g.vecB && !g.ones_n, b_n = !b_k && g.sBn == 1 && g.sBk != 1;        // B k-contiguous (float4 rows) / n-contiguous
    const dim3 bgrid(tm, (tn + kGemmWaves - 1) / kGemmWaves, Z), bblock(64 * kGemmWaves);
    if (many && a_m && b_k) hipLaunchKernelGGL((sac_gemm_big_kernel<false, false>), bgrid, bblock, 0, s, g);
    else if (many && a_k && b_k) hipLaunchKernelGGL((sac_gemm_big_kernel<true, false>), bgrid, bblock, 0, s, g);
    else if (many && a_m && b_n) hipLaunchKernelGGL((sac_gemm_big_kernel<false, true>), bgrid, bblock, 0, s, g);
    else if (many && a_k && b_n) hipLaunchKernelGGL((sac_gemm_big_kernel<true, true>), bgrid, bblock, 0, s, g);
    else if ((long long)tm * tn * Z >= 2048) hipLaunchKernelGGL(sac_gemm_kernel<false>, dim3(tm, (tn + kGemmWaves - 1) / kGemmWaves, Z), dim3(64 * kGemmWaves), 0, s, g);
    else hipLaunchKernelGGL(sac_gemm_kernel<true>, dim3(tm, tn, Z), dim3(64 * kGemmWaves), 0, s, g);
    return hipGetLastError();
}

}  // namespace dril

// dril_gemm.h — the generic strided fp32-MFMA contraction shared by the SAC path (dril_sac.hip) and the generic
// (any obs / action / hidden width) on-policy path (dril_generic.hip).  Kernels in dril_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

namespace dril {

// C[z](M x N) = epi(alpha * A[z](M x K) . B[z](K x N) + bias[z](M)), every operand addressed by element strides
struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; const float* aux;
    float* zout;                                  // when set: the pre-activation alpha * A.B + bias goes here too (C's strides), for activations whose derivative needs it (gelu, swish)
    int M, N, K;
    int sAm, sAk, sBk, sBn, sCm, sCn;             // element strides
    long long zA, zB, zC, zBias, zAux;            // batch (blockIdx.z) strides
    int zdivB;                                    // B uses batch index z / zdivB (several nets reading one input)
    int vecA, vecB;                               // operand has unit stride along k, 16-byte aligned rows and K % 4 == 0: float4 loads
    int use_lds;                                  // the LDS-staged (coalesced-load) split-K body for this contraction (set by gemm_prepare: K >= 64)
    int ldsA, ldsB;                               // access pattern of the LDS-staged split-K shape (set by gemm_prepare): 0 k-contiguous float4, 1 row-contiguous float4, 2 element by element
    int vecBn;                                    // B has unit stride along n with 16-byte aligned rows: float4 runs along n (LDS-tiled kernel only)
    int ones_n;                                   // B(:, N-1) == 1 (appends the bias column to a weight-gradient contraction)
    int epi; float alpha;
    int dbg;                                      // DRIL_GEMM_DBG bits (diagnostic, results wrong): 1 no global operand loads after the first chunk, 2 no MFMA, 4 no epilogue, 8 no staging stores
    int allow_split;                              // large contractions may run on the bf16 matrix cores with fp32-equivalent 3-piece operand splitting (generic on-policy path; SAC keeps the f32 MFMA)
};
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_TANH = 2, EPI_MASK_RELU = 3, EPI_MASK_TANH = 4,
       // round 3: the other activations whose derivative is a function of the OUTPUT (the reverse pass keeps activations, not pre-activations): NNlib sigmoid,
       // elu (alpha = 1), leakyrelu (a = 0.01), softplus — generic PPO path only (the reference takes any activation, layer_constructors.jl:8,56)
       EPI_SIGMOID = 5, EPI_ELU = 6, EPI_LEAKY = 7, EPI_SOFTPLUS = 8, EPI_MASK_SIGMOID = 9, EPI_MASK_ELU = 10, EPI_MASK_LEAKY = 11, EPI_MASK_SOFTPLUS = 12,
       // gelu (NNlib: the tanh form) and swish: the derivative is a function of the PRE-activation, which the forward of an update keeps beside the activation (GemmArgs.zout); their mask epilogues read it as `aux`
       EPI_GELU = 13, EPI_SWISH = 14, EPI_MASK_GELU = 15, EPI_MASK_SWISH = 16 };
__host__ __device__ inline bool epi_reads_aux(int epi) { return epi == EPI_MASK_RELU || epi == EPI_MASK_TANH || (epi >= EPI_MASK_SIGMOID && epi <= EPI_MASK_SOFTPLUS) || epi == EPI_MASK_GELU || epi == EPI_MASK_SWISH; }
inline bool activation_needs_preactivation(int act) { return act == 6 || act == 7; }
// activation codes of dril_config.activation -> the forward epilogue and the mask (act'(y)) epilogue
inline int epi_of_activation(int act) { return act == 1 ? EPI_RELU : act == 2 ? EPI_SIGMOID : act == 3 ? EPI_ELU : act == 4 ? EPI_LEAKY : act == 5 ? EPI_SOFTPLUS : act == 6 ? EPI_GELU : act == 7 ? EPI_SWISH : EPI_TANH; }
inline int mask_epi_of_activation(int act) { return act == 1 ? EPI_MASK_RELU : act == 2 ? EPI_MASK_SIGMOID : act == 3 ? EPI_MASK_ELU : act == 4 ? EPI_MASK_LEAKY : act == 5 ? EPI_MASK_SOFTPLUS : act == 6 ? EPI_MASK_GELU : act == 7 ? EPI_MASK_SWISH : EPI_MASK_TANH; }

GemmArgs gemm_args();
// one contraction, Z batches (blockIdx.z); picks the split-K / tile-parallel / LDS-tiled shape from the tile count
hipError_t launch_gemm(GemmArgs g, int Z, hipStream_t s);
// two independent contractions in one launch (a layer's [dW | db] and its dz): blockIdx.z < Za runs `a`
hipError_t launch_gemm_pair(GemmArgs a, int Za, GemmArgs b, int Zb, hipStream_t s);
// up to four independent contractions in one launch (the split-K shape, like the pair): contraction i runs Zs[i] batches
hipError_t launch_gemm_multi(const GemmArgs* gs, const int* Zs, int n, hipStream_t s);

}  // namespace dril

// microbenchmark: cost of a grid-wide barrier on MI355X (cooperative launch, 256 workgroups x 512 threads)
//   A: cooperative_groups grid.sync()      B: hand-rolled sense-reversing barrier on one L2 atomic counter (bounded spin)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(512) void k_cg(int n, float* out) {
    cg::grid_group g = cg::this_grid();
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.f; g.sync(); }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* gen, unsigned nblocks, unsigned& local_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        local_gen += 1;
        __threadfence();
        const unsigned prev = atomicAdd(counter, 1u);
        if (prev == nblocks - 1) { atomicExch(counter, 0u); __threadfence(); atomicExch(gen, local_gen); }
        else {
            unsigned spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local_gen) { __builtin_amdgcn_s_sleep(1); if (++spins > 50000000u) { ok = false; break; } }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(512) void k_hand(int n, float* out, unsigned* counter, unsigned* gen) {
    float v = threadIdx.x; unsigned lg = 0;
    for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.f; if (!grid_barrier(counter, gen, gridDim.x, lg)) break; }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    float* out; unsigned* ctr; hipMalloc(&out, 4096); hipMalloc(&ctr, 8); hipMemset(ctr, 0, 8);
    unsigned* counter = ctr; unsigned* gen = ctr + 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {64, 256}) {
        int n = 2000; float ms;
        void* args1[] = {&n, &out};
        hipLaunchCooperativeKernel((void*)k_cg, dim3(blocks), dim3(512), args1, 0, 0); hipDeviceSynchronize();
        hipEventRecord(e0); hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(blocks), dim3(512), args1, 0, 0); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1); printf("%3d blocks  cg::grid.sync()      : %.2f us per barrier (%s)\n", blocks, 1e3 * ms / n, hipGetErrorString(e));
        hipMemset(ctr, 0, 8);
        void* args2[] = {&n, &out, &counter, &gen};
        hipLaunchCooperativeKernel((void*)k_hand, dim3(blocks), dim3(512), args2, 0, 0); hipDeviceSynchronize(); hipMemset(ctr, 0, 8);
        hipEventRecord(e0); e = hipLaunchCooperativeKernel((void*)k_hand, dim3(blocks), dim3(512), args2, 0, 0); hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1); printf("%3d blocks  hand-rolled barrier  : %.2f us per barrier (%s)\n", blocks, 1e3 * ms / n, hipGetErrorString(e));
    }
    return 0;
}

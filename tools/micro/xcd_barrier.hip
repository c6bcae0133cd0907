// microbenchmark (round 3): what does a barrier cost when the workgroups that synchronise sit on ONE XCD, and what does a two-level
// (per-XCD counter, then the eight XCD leaders) barrier cost over the whole chip?  Workgroups are dealt to the XCDs round-robin by workgroup id (id % 8), so a
// launch of 8 n workgroups in which only those with id % 8 == x take part confines the n participants to XCD x (the others exit at once).
//   variants of the hand-rolled sense-reversing barrier (one counter + one generation word, s_sleep spin, bounded):
//     flat/agent   : every participant adds to one counter with agent-scope atomics                       (tools/micro/grid_barrier.hip, for reference)
//     xcd/agent    : n participants on one XCD, agent-scope atomics
//     xcd/wg       : n participants on one XCD, the read-modify-writes at workgroup scope (performed in that XCD's L2, no wider coherence action); the
//                    polling load stays agent scope so it cannot be served from a stale L1 line
//     two-level    : 8 n participants; per-XCD counter (xcd/wg form), the last arriver of an XCD adds to a chip-wide counter (agent scope), everybody polls the
//                    chip-wide generation word
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/xcd_barrier tools/micro/xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define SPIN_LIMIT 20000000u
template <int SCOPE> __device__ __forceinline__ unsigned add1(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, SCOPE); }
__device__ __forceinline__ unsigned poll(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int SCOPE>
__device__ __forceinline__ bool barrier1(unsigned* counter, unsigned* gen, unsigned n, unsigned& lg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        lg += 1;
        __threadfence();
        if (add1<SCOPE>(counter) == n - 1) { __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __threadfence(); __hip_atomic_store(gen, lg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        else { unsigned spins = 0; while (poll(gen) != lg) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { ok = false; break; } } }
        __threadfence();
    }
    __syncthreads();
    return ok;
}
// mode 0: all workgroups take part (flat); mode 1: only id % 8 == 0 (one XCD)
template <int SCOPE>
__global__ __launch_bounds__(512) void k_one(int n, int mode, float* out, unsigned* counter, unsigned* gen) {
    if (mode == 1 && (blockIdx.x & 7) != 0) return;
    const unsigned parts = mode == 1 ? gridDim.x / 8 : gridDim.x;
    float v = threadIdx.x; unsigned lg = 0;
    for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.f; if (!barrier1<SCOPE>(counter, gen, parts, lg)) { v = -1.f; break; } }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
// two-level: xc[8 * 16] per-XCD counters (64-byte apart), top counter + generation word
__global__ __launch_bounds__(512) void k_two(int n, float* out, unsigned* xc, unsigned* top, unsigned* gen) {
    const unsigned x = blockIdx.x & 7, per = gridDim.x / 8;
    float v = threadIdx.x; unsigned lg = 0; bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
        v = v * 1.0001f + 1.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            lg += 1;
            __threadfence();
            bool release = false;
            if (add1<__HIP_MEMORY_SCOPE_WORKGROUP>(xc + 16 * x) == per - 1) {
                __hip_atomic_store(xc + 16 * x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (add1<__HIP_MEMORY_SCOPE_AGENT>(top) == 7) { __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __threadfence(); __hip_atomic_store(gen, lg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); release = true; }
            }
            if (!release) { unsigned spins = 0; while (poll(gen) != lg) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { ok = false; break; } } }
            __threadfence();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = ok ? v : -1.f;
}
int main() {
    float* out; unsigned* ctr; hipMalloc(&out, 4096); hipMalloc(&ctr, 4096); 
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 2000;
    auto run = [&](const char* name, auto launch, int parts) {
        float ms; float host[1024];
        hipMemset(ctr, 0, 4096); launch(); hipDeviceSynchronize(); hipMemset(ctr, 0, 4096);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
        hipError_t e = hipGetLastError();
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(host, out, 4096, hipMemcpyDeviceToHost);
        bool timed_out = false; for (int i = 0; i < 1024; ++i) if (host[i] == -1.f) timed_out = true;
        printf("%-46s %3d participants: %6.2f us per barrier%s (%s)\n", name, parts, 1e3 * ms / n, timed_out ? "  ** SPIN LIMIT HIT: invalid **" : "", hipGetErrorString(e));
    };
    unsigned* counter = ctr; unsigned* gen = ctr + 64; unsigned* xc = ctr + 128; unsigned* top = ctr + 320;
    for (int parts : {8, 16, 32}) {
        hipMemset(out, 0, 4096);
        run("flat, agent-scope atomics", [&] { k_one<__HIP_MEMORY_SCOPE_AGENT><<<parts, 512>>>(n, 0, out, counter, gen); }, parts);
        run("one XCD, agent-scope atomics", [&] { k_one<__HIP_MEMORY_SCOPE_AGENT><<<8 * parts, 512>>>(n, 1, out, counter, gen); }, parts);
        run("one XCD, workgroup-scope read-modify-writes", [&] { k_one<__HIP_MEMORY_SCOPE_WORKGROUP><<<8 * parts, 512>>>(n, 1, out, counter, gen); }, parts);
    }
    for (int blocks : {64, 256}) {
        hipMemset(out, 0, 4096);
        run("flat, agent-scope atomics", [&] { k_one<__HIP_MEMORY_SCOPE_AGENT><<<blocks, 512>>>(n, 0, out, counter, gen); }, blocks);
        run("two-level (per-XCD counter, 8 leaders)", [&] { k_two<<<blocks, 512>>>(n, out, xc, top, gen); }, blocks);
    }
    return 0;
}

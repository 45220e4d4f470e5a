// Experiment (GPU): per-iteration cost of a barrier among the ~63 single-wave blocks that share one dim of an L = 1
// clique (15 independent groups), with the exchanged data moved by agent-scope relaxed atomic stores / loads (sc1,
// no cache-wide release/acquire fences).  build: hipcc --offload-arch=gfx950 -O3 -o groupsync_bench groupsync_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

struct Ctl { unsigned ctr[64 * 32]; unsigned err; };

__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my sc1 stores have completed
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *err = 1; break; }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// grid = (tiles, 1, groups); every block: write `payload` floats, barrier(group), read `reads` floats of other blocks,
// (optionally a second barrier as a two-phase reduce/update would need)
__global__ void __launch_bounds__(64) k(Ctl* c, float* data, int iters, int payload, int reads, int two_phase, float* sink) {
    const unsigned tiles = gridDim.x, g = blockIdx.z;
    float* mine = data + ((size_t)g * tiles + blockIdx.x) * payload;
    float acc = 0.f;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        for (int e = threadIdx.x; e < payload; e += 64) __hip_atomic_store(&mine[e], (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier(&c->ctr[g * 32], ++phase * tiles, &c->err);
        for (int e = threadIdx.x; e < reads; e += 64) {
            const unsigned other = (blockIdx.x + 1 + e % (tiles - 1)) % tiles;
            acc += __hip_atomic_load(&data[((size_t)g * tiles + other) * payload + (e % payload)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (two_phase) {
            for (int e = threadIdx.x; e < 8; e += 64) __hip_atomic_store(&mine[e], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            group_barrier(&c->ctr[g * 32], ++phase * tiles, &c->err);
            for (int e = threadIdx.x; e < 448; e += 64)
                acc += __hip_atomic_load(&data[((size_t)g * tiles + (e % tiles)) * payload + (e / tiles)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == -1.f) *sink = acc;
}

int main() {
    Ctl* c; CK(hipMalloc(&c, sizeof(Ctl)));
    float* data; CK(hipMalloc(&data, sizeof(float) * 64 * 64 * 4096));
    float* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int cfgs[][5] = {{63, 15, 512, 0, 0}, {63, 15, 512, 448, 0}, {63, 15, 512, 448, 1}, {63, 15, 0, 0, 0}, {63, 7, 512, 448, 1},
                           {32, 15, 512, 448, 1}, {63, 1, 512, 448, 1}, {16, 15, 512, 448, 1}};
    for (auto& cf : cfgs) {
        int tiles = cf[0], groups = cf[1], payload = cf[2] ? cf[2] : 1, reads = cf[3], two = cf[4], iters = 300;
        if (cf[2] == 0) payload = 1;
        void* args[] = {&c, &data, &iters, &payload, &reads, &two, &sink};
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl)));
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)k, dim3(tiles, 1, groups), dim3(64), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("tiles %2d x groups %2d, payload %3d floats, reads %3d, %s: %.2f us per iteration%s\n", tiles, groups, cf[2], reads,
               two ? "two barriers" : "one barrier ", ms * 1e3 / iters, h.err ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}

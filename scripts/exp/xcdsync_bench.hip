// Experiment (GPU): cost of a barrier among the blocks of ONE (clique, dim) group when they all sit on one XCD (the dim-major
// kernel's grid: blockIdx.x = XCD, blockIdx.y = block inside the group, blockIdx.z = octet of groups), with the exchanged data
// moved by sc1 stores / loads (device-scope relaxed atomics: L2 is the XCD's coherence point, no cache-wide fences).
// An iteration of a persistent one-layer training kernel would be: unit -> write the block's gradient copy -> barrier(group)
// -> read the group's copies (fused Adam) -> next unit.   build: hipcc --offload-arch=gfx950 -O3 -o xcdsync_bench xcdsync_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

struct Ctl { unsigned ctr[256 * 32]; unsigned err; };

__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my sc1 stores have completed (acknowledged by L2)
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
}

// same_xcd = 1: group = blockIdx.x + 8 * blockIdx.z, member = blockIdx.y (grid (8, members, groups / 8)): one XCD per group
// same_xcd = 0: group = blockIdx.z, member = blockIdx.x (grid (members, 1, groups)): a group's blocks land on different XCDs
__global__ void __launch_bounds__(256) k(Ctl* c, float* data, int iters, int payload, int members, int same_xcd, int work, float* sink) {
    const unsigned g = same_xcd ? blockIdx.x + 8 * blockIdx.z : blockIdx.z;
    const unsigned me = same_xcd ? blockIdx.y : blockIdx.x;
    float* mine = data + ((size_t)g * members + me) * payload;
    float acc = 0.f;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        float v = (float)it;
        for (int q = 0; q < work; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);          // stand-in for the unit's arithmetic
        for (int e = threadIdx.x; e < payload; e += 256) __hip_atomic_store(&mine[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier(&c->ctr[g * 32], ++phase * members, &c->err);
        for (int e = threadIdx.x; e < payload; e += 256)                                // fused Adam: every block sums the group's copies
            for (int m = 0; m < members; ++m)
                acc += __hip_atomic_load(&data[((size_t)g * members + m) * payload + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier(&c->ctr[g * 32 + 16], phase * members, &c->err);                 // nobody overwrites a copy that is still being read
    }
    if (acc == -1.f) *sink = acc;
}

// One barrier per iteration: the blocks ADD their partials into an L2-resident accumulator (three rotating buffers: after the
// barrier of iteration i every read of buffer i - 1 is over, so member 0 zeroes it for iteration i + 2), meet once, read the sums.
__global__ void __launch_bounds__(256) k1(Ctl* c, float* data, int iters, int payload, int members, int work, float* sink) {
    const unsigned g = blockIdx.x + 8 * blockIdx.z;
    const unsigned me = blockIdx.y;
    float* accum = data + (size_t)g * 3 * payload;
    float acc = 0.f;
    unsigned phase = 0;
    for (int it = 0; it < iters; ++it) {
        float v = (float)it;
        for (int q = 0; q < work; ++q) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        float* buf = accum + (size_t)(it % 3) * payload;
        for (int e = threadIdx.x; e < payload; e += 256) __hip_atomic_fetch_add(&buf[e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        group_barrier(&c->ctr[g * 32], ++phase * members, &c->err);
        for (int e = threadIdx.x; e < payload; e += 256) acc += __hip_atomic_load(&buf[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (me == 0) {
            float* old = accum + (size_t)((it + 2) % 3) * payload;
            for (int e = threadIdx.x; e < payload; e += 256) __hip_atomic_store(&old[e], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == -1.f) *sink = acc;
}

int main() {
    Ctl* c; CK(hipMalloc(&c, sizeof(Ctl)));
    float* data; CK(hipMalloc(&data, sizeof(float) * 256 * 16 * 1024));
    float* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // members, groups, payload, same_xcd, work
    const int cfgs[][5] = {{8, 16, 512, 1, 0}, {8, 16, 512, 0, 0}, {8, 16, 512, 1, 2000}, {8, 16, 512, 0, 2000}, {8, 96, 512, 1, 0}, {8, 96, 512, 0, 0},
                           {8, 16, 1, 1, 0}, {8, 16, 1, 0, 0}};
    for (auto& cf : cfgs) {
        int members = cf[0], groups = cf[1], payload = cf[2], same = cf[3], work = cf[4], iters = 500;
        void* args[] = {&c, &data, &iters, &payload, &members, &same, &work, &sink};
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl)));
            CK(hipEventRecord(e0));
            dim3 grid = same ? dim3(8, members, groups / 8) : dim3(members, 1, groups);
            CK(hipLaunchCooperativeKernel((const void*)k, grid, dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("%d blocks x %2d groups (%s), payload %3d floats, work %4d: %.2f us per iteration (two barriers + copies)%s\n", members, groups,
               same ? "one XCD per group " : "group across XCDs", payload, work, ms * 1e3 / iters, h.err ? "  [SPIN TIMEOUT]" : "");
    }
    const int cfg1[][4] = {{8, 16, 640, 0}, {8, 16, 640, 2000}, {8, 96, 640, 0}, {4, 16, 640, 0}, {2, 16, 640, 0}};     // members, groups, payload, work
    for (auto& cf : cfg1) {
        int members = cf[0], groups = cf[1], payload = cf[2], work = cf[3], iters = 500;
        void* args[] = {&c, &data, &iters, &payload, &members, &work, &sink};
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(c, 0, sizeof(Ctl)));
            CK(hipMemset(data, 0, sizeof(float) * 256 * 16 * 1024));
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)k1, dim3(8, members, groups / 8), dim3(256), args, 0, 0));
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("ONE barrier + atomic accumulation: %d blocks x %2d groups (one XCD per group), payload %3d floats, work %4d: %.2f us per iteration%s\n",
               members, groups, payload, work, ms * 1e3 / iters, h.err ? "  [SPIN TIMEOUT]" : "");
    }
    return 0;
}

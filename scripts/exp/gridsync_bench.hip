// Experiment (GPU): cost of a software grid barrier on MI355X for the shapes of a persistent training kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o gridsync_bench gridsync_bench.hip ; run: ./gridsync_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

struct Bar { unsigned top; unsigned pad[63]; unsigned sub[64 * 64]; unsigned err; };

__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target, unsigned* err) {
    unsigned spins = 0;
    // relaxed polling (no cache invalidate per poll), one acquire fence once the target is reached
    while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { *err = 1; return false; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// mode 0: flat counter; mode 1: two-level (groups of `gsz` blocks, the last arriver of a group bumps the top)
__global__ void bar_kernel(Bar* b, int iters, int mode, int ngroups, float* data, int payload) {
    const unsigned nb = gridDim.x;
    const unsigned g = blockIdx.x % ngroups;
    const unsigned gcount = nb / ngroups + ((g < nb % ngroups) ? 1u : 0u);
    for (int it = 0; it < iters; ++it) {
        if (payload) {   // every block writes a slab line and reads other blocks' lines (visibility through the barrier)
            for (int e = threadIdx.x; e < payload; e += blockDim.x) data[(size_t)blockIdx.x * payload + e] = (float)it;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (mode == 0) {
                __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(&b->top, (unsigned)(it + 1) * nb, &b->err);
            } else {
                const unsigned old = __hip_atomic_fetch_add(&b->sub[g * 64], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)(it + 1) * gcount)
                    __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(&b->top, (unsigned)(it + 1) * ngroups, &b->err);
            }
        }
        __syncthreads();
        if (payload) {
            float acc = 0.f;
            for (int e = threadIdx.x; e < 64; e += blockDim.x)
                acc += __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + e * 7 + 1) % nb) * payload]);
            if (acc < -1.f) data[0] = acc;
        }
    }
}

int main() {
    Bar* b; CK(hipMalloc(&b, sizeof(Bar)));
    float* data; CK(hipMalloc(&data, sizeof(float) * 4096 * 2048));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][2] = {{945, 64}, {128, 384}, {119, 512}, {64, 384}, {256, 64}, {32, 512}};
    for (auto& sh : shapes) {
        int nblk = sh[0], nthr = sh[1];
        int maxb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, bar_kernel, nthr, 0));
        for (int mode = 0; mode < 2; ++mode) for (int ng : {8, 32}) for (int payload : {0, 512}) {
            if (mode == 0 && ng != 8) continue;
            CK(hipMemset(b, 0, sizeof(Bar)));
            int iters = 200;
            void* args[] = {&b, &iters, &mode, &ng, &data, &payload};
            // warm-up + timed
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(b, 0, sizeof(Bar)));
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel((const void*)bar_kernel, dim3(nblk), dim3(nthr), args, 0, 0));
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Bar hb; CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
            printf("blocks %4d x %3d thr (occ %d/CU) mode %s groups %2d payload %4d: %.2f us per barrier round%s\n", nblk, nthr, maxb,
                   mode ? "two-level" : "flat     ", ng, payload, ms * 1e3 / iters, hb.err ? "  [SPIN TIMEOUT]" : "");
        }
    }
    return 0;
}

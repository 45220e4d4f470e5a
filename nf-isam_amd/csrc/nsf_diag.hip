// nsf_diag.hip -- DIAGNOSTIC library (libnfisam_diag.so), not part of the product's C ABI (include/nfisam_hip.h).
//
// One entry point: occupy the device the way a foreign process's long kernel would -- `blocks` blocks of 256 threads, each
// holding `lds_bytes` of LDS, spin for `seconds` of wall clock on `stream`.  Used by the test of the co-residency probe
// (tests/test_hip_parity.py: test_a_busy_device_is_probed_before_a_plan_takes_the_persistent_form) and by
// scripts/exp/occupy_check.py.  Round 6: moved out of libnfisam_hip.so (a kernel that can hold every CU for
// up to 30 s does not belong among the product library's exports).
#include <hip/hip_runtime.h>
#include <stddef.h>

__global__ void __launch_bounds__(256) nsf_diag_occupy_kernel(unsigned budget_ticks) {
    extern __shared__ unsigned occupy_lds[];
    if (threadIdx.x == 0) occupy_lds[0] = 0u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    for (;;) {
        // every wave of the block stays until its first thread says so (a block of the real kernel holds four waves too)
        if (threadIdx.x == 0 && __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)budget_ticks)
            __hip_atomic_store(&occupy_lds[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_load(&occupy_lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

extern "C" int nfisam_diag_occupy_device(int blocks, size_t lds_bytes, float seconds, void* stream) {
    if (blocks < 1 || !(seconds > 0.0f) || seconds > 30.0f || lds_bytes > (size_t)(160 * 1024)) return 1;
    if (hipFuncSetAttribute((const void*)nsf_diag_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)) != hipSuccess) return 2;
    hipLaunchKernelGGL(nsf_diag_occupy_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, (hipStream_t)stream, (unsigned)(seconds * 1e8f));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

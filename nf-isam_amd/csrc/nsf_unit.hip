// nsf_unit.hip -- one KERNEL unit of libnfisam_hip.so: the gfx950 kernels templated on (K bins, H hidden width) and
// their launchers, instantiated for the (K, H) pairs of unit NSF_UNIT (nsf_units.h).  Compiled once per unit
// (Makefile: -DNSF_UNIT=u); see nsf_host.h for how the units fit together.
//
// Work decomposition:
//   training   one layer        nsf_train1_kernel: one wave = one dim x T tiles, dim-major blocks, Adam fused into the next launch
//                               (every real NF-iSAM run; hidden_dim 4 / 8 / 16)
//              layers / dL/dx   tile-major: a block = W waves over ONE particle tile through all layers (in the density direction
//                               all D conditioners depend only on the layer input, src/flows/flows.py:77-83; layers are
//                               sequential: layer inputs live in LDS and waves meet at __syncthreads()):
//                               nsf_train3_kernel (two dims per wave, MFMA conditioner; latency regime, D 6..16),
//                               nsf_train2_kernel (two lanes per particle; latency regime, other D),
//                               nsf_train_kernel (one lane per particle; big launches, hidden_dim != 8, A/B)
//   inverse    one wave per 64 particles, dims sequential (true data dependence, flows.py:115-137)
//   walk       whole Bayes tree root -> leaves in one launch
#include <stddef.h>
#include <type_traits>
#include <vector>
#include <mutex>

#include "nsf_host.h"
#include "nsf_cond_mfma.h"
#include "nsf_half.h"
#include "nsf_bookkeep.h"

#ifndef NSF_UNIT
#error "compile with -DNSF_UNIT=<unit index> (see nsf_units.h)"
#endif
#define NSF_PASTE_(a, b) a##b
#define NSF_PASTE(a, b) NSF_PASTE_(a, b)
#define NSF_FOR_EACH_KH(X) NSF_PASTE(NSF_KH_, NSF_UNIT)(X)

// Gradient sink.  slab = false: accumulate into one shared buffer with float atomics (any number of
// tiles).  slab = true: this tile owns a private copy of the gradient buffer and every entry is written
// exactly once per iteration with a plain store; the Adam kernel sums the tiles in a fixed order
// (no memory-side atomic tail on small launches, bitwise-reproducible training).
typedef __attribute__((address_space(1))) float gfloat;       // device-memory float (global_ instead of flat_ instructions)
typedef __attribute__((ext_vector_type(4))) float vf4_t;
typedef __attribute__((address_space(1))) vf4_t gvf4_t;
__device__ __forceinline__ void gsink(float* dst, float v, bool slab) {
    if (slab) *dst = v; else atomicAdd(dst, v);
}
__device__ __forceinline__ void gsink(gfloat* dst, float v, bool slab) {
    if (slab) *dst = v; else __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gsink4(gfloat* dst, const __attribute__((ext_vector_type(4))) float& v, bool slab) {
    if (slab) {
        *(gvf4_t*)dst = v;
        return;
    }
    gsink(dst + 0, v.x, slab);
    gsink(dst + 1, v.y, slab);
    gsink(dst + 2, v.z, slab);
    gsink(dst + 3, v.w, slab);
}
__device__ __forceinline__ void gsink4(float* dst, const __attribute__((ext_vector_type(4))) float& v, bool slab) {
    if (slab) {   // every destination of a 4-group is 16-byte aligned (kernel-layout rows are multiples of 4 floats)
        *(__attribute__((ext_vector_type(4))) float*)dst = v;
        return;
    }
    gsink(dst + 0, v.x, slab);
    gsink(dst + 1, v.y, slab);
    gsink(dst + 2, v.z, slab);
    gsink(dst + 3, v.w, slab);
}

// =============================================================================================
// training / VJP kernels
// =============================================================================================
template <int K, int H, typename WP>
__device__ __forceinline__ void load_theta(WP lp, int i, const float* xin, int xstride, int lane,
                                           float (&h1)[H], float (&h2)[H],
                                           float (&th)[Layout<K, H>::PoP]) {
    using LY = Layout<K, H>;
    if (i == 0) {
        load_row<LY::PoP>(lp, th);
    } else {
        WP blk = lp + LY::off(i);
        cond_hidden<K, H, WP>(blk, i, xin, xstride, lane, h1, h2);
        cond_theta<K, H, WP>(blk, i, h2, th);
    }
}

// ---- weight-gradient GEMMs on the matrix cores -------------------------------------------------
// For one (layer, dim) unit the parameter gradients are sums over the wave's 64 particles of outer
// products:  dW2t[c][o] = sum_p h2ext[p][c] * gth[p][o],  dW1t[c][j] = sum_p h1ext[p][c] * ga2[p][j],
// dW0t[c][j] = sum_p xext[p][c] * ga1[p][j]   (ext = activation vector with a trailing 1 for the bias).
// With the particle index as the contraction dimension these are GEMMs, and
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain) does the cross-lane reduction for free:
// 4 particles per instruction.  The VALU phase holds "particle on the lane, feature in the
// register"; the MFMA operands need "feature on lane&15, particle on lane>>4", so the vectors take
// one trip through a wave-private LDS tile, feature-major with a row stride of XS = 66 floats
// (66 mod 32 = 2 makes both the 64-lane row writes and the 16x4 operand reads conflict-free).
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- diagnostic build only (-DNSF_STAMPS): per-wave s_memtime stamps at phase boundaries ---------
// which waves are stamped and into which of the 64 slots: tile group 0 of clique 0, slot = 4 * dim + wave (the dim-major
// kernel, whose grid is 1-D, redefines the two around its body)
#define STAMP_SEL (blockIdx.x == 0 && blockIdx.y == 0)
#define STAMP_SLOT ((w + blockIdx.z * (blockDim.x >> 6)) & 63)
#if defined(NSF_STAMPS) && NSF_UNIT == 0   // stamps exist in the K = 9, H = 8 unit only
__device__ unsigned long long g_stamps[64 * 32];
__device__ unsigned long long g_blk[4096 * 2];
__device__ __forceinline__ unsigned long long g_stamps_t0(int) { return 0ull; }
#if NSF_STAMPS == 3      // pinned: per-phase cycle sums in LDS (nsf_train1_kernel only); PSTAMP ties the phase's results to the stamp
#define STAMP_DECL unsigned long long sprev_ = 0ull;
#define STAMP(id) do { } while (0)
#define PSTAMP(id, va, vb)                                                                          \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(va), "+v"(vb)::"memory"); \
        if (lane == 0 && sprev_ != 0ull && w < 4) atomicAdd((unsigned*)&smem[PANEL_BASE - 64 + w * 16 + ((id) & 15)], (unsigned)(t_ - sprev_)); \
        sprev_ = t_;                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#elif NSF_STAMPS == 2      // light: the raw stamp only (no per-phase accumulators: they cost 32 registers and distort the kernel)
#define STAMP_DECL
#define STAMP(id)                                                                                   \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (STAMP_SEL && lane == 0)                                                                 \
            g_stamps[STAMP_SLOT * 32 + (id)] = t_;                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#else
#define STAMP_DECL unsigned long long sacc_[16] = {0ull}; unsigned long long sprev_ = 0ull;
#define STAMP(id)                                                                                   \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        if (sprev_ != 0ull) sacc_[(id) & 15] += t_ - sprev_;                                        \
        sprev_ = t_;                                                                                \
        if (STAMP_SEL && lane == 0 && (id) < 32) {                                                  \
            const int sw_ = STAMP_SLOT * 32;                                                        \
            g_stamps[sw_ + (id)] = t_;                                                              \
            g_stamps[sw_ + 16 + ((id) & 15)] = sacc_[(id) & 15];                                    \
        }                                                                                           \
        if (((id) == 0 || (id) == 9) && lane == 0 && w == 0) {                                                     \
            const unsigned bid_ = blockIdx.x + gridDim.x * (blockIdx.z + gridDim.z * blockIdx.y);                   \
            if (bid_ < 4096) g_blk[bid_ * 2 + ((id) == 9)] = __builtin_amdgcn_s_memrealtime();                       \
        }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#endif
extern "C" int nfisam_debug_read_blocks(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk), sizeof(unsigned long long) * 4096 * 2);
}
extern "C" int nfisam_debug_write_stamps(const unsigned long long* in) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), in, sizeof(unsigned long long) * 64 * 32);
}
#if NSF_STAMPS == 3
extern "C" int nfisam_debug_read_stg(unsigned long long* out, int zero) {
    if (zero) { unsigned long long z[32] = {0ull}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stg), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stg), sizeof(unsigned long long) * 32);
}
#endif
extern "C" int nfisam_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 32);
}
#else
#define STAMP(id) do { } while (0)
#define STAMP_DECL
#endif
#ifndef PSTAMP
#define PSTAMP(id, va, vb) STAMP(id)
#endif

// Per-iteration loss sums live behind the gradient slabs in the kgrad workspace: a ring of LOSS_RING
// iterations x LOSS_SLOTS words (spread so that hundreds of waves do not serialise on one address).  The
// bookkeeping kernel that closes a chunk of iterations consumes and clears them.
// staging rows per wave.  nsf_train_kernel: phase A stages one 16-row tile of gth at a time next to h2 (16 + H rows),
// phase B ga2 | ga1 | h1 (3H rows); the atomics sink transposes (H+1) x PoP floats through the same rows.
template <int K, int H>
struct StgRows {
    static constexpr int PoP = Layout<K, H>::PoP;
    static constexpr int a = 16 + H, b = 3 * H, c = ((H + 1) * PoP + 65) / 66;   // bias rows come from the shared `ones` row
    static constexpr int value = (a > b ? (a > c ? a : c) : (b > c ? b : c));
    static constexpr int split = PoP + 16 + 3 * H;      // nsf_train2_kernel: gth | h2 | pad to +16 | ga2 | ga1 | h1
};

// Lanes of ONE wave exchange data through LDS: the hardware executes a wave's LDS operations in
// order, but the compiler must be told that other lanes write between this lane's store and its
// later load (otherwise it forwards the lane's own store).  Wavefront-scope release/acquire
// fences + wave barrier: no instructions, only ordering.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// OCC = waves per SIMD the register allocation aims at (amdgpu_waves_per_eu): the throughput launches (many cliques
// per launch, 4-wave blocks, scalar-path weights) want 4 resident waves per SIMD to cover the scalar-load and LDS
// round trips of a wave's dependent instruction stream; the latency launches (one clique) run one or two waves per
// SIMD and keep every intermediate in registers.
template <int K, int H, bool MF, bool WL, int OCC>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC, 8))) nsf_train_kernel(TrainArgs a) {
    using LY = Layout<K, H>;
    using WP = typename std::conditional<WL, const float*, cfloat*>::type;
    constexpr int PoP = LY::PoP;
    constexpr int NT = (PoP + 15) / 16;                       // 16-row output tiles of gth
    constexpr int STG_ROWS = StgRows<K, H>::value;
    static_assert(!MF || H == 8, "MFMA gradient path packs ga2|ga1 into one 16-row operand tile: H = 8");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool batched = a.cliques != nullptr;
    const nfisam_clique* cp = batched ? (a.cliques + blockIdx.y) : nullptr;
    const float* x = batched ? cp->x : a.single.x;
    float* kparams = batched ? cp->kparams : a.single.kparams;
    float* G = batched ? cp->kgrad : a.single.kgrad;
    nfisam_train_state* st = batched ? cp->state : a.single.state;
    const int n = batched ? cp->n : a.single.n;
    const int D = batched ? cp->D : a.single.D;
    const int L = a.L;
    const float B = a.B;
    const bool slab = a.slab != 0;
    const int T = a.tiles_per_block > 1 ? a.tiles_per_block : 1;      // 64-particle tiles this block sums over (L == 1 only)
    const size_t gstride = (size_t)L * (size_t)(a.layer_stride > 0 ? a.layer_stride : LY::count(D));
    float* ring = G + (slab ? (size_t)gridDim.x : (size_t)1) * gstride;   // loss ring behind the gradient copies
    if (slab) G += (size_t)blockIdx.x * gstride;

    const int p0 = blockIdx.x * TILE * T;
    if (p0 >= n) return;
    // state words as per-lane loads issued now and consumed only after the prologue's global loads have
    // been issued too: one exposed memory round trip instead of two
    int st_stop = 0, st_step = 0;
    if (st != nullptr) {
        st_stop = __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_step = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    // With L == 1 the dims of a tile never exchange data: blockIdx.z picks a group of W dims so that
    // every wave runs exactly one unit (grid.z = ceil(D / W)).  L > 1 needs all dims in one block.
    const int dim_lo = blockIdx.z * W;
    const int dim_step = (gridDim.z > 1) ? D : W;       // one pass over [dim_lo + w] when grouped
    if (dim_lo >= D) return;
    const int DT = D * XS;
    STAMP_DECL
    STAMP(0);

    const int Pk = a.layer_stride > 0 ? a.layer_stride : LY::count(D);
    // parameter range this block needs: everything, or (L == 1, grouped) only its own dims' blocks
    const int dim_hi = (gridDim.z > 1) ? ((dim_lo + W < D) ? dim_lo + W : D) : D;
    const int w_lo = (gridDim.z > 1 && dim_lo > 0) ? LY::off(dim_lo) : 0;
    const int w_hi = (gridDim.z > 1) ? LY::off(dim_hi) : L * Pk;
    float* wlds = smem;               // [w_hi - w_lo] parameter copy (WL only), 16-byte aligned rows
    float* xs = smem + (WL ? a.wl_floats : 0);   // [L or T][D][XS] layer inputs / particle tiles, dimension-major
    const int gt = a.g_tiles ? DT : 0;   // dL/dx tiles exist only when a layer input gradient is needed
    float* g0 = xs + (L > T ? L : T) * DT;   // [D][XS]
    float* g1 = g0 + gt;              // [D][XS]
    float* ones = g1 + gt;            // [XS] constant 1 (bias column of the gradient GEMMs)
    float* stg = ones + XS + w * (STG_ROWS * XS);   // wave-private staging tile (MF only)

    // ---- prologue: ALL global loads first (particle tiles + parameter rows), one wait, then LDS ----
    // The T tiles are 64*T*D contiguous floats; element e belongs to particle e / D, column e % D (the
    // quotient by a reciprocal multiply: exact for e < 2^20).  Rows beyond n read as 0.
    {
        constexpr int XB = 16, WB = 4;               // loads in flight per lane: dwords of x, float4 of weights
        const int nx = D * TILE * T;
        const int rows = (n - p0) < TILE * T ? (n - p0) : TILE * T;
        const int lim = rows * D;                                     // floats of these tiles that exist
        const float* xt = x + (size_t)p0 * D;
        const float invD = 1.0f / (float)D;
        const int tot4 = WL ? ((w_hi - w_lo) >> 2) : 0;               // block offsets are multiples of 4 floats
        const f32x4* wsrc = (const f32x4*)(kparams + w_lo);
        f32x4* wdst = (f32x4*)wlds;
        int e0 = threadIdx.x, f0 = threadIdx.x;
        while (e0 < nx || f0 < tot4) {
            float xv[XB];
            f32x4 wv[WB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                xv[u] = (e < lim) ? xt[e] : 0.0f;
            }
            if constexpr (WL) {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int f = f0 + u * (int)blockDim.x;
                    wv[u] = (f < tot4) ? wsrc[f] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (st_stop != 0 || st_step + a.iter_idx >= a.max_iters) return;      // block-uniform (first consumer of the state loads)
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < nx) {
                    int pq = (int)(((float)e + 0.5f) * invD);
                    int k = e - pq * D;
                    if (k < 0) { k += D; pq -= 1; }
                    if (k >= D) { k -= D; pq += 1; }
                    xs[(pq >> 6) * DT + k * XS + (pq & 63)] = xv[u];
                }
            }
            if constexpr (WL) {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int f = f0 + u * (int)blockDim.x;
                    if (f < tot4) wdst[f] = wv[u];
                }
            }
            e0 += XB * (int)blockDim.x;
            f0 += WB * (int)blockDim.x;
        }
    }
    if (threadIdx.x < XS) ones[threadIdx.x] = 1.0f;
    __syncthreads();

    WP kp;
    if constexpr (WL) kp = wlds - w_lo; else kp = (cfloat*)kparams;
    STAMP(1);

    // ---- forward-only passes: layers 0 .. L-2 (the last layer is recomputed in backward) ---
    for (int l = 0; l + 1 < L; ++l) {
        WP lp = kp + (size_t)l * Pk;
        const float* xin = xs + l * DT;
        float* xout = xs + (l + 1) * DT;
        for (int i = dim_lo + w; i < D; i += dim_step) {
            float h1[H], h2[H], th[PoP];
            load_theta<K, H, WP>(lp, i, xin, XS, lane, h1, h2, th);
            Spline<K> S;
            float z, lad;
            spline_eval<K, PoP, false>(xin[i * XS + lane], th, B, S, z, lad);
            xout[i * XS + lane] = z;
        }
        STAMP(10);
        __syncthreads();
        STAMP(11);
    }

    // ---- backward with recompute, last layer first ------------------------------------------
    float lossv = 0.0f;
    float* gcur = g0;
    float* gprev = g1;
    const int r16 = lane & 15, kq = lane >> 4;
    for (int l = L - 1; l >= 0; --l) {
        const bool last = (l == L - 1);
        const bool need_gx = (l > 0) || (a.gx != nullptr);
        WP lp = kp + (size_t)l * Pk;
        float* Gl = G + (size_t)l * Pk;
        if (need_gx) {
            for (int e = threadIdx.x; e < DT; e += blockDim.x) gprev[e] = 0.0f;
            __syncthreads();
        }
        for (int i = dim_lo + w; i < D; i += dim_step) {
            // gradient accumulators of this dim: they persist over the block's T tiles (the MFMAs add into them), so a
            // block emits ONE gradient copy however many tiles it covers
            f32x4 cacc[NT], c1 = {0.f, 0.f, 0.f, 0.f}, c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            float r0 = 0.0f;                           // i == 0: init_param gradient of lane < PoP
            WP blk = lp + LY::off(i);
            float* Gb = Gl + LY::off(i);
            for (int tt = 0; tt < T; ++tt) {
                if (p0 + tt * TILE >= n) break;
                const float* xin = xs + (T > 1 ? tt : l) * DT;
                const int gp = p0 + tt * TILE + lane;
                const bool valid = gp < n;
                float h1[H], h2[H], th[PoP], gth[PoP];
                STAMP(2);
                load_theta<K, H, WP>(lp, i, xin, XS, lane, h1, h2, th);
                STAMP(3);
                Spline<K> S;
                float z, lad;
                spline_eval<K, PoP, false>(xin[i * XS + lane], th, B, S, z, lad);
                STAMP(4);
                float gz, gl;
                if (a.nll_mode) {
                    gl = -1.0f;
                    gz = last ? z : gcur[i * XS + lane];
                    if (valid) lossv += (last ? 0.5f * z * z : 0.0f) - lad;
                } else {
                    gl = (a.gl != nullptr && valid) ? a.gl[gp] : 0.0f;
                    gz = last ? (valid ? a.gz[(size_t)gp * D + i] : 0.0f) : gcur[i * XS + lane];
                }
                if (!valid) { gz = 0.0f; gl = 0.0f; }
                const float gxs = spline_backward<K, PoP>(S, B, gz, gl, gth);
                if (need_gx) atomicAdd(&gprev[i * XS + lane], gxs);
                STAMP(5);

                if (i == 0) {   // init_param: plain sum over particles of gth
                    constexpr int N0 = (PoP <= 32) ? 32 : 64;
                    float v[N0];
#pragma unroll
                    for (int t = 0; t < N0; ++t) v[t] = (t < PoP) ? gth[t] : 0.0f;
                    r0 += butterfly<N0>(v, lane);
                    continue;
                }
                // ---- per-particle back-propagation through the conditioner (VALU, scalar-path weights)
                float gh2[H], ga2[H], ga1[H];
                {
                    WP W2 = reload_ptr(blk + LY::oW2(i));
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        float wr[PoP];
                        load_row<PoP>(W2 + k * PoP, wr);
                        float acc = 0.0f;
#pragma unroll
                        for (int o = 0; o < PoP; ++o) acc = __builtin_fmaf(wr[o], gth[o], acc);
                        gh2[k] = acc;
                        if (k & 1) row_group_fence<WP>();
                    }
#pragma unroll
                    for (int k = 0; k < H; ++k) ga2[k] = gh2[k] * (1.0f - h2[k] * h2[k]);
                    WP W1 = reload_ptr(blk + LY::oW1(i));
#pragma unroll
                    for (int k = 0; k < H; ++k) {
                        float wr[H];
                        load_row<H>(W1 + k * H, wr);
                        float acc = 0.0f;
#pragma unroll
                        for (int j = 0; j < H; ++j) acc = __builtin_fmaf(wr[j], ga2[j], acc);
                        ga1[k] = acc * (1.0f - h1[k] * h1[k]);
                        if ((k & 7) == 7) row_group_fence<WP>();
                    }
                    if (need_gx) {
                        WP W0 = blk;
                        for (int k = 0; k < i; k += 4) {       // 4 rows in flight (rows >= i alias later weights: unused)
                            float wq[4][H], acc[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) load_row<H>(W0 + (k + u) * H, wq[u]);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc[u] = 0.0f;
#pragma unroll
                                for (int j = 0; j < H; ++j) acc[u] = __builtin_fmaf(wq[u][j], ga1[j], acc[u]);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (k + u < i) atomicAdd(&gprev[(k + u) * XS + lane], acc[u]);
                        }
                    }
                }
                STAMP(6);
                if constexpr (MF) {
                    // ======== phase A: dW2t | db2 = [h2, 1]^T (x) gth, one 16-row tile of gth per staging round ========
                    // (staging rows 0..15: the gth tile, rows 16..16+H-1: h2 -- 24 rows instead of PoP + H = 40: the
                    //  staging tile is what limits the resident waves per CU)
                    {
#pragma unroll
                        for (int k = 0; k < H; ++k) stg[(16 + k) * XS + lane] = h2[k];
                        const float* pa = stg + r16 * XS + kq;             // A: gth rows 16t + r16 (>= Po: unused outputs)
                        // B: [h2 | 1]; the bias column and the unused columns beyond it read the constant-one row
                        const float* pb = ((r16 < H) ? (stg + (16 + r16) * XS) : ones) + kq;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
#pragma unroll
                            for (int o = 0; o < 16; ++o) stg[o * XS + lane] = (16 * t + o < PoP) ? gth[(16 * t + o < PoP) ? 16 * t + o : 0] : 0.0f;
                            wave_lds_sync();
#pragma unroll
                            for (int s4 = 0; s4 < TILE; s4 += 4) cacc[t] = mfma4(pa[s4], pb[s4], cacc[t]);
                            wave_lds_sync();
                        }
                    }
                    STAMP(7);
                    // ======== phase B: dW1t | db1 = [h1,1]^T (x) ga2 ;  dW0t | db0 = [x,1]^T (x) ga1 ========
                    {
#pragma unroll
                        for (int j = 0; j < H; ++j) {
                            stg[j * XS + lane] = ga2[j];
                            stg[(H + j) * XS + lane] = ga1[j];
                            stg[(2 * H + j) * XS + lane] = h1[j];
                        }
                        wave_lds_sync();
                        const float* pa = stg + r16 * XS + kq;                   // A: rows 0..7 ga2, 8..15 ga1
                        const float* pb1 = ((r16 < H) ? (stg + (2 * H + r16) * XS) : ones) + kq;   // B1: [h1 | 1]
                        float areg[TILE / 4];
#pragma unroll
                        for (int s4 = 0; s4 < TILE; s4 += 4) {
                            areg[s4 / 4] = pa[s4];
                            c1 = mfma4(areg[s4 / 4], pb1[s4], c1);
                        }
                        // x tiles (16 input columns each); column i is the bias (ones row).  Tile 0 (columns 0..15) keeps
                        // its accumulator over the block's tiles; further column tiles (D > 16) add into the block's
                        // own gradient copy: the same lane owns the same entries for every tile, so a plain
                        // read-add-write is race-free (slab) / a float atomic (shared copy).
                        {
                            const float* pb0 = ((r16 < i) ? (xin + r16 * XS) : ones) + kq;
#pragma unroll
                            for (int s4 = 0; s4 < TILE; s4 += 4) c0 = mfma4(areg[s4 / 4], pb0[s4], c0);
                        }
                        for (int ct = 1; ct * 16 <= i; ++ct) {
                            const int cab = ct * 16 + r16;
                            const float* pb0 = ((cab < i) ? (xin + cab * XS) : ones) + kq;
                            f32x4 cx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int s4 = 0; s4 < TILE; s4 += 4) cx = mfma4(areg[s4 / 4], pb0[s4], cx);
                            if (kq >= 2 && cab <= i) {
                                float* dst = &Gb[cab * H + 4 * (kq - 2)];
                                if (slab && tt > 0) cx += *(const f32x4*)dst;
                                gsink4(dst, cx, slab);
                            }
                        }
                        wave_lds_sync();
                    }
                    STAMP(8);
                } else {
                    // ======== butterfly variant (round-1 v1): reduce-scatter over lanes with ds_bpermute (T == 1) ========
                    {
                        constexpr int TOT = (H + 1) * PoP;
                        float* Gw = Gb + LY::oW2(i);
#pragma unroll
                        for (int c = 0; c < (TOT + 63) / 64; ++c) {
                            float v[64];
#pragma unroll
                            for (int t = 0; t < 64; ++t) {
                                const int f = c * 64 + t;
                                const int k = f / PoP, o = f % PoP;
                                v[t] = (f < TOT) ? ((k < H) ? gth[o] * h2[k < H ? k : 0] : gth[o]) : 0.0f;
                            }
                            const float r = butterfly<64>(v, lane);
                            if (c * 64 + lane < TOT) gsink(&Gw[c * 64 + lane], r, slab);
                        }
                    }
                    {
                        constexpr int TOT = (H + 1) * H;
                        float* Gw = Gb + LY::oW1(i);
#pragma unroll
                        for (int c = 0; c < (TOT + 63) / 64; ++c) {
                            float v[64];
#pragma unroll
                            for (int t = 0; t < 64; ++t) {
                                const int f = c * 64 + t;
                                const int k = f / H, j = f % H;
                                v[t] = (f < TOT) ? ((k < H) ? ga2[j] * h1[k < H ? k : 0] : ga2[j]) : 0.0f;
                            }
                            const float r = butterfly<64>(v, lane);
                            if (c * 64 + lane < TOT) gsink(&Gw[c * 64 + lane], r, slab);
                        }
                    }
                    {
                        float* Gw = Gb;
                        constexpr int RPC = 64 / H;
                        const int tot = (i + 1) * H;
                        for (int kc = 0; kc <= i; kc += RPC) {
                            float v[64];
#pragma unroll
                            for (int r_ = 0; r_ < RPC; ++r_) {
                                const int k = kc + r_;
                                const float xk = (k < i) ? xin[k * XS + lane] : ((k == i) ? 1.0f : 0.0f);
#pragma unroll
                                for (int j = 0; j < H; ++j) v[r_ * H + j] = ga1[j] * xk;
                            }
                            const float r = butterfly<64>(v, lane);
                            if (kc * H + lane < tot) gsink(&Gw[kc * H + lane], r, slab);
                        }
                    }
                }
            }
            // ---- this dim's gradient copy: every entry written once (slab) or added once (atomics) per block ----
            if (i == 0) {
                if (lane < PoP) gsink(&Gl[lane], r0, slab);
                continue;
            }
            if constexpr (MF) {
                // C layout of the MFMAs: col = lane&15 (= input column c), rows 4*(lane>>4)+r (= four consecutive outputs)
                float* Gw = Gb + LY::oW2(i);
                if (slab) {
                    if (r16 <= H) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gw[r16 * PoP + 16 * t + 4 * kq], cacc[t], true);
                    }
                } else {
                    // atomics: transpose through LDS to the flat order (64 consecutive addresses per wave instruction)
                    if (r16 <= H) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) {   // scalar stores: same type as the float re-reads below
                                float* d = &stg[r16 * PoP + 16 * t + 4 * kq];
                                d[0] = cacc[t].x; d[1] = cacc[t].y; d[2] = cacc[t].z; d[3] = cacc[t].w;
                            }
                    }
                    wave_lds_sync();
                    constexpr int TOT = (H + 1) * PoP;
#pragma unroll
                    for (int c = 0; c < (TOT + 63) / 64; ++c) {
                        const int f = c * 64 + lane;
                        if (f < TOT) atomicAdd(&Gw[f], stg[f]);
                    }
                    wave_lds_sync();
                }
                // c1 rows 0..7 (kq < 2) are ga2[j], j = 4*kq + r ; flat f = c*H + j, c <= H
                if (kq < 2 && r16 <= H) gsink4(&(Gb + LY::oW1(i))[r16 * H + 4 * kq], c1, slab);
                // c0 rows 8..15 (kq >= 2) are ga1[j], j = 4*(kq-2)+r ; flat f = c*H + j, input column c = r16
                if (kq >= 2 && r16 <= i) gsink4(&Gb[r16 * H + 4 * (kq - 2)], c0, slab);
            }
        }
        STAMP(12);
        __syncthreads();
        STAMP(13);
        float* tmp = gcur; gcur = gprev; gprev = tmp;
    }

    if (a.gx != nullptr) {   // gcur now holds dL/dx of layer 0's input (T == 1)
        for (int e = threadIdx.x; e < D * TILE; e += blockDim.x) {
            const int p = e / D, k = e - p * D;
            const int q = p0 + p;
            if (q < n) a.gx[(size_t)q * D + k] = gcur[k * XS + p];
        }
    }
    STAMP(9);
    if (a.nll_mode) {
        const float tot = wave_sum(lossv);
        if (lane == 0) {
            float* dst = (st != nullptr) ? &ring[((st_step + a.iter_idx) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                   ((blockIdx.x * 7 + blockIdx.z * 13 + w) & (LOSS_SLOTS - 1))]
                                         : a.loss_sum;
            if (dst != nullptr) atomicAdd(dst, tot);
        }
    }
}

// One float per lane into row `ROW` of a [rows][XS] LDS tile whose row-0 address for this lane is `lane_addr` (bytes).
// Written as an explicit ds_write_b32 with the row offset as the instruction's immediate: left to the compiler, pairs of
// these stores become ds_write2_b32, whose 8-bit offsets need a fresh base-address VGPR every ~4 rows (seven address
// registers for a 16-row tile, at a point where the kernel has none to spare).  LDS operations of a wave execute in
// order, so later compiler-generated reads of the rows see the data; the "memory" clobber keeps the program order.
template <int ROW>
__device__ __forceinline__ void lds_row_store(unsigned lane_addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lane_addr), "v"(v), "n"(ROW * XS * 4) : "memory");
}
template <int ROW0, int N, int SRC0 = 0, int NSRC>
__device__ __forceinline__ void lds_rows_store(unsigned lane_addr, const float (&src)[NSRC]) {
    if constexpr (N > 0) {
        lds_row_store<ROW0>(lane_addr, (SRC0 < NSRC) ? src[SRC0 < NSRC ? SRC0 : 0] : 0.0f);
        lds_rows_store<ROW0 + 1, N - 1, SRC0 + 1, NSRC>(lane_addr, src);
    }
}

// The bookkeeping of a window-spanning launch's window (nsf_bookkeep.h) as a CALL: inlined into the training kernel its ~60
// registers would be live next to everything the two-wave builds keep around their loop (54-84 VGPRs spilled, some of them in the
// loop); as a call the caller's live registers are saved around it once per window, where it does not matter.
__device__ __attribute__((noinline)) void bookkeep_window(const BookArgs* b) { bookkeep_body(*b); }

// =============================================================================================
// dim-major training kernel (L == 1, NLL): ONE WAVE = ONE DIM x T TILES, a block = W waves of ONE (clique, dim).
//
// Why: in nsf_train_kernel the four waves of a block run four different dims and every wave walks through all of
// its dims, so a CU's resident waves read 12-16 different weight sets through the 16 KB scalar data cache: 58 % of
// the scalar loads miss (rocprofv3 SQC_DCACHE_HITS / _MISSES, 64-clique batch) and every miss is an exposed
// ~500-cycle round trip in front of an `s_waitcnt lgkmcnt(0)`.  Here the W waves of a block run the SAME (clique, dim)
// on different particle tiles, each wave sweeps T tiles with that one weight set (staged once per block into an LDS
// panel, nsf_cond_mfma.h) and keeps the weight-gradient MFMA accumulators across them; the block emits one gradient
// copy, and the one workgroup barrier of the main path is the panel's.
// LDS per wave: the particle tile [D][XS] + a 16-row staging tile + [H][XS] h1 (the [h|1] operands of the gradient
// GEMMs are parked in registers while the staging rows are re-used for the other operand): 10.3 KB at D = 15, so the
// register allocation (3 waves per SIMD) decides the occupancy, not LDS.
//
// Grid = (8, blocks per group, ceil(groups / 8)), group = (clique, dim), long dims first.  Workgroups are dealt
// round-robin over the 8 XCDs in linear order (x fastest), so blockIdx.x picks the XCD and the blocks of one group
// (blockIdx.y) share it and its L2: they read the same parameters and the same gradient copies (fused Adam), which the
// group's blocks of the previous launch wrote through that L2 if the dispatcher kept its rotation (a speed matter only,
// never correctness).  No integer division in the prologue: group / cliques is a multiply-high by a host-made
// reciprocal, tiles per wave and waves per block are powers of two, and the clique's descriptor arrives in ONE scalar
// load (from the device array, or -- single-clique calls -- from the kernel-argument segment itself).
// =============================================================================================
#undef STAMP_SEL
#undef STAMP_SLOT
#define STAMP_SEL (bx == 0 && by == 0)
#define STAMP_SLOT ((w + i * 4) & 63)
// Kernel arguments: what a block needs to FIND its work comes first, as scalars -- the unit is compiled with
// -amdgpu-kernarg-preload-count=16, so these 40 bytes are in SGPRs when the wave starts and the clique's descriptor can
// be requested with the first instruction (two dependent ~0.4 us round trips -- kernel arguments, then descriptor --
// become one).  Launches of at most TRAIN1_KERNARG_CLIQUES cliques carry the descriptors themselves in the argument
// segment (`few`): the kernel-argument segment's address is known at wave start as well.
struct Train1Head {
    const nfisam_clique* cliques;   // device array of descriptors, or nullptr: the descriptors are in `few`
    const uint32_t* panel_map;      // kernel-layout parameter index -> LDS word(s) of the conditioner panel (nsf_cond_mfma.h)
    unsigned magic_cliques;         // ceil(2^32 / cliques) (0: one clique): group / cliques without a division
    int groups;                     // (clique, dim) groups = cliques x largest D
    int grid_cliques;               // cliques of the launch
    int xrows;                      // rows of a wave's particle tile in LDS (largest D of the launch)
    int shifts;                     // log2 tiles per wave | log2 waves per block << 8
};
constexpr int TRAIN1_KERNARG_CLIQUES = 8;
struct Train1Few { nfisam_clique c[TRAIN1_KERNARG_CLIQUES]; };
constexpr int TRAIN1_FEW_OFFSET = 40 + (int)((sizeof(TrainArgs) + 7) / 8 * 8);   // kernel-argument offset of `few` (asserted on the host)

// PERSIST (chunk-persistent form, launches that are resident at once): the block stays for a.persist_iters iterations.  With
// one layer a (clique, dim) group is an optimisation problem of its own, so only the group's blocks (one XCD by the grid's
// construction: blockIdx.x = XCD) have to meet per iteration: every block writes its gradient copy as in the plain form,
// the group passes ONE barrier (a counter in the clique's workspace; the copies and the Adam state alternate between two
// buffers, so nobody overwrites what a slower block still reads), and the fused Adam update of the next iteration reads
// the copies back from the XCD's L2 (device-scope loads: the CU's vector cache may hold the lines of two iterations ago).
// The particle tile stays in LDS; no kernel boundary, no cold prologue.  Same arithmetic in the same order: bit-identical.
// Waves per SIMD the dim-major kernels are compiled for: three (168 VGPRs) -- what contended launches want (C3's 624 blocks are
// resident at once only at three blocks per CU; the 64-clique batch runs 3 waves per SIMD).  From num_knots 12 up (10 for the
// chunk-persistent form) that allocation spills (K = 12: 17 / 39 VGPRs, K = 15: 65 / 60; reloads in the middle of the dependent
// chain), which is what a LONE wave per SIMD feels: those (K, H <= 8) pairs get a second, LEAN instantiation with 256 VGPRs
// (no scratch) that the launcher takes whenever the launch is resident at two blocks per CU anyway -- single cliques, i.e. every
// fit of a real run (one Plaza-shaped clique, K = 15: 10.6 -> 9.8 us per iteration, K = 12: 9.7 -> 9.45; the reference's examples
// use K = 12 and 15 too: icra_paper/run_nfisam.py:151, toy_examples/R2RangeGaussian_example/five_node_range_gaussian_incremental.py:85).
#ifndef NSF_PLAIN_MAX_K3
#define NSF_PLAIN_MAX_K3 11
#endif
#ifndef NSF_PERSIST_MAX_K3
#define NSF_PERSIST_MAX_K3 9
#endif
#ifndef NSF_PERSIST_WAVES
#define NSF_PERSIST_WAVES 3      // resident waves per SIMD the chunk-persistent instantiation is compiled for (round 3: 2 -- it spilled at 3)
#endif
// ---- the one-launch-per-iteration form: ROUND 3'S KERNEL, VERBATIM ---------------------------------------------------------
// Round 4 rebuilt the chunk-persistent form (nsf_train1_kernel below: loop roots laundered per phase, flag-in-data exchange,
// divided update).  The same source compiled for ONE pass came out 2-5 % slower than round 3's kernel (167 instead of 163
// VGPRs, another schedule: C3 as one launch 14.6 vs 13.95 us, the 64-clique batch 86.0 vs 84.9, scripts/ab.py against a
// build of round 3's tree on one box), so the launches that run one iteration per launch -- every batch too large to be
// resident at once, the replicas' conveyor, the eager tail of a run -- keep round 3's text as it was.
template <int K, int H, bool LEAN = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((H == 16 || LEAN) ? 2 : 3, 8)))
nsf_train1_plain_kernel(const nfisam_clique* h_cliques, const uint32_t* h_panel_map, unsigned h_magic, int h_groups, int h_grid_cliques,
                  int h_xrows, int h_shifts, TrainArgs a, Train1Few few) {
    constexpr bool PERSIST = false;                           // (the text of round 3's kernel: its persistent branches compile away)
    using LY = Layout<K, H>;
    using CP = CondPanel<K, H>;
    constexpr int PoP = LY::PoP;
    constexpr int NT = (PoP + 15) / 16;
    constexpr int NS = TILE / 4;                              // MFMA k-steps over the 64 particles of a tile
    static_assert((H == 16 || H == 8 || H == 4) && NT <= 4, "H <= 8: ga2|ga1 share one 16-row operand tile; H = 16: one tile each, bias chains");
    constexpr bool WIDE_H = (H == 16);                        // no spare column for the bias in [h | 1]: db2 / db1 come from chains against a constant 1
    constexpr int QH = H / 4;                                 // row groups (of 4) of ga2 resp. ga1 in that tile
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int bx = blockIdx.y;                                // tile group inside the (clique, dim)
    // (chains: an iteration may be split into n launches on parallel graph branches, see nfisam_nsf_train_plan_create;
    //  launch c takes the octets of groups c, c + n, ...)
    const int grp = blockIdx.x + 8 * (blockIdx.z * ((h_shifts >> 24) & 0xff) + ((h_shifts >> 16) & 0xff));
    if (grp >= h_groups) return;                              // padding of the group count to a multiple of 8
    const int gq = h_magic != 0u ? (int)__umulhi((unsigned)grp, h_magic) : grp;    // grp / cliques
    const int by = grp - gq * h_grid_cliques;                 // clique
    const int i = h_xrows - 1 - gq;                           // this block's dim: the long ones first
    typedef const __attribute__((address_space(4))) nfisam_clique cclique;
    typedef const __attribute__((address_space(4))) char cchar;
    cclique* cp = (h_cliques != nullptr)
                      ? (cclique*)(h_cliques + by)
                      : (cclique*)((cchar*)__builtin_amdgcn_kernarg_segment_ptr() + TRAIN1_FEW_OFFSET) + by;
    (void)few;
    // the descriptor's pointers are device-memory pointers: say so (generic pointers would compile to flat_ loads and
    // atomics, which count against both memory counters)
    const gfloat* x = (const gfloat*)cp->x;
    const float* kparams = cp->kparams;
    gfloat* G = (gfloat*)cp->kgrad;
    const gfloat* own_m = (const gfloat*)cp->adam_m;
    const gfloat* own_v = (const gfloat*)cp->adam_v;
    typedef __attribute__((address_space(1))) nfisam_train_state gstate;
    gstate* st = (gstate*)cp->state;
    const int n = cp->n;
    const int D = cp->D;
    if (i >= D) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ws = (h_shifts >> 8) & 0xff, ts = h_shifts & 0xff;
    const int W = 1 << ws, T = 1 << ts;
    const int slot = (bx << ws) + w;                          // this wave's tile group
    const int p0 = slot << (6 + ts);
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
    STAMP(10);
#endif
    int st_stop = 0, st_step = 0;
    if (st != nullptr) {
        st_stop = __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_step = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float B = a.B;
    const bool slab = a.slab != 0;
    const size_t gstride = (size_t)LY::count(D);
    gfloat* ring = G + (slab ? (size_t)a.n_copies : (size_t)1) * gstride;
    // fused Adam (nsf_cond_mfma.h): gradient copies and optimiser state alternate between two buffers with the parity
    // of the iteration inside its chunk; the second set sits behind the loss ring: [copies][ring][64][copies][theta|m|v]
    gfloat* const G0 = G;
    gfloat* Gset1 = ring + LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
    gfloat* alt = Gset1 + (size_t)a.n_copies * gstride;
    int it = PERSIST ? 0 : a.iter_idx;                        // iteration inside the chunk
    int par = (a.fused_adam != 0) ? (it & 1) : 0;
    bool pending = a.fused_adam != 0 && it > 0;
    const gfloat* Gprev = par ? G0 : Gset1;                   // copy 0 of the previous iteration
    if (par) G = Gset1;
    if (slab) G += (size_t)bx * gstride;                      // one gradient copy per block
    const int xrows = h_xrows;                                // rows of a particle tile in LDS (largest D of the launch)
    const float* pan = smem + PANEL_BASE;                     // the block's conditioner panel (nsf_cond_mfma.h)
    // a row of ones for the bias / unused columns of the MFMA operands: a lane that must supply 1.0 READS it from here
    // (its operand pointer is selected once per tile) instead of selecting 1.0 over a loaded value at every k-step.
    // Every wave writes the same 68 words; its own LDS operations are in order, so it reads what it wrote.
    float* ones = smem + PANEL_BASE + CP::floats(xrows);
    float* tiles0 = ones + ONES_ROW;
    const int wave_floats = train1_wave_floats(xrows, H);
    // fixed-size rows first: their offsets from the wave's base are immediates of the LDS instructions (fewer address registers)
    float* stg = tiles0 + (size_t)w * wave_floats;            // [16][XS] staging rows
    float* hrow = stg + 16 * XS;                              // [H][XS] h1 of the tile (an operand of the last gradient GEMM)
    const unsigned stg_lane = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stg + lane);   // LDS byte address
    float* xt = hrow + H * XS;                                // [xrows][XS] particle tile, dimension-major
    float* ctacc = xt + xrows * XS;                           // D > 16: dW0 rows 16.. of the wave, summed over its tiles
    const int r16 = lane & 15, kq = lane >> 4;
    ones[lane] = 1.0f;
    if (lane < ONES_ROW - 64) ones[64 + lane] = 1.0f;
    const bool merged = (i <= 16 - (H + 1));                  // the two last gradient GEMMs share one operand tile (see phase B)
    gfloat* Gb = G + LY::off(i > 0 ? i : 1);
    const int members = (((n + (TILE << ts) - 1) >> (6 + ts)) + W - 1) >> ws;     // blocks of this (clique, dim) group with a tile
    if (PERSIST && bx >= members) return;
    const bool has_tile = p0 < n;

    STAMP_DECL
    STAMP(0);
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
    if (lane < 16) smem[PANEL_BASE - 64 + w * 16 + lane] = 0.0f;
    { float d0_ = 0.f, d1_ = 0.f; PSTAMP(0, d0_, d1_); }
#endif
    f32x4 cacc[NT], c1 = {0.f, 0.f, 0.f, 0.f}, c0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cb2[NT], cb1 = {0.f, 0.f, 0.f, 0.f};                 // WIDE_H only
#pragma unroll
    for (int t = 0; t < NT; ++t) { cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float r0 = 0.0f, lossv = 0.0f;

    // tile loader: every lane reads the columns 0..i of its own particle row (16-byte loads at the row's 4-byte
    // alignment; the conditioner's inputs and x_i itself) and drops them into the dimension-major LDS tile: no index
    // arithmetic, no column the dim does not need.  Rows beyond n re-read row n-1 (masked out of loss and gradient).
    // Fetch (global -> registers) and store (registers -> LDS) are separate so that the first tile's loads are in
    // flight while the block stages its weight panel.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    auto fetch = [&](int pt, int c0, float (&xr)[16]) {
        const int pr = (pt + lane < n) ? pt + lane : n - 1;
        const gfloat* row = x + (size_t)pr * D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = c0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k <= i) {                                 // wave-uniform
                if (k + 3 < D) {
                    const f32x4u u = *(const __attribute__((address_space(1))) f32x4u*)(row + k);
                    v = f32x4{u.x, u.y, u.z, u.w};
                } else {
                    v.x = row[k];
                    if (k + 1 < D) v.y = row[k + 1];
                    if (k + 2 < D) v.z = row[k + 2];
                }
            }
            xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
        }
    };
    auto store = [&](int c0, const float (&xr)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (c0 + u <= i) xt[(c0 + u) * XS + lane] = xr[u];
    };
    auto load_tile = [&](int pt, int c_first) {
        for (int c0 = c_first; c0 <= i; c0 += 16) {
            float xr[16];
            fetch(pt, c0, xr);
            store(c0, xr);
        }
    };
  for (;;) {                                                  // (PERSIST: the iterations of the chunk; else one pass)
    {
        // the one workgroup barrier in front of the tile loop: the block's waves share the (clique, dim) and so the panel
        float xr[16];
        const bool first = !PERSIST || it == 0;               // the particle tile stays in LDS between the iterations of a chunk
        if (!PERSIST && p0 < n) fetch(p0, 0, xr);               // (PERSIST: once per chunk, after the staging: fewer live registers)
        {
            FusedAdam fa;
            fa.grads = pending ? Gprev : nullptr;
            fa.gstride = gstride;
            fa.copies = members;
            // state before the pending update: the buffer of the previous iteration's parity (even: the clique's own)
            const gfloat* own_t = (const gfloat*)kparams;
            const bool src_alt = pending && par == 0;
            const gfloat* t_src = src_alt ? alt : own_t;
            fa.m_src = src_alt ? alt + gstride : own_m;
            fa.v_src = src_alt ? alt + 2 * gstride : own_v;
            const bool writer = pending && bx == 0;
            fa.t_dst = writer ? (src_alt ? (gfloat*)own_t : alt) : nullptr;
            fa.m_dst = writer ? (src_alt ? (gfloat*)own_m : alt + gstride) : nullptr;
            fa.v_dst = writer ? (src_alt ? (gfloat*)own_v : alt + 2 * gstride) : nullptr;
            if (!stage_cond_panel<K, H, PERSIST>(smem, (const float*)t_src, fa, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, a, n, it)) return;
            __syncthreads();
        }
        if (!PERSIST && p0 >= n) return;
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(11);
#endif
        if (first && has_tile) {
            if (PERSIST) fetch(p0, 0, xr);
            store(0, xr);
            load_tile(p0, 16);
        }
    }

    if constexpr (PERSIST) {                                  // (zeroed HERE, not at the loop's end: nothing of them is live across the staging)
        lossv = 0.0f; r0 = 0.0f;
        c1 = f32x4{0.f, 0.f, 0.f, 0.f}; c0 = c1; cb1 = c1;
#pragma unroll
        for (int t = 0; t < NT; ++t) { cacc[t] = c1; cb2[t] = c1; }
    }
    for (int tt = 0; tt < T && has_tile; ++tt) {
        const int pt = p0 + tt * TILE;
        if (pt >= n) break;
        PSTAMP(1, lossv, r0);
        if (tt > 0) load_tile(pt, 0);
        wave_lds_sync();
        PSTAMP(2, lossv, r0);
        const bool valid = pt + lane < n;
        float h1[H], h2[H], th[PoP], gth[PoP];
        if (i == 0) {
#pragma unroll
            for (int o = 0; o < PoP; o += 4) {
                const cm_f32x4 v4 = *(const cm_f32x4*)(pan + o);
                th[o] = v4[0]; th[o + 1] = v4[1]; th[o + 2] = v4[2]; th[o + 3] = v4[3];
            }
        } else {
            cond_forward_mfma<K, H>(pan, i, CP::s0_of(i), xt, XS, lane, lane, h1, h2, th);
            // operands of the gradient GEMMs, parked while the lanes are busy with the spline
            lds_rows_store<0, H, 0, H>(stg_lane, h2);
            lds_rows_store<16, H, 0, H>(stg_lane, h1);         // = hrow
        }
        PSTAMP(3, th[0], th[PoP - 1]);
        SplineT<K> S;
        float z, lad;
        spline_train_fwd<K, PoP>(xt[i * XS + lane], th, B, S, z, lad);
        PSTAMP(4, z, lad);
        if (valid) lossv += 0.5f * z * z - lad;
        spline_train_bwd<K, PoP>(S, B, valid ? z : 0.0f, valid ? -1.0f : 0.0f, gth);
        PSTAMP(5, gth[0], gth[LY::HP]);
        if (i == 0) {   // init_param: plain sum over particles of gth
            constexpr int N0 = (PoP <= 32) ? 32 : 64;
            float v[N0];
#pragma unroll
            for (int t = 0; t < N0; ++t) v[t] = (t < PoP) ? gth[t] : 0.0f;
            r0 += butterfly<N0>(v, lane);
            wave_lds_sync();                                    // the tile is overwritten by the next iteration's loads
            continue;
        }
        // ---- per-particle back-propagation through the conditioner (4x4x1 MFMA chains, nsf_cond_mfma.h) ----
        float ga2[H], ga1[H];
        cond_backward_mfma<K, H>(pan, lane, gth, h1, h2, ga2, ga1);
        PSTAMP(6, ga1[0], ga2[H - 1]);
        // ---- weight gradients on the matrix cores (see nsf_train_kernel); operand rows: lane&15 = feature,
        //      lane>>4 = particle inside the k-group of 4; the bias column (and the unused columns) multiply 1 ----
        // The 16 staging rows carry four generations of operands: [h2 | h1], gth tile 0, gth tile 1, [ga2 | ga1].  A
        // generation is read into registers completely, then the NEXT one is written before this one's MFMAs are
        // issued: the LDS writes complete under the 16 x 32 MFMA cycles and only the reads' round trip stays exposed.
        const float* pa = stg + r16 * XS + kq;
        float breg[NS], areg[NS];
        {
            wave_lds_sync();
            const float* pah = ((r16 < H) ? stg + r16 * XS : ones) + kq;
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) breg[s4] = pah[4 * s4];
            wave_lds_sync();
        }
        PSTAMP(7, breg[0], breg[NS - 1]);
        {   // phase A: dW2t | db2 = [h2, 1]^T (x) gth ;  phase B: dW1t | db1 = [h1,1]^T (x) ga2 ;  dW0t | db0 = [x,1]^T (x) ga1
            lds_rows_store<0, 16, 0, PoP>(stg_lane, gth);
            wave_lds_sync();
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
            wave_lds_sync();
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                if (t == 1 && NT > 1) lds_rows_store<0, 16, 16, PoP>(stg_lane, gth);
                if (t == 2 && NT > 2) lds_rows_store<0, 16, 32, PoP>(stg_lane, gth);
                if (t == 3 && NT > 3) lds_rows_store<0, 16, 48, PoP>(stg_lane, gth);
                if (t == NT) {
                    if constexpr (WIDE_H) {
                        lds_rows_store<0, 16, 0, H>(stg_lane, ga2);
                    } else {
                        lds_rows_store<0, H, 0, H>(stg_lane, ga2);
                        lds_rows_store<H, H, 0, H>(stg_lane, ga1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    cacc[t - 1] = mfma4(areg[s4], breg[s4], cacc[t - 1]);
                    if constexpr (WIDE_H) cb2[t - 1] = mfma4(areg[s4], 1.0f, cb2[t - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_sync();
                if (t < NT) {
#pragma unroll
                    for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
                    wave_lds_sync();
                }
            }
            if constexpr (WIDE_H) {
                // ga2 (16 rows) x [h1]: dW1t, and x 1: db1; then ga1 (16 rows) x [x_0 .. x_{i-1} | 1]: dW0t | db0
                const float* pb1 = hrow + r16 * XS + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
                wave_lds_sync();
                lds_rows_store<0, 16, 0, H>(stg_lane, ga1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    c1 = mfma4(areg[s4], pb1[4 * s4], c1);
                    cb1 = mfma4(areg[s4], 1.0f, cb1);
                }
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_sync();
                const float* pb0 = ((r16 < i) ? xt + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c0 = mfma4(areg[s4], pb0[4 * s4], c0);
                }
            } else if (merged) {
                // i <= 16 - (H + 1): the columns [h1 (H) | 1 | x_0 .. x_{i-1}] of BOTH products fit one 16-column operand
                // (they share the bias column): one MFMA chain instead of two.  Rows 0..H-1 (ga2) x columns 0..H give
                // dW1t | db1, rows H.. (ga1) x columns H.. give db0 | dW0t; the cross terms are not used.
                const int kx = r16 - (H + 1);
                const bool one = (r16 == H) || kx >= i;
                const float* pbm = (one ? ones : ((r16 < H) ? hrow + r16 * XS : xt + kx * XS)) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c1 = mfma4(areg[s4], pbm[4 * s4], c1);
                }
            } else {
                const float* pb0 = ((r16 < i) ? xt + r16 * XS : ones) + kq;     // input columns 0..15 (column i = bias)
                const float* pb1 = ((r16 < H) ? hrow + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c1 = mfma4(areg[s4], pb1[4 * s4], c1);
                    c0 = mfma4(areg[s4], pb0[4 * s4], c0);
                }
            }
            for (int ct = 1; ct * 16 <= i; ++ct) {              // D > 16: further column tiles add into the wave's own copy
                const int cab = ct * 16 + r16;
                const float* pb0 = ((cab < i) ? xt + cab * XS : ones) + kq;
                f32x4 cx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) cx = mfma4(areg[s4], pb0[4 * s4], cx);
                if ((WIDE_H || (kq >= QH && kq < 2 * QH)) && cab <= i) {
                    float* dst = &ctacc[(cab - 16) * H + 4 * (WIDE_H ? kq : kq - QH)];
                    if (slab) {
                        if (tt > 0) { cx.x += dst[0]; cx.y += dst[1]; cx.z += dst[2]; cx.w += dst[3]; }
                        dst[0] = cx.x; dst[1] = cx.y; dst[2] = cx.z; dst[3] = cx.w;
                    } else {
                        gsink4(&Gb[cab * H + 4 * (WIDE_H ? kq : kq - QH)], cx, false);
                    }
                }
            }
            wave_lds_sync();
        }
        PSTAMP(8, c1.x, cacc[0].x);
    }

    // ---- the gradient of this dim's parameter block ----
    if (slab) {
        // One copy per BLOCK: every wave lays its fragment out in parameter order in its own (now free) rows, 16 bytes per
        // store (the accumulators hold four consecutive parameters), the block's threads add the fragments in wave order
        // and write the copy with consecutive 16-byte stores.
        float* frag = stg;                                     // 24 rows: room for every dim's block
        if (!has_tile) {
            // (PERSIST: a wave without particles stays for the staging of the following iterations)
        } else if (i == 0) {
            if (lane < PoP) frag[lane] = r0;
        } else {
            float* fw = frag + LY::oW2(i);
            if (r16 <= H) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq + 3 < PoP) *(f32x4*)&fw[r16 * PoP + 16 * t + 4 * kq] = cacc[t];
            }
            if (kq < QH && r16 <= H) *(f32x4*)&(frag + LY::oW1(i))[r16 * H + 4 * kq] = c1;
            if constexpr (WIDE_H) {                           // the bias rows (every column of a bias chain holds the same sums) + dW0t | db0
                if (r16 == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (16 * t + 4 * kq + 3 < PoP) *(f32x4*)&fw[H * PoP + 16 * t + 4 * kq] = cb2[t];
                    *(f32x4*)&(frag + LY::oW1(i))[H * H + 4 * kq] = cb1;
                }
                if (r16 <= i) *(f32x4*)&frag[r16 * H + 4 * kq] = c0;
            } else if (merged) {                                     // columns H.. of the shared tile: bias first, then x_0..x_{i-1}
                const int k0 = (r16 == H) ? i : r16 - (H + 1);
                if (kq >= QH && kq < 2 * QH && (r16 == H || (r16 > H && k0 < i))) *(f32x4*)&frag[k0 * H + 4 * (kq - QH)] = c1;
            } else if (kq >= QH && kq < 2 * QH && r16 <= i) {
                *(f32x4*)&frag[r16 * H + 4 * (kq - QH)] = c0;
            }
            for (int e = lane; e < (i - 15) * H; e += 64) frag[16 * H + e] = ctacc[e];     // D > 16 (rows 16..i)
        }
        // the block's loss: the waves' sums added in wave order by ONE thread (below), then one atomic per block into a ring
        // slot shared by at most two blocks of the clique while D x blocks <= 256 (LOSS_SLOTS = 128) -- a sum of two floats does not depend on
        // their order, so the loss record is the same whichever way the launches and the waves happen to be timed
        {
            const float wtot = wave_sum(lossv);
            if (lane == 0) xt[64] = has_tile ? wtot : 0.0f;   // a padding word of the tile's first row
        }
        __syncthreads();                                      // waves without a tile left before the panel barrier
        const int waves_c = (n + (TILE << ts) - 1) >> (6 + ts);
        const int alive = (waves_c - (bx << ws) < W) ? waves_c - (bx << ws) : W;
        const int nj4 = ((i == 0) ? PoP : LY::block(i)) >> 2;
        gvf4_t* Gc = (gvf4_t*)(G + ((i == 0) ? 0 : LY::off(i)));
        if (threadIdx.x == 64 * (alive - 1)) {                 // (the block's LAST wave with a tile: it has the fewest fragments to add below)
            float bl = 0.0f;
            for (int ww = 0; ww < alive; ++ww) bl += tiles0[(size_t)ww * wave_floats + (16 + H) * XS + 64];
            gfloat* dst = (st != nullptr) ? &ring[((st_step + it) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                    (((i * members + bx) >> 1) & (LOSS_SLOTS - 1))]
                                          : (gfloat*)a.loss_sum;
            if (dst != nullptr) gsink(dst, bl, false);
        }
        for (int e = threadIdx.x; e < nj4 && w < alive; e += 64 * alive) {
            f32x4 sum = *(const f32x4*)&tiles0[4 * e];
            for (int ww = 1; ww < alive; ++ww) {
                const f32x4 o = *(const f32x4*)&tiles0[(size_t)ww * wave_floats + 4 * e];
                sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
            }
            Gc[e] = sum;
        }
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(12);
#endif
    } else if (i == 0) {
        if (lane < PoP) gsink(&G[lane], r0, false);
    } else {
        gfloat* Gw = Gb + LY::oW2(i);
        {
            // atomics: (H+1) x PoP floats through LDS in two halves of the staging tile, flat order (consecutive addresses)
            constexpr int TOT = (WIDE_H ? H : H + 1) * PoP;     // (H = 16: the bias row comes from its own chain, below)
            static_assert(TOT <= 16 * XS, "the transposed dW2 block fits the staging rows");
            if (r16 <= H) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq + 3 < PoP) {
                        float* d = &stg[r16 * PoP + 16 * t + 4 * kq];
                        d[0] = cacc[t].x; d[1] = cacc[t].y; d[2] = cacc[t].z; d[3] = cacc[t].w;
                    }
            }
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < (TOT + 63) / 64; ++c) {
                const int f = c * 64 + lane;
                if (f < TOT) gsink(&Gw[f], stg[f], false);
            }
        }
        if (kq < QH && r16 <= H) gsink4(&(Gb + LY::oW1(i))[r16 * H + 4 * kq], c1, false);
        if constexpr (WIDE_H) {
            if (r16 == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gw[H * PoP + 16 * t + 4 * kq], cb2[t], false);
                gsink4(&(Gb + LY::oW1(i))[H * H + 4 * kq], cb1, false);
            }
            if (r16 <= i) gsink4(&Gb[r16 * H + 4 * kq], c0, false);
        } else if (merged) {
            const int k0 = (r16 == H) ? i : r16 - (H + 1);
            if (kq >= QH && kq < 2 * QH && (r16 == H || (r16 > H && k0 < i))) gsink4(&Gb[k0 * H + 4 * (kq - QH)], c1, false);
        } else if (kq >= QH && kq < 2 * QH && r16 <= i) {
            gsink4(&Gb[r16 * H + 4 * (kq - QH)], c0, false);
        }
    }
    PSTAMP(9, lossv, r0);
    const float tot = slab ? 0.0f : wave_sum(lossv);
    if (lane == 0 && has_tile && !slab) {
        gfloat* dst = (st != nullptr) ? &ring[((st_step + it) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                ((slot * 7 + i * 13) & (LOSS_SLOTS - 1))]
                                      : (gfloat*)a.loss_sum;
        if (dst != nullptr) gsink(dst, tot, false);
    }
    if constexpr (!PERSIST) {
        break;
    } else {
        if (++it >= a.persist_iters) break;
        // ---- the group's blocks meet: this block's copy is in L2 (vmcnt: its stores are acknowledged), then everybody's is ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* ctr = (unsigned*)(ring + LOSS_RING * LOSS_SLOTS) + i;       // zero at the start of every chunk (nsf_bookkeep_kernel)
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it * members);
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {                   // a member that never became resident: give up loudly (non-finite loss -> domain error)
                    if (st != nullptr) ring[((st_step + it) & (LOSS_RING - 1)) * LOSS_SLOTS] = __builtin_nanf("");
                    break;
                }
            }
        }
        __syncthreads();
        par = it & 1;
        pending = a.fused_adam != 0;
        Gprev = par ? G0 : Gset1;
        G = (par ? Gset1 : G0) + (slab ? (size_t)bx * gstride : (size_t)0);
        Gb = G + LY::off(i > 0 ? i : 1);
    }
  }
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
    if (STAMP_SEL && lane < 16) g_stamps[STAMP_SLOT * 32 + 16 + lane] = (unsigned long long)((unsigned*)smem)[PANEL_BASE - 64 + w * 16 + lane];
#endif
}


// chunk-persistent form: floats of each of the three arrays (theta | m | v of the block's dim) kept in LDS behind the waves' tiles
template <int K, int H>
__host__ __device__ constexpr int persist_keep_stride(int max_D) {
    using LY = Layout<K, H>;
    const int b = max_D > 1 ? LY::block(max_D - 1) : 0;
    return ((b > LY::PoP ? b : LY::PoP) + 3) & ~3;
}
// WIDE (PERSIST only, round 5): groups of more than eight blocks -- a clique of more than 2048 particles, up to
// PERSIST_MAX_COPIES = 16 gradient copies per (clique, dim) -- whose tagged copies are fetched in two passes and summed in
// nsf_adam_kernel's lane-partial order for that many copies (nsf_cond_mfma.h: stage_cond_panel_persist*_wide).  A second
// instantiation rather than a run-time branch: the common case keeps the code round 4 measured.
template <int K, int H, bool PERSIST = false, bool LEAN = false, bool WIDE = false, bool SPL = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((H == 16 || LEAN || SPL || (PERSIST && NSF_PERSIST_WAVES < 3)) ? 2 : 3, 8)))
nsf_train1_kernel(const nfisam_clique* h_cliques, const uint32_t* h_panel_map, unsigned h_magic, int h_groups, int h_grid_cliques,
                  int h_xrows, int h_shifts, TrainArgs a, Train1Few few) {
    using LY = Layout<K, H>;
    using CP = CondPanel<K, H>;
    constexpr int PoP = LY::PoP;
    constexpr int NT = (PoP + 15) / 16;
    // SPL (round 6, nsf_half.h): two lanes per particle -- a wave covers 32 particles, lanes p and p + 32 share particle p
    constexpr int TP = SPL ? 32 : TILE;                       // particles per wave-tile
    constexpr int PSH = SPL ? 5 : 6;                          // log2 of it
    constexpr int NS = TP / 4;                                // MFMA k-steps over the particles of a tile
    static_assert(!SPL || (H == 8 && LY::HP == 16), "two lanes per particle: hidden_dim 8, sixteen theta columns per half (num_knots 9 .. 11)");
    static_assert((H == 16 || H == 8 || H == 4) && NT <= 4, "H <= 8: ga2|ga1 share one 16-row operand tile; H = 16: one tile each, bias chains");
    constexpr bool WIDE_H = (H == 16);                        // no spare column for the bias in [h | 1]: db2 / db1 come from chains against a constant 1
    constexpr int QH = H / 4;                                 // row groups (of 4) of ga2 resp. ga1 in that tile
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // Grid dim3(8, blocks per group, octets): workgroups are dealt round-robin over the 8 XCDs in linear order, so the
    // blocks of a (clique, dim) group (8 apart) share one XCD's L2.  A SPEED matter only, also for the chunk-persistent form
    // (its exchange goes through agent-scope write-through stores and loads, see below); h_shifts bit 7 (debug knob
    // NFISAM_PERSIST_SCATTER=1, PERSIST only) transposes the two grid axes so that a group's blocks are 1 apart, i.e. on
    // DIFFERENT XCDs: tests/test_hip_parity.py holds that launch to the same bits.
    const bool scatter = PERSIST && (h_shifts & 0x80) != 0;
    const int bx = scatter ? blockIdx.x : blockIdx.y;         // tile group inside the (clique, dim)
    // (chains: an iteration may be split into n launches on parallel graph branches, see nfisam_nsf_train_plan_create;
    //  launch c takes the octets of groups c, c + n, ...)
    const int grp = (scatter ? blockIdx.y : blockIdx.x) + 8 * (blockIdx.z * ((h_shifts >> 24) & 0xff) + ((h_shifts >> 16) & 0xff));
    if (grp >= h_groups) return;                              // padding of the group count to a multiple of 8
    const int gq = h_magic != 0u ? (int)__umulhi((unsigned)grp, h_magic) : grp;    // grp / cliques
    // clique: rotated by the dim's row, so that the dims of ONE clique spread over the XCDs also when the number of cliques is
    // a multiple of 8 (XCD = grp mod 8 would otherwise be the clique itself: the eight C3 cliques, D = 6 .. 12, put 48 .. 96
    // blocks on an XCD -- the widest clique filled its XCD's 96 places while the narrowest left half of its own idle)
    // (PERSIST only: the one-launch-per-iteration form re-reads the particle tile every launch, and a clique whose dims share
    //  an XCD fetches it into ONE L2; its C3 time did not change with the rotation -- 14.5 us either way)
    int by = grp - gq * h_grid_cliques;
    if constexpr (PERSIST) {
        by += gq;
        by -= (h_magic != 0u ? (int)__umulhi((unsigned)by, h_magic) : by) * h_grid_cliques;
    }
    const int i = h_xrows - 1 - gq;                           // this block's dim: the long ones first
    typedef const __attribute__((address_space(4))) nfisam_clique cclique;
    typedef const __attribute__((address_space(4))) char cchar;
    cclique* cp0 = (h_cliques != nullptr)
                      ? (cclique*)(h_cliques + by)
                      : (cclique*)((cchar*)__builtin_amdgcn_kernarg_segment_ptr() + TRAIN1_FEW_OFFSET) + by;
    (void)few;
    typedef __attribute__((address_space(1))) nfisam_train_state gstate;
    if (i >= cp0->D) return;
    const int lane0 = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ws = (h_shifts >> 8) & 0xff, ts = h_shifts & 0x3f;
    const int W = 1 << ws, T = 1 << ts;
#if defined(NSF_STAMPS)
    const int lane = lane0;                                   // (the stamp macros name `lane`; the loop below declares its own)
#endif
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
    STAMP(10);
#endif
    int st_stop = 0, st_step = 0;
    {
        gstate* st0 = (gstate*)cp0->state;
        if (st0 != nullptr) {
            st_stop = __hip_atomic_load(&st0->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st_step = __hip_atomic_load(&st0->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (PERSIST) {                                  // (uniform values that arrive in VGPRs: scalar registers around the loop)
        st_stop = __builtin_amdgcn_readfirstlane(st_stop);
        st_step = __builtin_amdgcn_readfirstlane(st_step);
    }
    {
        // a row of ones for the bias / unused columns of the MFMA operands: a lane that must supply 1.0 READS it from here
        // (its operand pointer is selected once per tile) instead of selecting 1.0 over a loaded value at every k-step.
        // Every wave writes the same 68 words; its own LDS operations are in order, so it reads what it wrote.
        float* ones0 = smem + PANEL_BASE + CP::floats(h_xrows);
        ones0[lane0] = 1.0f;
        if (lane0 < ONES_ROW - 64) ones0[64 + lane0] = 1.0f;
    }
    if (PERSIST && bx >= (((((int)cp0->n + (TP << ts) - 1) >> (PSH + ts)) + W - 1) >> ws)) return;   // a block without a tile
    if (PERSIST && threadIdx.x == 0) smem[1] = 0.0f;          // the block's abort word (words 1-3 in front of the panel are free)
    // (test knob NFISAM_PERSIST_DROP=1, h_shifts bit 6: block 1 of group 0 leaves at once -- a member that "never became
    //  resident"; its group must time out, raise the abort flag and end the run with NFISAM_ERR_STALL)
    if (PERSIST && (h_shifts & 0x40) != 0 && grp == 0 && bx == 1) return;

    STAMP_DECL
#if !(defined(NSF_STAMPS) && NSF_STAMPS == 2)
    STAMP(0);
#endif
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
    if (lane0 < 16 && w < 4) smem[PANEL_BASE - 64 + w * 16 + lane0] = 0.0f;
#endif
    f32x4 cacc[NT], c1 = {0.f, 0.f, 0.f, 0.f}, c0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cb2[NT], cb1 = {0.f, 0.f, 0.f, 0.f};                 // WIDE_H only
#pragma unroll
    for (int t = 0; t < NT; ++t) { cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb2[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float r0 = 0.0f, lossv = 0.0f;
    int it = PERSIST ? 0 : a.iter_idx;                        // iteration inside the chunk
    // ROOMY: a two-wave build (256 VGPRs: SPL, LEAN).  The three-wave builds re-derive everything per phase from laundered roots
    // so that nothing lives around the loop (below); a roomy build lets the compiler hoist the loop-invariant derivations
    // instead (round 6: one Plaza clique 7.63 -> 7.40 us per iteration).
    // (Tried on top of it and dropped, round 6: requesting the first look at the tagged copies at the TOP of the iteration, in
    //  front of the derivations -- the copies of the slower blocks are not out yet, the look fails and costs a second round
    //  trip, and 16 blocks x 256 threads x 32 early loads stand in the way of the very stores they wait for: one Plaza clique
    //  7.34 -> 11.2 us per iteration with sixteen copies, 7.41 -> 7.6 with eight.  The staging's own first look is just in time.)
    constexpr bool ROOMY = PERSIST && (SPL || LEAN);

  for (;;) {                                                  // (PERSIST: the iterations of the chunk; else one pass)
    // Everything the body needs is (re)derived INSIDE the loop from two laundered roots -- the thread index and the address
    // of the clique's descriptor: values defined in front of a loop and used in its body stay alive around the back
    // edge, ~35 VGPRs and ~150 SGPRs more than the one-pass kernel needs, which is what kept the chunk-persistent form
    // from three waves per SIMD (round 3: 37 VGPRs spilled to scratch, reloads in the middle of the dependent chain).
    // The descriptor comes back with one s_load_dwordx16 per iteration (scalar cache), under the staging's own loads.
    int tl_ = threadIdx.x;
    cclique* cp = cp0;
    // (PERSIST: the launch's argument block is read through a laundered pointer into the kernel-argument segment, field by
    //  field where a phase needs it -- ~20 scalars that would otherwise be loaded once and held around the whole loop; with the
    //  ~45 loop-carried scalars the body spilled 166 SGPRs into VGPR lanes, 350 v_readlane / v_writelane per iteration and wave)
    typedef const __attribute__((address_space(4))) TrainArgs cargs;
    cargs* ap_ = (cargs*)((cchar*)__builtin_amdgcn_kernarg_segment_ptr() + 40);
    if constexpr (PERSIST && !ROOMY) asm volatile("" : "+v"(tl_), "+s"(cp), "+s"(ap_));
#define AF(f) (PERSIST ? ap_->f : a.f)
    const int lane = tl_ & 63;
    // the descriptor's pointers are device-memory pointers: say so (generic pointers would compile to flat_ loads and
    // atomics, which count against both memory counters)
    const gfloat* x = (const gfloat*)cp->x;
    const float* kparams = cp->kparams;
    gfloat* G = (gfloat*)cp->kgrad;
    const gfloat* own_m = (const gfloat*)cp->adam_m;
    const gfloat* own_v = (const gfloat*)cp->adam_v;
    gstate* st = (gstate*)cp->state;
    const int n = cp->n;
    const int D = cp->D;
    const int slot = (bx << ws) + w;                          // this wave's tile group
    const int p0 = slot << (PSH + ts);
    float B = AF(B);
    // (round 6) What the loop's top and the staging read of the launch's arguments and of the descriptor, requested as ONE
    // batch of scalar loads: left to itself the compiler sinks every load to its use, behind the branches in between, and
    // a three-wave build's iteration began with five scalar-cache round trips one after the other (s_load .. s_waitcnt
    // lgkmcnt(0) x 5: ~1 k of the 1.4 k cycles of "loop top").  The empty asm is a use of all of them in one place.
    struct TopArgs { int slab, n_copies, fused_adam, persist_split, persist_spins, max_iters; float lr, b1, b2, eps, lb1, lb2; } ta;
    ta.slab = AF(slab); ta.n_copies = AF(n_copies); ta.fused_adam = AF(fused_adam); ta.persist_split = AF(persist_split);
    ta.persist_spins = AF(persist_spins); ta.max_iters = AF(max_iters);
    ta.lr = AF(adam.lr); ta.b1 = AF(adam.beta1); ta.b2 = AF(adam.beta2); ta.eps = AF(adam.eps); ta.lb1 = AF(log_b1); ta.lb2 = AF(log_b2);
    if constexpr (PERSIST && !ROOMY) {
        asm volatile("" ::"s"(x), "s"(kparams), "s"(G), "s"(own_m), "s"(own_v), "s"(st), "s"(n), "s"(D), "s"(B), "s"(ta.slab), "s"(ta.n_copies),
                     "s"(ta.fused_adam), "s"(ta.persist_split), "s"(ta.persist_spins), "s"(ta.max_iters), "s"(ta.lr), "s"(ta.b1), "s"(ta.b2),
                     "s"(ta.eps), "s"(ta.lb1), "s"(ta.lb2));
    }
    if constexpr (PERSIST && !ROOMY) asm volatile("" : "+s"(B));         // (the spline's per-bin constants are functions of B: not to be hoisted into VGPRs)
    const bool slab = ta.slab != 0;
    const size_t gstride = (size_t)LY::count(D);
    gfloat* ring = G + (slab ? (size_t)ta.n_copies : (size_t)1) * gstride;
    if constexpr (PERSIST) {
        // for the kernel that closes this chunk (nsf_adam_kernel with `fused_close`, nsf_kernels.hip): the clique's step / stop as this launch found them
        if (it == 0 && i == 0 && bx == 0 && tl_ == 0 && h_xrows <= SPAN_MAX_D) {
            unsigned* cw = (unsigned*)(ring + LOSS_RING * LOSS_SLOTS);
            cw[CLOSE_WORD_STEP] = (unsigned)st_step;
            cw[CLOSE_WORD_STOP] = (unsigned)st_stop;
        }
    }
    // fused Adam (nsf_cond_mfma.h): gradient copies and optimiser state alternate between two buffers with the parity
    // of the iteration inside its chunk; the second set sits behind the loss ring: [copies][ring][64][copies][theta|m|v]
    gfloat* const G0 = G;
    gfloat* Gset1 = ring + LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
    gfloat* alt = Gset1 + (size_t)ta.n_copies * gstride;
    // chunk-persistent form: two sets of TAGGED copies behind the second state buffer, 2 floats (value, tag) per parameter
    gfloat* const tg0 = alt + 3 * gstride;
    const size_t tg_set = (size_t)ta.n_copies * 2 * gstride;
    const int par = (ta.fused_adam != 0) ? (it & 1) : 0;
    const bool pending = ta.fused_adam != 0 && it > 0;
    const gfloat* Gprev = par ? G0 : Gset1;                   // copy 0 of the previous iteration
    if (par) G = Gset1;
    if (slab) G += (size_t)bx * gstride;                      // one gradient copy per block
    const int xrows = h_xrows;                                // rows of a particle tile in LDS (largest D of the launch)
    const float* pan = smem + PANEL_BASE;                     // the block's conditioner panel (nsf_cond_mfma.h)
    float* ones = smem + PANEL_BASE + CP::floats(xrows);      // (the row of ones, written above)
    float* tiles0 = ones + ONES_ROW;
    const int wave_floats = train1_wave_floats(xrows, H);
    // fixed-size rows first: their offsets from the wave's base are immediates of the LDS instructions (fewer address registers)
    float* stg = tiles0 + (size_t)w * wave_floats;            // [16][XS] staging rows
    float* hrow = stg + 16 * XS;                              // [H][XS] h1 of the tile (an operand of the last gradient GEMM)
    float* xt = hrow + H * XS;                                // [xrows][XS] particle tile, dimension-major
    float* ctacc = xt + xrows * XS;                           // D > 16: dW0 rows 16.. of the wave, summed over its tiles
    const bool merged = (i <= 16 - (H + 1));                  // the two last gradient GEMMs share one operand tile (see phase B)
    gfloat* Gb = G + LY::off(i > 0 ? i : 1);
    const int members = (((n + (TP << ts) - 1) >> (PSH + ts)) + W - 1) >> ws;     // blocks of this (clique, dim) group with a tile
    const bool has_tile = p0 < n && w < W;                    // (waves W .. : helper waves of a two-wave build, staging only)
    const unsigned stg_lane = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stg + lane);   // LDS byte address
    const int r16 = lane & 15, kq = lane >> 4;
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
    if (it == (PERSIST ? 0 : AF(iter_idx))) { float d0_ = 0.f, d1_ = 0.f; PSTAMP(0, d0_, d1_); }
#endif
    // tile loader: every lane reads the columns 0..i of its own particle row (16-byte loads at the row's 4-byte
    // alignment; the conditioner's inputs and x_i itself) and drops them into the dimension-major LDS tile: no index
    // arithmetic, no column the dim does not need.  Rows beyond n re-read row n-1 (masked out of loss and gradient).
    // Fetch (global -> registers) and store (registers -> LDS) are separate so that the first tile's loads are in
    // flight while the block stages its weight panel.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    auto fetch = [&](int pt, int c0, float (&xr)[16]) {
        const int pr = (pt + (lane & (TP - 1)) < n) ? pt + (lane & (TP - 1)) : n - 1;
        const gfloat* row = x + (size_t)pr * D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = c0 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k <= i) {                                 // wave-uniform
                if (k + 3 < D) {
                    const f32x4u u = *(const __attribute__((address_space(1))) f32x4u*)(row + k);
                    v = f32x4{u.x, u.y, u.z, u.w};
                } else {
                    v.x = row[k];
                    if (k + 1 < D) v.y = row[k + 1];
                    if (k + 2 < D) v.z = row[k + 2];
                }
            }
            xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
        }
    };
    auto store = [&](int c0, const float (&xr)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (c0 + u <= i) xt[(c0 + u) * XS + (lane & (TP - 1))] = xr[u];
    };
    auto load_tile = [&](int pt, int c_first) {
        for (int c0 = c_first; c0 <= i; c0 += 16) {
            float xr[16];
            fetch(pt, c0, xr);
            store(c0, xr);
        }
    };
    {
        // the one workgroup barrier in front of the tile loop: the block's waves share the (clique, dim) and so the panel
        float xr[16];
        const bool first = !PERSIST || it == 0;               // the particle tile stays in LDS between the iterations of a chunk
        if (!PERSIST && p0 < n) fetch(p0, 0, xr);               // (PERSIST: once per chunk, after the staging: fewer live registers)
        {
            FusedAdam fa;
            fa.grads = pending ? Gprev : nullptr;
            fa.gstride = gstride;
            fa.copies = members;
            // state before the pending update: the buffer of the previous iteration's parity (even: the clique's own)
            const gfloat* own_t = (const gfloat*)kparams;
            const bool src_alt = pending && par == 0;
            const gfloat* t_src = src_alt ? alt : own_t;
            fa.m_src = src_alt ? alt + gstride : own_m;
            fa.v_src = src_alt ? alt + 2 * gstride : own_v;
            const bool writer = pending && bx == 0;
            fa.t_dst = writer ? (src_alt ? (gfloat*)own_t : alt) : nullptr;
            fa.m_dst = writer ? (src_alt ? (gfloat*)own_m : alt + gstride) : nullptr;
            fa.v_dst = writer ? (src_alt ? (gfloat*)own_v : alt + 2 * gstride) : nullptr;
            if constexpr (PERSIST) {
                // flag-in-data exchange (nsf_cond_mfma.h: stage_cond_panel_persist): the previous iteration's copies arrive as
                // (value, tag) pairs, theta | m | v of the dim stay in this block's LDS between the iterations
                PersistAdam pa_;
                pa_.tagged = pending ? tg0 + (size_t)(par ^ 1) * tg_set : nullptr;
                pa_.cstride = 2 * gstride;
                pa_.copies = members;
                pa_.tag = (uint32_t)(st_step + it);            // = the writer's st_step + (it - 1) + 1
                pa_.m_src = own_m; pa_.v_src = own_v;          // (first iteration of a chunk: the state is in the clique's own arrays)
                pa_.t_dst = fa.t_dst; pa_.m_dst = fa.m_dst; pa_.v_dst = fa.v_dst;
                pa_.keep = tiles0 + (size_t)W * wave_floats;
                pa_.kstride = persist_keep_stride<K, H>(xrows);
                pa_.ctr = (unsigned*)(ring + LOSS_RING * LOSS_SLOTS) + i;
                pa_.spin_log2 = ta.persist_spins;
                pa_.max_iters = ta.max_iters;
                pa_.lr = ta.lr; pa_.beta1 = ta.b1; pa_.beta2 = ta.b2; pa_.eps = ta.eps;
                pa_.log_b1 = ta.lb1; pa_.log_b2 = ta.lb2;
                if (it == 1 && threadIdx.x == 0) {             // (diagnostic: the XCCs a group's blocks run on, bits 23-30 of the dim's control word)
                    unsigned xcc;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
                    __hip_atomic_fetch_or(pa_.ctr, 1u << (23 + (xcc & 7u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                pa_.looks = 0u;
                { float d0_ = lossv, d1_ = r0; PSTAMP(10, d0_, d1_); }       // (loop top -> here: descriptor, pointers)
                int rc_;
                if (ta.persist_split != 0) {
                    // every block records its slice of the state (fa.*_dst are the first block's otherwise)
                    const bool to_own = pending && par == 0;
                    pa_.t_dst = to_own ? (gfloat*)own_t : alt;
                    pa_.m_dst = to_own ? (gfloat*)own_m : alt + gstride;
                    pa_.v_dst = to_own ? (gfloat*)own_v : alt + 2 * gstride;
                    if constexpr (WIDE)
                        rc_ = stage_cond_panel_persist_split_wide<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it,
                                                                        bx, tg0 + 2 * tg_set);
                    else
                        rc_ = stage_cond_panel_persist_split<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it,
                                                                   bx, tg0 + 2 * tg_set);
                } else {
                    if constexpr (WIDE) {
                        if ((ROOMY || H == 16) && (int)blockDim.x > (64 << ws))    // (helper waves: one parameter per thread)
                            rc_ = stage_cond_panel_persist_solo_wide<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it);
                        else
                            rc_ = stage_cond_panel_persist_wide<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it);
                    }
                    else if ((ROOMY || H == 16) && (int)blockDim.x > (64 << ws))     // (helper waves: one parameter per thread)
                        rc_ = stage_cond_panel_persist_solo<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it);
                    else
                        rc_ = stage_cond_panel_persist<K, H>(smem, (const float*)own_t, pa_, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, n, it);
                }
                if (rc_ == 1) return;
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
                { float d0_ = smem[PANEL_BASE], d1_ = (float)rc_; PSTAMP(11, d0_, d1_); }   // the staging: loads, looks, Adam, LDS stores
                if (lane == 0 && w < 4) atomicAdd((unsigned*)&smem[PANEL_BASE - 64 + w * 16 + 13], pa_.looks);
#endif
                // A thread that gave up (2^persist_spins looks at copies that never came: a member of the group is not on the
                // machine, somebody else holds its place) has raised the group's abort flag; the block leaves as one, the
                // other members see the flag at their next look and leave too, nsf_bookkeep_kernel turns it into the STALL status.
                if (rc_ == 2) *(volatile int*)&smem[1] = 1;
                __syncthreads();
                if (*(volatile int*)&smem[1] != 0) return;
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
                { float d0_ = smem[PANEL_BASE], d1_ = smem[PANEL_BASE + 1]; PSTAMP(12, d0_, d1_); }   // the block's barrier behind the staging
#endif
            } else {
                if (!stage_cond_panel<K, H, false>(smem, (const float*)t_src, fa, h_panel_map, i, threadIdx.x, blockDim.x, st_step, st_stop, a, n, it)) return;
                __syncthreads();
            }
        }
        if (!PERSIST && p0 >= n) return;
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(11);
#endif
        if (first && has_tile) {
            if (PERSIST) fetch(p0, 0, xr);
            store(0, xr);
            load_tile(p0, 16);
        }
    }

    if constexpr (PERSIST) {                                  // (zeroed HERE, not at the loop's end: nothing of them is live across the staging)
        float z_ = 0.0f;
        asm volatile("" : "+v"(z_));                          // (a zero made in the loop: the compiler keeps hoisted zero vectors in registers otherwise)
        lossv = z_; r0 = z_;
        c1 = f32x4{z_, z_, z_, z_}; c0 = c1; cb1 = c1;
#pragma unroll
        for (int t = 0; t < NT; ++t) { cacc[t] = c1; cb2[t] = c1; }
    }
    // Issue priority by progress ("least slack first", PERSIST): a wave that is near the end of its iteration holds up the seven
    // other blocks of its (clique, dim) group -- they cannot stage the next iteration before its copy is out -- while a wave
    // that has just begun its unit has the whole unit of slack.  With three waves per SIMD taking turns at the issue port
    // regardless, the group's dependent chain ran at the pace of its unluckiest wave; with the priority raised phase by
    // phase (conditioner forward 0 -> spline and conditioner backward 1 -> gradient GEMMs 2 -> epilogue and the next
    // staging 3, the slice owners of the divided update included) C3 went from 13.2 to 11.2 us per iteration (scripts/ab.py,
    // four interleaved rounds on one box); a lone wave per SIMD (one Plaza clique) is indifferent.
    if constexpr (PERSIST) __builtin_amdgcn_s_setprio(0);
    for (int tt = 0; tt < T && has_tile; ++tt) {
        const int pt = p0 + tt * TP;
        if (pt >= n) break;
        PSTAMP(1, lossv, r0);
        if (tt > 0) load_tile(pt, 0);
        wave_lds_sync();
        // (PERSIST: a fresh derivation per phase -- a value derived once at the loop's top would have to stay in a register,
        //  or in scratch, from there to its last use in the epilogue)
        int tl2_ = threadIdx.x;
        if constexpr (PERSIST && !ROOMY) asm volatile("" : "+v"(tl2_));
        const int lane = tl2_ & 63;
        const unsigned stg_lane = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stg + lane);
        const int r16 = lane & 15, kq = lane >> 4;
        PSTAMP(2, lossv, r0);
        if constexpr (SPL) {
            // ---- two lanes per particle (nsf_half.h): lanes p and p + 32 share particle p of the wave's 32 ----
            constexpr int HP = LY::HP, HQ = H / 2;
            const int p = lane & 31;
            const bool up = lane >= 32;
            const bool validp = pt + p < n;
            const unsigned stg_p = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stg + p);
            float th[HP], gth[HP], h1o[HQ], h2o[HQ];
            if (i == 0) {
#pragma unroll
                for (int o = 0; o < HP; o += 4) {
                    const cm_f32x4 v4 = *(const cm_f32x4*)(pan + (up ? HP : 0) + o);
                    th[o] = v4[0]; th[o + 1] = v4[1]; th[o + 2] = v4[2]; th[o + 3] = v4[3];
                }
            } else {
                float h1[H], h2[H];
                cond_forward_half<K, H>(pan, i, CP::s0_of(i), xt, XS, lane, p, h1o, h2o, h1, h2, th);
                // operands of the gradient GEMMs, parked while the lanes are busy with the spline: the low lanes write the
                // h2 rows (0 ..), the high lanes the h1 rows (16 .. = hrow), both at the particle's column
                float hs[H];
#pragma unroll
                for (int t = 0; t < H; ++t) hs[t] = up ? h1[t] : h2[t];
                lds_rows_store<0, H, 0, H>(stg_p + (up ? 16u * XS * 4u : 0u), hs);
            }
            PSTAMP(3, th[0], th[HP - 1]);
            if constexpr (PERSIST) __builtin_amdgcn_s_setprio(1);
            SplineH<K> S;
            float z, lad;
            spline_half_fwd<K, HP>(xt[i * XS + p], th, up, B, S, z, lad);
            PSTAMP(4, z, lad);
            if (validp && !up) lossv += 0.5f * z * z - lad;       // (both lanes hold z and lad: the low one counts)
            spline_half_bwd<K, HP>(S, up, B, validp ? z : 0.0f, validp ? -1.0f : 0.0f, gth);
            PSTAMP(5, gth[0], gth[K]);
            if (i == 0) {   // init_param: plain sum over the half's particles of gth
                float v[HP];
#pragma unroll
                for (int t = 0; t < HP; ++t) v[t] = gth[t];
                r0 += butterfly_half<HP>(v, p);
                wave_lds_sync();
                continue;
            }
            float ga2o[HQ], ga1o[HQ];
            cond_backward_half<K, H>(pan, lane, gth, h1o, h2o, ga2o, ga1o);
            PSTAMP(6, ga1o[0], ga2o[HQ - 1]);
            if constexpr (PERSIST) __builtin_amdgcn_s_setprio(2);
            // ---- weight gradients on the matrix cores: the contraction runs over the wave's 32 particles (NS = 8 steps);
            //      BOTH 16-row tiles of gth are staged at once, the halves side by side (columns 0-31 | 32-63) ----
            const float* pa = stg + r16 * XS + kq;
            float breg[NS], ar0[NS], ar1[NS];
            {
                wave_lds_sync();
                const float* pah = ((r16 < H) ? stg + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) breg[s4] = pah[4 * s4];
                wave_lds_sync();
            }
            PSTAMP(7, breg[0], breg[NS - 1]);
            lds_rows_store<0, 16, 0, HP>(stg_lane, gth);
            wave_lds_sync();
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) { ar0[s4] = pa[4 * s4]; ar1[s4] = pa[32 + 4 * s4]; }
            wave_lds_sync();
            {   // [ga2 | ga1]: rows 0 .. H-1 | H .. 2H-1, a half writes its own units' rows at the particle's column
                const unsigned own = stg_p + (up ? (unsigned)HQ * XS * 4u : 0u);
                lds_rows_store<0, HQ, 0, HQ>(own, ga2o);
                lds_rows_store<H, HQ, 0, HQ>(own, ga1o);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) {
                cacc[0] = mfma4(ar0[s4], breg[s4], cacc[0]);
                cacc[1] = mfma4(ar1[s4], breg[s4], cacc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            wave_lds_sync();
            if (merged) {
                const int kx = r16 - (H + 1);
                const bool one = (r16 == H) || kx >= i;
                const float* pbm = (one ? ones : ((r16 < H) ? hrow + r16 * XS : xt + kx * XS)) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    ar0[s4] = pa[4 * s4];
                    c1 = mfma4(ar0[s4], pbm[4 * s4], c1);
                }
            } else {
                const float* pb0 = ((r16 < i) ? xt + r16 * XS : ones) + kq;     // input columns 0..15 (column i = bias)
                const float* pb1 = ((r16 < H) ? hrow + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    ar0[s4] = pa[4 * s4];
                    c1 = mfma4(ar0[s4], pb1[4 * s4], c1);
                    c0 = mfma4(ar0[s4], pb0[4 * s4], c0);
                }
            }
            wave_lds_sync();
            PSTAMP(8, c1.x, cacc[0].x);
            continue;
        }
        const bool valid = pt + lane < n;
        float h1[H], h2[H], th[PoP], gth[PoP];
        if (i == 0) {
#pragma unroll
            for (int o = 0; o < PoP; o += 4) {
                const cm_f32x4 v4 = *(const cm_f32x4*)(pan + o);
                th[o] = v4[0]; th[o + 1] = v4[1]; th[o + 2] = v4[2]; th[o + 3] = v4[3];
            }
        } else {
            cond_forward_mfma<K, H>(pan, i, CP::s0_of(i), xt, XS, lane, lane, h1, h2, th);
            // operands of the gradient GEMMs, parked while the lanes are busy with the spline
            lds_rows_store<0, H, 0, H>(stg_lane, h2);
            lds_rows_store<16, H, 0, H>(stg_lane, h1);         // = hrow
        }
        PSTAMP(3, th[0], th[PoP - 1]);
        if constexpr (PERSIST) __builtin_amdgcn_s_setprio(1);     // (behind the conditioner forward: a third of the unit done)
        SplineT<K> S;
        float z, lad;
        spline_train_fwd<K, PoP>(xt[i * XS + lane], th, B, S, z, lad);
        PSTAMP(4, z, lad);
        if (valid) lossv += 0.5f * z * z - lad;
        spline_train_bwd<K, PoP>(S, B, valid ? z : 0.0f, valid ? -1.0f : 0.0f, gth);
        PSTAMP(5, gth[0], gth[LY::HP]);
        if (i == 0) {   // init_param: plain sum over particles of gth
            constexpr int N0 = (PoP <= 32) ? 32 : 64;
            float v[N0];
#pragma unroll
            for (int t = 0; t < N0; ++t) v[t] = (t < PoP) ? gth[t] : 0.0f;
            r0 += butterfly<N0>(v, lane);
            wave_lds_sync();                                    // the tile is overwritten by the next iteration's loads
            continue;
        }
        // ---- per-particle back-propagation through the conditioner (4x4x1 MFMA chains, nsf_cond_mfma.h) ----
        float ga2[H], ga1[H];
        cond_backward_mfma<K, H>(pan, lane, gth, h1, h2, ga2, ga1);
        PSTAMP(6, ga1[0], ga2[H - 1]);
        if constexpr (PERSIST) __builtin_amdgcn_s_setprio(2);     // (the unit's last phase: the gradient GEMMs)
        // ---- weight gradients on the matrix cores (see nsf_train_kernel); operand rows: lane&15 = feature,
        //      lane>>4 = particle inside the k-group of 4; the bias column (and the unused columns) multiply 1 ----
        // The 16 staging rows carry four generations of operands: [h2 | h1], gth tile 0, gth tile 1, [ga2 | ga1].  A
        // generation is read into registers completely, then the NEXT one is written before this one's MFMAs are
        // issued: the LDS writes complete under the 16 x 32 MFMA cycles and only the reads' round trip stays exposed.
        const float* pa = stg + r16 * XS + kq;
        float breg[NS], areg[NS];
        {
            wave_lds_sync();
            const float* pah = ((r16 < H) ? stg + r16 * XS : ones) + kq;
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) breg[s4] = pah[4 * s4];
            wave_lds_sync();
        }
        PSTAMP(7, breg[0], breg[NS - 1]);
        {   // phase A: dW2t | db2 = [h2, 1]^T (x) gth ;  phase B: dW1t | db1 = [h1,1]^T (x) ga2 ;  dW0t | db0 = [x,1]^T (x) ga1
            lds_rows_store<0, 16, 0, PoP>(stg_lane, gth);
            wave_lds_sync();
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
            wave_lds_sync();
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                if (t == 1 && NT > 1) lds_rows_store<0, 16, 16, PoP>(stg_lane, gth);
                if (t == 2 && NT > 2) lds_rows_store<0, 16, 32, PoP>(stg_lane, gth);
                if (t == 3 && NT > 3) lds_rows_store<0, 16, 48, PoP>(stg_lane, gth);
                if (t == NT) {
                    if constexpr (WIDE_H) {
                        lds_rows_store<0, 16, 0, H>(stg_lane, ga2);
                    } else {
                        lds_rows_store<0, H, 0, H>(stg_lane, ga2);
                        lds_rows_store<H, H, 0, H>(stg_lane, ga1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    cacc[t - 1] = mfma4(areg[s4], breg[s4], cacc[t - 1]);
                    if constexpr (WIDE_H) cb2[t - 1] = mfma4(areg[s4], 1.0f, cb2[t - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_sync();
                if (t < NT) {
#pragma unroll
                    for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
                    wave_lds_sync();
                }
            }
            if constexpr (WIDE_H) {
                // ga2 (16 rows) x [h1]: dW1t, and x 1: db1; then ga1 (16 rows) x [x_0 .. x_{i-1} | 1]: dW0t | db0
                const float* pb1 = hrow + r16 * XS + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
                wave_lds_sync();
                lds_rows_store<0, 16, 0, H>(stg_lane, ga1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    c1 = mfma4(areg[s4], pb1[4 * s4], c1);
                    cb1 = mfma4(areg[s4], 1.0f, cb1);
                }
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_sync();
                const float* pb0 = ((r16 < i) ? xt + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c0 = mfma4(areg[s4], pb0[4 * s4], c0);
                }
            } else if (merged) {
                // i <= 16 - (H + 1): the columns [h1 (H) | 1 | x_0 .. x_{i-1}] of BOTH products fit one 16-column operand
                // (they share the bias column): one MFMA chain instead of two.  Rows 0..H-1 (ga2) x columns 0..H give
                // dW1t | db1, rows H.. (ga1) x columns H.. give db0 | dW0t; the cross terms are not used.
                const int kx = r16 - (H + 1);
                const bool one = (r16 == H) || kx >= i;
                const float* pbm = (one ? ones : ((r16 < H) ? hrow + r16 * XS : xt + kx * XS)) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c1 = mfma4(areg[s4], pbm[4 * s4], c1);
                }
            } else {
                const float* pb0 = ((r16 < i) ? xt + r16 * XS : ones) + kq;     // input columns 0..15 (column i = bias)
                const float* pb1 = ((r16 < H) ? hrow + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    areg[s4] = pa[4 * s4];
                    c1 = mfma4(areg[s4], pb1[4 * s4], c1);
                    c0 = mfma4(areg[s4], pb0[4 * s4], c0);
                }
            }
            for (int ct = 1; ct * 16 <= i; ++ct) {              // D > 16: further column tiles add into the wave's own copy
                const int cab = ct * 16 + r16;
                const float* pb0 = ((cab < i) ? xt + cab * XS : ones) + kq;
                f32x4 cx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) cx = mfma4(areg[s4], pb0[4 * s4], cx);
                if ((WIDE_H || (kq >= QH && kq < 2 * QH)) && cab <= i) {
                    float* dst = &ctacc[(cab - 16) * H + 4 * (WIDE_H ? kq : kq - QH)];
                    if (slab) {
                        if (tt > 0) { cx.x += dst[0]; cx.y += dst[1]; cx.z += dst[2]; cx.w += dst[3]; }
                        dst[0] = cx.x; dst[1] = cx.y; dst[2] = cx.z; dst[3] = cx.w;
                    } else {
                        gsink4(&Gb[cab * H + 4 * (WIDE_H ? kq : kq - QH)], cx, false);
                    }
                }
            }
            wave_lds_sync();
        }
        PSTAMP(8, c1.x, cacc[0].x);
    }

    // ---- the gradient of this dim's parameter block ----
    if constexpr (PERSIST) __builtin_amdgcn_s_setprio(3);     // (from here to the end of the next staging: the group's exchange)
    int tl3_ = threadIdx.x;
    cclique* cpe = cp0;
    cargs* ape_ = (cargs*)((cchar*)__builtin_amdgcn_kernarg_segment_ptr() + 40);
    if constexpr (PERSIST && !ROOMY) asm volatile("" : "+v"(tl3_), "+s"(cpe), "+s"(ape_));
    const int lane_e = tl3_ & 63, r16_e = lane_e & 15, kq_e = lane_e >> 4;
  {
    // (PERSIST: what the epilogue needs of the clique's descriptor and of the launch's arguments is derived AGAIN here, from
    //  laundered pointers, under the same names: the versions of the loop's top die in front of the tile loop instead of
    //  sitting in scalar registers -- i.e. in spill lanes -- across it)
#undef AF
#define AF(f) (PERSIST ? ape_->f : a.f)
    gfloat* G = (gfloat*)cpe->kgrad;
    gstate* st = (gstate*)cpe->state;
    const int n = cpe->n;
    const bool slab = AF(slab) != 0;
    const size_t gstride = (size_t)LY::count((int)cpe->D);
    gfloat* ring = G + (slab ? (size_t)AF(n_copies) : (size_t)1) * gstride;
    gfloat* Gset1 = ring + LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS;
    gfloat* const tg0 = Gset1 + (size_t)AF(n_copies) * gstride + 3 * gstride;
    const size_t tg_set = (size_t)AF(n_copies) * 2 * gstride;
    const int par = (AF(fused_adam) != 0) ? (it & 1) : 0;
    if (par) G = Gset1;
    if (slab) G += (size_t)bx * gstride;
    gfloat* Gb = G + LY::off(i > 0 ? i : 1);
    const int members = (((n + (TP << ts) - 1) >> (PSH + ts)) + W - 1) >> ws;
    const bool has_tile = (((bx << ws) + w) << (PSH + ts)) < n && w < W;
    if (slab) {
        // One copy per BLOCK: every wave lays its fragment out in parameter order in its own (now free) rows, 16 bytes per
        // store (the accumulators hold four consecutive parameters), the block's threads add the fragments in wave order
        // and write the copy with consecutive 16-byte stores.
        float* frag = stg;                                     // 24 rows: room for every dim's block
        if (!has_tile) {
            // (PERSIST: a wave without particles stays for the staging of the following iterations)
        } else if (i == 0) {
            if constexpr (SPL) {                               // (lane (half, p < 16) holds the half's column p)
                if ((lane_e & 31) < LY::HP) frag[LY::HP * (lane_e >> 5) + (lane_e & 31)] = r0;
            } else {
                if (lane_e < PoP) frag[lane_e] = r0;
            }
        } else {
            float* fw = frag + LY::oW2(i);
            if (r16_e <= H) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq_e + 3 < PoP) *(f32x4*)&fw[r16_e * PoP + 16 * t + 4 * kq_e] = cacc[t];
            }
            if (kq_e < QH && r16_e <= H) *(f32x4*)&(frag + LY::oW1(i))[r16_e * H + 4 * kq_e] = c1;
            if constexpr (WIDE_H) {                           // the bias rows (every column of a bias chain holds the same sums) + dW0t | db0
                if (r16_e == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (16 * t + 4 * kq_e + 3 < PoP) *(f32x4*)&fw[H * PoP + 16 * t + 4 * kq_e] = cb2[t];
                    *(f32x4*)&(frag + LY::oW1(i))[H * H + 4 * kq_e] = cb1;
                }
                if (r16_e <= i) *(f32x4*)&frag[r16_e * H + 4 * kq_e] = c0;
            } else if (merged) {                                     // columns H.. of the shared tile: bias first, then x_0..x_{i-1}
                const int k0 = (r16_e == H) ? i : r16_e - (H + 1);
                if (kq_e >= QH && kq_e < 2 * QH && (r16_e == H || (r16_e > H && k0 < i))) *(f32x4*)&frag[k0 * H + 4 * (kq_e - QH)] = c1;
            } else if (kq_e >= QH && kq_e < 2 * QH && r16_e <= i) {
                *(f32x4*)&frag[r16_e * H + 4 * (kq_e - QH)] = c0;
            }
            for (int e = lane_e; e < (i - 15) * H; e += 64) frag[16 * H + e] = ctacc[e];     // D > 16 (rows 16..i)
        }
        // the block's loss: the waves' sums added in wave order by ONE thread (below), then one atomic per block into a ring
        // slot shared by at most two blocks of the clique while D x blocks <= 256 (LOSS_SLOTS = 128) -- a sum of two floats does not depend on
        // their order, so the loss record is the same whichever way the launches and the waves happen to be timed
        {
            const float wtot = wave_sum(lossv);
            if (lane_e == 0 && w < W) xt[64] = has_tile ? wtot : 0.0f;   // a padding word of the tile's first row (helper waves have no rows)
        }
        __syncthreads();                                      // waves without a tile left before the panel barrier
        const int waves_c = (n + (TP << ts) - 1) >> (PSH + ts);
        const int alive = (waves_c - (bx << ws) < W) ? waves_c - (bx << ws) : W;
        const int nj4 = ((i == 0) ? PoP : LY::block(i)) >> 2;
        gvf4_t* Gc = (gvf4_t*)(G + ((i == 0) ? 0 : LY::off(i)));
        if (threadIdx.x == 64 * (alive - 1)) {                 // (the block's LAST wave with a tile: it has the fewest fragments to add below)
            float bl = 0.0f;
            for (int ww = 0; ww < alive; ++ww) bl += tiles0[(size_t)ww * wave_floats + (16 + H) * XS + 64];
            gfloat* dst = (st != nullptr) ? &ring[((st_step + it) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                    (((i * members + bx) >> 1) & (LOSS_SLOTS - 1))]
                                          : (gfloat*)AF(loss_sum);
            if (dst != nullptr) gsink(dst, bl, false);
        }
        for (int e = threadIdx.x; e < nj4 && w < alive; e += 64 * alive) {
            f32x4 sum = *(const f32x4*)&tiles0[4 * e];
            for (int ww = 1; ww < alive; ++ww) {
                const f32x4 o = *(const f32x4*)&tiles0[(size_t)ww * wave_floats + 4 * e];
                sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
            }
            // PERSIST, not the chunk's last iteration: the group's other blocks read this copy in the next iteration of the
            // SAME launch -- every word goes out with its tag (the iteration's number in the run) as an 8-byte pair, by
            // agent-scope (sc1) write-through stores that nobody waits for: the readers poll the tags themselves.
            // The chunk's LAST copy is read by the next kernel (nsf_adam_kernel, close_chunk): the plain layout.
            // (window-spanning launch, ROOMY builds: an iteration that ends a window may be the run's last -- the rule is evaluated
            //  behind it -- so it leaves BOTH layouts; an iteration that exhausts the budget leaves the plain one only)
            bool more_ = PERSIST && it + 1 < AF(persist_iters), plain_ = !more_;
            if constexpr (ROOMY) {
                const int sw_ = AF(span_window);
                if (sw_ > 0) {
                    const int done_ = st_step + it + 1;
                    if (done_ >= AF(max_iters)) more_ = false;
                    plain_ = !more_ || (done_ % sw_) == 0;
                }
            }
            if (more_) {
                const float tagf = __uint_as_float((uint32_t)(st_step + it + 1));
                gfloat* dst = tg0 + (size_t)par * tg_set + (size_t)bx * 2 * gstride + 2 * (size_t)(((i == 0) ? 0 : LY::off(i)) + 4 * e);
                const f32x4 lo = {sum.x, tagf, sum.y, tagf}, hi = {sum.z, tagf, sum.w, tagf};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" ::"v"(dst), "v"(lo), "v"(hi) : "memory");
            }
            if (plain_) Gc[e] = sum;
        }
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(12);
#endif
    } else if (i == 0) {
        if constexpr (SPL) {
            if ((lane_e & 31) < LY::HP) gsink(&G[LY::HP * (lane_e >> 5) + (lane_e & 31)], r0, false);
        } else {
            if (lane_e < PoP) gsink(&G[lane_e], r0, false);
        }
    } else {
        gfloat* Gw = Gb + LY::oW2(i);
        {
            // atomics: (H+1) x PoP floats through LDS in two halves of the staging tile, flat order (consecutive addresses)
            constexpr int TOT = (WIDE_H ? H : H + 1) * PoP;     // (H = 16: the bias row comes from its own chain, below)
            static_assert(TOT <= 16 * XS, "the transposed dW2 block fits the staging rows");
            if (r16_e <= H) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq_e + 3 < PoP) {
                        float* d = &stg[r16_e * PoP + 16 * t + 4 * kq_e];
                        d[0] = cacc[t].x; d[1] = cacc[t].y; d[2] = cacc[t].z; d[3] = cacc[t].w;
                    }
            }
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < (TOT + 63) / 64; ++c) {
                const int f = c * 64 + lane_e;
                if (f < TOT) gsink(&Gw[f], stg[f], false);
            }
        }
        if (kq_e < QH && r16_e <= H) gsink4(&(Gb + LY::oW1(i))[r16_e * H + 4 * kq_e], c1, false);
        if constexpr (WIDE_H) {
            if (r16_e == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (16 * t + 4 * kq_e + 3 < PoP) gsink4(&Gw[H * PoP + 16 * t + 4 * kq_e], cb2[t], false);
                gsink4(&(Gb + LY::oW1(i))[H * H + 4 * kq_e], cb1, false);
            }
            if (r16_e <= i) gsink4(&Gb[r16_e * H + 4 * kq_e], c0, false);
        } else if (merged) {
            const int k0 = (r16_e == H) ? i : r16_e - (H + 1);
            if (kq_e >= QH && kq_e < 2 * QH && (r16_e == H || (r16_e > H && k0 < i))) gsink4(&Gb[k0 * H + 4 * (kq_e - QH)], c1, false);
        } else if (kq_e >= QH && kq_e < 2 * QH && r16_e <= i) {
            gsink4(&Gb[r16_e * H + 4 * (kq_e - QH)], c0, false);
        }
    }
    PSTAMP(9, lossv, r0);
    const float tot = slab ? 0.0f : wave_sum(lossv);
    if (lane_e == 0 && has_tile && !slab) {
        gfloat* dst = (st != nullptr) ? &ring[((st_step + it) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                ((((bx << ws) + w) * 7 + i * 13) & (LOSS_SLOTS - 1))]
                                      : (gfloat*)AF(loss_sum);
        if (dst != nullptr) gsink(dst, tot, false);
    }
    // ---- window-spanning launch (round 6; two-wave builds): the clique's blocks close the window THEMSELVES ---------------------
    // One launch runs the fit's windows back to back: a 50-iteration chunk of a real fit is 339 us of launch + ~36 us of fixed cost
    // (the launch's cold first iteration and drain, the gap between two graph launches, the closing Adam and bookkeeping kernels).
    // At a window's end every wave waits for its stores and the loss atomic to be acknowledged, the block passes a barrier, ONE lane
    // takes a ticket at a counter of the clique's workspace; the block whose ticket is the window's last runs the bookkeeping
    // (nsf_bookkeep.h: the code of nsf_bookkeep_kernel -- loss record, stop rule, step, the host's mirror) behind an agent-scope
    // acquire, and publishes the window's number with the stop bit; everybody polls that word and either goes on to the next
    // window or leaves.  (MI355X_MICROARCH.md, Valid forms: sc1 / atomic payload, vmcnt(0) in every storing wave, workgroup
    // barrier, one lane's counter add; the last arriver, told by the value its add returned, reads behind an acquire.)
    if constexpr (ROOMY) {
        const int sw_ = AF(span_window);
        if (sw_ > 0) {
            const int done_ = st_step + it + 1;                           // iterations of the run this block has finished
            const bool budget_end = done_ >= AF(max_iters) || it + 1 >= AF(persist_iters);
            if ((done_ % sw_) == 0 || budget_end) {
                unsigned* words = (unsigned*)(ring + LOSS_RING * LOSS_SLOTS);
                const unsigned win_no = (unsigned)((it + sw_) / sw_);     // windows of this launch closed so far, this one included
                const unsigned total = (unsigned)((int)cpe->D * members);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) {
                    const unsigned old = __hip_atomic_fetch_add(&words[SPAN_WORD_TICKET], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *(volatile int*)&smem[2] = (old + 1u == total * win_no) ? 1 : 0;
                }
                __syncthreads();
                if (*(volatile int*)&smem[2] != 0) {                      // (block-uniform) the clique's last block of this window
                    if (threadIdx.x == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __syncthreads();
                    BookArgs bk;
                    bk.ring = (float*)ring; bk.iter_loss = cpe->iter_loss; bk.st = (nfisam_train_state*)cpe->state;
                    bk.mirror = nullptr;                                   // (published below, BEHIND the decision: off the clique's critical path)
                    bk.n = n; bk.D = (int)cpe->D; bk.chunk = sw_; bk.zero_counters = 0;
                    bk.cfg.lr = AF(adam.lr); bk.cfg.beta1 = AF(adam.beta1); bk.cfg.beta2 = AF(adam.beta2); bk.cfg.eps = AF(adam.eps);
                    bk.cfg.max_iters = AF(adam.max_iters); bk.cfg.average_window = AF(adam.average_window);
                    bk.cfg.loss_delta_tol = AF(adam.loss_delta_tol); bk.cfg.reserved = 0;
                    bookkeep_window(&bk);
                    nfisam_train_state* sp = (nfisam_train_state*)cpe->state;
                    int stop_now = 0;
                    if (threadIdx.x == 0) {
                        stop_now = sp->stop;                              // (this thread wrote it in the body)
                        // The decision goes out FIRST: the other blocks need nothing else of what this block wrote (the loss
                        // record, the state and the zeroed ring rows are next read by the block that closes the NEXT window, which
                        // takes its ticket behind this block's next one, i.e. behind the release fence below).
                        __hip_atomic_store(&words[SPAN_WORD_DECISION], (win_no << 1) | (stop_now != 0 ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        words[SPAN_WORD_LAST_T] = (unsigned)done_;        // (for the closing Adam kernel: the update it applies is number done_,
                        words[SPAN_WORD_LAST_PARITY] = (unsigned)(it & 1);   //  its copies and source state sit in the buffers of this parity)
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores of the body are acknowledged
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        nfisam_train_state* mir = AF(span_mirror);
                        if (mir != nullptr) {                             // the host's mirror, as nsf_bookkeep.h publishes it
                            nfisam_train_state* m = mir + by;
                            const int seq = __hip_atomic_load(&m->reserved[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
                            m->step = sp->step; m->stop = stop_now; m->have_avg = sp->have_avg; m->loss_avg = sp->loss_avg; m->domain_err = sp->domain_err;
                            m->reserved[1] = sp->reserved[1];
                            m->reserved[2] = sp->reserved[2];
                            __hip_atomic_store(&m->reserved[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
                if (threadIdx.x == 0) {
                    unsigned spins = 0, dec = 0u;
                    for (;;) {
                        dec = __hip_atomic_load(&words[SPAN_WORD_DECISION], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((dec >> 1) == win_no) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << (AF(persist_spins) + 5))) {   // a member of the clique never arrived: leave, loudly (STALL)
                            __hip_atomic_fetch_or(&words[i], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            dec = 1u;
                            break;
                        }
                    }
                    *(volatile int*)&smem[3] = (int)(dec & 1u);
                }
                __syncthreads();
                if (*(volatile int*)&smem[3] != 0 || budget_end) break;
            }
        }
    }
  }
    if constexpr (!PERSIST) {
        break;
    } else {
        if (++it >= AF(persist_iters)) break;                 // (no barrier: the next staging waits for the copies' tags)
    }
  }
#if defined(NSF_STAMPS) && NSF_STAMPS == 3 && NSF_UNIT == 0
    if (STAMP_SEL && lane0 < 16 && w < 4) g_stamps[STAMP_SLOT * 32 + 16 + lane0] = (unsigned long long)((unsigned*)smem)[PANEL_BASE - 64 + w * 16 + lane0];
#endif
}
#undef AF

#undef STAMP_SEL
#undef STAMP_SLOT
#define STAMP_SEL (blockIdx.x == 0 && blockIdx.y == 0)
#define STAMP_SLOT ((w + blockIdx.z * (blockDim.x >> 6)) & 63)

// =============================================================================================
// training / VJP kernel, two lanes per particle (nsf_split.h): a wave covers 32 particles, so the same
// clique spreads over twice as many waves / CUs and every unit issues about half the instructions.
// Same arguments, LDS plan and gradient sinks as nsf_train_kernel, with TILE2 / XS2 in place of TILE / XS.
// =============================================================================================
template <int K, int H>
__device__ __forceinline__ void load_theta2(const float* lp, int i, const float* xin, int p, int hf,
                                            float (&h1m)[H / 2], float (&h1o)[H / 2], float (&h2m)[H / 2],
                                            float (&h2o)[H / 2], float (&th)[hp_of(K)]) {
    using LY = Layout<K, H>;
    if (i == 0) {
        load_row_used<LY::HP, K + LY::ND0>(lp + LY::HP * hf, th);
    } else {
        const float* blk = lp + LY::off(i);
        cond_hidden2<K, H>(blk, i, xin, p, hf, h1m, h1o, h2m, h2o);
        cond_theta2<K, H>(blk, i, hf, h2m, h2o, th);
    }
}

template <int K, int H, bool WL>
__global__ void __launch_bounds__(512) nsf_train2_kernel(TrainArgs a) {
    using LY = Layout<K, H>;
    constexpr int PoP = LY::PoP, HP = LY::HP, HH = H / 2;
    constexpr int NT = (PoP + 15) / 16;                       // 16-row output tiles of gth
    constexpr int UO = K + LY::ND0;                           // used outputs per half (the rest is padding)
    static_assert(H == 8, "the gradient GEMMs pack ga2|ga1 into one 16-row operand tile: H = 8");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool batched = a.cliques != nullptr;
    const nfisam_clique* cp = batched ? (a.cliques + blockIdx.y) : nullptr;
    const float* x = batched ? cp->x : a.single.x;
    float* kparams = batched ? cp->kparams : a.single.kparams;
    float* G = batched ? cp->kgrad : a.single.kgrad;
    nfisam_train_state* st = batched ? cp->state : a.single.state;
    const int n = batched ? cp->n : a.single.n;
    const int D = batched ? cp->D : a.single.D;
    const int L = a.L;
    const float B = a.B;
    const bool slab = a.slab != 0;
    const size_t gstride = (size_t)L * (size_t)(a.layer_stride > 0 ? a.layer_stride : LY::count(D));
    float* ring = G + (slab ? (size_t)gridDim.x : (size_t)1) * gstride;   // loss ring behind the gradient copies
    if (slab) G += (size_t)blockIdx.x * gstride;

    const int p0 = blockIdx.x * TILE2;
    if (p0 >= n) return;
    int st_stop = 0, st_step = 0;
    if (st != nullptr) {
        st_stop = __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_step = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int lane = threadIdx.x & 63;
    const int hf = lane & 1, p = lane >> 1;               // half (axis) and particle of this lane
    const int mo = HH * hf, oo = HH - mo;                 // my / the partner's hidden-unit offset
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int dim_lo = blockIdx.z * W;
    const int dim_step = (gridDim.z > 1) ? D : W;
    // Which dim a wave takes.  A workgroup's waves go to the four SIMDs cyclically, so waves w and w + 4 share one; with
    // 5..8 waves the SIMDs that host two of them set the pace of every layer (55 % of the wave-cycles of C2 were barrier
    // waits).  Units get dearer with the dim (layer 0 contracts i inputs, dL/dx has i outputs) and dim 0 has no
    // conditioner at all, so the PAIRED waves take the cheapest dims and the waves that own a SIMD the dearest:
    // rank of wave w in the order (0, 4, 1, 5, 2, 6, 3, 7).  NFISAM_DIM_PAIRING=0 in the launcher keeps w.
    const int wdim = (a.pair_dims != 0 && W > 4 && W <= 8) ? (w < 4 ? w + (w < W - 4 ? w : W - 4) : 2 * (w - 4) + 1) : w;
    if (dim_lo >= D) return;
    const int gp = p0 + p;
    const bool valid = gp < n;
    const int DT = D * XS2;
    STAMP_DECL
    STAMP(0);

    const int Pk = a.layer_stride > 0 ? a.layer_stride : LY::count(D);
    const int dim_hi = (gridDim.z > 1) ? ((dim_lo + W < D) ? dim_lo + W : D) : D;
    const int w_lo = (gridDim.z > 1 && dim_lo > 0) ? LY::off(dim_lo) : 0;
    const int w_hi = (gridDim.z > 1) ? LY::off(dim_hi) : L * Pk;
    float* wlds = smem;               // [w_hi - w_lo] parameter copy (WL only)
    float* xs = smem + (WL ? a.wl_floats : 0);   // [L][D][XS2] layer inputs, dimension-major
    const int gt = a.g_tiles ? DT : 0;   // dL/dx tiles exist only when a layer input gradient is needed
    float* g0 = xs + L * DT;          // [D][XS2]
    float* g1 = g0 + gt;              // [D][XS2]
    float* ones = g1 + gt;            // [XS2]
    float* stg = ones + XS2 + w * (StgRows<K, H>::split * XS2);   // wave-private staging tile

    // ---- prologue: all global loads first (particle tile + parameter rows), one wait, then LDS ----
    {
        constexpr int XB = 8, WB = 4;
        const int nx = D * TILE2;
        const int lim = ((n - p0) < TILE2 ? (n - p0) : TILE2) * D;
        const float* xt = x + (size_t)p0 * D;
        const float invD = 1.0f / (float)D;
        const int tot4 = WL ? ((w_hi - w_lo) >> 2) : 0;
        const f32x4* wsrc = (const f32x4*)(kparams + w_lo);
        f32x4* wdst = (f32x4*)wlds;
        int e0 = threadIdx.x, f0 = threadIdx.x;
        while (e0 < nx || f0 < tot4) {
            float xv[XB];
            f32x4 wv[WB];
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                xv[u] = (e < lim) ? xt[e] : 0.0f;
            }
            if constexpr (WL) {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int f = f0 + u * (int)blockDim.x;
                    wv[u] = (f < tot4) ? wsrc[f] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (st_stop != 0 || st_step + a.iter_idx >= a.max_iters) return;      // block-uniform
#pragma unroll
            for (int u = 0; u < XB; ++u) {
                const int e = e0 + u * (int)blockDim.x;
                if (e < nx) {
                    int pq = (int)(((float)e + 0.5f) * invD);
                    int k = e - pq * D;
                    if (k < 0) { k += D; pq -= 1; }
                    if (k >= D) { k -= D; pq += 1; }
                    xs[k * XS2 + pq] = xv[u];
                }
            }
            if constexpr (WL) {
#pragma unroll
                for (int u = 0; u < WB; ++u) {
                    const int f = f0 + u * (int)blockDim.x;
                    if (f < tot4) wdst[f] = wv[u];
                }
            }
            e0 += XB * (int)blockDim.x;
            f0 += WB * (int)blockDim.x;
        }
    }
    if (threadIdx.x < XS2) ones[threadIdx.x] = 1.0f;
    __syncthreads();

    const float* kp;
    if constexpr (WL) kp = wlds - w_lo; else kp = kparams;
    STAMP(1);

    // ---- forward-only passes: layers 0 .. L-2 (the last layer is recomputed in backward) ---
    for (int l = 0; l + 1 < L; ++l) {
        const float* lp = kp + (size_t)l * Pk;
        const float* xin = xs + l * DT;
        float* xout = xs + (l + 1) * DT;
        for (int i = dim_lo + wdim; i < D; i += dim_step) {
            float h1m[HH], h1o[HH], h2m[HH], h2o[HH], th[HP];
            load_theta2<K, H>(lp, i, xin, p, hf, h1m, h1o, h2m, h2o, th);
            Spline2<K> S;
            float z, lad;
            spline_eval2<K, false>(xin[i * XS2 + p], th, hf, B, S, z, lad);
            if (hf == 0) xout[i * XS2 + p] = z;
        }
        STAMP(10);
        __syncthreads();
        STAMP(11);
    }

    // ---- backward with recompute, last layer first ------------------------------------------
    float lossv = 0.0f;
    float* gcur = g0;
    float* gprev = g1;
    const int r16 = lane & 15, kq = lane >> 4;
    for (int l = L - 1; l >= 0; --l) {
        const bool last = (l == L - 1);
        const bool need_gx = (l > 0) || (a.gx != nullptr);
        const float* lp = kp + (size_t)l * Pk;
        float* Gl = G + (size_t)l * Pk;
        const float* xin = xs + l * DT;
        if (need_gx) {
            for (int e = threadIdx.x; e < DT; e += blockDim.x) gprev[e] = 0.0f;
            __syncthreads();
        }
        for (int i = dim_lo + wdim; i < D; i += dim_step) {
            float h1m[HH], h1o[HH], h2m[HH], h2o[HH], th[HP], gth[HP];
            STAMP(2);
            load_theta2<K, H>(lp, i, xin, p, hf, h1m, h1o, h2m, h2o, th);
            STAMP(3);
            Spline2<K> S;
            float z, lad;
            spline_eval2<K, false>(xin[i * XS2 + p], th, hf, B, S, z, lad);
            STAMP(4);
            float gz, gl;
            if (a.nll_mode) {
                gl = -1.0f;
                gz = last ? z : gcur[i * XS2 + p];
                if (valid && hf == 0) lossv += (last ? 0.5f * z * z : 0.0f) - lad;
            } else {
                gl = (a.gl != nullptr && valid) ? a.gl[gp] : 0.0f;
                gz = last ? (valid ? a.gz[(size_t)gp * D + i] : 0.0f) : gcur[i * XS2 + p];
            }
            if (!valid) { gz = 0.0f; gl = 0.0f; }
            const float gxs = spline_backward2<K>(S, hf, B, gz, gl, gth);
            if (need_gx && hf == 0) atomicAdd(&gprev[i * XS2 + p], gxs);
            STAMP(5);

            if (i == 0) {   // init_param: plain sum over particles of gth
                constexpr int N0 = (HP <= 8) ? 8 : ((HP <= 16) ? 16 : 32);
                float v[N0];
#pragma unroll
                for (int t = 0; t < N0; ++t) v[t] = (t < HP) ? gth[t] : 0.0f;
                const float r = butterfly2<N0>(v, p);
                if (p < HP) gsink(&Gl[HP * hf + p], r, slab);
                continue;
            }
            const float* blk = lp + LY::off(i);
            float* Gb = Gl + LY::off(i);
            // ---- per-particle back-propagation through the conditioner (VALU, partial sums + DPP) ----
            float ga2m[HH], ga1m[HH];
            {
                const float* W2 = blk + LY::oW2(i) + HP * hf;
                float pm[HH], po[HH];
#pragma unroll
                for (int kk = 0; kk < HH; ++kk) {
                    float wr[HP], wo[HP];
                    load_row_used<HP, UO>(W2 + (mo + kk) * PoP, wr);
                    load_row_used<HP, UO>(W2 + (oo + kk) * PoP, wo);
                    float am = 0.0f, ao = 0.0f;
#pragma unroll
                    for (int o = 0; o < UO; ++o) {
                        am = __builtin_fmaf(wr[o], gth[o], am);
                        ao = __builtin_fmaf(wo[o], gth[o], ao);
                    }
                    pm[kk] = am; po[kk] = ao;
                }
#pragma unroll
                for (int kk = 0; kk < HH; ++kk) {
                    const float gh2 = pm[kk] + pswap(po[kk]);
                    ga2m[kk] = gh2 * (1.0f - h2m[kk] * h2m[kk]);
                }
                const float* W1 = blk + LY::oW1(i) + mo;
                float qm[HH], qo[HH];
#pragma unroll
                for (int kk = 0; kk < HH; ++kk) {
                    float wr[HH], wo[HH];
                    load_row<HH>(W1 + (mo + kk) * H, wr);
                    load_row<HH>(W1 + (oo + kk) * H, wo);
                    float am = 0.0f, ao = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < HH; ++jj) {
                        am = __builtin_fmaf(wr[jj], ga2m[jj], am);
                        ao = __builtin_fmaf(wo[jj], ga2m[jj], ao);
                    }
                    qm[kk] = am; qo[kk] = ao;
                }
#pragma unroll
                for (int kk = 0; kk < HH; ++kk) {
                    const float gh1 = qm[kk] + pswap(qo[kk]);
                    ga1m[kk] = gh1 * (1.0f - h1m[kk] * h1m[kk]);
                }
                if (need_gx) {
                    const float* W0 = blk + mo;
                    for (int k = 0; k < i; k += 2) {       // the pair splits two input columns (row i aliases b0: unused)
                        float w0[HH], w1[HH];
                        load_row<HH>(W0 + k * H, w0);
                        load_row<HH>(W0 + (k + 1) * H, w1);
                        float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
                        for (int jj = 0; jj < HH; ++jj) {
                            r0 = __builtin_fmaf(w0[jj], ga1m[jj], r0);
                            r1 = __builtin_fmaf(w1[jj], ga1m[jj], r1);
                        }
                        const float t0 = r0 + pswap(r0), t1 = r1 + pswap(r1);
                        const int kk = k + hf;
                        if (kk < i) atomicAdd(&gprev[kk * XS2 + p], hf ? t1 : t0);
                    }
                }
            }
            STAMP(6);
            // ======== weight gradients on the matrix cores: one staging round for all three GEMMs ========
            //   dW2t | db2 = [h2, 1]^T (x) gth ;  dW1t | db1 = [h1, 1]^T (x) ga2 ;  dW0t | db0 = [x, 1]^T (x) ga1
            {
                constexpr int R2 = PoP + 16;                   // first row of the ga2 | ga1 | h1 group
#pragma unroll
                for (int o = 0; o < HP; ++o) stg[(HP * hf + o) * XS2 + p] = gth[o];
#pragma unroll
                for (int kk = 0; kk < HH; ++kk) {
                    stg[(PoP + mo + kk) * XS2 + p] = h2m[kk];
                    stg[(R2 + mo + kk) * XS2 + p] = ga2m[kk];
                    stg[(R2 + H + mo + kk) * XS2 + p] = ga1m[kk];
                    stg[(R2 + 2 * H + mo + kk) * XS2 + p] = h1m[kk];
                }
                wave_lds_sync();
                // operand rows: lane&15 = feature, lane>>4 = particle within the k-group of 4; the bias column
                // (and the unused columns beyond it) read the constant-one row
                const float* pa = stg + r16 * XS2 + kq;                                    // gth rows 16t + r16
                const float* pb = ((r16 < H) ? (stg + (PoP + r16) * XS2) : ones) + kq;     // [h2 | 1]
                const float* pa2 = stg + (R2 + r16) * XS2 + kq;                            // rows 0..7 ga2, 8..15 ga1
                const float* pb1 = ((r16 < H) ? (stg + (R2 + 2 * H + r16) * XS2) : ones) + kq;   // [h1 | 1]
                f32x4 cacc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) cacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 c1 = {0.f, 0.f, 0.f, 0.f};
                float areg[TILE2 / 4];
#pragma unroll
                for (int s4 = 0; s4 < TILE2; s4 += 4) {
                    const float b = pb[s4];
#pragma unroll
                    for (int t = 0; t < NT; ++t) cacc[t] = mfma4(pa[t * 16 * XS2 + s4], b, cacc[t]);
                    areg[s4 / 4] = pa2[s4];
                    c1 = mfma4(areg[s4 / 4], pb1[s4], c1);
                }
                STAMP(7);
                // C layout: col = lane&15 (= c), rows 4*(lane>>4)+r: four consecutive outputs per lane
                float* Gw2 = Gb + LY::oW2(i);
                if (slab) {
                    if (r16 <= H) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gw2[r16 * PoP + 16 * t + 4 * kq], cacc[t], true);
                    }
                } else {
                    // atomics: transpose through LDS to the flat order so that a wave's 64 atomics hit 256
                    // consecutive bytes (scattered float atomics serialise in the memory pipeline)
                    wave_lds_sync();
                    if (r16 <= H) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) {
                                float* d = &stg[r16 * PoP + 16 * t + 4 * kq];
                                d[0] = cacc[t].x; d[1] = cacc[t].y; d[2] = cacc[t].z; d[3] = cacc[t].w;
                            }
                    }
                    wave_lds_sync();
                    constexpr int TOT = (H + 1) * PoP;
#pragma unroll
                    for (int c = 0; c < (TOT + 63) / 64; ++c) {
                        const int f = c * 64 + lane;
                        if (f < TOT) atomicAdd(&Gw2[f], stg[f]);
                    }
                }
                // rows 0..7 (kq < 2) of c1 are ga2[j], j = 4*kq + r ; flat f = c*H + j, c <= H
                if (kq < 2 && r16 <= H) gsink4(&(Gb + LY::oW1(i))[r16 * H + 4 * kq], c1, slab);
                // x tiles (16 input columns each); column i is the bias (ones row)
                float* Gw0 = Gb;
                for (int ct = 0; ct * 16 <= i; ++ct) {
                    const int cab = ct * 16 + r16;
                    const float* pb0 = ((cab < i) ? (xin + cab * XS2) : ones) + kq;
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s4 = 0; s4 < TILE2; s4 += 4) c0 = mfma4(areg[s4 / 4], pb0[s4], c0);
                    // rows 8..15 (kq >= 2) are ga1[j], j = 4*(kq-2)+r ; flat f = cab*H + j
                    if (kq >= 2 && cab <= i) gsink4(&Gw0[cab * H + 4 * (kq - 2)], c0, slab);
                }
                wave_lds_sync();
            }
            STAMP(8);
        }
        STAMP(12);
        __syncthreads();
        STAMP(13);
        float* tmp = gcur; gcur = gprev; gprev = tmp;
    }

    if (a.gx != nullptr) {   // gcur now holds dL/dx of layer 0's input
        for (int e = threadIdx.x; e < D * TILE2; e += blockDim.x) {
            const int pp = e / D, k = e - pp * D;
            const int q = p0 + pp;
            if (q < n) a.gx[(size_t)q * D + k] = gcur[k * XS2 + pp];
        }
    }
    STAMP(9);
    if (a.nll_mode) {
        const float tot = wave_sum(lossv);
        if (lane == 0) {
            float* dst = (st != nullptr) ? &ring[((st_step + a.iter_idx) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                   ((blockIdx.x * 7 + blockIdx.z * 13 + w) & (LOSS_SLOTS - 1))]
                                         : a.loss_sum;
            if (dst != nullptr) atomicAdd(dst, tot);
        }
    }
}

// =============================================================================================
// training / VJP kernel, TWO DIMS PER WAVE (multi-layer flows and VJP launches in the latency regime).
//
// The layers of a flow are sequential and every layer needs every dim of its input (src/flows/models.py:11-24), so a
// block owns a particle tile through all layers and its waves meet at a barrier per layer: the time of a launch is
// (2L - 1) x the time of ONE (layer, dim) unit on the slowest SIMD.  nsf_train2_kernel put one dim on a wave (two lanes
// per particle, VALU conditioner with per-lane weights from LDS): D = 6 means six waves on four SIMDs, and a unit took
// ~13 k cycles.  Here a wave is the dim-major kernel's unit (nsf_train1_kernel: conditioner on 4x4x1 MFMA chains, lean
// spline, operand staging for the gradient GEMMs) with the two halves of the wave on two different dims of the same 32
// particles (nsf_cond_mfma.h, PairPanel): D = 6 is three waves on three SIMDs, ~10 k cycles per unit.
//   grid (32-particle tiles, cliques), block = W = ceil(D / 2) <= 8 waves; wave w owns the dims 2w, 2w + 1
//   LDS: ones | layer inputs [L][D][XS2] | dL/dx buffers 2 x [pair_g_rows(D)][XS2] | per wave: staging [16][XS] + h1 [H][XS] |
//        panels [L][D][PairPanel::floats(D)] (each wave stages and reads only its own dims')
// Same arguments, gradient sinks, loss ring and workspace layout as nsf_train2_kernel (a drop-in at the launch site).
// =============================================================================================
// rows of one dL/dx buffer: row k = the gradient through dim k's own spline argument, row D + i(i-1)/2 + k = dim i's
// conditioner's contribution to input k < i.  Every row has ONE writer (plain stores, nothing to zero); the reader of
// dim k adds its column of the triangle in a fixed order -- reproducible, and no LDS atomics in the wave's stream.
__host__ __device__ static inline int pair_g_rows(int D) { return D + D * (D - 1) / 2; }
__host__ __device__ static inline int pair_tile_floats(int L, int D, int g_tiles) {
    return (ONES_ROW + (L * D + 2 * g_tiles * pair_g_rows(D)) * XS2 + 3) & ~3;
}
constexpr int PAIR_WAVE_FLOATS = (16 + 8) * XS;             // staging rows + h1 rows (H <= 8)
// hidden_dim 16 (round 5): sixteen h1 rows; and the panels of ONE layer resident at a time (9.1 KB per (layer, dim): all layers
// of C2's shape would be 218 KB) -- a wave brings the panels of its two dims at the top of every stage (they are wave-private:
// no barrier), from the clique's panel image (16-byte copies) or, iteration 0 of a chunk / VJP calls, from the parameters
template <int H> __host__ __device__ constexpr int pair_wave_floats() { return (H == 16) ? (16 + 16) * XS : PAIR_WAVE_FLOATS; }
// one layer's panels of the wave's dims out of the panel image into LDS: rounds of 20 sixteen-byte words per lane, all loads of a
// round in flight together (two panels of hidden_dim 16 at num_knots 9 are 18 words per lane: ONE memory round trip per stage;
// three rounds of 8 words cost C2's shape 57.1 instead of the us per iteration below)
__device__ __forceinline__ void copy_pair_panels(const float* image_generic, size_t pstride, int PS, int iA, int nd, int l, int lane, float* lay0) {
    typedef const __attribute__((address_space(1))) cm_f32x4* gv4;
    constexpr int U = 20;
    const int n4 = (nd * PS) >> 2;
    gv4 src = (gv4)(image_generic + (size_t)l * pstride + (size_t)iA * PS);
    cm_f32x4* dst = (cm_f32x4*)(lay0 + (size_t)iA * PS);
    for (int base = 0; base < n4; base += 64 * U) {
        cm_f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int e = base + lane + 64 * u; v[u] = src[e < n4 ? e : 0]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int e = base + lane + 64 * u; if (e < n4) dst[e] = v[u]; }
    }
}
static_assert(PAIR_MAP_OFFSETS == PAIR_MAX_D + 1, "one map per clique width 0 .. PAIR_MAX_D");
struct PairMapOffsets { uint32_t at[PAIR_MAX_D + 1]; };     // word offset of clique width D's map in the table (kernel argument: no dependent load)

template <int K, int H>
__global__ void __launch_bounds__(512) nsf_train3_kernel(TrainArgs a, const uint32_t* pair_map, PairMapOffsets offs) {
    using LY = Layout<K, H>;
    using PP = PairPanel<K, H>;
    constexpr int PoP = LY::PoP;
    constexpr int NT = (PoP + 15) / 16;
    constexpr int NS = TILE / 4, NSH = NS / 2;                // MFMA k-steps over the wave's 64 columns / over one dim's 32
    constexpr int QH = H / 4;
    static_assert((H == 16 || H == 8 || H == 4) && NT <= 4, "H <= 8: ga2|ga1 share one 16-row operand tile; H = 16: one tile each, bias chains");
    constexpr bool WIDE_H = (H == 16);                        // as in nsf_train1_kernel: no spare column for the bias in [h | 1]
    constexpr bool STREAM = WIDE_H;                           // the panels of ONE layer resident (pair_wave_floats<H>, above)
    constexpr int PWF = pair_wave_floats<H>();
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // the clique's descriptor in ONE scalar load: from the device array, or (single-clique calls) from the kernel-argument
    // segment itself (TrainArgs is the first argument)
    typedef const __attribute__((address_space(4))) nfisam_clique cclique;
    typedef const __attribute__((address_space(4))) char cchar;
    cclique* cp = (a.cliques != nullptr) ? (cclique*)(a.cliques + blockIdx.y)
                                         : (cclique*)((cchar*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TrainArgs, single));
    const gfloat* x = (const gfloat*)cp->x;
    const float* kparams = cp->kparams;
    gfloat* G = (gfloat*)cp->kgrad;
    typedef __attribute__((address_space(1))) nfisam_train_state gstate;
    gstate* st = (gstate*)cp->state;
    const int n = cp->n;
    const int D = cp->D;
    const int L = a.L;
    const float B = a.B;
    const bool slab = a.slab != 0;
    const int Pk = a.layer_stride > 0 ? a.layer_stride : LY::count(D);
    const size_t gstride = (size_t)L * (size_t)Pk;
    gfloat* ring = G + (slab ? (size_t)gridDim.x : (size_t)1) * gstride;   // loss ring behind the gradient copies
    if (slab) G += (size_t)blockIdx.x * gstride;

    const int p0 = blockIdx.x * TILE2;
    if (p0 >= n) return;
    int st_stop = 0, st_step = 0;
    if (st != nullptr) {
        st_stop = __hip_atomic_load(&st->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_step = __hip_atomic_load(&st->step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 5, p = lane & 31;                  // which dim of the pair, which particle of the tile
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int npairs = (D + 1) >> 1;
    const int gp = p0 + p;
    const bool valid = gp < n;
    const int DT = D * XS2;
    const int PS = PP::floats(D), s0 = PP::s0(D), oW0N = PP::oW0N(D);

    float* ones = smem;                                       // [ONES_ROW]
    float* xs = ones + ONES_ROW;                              // [L][D][XS2] layer inputs, dimension-major
    const int gt = a.g_tiles ? pair_g_rows(D) * XS2 : 0;
    float* g0 = xs + L * DT;
    float* g1 = g0 + gt;
    float* stg = smem + pair_tile_floats(L, a.xrows, a.g_tiles) + (size_t)w * PWF;   // (sized for the launch's widest clique)
    float* hrow = stg + 16 * XS;
    float* panels = smem + pair_tile_floats(L, a.xrows, a.g_tiles) + (size_t)W * PWF;
    const unsigned stg_lane = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(stg + lane);
    const int r16 = lane & 15, kq = lane >> 4;
    STAMP_DECL
    STAMP(0);

    const float* image = (const float*)(ring + LOSS_RING * LOSS_SLOTS + FUSED_COUNTERS);   // [L][D][PS], see build_pair_map
    // ---- prologue: own panels (all layers), the particle tile (every wave writes all of it: same values, no barrier) ----
    {
        const uint32_t* map = pair_map + offs.at[D];
        const int lim = ((n - p0) < TILE2 ? (n - p0) : TILE2) * D;
        const gfloat* xt = x + (size_t)p0 * D;
        constexpr int XB = 8;                                 // D <= 16: 32 x 16 / 64 elements per lane
        float xv[XB];
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const int e = lane + 64 * u;
            if (64 * u < D * TILE2) xv[u] = xt[e < lim ? e : 0];        // (wave-uniform guard; rows past n are zeroed below)
        }
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(14);
#endif
        if (w < npairs) {                                     // W = ceil(max D / 2): a wave owns ONE pair (or none, in a narrower clique)
            const int j = w;
            if constexpr (STREAM) {                            // the first stage's layer: 0 (forward passes ahead) or the only one
                if (a.pair_image) copy_pair_panels(image, (size_t)D * PS, PS, 2 * j, (2 * j + 1 < D) ? 2 : 1, 0, lane, panels);
                else stage_pair_panels<K, H>(panels, PS, (size_t)D * PS, kparams, (size_t)Pk, map, 2 * j, (2 * j + 1 < D) ? 2 : 1, 1, lane);
            } else if (a.pair_image) {                         // layer 0 now, layer l + 1 under forward stage l
                cm_f32x4 pv[PAIR_PANEL_WORDS];
                load_pair_panels(image, (size_t)D * PS, PS, 2 * j, (2 * j + 1 < D) ? 2 : 1, 0, lane, pv);
                store_pair_panels(panels, (size_t)D * PS, PS, 2 * j, (2 * j + 1 < D) ? 2 : 1, 0, lane, pv);
            } else
                stage_pair_panels<K, H>(panels, PS, (size_t)D * PS, kparams, (size_t)Pk, map, 2 * j, (2 * j + 1 < D) ? 2 : 1, L, lane);
        }
        // the state words were requested at kernel entry and are LOOKED AT only now, behind the prologue's own loads (the
        // empty asm is a use in front of the branch: the compiler does not pull the wait up to the loads)
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(3);
#endif
        asm volatile("" : "+v"(st_stop), "+v"(st_step));
        if (st_stop != 0 || st_step + a.iter_idx >= a.max_iters) return;      // block-uniform
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(15);
#endif
        const float invD = 1.0f / (float)D;
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const int e = lane + 64 * u;
            if (e < D * TILE2) {
                int pq = (int)(((float)e + 0.5f) * invD);
                int k = e - pq * D;
                if (k < 0) { k += D; pq -= 1; }
                if (k >= D) { k -= D; pq += 1; }
                xs[k * XS2 + pq] = (e < lim) ? xv[u] : 0.0f;
            }
        }
        ones[lane] = 1.0f;
        if (lane < ONES_ROW - 64) ones[64 + lane] = 1.0f;
        wave_lds_sync();
    }
    STAMP(1);
    // STREAM: layer `l`'s panels of this wave's dims into the (one) resident slot; wave-private, the wave's own LDS operations
    // are in order -- no barrier
    int resident = 0;
    auto bring_layer = [&](int l) {
        if (w < npairs && l != resident) {
            const int j = w, nd = (2 * j + 1 < D) ? 2 : 1;
            if (a.pair_image) copy_pair_panels(image, (size_t)D * PS, PS, 2 * j, nd, l, lane, panels);
            else stage_pair_panels<K, H>(panels, PS, (size_t)D * PS, kparams + (size_t)l * Pk, (size_t)Pk, pair_map + offs.at[D], 2 * j, nd, 1, lane);
            wave_lds_sync();
        }
        resident = l;
    };

    // ---- forward-only passes: layers 0 .. L-2.  With a workspace behind kgrad (training plans) the wave parks what the
    //      backward pass needs of each of them in device memory (NF coalesced rows of 64 floats; it stays in this XCD's L2) and
    //      reads it back one layer ahead of its use; otherwise the backward pass recomputes conditioner and spline. ---
    constexpr int NF = pair_stash_fields(K, H);
    // (hidden_dim 16 parks its 2 x 16 + 2 K + 12 floats per lane too while they fit the register file next to the GEMMs' operands:
    //  248 VGPRs at num_knots 9, scratch from 12 up -- there the backward stages recompute conditioner and spline)
    const bool stash = (!WIDE_H || K <= 11) && a.pair_stash != 0;
    gfloat* stash_w = (gfloat*)image + (size_t)L * D * PS + (((size_t)blockIdx.x * (L - 1)) * npairs + w) * (NF * 64) + 4 * lane;
    const size_t stash_layer = (size_t)npairs * (NF * 64);
    float lossv = 0.0f;
    for (int l = 0; l + 1 < L; ++l) {
        const float* xin = xs + l * DT;
        float* xout = xs + (l + 1) * DT;
        if constexpr (STREAM) bring_layer(l);
        if (w < npairs) {                                     // W = ceil(max D / 2): a wave owns ONE pair (or none, in a narrower clique)
            const int j = w;
            const int i = 2 * j + sub;
            const bool dim_ok = i < D;
            const int ic = dim_ok ? i : D - 1;
            const int imax = (2 * j + 1 < D) ? 2 * j + 1 : D - 1;
            const float* pan = panels + ((size_t)(STREAM ? 0 : l) * D + ic) * PS;
            cm_f32x4 pnext[PAIR_PANEL_WORDS];
            if (!STREAM && a.pair_image) load_pair_panels(image, (size_t)D * PS, PS, 2 * j, (2 * j + 1 < D) ? 2 : 1, l + 1, lane, pnext);
            float h1[H], h2[H], th[PoP];
            cond_forward_mfma<K, H>(pan, imax, s0, xin, XS2, lane, p, h1, h2, th);
            SplineT<K> S;
            float z, lad;
            spline_train_fwd<K, PoP>(xin[ic * XS2 + p], th, B, S, z, lad);
            if (dim_ok) xout[i * XS2 + p] = z;
            if (!STREAM && a.pair_image) store_pair_panels(panels, (size_t)D * PS, PS, 2 * j, (2 * j + 1 < D) ? 2 : 1, l + 1, lane, pnext);
            if (stash) {
                float sv[NF];
                stash_pack<K, H>(h1, h2, S, sv);
                gvf4_t* dst = (gvf4_t*)(stash_w + (size_t)l * stash_layer);
#pragma unroll
                for (int f = 0; f < NF; f += 4) dst[(f >> 2) * 64] = vf4_t{sv[f], sv[f + 1], sv[f + 2], sv[f + 3]};
                if (a.nll_mode && valid && dim_ok) lossv -= lad;
            }
        }
        STAMP(10);
        __syncthreads();
        STAMP(11);
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(16 + (l & 7));
#endif
    }

    // ---- backward, last layer first ------------------------------------------
    float nxt[NF];                                            // parked state of the next stage's layer, in flight during this stage's GEMMs
#pragma unroll
    for (int f = 0; f < NF; ++f) nxt[f] = 0.0f;
    float* gcur = g0;
    float* gprev = g1;
    // dL/d(input k) of the layer above, for the lane's particle: own spline argument + the conditioners of the dims k+1 .. D-1
    // (the loop starts behind the pair's FIRST dim; the second dim skips one term)
    auto upstream = [&](const float* gb, int k, int first) {
        float acc = gb[k * XS2 + p];
        for (int src = first + 1; src < D; ++src) {
            const float v = gb[(D + ((src * (src - 1)) >> 1) + (src > k ? k : 0)) * XS2 + p];
            acc += (src > k) ? v : 0.0f;
        }
        return acc;
    };
    for (int l = L - 1; l >= 0; --l) {
        const bool last = (l == L - 1);
        const bool need_gx = (l > 0) || (a.gx != nullptr);
        gfloat* Gl = G + (size_t)l * Pk;
        const float* xin = xs + l * DT;
        if constexpr (STREAM) bring_layer(l);
        if (w < npairs) {                                     // W = ceil(max D / 2): a wave owns ONE pair (or none, in a narrower clique)
            const int j = w;
            const int i = 2 * j + sub;
            const bool dim_ok = i < D;
            const int ic = dim_ok ? i : D - 1;
            const int imax = (2 * j + 1 < D) ? 2 * j + 1 : D - 1;
            const float* pan = panels + ((size_t)(STREAM ? 0 : l) * D + ic) * PS;
            float h1[H], h2[H], gth[PoP];
            SplineT<K> S;
            float z = 0.0f, lad = 0.0f;
            STAMP(2);
            const bool parked = stash && !last;                // wave-uniform
            if (parked) {
                stash_unpack<K, H>(nxt, h1, h2, S);
            } else {
                float th[PoP];
                cond_forward_mfma<K, H>(pan, imax, s0, xin, XS2, lane, p, h1, h2, th);
                spline_train_fwd<K, PoP>(xin[ic * XS2 + p], th, B, S, z, lad);
            }
            // operands of the gradient GEMMs
            lds_rows_store<0, H, 0, H>(stg_lane, h2);
            lds_rows_store<16, H, 0, H>(stg_lane, h1);         // = hrow
            STAMP(4);
            float gz, gl;
            if (a.nll_mode) {
                gl = -1.0f;
                gz = last ? z : upstream(gcur, ic, 2 * j);
                if (valid && dim_ok && !parked) lossv += (last ? 0.5f * z * z : 0.0f) - lad;
            } else {
                gl = (a.gl != nullptr && valid) ? a.gl[gp] : 0.0f;
                gz = last ? ((valid && dim_ok) ? a.gz[(size_t)gp * D + i] : 0.0f) : upstream(gcur, ic, 2 * j);
            }
            if (!valid || !dim_ok) { gz = 0.0f; gl = 0.0f; }
            const float gxs = spline_train_bwd<K, PoP>(S, B, gz, gl, gth);
            if (need_gx && dim_ok) gprev[i * XS2 + p] = gxs;
            STAMP(5);
            // ---- per-particle back-propagation through the conditioner (4x4x1 MFMA chains, nsf_cond_mfma.h) ----
            float ga2[H], ga1[H];
            cond_backward_mfma<K, H>(pan, lane, gth, h1, h2, ga2, ga1);
            if (need_gx) {                                       // dL/dx_k of the layer input, k < i: zero rows of W0N beyond
                constexpr int NG4 = (PAIR_MAX_D - 1 + 3) / 4;
                cm_f32x4 gxk[NG4];
                cond_input_grad<K, H, NG4>(pan, oW0N, lane, ga1, gxk, (imax + 3) >> 2);
                float* tri = gprev + (D + ((ic * (ic - 1)) >> 1)) * XS2 + p;
#pragma unroll
                for (int g = 0; g < NG4; ++g)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (4 * g + u < i && dim_ok) tri[(4 * g + u) * XS2] = gxk[g][u];
            }
            // ---- weight gradients on the matrix cores: the staging columns 0-31 are the first dim's particles, 32-63 the
            //      second's -- two accumulator sets, the k-steps of a chain split between them (see nsf_train1_kernel) ----
            STAMP(6);
            if (stash && l > 0) {                                // the next stage's layer: requested now, used after the barrier
                const gvf4_t* src = (const gvf4_t*)(stash_w + (size_t)(l - 1) * stash_layer);
#pragma unroll
                for (int f = 0; f < NF; f += 4) {
                    const vf4_t v4 = src[(f >> 2) * 64];
                    nxt[f] = v4.x; nxt[f + 1] = v4.y; nxt[f + 2] = v4.z; nxt[f + 3] = v4.w;
                }
            }
            const int iA = 2 * j, iB = 2 * j + 1;
            const float* pa = stg + r16 * XS + kq;
            float breg[NS], areg[NS];
            {
                wave_lds_sync();
                const float* pah = ((r16 < H) ? stg + r16 * XS : ones) + kq;
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) breg[s4] = pah[4 * s4];
                wave_lds_sync();
            }
            f32x4 cacc[2][NT], c1[2], c0[2];
            f32x4 cb2[2][NT], cb1[2];                             // WIDE_H only: the bias chains (operand 1)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { cacc[hb][t] = f32x4{0.f, 0.f, 0.f, 0.f}; cb2[hb][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                c1[hb] = f32x4{0.f, 0.f, 0.f, 0.f};
                c0[hb] = f32x4{0.f, 0.f, 0.f, 0.f};
                cb1[hb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // A generation of the 16 staging rows is read into registers completely; the NEXT one is written and its reads are
            // issued BEFORE this one's MFMAs, so that the LDS round trip runs under the 16 x 32 MFMA cycles (a lone wave per
            // SIMD has nothing else to hide it under); the last chains' second operands are requested the same way.
            const bool merged = !WIDE_H && (imax <= 16 - (H + 1));
            lds_rows_store<0, 16, 0, PoP>(stg_lane, gth);
            wave_lds_sync();
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) areg[s4] = pa[4 * s4];
            wave_lds_sync();
            float anext[NS], bm[NS];
#pragma unroll
            for (int s4 = 0; s4 < NS; ++s4) bm[s4] = 0.0f;
#pragma unroll
            for (int t = 1; t <= NT; ++t) {
                if (t == 1 && NT > 1) lds_rows_store<0, 16, 16, PoP>(stg_lane, gth);
                if (t == 2 && NT > 2) lds_rows_store<0, 16, 32, PoP>(stg_lane, gth);
                if (t == 3 && NT > 3) lds_rows_store<0, 16, 48, PoP>(stg_lane, gth);
                if (t == NT) {
                    if constexpr (WIDE_H) {
                        lds_rows_store<0, 16, 0, H>(stg_lane, ga2);
                    } else {
                        lds_rows_store<0, H, 0, H>(stg_lane, ga2);
                        lds_rows_store<H, H, 0, H>(stg_lane, ga1);
                    }
                }
                wave_lds_sync();
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) anext[s4] = pa[4 * s4];
                if (t == NT && merged) {
                    // columns [h1 (H) | 1 | x_0 .. x_{i-1}] of both products in one 16-column operand (see nsf_train1_kernel)
                    const int kx = r16 - (H + 1);
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const int ih = 2 * j + hb;
                        const bool one = (r16 == H) || kx >= ih;
                        const float* pbm = (one ? ones : ((r16 < H) ? hrow + r16 * XS + 32 * hb : xin + kx * XS2)) + kq;
#pragma unroll
                        for (int s = 0; s < NSH; ++s) bm[hb * NSH + s] = pbm[4 * s];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    cacc[s4 / NSH][t - 1] = mfma4(areg[s4], breg[s4], cacc[s4 / NSH][t - 1]);
                    if constexpr (WIDE_H) cb2[s4 / NSH][t - 1] = mfma4(areg[s4], 1.0f, cb2[s4 / NSH][t - 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_sync();
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) areg[s4] = anext[s4];
            }
            STAMP(7);
            if constexpr (WIDE_H) {
                // areg = the sixteen rows of ga2: x [h1] -> dW1t, x 1 -> db1; then ga1 (sixteen rows) x [x_0 .. x_{i-1} | 1] -> dW0t | db0
                lds_rows_store<0, 16, 0, H>(stg_lane, ga1);       // (the rows of ga2 are in registers)
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    const float* pb1 = hrow + r16 * XS + 32 * (s4 / NSH) + kq;
                    c1[s4 / NSH] = mfma4(areg[s4], pb1[4 * (s4 % NSH)], c1[s4 / NSH]);
                    cb1[s4 / NSH] = mfma4(areg[s4], 1.0f, cb1[s4 / NSH]);
                }
                wave_lds_sync();
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    const int ih = 2 * j + s4 / NSH;
                    const float* pb0 = ((r16 < ih) ? xin + r16 * XS2 : ones) + kq;
                    areg[s4] = pa[4 * s4];
                    c0[s4 / NSH] = mfma4(areg[s4], pb0[4 * (s4 % NSH)], c0[s4 / NSH]);
                }
            } else if (merged) {
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) c1[s4 / NSH] = mfma4(areg[s4], bm[s4], c1[s4 / NSH]);
            } else {
                const float* pb0[2];
                const float* pb1[2];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const int ih = 2 * j + hb;
                    pb0[hb] = ((r16 < ih) ? xin + r16 * XS2 : ones) + kq;       // input columns 0..15 (column i = bias)
                    pb1[hb] = ((r16 < H) ? hrow + r16 * XS + 32 * hb : ones) + kq;
                }
#pragma unroll
                for (int s4 = 0; s4 < NS; ++s4) {
                    c1[s4 / NSH] = mfma4(areg[s4], pb1[s4 / NSH][4 * (s4 % NSH)], c1[s4 / NSH]);
                    c0[s4 / NSH] = mfma4(areg[s4], pb0[s4 / NSH][4 * (s4 % NSH)], c0[s4 / NSH]);
                }
            }
            wave_lds_sync();
            // ---- the two dims' parameter-block gradients of this layer (C layout: col = lane&15, rows 4*(lane>>4)+r) ----
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int ih = hb ? iB : iA;
                if (ih >= D) continue;
                if (ih == 0) {                                  // the spline parameters of dim 0: db2 alone
                    if (r16 == (WIDE_H ? 0 : H)) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gl[16 * t + 4 * kq], WIDE_H ? cb2[hb][t] : cacc[hb][t], slab);
                    }
                    continue;
                }
                gfloat* Gb = Gl + LY::off(ih);
                gfloat* Gw2 = Gb + LY::oW2(ih);
                if (r16 <= H) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gw2[r16 * PoP + 16 * t + 4 * kq], cacc[hb][t], slab);
                }
                if (kq < QH && r16 <= H) gsink4(&(Gb + LY::oW1(ih))[r16 * H + 4 * kq], c1[hb], slab);
                if constexpr (WIDE_H) {                         // the bias rows (every column of a bias chain holds the same sums) + dW0t | db0
                    if (r16 == 0) {
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            if (16 * t + 4 * kq + 3 < PoP) gsink4(&Gw2[H * PoP + 16 * t + 4 * kq], cb2[hb][t], slab);
                        gsink4(&(Gb + LY::oW1(ih))[H * H + 4 * kq], cb1[hb], slab);
                    }
                    if (r16 <= ih) gsink4(&Gb[r16 * H + 4 * kq], c0[hb], slab);
                } else if (merged) {                          // columns H.. of the shared tile: bias first, then x_0..x_{i-1}
                    const int k0 = (r16 == H) ? ih : r16 - (H + 1);
                    if (kq >= QH && kq < 2 * QH && (r16 == H || (r16 > H && k0 < ih))) gsink4(&Gb[k0 * H + 4 * (kq - QH)], c1[hb], slab);
                } else if (kq >= QH && kq < 2 * QH && r16 <= ih) {
                    gsink4(&Gb[r16 * H + 4 * (kq - QH)], c0[hb], slab);
                }
            }
            STAMP(8);
        }
        STAMP(12);
        __syncthreads();
        STAMP(13);
#if defined(NSF_STAMPS) && NSF_STAMPS == 2
        STAMP(24 + (l & 7));
#endif
        float* tmp = gcur; gcur = gprev; gprev = tmp;
    }

    if (a.gx != nullptr) {   // gcur now holds dL/dx of layer 0's input
        for (int e = threadIdx.x; e < D * TILE2; e += blockDim.x) {
            const int pp = e / D, k = e - pp * D;
            const int q = p0 + pp;
            if (q < n) {
                float acc = gcur[k * XS2 + pp];
                for (int src = k + 1; src < D; ++src) acc += gcur[(D + ((src * (src - 1)) >> 1) + k) * XS2 + pp];
                a.gx[(size_t)q * D + k] = acc;
            }
        }
    }
    STAMP(9);
    if (a.nll_mode) {
        // the block's loss: the waves' sums added in wave order by one thread, ONE atomic per block into a ring slot that two
        // blocks share while the launch has <= 256 tiles (a sum of two floats does not depend on their order: the loss
        // record is the same however the blocks happen to be timed; a third of the atomics to drain at the kernel's end)
        __shared__ float s_wave_loss[8];
        const float tot = wave_sum(lossv);
        if (lane == 0) s_wave_loss[w] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            float bl = s_wave_loss[0];
            for (int ww = 1; ww < (int)(blockDim.x >> 6); ++ww) bl += s_wave_loss[ww];
            gfloat* dst = (st != nullptr) ? &ring[((st_step + a.iter_idx) & (LOSS_RING - 1)) * LOSS_SLOTS +
                                                    ((blockIdx.x >> 1) & (LOSS_SLOTS - 1))]
                                          : (gfloat*)a.loss_sum;
            if (dst != nullptr) gsink(dst, bl, false);
        }
    }
}

// =============================================================================================
// inference: forward (density direction)
// =============================================================================================
template <int K, int H>
__global__ void __launch_bounds__(512) nsf_forward_kernel(const float* __restrict__ x, const float* kparams,
                                                           int n, int D, float B, int L, int layer_stride,
                                                           float* __restrict__ z, float* __restrict__ logdet,
                                                           float* __restrict__ logprob) {
    using LY = Layout<K, H>;
    constexpr int PoP = LY::PoP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p0 = blockIdx.x * TILE;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int DT = D * TILE;
    float* xa = smem;            // [D][TILE]
    float* xb = xa + DT;         // [D][TILE]
    float* ldacc = xb + DT;      // [TILE]
    for (int e = threadIdx.x; e < DT; e += blockDim.x) {
        const int p = e / D, k = e - p * D;
        const int q = p0 + p;
        xa[k * TILE + p] = (q < n) ? x[(size_t)q * D + k] : 0.0f;
    }
    if (threadIdx.x < TILE) ldacc[threadIdx.x] = 0.0f;
    __syncthreads();
    cfloat* kp = (cfloat*)kparams;
    const int Pk = layer_stride > 0 ? layer_stride : LY::count(D);
    float* xin = xa;
    float* xout = xb;
    for (int l = 0; l < L; ++l) {
        cfloat* lp = kp + (size_t)l * Pk;
        float ld = 0.0f;
        for (int i = w; i < D; i += W) {
            float h1[H], h2[H], th[PoP];
            load_theta<K, H, cfloat*>(lp, i, xin, TILE, lane, h1, h2, th);
            Spline<K> S;
            float zz, lad;
            spline_eval<K, PoP, false>(xin[i * TILE + lane], th, B, S, zz, lad);
            xout[i * TILE + lane] = zz;
            ld += lad;
        }
        atomicAdd(&ldacc[lane], ld);
        __syncthreads();
        float* tmp = xin; xin = xout; xout = tmp;
    }
    // xin holds z
    if (z != nullptr) {
        for (int e = threadIdx.x; e < DT; e += blockDim.x) {
            const int p = e / D, k = e - p * D;
            const int q = p0 + p;
            if (q < n) z[(size_t)q * D + k] = xin[k * TILE + p];
        }
    }
    if (threadIdx.x < TILE && p0 + (int)threadIdx.x < n) {
        const int q = p0 + threadIdx.x;
        const float ld = ldacc[threadIdx.x];
        if (logdet != nullptr) logdet[q] = ld;
        if (logprob != nullptr) {
            float zz = 0.0f;
            for (int k = 0; k < D; ++k) { const float t = xin[k * TILE + threadIdx.x]; zz += t * t; }
            logprob[q] = -0.5f * zz - 0.5f * (float)D * 1.8378770664093453f + ld;
        }
    }
}

// =============================================================================================
// inference: inverse / conditional sampling (one wave per 64 particles, dims sequential)
// =============================================================================================
__device__ __forceinline__ float wrap_pi(float t) {   // src/utils/Functions.py:20-21 (python % semantics)
    const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
    float r = fmodf(t + pi, two_pi);
    if (r < 0.0f) r += two_pi;
    return r - pi;
}

template <int K, int H>
__global__ void __launch_bounds__(64) nsf_inverse_kernel(const float* __restrict__ zin, const float* __restrict__ x_sep,
                                                         const float* kparams, int n, int D, int Ds, float B, int L,
                                                         int layer_stride, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                         const uint8_t* __restrict__ circ,
                                                         float* __restrict__ x_out, float* __restrict__ logdet) {
    using LY = Layout<K, H>;
    constexpr int PoP = LY::PoP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int p0 = blockIdx.x * TILE;
    const int F = D - Ds;
    float* xs = smem;                 // [D][TILE]  row being reconstructed
    float* zs = xs + D * TILE;        // [F][TILE]  latent of the current layer
    // L > 1 with given columns: ych[l] = the given columns as layer l sees them (pushed through the marginal flow of
    // layers 0..l-1; the flow is autoregressive, so they depend on the given columns only).  The reference conditions
    // every layer on the raw columns (src/slam/NFiSAM.py:151-152), which inverts no composition.
    const bool chain = (L > 1 && Ds > 0);
    float* ych = chain ? (zs + F * TILE) : xs;      // [L][Ds][TILE]
    // given columns: normalise (NFiSAM.py:96-106)
    for (int e = lane; e < Ds * TILE; e += TILE) {
        const int p = e / Ds, k = e - p * Ds;
        const int q = p0 + p;
        float v = (q < n) ? x_sep[(size_t)q * Ds + k] : 0.0f;
        if (mean != nullptr) {
            const float d = v - mean[k];
            v = ((circ != nullptr && circ[k]) ? wrap_pi(d) : d) / stdv[k];
        }
        ych[k * TILE + p] = v;
    }
    for (int e = lane; e < F * TILE; e += TILE) {
        const int p = e / F, k = e - p * F;
        const int q = p0 + p;
        zs[k * TILE + p] = (q < n) ? zin[(size_t)q * F + k] : 0.0f;
    }
    __syncthreads();
    cfloat* kp = (cfloat*)kparams;
    const int Pk = layer_stride > 0 ? layer_stride : LY::count(D);
    if (chain) {
        for (int l = 0; l + 1 < L; ++l) {
            cfloat* lp = kp + (size_t)l * Pk;
            const float* yin = ych + (size_t)l * Ds * TILE;
            float* yout = ych + (size_t)(l + 1) * Ds * TILE;
            for (int i = 0; i < Ds; ++i) {
                float h1[H], h2[H], th[PoP];
                load_theta<K, H, cfloat*>(lp, i, yin, TILE, lane, h1, h2, th);
                Spline<K> S;
                float yi, lad;
                spline_eval<K, PoP, false>(yin[i * TILE + lane], th, B, S, yi, lad);
                yout[i * TILE + lane] = yi;
            }
        }
    }
    float ld = 0.0f;
    for (int l = L - 1; l >= 0; --l) {
        cfloat* lp = kp + (size_t)l * Pk;
        if (chain) for (int i = 0; i < Ds; ++i) xs[i * TILE + lane] = ych[((size_t)l * Ds + i) * TILE + lane];
        for (int i = Ds; i < D; ++i) {
            float h1[H], h2[H], th[PoP];
            load_theta<K, H, cfloat*>(lp, i, xs, TILE, lane, h1, h2, th);
            Spline<K> S;
            float xi, lad;
            spline_eval<K, PoP, true>(zs[(i - Ds) * TILE + lane], th, B, S, xi, lad);
            xs[i * TILE + lane] = xi;     // only this lane reads its own column entries
            ld += lad;
        }
        if (l > 0) for (int i = Ds; i < D; ++i) zs[(i - Ds) * TILE + lane] = xs[i * TILE + lane];
    }
    __syncthreads();
    for (int e = lane; e < F * TILE; e += TILE) {
        const int p = e / F, k = e - p * F;
        const int q = p0 + p;
        if (q < n) {
            float v = xs[(Ds + k) * TILE + p];
            if (mean != nullptr) {
                v = v * stdv[Ds + k] + mean[Ds + k];
                if (circ != nullptr && circ[Ds + k]) v = wrap_pi(v);
            }
            x_out[(size_t)q * F + k] = v;
        }
    }
    if (logdet != nullptr && p0 + lane < n) logdet[p0 + lane] = ld;
}

// =============================================================================================
// posterior traversal of a whole Bayes tree in ONE launch (SURVEY.md §8 f-1;
// reference: FactorGraphSolver.sample_posterior, src/slam/FactorGraphSolver.py:497-550, which
// makes one host-synchronised conditional-sampling call per clique).
// Sample j of a child clique is conditioned on sample j of its parent only, so a wave of 64
// samples can walk all cliques root -> leaves on its own: no inter-block synchronisation, samples
// never leave the device.  St / Zt are COLUMN-major [total_dim][n] (one variable column = one
// coalesced run over particles); St receives the un-normalised posterior samples.
// =============================================================================================
template <int K, int H>
__global__ void __launch_bounds__(64) nsf_posterior_walk_kernel(const nfisam_post_clique* __restrict__ table,
                                                                int n_cliques, const int32_t* __restrict__ cols,
                                                                const float* __restrict__ obs, float B, int L, int n,
                                                                int dmax, const float* __restrict__ Zt, float* __restrict__ St) {
    using LY = Layout<K, H>;
    constexpr int PoP = LY::PoP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int p = blockIdx.x * TILE + lane;
    const bool valid = p < n;
    const size_t pp = valid ? (size_t)p : 0;
    float* xs = smem;                       // [Dmax][TILE]
    float* ych = xs + (size_t)dmax * TILE;  // [L][Dmax][TILE] given columns per layer (L > 1 only, see nsf_inverse_kernel)
    int zoff = 0;                           // latent rows are consumed in walk order
    for (int c = 0; c < n_cliques; ++c) {
        const nfisam_post_clique q = table[c];
        const int n_obs = q.n_obs, Ds = q.n_obs + q.n_sep, D = Ds + q.n_frontal;
        const float* mean = q.mean;
        const float* stdv = q.std;
        const uint8_t* circ = q.circular;
        const bool chain = (L > 1 && Ds > 0);
        float* y0 = chain ? ych : xs;
        // given columns: true observations (same for every sample) then the separator samples
        for (int k = 0; k < Ds; ++k) {
            float v = (k < n_obs) ? obs[q.obs_off + k] : St[(size_t)cols[q.sep_off + (k - n_obs)] * n + pp];
            const float d = v - mean[k];
            y0[k * TILE + lane] = (circ[k] ? wrap_pi(d) : d) / stdv[k];
        }
        cfloat* kp = (cfloat*)q.kparams;
        const int Pk = LY::count(q.D_model);
        if (chain) {
            for (int l = 0; l + 1 < L; ++l) {
                cfloat* lp = kp + (size_t)l * Pk;
                const float* yin = ych + (size_t)l * dmax * TILE;
                float* yout = ych + (size_t)(l + 1) * dmax * TILE;
                for (int i = 0; i < Ds; ++i) {
                    float h1[H], h2[H], th[PoP];
                    load_theta<K, H, cfloat*>(lp, i, yin, TILE, lane, h1, h2, th);
                    Spline<K> S;
                    float yi, lad;
                    spline_eval<K, PoP, false>(yin[i * TILE + lane], th, B, S, yi, lad);
                    yout[i * TILE + lane] = yi;
                }
            }
        }
        for (int l = L - 1; l >= 0; --l) {
            cfloat* lp = kp + (size_t)l * Pk;
            if (chain) for (int i = 0; i < Ds; ++i) xs[i * TILE + lane] = ych[((size_t)l * dmax + i) * TILE + lane];
            for (int i = Ds; i < D; ++i) {
                float h1[H], h2[H], th[PoP];
                load_theta<K, H, cfloat*>(lp, i, xs, TILE, lane, h1, h2, th);
                // layer L-1 consumes the latent draw; lower layers consume the previous layer's output
                const float zin = (l == L - 1) ? Zt[(size_t)(zoff + (i - Ds)) * n + pp] : xs[i * TILE + lane];
                Spline<K> S;
                float xi, lad;
                spline_eval<K, PoP, true>(zin, th, B, S, xi, lad);
                xs[i * TILE + lane] = xi;
            }
        }
        for (int i = Ds; i < D; ++i) {
            float v = xs[i * TILE + lane] * stdv[i] + mean[i];
            if (circ[i]) v = wrap_pi(v);
            if (valid) St[(size_t)cols[q.front_off + (i - Ds)] * n + p] = v;
        }
        zoff += q.n_frontal;
        // the next clique may read the columns just written by THIS lane only: program order suffices
    }
}

// ---------------------------------------------------------------------------------------------
// The same walk, latency-engineered for trees of hundreds of small cliques (one trained flow each, all
// parameters cold): a wave covers 32 samples with two lanes per sample (nsf_split.h), and while it computes
// clique c from LDS it has the parameters, normalisation constants, column indices and latent draws of clique
// c+1 in flight into registers (software pipeline, one LDS double buffer per wave).  Only the separator
// samples (produced by the ancestors a moment earlier) are read on the critical path.
// ---------------------------------------------------------------------------------------------
struct WalkArgs {
    const nfisam_post_clique* table;
    const int32_t* cols;
    const float* obs;
    const float* Zt;
    float* St;
    float B;
    int n_cliques, L, n;
    int wmax;      // floats of one parameter buffer
    int dmax;      // largest clique dimension
};

template <int K, int H>
__device__ __forceinline__ int walk_range_start(int Ds) { return Ds > 0 ? Layout<K, H>::off(Ds) : 0; }

// A 64-byte clique descriptor fetched as 16 dwords by lanes 0..15 (vector memory: returns in order with the
// other prefetch loads, unlike a scalar load that every later LDS wait would have to wait for), made
// wave-uniform with readlane.
__device__ __forceinline__ int desc_fetch(const nfisam_post_clique* e, int lane) {
    return (lane < 16) ? ((const int*)e)[lane] : 0;
}
__device__ __forceinline__ nfisam_post_clique desc_uniform(int tw) {
    static_assert(sizeof(nfisam_post_clique) == 64, "descriptor is 16 dwords");
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = __builtin_amdgcn_readlane(tw, k);
    nfisam_post_clique q;
    __builtin_memcpy(&q, d, sizeof(q));
    return q;
}

template <int K, int H>
__global__ void __launch_bounds__(64) nsf_posterior_walk2_kernel(WalkArgs a) {
    using LY = Layout<K, H>;
    constexpr int HP = LY::HP, HH = H / 2;
    constexpr int R = 8;                       // float4 of parameters in flight per lane
    constexpr int FP = 4;                      // latent columns in flight per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int hf = lane & 1, p = lane >> 1;
    const int gp = blockIdx.x * TILE2 + p;
    const bool valid = gp < a.n;
    const size_t pp = valid ? (size_t)gp : 0;
    const int n = a.n, L = a.L, dmax = a.dmax;
    const float B = a.B;
    float* wb = smem;                                   // [2][wmax] parameters of the current / next clique
    float* aux = wb + 2 * (size_t)a.wmax;               // [2][4][dmax]: mean, std, circular, obs
    int* colb = (int*)(aux + 8 * dmax);                 // [2][dmax] separator then frontal column indices
    float* zb = (float*)(colb + 2 * dmax);              // [2][dmax][XS2] latent draws
    float* xs = zb + 2 * dmax * XS2;                    // [dmax][XS2] row being reconstructed

    // ---- clique 0: plain synchronous loads into buffer 0 ------------------------------------------
    nfisam_post_clique q = a.table[0];
    {
        const int Ds = q.n_obs + q.n_sep, D = Ds + q.n_frontal;
        const int start = walk_range_start<K, H>(Ds), len = LY::off(D) - start, Pk = LY::count(q.D_model);
        const int len4 = len >> 2;
        for (int l = 0; l < L; ++l) {
            const f32x4* src = (const f32x4*)(q.kparams + (size_t)l * Pk + start);
            f32x4* dst = (f32x4*)(wb + (size_t)l * len);
            for (int f = lane; f < len4; f += 64) dst[f] = src[f];
        }
        if (lane < D) {
            aux[lane] = q.mean[lane]; aux[dmax + lane] = q.std[lane]; aux[2 * dmax + lane] = q.circular[lane] ? 1.0f : 0.0f;
        }
        if (lane < q.n_obs) aux[3 * dmax + lane] = a.obs[q.obs_off + lane];
        if (lane < q.n_sep + q.n_frontal) colb[lane] = a.cols[q.sep_off + lane];
        for (int j = 0; j < q.n_frontal; ++j) if (hf == 0) zb[j * XS2 + p] = a.Zt[(size_t)j * n + pp];
    }
    nfisam_post_clique qn = q;
    if (a.n_cliques > 1) qn = desc_uniform(desc_fetch(a.table + 1, lane));
    wave_lds_sync();
    int zoff = 0;
    for (int c = 0; c < a.n_cliques; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        const int n_obs = q.n_obs, Ds = q.n_obs + q.n_sep, D = Ds + q.n_frontal;
        const float* ax = aux + cur * 4 * dmax;
        const int* cl = colb + cur * dmax;
        const bool more = (c + 1 < a.n_cliques);
        // ---- 1. given columns: true observations, then the separator samples written by the ancestors.  The
        //         even lane of a pair is the one that wrote them (program order makes them visible to it).
        // All loads are issued before any is consumed (one memory round trip instead of one per column): lanes 0..31
        // take the wave's 32 samples of column k, lanes 32..63 those of column k + 1 (128 contiguous bytes each).
        // The values were stored by this wave: agent-scope loads read them back from L2.
        {
            const int half = lane >> 5, pl = lane & 31;
            const int gq = blockIdx.x * TILE2 + pl;
            const size_t pq = (gq < n) ? (size_t)gq : 0;
            constexpr int KMAX = 16;                       // columns in flight per lane: covers Ds <= 32
            float v[KMAX];
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const int k = 2 * j + half;
                v[j] = 0.0f;
                if (k < Ds) {
                    if (k < n_obs) v[j] = ax[3 * dmax + k];
                    else v[j] = __hip_atomic_load(&a.St[(size_t)cl[k - n_obs] * n + pq], __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const int k = 2 * j + half;
                if (k < Ds) {
                    const float d = v[j] - ax[k];
                    xs[k * XS2 + pl] = ((ax[2 * dmax + k] != 0.0f) ? wrap_pi(d) : d) / ax[dmax + k];
                }
            }
            for (int k = 2 * KMAX + half; k < Ds; k += 2) {          // very wide separators: the rest, one by one
                float vv = (k < n_obs) ? ax[3 * dmax + k]
                                       : __hip_atomic_load(&a.St[(size_t)cl[k - n_obs] * n + pq], __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                const float d = vv - ax[k];
                xs[k * XS2 + pl] = ((ax[2 * dmax + k] != 0.0f) ? wrap_pi(d) : d) / ax[dmax + k];
            }
        }
        wave_lds_sync();
        // ---- 2. next clique: everything that does not depend on this one goes in flight now ----------
        const int tw2 = (c + 2 < a.n_cliques) ? desc_fetch(a.table + c + 2, lane) : 0;
        const int nDs = qn.n_obs + qn.n_sep, nD = nDs + qn.n_frontal;
        const int nstart = walk_range_start<K, H>(nDs), nlen = LY::off(nD) - nstart, nPk = LY::count(qn.D_model);
        const int nlen4 = nlen >> 2, ntot4 = more ? L * nlen4 : 0;
        f32x4 pf[R];
        float pm = 0.0f, ps = 1.0f, pc = 0.0f, po = 0.0f, pz[FP];
        int pcol = 0;
        const int nz0 = zoff + q.n_frontal;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int f = lane + 64 * r;
            pf[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < ntot4) {
                const int l = f / nlen4, o = f - l * nlen4;
                pf[r] = ((const f32x4*)(qn.kparams + (size_t)l * nPk + nstart))[o];
            }
        }
        if (more) {
            if (lane < nD) { pm = qn.mean[lane]; ps = qn.std[lane]; pc = qn.circular[lane] ? 1.0f : 0.0f; }
            if (lane < qn.n_obs) po = a.obs[qn.obs_off + lane];
            if (lane < qn.n_sep + qn.n_frontal) pcol = a.cols[qn.sep_off + lane];
#pragma unroll
            for (int j = 0; j < FP; ++j) pz[j] = (j < qn.n_frontal) ? a.Zt[(size_t)(nz0 + j) * n + pp] : 0.0f;
        }
        // ---- 3. this clique, from LDS only --------------------------------------------------------------
        {
            const int start = walk_range_start<K, H>(Ds), len = LY::off(D) - start;
            const float* zc = zb + cur * dmax * XS2;
            for (int l = L - 1; l >= 0; --l) {
                const float* lp = wb + (size_t)cur * a.wmax + (size_t)l * len - start;
                for (int i = Ds; i < D; ++i) {
                    float h1m[HH], h1o[HH], h2m[HH], h2o[HH], th[HP];
                    load_theta2<K, H>(lp, i, xs, p, hf, h1m, h1o, h2m, h2o, th);
                    const float zin = (l == L - 1) ? zc[(i - Ds) * XS2 + p] : xs[i * XS2 + p];
                    Spline2<K> S;
                    float xi, lad;
                    spline_eval2<K, true>(zin, th, hf, B, S, xi, lad);
                    if (hf == 0) xs[i * XS2 + p] = xi;
                    wave_lds_sync();
                }
            }
            for (int i = Ds; i < D; ++i) {
                float v = xs[i * XS2 + p] * ax[dmax + i] + ax[i];
                if (ax[2 * dmax + i] != 0.0f) v = wrap_pi(v);
                if (valid && hf == 0) a.St[(size_t)cl[q.n_sep + (i - Ds)] * n + gp] = v;
            }
        }
        zoff = nz0;
        // ---- 4. land the next clique's data in the other buffer -----------------------------------------
        if (more) {
            float* wn = wb + (size_t)nxt * a.wmax;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int f = lane + 64 * r;
                if (f < ntot4) ((f32x4*)wn)[f] = pf[r];          // layer l at l*nlen: f = l*nlen4 + o
            }
            for (int f = lane + 64 * R; f < ntot4; f += 64) {      // big cliques: the rest, synchronously
                const int l = f / nlen4, o = f - l * nlen4;
                ((f32x4*)wn)[f] = ((const f32x4*)(qn.kparams + (size_t)l * nPk + nstart))[o];
            }
            float* an = aux + nxt * 4 * dmax;
            if (lane < nD) { an[lane] = pm; an[dmax + lane] = ps; an[2 * dmax + lane] = pc; }
            if (lane < qn.n_obs) an[3 * dmax + lane] = po;
            if (lane < qn.n_sep + qn.n_frontal) colb[nxt * dmax + lane] = pcol;
            float* zn = zb + nxt * dmax * XS2;
#pragma unroll
            for (int j = 0; j < FP; ++j) if (j < qn.n_frontal && hf == 0) zn[j * XS2 + p] = pz[j];
            for (int j = FP; j < qn.n_frontal; ++j) if (hf == 0) zn[j * XS2 + p] = a.Zt[(size_t)(nz0 + j) * n + pp];
            wave_lds_sync();
        }
        q = qn;
        if (c + 2 < a.n_cliques) qn = desc_uniform(tw2);
    }
}


// =============================================================================================
// launchers of one (K, H) pair
// =============================================================================================
template <int KK, int HH>
static int unit_forward(const float* x, const float* kparams, int n, int D, float B, int L, int layer_stride, float* z,
                        float* logdet, float* logprob, hipStream_t s) {
    const int W = pick_waves(D);
    const size_t lds = ((size_t)2 * D * TILE + TILE) * sizeof(float);
    int rc = set_lds(nsf_forward_kernel<KK, HH>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((nsf_forward_kernel<KK, HH>), dim3((n + TILE - 1) / TILE), dim3(64 * W), lds, s, x, kparams, n, D, B, L,
                       layer_stride, z, logdet, logprob);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

template <int KK, int HH>
static int unit_inverse(const float* z, const float* x_sep, const float* kparams, int n, int D, int Ds, float B, int L,
                        int layer_stride, const float* mean, const float* stdv, const uint8_t* circular, float* x_out,
                        float* logdet, hipStream_t s) {
    const size_t lds = ((size_t)D + (D - Ds) + ((L > 1 && Ds > 0) ? (size_t)L * Ds : 0)) * TILE * sizeof(float);
    int rc = set_lds(nsf_inverse_kernel<KK, HH>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((nsf_inverse_kernel<KK, HH>), dim3((n + TILE - 1) / TILE), dim3(64), lds, s, z, x_sep, kparams, n, D, Ds,
                       B, L, layer_stride, mean, stdv, circular, x_out, logdet);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

template <int KK, int HH>
static int unit_walk(const nfisam_post_clique* table, int n_cliques, const int32_t* cols, const float* obs, int max_D, float B,
                     int L, int n, const float* Zt, float* St, hipStream_t s) {
    // pipelined two-lanes-per-sample walk (L == 1; hidden widths 8 and 16: a lane takes H / 2 hidden units, whole 16-byte rows)
    // when its LDS double buffer fits (a clique needs at most the parameter blocks of all its max_D dims), else the plain
    // one-lane-per-sample walk
    const size_t wmax = (size_t)L * (size_t)Layout<KK, HH>::off(max_D);
    const size_t lds2 = (2 * wmax + 8 * (size_t)max_D + 2 * (size_t)max_D + (size_t)3 * max_D * XS2) * sizeof(float);
    const char* walk_env = getenv("NFISAM_WALK");          // "plain" forces the one-lane walk (tests, A/B)
    const bool force_plain = (walk_env != nullptr && strcmp(walk_env, "plain") == 0);
    if constexpr (HH % 8 == 0) {
        if (L == 1 && lds2 <= 150 * 1024 && max_D <= 64 && !force_plain) {
            int rc = set_lds(nsf_posterior_walk2_kernel<KK, HH>, lds2);
            if (rc) return rc;
            WalkArgs wa;
            wa.table = table; wa.cols = cols; wa.obs = obs; wa.Zt = Zt; wa.St = St; wa.B = B;
            wa.n_cliques = n_cliques; wa.L = L; wa.n = n; wa.wmax = (int)wmax; wa.dmax = max_D;
            hipLaunchKernelGGL((nsf_posterior_walk2_kernel<KK, HH>), dim3((n + TILE2 - 1) / TILE2), dim3(64), lds2, s, wa);
            HIP_TRY(hipGetLastError());
            return NFISAM_OK;
        }
    }
    const size_t lds = (size_t)max_D * TILE * sizeof(float) * (L > 1 ? (size_t)(L + 1) : 1);
    int rc = set_lds(nsf_posterior_walk_kernel<KK, HH>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((nsf_posterior_walk_kernel<KK, HH>), dim3((n + TILE - 1) / TILE), dim3(64), lds, s, table, n_cliques, cols,
                       obs, B, L, n, max_D, Zt, St);
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

template <int KK, int HH, bool MF, bool WL, int OCC>
static int launch_train_variant(const TrainArgs& a, int n_cliques, int max_n, int W, int groups, size_t lds, hipStream_t s) {
    int rc = set_lds(nsf_train_kernel<KK, HH, MF, WL, OCC>, lds);
    if (rc) return rc;
    const int T = a.tiles_per_block > 1 ? a.tiles_per_block : 1;
    hipLaunchKernelGGL((nsf_train_kernel<KK, HH, MF, WL, OCC>), dim3((max_n + TILE * T - 1) / (TILE * T), n_cliques, groups),
                       dim3(64 * W), lds, s, a);
    return NFISAM_OK;
}

// parameter floats one block must hold: all layers, or (dims spread over grid.z) its own dims' blocks
template <int KK, int HH>
static size_t block_weight_floats(const TrainArgs& a, int max_D, int W, int groups) {
    const size_t stride = a.layer_stride > 0 ? (size_t)a.layer_stride : kcount(max_D, KK, HH);
    size_t wfloats = (size_t)a.L * stride;
    if (groups > 1) {
        const int lo = ((max_D - 1) / W) * W;            // the last group holds the largest blocks
        wfloats = (size_t)Layout<KK, HH>::off(max_D) - (lo > 0 ? (size_t)Layout<KK, HH>::off(lo) : 0);
        if (W >= max_D) wfloats = (size_t)Layout<KK, HH>::off(max_D);
        const size_t first = (size_t)Layout<KK, HH>::off(W < max_D ? W : max_D);
        if (first > wfloats) wfloats = first;
    }
    return wfloats;
}

// device-resident map of the pair kernel's panels (nsf_cond_mfma.h: build_pair_map), one per device and (K, H)
template <int KK, int HH>
struct PairMap {
    static int get(const uint32_t** out, PairMapOffsets* offs) {
        static const uint32_t* maps[16] = {nullptr};
        static PairMapOffsets table;
        int devn = 0;
        HIP_TRY(hipGetDevice(&devn));
        if (devn < 0 || devn >= 16) return NFISAM_ERR_ARG;
        if (maps[devn] == nullptr) {
            const size_t cnt = pair_map_words<KK, HH>();
            std::vector<uint32_t> host(cnt);
            build_pair_map<KK, HH>(host.data());
            for (int D = 0; D <= PAIR_MAX_D; ++D) table.at[D] = host[D];
            uint32_t* d = nullptr;
            HIP_TRY(hipMalloc((void**)&d, cnt * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(d, host.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice));
            maps[devn] = d;
        }
        if (out != nullptr) *out = maps[devn];
        if (offs != nullptr) *offs = table;
        return NFISAM_OK;
    }
};
// LDS bytes of nsf_train3_kernel; 0: the launch does not fit it (too wide, or the panels of all layers exceed the CU's LDS)
template <int KK, int HH>
static size_t pair_kernel_lds(int L, int max_D) {
    if constexpr (HH == 16) {
        // hidden_dim 16 (round 5): ONE layer's panels resident (9.1 KB per dim at K = 9), sixteen h1 rows per wave
        const char* pe = getenv("NFISAM_PAIR");
        if ((pe != nullptr && pe[0] == '0') || max_D > PAIR_MAX_D || max_D < 1) return 0;
        const int W = ((max_D + 1) / 2 < 8) ? (max_D + 1) / 2 : 8;
        const size_t fl = (size_t)pair_tile_floats(L, max_D, 1) + (size_t)W * pair_wave_floats<16>() + (size_t)max_D * PairPanel<KK, HH>::floats(max_D);
        return fl * sizeof(float) <= 160 * 1024 ? fl * sizeof(float) : 0;
    } else if constexpr (HH != 8 && HH != 4) {
        return 0;
    } else {
        // Narrow cliques stay with nsf_train2_kernel: up to four dims are one wave per SIMD there too, and its units get
        // cheaper with the dim (measured, us per iteration split / pair: D 3, L 4: 28.8 / 30.6; D 4: 30.8 / 31.3; D 5: 33.9 /
        // 33.6; D 6 (C2): 39.1 / 34.1; D 8, L 3: 36.5 / 31.1).  NFISAM_PAIR=0 | 1 forces one of them (A/B, tests).
        const char* pe = getenv("NFISAM_PAIR");
        const int min_D = ((pe != nullptr && pe[0] == '1') || HH != 8) ? 1 : 6;      // (hidden_dim 4 has no two-lanes-per-particle kernel to stay with)
        if ((pe != nullptr && pe[0] == '0') || max_D > PAIR_MAX_D || max_D < min_D) return 0;
        const int W = ((max_D + 1) / 2 < 8) ? (max_D + 1) / 2 : 8;
        const size_t fl = (size_t)pair_tile_floats(L, max_D, 1) + (size_t)W * PAIR_WAVE_FLOATS +
                          (size_t)L * max_D * PairPanel<KK, HH>::floats(max_D);
        return fl * sizeof(float) <= 160 * 1024 ? fl * sizeof(float) : 0;
    }
}
template <int KK, int HH>
static int unit_pair_map(const uint32_t** map, uint32_t* offsets) {
    if constexpr (HH != 8 && HH != 4 && HH != 16) {
        return NFISAM_ERR_ARG;
    } else {
        PairMapOffsets o;
        const int rc = PairMap<KK, HH>::get(map, &o);
        if (rc == NFISAM_OK && offsets != nullptr) memcpy(offsets, o.at, sizeof(o.at));
        return rc;
    }
}

template <int KK, int HH>
static int unit_train2(TrainArgs a, int n_cliques, int max_n, int max_D, hipStream_t s) {
    if constexpr (HH != 8 && HH != 4 && HH != 16) {
        return NFISAM_ERR_ARG;
    } else {
        const long tiles = (long)((max_n + TILE2 - 1) / TILE2) * n_cliques;
        if (!(a.L == 1 && a.gx == nullptr)) {
            // layers / dL/dx couple the dims of a tile: two dims per wave on the MFMA conditioner (nsf_train3_kernel)
            const size_t lds3 = pair_kernel_lds<KK, HH>(a.L, max_D);
            if (lds3 > 0) {
                const uint32_t* map = nullptr;
                PairMapOffsets offs;
                int rc = PairMap<KK, HH>::get(&map, &offs);
                if (rc) return rc;
                a.g_tiles = 1;
                a.xrows = max_D;
                {
                    const char* se = getenv("NFISAM_PAIR_STASH");
                    a.pair_stash = ((HH != 16 || KK <= 11) && a.pair_ws && a.L > 1 && pair_stash_fits(max_n, max_D) && !(se != nullptr && se[0] == '0')) ? 1 : 0;
                }
                const int W = ((max_D + 1) / 2 < 8) ? (max_D + 1) / 2 : 8;
                rc = set_lds(nsf_train3_kernel<KK, HH>, lds3);
                if (rc) return rc;
                hipLaunchKernelGGL((nsf_train3_kernel<KK, HH>), dim3((max_n + TILE2 - 1) / TILE2, n_cliques), dim3(64 * W), lds3, s, a, map, offs);
                HIP_TRY(hipGetLastError());
                return NFISAM_OK;
            }
        }
        if constexpr (HH != 8) {
            return NFISAM_ERR_ARG;                             // (train_tile sends hidden_dim 4 here only when the pair kernel takes the launch)
        } else {
        // L == 1 and no dL/dx requested: the dims of a tile never exchange data, so a small (latency-bound) launch
        // turns every (tile, dim) unit into its own single-wave block; big batches keep a tile's dims together.
        const bool independent_dims = (a.L == 1 && a.gx == nullptr);
        int W = pick_waves(max_D), groups = 1;
        // (measured: single-wave blocks beat 2-4 dims per block by 1-5 % here, although they dispatch more slowly)
        if (independent_dims && tiles * max_D <= 2048) { W = 1; groups = max_D; }
        else if (independent_dims && tiles * W <= 4096) groups = (max_D + W - 1) / W;
        a.g_tiles = independent_dims ? 0 : 1;
        {
            const char* pe = getenv("NFISAM_DIM_PAIRING");
            a.pair_dims = (pe != nullptr && pe[0] == '0') ? 0 : 1;
        }
        const size_t tile_floats = (((size_t)a.L + 2 * a.g_tiles) * max_D + 1 + (size_t)W * StgRows<KK, HH>::split) * XS2;
        const size_t wfloats = block_weight_floats<KK, HH>(a, max_D, W, groups);
        // the lanes of a pair read different weight rows: LDS copy whenever it fits, global loads otherwise
        const bool wl = weights_mode() != 0 && (tile_floats + wfloats) * sizeof(float) <= 150 * 1024;
        a.wl_floats = wl ? (int)wfloats : 0;
        const size_t lds = (tile_floats + (wl ? wfloats : 0)) * sizeof(float);
        int rc;
        if (wl) {
            rc = set_lds(nsf_train2_kernel<KK, HH, true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((nsf_train2_kernel<KK, HH, true>), dim3((max_n + TILE2 - 1) / TILE2, n_cliques, groups), dim3(64 * W),
                               lds, s, a);
        } else {
            rc = set_lds(nsf_train2_kernel<KK, HH, false>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((nsf_train2_kernel<KK, HH, false>), dim3((max_n + TILE2 - 1) / TILE2, n_cliques, groups), dim3(64 * W),
                               lds, s, a);
        }
        HIP_TRY(hipGetLastError());
        return NFISAM_OK;
        }
    }
}

// device-resident panel map of one (K, H) (nsf_cond_mfma.h: build_panel_map), one per device, built on first use.
// The first use must be outside a stream capture (hipMalloc + a synchronous copy): plan creation calls `prepare` before
// it starts capturing.
constexpr int PANEL_MAP_MAX_D = 96;
template <int KK, int HH>
struct PanelMap {
    static int get(int max_D, const uint32_t** out) {
        static const uint32_t* maps[16] = {nullptr};
        int devn = 0;
        HIP_TRY(hipGetDevice(&devn));
        if (devn < 0 || devn >= 16 || max_D > PANEL_MAP_MAX_D) return NFISAM_ERR_ARG;
        if (maps[devn] == nullptr) {
            const size_t cnt = Layout<KK, HH>::count(PANEL_MAP_MAX_D);
            std::vector<uint32_t> host(cnt);
            build_panel_map<KK, HH>(host.data(), PANEL_MAP_MAX_D);
            uint32_t* d = nullptr;
            HIP_TRY(hipMalloc((void**)&d, cnt * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(d, host.data(), cnt * sizeof(uint32_t), hipMemcpyHostToDevice));
            maps[devn] = d;
        }
        if (out != nullptr) *out = maps[devn];
        return NFISAM_OK;
    }
};
template <int KK, int HH, bool PERSIST> static bool lean_launch_fits(long blocks, int max_D);
template <int KK, int HH> constexpr bool lean_plain_v = HH <= 8 && KK > NSF_PLAIN_MAX_K3;
template <int KK, int HH> constexpr bool lean_persist_v = HH <= 8 && KK > NSF_PERSIST_MAX_K3;
// round 6: the two-wave build of the chunk-persistent form also serves LONE launches of num_knots 9 (a block per CU at most:
// nothing to gain from a third wave per SIMD) -- it has the registers to request the first look at the tagged copies at the
// top of the iteration -- tried and dropped -- and to let the compiler hoist the loop's derivations (nsf_train1_kernel: ROOMY)
// (... and of every other num_knots at hidden_dim <= 8: the helper waves need a two-wave build)
template <int KK, int HH> constexpr bool lean_persist_inst_v = HH <= 8;
template <int KK, int HH>
static int unit_prepare(int max_D) {
    if constexpr (HH == 8 || HH == 4 || HH == 16) {
        {
            const int rc = PairMap<KK, HH>::get(nullptr, nullptr);
            if (rc) return rc;
        }
        // (the occupancy answers the launcher will want: asked here, outside the capture of the plan's graph)
        (void)device_cus();
        if constexpr (lean_persist_inst_v<KK, HH>) (void)lean_launch_fits<KK, HH, true>(1, max_D);
        if constexpr (lean_plain_v<KK, HH>) (void)lean_launch_fits<KK, HH, false>(1, max_D);
        return max_D <= PANEL_MAP_MAX_D ? PanelMap<KK, HH>::get(max_D, nullptr) : NFISAM_OK;
    } else {
        return NFISAM_OK;
    }
}

template <int KK, int HH>
static size_t train1_lds_bytes(int max_D, int W, bool persist = false) {
    return ((size_t)PANEL_BASE + CondPanel<KK, HH>::floats(max_D) + ONES_ROW + (size_t)W * train1_wave_floats(max_D, HH) +
            (persist ? (size_t)3 * persist_keep_stride<KK, HH>(max_D) : 0)) * sizeof(float);
}
// co-resident blocks of nsf_train1_kernel<K, H, true> on this device (what its registers and LDS allow per CU x CUs)
// blocks of 256 threads with `lds` bytes of dynamic LDS that the current device holds at once for `kernel`
template <typename Kern>
static long resident_blocks(Kern kernel, size_t lds) {
    if (set_lds(kernel, lds) != NFISAM_OK) return 0;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess) return 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return (long)per_cu * (long)prop.multiProcessorCount;
}
template <int KK, int HH>
static long unit_persist_places(int max_D) {
    if constexpr (HH != 8 && HH != 4 && HH != 16) {
        return 0;
    } else {
        return resident_blocks(nsf_train1_kernel<KK, HH, true>, train1_lds_bytes<KK, HH>(max_D, 4, true));
    }
}
// the LEAN instantiations (two waves per SIMD, no scratch) exist for the (K, H) pairs whose three-wave build spills
// (lean_plain_v / lean_persist_v: declared in front of unit_prepare)
// does a launch of `blocks` four-wave blocks fit the device at the lean build's occupancy (all resident: nothing gained by a
// third wave per SIMD)?  Cached per max_D; the chunk-persistent form keeps the 1/8 margin of persist_shape (nsf_kernels.hip).
template <int KK, int HH, bool PERSIST>
static bool lean_launch_fits(long blocks, int max_D) {
    // places[device][max_D] (0: not asked yet, -1: none), under a lock: replica worker threads and multi-device processes
    // ask concurrently.  Warmed by unit_prepare (plan creation, OUTSIDE any stream capture: the first query per key calls
    // hipGetDeviceProperties and the occupancy API).
    constexpr int MAXDEV = 16;
    static std::mutex mu;
    static long places[MAXDEV][FUSED_COUNTERS + 1] = {};
    if (max_D < 1 || max_D > FUSED_COUNTERS) return false;
    const char* le = getenv("NFISAM_LEAN");                  // "0": never (A/B, tests; read per call)
    if (le != nullptr && le[0] == '0') return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return false;
    long pl;
    {
        std::lock_guard<std::mutex> lk(mu);
        pl = places[dev][max_D];
    }
    if (pl == 0) {
        if constexpr (PERSIST) pl = resident_blocks(nsf_train1_kernel<KK, HH, true, true>, train1_lds_bytes<KK, HH>(max_D, 4, true));
        else pl = resident_blocks(nsf_train1_plain_kernel<KK, HH, true>, train1_lds_bytes<KK, HH>(max_D, 4, false));
        if (pl <= 0) pl = -1;
        std::lock_guard<std::mutex> lk(mu);
        places[dev][max_D] = pl;
    }
    return pl > 0 && blocks <= (PERSIST ? pl - pl / 8 : pl);
}

template <int KK, int HH>
static int unit_train1(TrainArgs a, int n_cliques, int max_n, int max_D, hipStream_t s) {
    if constexpr (HH != 8 && HH != 4 && HH != 16) {
        return NFISAM_ERR_ARG;
    } else {
        // one wave = one dim x T tiles, dim-major blocks (nsf_train1_kernel)
        const int T = a.tiles_per_block > 1 ? a.tiles_per_block : 1;
        a.tiles_per_block = T;
        const int W = a.waves > 0 ? a.waves : dim_major_waves(n_cliques, max_n, max_D, T);
        if ((T & (T - 1)) != 0 || (W & (W - 1)) != 0 || W > 8 || T > 8) return NFISAM_ERR_ARG;
        a.t_shift = __builtin_ctz((unsigned)T);
        a.w_shift = __builtin_ctz((unsigned)W);
        a.xrows = max_D;
        // two lanes per particle (nsf_half.h; the shape decision is train_shape's, nsf_kernels.hip): 32 particles per wave
        constexpr bool half_kh = (HH == 8 && hp_of(KK) == 16);
        const bool spl = a.half != 0;
        if (spl && (!half_kh || T != 1 || !a.slab || max_D > 16 || a.L != 1)) return NFISAM_ERR_ARG;
        const int TP = spl ? 32 : TILE;                       // particles per wave-tile
        const int waves = (max_n + TP * T - 1) / (TP * T);
        const int gx = (waves + W - 1) / W;
        a.n_copies = gx;                                       // one gradient copy per block (a.slab = TP * T * W particles)
        a.grid_cliques = n_cliques;
        a.groups = n_cliques * max_D;                          // (clique, dim) groups of gx blocks, padded to the 8 XCDs
        if ((long)a.groups * (long)n_cliques >= (1L << 31) || gx > 65535 || (a.groups + 7) / 8 > 65535) return NFISAM_ERR_ARG;
        a.magic_cliques = n_cliques > 1 ? (unsigned)(((1ull << 32) + (unsigned)n_cliques - 1) / (unsigned)n_cliques) : 0u;
        int rc = PanelMap<KK, HH>::get(max_D, &a.panel_map);
        if (rc) return rc;
        const size_t lds = train1_lds_bytes<KK, HH>(max_D, W, a.persist_iters > 0);
        size_t lds_launch = lds;
        if (const char* pe = getenv("NFISAM_LDS_PAD_KB")) lds_launch += (size_t)atoi(pe) * 1024;   // experiments: fewer blocks per CU
        const bool persist = a.persist_iters > 0;
        if (persist && (T != 1 || !a.slab || !a.fused_adam || a.L != 1 || max_D > FUSED_COUNTERS)) return NFISAM_ERR_ARG;
        // the (clique, dim, 256 particles) blocks this launch really has
        long real_blocks = 0;
        {
            const nfisam_clique* hc = (a.cliques == nullptr) ? &a.single : a.host_cliques;
            if (hc != nullptr)
                for (int c = 0; c < n_cliques; ++c) real_blocks += (long)hc[c].D * ((hc[c].n + W * T * TP - 1) / (W * T * TP));
            else
                real_blocks = (long)n_cliques * max_D * gx;
        }
        // LEAN build (two waves per SIMD, no scratch: see the kernels' attribute) when the launch is resident at that occupancy anyway
        bool lean = false;
        if constexpr (lean_persist_v<KK, HH>) { if (persist && W == 4 && !spl) lean = lean_launch_fits<KK, HH, true>(real_blocks, max_D); }
        else if constexpr (lean_persist_inst_v<KK, HH>) {
            static const bool lone_lean = !(getenv("NFISAM_LONE_LEAN") != nullptr && getenv("NFISAM_LONE_LEAN")[0] == '0');
            if (persist && W == 4 && !spl && lone_lean && real_blocks <= 256) lean = lean_launch_fits<KK, HH, true>(real_blocks, max_D);
        }
        if constexpr (lean_plain_v<KK, HH>) { if (!persist && W == 4 && !spl) lean = lean_launch_fits<KK, HH, false>(real_blocks, max_D); }
        // a window-spanning launch closes its windows in the kernel: that code exists in the two-wave builds (nsf_train1_kernel: ROOMY)
        if (a.span_window > 0 && !(persist && (spl || lean) && W == 4 && max_D <= SPAN_MAX_D && n_cliques == 1)) return NFISAM_ERR_ARG;
        const bool wide = persist && gx > 8;                   // groups of 9 .. 16 blocks (n > 2048): the WIDE instantiation
        // helper waves (round 6): a two-wave build whose blocks get a CU each is launched with eight waves per block -- waves 4 .. 7
        // own no particles and take part in the staging only (one parameter per thread: stage_cond_panel_persist_solo)
        static const bool helpers_on = !(getenv("NFISAM_HELPERS") != nullptr && getenv("NFISAM_HELPERS")[0] == '0');
        // (MI355X: 256 CUs -> at most 224 such blocks, 240 for groups of nine to sixteen.  The dispatcher deals workgroups round-robin to
        //  the eight XCDs, so what must fit is the busiest XCD's share -- (octets of groups) x (blocks per group), padding included -- into
        //  its cus / 8 CUs: at most 7/8 of them by default; the sixteen-copy launches of NFISAM_HALF=2 may fill an XCD)
        const long cus = device_cus();
        const long per_xcd = (long)((a.groups + 7) / 8) * gx;
        const bool helpers = helpers_on && persist && (spl || lean || HH == 16) && W == 4 &&      // (hidden_dim 16 compiles to two waves per SIMD anyway)
                             real_blocks <= (wide ? cus - cus / 16 : cus - cus / 8) &&
                             per_xcd <= (wide ? cus / 8 : cus / 8 - cus / 64);       // (also the window-spanning launch: the in-kernel bookkeeping's work is its first four waves')
        const int BW = helpers ? 2 * W : W;                    // waves per block
        if (persist && gx > PERSIST_MAX_COPIES) return NFISAM_ERR_ARG;
        if constexpr (half_kh) {
            if (spl) {
                // (the one-launch-per-iteration twin is the same template with PERSIST = false: the same arithmetic in the same order)
                if (persist && wide) rc = set_lds(nsf_train1_kernel<KK, HH, true, false, true, true>, lds_launch);
                else if (persist) rc = set_lds(nsf_train1_kernel<KK, HH, true, false, false, true>, lds_launch);
                else rc = set_lds(nsf_train1_kernel<KK, HH, false, false, false, true>, lds_launch);
                if (rc) return rc;
            }
        }
        if (spl) {
        } else if (persist && wide) {
            rc = set_lds(nsf_train1_kernel<KK, HH, true, false, true>, lds_launch);
            if constexpr (lean_persist_inst_v<KK, HH>) { if (lean) rc = set_lds(nsf_train1_kernel<KK, HH, true, true, true>, lds_launch); }
        } else if (persist) {
            rc = set_lds(nsf_train1_kernel<KK, HH, true>, lds_launch);
            if constexpr (lean_persist_inst_v<KK, HH>) { if (lean) rc = set_lds(nsf_train1_kernel<KK, HH, true, true>, lds_launch); }
        } else {
            rc = set_lds(nsf_train1_plain_kernel<KK, HH>, lds_launch);
            if constexpr (lean_plain_v<KK, HH>) { if (lean) rc = set_lds(nsf_train1_plain_kernel<KK, HH, true>, lds_launch); }
        }
        if (rc) return rc;
        // few cliques: their descriptors travel in the kernel arguments (host copy: the plan's, or the single one)
        static_assert(offsetof(Train1Head, shifts) == 32 && sizeof(Train1Head) == 40, "scalar head of the kernel arguments");
        Train1Few few;
        memset(&few, 0, sizeof(few));
        const nfisam_clique* dev = a.cliques;
        const nfisam_clique* host = (a.cliques == nullptr) ? &a.single : a.host_cliques;
        if (host != nullptr && n_cliques <= TRAIN1_KERNARG_CLIQUES) {
            memcpy(few.c, host, sizeof(nfisam_clique) * (size_t)n_cliques);
            dev = nullptr;
        } else if (dev == nullptr) {
            return NFISAM_ERR_ARG;
        }
        const int nch = a.n_chains > 1 ? a.n_chains : 1, ch = a.n_chains > 1 ? a.chain : 0;
        const int octets = (a.groups + 7) / 8;
        const int gz = (octets - ch + nch - 1) / nch;          // octets ch, ch + nch, ...
        if (nch > 255 || ch < 0 || ch >= nch) return NFISAM_ERR_ARG;
        {   // contended launches (more than one block per CU: several waves share a SIMD) divide the Adam update among a group's blocks
            static const char* se = getenv("NFISAM_PERSIST_SPLIT");
            long blocks = 0;
            const nfisam_clique* hc = (a.cliques == nullptr) ? &a.single : a.host_cliques;
            for (int c = 0; c < n_cliques && hc != nullptr; ++c) blocks += (long)hc[c].D * ((hc[c].n + W * TP - 1) / (W * TP));
            a.persist_split = (se != nullptr) ? (se[0] == '1') : (blocks > 256);
        }
        static const int spin_log2 = getenv("NFISAM_PERSIST_SPINS") != nullptr ? atoi(getenv("NFISAM_PERSIST_SPINS")) : 15;           // (~1 us per look: a member that never arrives costs tens of milliseconds, not seconds -- round 4: 22)
        a.persist_spins = spin_log2 < 1 ? 1 : (spin_log2 > 30 ? 30 : spin_log2);
        static const bool drop = getenv("NFISAM_PERSIST_DROP") != nullptr && getenv("NFISAM_PERSIST_DROP")[0] == '1';                  // (test knob)
        static const bool scatter = getenv("NFISAM_PERSIST_SCATTER") != nullptr && getenv("NFISAM_PERSIST_SCATTER")[0] == '1';   // (test knob)
        const int pshifts = a.t_shift | (scatter ? 0x80 : 0) | (drop ? 0x40 : 0) | (a.w_shift << 8) | (ch << 16) | (nch << 24);
        const int shifts = a.t_shift | (a.w_shift << 8) | (ch << 16) | (nch << 24);
        bool launched = false;
        if constexpr (half_kh) {
            if (gz > 0 && spl) {
                if (persist && wide)
                    hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true, false, true, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                                       dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
                else if (persist)
                    hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true, false, false, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                                       dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
                else
                    hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, false, false, false, true>), dim3(8, gx, gz), dim3(64 * W), lds_launch, s, dev, a.panel_map,
                                       a.magic_cliques, a.groups, a.grid_cliques, a.xrows, shifts, a, few);
                launched = true;
            }
        }
        if (spl && !launched && gz > 0) return NFISAM_ERR_ARG;
        if constexpr (lean_persist_inst_v<KK, HH>) {
            if (gz > 0 && persist && lean) {
                if (wide)
                    hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true, true, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                                       dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
                else
                    hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                                       dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
                launched = true;
            }
        }
        if (!launched && gz > 0 && persist && wide) {
            hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true, false, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                               dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
            launched = true;
        }
        if constexpr (lean_plain_v<KK, HH>) {
            if (gz > 0 && !persist && lean) {
                hipLaunchKernelGGL((nsf_train1_plain_kernel<KK, HH, true>), dim3(8, gx, gz), dim3(64 * W), lds_launch, s, dev, a.panel_map,
                                   a.magic_cliques, a.groups, a.grid_cliques, a.xrows, shifts, a, few);
                launched = true;
            }
        }
        if (launched) {
        } else if (gz > 0 && persist)
            hipLaunchKernelGGL((nsf_train1_kernel<KK, HH, true>), scatter ? dim3(gx, 8, gz) : dim3(8, gx, gz), dim3(64 * BW), lds_launch, s,
                               dev, a.panel_map, a.magic_cliques, a.groups, a.grid_cliques, a.xrows, pshifts, a, few);
        else if (gz > 0)
            hipLaunchKernelGGL((nsf_train1_plain_kernel<KK, HH>), dim3(8, gx, gz), dim3(64 * W), lds_launch, s, dev, a.panel_map,
                               a.magic_cliques, a.groups, a.grid_cliques, a.xrows, shifts, a, few);
        HIP_TRY(hipGetLastError());
        return NFISAM_OK;
    }
}

// gradient kernel of one training iteration / VJP: picks the kernel family and block shape for the launch
template <int KK, int HH>
static int unit_train(const TrainArgs& a_in, int n_cliques, int max_n, int max_D, hipStream_t s) {
    if (a_in.tile == TILE2) return unit_train2<KK, HH>(a_in, n_cliques, max_n, max_D, s);
    TrainArgs a = a_in;
    const bool mf = use_mfma_grad(HH);
    const long tiles = (long)((max_n + TILE - 1) / TILE) * n_cliques;
    // L == 1 and no dL/dx requested: the dims of a tile never exchange data.  When the launch is small
    // (latency-bound) every (tile, dim) unit becomes its own single-wave block, so that every wave has
    // a SIMD to itself; with thousands of waves in flight tiles keep their dims together instead.
    const bool independent_dims = (a.L == 1 && a.gx == nullptr);
    if (independent_dims && a.nll_mode && a.gz == nullptr && is_dim_major(n_cliques, max_n, max_D, a.L, TILE, HH))
        return unit_train1<KK, HH>(a, n_cliques, max_n, max_D, s);
    int W = pick_waves(max_D), groups = 1, T = 1;
    if (independent_dims && tiles * max_D <= 1024) { W = 1; groups = max_D; }
    else if (independent_dims) {
        // 4 waves per block: a wave's staging tile is 6.3 KB (24 rows), so 16 waves fit a CU next to the particle tiles.
        // Up to 256 tiles every wave still gets a single unit (dims spread over grid.z); larger launches let a wave
        // loop over its dims and amortise the tile load.
        W = 4;
        const char* e = getenv("NFISAM_BIG_W");
        if (e != nullptr && atoi(e) >= 1 && atoi(e) <= 8) W = atoi(e);
        if (W > max_D) W = max_D;
        if (tiles <= 256) groups = (max_D + W - 1) / W;
        if (groups == 1 && mf && a.tiles_per_block > 1) T = a.tiles_per_block;
    }
    a.tiles_per_block = T;
    a.g_tiles = independent_dims ? 0 : 1;
    const size_t xt = (size_t)(a.L > T ? a.L : T);
    const size_t tile_floats = ((xt + 2 * a.g_tiles) * max_D + 1 + (mf ? (size_t)W * StgRows<KK, HH>::value : 0)) * XS;
    const size_t wfloats = block_weight_floats<KK, HH>(a, max_D, W, groups);
    // LDS copy of the parameters: pays when few waves share a SIMD (nothing hides a cold scalar-cache
    // miss per weight row); it must fit next to the tiles.  Large batches keep the scalar path.
    const int wm = weights_mode();
    const long blocks = ((tiles + T - 1) / T) * groups;
    const bool fits = (tile_floats + wfloats) * sizeof(float) <= 150 * 1024;
    const bool wl = fits && (wm == 1 || (wm == -1 && blocks * W <= 4096 && !(independent_dims && blocks * W > 1024)));
    a.wl_floats = wl ? (int)wfloats : 0;
    const size_t lds = (tile_floats + (wl ? wfloats : 0)) * sizeof(float);
    int rc;
    if constexpr (HH == 8) {
        if (mf) {
            if (wl) rc = launch_train_variant<KK, HH, true, true, 1>(a, n_cliques, max_n, W, groups, lds, s);
            else rc = launch_train_variant<KK, HH, true, false, 1>(a, n_cliques, max_n, W, groups, lds, s);
            if (rc) return rc;
            HIP_TRY(hipGetLastError());
            return NFISAM_OK;
        }
    }
    if (wl) rc = launch_train_variant<KK, HH, false, true, 1>(a, n_cliques, max_n, W, groups, lds, s);
    else rc = launch_train_variant<KK, HH, false, false, 1>(a, n_cliques, max_n, W, groups, lds, s);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return NFISAM_OK;
}

// ---- the unit's table -------------------------------------------------------------------------------------------
#define NSF_OPS_ENTRY(k, h) \
    {k, h, unit_forward<k, h>, unit_inverse<k, h>, unit_walk<k, h>, unit_train<k, h>, unit_prepare<k, h>, pair_kernel_lds<k, h>, unit_pair_map<k, h>, unit_persist_places<k, h>},
static const NsfUnitOps g_unit_ops[] = {NSF_FOR_EACH_KH(NSF_OPS_ENTRY)};

#define NSF_UNIT_FN_(u) nsf_unit_ops_u##u
#define NSF_UNIT_FN(u) NSF_UNIT_FN_(u)
extern "C" const NsfUnitOps* NSF_UNIT_FN(NSF_UNIT)(int K, int H) {
    for (const NsfUnitOps& o : g_unit_ops)
        if (o.K == K && o.H == H) return &o;
    return nullptr;
}

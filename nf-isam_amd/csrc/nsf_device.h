// nsf_device.h — per-lane device math of the autoregressive rational-quadratic spline flow
// for gfx950 (wave64).  One lane = one particle; everything here is straight-line register
// code with compile-time K (bins) and H (hidden width).
//
// What it computes (reference, paths relative to the NF-iSAM root):
//   conditioner  theta_i = W2 tanh(W1 tanh(W0 x[:i] + b0) + b1) + b2      src/flows/flows.py:26-41,82-83
//   knots        softmax -> min-size mix -> cumsum -> [-B,B], ends pinned src/flows/utils.py:85-103
//   derivatives  1e-3 + softplus(logit), boundary logits = const (slope 1) src/flows/utils.py:41-44,94
//   bin search   last knot j with v >= knot_j (last knot bumped 1e-6)      src/flows/utils.py:17-22
//   RQ forward / inverse / log|det|                                        src/flows/utils.py:123-164
//   tails        |v| > B (or NaN): identity, log-det 0                     src/flows/utils.py:31-49
// The backward formulas are hand-derived (DESIGN.md "Backward"); the reference uses autograd.
//
// Weights are wave-uniform: they are read through the scalar cache (address space 4 ->
// s_load_dwordx{8,16}) and used as SGPR operands of v_fma, so they cost neither VGPRs nor
// LDS bandwidth.  The per-particle row x[0..i) lives in LDS, dimension-major [k][64].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace nsf {

typedef const __attribute__((address_space(4))) float cfloat;   // scalar-path (invariant) memory

constexpr float kMinBin = 1e-3f;      // DEFAULT_MIN_BIN_WIDTH / HEIGHT (utils.py:12-13)
constexpr float kMinDeriv = 1e-3f;    // DEFAULT_MIN_DERIVATIVE (utils.py:14)
constexpr float kBoundLogit = 0.53974175453186035f;  // log(exp(1 - 1e-3) - 1)  (utils.py:42)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int TILE = 64;              // particles per wave-tile (one per lane)

__host__ __device__ constexpr int pad4(int v) { return (v + 3) & ~3; }

// ---- kernel-layout offsets (see include/nfisam_hip.h) --------------------------------------
// The 3K-1 spline logits of one dim are stored as two halves of HP floats:
//   half 0 = [ K width logits  | first ND0 derivative logits | 0-pad ]
//   half 1 = [ K height logits | last  ND1 derivative logits | 0-pad ]
// Widths and heights go through the same softmax -> cumsum -> knot pipeline, so in the training kernel
// two neighbouring lanes share one particle, one half each (nsf_split.h); the one-lane-per-particle
// inference kernels address the same layout through iw/ih/idv.
__host__ __device__ constexpr int nd0_of(int K) { return K / 2; }                 // ceil((K-1)/2)
__host__ __device__ constexpr int hp_of(int K) { return pad4(K + nd0_of(K)); }
__host__ __device__ constexpr int pop_of(int K) { return 2 * hp_of(K); }
__host__ __device__ constexpr int out_col(int K, int o) {     // reference output index o in [0,3K-1) -> column
    return o < K ? o : (o < 2 * K ? hp_of(K) + (o - K)
                                   : ((o - 2 * K) < nd0_of(K) ? K + (o - 2 * K) : hp_of(K) + K + (o - 2 * K - nd0_of(K))));
}
template <int K, int H>
struct Layout {
    static constexpr int Po = 3 * K - 1;
    static constexpr int ND0 = nd0_of(K), ND1 = K - 1 - ND0;
    static constexpr int HP = hp_of(K);
    static constexpr int PoP = pop_of(K);
    __host__ __device__ static constexpr int iw(int j) { return j; }                // width logit j
    __host__ __device__ static constexpr int ih(int j) { return HP + j; }           // height logit j
    __host__ __device__ static constexpr int idv(int j) { return j < ND0 ? K + j : HP + K + (j - ND0); }  // interior knot j+1
    static constexpr int kFixed = H + H * H + H + H * PoP + PoP;   // block size without W0t
    __host__ __device__ static constexpr int block(int i) { return i * H + kFixed; }
    __host__ __device__ static constexpr int off(int i) {          // offset of dim i's block, i >= 1
        return PoP + (i - 1) * kFixed + H * ((i - 1) * i / 2);
    }
    __host__ __device__ static constexpr int count(int D) { return off(D); }
    // offsets inside dim i's block
    __host__ __device__ static constexpr int oW0(int) { return 0; }
    __host__ __device__ static constexpr int ob0(int i) { return i * H; }
    __host__ __device__ static constexpr int oW1(int i) { return i * H + H; }
    __host__ __device__ static constexpr int ob1(int i) { return i * H + H + H * H; }
    __host__ __device__ static constexpr int oW2(int i) { return i * H + H + H * H + H; }
    __host__ __device__ static constexpr int ob2(int i) { return i * H + H + H * H + H + H * PoP; }
};

// ---- fast scalar math (1-ulp hardware transcendentals) --------------------------------------
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * kLn2; }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ftanh(float x) {
    // 1 - 2/(1+e^{2x}); saturates correctly for |x| large (e -> inf gives 1, e -> 0 gives -1)
    const float e = __builtin_amdgcn_exp2f(x * (2.0f * kLog2e));
    return 1.0f - 2.0f * frcp(1.0f + e);
}
__device__ __forceinline__ float fsoftplus(float x) {   // max(x,0) + log(1 + e^{-|x|})
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * kLog2e);
    return fmaxf(x, 0.0f) + flog(1.0f + e);
}
__device__ __forceinline__ float fsigmoid(float x) { return frcp(1.0f + fexp(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// cross-lane reduce-scatter: on return lane l holds the wave total of input v[l & (N-1)].
// log2(N) exchange steps move N-1 values in total (vs 6 per value for a plain wave reduction).
template <int N>
__device__ __forceinline__ float butterfly(float (&v)[N], int lane) {
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int t = 0; t < half; ++t) {
            const float lo = v[t], hi = v[t + half];   // load first: keeps v[] in registers (no select-of-address)
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            v[t] = keep + __shfl_xor(send, half, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int off = N; off < 64; off <<= 1) r += __shfl_xor(r, off, 64);
    return r;
}


// ---- Adam (torch.optim.Adam as the reference drives it, NFiSAM.py:425, 476-478) ---------------------------------
// One copy of the arithmetic for the stand-alone Adam kernel and the update fused into the dim-major training
// kernel's tail: explicit round-to-nearest operations, so that -ffp-contract cannot fuse them differently in the two
// places and both produce the same bits from the same gradient sum.
struct AdamCoef { float b1, b2, step_size, inv_bc2s, eps, inv_n; };
// a product that must not be fused into the addition that consumes it: the empty asm makes it opaque
// (`#pragma clang fp contract(off)` is not honoured once the function is inlined into a -ffp-contract=fast kernel)
__device__ __forceinline__ float rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
// 1 - beta^t from ln(beta): explicit operations only (a library expm1f inlined into two kernels came out an ulp apart),
// relative error < 5e-7: the series of -expm1(x) while |x| < 1/4, else 1 - 2^(x log2 e) on the hardware exp2
__device__ __forceinline__ float one_minus_pow(float log_b, int t) {
    const float x = rounded((float)t * log_b);
    if (x > -0.25f) {
        float p = rounded(x * (1.0f / 5040.0f)) + (1.0f / 720.0f);
        p = rounded(p * x) + (1.0f / 120.0f);
        p = rounded(p * x) + (1.0f / 24.0f);
        p = rounded(p * x) + (1.0f / 6.0f);
        p = rounded(p * x) + 0.5f;
        p = rounded(p * x) + 1.0f;
        return -rounded(p * x);
    }
    return 1.0f - __builtin_amdgcn_exp2f(rounded(x * 1.4426950408889634f));
}
__device__ __forceinline__ AdamCoef adam_coef(float lr, float b1, float b2, float eps, float log_b1, float log_b2, int t, int n) {
    AdamCoef k;
    const float bc1 = one_minus_pow(log_b1, t), bc2 = one_minus_pow(log_b2, t);
    k.b1 = b1; k.b2 = b2; k.eps = eps;
    k.step_size = lr / bc1;
    k.inv_bc2s = 1.0f / sqrtf(bc2);
    k.inv_n = 1.0f / (float)n;
    return k;
}
__device__ __forceinline__ void adam_update(const AdamCoef& k, float gsum, float& m, float& v, float& theta) {
    const float g = rounded(gsum * k.inv_n);
    const float c1 = 1.0f - k.b1, c2 = 1.0f - k.b2;
    const float m1 = rounded(k.b1 * m), m2 = rounded(c1 * g);
    m = m1 + m2;
    const float v1 = rounded(k.b2 * v), v2 = rounded(rounded(c2 * g) * g);
    v = v1 + v2;
    const float sq = rounded(sqrtf(v) * k.inv_bc2s);
    const float denom = sq + k.eps;
    const float num = rounded(k.step_size * m);
    const float upd = rounded(num / denom);
    theta = theta - upd;
}

// ---- weight rows ------------------------------------------------------------------------------
// A row of N (multiple of 4) wave-uniform weights.  Scalar-cache path: plain indexing (the
// compiler merges into s_load_dwordx{4,8,16}); LDS path: explicit 16-byte broadcast reads
// (ds_read_b128), every row of the kernel layout being 16-byte aligned by construction.
typedef __attribute__((ext_vector_type(4))) float wf4;
template <int N>
__device__ __forceinline__ void load_row(cfloat* row, float (&w)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) w[j] = row[j];
}
template <int N>
__device__ __forceinline__ void load_row(const float* row, float (&w)[N]) {
    static_assert(N % 4 == 0, "rows are padded to multiples of 4");
#pragma unroll
    for (int j = 0; j < N; j += 4) {
        const wf4 v = *(const wf4*)(row + j);
        w[j] = v.x; w[j + 1] = v.y; w[j + 2] = v.z; w[j + 3] = v.w;
    }
}

// The first U of the N floats of a row; the tail group is fetched with a narrower load, so that no loaded register
// is dead: a partly dead 16-byte load lets the register allocator overlap its destination with the next load's,
// and that write-after-write pair forces a full s_waitcnt lgkmcnt(0) between the two loads.
template <int N, int U>
__device__ __forceinline__ void load_row_used(const float* row, float (&w)[N]) {
    static_assert(U <= N && U > 0, "used prefix");
#pragma unroll
    for (int j = 0; j + 4 <= U; j += 4) {
        const wf4 v = *(const wf4*)(row + j);
        w[j] = v.x; w[j + 1] = v.y; w[j + 2] = v.z; w[j + 3] = v.w;
    }
    constexpr int R = U & 3, J = U - R;
    if constexpr (R == 1) {
        w[J] = row[J];
    } else if constexpr (R == 2) {
        const float2 v = *(const float2*)(row + J);
        w[J] = v.x; w[J + 1] = v.y;
    } else if constexpr (R == 3) {
        const float2 v = *(const float2*)(row + J);
        w[J] = v.x; w[J + 1] = v.y; w[J + 2] = row[J + 2];
    }
#pragma unroll
    for (int j = U; j < N; ++j) w[j] = 0.0f;
}

// Scalar-path weight rows live in SGPRs (102 per wave).  Left alone, hipcc hoists the s_loads of every later row of a
// fully unrolled loop above the FMAs of the current rows and then SPILLS the loaded weights to VGPR lanes
// (v_writelane / v_readlane: two extra VALU instructions per weight, ~1000 per unit).  A scheduling fence after every
// group of rows keeps at most ~64 weight SGPRs in flight; the fence emits no instruction.
template <typename WP>
__device__ __forceinline__ void row_group_fence() {
    if constexpr (std::is_same<WP, cfloat*>::value) __builtin_amdgcn_sched_barrier(0);
}

// The backward pass re-reads W2 / W1 that the forward pass of the same unit has read.  Scalar-path loads are
// invariant loads, so hipcc would CSE the two and keep ~330 weight SGPRs alive from the forward conditioner across the
// whole spline to the backward conditioner -- i.e. spill every weight to a VGPR lane (v_writelane) and fetch it back
// (v_readlane) in front of every FMA: two extra VALU instructions per weight.  Passing the pointer through an empty
// asm hides the equality; the re-load hits the scalar cache.
__device__ __forceinline__ cfloat* reload_ptr(cfloat* p) {
    const uint64_t v = (uint64_t)p;                     // wave-uniform by construction: make that explicit for the "s" constraint
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (cfloat*)(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ const float* reload_ptr(const float* p) { return p; }

// ---- scalar-path GEMV with the NEXT weight chunk in flight ----------------------------------------------------
// A scalar load returns out of order with the wave's other LSU traffic, so its consumer waits for `lgkmcnt(0)`:
// load -> wait -> FMAs -> load -> wait ... exposes one scalar-cache round trip (~200-300 cycles) per chunk, and the
// conditioner is ~10 chunks per direction.  Here chunk c+1 is requested right after chunk c has ARRIVED and before it
// is consumed, so its round trip runs under the FMAs of chunk c.  Chunks are 32 consecutive floats (two
// s_load_dwordx16: 64 SGPRs for the pair in flight + in use, of 102); `arrived()` is an empty asm that reads a chunk,
// which makes the compiler place the wait there, and the scheduling fences keep the order request / consume.
template <int N>
__device__ __forceinline__ void chunk_arrived(const float (&w)[N]) {
    asm volatile("" ::"s"(w[0]), "s"(w[N - 1]));
}
// The accumulators pass through an empty asm at every chunk boundary: without it the SLP vectoriser fuses the whole
// eight-chunk FMA chain into one tree and emits it after the LAST load, which keeps every chunk alive (and spilled).
template <int N>
__device__ __forceinline__ void pin_rows(float (&acc)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) asm volatile("" : "+v"(acc[j]));
}
// 32 consecutive floats as two VOLATILE 64-byte scalar loads: a volatile access is neither split, merged with its
// neighbours, hoisted nor shared with the other pass's load of the same row, so it stays where the pipeline puts it
typedef __attribute__((ext_vector_type(16))) float wf16;
__device__ __forceinline__ void load_chunk(cfloat* p, float (&w)[32]) {
    typedef const volatile __attribute__((address_space(4))) wf16* vp;
    const wf16 lo = *(vp)p, hi = *(vp)(p + 16);
#pragma unroll
    for (int j = 0; j < 16; ++j) { w[j] = lo[j]; w[16 + j] = hi[j]; }
}
// acc[j] += sum_k W[k][j] * h[k]   for a contiguous row-major W[ROWS][N] in scalar-path memory, N * G = 32 floats per chunk
template <int N, int ROWS>
__device__ __forceinline__ void scalar_gemv_cols(cfloat* W, const float (&h)[ROWS], float (&acc)[N]) {
    constexpr int G = (N >= 32) ? 1 : 32 / N, CH = G * N, NC = ROWS / G;
    static_assert(ROWS % G == 0 && N <= 32 && 32 % N == 0, "chunking");
    float wa[CH], wb[CH];
    load_chunk(W, wa);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float(&cur)[CH] = (c & 1) ? wb : wa;
        float(&nxt)[CH] = (c & 1) ? wa : wb;
        chunk_arrived<CH>(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) load_chunk(W + (c + 1) * CH, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < G; ++r)
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] = __builtin_fmaf(cur[r * N + j], h[c * G + r], acc[j]);
        pin_rows<N>(acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// out[k] = sum_j W[k][j] * g[j]   (the transposed use of the same rows in the backward pass); four partial sums per row
template <int N, int ROWS>
__device__ __forceinline__ void scalar_gemv_rows(cfloat* W, const float (&g)[N], float (&out)[ROWS]) {
    constexpr int G = (N >= 32) ? 1 : 32 / N, CH = G * N, NC = ROWS / G;
    static_assert(ROWS % G == 0 && N <= 32 && 32 % N == 0, "chunking");
    float wa[CH], wb[CH];
    load_chunk(W, wa);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float(&cur)[CH] = (c & 1) ? wb : wa;
        float(&nxt)[CH] = (c & 1) ? wa : wb;
        chunk_arrived<CH>(cur);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) load_chunk(W + (c + 1) * CH, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < G; ++r) {
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
            for (int j = 0; j + 3 < N; j += 4) {
                p0 = __builtin_fmaf(cur[r * N + j], g[j], p0);
                p1 = __builtin_fmaf(cur[r * N + j + 1], g[j + 1], p1);
                p2 = __builtin_fmaf(cur[r * N + j + 2], g[j + 2], p2);
                p3 = __builtin_fmaf(cur[r * N + j + 3], g[j + 3], p3);
            }
            out[c * G + r] = (p0 + p1) + (p2 + p3);
            asm volatile("" : "+v"(out[c * G + r]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- conditioner ----------------------------------------------------------------------------
// xs: LDS, dimension-major with row stride `xstride`: xs[k * xstride + lane] = x_k of this lane's particle.
// WP = weight pointer type: `cfloat*` (scalar-cache path, SGPR operands) or `const float*` into an
// LDS copy of the parameters (broadcast ds_read_b128; used when one wave per SIMD must not expose
// a cold scalar-cache miss per weight row).
template <int K, int H, typename WP>
__device__ __forceinline__ void cond_hidden(WP blk, int i, const float* xs, int xstride, int lane,
                                            float (&h1)[H], float (&h2)[H]) {
    using LY = Layout<K, H>;
    float a[H], wr[H];
    load_row<H>(blk + LY::ob0(i), a);
    WP W0 = blk;
    // input layer: the only loop whose trip count depends on the dim.  Four (x_k, weight row) pairs are
    // fetched together so that their LDS / scalar-cache latencies overlap; rows k >= i are still inside
    // this dim's block (they alias b0 / W1) and are multiplied by 0.
    for (int k = 0; k < i; k += 4) {
        float xk[4], wq[4][H];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xk[u] = (k + u < i) ? xs[(k + u) * xstride + lane] : 0.0f;
            load_row<H>(W0 + (k + u) * H, wq[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < H; ++j) a[j] = __builtin_fmaf(wq[u][j], xk[u], a[j]);
        }
        row_group_fence<WP>();
    }
#pragma unroll
    for (int j = 0; j < H; ++j) h1[j] = ftanh(a[j]);
    WP W1 = blk + LY::oW1(i);
    load_row<H>(blk + LY::ob1(i), a);
    if constexpr (std::is_same<WP, cfloat*>::value && (H == 8 || H == 16)) {
        scalar_gemv_cols<H, H>(W1, h1, a);
    } else {
#pragma unroll
        for (int k = 0; k < H; ++k) {
            load_row<H>(W1 + k * H, wr);
#pragma unroll
            for (int j = 0; j < H; ++j) a[j] = __builtin_fmaf(wr[j], h1[k], a[j]);
            if ((k & 7) == 7) row_group_fence<WP>();
        }
    }
#pragma unroll
    for (int j = 0; j < H; ++j) h2[j] = ftanh(a[j]);
}

template <int K, int H, typename WP>
__device__ __forceinline__ void cond_theta(WP blk, int i, const float (&h2)[H],
                                           float (&th)[Layout<K, H>::PoP]) {
    using LY = Layout<K, H>;
    WP W2 = blk + LY::oW2(i);
    if constexpr (std::is_same<WP, cfloat*>::value && LY::PoP == 32) {
        // b2 is row H of the same block ([H + 1][PoP] contiguous): the bias rides the pipeline as one more chunk
        float hx[H + 1];
#pragma unroll
        for (int k = 0; k < H; ++k) hx[k] = h2[k];
        hx[H] = 1.0f;
#pragma unroll
        for (int o = 0; o < LY::PoP; ++o) th[o] = 0.0f;
        scalar_gemv_cols<LY::PoP, H + 1>(W2, hx, th);
    } else {
        load_row<LY::PoP>(blk + LY::ob2(i), th);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            float wr[LY::PoP];
            load_row<LY::PoP>(W2 + k * LY::PoP, wr);
#pragma unroll
            for (int o = 0; o < LY::PoP; ++o) th[o] = __builtin_fmaf(wr[o], h2[k], th[o]);
            if (k & 1) row_group_fence<WP>();
        }
    }
}

// ---- spline ----------------------------------------------------------------------------------
template <int K>
struct Spline {
    float pw[K], ph[K];               // softmax probabilities of widths / heights
    float Xk, dx, Yk, dy, d0, d1;     // selected bin: left knots, sizes, end derivatives
    float ud0, ud1;                   // derivative logits at knots k, k+1
    float t;                          // position inside the bin
    int k;
    bool inside;
};

// Rational-quadratic map inside one bin (utils.py:123-164).  Forward: v is x, returns z and
// log dz/dx.  Inverse: v is z, solves the quadratic with the numerically stable root
// 2c / (-b - sqrt(b^2 - 4ac)) (utils.py:126-136) and returns x and -log dz/dx.
template <bool INV>
__device__ __forceinline__ void rq_math(float vs, float Xk, float dx, float Yk, float dy, float d0, float d1,
                                        float& t_out, float& out, float& lad) {
    const float idx = frcp(dx);
    const float s = dy * idx, sig = d0 + d1 - 2.0f * s;
    float t;
    if (INV) {
        const float dl = vs - Yk;
        const float a = dl * sig + dy * (s - d0);
        const float b = dy * d0 - dl * sig;
        const float c = -s * dl;
        const float disc = fmaxf(b * b - 4.0f * a * c, 0.0f);
        t = (2.0f * c) * frcp(-b - __builtin_sqrtf(disc));
        out = t * dx + Xk;
    } else {
        t = (vs - Xk) * idx;
    }
    t_out = t;
    const float q = t * (1.0f - t), omt = 1.0f - t;
    const float den = s + sig * q;
    const float M = d1 * t * t + 2.0f * s * q + d0 * omt * omt;
    const float l = flog(s * s * M) - 2.0f * flog(den);
    if (INV) {
        lad = -l;
    } else {
        out = Yk + dy * (s * t * t + d0 * q) * frcp(den);
        lad = l;
    }
}

template <int K, int PoP, bool INV>
__device__ __forceinline__ void spline_eval(float v, const float (&th)[PoP], float B, Spline<K>& S,
                                            float& out, float& lad) {
    using LY = Layout<K, 8>;          // only the output-column accessors are used (independent of H)
    static_assert(PoP == LY::PoP, "theta rows follow the kernel layout");
    S.inside = (v >= -B) && (v <= B);             // false for NaN (utils.py:31)
    const float vs = S.inside ? v : 0.0f;
    float mw = th[LY::iw(0)], mh = th[LY::ih(0)];
#pragma unroll
    for (int j = 1; j < K; ++j) { mw = fmaxf(mw, th[LY::iw(j)]); mh = fmaxf(mh, th[LY::ih(j)]); }
    float sw = 0.0f, sh = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        S.pw[j] = fexp(th[LY::iw(j)] - mw); sw += S.pw[j];
        S.ph[j] = fexp(th[LY::ih(j)] - mh); sh += S.ph[j];
    }
    const float iw = frcp(sw), ih = frcp(sh);
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    float cx = 0.0f, cy = 0.0f, Xl = -B, Yl = -B;
    S.k = 0; S.Xk = -B; S.Yk = -B; S.dx = 1.0f; S.dy = 1.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        S.pw[j] *= iw; S.ph[j] *= ih;
        cx += kMinBin + mix * S.pw[j];
        cy += kMinBin + mix * S.ph[j];
        const float Xr = (j == K - 1) ? B : twoB * cx - B;     // last knot pinned (utils.py:90-91)
        const float Yr = (j == K - 1) ? B : twoB * cy - B;
        const bool sel = INV ? (vs >= Yl) : (vs >= Xl);         // monotone knots: last true wins
        if (sel) { S.k = j; S.Xk = Xl; S.dx = Xr - Xl; S.Yk = Yl; S.dy = Yr - Yl; }
        Xl = Xr; Yl = Yr;
    }
    S.ud0 = kBoundLogit; S.ud1 = kBoundLogit;
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        const float dj = th[LY::idv(j)];
        if (S.k == j + 1) S.ud0 = dj;
        if (S.k == j) S.ud1 = dj;
    }
    S.d0 = kMinDeriv + fsoftplus(S.ud0);
    S.d1 = kMinDeriv + fsoftplus(S.ud1);
    rq_math<INV>(vs, S.Xk, S.dx, S.Yk, S.dy, S.d0, S.d1, S.t, out, lad);
    if (!S.inside) { out = v; lad = 0.0f; }
}

// Backward of the forward spline.  Upstream gz = dL/dz, gl = dL/dlogdet.  Writes gth[0..Po)
// (gth[Po..PoP) = 0) and returns dL/dx through the spline's own argument.
template <int K, int PoP>
__device__ __forceinline__ float spline_backward(const Spline<K>& S, float B, float gz, float gl,
                                                 float (&gth)[PoP]) {
    using LY = Layout<K, 8>;
    static_assert(PoP == LY::PoP, "theta rows follow the kernel layout");
#pragma unroll
    for (int o = 0; o < PoP; ++o) gth[o] = 0.0f;      // pads stay 0
    const int k = S.k;
    const float w = S.dx, h = S.dy, d0 = S.d0, d1 = S.d1, t = S.t;
    const float iw = frcp(w);
    const float s = h * iw, sig = d0 + d1 - 2.0f * s, q = t * (1.0f - t), omt = 1.0f - t, o2t = 1.0f - 2.0f * t;
    const float N = s * t * t + d0 * q, den = s + sig * q;
    const float iden = frcp(den), u = N * iden, iden2 = iden * iden;
    const float u_t = ((2.0f * s * t + d0 * o2t) * den - N * sig * o2t) * iden2;
    const float u_s = (t * t * den - N * (1.0f - 2.0f * q)) * iden2;
    const float u_d0 = q * (den - N) * iden2;
    const float u_d1 = -N * q * iden2;
    const float M = d1 * t * t + 2.0f * s * q + d0 * omt * omt;
    const float iM = frcp(M);
    const float M_t = 2.0f * d1 * t + 2.0f * s * o2t - 2.0f * d0 * omt;
    const float ld_t = M_t * iM - 2.0f * sig * o2t * iden;
    const float ld_s = 2.0f * frcp(s) + 2.0f * q * iM - 2.0f * (1.0f - 2.0f * q) * iden;
    const float ld_d0 = omt * omt * iM - 2.0f * q * iden;
    const float ld_d1 = t * t * iM - 2.0f * q * iden;
    const float gzh = gz * h;
    const float G_t = gzh * u_t + gl * ld_t;
    const float G_s = gzh * u_s + gl * ld_s;
    const float G_d0 = gzh * u_d0 + gl * ld_d0;
    const float G_d1 = gzh * u_d1 + gl * ld_d1;
    const float g_x = G_t * iw;
    const float g_w = -(G_t * t + G_s * s) * iw;          // d/d(bin width)  at fixed left knot
    const float g_h = gz * u + G_s * iw;                  // d/d(bin height) at fixed left knot
    const bool lo = (k >= 1), hi = (k + 1 <= K - 1);      // end knots are pinned: no gradient
    const float gXk = lo ? (-g_x - g_w) : 0.0f, gXk1 = hi ? g_w : 0.0f;
    const float gYk = lo ? (gz - g_h) : 0.0f, gYk1 = hi ? g_h : 0.0f;
    const float scale = 2.0f * B * (1.0f - kMinBin * (float)K);
    const float cw1 = scale * (gXk + gXk1), cw2 = scale * gXk1;
    const float ch1 = scale * (gYk + gYk1), ch2 = scale * gYk1;
    float dotw = 0.0f, doth = 0.0f;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const float cw = (m < k) ? cw1 : ((m == k) ? cw2 : 0.0f);
        const float ch = (m < k) ? ch1 : ((m == k) ? ch2 : 0.0f);
        dotw = __builtin_fmaf(S.pw[m], cw, dotw);
        doth = __builtin_fmaf(S.ph[m], ch, doth);
    }
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const float cw = (m < k) ? cw1 : ((m == k) ? cw2 : 0.0f);
        const float ch = (m < k) ? ch1 : ((m == k) ? ch2 : 0.0f);
        gth[LY::iw(m)] = S.pw[m] * (cw - dotw);
        gth[LY::ih(m)] = S.ph[m] * (ch - doth);
    }
    const float gd0 = G_d0 * fsigmoid(S.ud0), gd1 = G_d1 * fsigmoid(S.ud1);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        float v = 0.0f;
        if (k == j + 1) v = gd0;
        if (k == j) v = gd1;
        gth[LY::idv(j)] = v;
    }
    if (!S.inside) {
#pragma unroll
        for (int o = 0; o < PoP; ++o) gth[o] = 0.0f;
        return gz;
    }
    return g_x;
}


// ---- the spline of the NLL training kernels, written for VALU instruction count --------------------------------------
// Same mathematics as spline_eval<..., false> + spline_backward (utils.py:85-164 and its hand-derived backward), arranged
// so that a wave issues ~500 instead of ~610 VALU slots per particle:
//  * the softmax terms stay UNNORMALISED (e_m = exp(th_m - max)); knot j = -B + j 2B minbin + c E_j with E_j the running
//    sum of e (the last of them is the softmax denominator: no separate sum pass) and c = 2B (1 - K minbin) / E_K;
//  * the bin is found by K - 1 comparisons whose results stay in scalar registers (sel[j] = v >= knot_j, monotone) and
//    select the two knots of the bin on either axis and its two derivative logits; the backward pass re-uses them as
//    "m < k / m == k" masks instead of comparing a bin index again;
//  * sum_m p_m c_m of the softmax backward needs only P(m < k) and p_k, both known from the selected knots;
//  * outside [-B, B] the upstream gradients are zeroed once instead of every output.
// lean forward + backward of the spline for the NLL training kernel
template <int K>
struct SplineT {
    float ew[K], eh[K];             // UNNORMALISED softmax terms exp(th - max)
    float iw, ih;                   // 1 / sum
    float Xk, dx, Yk, dy, d0, d1, ud0, ud1, t;
    bool sel[K];                    // sel[j] = v >= X_j (j >= 1; monotone: true up to the bin)
    bool inside;
};
template <int K, int PoP>
__device__ __forceinline__ void spline_train_fwd(float v, const float (&th)[PoP], float B, SplineT<K>& S, float& z, float& lad) {
    using LY = Layout<K, 8>;
    S.inside = (v >= -B) && (v <= B);
    const float vs = S.inside ? v : 0.0f;
    float mw = th[LY::iw(0)], mh = th[LY::ih(0)];
#pragma unroll
    for (int j = 1; j < K; ++j) { mw = fmaxf(mw, th[LY::iw(j)]); mh = fmaxf(mh, th[LY::ih(j)]); }
    const float nmw = -mw * kLog2e, nmh = -mh * kLog2e;
    float Ew[K + 1], Eh[K + 1];     // unnormalised cumulative sums: E[j] = sum_{m<j} e_m
    Ew[0] = 0.0f; Eh[0] = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        S.ew[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(th[LY::iw(j)], kLog2e, nmw));
        S.eh[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(th[LY::ih(j)], kLog2e, nmh));
        Ew[j + 1] = Ew[j] + S.ew[j];
        Eh[j + 1] = Eh[j] + S.eh[j];
    }
    S.iw = frcp(Ew[K]); S.ih = frcp(Eh[K]);
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    const float cw = twoB * mix * S.iw, ch = twoB * mix * S.ih;
    // knot j = -B + j * 2B * minbin + c * E[j]   (j = 1..K-1; knot 0 = -B, knot K = B pinned: utils.py:85-92)
    float Xk = -B, Xk1 = B, Yk = -B, Yk1 = B;
    float X1 = __builtin_fmaf(cw, Ew[1], twoB * kMinBin - B), Y1 = __builtin_fmaf(ch, Eh[1], twoB * kMinBin - B);
    if (K > 1) { Xk1 = X1; Yk1 = Y1; }
    S.ud0 = kBoundLogit; S.ud1 = (K > 1) ? th[LY::idv(0)] : kBoundLogit;
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const float Xj = X1, Yj = Y1;
        float Xn = B, Yn = B;
        if (j + 1 < K) {
            const float cj = (float)(j + 1) * twoB * kMinBin - B;
            Xn = __builtin_fmaf(cw, Ew[j + 1], cj);
            Yn = __builtin_fmaf(ch, Eh[j + 1], cj);
        }
        const bool s = vs >= Xj;
        S.sel[j] = s;
        Xk = s ? Xj : Xk; Xk1 = s ? Xn : Xk1;
        Yk = s ? Yj : Yk; Yk1 = s ? Yn : Yk1;
        S.ud0 = s ? th[LY::idv(j - 1)] : S.ud0;
        S.ud1 = s ? ((j + 1 < K) ? th[LY::idv(j < K - 1 ? j : 0)] : kBoundLogit) : S.ud1;
        X1 = Xn; Y1 = Yn;
    }
    S.Xk = Xk; S.dx = Xk1 - Xk; S.Yk = Yk; S.dy = Yk1 - Yk;
    S.d0 = kMinDeriv + fsoftplus(S.ud0);
    S.d1 = kMinDeriv + fsoftplus(S.ud1);
    rq_math<false>(vs, S.Xk, S.dx, S.Yk, S.dy, S.d0, S.d1, S.t, z, lad);
    if (!S.inside) { z = v; lad = 0.0f; }
}
// gth = dL/dtheta for upstream gz = dL/dz, gl = dL/dlogdet (pads of the layout are written 0); returns dL/dx through the
// spline's own argument (outside the interval the spline is the identity: gz itself) -- multi-layer kernels only
template <int K, int PoP>
__device__ __forceinline__ float spline_train_bwd(const SplineT<K>& S, float B, float gz_in, float gl_in, float (&gth)[PoP]) {
    using LY = Layout<K, 8>;
    const float gz = S.inside ? gz_in : 0.0f, gl = S.inside ? gl_in : 0.0f;     // outside the interval: identity, no parameter gradient
    const float w = S.dx, h = S.dy, d0 = S.d0, d1 = S.d1, t = S.t;
    const float iw = frcp(w);
    const float s = h * iw, sig = d0 + d1 - 2.0f * s, q = t * (1.0f - t), omt = 1.0f - t, o2t = 1.0f - 2.0f * t;
    const float N = s * t * t + d0 * q, den = s + sig * q;
    const float iden = frcp(den), u = N * iden, iden2 = iden * iden;
    const float u_t = ((2.0f * s * t + d0 * o2t) * den - N * sig * o2t) * iden2;
    const float u_s = (t * t * den - N * (1.0f - 2.0f * q)) * iden2;
    const float u_d0 = q * (den - N) * iden2;
    const float u_d1 = -N * q * iden2;
    const float M = d1 * t * t + 2.0f * s * q + d0 * omt * omt;
    const float iM = frcp(M);
    const float M_t = 2.0f * d1 * t + 2.0f * s * o2t - 2.0f * d0 * omt;
    const float ld_t = M_t * iM - 2.0f * sig * o2t * iden;
    const float ld_s = 2.0f * frcp(s) + 2.0f * q * iM - 2.0f * (1.0f - 2.0f * q) * iden;
    const float ld_d0 = omt * omt * iM - 2.0f * q * iden;
    const float ld_d1 = t * t * iM - 2.0f * q * iden;
    const float gzh = gz * h;
    const float G_t = gzh * u_t + gl * ld_t;
    const float G_s = gzh * u_s + gl * ld_s;
    const float G_d0 = gzh * u_d0 + gl * ld_d0;
    const float G_d1 = gzh * u_d1 + gl * ld_d1;
    const float g_x = G_t * iw;
    const float g_w = -(G_t * t + G_s * s) * iw;
    const float g_h = gz * u + G_s * iw;
    const bool lo = (K > 1) ? S.sel[K > 1 ? 1 : 0] : false;            // k >= 1
    const bool hi = (K > 1) ? !S.sel[K - 1] : false;                    // k + 1 <= K - 1
    const float gXk = lo ? (-g_x - g_w) : 0.0f, gXk1 = hi ? g_w : 0.0f;
    const float gYk = lo ? (gz - g_h) : 0.0f, gYk1 = hi ? g_h : 0.0f;
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    const float scale = twoB * mix;
    const float cw1 = scale * (gXk + gXk1), cw2 = scale * gXk1;
    const float ch1 = scale * (gYk + gYk1), ch2 = scale * gYk1;
    // softmax backward: gth_m = p_m (c_m - sum_j p_j c_j), c_m = c1 (m < k), c2 (m = k), 0 (m > k).  The sum needs only
    // P(<k) = sum_{m<k} p_m and p_k, both known from the selected knots: X_k + B = 2B (k minbin + mix P(<k)), dx = 2B (minbin + mix p_k)
    int kc = 0;
#pragma unroll
    for (int j = 1; j < K; ++j) kc += S.sel[j] ? 1 : 0;
    const float kf = (float)kc, tk = twoB * kMinBin, r = frcp(scale);
    const float Pw = __builtin_fmaf(-kf, tk, S.Xk + B), Ph = __builtin_fmaf(-kf, tk, S.Yk + B);
    const float dotw = r * __builtin_fmaf(cw1, Pw, cw2 * (w - tk)), doth = r * __builtin_fmaf(ch1, Ph, ch2 * (h - tk));
    const float aw1 = S.iw * (cw1 - dotw), aw2 = S.iw * (cw2 - dotw), aw3 = -S.iw * dotw;
    const float ah1 = S.ih * (ch1 - doth), ah2 = S.ih * (ch2 - doth), ah3 = -S.ih * doth;
#pragma unroll
    for (int o = 0; o < PoP; ++o) gth[o] = 0.0f;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const bool ge = (m == 0) ? true : S.sel[m];                  // k >= m
        const bool gt = (m + 1 < K) ? S.sel[m + 1] : false;          // k >  m
        gth[LY::iw(m)] = S.ew[m] * (gt ? aw1 : (ge ? aw2 : aw3));
        gth[LY::ih(m)] = S.eh[m] * (gt ? ah1 : (ge ? ah2 : ah3));
    }
    const float gd0 = G_d0 * fsigmoid(S.ud0), gd1 = G_d1 * fsigmoid(S.ud1);
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {     // interior knot j + 1: k == j + 1 -> gd0, k == j -> gd1
        const bool gej = (j == 0) ? true : S.sel[j];
        const bool gej1 = S.sel[j + 1];
        const bool gej2 = (j + 2 < K) ? S.sel[j + 2] : false;
        gth[LY::idv(j)] = gej2 ? 0.0f : (gej1 ? gd0 : (gej ? gd1 : 0.0f));
    }
    return S.inside ? g_x : gz_in;
}

}  // namespace nsf

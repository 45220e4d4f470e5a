// nsf_half.h -- "two lanes per particle" ON the dim-major MFMA training kernel (round 6; gfx950, wave64).
//
// A lone wave per SIMD issues one vector instruction per ~4 cycles however independent its instructions are
// (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'), so the latency of a (clique, dim, tile) unit is the length of its
// per-LANE program.  One fit of a real NF-iSAM run (n = 2000, D = 15: 480 waves on 1024 SIMDs) is exactly that regime, and
// half the chip idles.  Here a wave covers 32 particles and the lanes p and p + 32 share particle p:
//   lanes  0-31 ("low half")   hidden units 0 .. H/2-1,  the first HP theta columns = [K width logits  | first ND0 slope logits]
//   lanes 32-63 ("high half")  hidden units H/2 .. H-1,  the last  HP theta columns = [K height logits | last  ND1 slope logits]
// The 4x4x1 MFMA blocks are four lanes wide, so a block sits inside one half and the halves differ only in the panel rows
// their lanes read (as the two dims of a wave do in nsf_train3_kernel).  Widths and heights go through the same
// softmax -> cumulative sum -> knot pipeline (nsf_split.h), the scalar algebra of the rational-quadratic bin is computed by
// both lanes.  What crosses between the halves goes through v_permlane32_swap_b32 (one instruction exchanges the upper half
// of one register with the lower half of another):
//   all-gather      hgather(v, lo, hi): every lane gets the low lane's and the high lane's v        (hidden activations, the
//                                                                                                    selected knots and slope logits)
//   reduce-scatter  hswap(a, b); a + b: the low lanes get the total of a, the high lanes of b       (dL/dh of the backward pass)
// The weight-gradient GEMMs contract over the wave's 32 particles: half the 16x16x4 MFMA steps of a 64-particle tile, and both
// 16-row tiles of dL/dtheta are staged at once (the halves write side by side).
// Reference: src/flows/flows.py:26-41,65-93 (conditioner, forward), src/flows/utils.py:25-164 (spline); backward hand-derived
// (DESIGN.md "Backward"), as in nsf_device.h.
#pragma once
#include "nsf_cond_mfma.h"

namespace nsf {

// a[32..63] <-> b[0..31]  (the `s_nop 1` covers the two wait states a VALU write of either operand needs in front of the swap)
__device__ __forceinline__ void hswap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// lo = v of the pair's low lane, hi = v of its high lane, on both lanes
__device__ __forceinline__ void hgather(float v, float& lo, float& hi) {
    lo = v; hi = v;
    hswap(lo, hi);
}
// the low lanes receive a_low + a_high, the high lanes b_low + b_high
__device__ __forceinline__ float hreduce(float a, float b) {
    hswap(a, b);
    return a + b;
}

template <int K, int H>
struct HalfCfg {
    using LY = Layout<K, H>;
    using CP = CondPanel<K, H>;
    static_assert(H % 8 == 0, "a half owns whole groups of four hidden units");
    static constexpr int HQ = H / 2;            // hidden units per half
    static constexpr int GO = H / 8;            // groups of four of them
    static constexpr int HP = LY::HP;           // theta columns per half
    static constexpr int G2O = HP / 4;
    static constexpr int SG2 = (G2O % 4 == 0) ? 4 : ((G2O % 2 == 0) ? 2 : 1);
};
// theta column o of a half is a spline parameter of EITHER half (the MFMA that contracts it is skipped for the wave otherwise;
// the high half's extra column, K even, carries a zero gradient)
template <int K>
struct HalfColUsed {
    __host__ __device__ static constexpr bool at(int o) { return o < K + nd0_of(K); }
};

// theta (the lane's HP columns) and the hidden activations of dim i; xt = the tile [rows][xs], the lane's particle in column col.
// h1 / h2: all H units (after the all-gather); h1o / h2o: the lane's own H/2.
template <int K, int H>
__device__ __forceinline__ void cond_forward_half(const float* pan, int i, int s0, const float* xt, int xs, int lane, int col,
                                                  float (&h1o)[H / 2], float (&h2o)[H / 2], float (&h1)[H], float (&h2)[H],
                                                  float (&th)[Layout<K, H>::HP]) {
    using C = HalfCfg<K, H>;
    using CP = CondPanel<K, H>;
    constexpr int ST = CP::ST, GO = C::GO, HQ = C::HQ, HP = C::HP;
    const int r = lane & 3, up = lane >> 5;
    const int g0 = up * GO;                     // the half's first group of hidden units
    const cm_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {   // layer 0: eight inputs per round (zero weights past i, the clamped row keeps the activation finite), two chains per group
        cm_f32x4 a1[GO][2];
#pragma unroll
        for (int g = 0; g < GO; ++g) {
            a1[g][0] = *(const cm_f32x4*)(pan + CP::ob0 + 4 * (g0 + g));
            a1[g][1] = zero;
        }
        const float* w0 = pan + CP::oW0T + (4 * g0 + r) * s0;
        for (int k0 = 0; k0 < i; k0 += 8) {
            float xk[8];
            cm_f32x4 a4[GO][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u; xk[u] = xt[(k < i ? k : i) * xs + col]; }
#pragma unroll
            for (int g = 0; g < GO; ++g)
#pragma unroll
                for (int q = 0; q < 2; ++q) a4[g][q] = *(const cm_f32x4*)(w0 + 4 * g * s0 + k0 + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < GO; ++g)
#pragma unroll
                    for (int q = 0; q < 2; ++q) a1[g][q] = mfma1(a4[g][q][u], xk[4 * q + u], a1[g][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < GO; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) h1o[4 * g + u] = ftanh_scaled(a1[g][0][u] + a1[g][1][u]);
#pragma unroll
        for (int t = 0; t < HQ; ++t) hgather(h1o[t], h1[t], h1[HQ + t]);
    }
    {
        cm_f32x4 a2[GO * 2];
#pragma unroll
        for (int g = 0; g < GO; ++g) {
            a2[2 * g] = *(const cm_f32x4*)(pan + CP::ob1 + 4 * (g0 + g));
            a2[2 * g + 1] = zero;
        }
        mfma_rows<GO, H, GO, 2, EveryCol>(pan + CP::oW1T + (4 * g0 + r) * ST, ST, a2, h1);
#pragma unroll
        for (int g = 0; g < GO; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) h2o[4 * g + u] = ftanh_scaled(a2[2 * g][u] + a2[2 * g + 1][u]);
#pragma unroll
        for (int t = 0; t < HQ; ++t) hgather(h2o[t], h2[t], h2[HQ + t]);
    }
    {
        cm_f32x4 t[C::G2O];
#pragma unroll
        for (int g = 0; g < C::G2O; ++g) t[g] = *(const cm_f32x4*)(pan + CP::ob2 + HP * up + 4 * g);
        mfma_rows<C::G2O, H, C::SG2, 1, EveryCol>(pan + CP::oW2T + (HP * up + r) * ST, ST, t, h2);
#pragma unroll
        for (int g = 0; g < C::G2O; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) th[4 * g + u] = t[g][u];
    }
}

// ga2o / ga1o = dL/da2, dL/da1 of the lane's OWN hidden units from gth = dL/dtheta of the lane's own columns
template <int K, int H>
__device__ __forceinline__ void cond_backward_half(const float* pan, int lane, const float (&gth)[Layout<K, H>::HP],
                                                   const float (&h1o)[H / 2], const float (&h2o)[H / 2],
                                                   float (&ga2o)[H / 2], float (&ga1o)[H / 2]) {
    using C = HalfCfg<K, H>;
    using CP = CondPanel<K, H>;
    constexpr int GH = CP::GH, HQ = C::HQ, HP = C::HP;
    const int r = lane & 3, up = lane >> 5;
    const cm_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    {   // partial sums over the half's columns for ALL hidden units, then the halves' sums meet (reduce-scatter)
        cm_f32x4 s[GH * 2];
#pragma unroll
        for (int g = 0; g < GH * 2; ++g) s[g] = zero;
        mfma_rows<GH, HP, GH, 2, HalfColUsed<K>>(pan + CP::oW2N + r * CP::NS2 + HP * up, CP::NS2, s, gth);
        float ps[H];
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) ps[4 * g + u] = s[2 * g][u] + s[2 * g + 1][u];
#pragma unroll
        for (int t = 0; t < HQ; ++t) ga2o[t] = hreduce(ps[t], ps[HQ + t]) * (1.0f - h2o[t] * h2o[t]);
    }
    {   // the half's own units of ga2 contracted into ALL units of layer 1, then the same meeting
        cm_f32x4 t[GH];
#pragma unroll
        for (int g = 0; g < GH; ++g) t[g] = zero;
        mfma_rows<GH, HQ, GH, 1, EveryCol>(pan + CP::oW1N + r * CP::ST + HQ * up, CP::ST, t, ga2o);
        float ps[H];
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) ps[4 * g + u] = t[g][u];
#pragma unroll
        for (int tq = 0; tq < HQ; ++tq) ga1o[tq] = hreduce(ps[tq], ps[HQ + tq]) * (1.0f - h1o[tq] * h1o[tq]);
    }
}

// ---- the spline, one axis per lane (the instruction-count form of nsf_device.h: SplineT) --------------------------------------
template <int K>
struct SplineH {
    float e[K];                     // UNNORMALISED softmax terms of the lane's axis, exp(th - max)
    float inv;                      // 1 / their sum
    float L, sz;                    // the selected bin on the lane's axis: left knot, size
    float Xk, dx, Yk, dy, d0, d1, ud0, ud1, t;
    int k, kp;                      // bin; bin relative to the half's slope logits
    bool sel[K];                    // sel[j] = k >= j (j >= 1)
    bool inside;
};
// th = the lane's HP logits [K sizes | its slope logits | pad]; v = x (the bin is searched on the x axis: the low half's knots)
template <int K, int HP>
__device__ __forceinline__ void spline_half_fwd(float v, const float (&th)[HP], bool up, float B, SplineH<K>& S, float& z, float& lad) {
    constexpr int ND0 = nd0_of(K), ND1 = K - 1 - ND0;
    S.inside = (v >= -B) && (v <= B);
    const float vs = S.inside ? v : 0.0f;
    float m = th[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmaxf(m, th[j]);
    const float nm = -m * kLog2e;
    float E[K + 1];
    E[0] = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        S.e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(th[j], kLog2e, nm));
        E[j + 1] = E[j] + S.e[j];
    }
    S.inv = frcp(E[K]);
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    const float c = twoB * mix * S.inv;
    // knot j = -B + j * 2B * minbin + c * E[j]   (knot 0 = -B, knot K = B pinned: utils.py:85-92)
    float kn[K + 1];
    kn[0] = -B; kn[K] = B;
#pragma unroll
    for (int j = 1; j < K; ++j) kn[j] = __builtin_fmaf(c, E[j], (float)j * twoB * kMinBin - B);
    int kc = 0;
#pragma unroll
    for (int j = 1; j < K; ++j) kc += (vs >= kn[j]) ? 1 : 0;          // (meaningful on the low half: the x axis)
    {
        float kl = __int_as_float(kc), kh = kl;
        hswap(kl, kh);
        S.k = __float_as_int(kl);                                      // the low lane's count on both lanes
    }
    const int k = S.k;
    float Lk = kn[0], Rk = kn[1];
    S.sel[0] = true;
#pragma unroll
    for (int j = 1; j < K; ++j) {
        const bool s = k >= j;
        S.sel[j] = s;
        Lk = s ? kn[j] : Lk;
        Rk = s ? kn[j + 1] : Rk;
    }
    S.L = Lk; S.sz = Rk - Lk;
    hgather(S.L, S.Xk, S.Yk);
    hgather(S.sz, S.dx, S.dy);
    // slope logits: the low half holds those of the interior knots 1 .. ND0, the high half of ND0 + 1 .. K - 1
    S.kp = k - (up ? ND0 : 0);
    float m0 = kBoundLogit, m1 = kBoundLogit;
#pragma unroll
    for (int jj = 0; jj < ND0; ++jj) {
        const bool mine = (jj < ND1) ? true : !up;                     // (K even: the high half's last column is padding)
        const float dj = th[K + jj];
        m0 = (mine && S.kp == jj + 1) ? dj : m0;
        m1 = (mine && S.kp == jj) ? dj : m1;
    }
    float m0l, m0h, m1l, m1h;
    hgather(m0, m0l, m0h);
    hgather(m1, m1l, m1h);
    S.ud0 = (k <= ND0) ? m0l : m0h;
    S.ud1 = (k + 1 <= ND0) ? m1l : m1h;
    S.d0 = kMinDeriv + fsoftplus(S.ud0);
    S.d1 = kMinDeriv + fsoftplus(S.ud1);
    rq_math<false>(vs, S.Xk, S.dx, S.Yk, S.dy, S.d0, S.d1, S.t, z, lad);
    if (!S.inside) { z = v; lad = 0.0f; }
}
// gth = dL/dtheta of the lane's HP columns for upstream gz = dL/dz, gl = dL/dlogdet (pads written 0)
template <int K, int HP>
__device__ __forceinline__ void spline_half_bwd(const SplineH<K>& S, bool up, float B, float gz_in, float gl_in, float (&gth)[HP]) {
    constexpr int ND0 = nd0_of(K), ND1 = K - 1 - ND0;
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
    // the lane's axis: gradient w.r.t. the bin's left / right knot (end knots are pinned)
    const float gL = lo ? (up ? (gz - g_h) : (-g_x - g_w)) : 0.0f;
    const float gR = hi ? (up ? g_h : g_w) : 0.0f;
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    const float scale = twoB * mix;
    const float c1 = scale * (gL + gR), c2 = scale * gR;
    // softmax backward: gth_m = p_m (c_m - sum_j p_j c_j), c_m = c1 (m < k), c2 (m = k), 0 (m > k); the sum needs only
    // P(<k) and p_k, both known from the selected knot and bin size (nsf_device.h: spline_train_bwd)
    const float kf = (float)S.k, tk = twoB * kMinBin, r = frcp(scale);
    const float P = __builtin_fmaf(-kf, tk, S.L + B);
    const float dot = r * __builtin_fmaf(c1, P, c2 * (S.sz - tk));
    const float a1 = S.inv * (c1 - dot), a2 = S.inv * (c2 - dot), a3 = -S.inv * dot;
#pragma unroll
    for (int o = 0; o < HP; ++o) gth[o] = 0.0f;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        const bool ge = (m == 0) ? true : S.sel[m];                  // k >= m
        const bool gt = (m + 1 < K) ? S.sel[m + 1] : false;          // k >  m
        gth[m] = S.e[m] * (gt ? a1 : (ge ? a2 : a3));
    }
    const float gd0 = G_d0 * fsigmoid(S.ud0), gd1 = G_d1 * fsigmoid(S.ud1);
#pragma unroll
    for (int jj = 0; jj < ND0; ++jj) {
        const bool mine = (jj < ND1) ? true : !up;
        gth[K + jj] = (mine && S.kp == jj + 1) ? gd0 : ((mine && S.kp == jj) ? gd1 : 0.0f);
    }
}

// reduce-scatter over the 32 particles of ONE half: on return the lane with particle index p holds the half's total of
// v[p & (N-1)].  N a power of two <= 32.
template <int N>
__device__ __forceinline__ float butterfly_half(float (&v)[N], int p) {
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1) {
        const bool upl = (p & half) != 0;
#pragma unroll
        for (int t = 0; t < half; ++t) {
            const float lo = v[t], hi = v[t + half];
            const float keep = upl ? hi : lo;
            const float send = upl ? lo : hi;
            v[t] = keep + __shfl_xor(send, half, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int off = N; off < 32; off <<= 1) r += __shfl_xor(r, off, 64);
    return r;
}

}  // namespace nsf

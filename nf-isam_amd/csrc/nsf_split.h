// nsf_split.h — "two lanes per particle" device math of the spline flow (gfx950, wave64).
//
// The training kernel is bound by VALU issue per (layer, dim, tile) unit while most of the chip's SIMDs
// are idle on a single clique (4096 particles x 6 dims = 384 units for 1024 SIMDs).  Here a wave covers 32
// particles and the two lanes (2p, 2p+1) of a pair split one particle's unit:
//   conditioner  each lane computes H/2 hidden units per layer and its HP output logits; hidden
//                activations are exchanged with one DPP quad_perm move each
//   spline       lane 0 owns the x axis (width logits), lane 1 the y axis (height logits): softmax, cumulative
//                knots and their gradients are the same instruction stream on different data; the selected
//                bin's (left knot, size) and the two derivative logits are exchanged with DPP
//   backward     gradient w.r.t. hidden units: partial sums over the lane's own outputs, combined with a
//                DPP add; weight gradients: MFMA over the 32 particles (nsf_kernels.hip)
// Math and reference citations as in nsf_device.h; the parameter columns follow Layout's half order.
#pragma once
#include "nsf_device.h"

namespace nsf {

constexpr int TILE2 = 32;             // particles per wave (two lanes each)
constexpr int XS2 = 34;               // LDS row stride of [feature][particle] tiles (34 mod 32 = 2)

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
// value held by the partner lane (lane ^ 1): quad_perm [1,0,3,2]
__device__ __forceinline__ float pswap(float v) { return dppf<0xB1>(v); }

// ---- conditioner ----------------------------------------------------------------------------
// blk: this dim's parameter block (LDS or global), xs: [k][XS2] layer input, p = lane >> 1, hf = lane & 1.
// h*m = the lane's own H/2 hidden units (global index HH*hf + kk), h*o = the partner's.
template <int K, int H>
__device__ __forceinline__ void cond_hidden2(const float* blk, int i, const float* xs, int p, int hf,
                                             float (&h1m)[H / 2], float (&h1o)[H / 2],
                                             float (&h2m)[H / 2], float (&h2o)[H / 2]) {
    using LY = Layout<K, H>;
    constexpr int HH = H / 2;
    const int mo = HH * hf, oo = HH - mo;
    float a[HH], wr[HH];
    load_row<HH>(blk + LY::ob0(i) + mo, a);
    const float* W0 = blk + mo;
    for (int k = 0; k < i; k += 4) {           // rows k >= i alias later weights of the block: multiplied by 0
        float xk[4], wq[4][HH];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xk[u] = (k + u < i) ? xs[(k + u) * XS2 + p] : 0.0f;
            load_row<HH>(W0 + (k + u) * H, wq[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < HH; ++j) a[j] = __builtin_fmaf(wq[u][j], xk[u], a[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < HH; ++j) { h1m[j] = ftanh(a[j]); h1o[j] = pswap(h1m[j]); }
    const float* W1 = blk + LY::oW1(i) + mo;
    load_row<HH>(blk + LY::ob1(i) + mo, a);
#pragma unroll
    for (int kk = 0; kk < HH; ++kk) {
        load_row<HH>(W1 + (mo + kk) * H, wr);
#pragma unroll
        for (int j = 0; j < HH; ++j) a[j] = __builtin_fmaf(wr[j], h1m[kk], a[j]);
    }
#pragma unroll
    for (int kk = 0; kk < HH; ++kk) {
        load_row<HH>(W1 + (oo + kk) * H, wr);
#pragma unroll
        for (int j = 0; j < HH; ++j) a[j] = __builtin_fmaf(wr[j], h1o[kk], a[j]);
    }
#pragma unroll
    for (int j = 0; j < HH; ++j) { h2m[j] = ftanh(a[j]); h2o[j] = pswap(h2m[j]); }
}

template <int K, int H>
__device__ __forceinline__ void cond_theta2(const float* blk, int i, int hf, const float (&h2m)[H / 2],
                                            const float (&h2o)[H / 2], float (&th)[hp_of(K)]) {
    using LY = Layout<K, H>;
    constexpr int HH = H / 2, HP = LY::HP;
    const int mo = HH * hf, oo = HH - mo;
    constexpr int UO = K + LY::ND0;                           // used outputs per half (the rest is padding)
    const float* W2 = blk + LY::oW2(i) + HP * hf;
    load_row_used<HP, UO>(blk + LY::ob2(i) + HP * hf, th);
#pragma unroll
    for (int kk = 0; kk < HH; ++kk) {
        float wr[HP];
        load_row_used<HP, UO>(W2 + (mo + kk) * LY::PoP, wr);
#pragma unroll
        for (int o = 0; o < UO; ++o) th[o] = __builtin_fmaf(wr[o], h2m[kk], th[o]);
    }
#pragma unroll
    for (int kk = 0; kk < HH; ++kk) {
        float wr[HP];
        load_row_used<HP, UO>(W2 + (oo + kk) * LY::PoP, wr);
#pragma unroll
        for (int o = 0; o < UO; ++o) th[o] = __builtin_fmaf(wr[o], h2o[kk], th[o]);
    }
}

// ---- spline ----------------------------------------------------------------------------------
template <int K>
struct Spline2 {
    float p[K];                       // softmax probabilities of this lane's axis (widths or heights)
    float Xk, dx, Yk, dy, d0, d1;     // selected bin (both lanes hold all of it)
    float ud0, ud1, t;
    int k;
    bool inside;
};

// global index of the lane's jj-th derivative logit (interior knot index - 1), or a value no bin matches
template <int K>
__device__ __forceinline__ int deriv_index(int jj, int hf) {
    constexpr int ND0 = nd0_of(K), ND1 = K - 1 - ND0;
    return hf ? ((jj < ND1) ? ND0 + jj : -100) : jj;
}

// th: the lane's HP logits [K sizes | its derivative logits | pad].  INV = false: v is x (bin search on the
// x axis, done by lane 0); INV = true: v is z (search on the y axis, lane 1).  Both lanes return out and lad.
template <int K, bool INV>
__device__ __forceinline__ void spline_eval2(float v, const float (&th)[hp_of(K)], int hf, float B,
                                             Spline2<K>& S, float& out, float& lad) {
    constexpr int ND0 = nd0_of(K);
    S.inside = (v >= -B) && (v <= B);             // false for NaN (utils.py:31)
    const float vs = S.inside ? v : 0.0f;
    float m = th[0];
#pragma unroll
    for (int j = 1; j < K; ++j) m = fmaxf(m, th[j]);
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) { S.p[j] = fexp(th[j] - m); s += S.p[j]; }
    const float inv = frcp(s);
    const float mix = 1.0f - kMinBin * (float)K, twoB = 2.0f * B;
    float kn[K + 1];
    kn[0] = -B;
    float c = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        S.p[j] *= inv;
        c += kMinBin + mix * S.p[j];
        kn[j + 1] = (j == K - 1) ? B : twoB * c - B;          // last knot pinned (utils.py:90-91)
    }
    int km = 0;
#pragma unroll
    for (int j = 1; j < K; ++j) if (vs >= kn[j]) km = j;      // monotone knots: last true wins
    const int k = INV ? dppi<0xF5>(km) : dppi<0xA0>(km);       // the searching lane's bin, to both lanes
    float lft = kn[0], rgt = kn[1];
#pragma unroll
    for (int j = 1; j < K; ++j) if (k >= j) { lft = kn[j]; rgt = kn[j + 1]; }
    const float msz = rgt - lft;
    const float ol = pswap(lft), osz = pswap(msz);
    S.k = k;
    S.Xk = hf ? ol : lft;  S.dx = hf ? osz : msz;
    S.Yk = hf ? lft : ol;  S.dy = hf ? msz : osz;
    float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int jj = 0; jj < ND0; ++jj) {
        const int g = deriv_index<K>(jj, hf);
        const float dj = th[K + jj];
        if (g == k - 1) m0 = dj;
        if (g == k) m1 = dj;
    }
    const float o0 = pswap(m0), o1 = pswap(m1);
    const int lo = hf ? ND0 : 0, hi = hf ? K - 1 : ND0;
    S.ud0 = (k == 0) ? kBoundLogit : ((k - 1 >= lo && k - 1 < hi) ? m0 : o0);
    S.ud1 = (k >= K - 1) ? kBoundLogit : ((k >= lo && k < hi) ? m1 : o1);
    S.d0 = kMinDeriv + fsoftplus(S.ud0);
    S.d1 = kMinDeriv + fsoftplus(S.ud1);
    rq_math<INV>(vs, S.Xk, S.dx, S.Yk, S.dy, S.d0, S.d1, S.t, out, lad);
    if (!S.inside) { out = v; lad = 0.0f; }
}

// Backward of the forward spline: the lane's HP entries of dL/dtheta; returns dL/dx through the spline's
// own argument (both lanes).  Same algebra as spline_backward (nsf_device.h).
template <int K>
__device__ __forceinline__ float spline_backward2(const Spline2<K>& S, int hf, float B, float gz, float gl,
                                                  float (&gth)[hp_of(K)]) {
    constexpr int ND0 = nd0_of(K), HP = hp_of(K);
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
    // this lane's axis: gradient w.r.t. the bin's left / right knot
    const float gL = lo ? (hf ? (gz - g_h) : (-g_x - g_w)) : 0.0f;
    const float gR = hi ? (hf ? g_h : g_w) : 0.0f;
    const float scale = 2.0f * B * (1.0f - kMinBin * (float)K);
    const float c1 = scale * (gL + gR), c2 = scale * gR;
    float dot = 0.0f;
#pragma unroll
    for (int mm = 0; mm < K; ++mm) {
        const float cc = (mm < k) ? c1 : ((mm == k) ? c2 : 0.0f);
        dot = __builtin_fmaf(S.p[mm], cc, dot);
    }
#pragma unroll
    for (int mm = 0; mm < K; ++mm) {
        const float cc = (mm < k) ? c1 : ((mm == k) ? c2 : 0.0f);
        gth[mm] = S.p[mm] * (cc - dot);
    }
    const float gd0 = G_d0 * fsigmoid(S.ud0), gd1 = G_d1 * fsigmoid(S.ud1);
#pragma unroll
    for (int jj = 0; jj < ND0; ++jj) {
        const int g = deriv_index<K>(jj, hf);
        float v = 0.0f;
        if (g == k - 1) v = gd0;
        if (g == k) v = gd1;
        gth[K + jj] = v;
    }
#pragma unroll
    for (int o = K + ND0; o < HP; ++o) gth[o] = 0.0f;
    if (!S.inside) {
#pragma unroll
        for (int o = 0; o < HP; ++o) gth[o] = 0.0f;
        return gz;
    }
    return g_x;
}

// reduce-scatter over the 32 particles of a wave (lanes of equal parity): on return the lane with
// particle index p holds the total of v[p & (N-1)] of its half.  N a power of two <= 32.
template <int N>
__device__ __forceinline__ float butterfly2(float (&v)[N], int p) {
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1) {
        const bool up = (p & half) != 0;
#pragma unroll
        for (int t = 0; t < half; ++t) {
            const float lo = v[t], hi = v[t + half];
            const float keep = up ? hi : lo;
            const float send = up ? lo : hi;
            v[t] = keep + __shfl_xor(send, 2 * half, 64);
        }
    }
    float r = v[0];
#pragma unroll
    for (int off = N; off < 32; off <<= 1) r += __shfl_xor(r, 2 * off, 64);
    return r;
}

}  // namespace nsf

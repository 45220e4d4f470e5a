/* ORACLE (test infrastructure, NOT product code).
 *
 * Body of the plain-C restatement of the NF-iSAM flow hot path; included twice by
 * nsf_oracle.c with R = float (suffix _f) and R = double (suffix _d).
 *
 * Reference (paths relative to /root/reference):
 *   conditioner MLP                    src/flows/flows.py:26-41,77-83
 *   knots / bin search / RQ spline     src/flows/utils.py:17-22,25-66,69-164
 *   layer chaining, N(0,I) prior       src/flows/models.py:11-35 ; src/flows/prior_dist.py:5-26
 *   NLL, Adam, window early stop       src/slam/NFiSAM.py:425,451-491
 * The reference obtains gradients with torch autograd; the analytic backward below is
 * derived by hand (DESIGN.md "Backward") and pinned against the autograd gradients stored
 * in tests/golden (tests/test_oracle_golden.py).
 *
 * Blob layout ("torch layout", one block per flow layer):
 *   init_param[Po] | for i=1..D-1: W0[H][i] b0[H] W1[H][H] b1[H] W2[Po][H] b2[Po],  Po = 3K-1
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define MAXK 64
#define MAXH 64
#define MAXD 256

static R FN(r_exp)(R v) { return (R)exp((double)v); }
static R FN(r_log)(R v) { return (R)log((double)v); }
static R FN(r_tanh)(R v) { return (R)tanh((double)v); }
static R FN(r_sqrt)(R v) { return (R)sqrt((double)v); }
static R FN(r_softplus)(R v) { /* torch softplus, threshold 20 */
    return v > (R)20 ? v : (R)log1p(exp((double)v));
}
static R FN(r_sigmoid)(R v) { return (R)(1.0 / (1.0 + exp(-(double)v))); }

typedef struct {
    /* everything the forward pass of one (particle, dim) element leaves behind */
    int inside, k;
    R p_w[MAXK], p_h[MAXK];      /* softmax probabilities */
    R Xk, dx, Yk, dy, d0, d1;    /* selected bin */
    R ud0, ud1;                  /* unnormalised derivative logits at knots k, k+1 */
    R t, out, lad;
} FN(Spl);

/* knots from unnormalised logits: X[0..K], probabilities p[0..K-1] (utils.py:85-92) */
static void FN(knots)(const R* un, int K, R B, R min_size, R* p, R* X) {
    R mx = un[0];
    for (int j = 1; j < K; ++j) if (un[j] > mx) mx = un[j];
    R s = 0;
    for (int j = 0; j < K; ++j) { p[j] = FN(r_exp)(un[j] - mx); s += p[j]; }
    R c = 0;
    X[0] = -B;
    for (int j = 0; j < K; ++j) {
        p[j] /= s;
        c += min_size + ((R)1 - min_size * (R)K) * p[j];
        X[j + 1] = (R)2 * B * c - B;
    }
    X[K] = B;
}

/* bin index: #knots <= v, minus 1, with the last knot bumped by 1e-6 (utils.py:17-22) */
static int FN(bin)(const R* X, int K, R v) {
    int cnt = 0;
    for (int j = 0; j < K; ++j) cnt += (v >= X[j]);
    cnt += (v >= X[K] + (R)1e-6);
    int k = cnt - 1;
    if (k < 0) k = 0;
    if (k > K - 1) k = K - 1;
    return k;
}

static void FN(spline_eval)(R v, const R* theta, int K, R B, int inverse, FN(Spl)* S) {
    S->inside = (v >= -B) && (v <= B);
    if (!S->inside) { S->out = v; S->lad = 0; return; }
    R X[MAXK + 1], Y[MAXK + 1];
    FN(knots)(theta, K, B, (R)1e-3, S->p_w, X);
    FN(knots)(theta + K, K, B, (R)1e-3, S->p_h, Y);
    int k = FN(bin)(inverse ? Y : X, K, v);
    S->k = k;
    const R cst = (R)log(exp(1.0 - 1e-3) - 1.0);
    S->ud0 = (k == 0) ? cst : theta[2 * K + k - 1];
    S->ud1 = (k + 1 == K) ? cst : theta[2 * K + k];
    S->d0 = (R)1e-3 + FN(r_softplus)(S->ud0);
    S->d1 = (R)1e-3 + FN(r_softplus)(S->ud1);
    S->Xk = X[k]; S->dx = X[k + 1] - X[k];
    S->Yk = Y[k]; S->dy = Y[k + 1] - Y[k];
    R s = S->dy / S->dx, sig = S->d0 + S->d1 - (R)2 * s, t;
    if (inverse) {
        R dl = v - S->Yk;
        R a = dl * sig + S->dy * (s - S->d0);
        R b = S->dy * S->d0 - dl * sig;
        R c = -s * dl;
        R disc = b * b - (R)4 * a * c;
        t = ((R)2 * c) / (-b - FN(r_sqrt)(disc));
        S->out = t * S->dx + S->Xk;
    } else {
        t = (v - S->Xk) / S->dx;
    }
    S->t = t;
    R q = t * ((R)1 - t), den = s + sig * q;
    R M = S->d1 * t * t + (R)2 * s * q + S->d0 * ((R)1 - t) * ((R)1 - t);
    R lad = FN(r_log)(s * s * M) - (R)2 * FN(r_log)(den);
    if (inverse) S->lad = -lad;
    else { S->out = S->Yk + S->dy * (s * t * t + S->d0 * q) / den; S->lad = lad; }
}

/* backward of the FORWARD spline: upstream gz = dL/dz, gl = dL/dlad.
 * gtheta[3K-1] is OVERWRITTEN; returns dL/dx (through the spline's own argument only). */
static R FN(spline_backward)(const FN(Spl)* S, int K, R B, R gz, R gl, R* gtheta) {
    for (int j = 0; j < 3 * K - 1; ++j) gtheta[j] = 0;
    if (!S->inside) return gz;
    int k = S->k;
    R w = S->dx, h = S->dy, d0 = S->d0, d1 = S->d1, t = S->t;
    R s = h / w, sig = d0 + d1 - (R)2 * s, q = t * ((R)1 - t), omt = (R)1 - t, o2t = (R)1 - (R)2 * t;
    R N = s * t * t + d0 * q, den = s + sig * q, u = N / den, iden2 = (R)1 / (den * den);
    R u_t = (((R)2 * s * t + d0 * o2t) * den - N * sig * o2t) * iden2;
    R u_s = (t * t * den - N * ((R)1 - (R)2 * q)) * iden2;
    R u_d0 = q * (den - N) * iden2;
    R u_d1 = -N * q * iden2;
    R M = d1 * t * t + (R)2 * s * q + d0 * omt * omt;
    R M_t = (R)2 * d1 * t + (R)2 * s * o2t - (R)2 * d0 * omt;
    R ld_t = M_t / M - (R)2 * sig * o2t / den;
    R ld_s = (R)2 / s + (R)2 * q / M - (R)2 * ((R)1 - (R)2 * q) / den;
    R ld_d0 = omt * omt / M - (R)2 * q / den;
    R ld_d1 = t * t / M - (R)2 * q / den;
    R G_t = gz * h * u_t + gl * ld_t;
    R G_s = gz * h * u_s + gl * ld_s;
    R G_d0 = gz * h * u_d0 + gl * ld_d0;
    R G_d1 = gz * h * u_d1 + gl * ld_d1;
    R g_x = G_t / w;
    R g_a = -G_t / w;                       /* d/dX_k   at fixed bin width   */
    R g_w = -G_t * t / w - G_s * s / w;     /* d/d(bin width)                */
    R g_c = gz;                             /* d/dY_k   at fixed bin height  */
    R g_h = gz * u + G_s / w;               /* d/d(bin height)               */
    /* knots: X_k gets g_a - g_w, X_{k+1} gets g_w; end knots are pinned (no gradient) */
    R gXk = (k >= 1) ? (g_a - g_w) : 0, gXk1 = (k + 1 <= K - 1) ? g_w : 0;
    R gYk = (k >= 1) ? (g_c - g_h) : 0, gYk1 = (k + 1 <= K - 1) ? g_h : 0;
    R scale = (R)2 * B * ((R)1 - (R)1e-3 * (R)K);
    {   /* widths: dL/dp_m = scale * [ (gXk+gXk1) 1(m<k) + gXk1 1(m==k) ] ; softmax backward */
        R c1 = scale * (gXk + gXk1), c2 = scale * gXk1, dot = 0;
        for (int m = 0; m < K; ++m) dot += S->p_w[m] * (m < k ? c1 : (m == k ? c2 : 0));
        for (int m = 0; m < K; ++m) gtheta[m] = S->p_w[m] * ((m < k ? c1 : (m == k ? c2 : 0)) - dot);
    }
    {
        R c1 = scale * (gYk + gYk1), c2 = scale * gYk1, dot = 0;
        for (int m = 0; m < K; ++m) dot += S->p_h[m] * (m < k ? c1 : (m == k ? c2 : 0));
        for (int m = 0; m < K; ++m) gtheta[K + m] = S->p_h[m] * ((m < k ? c1 : (m == k ? c2 : 0)) - dot);
    }
    if (k >= 1) gtheta[2 * K + k - 1] += G_d0 * FN(r_sigmoid)(S->ud0);
    if (k + 1 <= K - 1) gtheta[2 * K + k] += G_d1 * FN(r_sigmoid)(S->ud1);
    return g_x;
}

static size_t FN(dim_block)(int i, int K, int H) {
    size_t Po = 3 * (size_t)K - 1;
    return (size_t)i * H + H + (size_t)H * H + H + (size_t)H * Po + Po;
}
static size_t FN(dim_off)(int i, int K, int H) { /* offset of dim i's block (i>=1) */
    size_t off = 3 * (size_t)K - 1;
    for (int m = 1; m < i; ++m) off += FN(dim_block)(m, K, H);
    return off;
}
size_t FN(nsf_oracle_param_count)(int D, int K, int H) { return FN(dim_off)(D, K, H); }

/* conditioner for dim i of one particle; keeps activations for backward */
static void FN(cond_fwd)(const R* xrow, int i, const R* blk, int K, int H, R* h1, R* h2, R* theta) {
    int Po = 3 * K - 1;
    const R *W0 = blk, *b0 = W0 + (size_t)H * i, *W1 = b0 + H, *b1 = W1 + (size_t)H * H,
            *W2 = b1 + H, *b2 = W2 + (size_t)Po * H;
    for (int j = 0; j < H; ++j) {
        R a = b0[j];
        for (int k = 0; k < i; ++k) a += W0[(size_t)j * i + k] * xrow[k];
        h1[j] = FN(r_tanh)(a);
    }
    for (int j = 0; j < H; ++j) {
        R a = b1[j];
        for (int k = 0; k < H; ++k) a += W1[(size_t)j * H + k] * h1[k];
        h2[j] = FN(r_tanh)(a);
    }
    for (int o = 0; o < Po; ++o) {
        R a = b2[o];
        for (int k = 0; k < H; ++k) a += W2[(size_t)o * H + k] * h2[k];
        theta[o] = a;
    }
}

/* accumulate parameter gradients of dim i's block from gtheta; optionally gx[0..i-1] += ... */
static void FN(cond_bwd)(const R* xrow, int i, const R* blk, R* gblk, int K, int H, const R* h1,
                         const R* h2, const R* gtheta, R* gx) {
    int Po = 3 * K - 1;
    const R *W0 = blk, *W1 = W0 + (size_t)H * i + H, *W2 = W1 + (size_t)H * H + H;
    R *gW0 = gblk, *gb0 = gW0 + (size_t)H * i, *gW1 = gb0 + H, *gb1 = gW1 + (size_t)H * H,
      *gW2 = gb1 + H, *gb2 = gW2 + (size_t)Po * H;
    R gh2[MAXH], ga2[MAXH], gh1[MAXH], ga1[MAXH];
    for (int k = 0; k < H; ++k) gh2[k] = 0;
    for (int o = 0; o < Po; ++o) {
        gb2[o] += gtheta[o];
        for (int k = 0; k < H; ++k) {
            gW2[(size_t)o * H + k] += gtheta[o] * h2[k];
            gh2[k] += W2[(size_t)o * H + k] * gtheta[o];
        }
    }
    for (int k = 0; k < H; ++k) { ga2[k] = gh2[k] * ((R)1 - h2[k] * h2[k]); gh1[k] = 0; }
    for (int j = 0; j < H; ++j) {
        gb1[j] += ga2[j];
        for (int k = 0; k < H; ++k) {
            gW1[(size_t)j * H + k] += ga2[j] * h1[k];
            gh1[k] += W1[(size_t)j * H + k] * ga2[j];
        }
    }
    for (int k = 0; k < H; ++k) ga1[k] = gh1[k] * ((R)1 - h1[k] * h1[k]);
    for (int j = 0; j < H; ++j) {
        gb0[j] += ga1[j];
        for (int k = 0; k < i; ++k) {
            gW0[(size_t)j * i + k] += ga1[j] * xrow[k];
            if (gx) gx[k] += W0[(size_t)j * i + k] * ga1[j];
        }
    }
}

/* ---- forward through L layers: x[n,D] -> z[n,D], logdet[n] (either output may be NULL) */
int FN(nsf_oracle_forward)(const R* x, const R* blob, int n, int D, int K, int H, R B, int L,
                           R* z, R* logdet) {
    if (K > MAXK || H > MAXH || D > MAXD || K < 1 || D < 1) return 1;
    size_t P = FN(nsf_oracle_param_count)(D, K, H);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < n; ++p) {
        R cur[MAXD], nxt[MAXD], h1[MAXH], h2[MAXH], theta[3 * MAXK];
        FN(Spl) S;
        for (int i = 0; i < D; ++i) cur[i] = x[(size_t)p * D + i];
        R ld = 0;
        for (int l = 0; l < L; ++l) {
            const R* lb = blob + (size_t)l * P;
            for (int i = 0; i < D; ++i) {
                const R* th = lb;
                if (i > 0) { FN(cond_fwd)(cur, i, lb + FN(dim_off)(i, K, H), K, H, h1, h2, theta); th = theta; }
                FN(spline_eval)(cur[i], th, K, B, 0, &S);
                nxt[i] = S.out; ld += S.lad;
            }
            for (int i = 0; i < D; ++i) cur[i] = nxt[i];
        }
        if (z) for (int i = 0; i < D; ++i) z[(size_t)p * D + i] = cur[i];
        if (logdet) logdet[p] = ld;
    }
    return 0;
}

/* ---- generic backward: given gz[n,D] (dLoss/dz) and gl[n] (dLoss/dlogdet) accumulate
 * grad[P*L] (must be zeroed by the caller) and, if gx != NULL, write dLoss/dx[n,D].
 * nll_mode != 0: ignore gz/gl and use the NLL loss  mean_p( 0.5|z|^2 + D/2 log 2pi - logdet ):
 * gz = z/n, gl = -1/n; *loss receives the loss; logprob[n] (optional) the per-particle log p. */
int FN(nsf_oracle_backward)(const R* x, const R* blob, int n, int D, int K, int H, R B, int L,
                            const R* gz_in, const R* gl_in, int nll_mode, R* grad, R* gx_out,
                            double* loss, R* logprob) {
    if (K > MAXK || H > MAXH || D > MAXD || K < 1 || D < 1 || L < 1 || L > 16) return 1;
    size_t P = FN(nsf_oracle_param_count)(D, K, H);
    int Po = 3 * K - 1;
    double loss_acc = 0;
#pragma omp parallel
    {
        R* gl_local = (R*)calloc(P * (size_t)L, sizeof(R));
        double loss_local = 0;
#pragma omp for schedule(static)
        for (int p = 0; p < n; ++p) {
            R xs[17][MAXD], g[MAXD], gprev[MAXD], h1[MAXH], h2[MAXH], theta[3 * MAXK], gth[3 * MAXK];
            FN(Spl) S;
            for (int i = 0; i < D; ++i) xs[0][i] = x[(size_t)p * D + i];
            R ld = 0;
            for (int l = 0; l < L; ++l) {            /* forward, keep every layer input */
                const R* lb = blob + (size_t)l * P;
                for (int i = 0; i < D; ++i) {
                    const R* th = lb;
                    if (i > 0) { FN(cond_fwd)(xs[l], i, lb + FN(dim_off)(i, K, H), K, H, h1, h2, theta); th = theta; }
                    FN(spline_eval)(xs[l][i], th, K, B, 0, &S);
                    xs[l + 1][i] = S.out; ld += S.lad;
                }
            }
            R gl;
            if (nll_mode) {
                R zz = 0;
                for (int i = 0; i < D; ++i) { zz += xs[L][i] * xs[L][i]; g[i] = xs[L][i] / (R)n; }
                gl = (R)-1 / (R)n;
                double lp = -0.5 * (double)zz - 0.5 * D * log(2 * M_PI) + (double)ld;
                loss_local += -lp;
                if (logprob) logprob[p] = (R)lp;
            } else {
                for (int i = 0; i < D; ++i) g[i] = gz_in[(size_t)p * D + i];
                gl = gl_in ? gl_in[p] : 0;
            }
            for (int l = L - 1; l >= 0; --l) {       /* backward with recompute */
                const R* lb = blob + (size_t)l * P;
                R* glb = gl_local + (size_t)l * P;
                for (int i = 0; i < D; ++i) gprev[i] = 0;
                for (int i = 0; i < D; ++i) {
                    const R* th = lb;
                    if (i > 0) { FN(cond_fwd)(xs[l], i, lb + FN(dim_off)(i, K, H), K, H, h1, h2, theta); th = theta; }
                    FN(spline_eval)(xs[l][i], th, K, B, 0, &S);
                    gprev[i] += FN(spline_backward)(&S, K, B, g[i], gl, gth);
                    if (i == 0) for (int o = 0; o < Po; ++o) glb[o] += gth[o];
                    else FN(cond_bwd)(xs[l], i, lb + FN(dim_off)(i, K, H), glb + FN(dim_off)(i, K, H),
                                      K, H, h1, h2, gth, gprev);
                }
                for (int i = 0; i < D; ++i) g[i] = gprev[i];
            }
            if (gx_out) for (int i = 0; i < D; ++i) gx_out[(size_t)p * D + i] = g[i];
        }
#pragma omp critical
        {
            for (size_t j = 0; j < P * (size_t)L; ++j) grad[j] += gl_local[j];
            loss_acc += loss_local;
        }
        free(gl_local);
    }
    if (loss) *loss = loss_acc / n;
    return 0;
}

/* ---- inverse: z[n,D-Ds] (+ x_sep[n,Ds] or NULL) -> x_free[n,D-Ds], logdet[n] (optional).
 * L == 1 is the reference (src/flows/flows.py:115-137, src/slam/NFiSAM.py:140-155).  For L > 1 the reference
 * conditions every layer on the same raw x_sep (NFiSAM.py:151-152), which inverts nothing (its multi-layer
 * forward is scrambled as well).  Here the given columns are pushed through the marginal flow
 * of layers 0..l-1 first (the flow is autoregressive, so the first Ds columns of every layer's input depend on
 * x_sep only), and layer l is conditioned on them: forward(concat(x_sep, inverse(z, x_sep))) returns z. */
int FN(nsf_oracle_inverse)(const R* z, const R* x_sep, const R* blob, int n, int D, int Ds, int K,
                           int H, R B, int L, R* x_out, R* logdet) {
    if (K > MAXK || H > MAXH || D > MAXD || Ds < 0 || Ds >= D || L < 1 || L > 16) return 1;
    size_t P = FN(nsf_oracle_param_count)(D, K, H);
    int F = D - Ds;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < n; ++p) {
        R ys[16][MAXD], row[MAXD], cur[MAXD], h1[MAXH], h2[MAXH], theta[3 * MAXK];
        FN(Spl) S;
        for (int i = 0; i < Ds; ++i) ys[0][i] = x_sep[(size_t)p * Ds + i];
        for (int l = 0; l + 1 < L; ++l) {           /* given columns as seen by layer l + 1 */
            const R* lb = blob + (size_t)l * P;
            for (int i = 0; i < Ds; ++i) {
                const R* th = lb;
                if (i > 0) { FN(cond_fwd)(ys[l], i, lb + FN(dim_off)(i, K, H), K, H, h1, h2, theta); th = theta; }
                FN(spline_eval)(ys[l][i], th, K, B, 0, &S);
                ys[l + 1][i] = S.out;
            }
        }
        for (int i = 0; i < F; ++i) cur[i] = z[(size_t)p * F + i];
        R ld = 0;
        for (int l = L - 1; l >= 0; --l) {
            const R* lb = blob + (size_t)l * P;
            for (int i = 0; i < Ds; ++i) row[i] = ys[l][i];
            for (int i = Ds; i < D; ++i) {
                const R* th = lb;
                if (i > 0) { FN(cond_fwd)(row, i, lb + FN(dim_off)(i, K, H), K, H, h1, h2, theta); th = theta; }
                FN(spline_eval)(cur[i - Ds], th, K, B, 1, &S);
                row[i] = S.out; ld += S.lad;
            }
            for (int i = 0; i < F; ++i) cur[i] = row[Ds + i];
        }
        for (int i = 0; i < F; ++i) x_out[(size_t)p * F + i] = cur[i];
        if (logdet) logdet[p] = ld;
    }
    return 0;
}

/* ---- torch.optim.Adam step (defaults: no amsgrad, no weight decay); t is 1-based */
void FN(nsf_oracle_adam)(R* theta, const R* grad, R* m, R* v, size_t P, R lr, R b1, R b2, R eps, int t) {
    double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    R step = (R)((double)lr / bc1), bc2s = (R)sqrt(bc2);
    for (size_t j = 0; j < P; ++j) {
        m[j] = b1 * m[j] + ((R)1 - b1) * grad[j];
        v[j] = b2 * v[j] + ((R)1 - b2) * grad[j] * grad[j];
        R denom = FN(r_sqrt)(v[j]) / bc2s + eps;
        theta[j] -= step * m[j] / denom;
    }
}

/* ---- full-batch training loop with the reference's window early stop (NFiSAM.py:451-491).
 * iter_loss[max_iters] is zero-filled first; returns iterations run in *iters_run. */
int FN(nsf_oracle_train)(const R* x, R* blob, R* m, R* v, int n, int D, int K, int H, R B, int L,
                         R lr, int max_iters, int average_window, R loss_delta_tol, int early_stop,
                         R* iter_loss, int* iters_run) {
    size_t P = FN(nsf_oracle_param_count)(D, K, H) * (size_t)L;
    R* grad = (R*)malloc(P * sizeof(R));
    for (int i = 0; i < max_iters; ++i) iter_loss[i] = 0;
    int have_avg = 0, it = 0;
    R loss_avg = 0;
    for (int i = 0; i < max_iters; ++i) {
        memset(grad, 0, P * sizeof(R));
        double loss;
        int rc = FN(nsf_oracle_backward)(x, blob, n, D, K, H, B, L, NULL, NULL, 1, grad, NULL, &loss, NULL);
        if (rc) { free(grad); return rc; }
        iter_loss[i] = (R)loss;
        FN(nsf_oracle_adam)(blob, grad, m, v, P, lr, (R)0.9, (R)0.999, (R)1e-8, i + 1);
        it = i + 1;
        if (early_stop && (i + 1) % average_window == 0) {
            R s = 0;
            for (int j = i - average_window + 1; j <= i; ++j) s += iter_loss[j];
            R nw = s / (R)average_window;
            if (have_avg && loss_avg != 0) {
                R delta = (R)fabs(1.0 - (double)(nw / loss_avg));
                if (delta < loss_delta_tol) break;
            }
            loss_avg = nw; have_avg = 1;
        }
    }
    *iters_run = it;
    free(grad);
    return 0;
}

#undef MAXK
#undef MAXH
#undef MAXD
#undef FN
#undef CAT
#undef CAT_

/* TEST INFRASTRUCTURE — CPU restatement of the reference forward pass.  NOT part of the product path.
 *
 * Included twice by tip_oracle.c with REAL = float / double and FN(name) = name##_f32 / name##_f64.
 * Every block cites the reference line it restates (paths relative to /root/reference;
 * "torch:" = the PyTorch built-in the reference delegates to, pin pytorch==1.7.1 README.md:35,
 * validated here against torch 2.10.0 CPU).  Plain loops, one rounding per operation, no blocking,
 * no fused tricks: this file is the arithmetic specification the HIP kernels are checked against.
 */

/* y[n] = b[n] + sum_k x[k] * W[n*K + k]            (torch nn.Linear: y = x W^T + b) */
static void FN(linear_row)(const REAL* x, const REAL* W, const REAL* b, REAL* y, int N, int K) {
    for (int n = 0; n < N; ++n) {
        REAL acc = b ? b[n] : (REAL)0;
        const REAL* w = W + (size_t)n * K;
        for (int k = 0; k < K; ++k) acc += x[k] * w[k];
        y[n] = acc;
    }
}

/* torch: nn.LayerNorm(d, eps=1e-5), biased variance, affine (transformer.py post-norm branch) */
static void FN(layernorm_row)(REAL* x, const REAL* g, const REAL* be, int D) {
    REAL mean = 0;
    for (int i = 0; i < D; ++i) mean += x[i];
    mean /= (REAL)D;
    REAL var = 0;
    for (int i = 0; i < D; ++i) { REAL d = x[i] - mean; var += d * d; }
    var /= (REAL)D;
    REAL rstd = (REAL)1 / SQRT(var + (REAL)1e-5);
    for (int i = 0; i < D; ++i) x[i] = (x[i] - mean) * rstd * g[i] + be[i];
}

/* One window (one batch element).  Scratch is caller-provided so the batch loop can be OpenMP'd. */
static void FN(window)(const tip_oracle_cfg* c, const REAL* const* w, const REAL* x_imu, const REAL* x_s,
                       REAL* y, int T, const REAL* keep_mask, REAL keep_scale,
                       REAL* tap_in, REAL* tap_layers, REAL* tap_rnn, size_t tap_layer_stride, REAL* scratch) {
    const int NI = c->n_imu_total, S = c->n_state, D = c->d_model, H = c->n_heads, F = c->d_ff, L = c->n_layers;
    const int R = c->d_rnn, In = NI + S, dh = D / H;
    REAL* u    = scratch;                 /* [In]        */
    REAL* z    = u + In;                  /* [T][D] current activations (time-major inside one window) */
    REAL* qkv  = z + (size_t)T * D;       /* [T][3D]     */
    REAL* att  = qkv + (size_t)T * 3 * D; /* [T][D]      */
    REAL* tmp  = att + (size_t)T * D;     /* [max(D,F,R)]*/
    REAL* tmp2 = tmp + (size_t)(F > R ? (F > D ? F : D) : (R > D ? R : D)); /* [D] */
    REAL* p    = tmp2 + D;                /* [T] softmax row */
    REAL* h    = p + T;                   /* [R] */
    REAL* hn   = h + R;                   /* [R] */

    const REAL* Win = w[0];
    const REAL* bin = w[1];
    for (int t = 0; t < T; ++t) {
        /* simple_transformer_with_state.py:63-78: clone; x_s[isnan]=0; Dropout(in_dropout=0) on x_imu;
         * x_s[:,:,108:111] *= 0; Dropout(past_state_dropout) on x_s (here: explicit keep-mask, scale 1/(1-p));
         * cat((x_imu, x_s), dim=2). */
        for (int i = 0; i < NI; ++i) u[i] = x_imu[(size_t)t * NI + i];
        for (int i = 0; i < S; ++i) {
            REAL v = x_s[(size_t)t * S + i];
            if (v != v) v = 0;                                 /* :65 */
            if (i >= c->rootv_begin && i < c->rootv_end) v *= (REAL)0; /* :75 (18*6 .. 18*6+3) */
            if (keep_mask) v = v * keep_mask[(size_t)t * S + i] * keep_scale; /* :77 */
            u[NI + i] = v;
        }
        /* :79 in_linear */
        FN(linear_row)(u, Win, bin, tmp, D, In);
        /* :88-89 x.reshape(T,B,H,D/H).transpose(2,3).reshape(T,B,D):
         * new channel a*H + b  <-  old channel b*(D/H) + a,  a in [0,D/H), b in [0,H) */
        for (int a = 0; a < dh; ++a)
            for (int b = 0; b < H; ++b) z[(size_t)t * D + a * H + b] = tmp[b * dh + a];
    }
    if (tap_in) for (size_t i = 0; i < (size_t)T * D; ++i) tap_in[i] = z[i];

    /* :91 tf_encode(x, mask): L x nn.TransformerEncoderLayer, post-norm, ReLU
     * (torch: nn/modules/transformer.py _sa_block/_ff_block; functional.py multi_head_attention_forward) */
    const REAL scale = (REAL)1 / SQRT((REAL)dh);
    for (int l = 0; l < L; ++l) {
        const REAL* const* lw = w + 2 + 12 * l;
        const REAL *Wqkv = lw[0], *bqkv = lw[1], *Wo = lw[2], *bo = lw[3], *W1 = lw[4], *b1 = lw[5];
        const REAL *W2 = lw[6], *b2 = lw[7], *g1 = lw[8], *be1 = lw[9], *g2 = lw[10], *be2 = lw[11];
        /* packed in-projection: rows 0:D = Q, D:2D = K, 2D:3D = V */
        for (int t = 0; t < T; ++t) FN(linear_row)(z + (size_t)t * D, Wqkv, bqkv, qkv + (size_t)t * 3 * D, 3 * D, D);
        /* head hd uses channels [hd*dh, (hd+1)*dh); causal mask :56-58,:85 (key j visible iff j <= i) */
        for (int hd = 0; hd < H; ++hd) {
            for (int i = 0; i < T; ++i) {
                const REAL* q = qkv + (size_t)i * 3 * D + hd * dh;
                REAL mx = -(REAL)INFINITY;
                for (int j = 0; j <= i; ++j) {
                    const REAL* k = qkv + (size_t)j * 3 * D + D + hd * dh;
                    REAL s = 0;
                    for (int e = 0; e < dh; ++e) s += (q[e] * scale) * k[e];
                    p[j] = s;
                    if (s > mx) mx = s;
                }
                REAL den = 0;
                for (int j = 0; j <= i; ++j) { p[j] = EXP(p[j] - mx); den += p[j]; }
                for (int e = 0; e < dh; ++e) {
                    REAL o = 0;
                    for (int j = 0; j <= i; ++j) o += (p[j] / den) * qkv[(size_t)j * 3 * D + 2 * D + hd * dh + e];
                    att[(size_t)i * D + hd * dh + e] = o;
                }
            }
        }
        for (int t = 0; t < T; ++t) {
            REAL* zr = z + (size_t)t * D;
            /* out_proj, residual, norm1 */
            FN(linear_row)(att + (size_t)t * D, Wo, bo, tmp2, D, D);
            for (int i = 0; i < D; ++i) zr[i] += tmp2[i];
            FN(layernorm_row)(zr, g1, be1, D);
            /* linear2(relu(linear1(x))), residual, norm2 */
            FN(linear_row)(zr, W1, b1, tmp, F, D);
            for (int i = 0; i < F; ++i) tmp[i] = tmp[i] > 0 ? tmp[i] : (REAL)0;
            FN(linear_row)(tmp, W2, b2, tmp2, D, F);
            for (int i = 0; i < D; ++i) zr[i] += tmp2[i];
            FN(layernorm_row)(zr, g2, be2, D);
        }
        if (tap_layers) for (size_t i = 0; i < (size_t)T * D; ++i) tap_layers[(size_t)l * tap_layer_stride + i] = z[i];
    }

    const REAL* const* tw = w + 2 + 12 * L;
    if (c->with_rnn) {
        /* :98-99 nn.RNN tanh, h0 = 0: h_t = tanh(W_ih x_t + b_ih + W_hh h_{t-1} + b_hh) (torch: nn/modules/rnn.py) */
        const REAL *Wih = tw[0], *Whh = tw[1], *bih = tw[2], *bhh = tw[3], *Wout = tw[4], *bout = tw[5];
        for (int i = 0; i < R; ++i) h[i] = 0;
        for (int t = 0; t < T; ++t) {
            FN(linear_row)(z + (size_t)t * D, Wih, bih, tmp, R, D);
            FN(linear_row)(h, Whh, bhh, hn, R, R);
            for (int i = 0; i < R; ++i) h[i] = TANH(tmp[i] + hn[i]);
            if (tap_rnn) for (int i = 0; i < R; ++i) tap_rnn[(size_t)t * R + i] = h[i];
            /* :102 self.linear on every t */
            FN(linear_row)(h, Wout, bout, y + (size_t)t * S, S, R);
        }
    } else {
        const REAL *Wout = tw[0], *bout = tw[1];
        for (int t = 0; t < T; ++t) FN(linear_row)(z + (size_t)t * D, Wout, bout, y + (size_t)t * S, S, D);
    }
}

size_t FN(tip_oracle_scratch_elems)(const tip_oracle_cfg* c, int T) {
    const int In = c->n_imu_total + c->n_state, D = c->d_model, F = c->d_ff, R = c->d_rnn;
    int mx = D; if (F > mx) mx = F; if (R > mx) mx = R;
    return (size_t)In + (size_t)T * D + (size_t)T * 3 * D + (size_t)T * D + mx + D + T + 2 * (size_t)R + 64;
}

/* Batch driver.  taps are nullable: tap_in [B,T,D], tap_layers [L,B,T,D], tap_rnn [B,T,R]. */
int FN(tip_oracle_forward)(const tip_oracle_cfg* c, const REAL* const* w, const REAL* x_imu, const REAL* x_s,
                           REAL* y, int B, int T, const REAL* keep_mask, REAL keep_scale,
                           REAL* tap_in, REAL* tap_layers, REAL* tap_rnn, int nthreads) {
    if (!c || !w || !x_imu || !x_s || !y || B < 0 || T < 1) return -1;
    if (c->d_model % c->n_heads) return -2;
    const int NI = c->n_imu_total, S = c->n_state, D = c->d_model, R = c->d_rnn;
    const size_t se = FN(tip_oracle_scratch_elems)(c, T);
    if (nthreads < 1) nthreads = 1;
    REAL* scratch = (REAL*)malloc(se * sizeof(REAL) * (size_t)nthreads);
    if (!scratch) return -3;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
#ifdef _OPENMP
        REAL* sc = scratch + se * (size_t)omp_get_thread_num();
#else
        REAL* sc = scratch;
#endif
        FN(window)(c, w, x_imu + (size_t)b * T * NI, x_s + (size_t)b * T * S, y + (size_t)b * T * S, T,
                   keep_mask ? keep_mask + (size_t)b * T * S : NULL, keep_scale,
                   tap_in ? tap_in + (size_t)b * T * D : NULL,
                   tap_layers ? tap_layers + (size_t)b * T * D : NULL,
                   tap_rnn ? tap_rnn + (size_t)b * T * R : NULL,
                   (size_t)B * T * D, sc);
    }
    free(scratch);
    return 0;
}

/*
 * pxo_core.c -- ORACLE (test infrastructure, never shipped): signal
 * conversion/pooling, recurrent nets, HMM Viterbi, barcode window.
 * See pxo.h for the pinning status of each row.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -mavx2 -mfma (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: every fused multiply-add below is an explicit
 * fmaf(); everything written a*b+c is two roundings, as NumPy/TF do it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "pxo.h"

/* ------------------------------------------------------------------------ *
 * a1  DAQ counts -> pA.  fast5_file.py:130-131:
 *     np.array(range / digitisation * (rawsignal + offset), dtype=float32)
 *     int16 + python float -> float64; product in float64; one cast.
 * ------------------------------------------------------------------------ */
static inline float raw2pa(int16_t raw, double k, double offset)
{
    return (float)(k * ((double)raw + offset));
}

void pxo_raw_to_pa(const int16_t* raw, int64_t n, const pxg_calib* cal, float* out)
{
    const double k = cal->range / cal->digitisation;
    for (int64_t i = 0; i < n; i++)
        out[i] = raw2pa(raw[i], k, cal->offset);
}

/* ------------------------------------------------------------------------ *
 * NumPy's float32 add.reduce along a contiguous axis (pairwise_sum in
 * numpy/core/src/umath/loops_utils.h.src): n < 8 plain loop; n <= 128 eight
 * running partial sums, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
 * tail added one by one; larger n split in halves (multiple of 8).
 * Used by mean(axis=1, dtype=float32) at signal_loader.py:224-225,246-247.
 * ------------------------------------------------------------------------ */
float pxo_np_sum_f32(const float* a, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++)
            res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int k = 0; k < 8; k++)
            r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++)
                r[k] += a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++)
            res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pxo_np_sum_f32(a, n2) + pxo_np_sum_f32(a + n2, n - n2);
}

/* mean of one stride-block of raw samples: float32 add.reduce then a float32
 * true-divide by the count (numpy _methods._mean) */
static float block_mean(const int16_t* raw, int stride, double k, double offset)
{
    float pa[128];
    for (int j = 0; j < stride; j++)
        pa[j] = raw2pa(raw[j], k, offset);
    /* the reduction result starts from the additive identity 0.0f */
    float s = 0.0f + pxo_np_sum_f32(pa, stride);
    return s / (float)stride;
}

/* ------------------------------------------------------------------------ *
 * a2  signal_loader.py:212-231  load_padded_signal_head(30000, 15, 9000)
 * ------------------------------------------------------------------------ */
int pxo_head_pool(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                  int length_limit, int stride, int min_length, float* out)
{
    const double k = cal->range / cal->digitisation;
    int64_t L = n_raw < length_limit ? n_raw : length_limit; /* :213 */
    L -= L % stride;                                         /* :214 */
    const int n_out = length_limit / stride;
    for (int i = 0; i < n_out; i++)
        out[i] = 0.0f;
    if (L < min_length)                                      /* :220-222 */
        return PXG_ST_SCALER_SIGNAL_TOO_SHORT;
    const int n_means = (int)(L / stride);
    const int pad = n_out - n_means;                         /* left pad :227-229 */
    for (int i = 0; i < n_means; i++)
        out[pad + i] = block_mean(raw + (int64_t)i * stride, stride, k, cal->offset);
    return PXG_ST_OKAY;
}

/* ------------------------------------------------------------------------ *
 * a5  signal_loader.py:233-264  load_signal(pool=15): block means then
 *     np.poly1d(float32[scale, shift])(x) = fl(fl(scale*x) + shift)  (:262)
 * ------------------------------------------------------------------------ */
int64_t pxo_pool_scale(const int16_t* raw, int64_t n_raw, const pxg_calib* cal,
                       int stride, float scale, float shift, float* out)
{
    const double k = cal->range / cal->digitisation;
    const int64_t P = n_raw / stride;
    for (int64_t p = 0; p < P; p++) {
        float m = block_mean(raw + p * stride, stride, k, cal->offset);
        float y = scale * m;
        out[p] = y + shift;
    }
    return P;
}

/* ------------------------------------------------------------------------ *
 * Canonical float32 activation kit (DESIGN.md "Canonical LSTM arithmetic").
 * Only IEEE +,*,fma, floor and integer ops, so the HIP kernels reproduce it bit
 * for bit.
 *   sigmoid: cubic Hermite spline on 1024 segments of width 1/16 over
 *            [-32, 32); coefficients from a deterministic float64 exp (no
 *            libm), rounded to float32; interpolation error < 1e-8.
 *            u = 16*z (exact), seg = floor(u), s = u - seg (exact),
 *            sigma = c0 + s*(c1 + s*(c2 + s*c3))   (three fma)
 *   tanh(x) = fl(2*sigmoid(2x) - 1)               (one fma)
 *   expf:    Cody-Waite + degree-7 Taylor, only used by the final softmax.
 * ------------------------------------------------------------------------ */
static inline float bits2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t f2bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

#define SIG_NSEG 1024
#define SIG_HALF 512
static float g_sig_tab[SIG_NSEG][4];
static int g_sig_ready = 0;

/* exp(x) for |x| <= 40 in float64 from +,*,fma only: identical on every IEEE
 * machine (the table must be the same in the library and in this oracle) */
static double det_exp(double x)
{
    const double magic = 6755399441055744.0;        /* 1.5 * 2^52 */
    double kf = (x * 1.4426950408889634 + magic) - magic;
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double fact = 6227020800.0;                      /* 13! */
    double p = 1.0 / fact;
    for (int n = 13; n >= 1; n--) {
        fact /= (double)n;
        p = fma(p, r, 1.0 / fact);
    }
    int64_t bits;
    memcpy(&bits, &p, 8);
    bits += (int64_t)kf << 52;
    memcpy(&p, &bits, 8);
    return p;
}

void pxo_sigmoid_table(float* out)
{
    const double h = 1.0 / 16.0;
    for (int i = 0; i < SIG_NSEG; i++) {
        const double z0 = (double)(i - SIG_HALF) * h, z1 = z0 + h;
        const double s0 = 1.0 / (1.0 + det_exp(-z0)), s1 = 1.0 / (1.0 + det_exp(-z1));
        const double d0 = s0 * (1.0 - s0), d1 = s1 * (1.0 - s1);
        out[4 * i + 0] = (float)s0;
        out[4 * i + 1] = (float)(h * d0);
        out[4 * i + 2] = (float)(3.0 * (s1 - s0) - h * (2.0 * d0 + d1));
        out[4 * i + 3] = (float)(2.0 * (s0 - s1) + h * (d0 + d1));
    }
}

/* Position of a look-up argument inside its table segment: u - floor(u), kept BELOW 1 -- for a tiny negative u
 * (-2^-25 < u < 0) the float subtraction u - (-1) rounds to 1.0, which is not a position inside segment -1; the
 * definition takes the largest float below 1 there.  (This is what the GPU's v_fract_f32 returns for every float,
 * checked exhaustively: tools/ubench/fract_check.hip; rounds 1-3 evaluated the segment's cubic AT 1.0 in that case,
 * about twice per 10 000-read batch.) */
static inline float sig_position(float u, float fl)
{
    const float s = u - fl;
    return s >= 1.0f ? 0.99999994f : s;
}

/* sigmoid(zscale/16 * z): zscale = 16 for the gates, 32 inside tanh */
static inline float sig_lookup(float z, float zscale, float zlo, float zhi)
{
    if (!g_sig_ready) {
        pxo_sigmoid_table(&g_sig_tab[0][0]);
        g_sig_ready = 1;
    }
    z = fminf(fmaxf(z, zlo), zhi);
    const float u = z * zscale;
    const float fl = floorf(u);
    const float s = sig_position(u, fl);
    const float* c = g_sig_tab[(int)fl + SIG_HALF];
    float p = fmaf(c[3], s, c[2]);
    p = fmaf(p, s, c[1]);
    return fmaf(p, s, c[0]);
}

float pxo_sigmoid(float x)
{
    return sig_lookup(x, 16.0f, -32.0f, 31.999998f);
}

float pxo_tanh(float x)
{
    const float s = sig_lookup(x, 32.0f, -16.0f, 15.999999f);
    return fmaf(2.0f, s, -1.0f);
}

float pxo_expf(float x)
{
    x = fminf(fmaxf(x, -87.0f), 87.0f);
    const float magic = 12582912.0f; /* 1.5 * 2^23: round-to-nearest-even */
    float t = fmaf(x, 1.44269504088896341f, magic);
    float n = t - magic;
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.98412698412698413e-4f;           /* 1/5040 */
    p = fmaf(p, r, 1.38888888888888894e-3f);     /* 1/720  */
    p = fmaf(p, r, 8.33333333333333322e-3f);     /* 1/120  */
    p = fmaf(p, r, 4.16666666666666644e-2f);     /* 1/24   */
    p = fmaf(p, r, 1.66666666666666657e-1f);     /* 1/6    */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return bits2f(f2bits(p) + ((int32_t)n << 23));
}

/* ------------------------------------------------------------------------ *
 * One Keras LSTM cell step (recurrent_v2 / LSTMCell.call; gate blocks
 * i,f,c,o; activation tanh, recurrent_activation sigmoid):
 *     z = x.W + h.U + b ; i,f,o = sigmoid ; g = tanh(z_c)
 *     c' = f*c + i*g ; h' = o*tanh(c')
 * Canonical evaluation order (what the MFMA kernels reproduce):
 *   accumulator start  scalar input: fl(fl(x*W_j) + b_j)
 *                      vector input: b_j
 *   then ONE fma chain over the input rows k = 0.. (vector input only)
 *   followed by the recurrent rows k = 0..units-1.
 *   c' = fl(fl(f*c) + fl(i*g)),  h' = fl(o * tanh(c')).
 * ------------------------------------------------------------------------ */
static void lstm_step(const pxg_lstm_layer* L, const float* x, float* h, float* c,
                      float* z)
{
    const int H = L->units, G = 4 * H, I = L->input_dim;
    if (I == 1) {
        const float xv = x[0];
        for (int j = 0; j < G; j++) {
            float xw = xv * L->kernel[j];
            z[j] = xw + L->bias[j];
        }
    } else {
        for (int j = 0; j < G; j++)
            z[j] = L->bias[j];
        for (int k = 0; k < I; k++) {
            const float xk = x[k];
            const float* w = L->kernel + (size_t)k * G;
            for (int j = 0; j < G; j++)
                z[j] = fmaf(xk, w[j], z[j]);
        }
    }
    for (int k = 0; k < H; k++) {
        const float hk = h[k];
        const float* u = L->recurrent + (size_t)k * G;
        for (int j = 0; j < G; j++)
            z[j] = fmaf(hk, u[j], z[j]);
    }
    for (int u = 0; u < H; u++) {
        float ig = pxo_sigmoid(z[u]);
        float fg = pxo_sigmoid(z[H + u]);
        float gg = pxo_tanh(z[2 * H + u]);
        float og = pxo_sigmoid(z[3 * H + u]);
        float fc = fg * c[u];
        float in = ig * gg;
        float cn = fc + in;
        c[u] = cn;
        h[u] = og * pxo_tanh(cn);
    }
}

void pxo_lstm_layer(const pxg_lstm_layer* L, const float* x, int T, int reverse,
                    float* seq_out, float* h_last)
{
    const int H = L->units, I = L->input_dim;
    float* h = (float*)calloc(H, sizeof(float));
    float* c = (float*)calloc(H, sizeof(float));
    float* z = (float*)malloc(sizeof(float) * 4 * H);
    for (int s = 0; s < T; s++) {
        const int t = reverse ? T - 1 - s : s;
        lstm_step(L, x + (size_t)t * I, h, c, z);
        if (seq_out)
            memcpy(seq_out + (size_t)t * H, h, sizeof(float) * H);
    }
    if (h_last)
        memcpy(h_last, h, sizeof(float) * H);
    free(h); free(c); free(z);
}

/* ------------------------------------------------------------------------ *
 * Canonical arithmetic "q8" (pxg_config.lstm_arith == PXG_LSTM_Q8).
 * The Keras equations above with the matrix products evaluated in EXACT fixed point, so that
 * the result does not depend on the order in which a machine adds the products:
 *   hidden state  h_f = fl(o * tanh(c')) as before, then q = rint(h_f * 2^22) (ties to even),
 *                 |q| <= 2^22; everything downstream sees h = q * 2^-22.
 *   weights       per layer and GATE BLOCK (the i, f, c, o columns), all rows that multiply an h
 *                 (vector-input rows, then recurrent rows) share one exponent p = the largest
 *                 integer with max|W_block| * 2^p <= 8355711 (= 127 + 127*256 + 127*65536);
 *                 Wq = rint(W * 2^p).  (Round 4 had one p per layer; per block the coherent
 *                 2^-23-of-the-largest-weight perturbation is taken against each block's own
 *                 largest weight: tools/decision_flips_gpu.py)
 *   digits        v = d0 + 256 d1 + 65536 d2 with every d in [-128, 127] (balanced base 256,
 *                 unique); w0..w2 of Wq, h0..h2 of q.
 *   products      the eight leading digit products, summed EXACTLY (they are small integers):
 *                   A0 = sum_k w2 h2                      (weight 2^32)
 *                   A1 = sum_k w2 h1 + w1 h2              (weight 2^24)
 *                   A2 = sum_k w2 h0 + w1 h1 + w0 h2      (weight 2^16)
 *                   A3 = sum_k w1 h0 + w0 h1              (weight 2^8)
 *                 (w0 h0, below 2^-29 of a full-scale product, is dropped).
 *   pre-activation, in table units (16 z for i, f, o; 32 z for the cell candidate):
 *                   V = A0 * 256 + A1,  U = A2 * 256 + A3   (int32; no overflow for K <= 192:
 *                                                            |h2| <= 64, every other digit <= 128)
 *                   t = fma((float)V, 65536, (float)U)      -- both conversions round to nearest even
 *                   u = fma(t, g * 2^(-p-14), start),  start = fma(x, g W_x, g b) (scalar input)
 *                                                            or g b          (vector input)
 *   gates         the spline sigmoid on u as above; C = 32 c carried; identical cell update.
 * Dense layers read h = q * 2^-22 (exact in float32) and keep their fma chain.
 * Here the integer sums are accumulated in float32 lanes -- every partial sum is an integer
 * below 2^24, so each operation is exact and the compiler may vectorise freely.
 * ------------------------------------------------------------------------ */
#define Q_HSCALE 4194304.0f        /* 2^22 */
#define Q_WMAX 8355711.0

typedef struct {
    int K, G, p[4];                /* p: one exponent per gate block */
    float* w[3];                   /* digit planes [K][G] as float (exact small integers) */
    float S[4];                    /* per gate block: g * 2^(-p-14) */
} qmat;

static void q_digits(int32_t v, int32_t d[3])
{
    d[0] = (int8_t)(uint8_t)(v & 255);
    const int32_t r1 = (v - d[0]) / 256;           /* exact */
    d[1] = (int8_t)(uint8_t)(r1 & 255);
    d[2] = (r1 - d[1]) / 256;                       /* exact; in [-128, 127] for |v| <= 8355711 */
}

static qmat* qmat_build(const pxg_lstm_layer* L)
{
    const int H = L->units, G = 4 * H, I = L->input_dim;
    const int Kin = I == 1 ? 0 : I, K = Kin + H;
    qmat* M = (qmat*)calloc(1, sizeof(qmat));
    M->K = K; M->G = G;
    /* one exponent per gate block (i, f, c, o): the largest with max|W_gate| * 2^p <= Q_WMAX */
    for (int g = 0; g < 4; g++) {
        double m = 0.0;
        for (int k = 0; k < K; k++) {
            const float* row = k < Kin ? L->kernel + (size_t)k * G : L->recurrent + (size_t)(k - Kin) * G;
            for (int j = g * H; j < (g + 1) * H; j++)
                if (fabs((double)row[j]) > m) m = fabs((double)row[j]);
        }
        int p = 0;
        if (m > 0.0) {
            p = 40;
            while (ldexp(m, p) > Q_WMAX) p--;
        }
        M->p[g] = p;
    }
    for (int d = 0; d < 3; d++) M->w[d] = (float*)malloc(sizeof(float) * (size_t)K * G);
    for (int k = 0; k < K; k++) {
        const float* row = k < Kin ? L->kernel + (size_t)k * G : L->recurrent + (size_t)(k - Kin) * G;
        for (int j = 0; j < G; j++) {
            int32_t dg[3];
            q_digits((int32_t)rint(ldexp((double)row[j], M->p[j / H])), dg);
            for (int d = 0; d < 3; d++) M->w[d][(size_t)k * G + j] = (float)dg[d];
        }
    }
    for (int g = 0; g < 4; g++) M->S[g] = (float)ldexp(g == 2 ? 32.0 : 16.0, -M->p[g] - 14);
    return M;
}

static void qmat_free(qmat* M)
{
    for (int d = 0; d < 3; d++) free(M->w[d]);
    free(M);
}

/* the digit planes of a layer are built once per thread and kept (8 slots, keyed by a hash of
 * the weights themselves: the same addresses may carry other weights later) */
static uint64_t q_hash(const pxg_lstm_layer* L)
{
    const int G = 4 * L->units;
    uint64_t h = 1469598103934665603ull ^ (uint64_t)L->units ^ ((uint64_t)L->input_dim << 20);
    const size_t n[3] = { (size_t)L->input_dim * G, (size_t)L->units * G, (size_t)G };
    const float* a[3] = { L->kernel, L->recurrent, L->bias };
    for (int i = 0; i < 3; i++)
        for (size_t k = 0; k < n[i]; k++) {
            uint32_t b;
            memcpy(&b, a[i] + k, 4);
            h = (h ^ b) * 1099511628211ull;
        }
    return h;
}

static qmat* qmat_get(const pxg_lstm_layer* L)
{
    static __thread struct { uint64_t key; qmat* M; } slot[8];
    static __thread int next;
    const uint64_t key = q_hash(L);
    for (int i = 0; i < 8; i++)
        if (slot[i].M && slot[i].key == key) return slot[i].M;
    const int i = next;
    next = (next + 1) & 7;
    if (slot[i].M) qmat_free(slot[i].M);
    slot[i].key = key;
    slot[i].M = qmat_build(L);
    return slot[i].M;
}

/* spline sigmoid of an argument already in table units (u = 16 z) */
static inline float sig_lookup_u(float u)
{
    if (!g_sig_ready) {
        pxo_sigmoid_table(&g_sig_tab[0][0]);
        g_sig_ready = 1;
    }
    u = fminf(fmaxf(u, -512.0f), 511.99997f);
    const float fl = floorf(u);
    const float s = sig_position(u, fl);
    const float* c = g_sig_tab[(int)fl + SIG_HALF];
    float p = fmaf(c[3], s, c[2]);
    p = fmaf(p, s, c[1]);
    return fmaf(p, s, c[0]);
}

/* one step: qin = the vector input's q values (NULL for a scalar input xs), qh / C = state,
 * scratch = 4 * G floats */
static void lstm_step_q(const qmat* M, const pxg_lstm_layer* L, const float* xs, const int32_t* qin,
                        int32_t* qh, float* C, float* scratch)
{
    const int H = L->units, G = M->G, K = M->K, Kin = K - H;
    float *a0 = scratch, *a1 = scratch + G, *a2 = scratch + 2 * G, *a3 = scratch + 3 * G, *u = scratch + 4 * G;
    memset(scratch, 0, sizeof(float) * 4 * (size_t)G);
    for (int k = 0; k < K; k++) {
        int32_t d[3];
        q_digits(k < Kin ? qin[k] : qh[k - Kin], d);
        const float h0 = (float)d[0], h1 = (float)d[1], h2 = (float)d[2];
        const float *w0 = M->w[0] + (size_t)k * G, *w1 = M->w[1] + (size_t)k * G, *w2 = M->w[2] + (size_t)k * G;
        for (int j = 0; j < G; j++) {          /* exact: integers below 2^24 */
            a0[j] = fmaf(w2[j], h2, a0[j]);
            a1[j] = fmaf(w2[j], h1, fmaf(w1[j], h2, a1[j]));
            a2[j] = fmaf(w2[j], h0, fmaf(w1[j], h1, fmaf(w0[j], h2, a2[j])));
            a3[j] = fmaf(w1[j], h0, fmaf(w0[j], h1, a3[j]));
        }
    }
    for (int j = 0; j < G; j++) {
        const int gate = j / H;
        const float g = gate == 2 ? 32.0f : 16.0f;
        const float start = L->input_dim == 1 ? fmaf(xs[0], L->kernel[j] * g, L->bias[j] * g) : L->bias[j] * g;
        const int32_t V = (int32_t)a0[j] * 256 + (int32_t)a1[j];
        const int32_t U = (int32_t)a2[j] * 256 + (int32_t)a3[j];
        const float t = fmaf((float)V, 65536.0f, (float)U);
        u[j] = fmaf(t, M->S[gate], start);
    }
    for (int n = 0; n < H; n++) {
        const float ig = sig_lookup_u(u[n]);
        const float fg = sig_lookup_u(u[H + n]);
        const float Gc = fmaf(64.0f, sig_lookup_u(u[2 * H + n]), -32.0f);
        const float og = sig_lookup_u(u[3 * H + n]);
        const float fc = fg * C[n];
        const float in = ig * Gc;
        const float cn = fc + in;
        C[n] = cn;
        const float th = fmaf(2.0f, sig_lookup_u(cn), -1.0f);
        const float hf = og * th;
        qh[n] = (int32_t)rintf(hf * Q_HSCALE);
    }
}

/* one layer over a sequence; x: float scalars [T] (input_dim 1) or q values [T][I];
 * seq_out (q values, [T][H], stored at the original time index) and q_last may be NULL */
void pxo_lstm_layer_q(const pxg_lstm_layer* L, const float* xs, const int32_t* xq, int T, int reverse,
                      int32_t* seq_out, int32_t* q_last)
{
    const int H = L->units, I = L->input_dim;
    const qmat* M = qmat_get(L);
    int32_t* qh = (int32_t*)calloc(H, sizeof(int32_t));
    float* C = (float*)calloc(H, sizeof(float));
    float* scratch = (float*)malloc(sizeof(float) * 5 * (size_t)M->G);
    for (int s = 0; s < T; s++) {
        const int t = reverse ? T - 1 - s : s;
        lstm_step_q(M, L, I == 1 ? xs + t : NULL, I == 1 ? NULL : xq + (size_t)t * I, qh, C, scratch);
        if (seq_out)
            memcpy(seq_out + (size_t)t * H, qh, sizeof(int32_t) * H);
    }
    if (q_last)
        memcpy(q_last, qh, sizeof(int32_t) * H);
    free(qh); free(C); free(scratch);
}

static void q_to_float(const int32_t* q, int n, float* out)
{
    for (int i = 0; i < n; i++) out[i] = (float)q[i] * (1.0f / Q_HSCALE);    /* exact */
}

static void dense_forward(const pxg_dense_layer* D, const float* x, float* out)
{
    for (int j = 0; j < D->out_dim; j++) {
        float acc = D->bias[j];
        for (int k = 0; k < D->in_dim; k++)
            acc = fmaf(x[k], D->kernel[(size_t)k * D->out_dim + j], acc);
        out[j] = acc;
    }
}

/* a4  scaler-r3: LSTM(48, seq) -> LSTM(48) -> Dense(2, linear); Dropout is
 *     identity at inference */
void pxo_scaler_forward(const pxg_config* cfg, const float* x, int T, float* pred)
{
    const int H1 = cfg->scaler_lstm1.units, H2 = cfg->scaler_lstm2.units;
    if (cfg->lstm_arith == PXG_LSTM_Q8) {
        int32_t* seq = (int32_t*)malloc(sizeof(int32_t) * (size_t)T * H1);
        int32_t* q2 = (int32_t*)malloc(sizeof(int32_t) * H2);
        float* h2 = (float*)malloc(sizeof(float) * H2);
        pxo_lstm_layer_q(&cfg->scaler_lstm1, x, NULL, T, 0, seq, NULL);
        pxo_lstm_layer_q(&cfg->scaler_lstm2, NULL, seq, T, 0, NULL, q2);
        q_to_float(q2, H2, h2);
        dense_forward(&cfg->scaler_dense, h2, pred);
        free(seq); free(q2); free(h2);
        return;
    }
    float* seq = (float*)malloc(sizeof(float) * (size_t)T * H1);
    float* h2 = (float*)malloc(sizeof(float) * H2);
    pxo_lstm_layer(&cfg->scaler_lstm1, x, T, 0, seq, NULL);
    pxo_lstm_layer(&cfg->scaler_lstm2, seq, T, 0, NULL, h2);
    dense_forward(&cfg->scaler_dense, h2, pred);
    free(seq); free(h2);
}

/* a4  signal_loader.py:98-109.  poly1d([std, mean]) with float64 coefficients
 *     applied to a float32 array under NumPy-1.x value-based casting stays
 *     float32: fl(fl(fl32(std)*p) + fl32(mean)); the QC bounds are float64
 *     scalars compared against a float32 ARRAY, hence rounded to float32
 *     first; both ends inclusive (:70-73). */
int pxo_scaler_transform(const pxg_config* cfg, const float* pred, float* ss)
{
    const float s_mean = (float)cfg->scaler_xfrm[0], s_std = (float)cfg->scaler_xfrm[1];
    const float h_mean = (float)cfg->scaler_xfrm[2], h_std = (float)cfg->scaler_xfrm[3];
    float a = s_std * pred[0];
    float scale = a + s_mean;
    float b = h_std * pred[1];
    float shift = b + h_mean;
    ss[0] = scale;
    ss[1] = shift;
    const int ok = scale >= (float)cfg->scaler_qc_scale[0] &&
                   scale <= (float)cfg->scaler_qc_scale[1] &&
                   shift >= (float)cfg->scaler_qc_shift[0] &&
                   shift <= (float)cfg->scaler_qc_shift[1];
    return ok ? PXG_ST_OKAY : PXG_ST_SCALING_QC_FAIL;
}

/* a12 demux-tetra-r4: Bidirectional(LSTMCell48, concat, seq) -> LSTMCell64 ->
 *     Dense(5, softmax); GaussianNoise/Dropout identity.
 *     softmax = exp(z - max) / sum, sum taken left to right. */
void pxo_demux_forward(const pxg_config* cfg, const float* x, int T, float* probs)
{
    const int Hf = cfg->demux_fwd.units, Hb = cfg->demux_bwd.units;
    const int Ht = cfg->demux_top.units, C = cfg->demux_dense.out_dim;
    float* sf = (float*)malloc(sizeof(float) * (size_t)T * Hf);
    float* sb = (float*)malloc(sizeof(float) * (size_t)T * Hb);
    float* cat = (float*)malloc(sizeof(float) * (size_t)T * (Hf + Hb));
    float* ht = (float*)malloc(sizeof(float) * Ht);
    float z[PXG_MAX_CLASSES], e[PXG_MAX_CLASSES];
    if (cfg->lstm_arith == PXG_LSTM_Q8) {
        int32_t* qf = (int32_t*)malloc(sizeof(int32_t) * (size_t)T * Hf);
        int32_t* qb = (int32_t*)malloc(sizeof(int32_t) * (size_t)T * Hb);
        int32_t* qc = (int32_t*)malloc(sizeof(int32_t) * (size_t)T * (Hf + Hb));
        int32_t* qt = (int32_t*)malloc(sizeof(int32_t) * Ht);
        pxo_lstm_layer_q(&cfg->demux_fwd, x, NULL, T, 0, qf, NULL);
        pxo_lstm_layer_q(&cfg->demux_bwd, x, NULL, T, 1, qb, NULL);
        for (int t = 0; t < T; t++) {
            memcpy(qc + (size_t)t * (Hf + Hb), qf + (size_t)t * Hf, sizeof(int32_t) * Hf);
            memcpy(qc + (size_t)t * (Hf + Hb) + Hf, qb + (size_t)t * Hb, sizeof(int32_t) * Hb);
        }
        pxo_lstm_layer_q(&cfg->demux_top, NULL, qc, T, 0, NULL, qt);
        q_to_float(qt, Ht, ht);
        free(qf); free(qb); free(qc); free(qt);
    } else {
    pxo_lstm_layer(&cfg->demux_fwd, x, T, 0, sf, NULL);
    pxo_lstm_layer(&cfg->demux_bwd, x, T, 1, sb, NULL);
    for (int t = 0; t < T; t++) {
        memcpy(cat + (size_t)t * (Hf + Hb), sf + (size_t)t * Hf, sizeof(float) * Hf);
        memcpy(cat + (size_t)t * (Hf + Hb) + Hf, sb + (size_t)t * Hb, sizeof(float) * Hb);
    }
    pxo_lstm_layer(&cfg->demux_top, cat, T, 0, NULL, ht);
    }
    dense_forward(&cfg->demux_dense, ht, z);
    float m = z[0];
    for (int j = 1; j < C; j++)
        if (z[j] > m)
            m = z[j];
    float s = 0.0f;
    for (int j = 0; j < C; j++) {
        e[j] = pxo_expf(z[j] - m);
        s = (j == 0) ? e[0] : s + e[j];
    }
    for (int j = 0; j < C; j++)
        probs[j] = e[j] / s;   /* IEEE division, once per read */
    free(sf); free(sb); free(cat); free(ht);
}

/* ------------------------------------------------------------------------ *
 * a6/a7  HMM + Viterbi, pomegranate >= 0.10 semantics:
 *   NormalDistribution.log_probability(x) =
 *        -log(sigma*sqrt(2*pi)) - (x-mu)^2 * (1/(2*sigma^2))        (float64)
 *   GeneralMixtureModel: lp = NEGINF; for each component
 *        lp = pair_lse(lp, logpdf_k(x) + log(w_k / sum w))
 *   pair_lse(a,b) = b if a==-inf; a if b==-inf;
 *                   a + log(exp(b-a)+1) if a > b else b + log(exp(a-b)+1)
 *   viterbi: v[0][s] = log(pi_s) + e_s(x_0);
 *            v[t][s] = max_k(v[t-1][k] + log a_ks)  (strict '>' scanning the
 *            in-edges in name-sorted source order, start from -inf) + e_s(x_t)
 *   no end state: terminate at the first maximum of the last column scanned in
 *   name-sorted order; trace back.
 * ------------------------------------------------------------------------ */
#define SQRT_2_PI 2.50662827463

static double pair_lse(double a, double b)
{
    if (a == INFINITY || b == INFINITY)
        return INFINITY;
    if (a == -INFINITY)
        return b;
    if (b == -INFINITY)
        return a;
    if (a > b)
        return a + log(exp(b - a) + 1.0);
    return b + log(exp(a - b) + 1.0);
}

double pxo_hmm_emission(const pxg_hmm* hmm, int s, double x)
{
    const int nm = hmm->n_mix[s];
    if (nm == 1) {
        const double sd = hmm->mix_sigma[s][0];
        const double lssp = -log(sd * SQRT_2_PI);
        const double tss = 1.0 / (2.0 * sd * sd);
        const double d = x - hmm->mix_mu[s][0];
        return lssp - (d * d) * tss;
    }
    double wsum = 0.0;
    for (int k = 0; k < nm; k++)
        wsum += hmm->mix_weight[s][k];
    double lp = -INFINITY;
    for (int k = 0; k < nm; k++) {
        const double sd = hmm->mix_sigma[s][k];
        const double lssp = -log(sd * SQRT_2_PI);
        const double tss = 1.0 / (2.0 * sd * sd);
        const double d = x - hmm->mix_mu[s][k];
        const double l = lssp - (d * d) * tss;
        lp = pair_lse(lp, l + log(hmm->mix_weight[s][k] / wsum));
    }
    return lp;
}

double pxo_viterbi(const pxg_hmm* hmm, const float* x, int T, int32_t* path)
{
    const int S = hmm->n_states;
    int order[PXG_MAX_STATES]; /* states in name-sorted order */
    for (int s = 0; s < S; s++)
        order[hmm->name_rank[s]] = s;
    double logtr[PXG_MAX_STATES][PXG_MAX_STATES];
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++)
            logtr[i][j] = hmm->trans[i][j] > 0.0 ? log(hmm->trans[i][j]) : -INFINITY;
    int8_t* bp = (int8_t*)malloc((size_t)(T > 0 ? T : 1) * S);
    double v[PXG_MAX_STATES], nv[PXG_MAX_STATES];
    if (T <= 0) {
        free(bp);
        return -INFINITY;
    }
    for (int s = 0; s < S; s++) {
        double lp = hmm->start_prob[s] > 0.0 ? log(hmm->start_prob[s]) : -INFINITY;
        double best = -INFINITY;
        if (lp > best)
            best = lp;
        v[s] = best + pxo_hmm_emission(hmm, s, (double)x[0]);
        bp[s] = -1;
    }
    for (int t = 1; t < T; t++) {
        const double xt = (double)x[t];
        for (int s = 0; s < S; s++) {
            double best = -INFINITY;
            int arg = -1;
            for (int r = 0; r < S; r++) {
                const int k = order[r];
                if (hmm->trans[k][s] <= 0.0)
                    continue;
                const double cand = v[k] + logtr[k][s];
                if (cand > best) {
                    best = cand;
                    arg = k;
                }
            }
            nv[s] = best + pxo_hmm_emission(hmm, s, xt);
            bp[(size_t)t * S + s] = (int8_t)arg;
        }
        memcpy(v, nv, sizeof(double) * S);
    }
    int end = order[0];
    for (int r = 1; r < S; r++)
        if (v[order[r]] > v[end])
            end = order[r];
    const double logp = v[end];
    int s = end;
    for (int t = T - 1; t >= 0; t--) {
        path[t] = s;
        if (t > 0) {
            int p = bp[(size_t)t * S + s];
            s = p < 0 ? s : p;
        }
    }
    free(bp);
    return logp;
}

/* a8  signal_analyzer.py:354-362: groupby runs of equal states; dict
 *     name -> (first, last) right-inclusive; a later run overwrites. */
void pxo_segments(const int32_t* path, int T, int32_t* seg_first, int32_t* seg_last)
{
    for (int s = 0; s < PXG_N_SEGMENTS; s++)
        seg_first[s] = seg_last[s] = -1;
    int t = 0;
    while (t < T) {
        int e = t;
        while (e + 1 < T && path[e + 1] == path[t])
            e++;
        seg_first[path[t]] = t;
        seg_last[path[t]] = e;
        t = e + 1;
    }
}

/* ------------------------------------------------------------------------ *
 * a11 barcoding.py:77-81
 *     med = np.median(sig)  (float32; even n: fl(fl(a+b)/2))
 *     mad = np.median(|sig - med|)
 *     (sig - med) / max(0.01, mad * 1.4826)
 *       mad*1.4826 is float32-scalar * python-float = float64 (NumPy 1.x);
 *       max() keeps the double; array / double-scalar casts it to float32.
 * ------------------------------------------------------------------------ */
static int cmp_f32(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

static float median_f32(const float* v, int n, float* tmp)
{
    memcpy(tmp, v, sizeof(float) * n);
    qsort(tmp, n, sizeof(float), cmp_f32);
    /* a zero median is +0: which of several +-0 samples a partition (NumPy's introselect, qsort
     * here, the GPU's radix select) leaves in the middle is not defined by any of them */
    float m;
    if (n & 1) {
        m = tmp[n / 2];
    } else {
        const float s = tmp[n / 2 - 1] + tmp[n / 2];
        m = s / 2.0f;
    }
    return m == 0.0f ? 0.0f : m;
}

void pxo_normalize_signal(const float* sig, int n, float* out)
{
    float* tmp = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    float* dev = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    const float med = median_f32(sig, n, tmp);
    for (int i = 0; i < n; i++)
        dev[i] = fabsf(sig[i] - med);
    const float mad = median_f32(dev, n, tmp);
    const double dd = fmax(0.01, (double)mad * 1.4826);
    const float div = (float)dd;
    for (int i = 0; i < n; i++)
        out[i] = (sig[i] - med) / div;
    free(tmp); free(dev);
}

/* a9+a10  signal_analyzer.py:445-448 + barcoding.py:83-101 */
int pxo_barcode_window(const pxg_config* cfg, const float* sig, int len, float* out)
{
    const int trim = cfg->signal_trim_length;
    if (len <= 0)                                              /* :447 */
        return 0;
    if (!(cfg->minimum_dna_length <= len && len <= cfg->maximum_dna_length))
        return 0;                                              /* :87-88 */
    if (len > trim) {
        pxo_normalize_signal(sig + (len - trim), trim, out);   /* :91-92 */
    } else if (len < trim) {
        const int pad = trim - len;                            /* :93-96 */
        for (int i = 0; i < pad; i++)
            out[i] = cfg->pad_filler;
        pxo_normalize_signal(sig, len, out + pad);
    } else {
        pxo_normalize_signal(sig, len, out);
    }
    return 1;
}

/* a13 barcoding.py:72-75: bisect_right over the float64 calibration list,
 *     the float32 score promoted to double for every comparison. */
int pxo_phred(const pxg_config* cfg, float score)
{
    if (score <= 0.0f)
        return 0;
    const double x = (double)score;
    int lo = 0, hi = cfg->n_calibration;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < cfg->calibration[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

/* a12 tail  barcoding.py:108-118 */
void pxo_barcode_call(const pxg_config* cfg, const float* probs, pxg_read_result* r)
{
    const int C = cfg->demux_dense.out_dim;
    int arg = 0;
    for (int j = 1; j < C; j++)
        if (probs[j] > probs[arg])
            arg = j;
    const int label = arg - cfg->number_of_decoy_labels;
    const float score = probs[arg];
    for (int j = 0; j < PXG_MAX_CLASSES; j++)
        r->probs[j] = j < C ? probs[j] : 0.0f;
    r->bc_label = (int8_t)label;
    r->bc_score = score;
    r->bc_called = (label >= 0 && (double)score >= cfg->score_threshold) ? 1 : 0;
    r->bc_phred = (uint8_t)pxo_phred(cfg, score);
}

/* oracle/zv_oracle.c — TEST INFRASTRUCTURE ONLY (see zv_oracle.h).
 *
 * CPU restatement of what the reference computes on its ggml CPU backend.  Every function cites
 * the reference lines it follows.  Activations are kept in the reference's own layouts:
 *   cf  = channels-first  [C][L]  (ggml ne = [L, C]; what ggml_conv_1d consumes/produces)
 *   tm  = token-major     [N][E]  (ggml ne = [E, N]; what the encoder's linear layers use)
 *
 * Numeric contract (distilled from ggml):
 *   conv  : operands f16 (activations rounded RNE by im2col, ggml.c:3776 + ggml-cpu.c:9952), products
 *           exact in f32, f32 accumulation; bias added afterwards in f32
 *   linear/attention: pure f32 (FMA accumulation on an AVX2+FMA build)
 *   norm  : mean / biased variance over the contiguous axis accumulated in double, 1/sqrtf(var+eps)
 *   softmax: max-subtracted, exp by ggml_v_expf on full 8-lane groups + libm expf on the tail,
 *           sum in double
 */
#include "zv_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__F16C__)
#include <immintrin.h>
#endif

/* built with -ffp-contract=off (oracle/Makefile): the restatement must not be contracted or
 * re-associated by the compiler; fused multiply-adds are written explicitly (fmaf) where ggml has them */

#define ZVO_MAX_TENSORS 1024

typedef struct
{
    char        name[64];
    const void *data;
    int         dtype;
    int64_t     ne[4];
} zvo_tensor;

struct zvo_ctx
{
    zvo_tensor t[ZVO_MAX_TENSORS];
    int        n;
    int        order;
    int        f16_inputs;
    int        threads;
};

static __thread char g_err[256];

static int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

const char *zvo_last_error(void) { return g_err; }

zvo_ctx *zvo_new(void)
{
    zvo_ctx *c = (zvo_ctx *)calloc(1, sizeof(zvo_ctx));
    c->order = ZVO_ORDER_GGML_AVX2;
    c->f16_inputs = 1;
    c->threads = 0;
    return c;
}

void zvo_free(zvo_ctx *c) { free(c); }
void zvo_set_order(zvo_ctx *c, int order) { c->order = order; }
void zvo_set_f16_inputs(zvo_ctx *c, int on) { c->f16_inputs = on; }
void zvo_set_threads(zvo_ctx *c, int n) { c->threads = n; }

int zvo_set_tensor(zvo_ctx *c, const char *name, const void *data, int dtype, int n_dims, const int64_t *ne)
{
    if (c->n >= ZVO_MAX_TENSORS) return fail("too many tensors");
    if (strlen(name) >= 64) return fail("tensor name too long: %s", name);
    zvo_tensor *t = &c->t[c->n++];
    strcpy(t->name, name);
    t->data = data;
    t->dtype = dtype;
    for (int i = 0; i < 4; i++) t->ne[i] = i < n_dims ? ne[i] : 1;
    return 0;
}

/* checked_get_tensor, reference src/utils.cpp:9-17 (error instead of throw) */
static const zvo_tensor *get(zvo_ctx *c, const char *fmt, ...)
{
    char name[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof(name), fmt, ap);
    va_end(ap);
    for (int i = 0; i < c->n; i++)
        if (strcmp(c->t[i].name, name) == 0) return &c->t[i];
    fail("tensor '%s' not found", name);
    return NULL;
}

static void apply_threads(zvo_ctx *c)
{
#ifdef _OPENMP
    if (c->threads > 0) omp_set_num_threads(c->threads);
#else
    (void)c;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* f16 <-> f32: ggml uses F16C's _cvtss_sh(x, 0) / _cvtsh_ss (ggml-impl.h:345-346): RNE        */

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline float f16_to_f32(uint16_t h)
{
#if defined(__F16C__)
    return _cvtsh_ss(h);
#else
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF;
    if (exp == 0)
    {
        if (man == 0) return bits_f32(sign);
        int e = -1;
        do { e++; man <<= 1; } while ((man & 0x400) == 0);
        return bits_f32(sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13));
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 127 - 15) << 23) | (man << 13));
#endif
}

static inline uint16_t f32_to_f16(float f)
{
#if defined(__F16C__)
    return _cvtss_sh(f, 0);
#else
    uint32_t x = f32_bits(f), sign = (x >> 16) & 0x8000;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00 | (x > 0x7F800000u ? 0x200 | ((x >> 13) & 0x3FF) : 0));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00);                 /* rounds to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                             /* rounds to zero */
    int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7FFFFFu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t half = 1u << (shift - 1), rem = man & ((1u << shift) - 1);
    uint32_t r = man >> shift;
    if (rem > half || (rem == half && (r & 1))) r++;
    if (e < -14) return (uint16_t)(sign | r);                               /* subnormal (r may carry into exp) */
    return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (r - 0x400)));
#endif
}

static inline float round_f16(float f) { return f16_to_f32(f32_to_f16(f)); }

/* ------------------------------------------------------------------------------------------ */
/* dot products                                                                                */

/* GGML_F32x8_REDUCE (ggml-cpu.c:675-693): acc0+=acc2, acc1+=acc3, acc0+=acc1, lo+hi, hadd, hadd */
static inline float reduce_4x8(float acc[4][8])
{
    float a[8], t0[4];
    for (int l = 0; l < 8; l++) a[l] = (acc[0][l] + acc[2][l]) + (acc[1][l] + acc[3][l]);
    for (int l = 0; l < 4; l++) t0[l] = a[l] + a[l + 4];
    return (t0[0] + t0[1]) + (t0[2] + t0[3]);
}

/* ggml_vec_dot_f16 on AVX2 (ggml-cpu.c:1463-1503): operands are f16 values held as float here.
 * products of two f16 are exact in f32, so fma == mul+add and no FMA is needed for exactness.   */
static float dot_f16ops(const float *x, const float *y, int n, int order)
{
    if (order == ZVO_ORDER_GGML_AVX2)
    {
        float acc[4][8];
        memset(acc, 0, sizeof(acc));
        const int np = n & ~31;
        for (int i = 0; i < np; i += 32)
            for (int a = 0; a < 4; a++)
                for (int l = 0; l < 8; l++)
                    acc[a][l] += x[i + 8 * a + l] * y[i + 8 * a + l];
        double sumf = (double)reduce_4x8(acc);
        for (int i = np; i < n; i++) sumf += (double)(x[i] * y[i]);
        return (float)sumf;
    }
    if (order == ZVO_ORDER_SEQ_F32)
    {
        float s = 0.0f;
        for (int i = 0; i < n; i++) s += x[i] * y[i];
        return s;
    }
    double s = 0.0;
    for (int i = 0; i < n; i++) s += (double)x[i] * (double)y[i];
    return (float)s;
}

/* ggml_vec_dot_f32 on AVX2+FMA (ggml-cpu.c:1352-1393): fused multiply-add per lane; the scalar
 * tail `sumf += x[i]*y[i]` stays an unfused multiply + add in the x86-64-v3 build of oracle/_ref
 * (verified bit for bit: with a fused tail ~44 % of the encoder features are 1 ulp off).          */
static float dot_f32(const float *x, const float *y, int n, int order)
{
    if (order == ZVO_ORDER_GGML_AVX2)
    {
        float acc[4][8];
        memset(acc, 0, sizeof(acc));
        const int np = n & ~31;
        for (int i = 0; i < np; i += 32)
            for (int a = 0; a < 4; a++)
                for (int l = 0; l < 8; l++)
                    acc[a][l] = fmaf(x[i + 8 * a + l], y[i + 8 * a + l], acc[a][l]);
        float sumf = reduce_4x8(acc);
        for (int i = np; i < n; i++) sumf += x[i] * y[i];
        return sumf;
    }
    if (order == ZVO_ORDER_SEQ_F32)
    {
        float s = 0.0f;
        for (int i = 0; i < n; i++) s += x[i] * y[i];
        return s;
    }
    double s = 0.0;
    for (int i = 0; i < n; i++) s += (double)x[i] * (double)y[i];
    return (float)s;
}

/* ------------------------------------------------------------------------------------------ */
/* element-wise helpers (ggml-cpu.c:1740-1750)                                                 */

static inline float lrelu(float x, float ns) { return ((x > 0.f) ? x : 0.f) + ns * ((x < 0.0f) ? x : 0.f); }

static void lrelu_inplace(float *x, size_t n, float ns)
{
    for (size_t i = 0; i < n; i++) x[i] = lrelu(x[i], ns);
}

/* ggml_norm (ggml-cpu.c:6880-6929) over rows of length n */
void zvo_norm_rows(const float *x, int rows, int n, float eps, float *y)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++)
    {
        const float *xr = x + (size_t)r * n;
        float *yr = y + (size_t)r * n;
        double sum = 0.0;
        for (int i = 0; i < n; i++) sum += (double)xr[i];
        float mean = (float)(sum / n);
        double sum2 = 0.0;
        for (int i = 0; i < n; i++)
        {
            float v = xr[i] - mean;
            yr[i] = v;
            sum2 += (double)(v * v);
        }
        float variance = (float)(sum2 / n);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int i = 0; i < n; i++) yr[i] *= scale;
    }
}

static void transpose(const float *x, int rows, int cols, float *y) /* y[c][r] = x[r][c] */
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < cols; c++)
        for (int r = 0; r < rows; r++) y[(size_t)c * rows + r] = x[(size_t)r * cols + c];
}

/* per-channel affine on cf data: x[c][t] = x[c][t]*w[c] (+ b[c]) — two separate f32 roundings as
 * the reference runs ggml_mul then ggml_add (src/stylettsdec.cpp:97-98,195-196)                   */
static void chan_mul_add(float *x, int C, int L, const float *w, const float *b)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++)
    {
        float *xr = x + (size_t)c * L;
        for (int t = 0; t < L; t++) xr[t] = xr[t] * w[c];
        if (b)
            for (int t = 0; t < L; t++) xr[t] = xr[t] + b[c];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* conv1d = im2col(F16) + mul_mat (ggml.c:3769-3786, ggml-cpu.c:9890-9961, :7377-7554)          */

static int conv_cf(zvo_ctx *c, const float *x, int L, int IC, const uint16_t *w, int OC, int K,
                   int pad, int dil, const float *bias, float *out)
{
    const int OL = L + 2 * pad - dil * (K - 1);
    if (OL <= 0) return fail("conv: non-positive output length");
    const int Lp = L + 2 * pad;
    const int n = IC * K;
    /* time-major padded copy of the (rounded) input: xt[tp][ic], tp = t + pad */
    float *xt = (float *)calloc((size_t)Lp * IC, sizeof(float));
    float *wf = (float *)malloc((size_t)OC * n * sizeof(float));
    if (!xt || !wf) { free(xt); free(wf); return fail("conv: out of memory"); }
    const int f16in = c->f16_inputs;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < L; t++)
        for (int ic = 0; ic < IC; ic++)
        {
            float v = x[(size_t)ic * L + t];
            xt[(size_t)(t + pad) * IC + ic] = f16in ? round_f16(v) : v;
        }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < OC * n; i++) wf[i] = f16_to_f32(w[i]);   /* [oc][ic*K + k], k fastest */

    const int order = c->order;
#pragma omp parallel
    {
        float *col = (float *)malloc((size_t)n * sizeof(float));
#pragma omp for schedule(static)
        for (int t = 0; t < OL; t++)
        {
            for (int ic = 0; ic < IC; ic++)
                for (int k = 0; k < K; k++) col[ic * K + k] = xt[(size_t)(t + k * dil) * IC + ic];
            for (int oc = 0; oc < OC; oc++)
            {
                float s = dot_f16ops(col, wf + (size_t)oc * n, n, order);
                out[(size_t)oc * OL + t] = bias ? s + bias[oc] : s;
            }
        }
        free(col);
    }
    free(xt);
    free(wf);
    return 0;
}

int zvo_conv1d(zvo_ctx *c, const float *x, int L, int IC, const uint16_t *w, int OC, int K, int pad,
               int dil, const float *bias, float *out)
{
    apply_threads(c);
    return conv_cf(c, x, L, IC, w, OC, K, pad, dil, bias, out);
}

/* conv by tensor names: weight `<wname>` f16 ne [K, IC, OC], bias `<bname>` f32 ne [OC] or none */
static int conv_named(zvo_ctx *c, const float *x, int L, int IC, const zvo_tensor *w, const zvo_tensor *b,
                      int pad, int dil, float *out, int *OC_out)
{
    if (!w) return -1;
    if (w->dtype != ZVO_F16) return fail("%s: conv weight must be f16", w->name);
    if (w->ne[1] != IC) return fail("%s: IC mismatch (%d vs %lld)", w->name, IC, (long long)w->ne[1]);
    const int K = (int)w->ne[0], OC = (int)w->ne[2];
    if (b && b->ne[0] != OC) return fail("%s: bias length mismatch", b->name);
    if (OC_out) *OC_out = OC;
    return conv_cf(c, x, L, IC, (const uint16_t *)w->data, OC, K, pad, dil, b ? (const float *)b->data : NULL, out);
}

/* ------------------------------------------------------------------------------------------ */
/* HiFi-GAN (reference src/hifigan.cpp)                                                         */

/* conv_transpose1d, src/hifigan.cpp:22-71: zero-stuff to (L-1)*s+1+2*(k-1-p)+op, plain conv with the
 * stored (pre-flipped) kernel, + bias.  Gaps are true zeros (SURVEY Appx C-H1).                   */
static int conv_transpose_cf(zvo_ctx *c, const float *x, int L, int IC, int idx, int stride, float **out, int *OC, int *OL)
{
    const zvo_tensor *w = get(c, "_meldec.upsamples.%d.1.w", idx);
    const zvo_tensor *b = get(c, "_meldec.upsamples.%d.1.b", idx);
    if (!w || !b) return -1;
    const int K = (int)w->ne[0];
    const int padding = stride / 2 + stride % 2, output_padding = stride % 2;   /* src/hifigan.cpp:295-296 */
    const int up_len = (L - 1) * stride + 1;
    const int off = (K - 1) - padding;
    const int padded = up_len + 2 * off + output_padding;
    float *u = (float *)calloc((size_t)IC * padded, sizeof(float));
    if (!u) return fail("out of memory");
    for (int ic = 0; ic < IC; ic++)
        for (int i = 0; i < L; i++) u[(size_t)ic * padded + off + (size_t)i * stride] = x[(size_t)ic * L + i];
    *OC = (int)w->ne[2];
    *OL = padded - (K - 1);
    *out = (float *)malloc((size_t)(*OC) * (*OL) * sizeof(float));
    int rc = conv_named(c, u, padded, IC, w, b, 0, 1, *out, NULL);
    free(u);
    return rc;
}

/* HiFiGANResidualBlock, src/hifigan.cpp:74-185 */
static int resblock_cf(zvo_ctx *c, const float *x, int L, int C, int idx, const int *dils, int ndil, float *y)
{
    size_t n = (size_t)C * L;
    float *xt = (float *)malloc(n * sizeof(float)), *xt2 = (float *)malloc(n * sizeof(float));
    memcpy(y, x, n * sizeof(float));
    int rc = 0;
    for (int d = 0; d < ndil && rc == 0; d++)
    {
        const zvo_tensor *w1 = get(c, "_meldec.blocks.%d.convs1.%d.1.w", idx, d);
        const zvo_tensor *b1 = get(c, "_meldec.blocks.%d.convs1.%d.1.b", idx, d);
        const zvo_tensor *w2 = get(c, "_meldec.blocks.%d.convs2.%d.1.w", idx, d);
        const zvo_tensor *b2 = get(c, "_meldec.blocks.%d.convs2.%d.1.b", idx, d);
        if (!w1 || !b1 || !w2 || !b2) { rc = -1; break; }
        for (size_t i = 0; i < n; i++) xt[i] = lrelu(y[i], 0.1f);
        int K = (int)w1->ne[0];
        rc = conv_named(c, xt, L, C, w1, b1, (K - 1) / 2 * dils[d], dils[d], xt2, NULL);
        if (rc) break;
        lrelu_inplace(xt2, n, 0.1f);
        K = (int)w2->ne[0];
        rc = conv_named(c, xt2, L, C, w2, b2, (K - 1) / 2, 1, xt, NULL);
        if (rc) break;
        for (size_t i = 0; i < n; i++) y[i] = y[i] + xt[i];
    }
    free(xt);
    free(xt2);
    return rc;
}

/* HiFiGAN::HiFiGAN graph + eval, src/hifigan.cpp:187-377 */
int zvo_vocoder(zvo_ctx *c, const float *mel, int T, float *wav)
{
    apply_threads(c);
    static const int scales[4] = {5, 5, 4, 3};          /* src/zerovox.cpp:129 */
    static const int dils[3] = {1, 3, 5};               /* src/zerovox.cpp:132-134 */
    const int n_up = 4, n_rb = 3, ksz = 7;
    const zvo_tensor *mean = get(c, "hifigan.mean"), *scale = get(c, "hifigan.scale");
    const zvo_tensor *iw = get(c, "_meldec.input_conv.w"), *ib = get(c, "_meldec.input_conv.b");
    const zvo_tensor *ow = get(c, "_meldec.output_conv.1.w"), *ob = get(c, "_meldec.output_conv.1.b");
    if (!mean || !scale || !iw || !ib || !ow || !ob) return -1;
    const int M = (int)mean->ne[0];
    const float *mu = (const float *)mean->data, *sc = (const float *)scale->data;

    /* (mel - mean) / scale, then transpose to cf (src/hifigan.cpp:242-246) */
    float *x = (float *)malloc((size_t)M * T * sizeof(float));
    for (int t = 0; t < T; t++)
        for (int m = 0; m < M; m++) x[(size_t)m * T + t] = (mel[(size_t)t * M + m] - mu[m]) / sc[m];

    int C = (int)iw->ne[2], L = T;
    float *cur = (float *)malloc((size_t)C * L * sizeof(float));
    int rc = conv_named(c, x, L, M, iw, ib, (ksz - 1) / 2, 1, cur, NULL);
    free(x);

    for (int i = 0; i < n_up && rc == 0; i++)
    {
        lrelu_inplace(cur, (size_t)C * L, 0.1f);
        float *up = NULL;
        int OC = 0, OL = 0;
        rc = conv_transpose_cf(c, cur, L, C, i, scales[i], &up, &OC, &OL);
        free(cur);
        cur = NULL;
        if (rc) { free(up); break; }
        C = OC;
        L = OL;
        size_t n = (size_t)C * L;
        float *cs = NULL, *y = (float *)malloc(n * sizeof(float));
        for (int j = 0; j < n_rb && rc == 0; j++)
        {
            rc = resblock_cf(c, up, L, C, i * n_rb + j, dils, 3, y);
            if (rc) break;
            if (!cs) { cs = y; y = (float *)malloc(n * sizeof(float)); }
            else for (size_t e = 0; e < n; e++) cs[e] = cs[e] + y[e];
        }
        free(y);
        free(up);
        if (rc) { free(cs); break; }
        const float inv = (float)(1.0 / (float)n_rb);            /* src/hifigan.cpp:315 */
        for (size_t e = 0; e < n; e++) cs[e] *= inv;
        cur = cs;
    }
    if (rc) { free(cur); return rc; }

    lrelu_inplace(cur, (size_t)C * L, (float)1e-2);              /* src/hifigan.cpp:324 */
    float *o = (float *)malloc((size_t)L * sizeof(float));
    rc = conv_named(c, cur, L, C, ow, ob, (ksz - 1) / 2, 1, o, NULL);
    free(cur);
    if (rc == 0)
        for (int t = 0; t < L; t++) wav[t] = tanhf(o[t]);
    free(o);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* f32 linear layer = ggml_mul_mat(W, x) (+ bias): W ne [in, out] (row o contiguous), x tm [n][in] */

static void linear_tm(zvo_ctx *c, const float *x, int n, int in, const float *W, const float *b, int out, float *y)
{
    const int order = c->order;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++)
        for (int o = 0; o < out; o++)
        {
            float s = dot_f32(W + (size_t)o * in, x + (size_t)i * in, in, order);
            y[(size_t)i * out + o] = b ? s + b[o] : s;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* StyleTTS decoder (reference src/stylettsdec.cpp)                                             */

/* InstanceNorm1d with affine on cf data: ggml_norm over T, then *w, +b (src/stylettsdec.cpp:94-98) */
static void instnorm_affine_cf(const float *x, int C, int L, const float *w, const float *b, float *y)
{
    zvo_norm_rows(x, C, L, 1e-5f, y);
    chan_mul_add(y, C, L, w, b);
}

/* ResBlk1d::graph, src/stylettsdec.cpp:69-149 */
static int resblk1d(zvo_ctx *c, const float *x, int L, int idx, int dim_in, int dim_out, float *out)
{
    const int learned_sc = dim_in != dim_out;
    const zvo_tensor *w1 = get(c, "_mel_decoder.encode.%d.conv1.w", idx), *b1 = get(c, "_mel_decoder.encode.%d.conv1.b", idx);
    const zvo_tensor *w2 = get(c, "_mel_decoder.encode.%d.conv2.w", idx), *b2 = get(c, "_mel_decoder.encode.%d.conv2.b", idx);
    const zvo_tensor *n1w = get(c, "_mel_decoder.encode.%d.norm1.w", idx), *n1b = get(c, "_mel_decoder.encode.%d.norm1.b", idx);
    const zvo_tensor *n2w = get(c, "_mel_decoder.encode.%d.norm2.w", idx), *n2b = get(c, "_mel_decoder.encode.%d.norm2.b", idx);
    const zvo_tensor *wsc = learned_sc ? get(c, "_mel_decoder.encode.%d.conv1x1.w", idx) : NULL;
    if (!w1 || !b1 || !w2 || !b2 || !n1w || !n1b || !n2w || !n2b || (learned_sc && !wsc)) return -1;

    size_t nin = (size_t)dim_in * L, nout = (size_t)dim_out * L;
    float *y = (float *)malloc(nout * sizeof(float));
    float *r = (float *)malloc(nin * sizeof(float)), *r2 = (float *)malloc(nin * sizeof(float));
    int rc = 0;
    if (learned_sc) rc = conv_named(c, x, L, dim_in, wsc, NULL, 0, 1, y, NULL);
    else memcpy(y, x, nin * sizeof(float));
    if (!rc)
    {
        instnorm_affine_cf(x, dim_in, L, (const float *)n1w->data, (const float *)n1b->data, r);
        lrelu_inplace(r, nin, 0.2f);
        rc = conv_named(c, r, L, dim_in, w1, b1, 1, 1, r2, NULL);
    }
    if (!rc)
    {
        instnorm_affine_cf(r2, dim_in, L, (const float *)n2w->data, (const float *)n2b->data, r);
        lrelu_inplace(r, nin, 0.2f);
        rc = conv_named(c, r, L, dim_in, w2, b2, 1, 1, out, NULL);
    }
    if (!rc)
    {
        const float s = (float)(1.0 / sqrt(2.0));                 /* src/stylettsdec.cpp:146 */
        for (size_t i = 0; i < nout; i++) out[i] = (out[i] + y[i]) * s;
    }
    free(y);
    free(r);
    free(r2);
    return rc;
}

/* AdaIN1d::graph, src/stylettsdec.cpp:171-200 (x is overwritten, like ggml_norm_inplace) */
static int adain1d(zvo_ctx *c, float *x, int C, int L, const float *s, int E, int idx0, int idx1)
{
    const zvo_tensor *fw = get(c, "_mel_decoder.decode.%d.norm%d.fc.w", idx0, idx1);
    const zvo_tensor *fb = get(c, "_mel_decoder.decode.%d.norm%d.fc.b", idx0, idx1);
    if (!fw || !fb) return -1;
    if (fw->ne[0] != E || fw->ne[1] != 2 * C) return fail("%s: shape mismatch", fw->name);
    float *h = (float *)malloc((size_t)2 * C * sizeof(float));
    linear_tm(c, s, 1, E, (const float *)fw->data, (const float *)fb->data, 2 * C, h);
    for (int i = 0; i < C; i++) h[i] = h[i] + 1.0f;               /* gamma += one */
    zvo_norm_rows(x, C, L, 1e-5f, x);
    chan_mul_add(x, C, L, h, h + C);
    free(h);
    return 0;
}

/* AdainResBlk1d::graph, src/stylettsdec.cpp:242-304 */
static int adainresblk1d(zvo_ctx *c, const float *x, int L, const float *s, int E, int idx, int dim_in, int dim_out, float *out)
{
    const int learned_sc = dim_in != dim_out;
    const zvo_tensor *w1 = get(c, "_mel_decoder.decode.%d.conv1.w", idx), *b1 = get(c, "_mel_decoder.decode.%d.conv1.b", idx);
    const zvo_tensor *w2 = get(c, "_mel_decoder.decode.%d.conv2.w", idx), *b2 = get(c, "_mel_decoder.decode.%d.conv2.b", idx);
    const zvo_tensor *wsc = learned_sc ? get(c, "_mel_decoder.decode.%d.conv1x1.w", idx) : NULL;
    if (!w1 || !b1 || !w2 || !b2 || (learned_sc && !wsc)) return -1;
    size_t nin = (size_t)dim_in * L, nout = (size_t)dim_out * L;
    float *a = (float *)malloc(nin * sizeof(float)), *t = (float *)malloc(nout * sizeof(float));
    memcpy(a, x, nin * sizeof(float));
    int rc = adain1d(c, a, dim_in, L, s, E, idx, 1);
    if (!rc)
    {
        lrelu_inplace(a, nin, 0.2f);
        rc = conv_named(c, a, L, dim_in, w1, b1, 1, 1, t, NULL);
    }
    if (!rc) rc = adain1d(c, t, dim_out, L, s, E, idx, 2);
    if (!rc)
    {
        lrelu_inplace(t, nout, 0.2f);
        rc = conv_named(c, t, L, dim_out, w2, b2, 1, 1, out, NULL);
    }
    if (!rc)
    {
        const float sq = (float)(1 / sqrt(2.0));                  /* src/stylettsdec.cpp:301 */
        if (learned_sc)
        {
            rc = conv_named(c, x, L, dim_in, wsc, NULL, 0, 1, t, NULL);
            if (!rc)
                for (size_t i = 0; i < nout; i++) out[i] = (out[i] + t[i]) * sq;
        }
        else
            for (size_t i = 0; i < nout; i++) out[i] = (out[i] + x[i]) * sq;
    }
    free(a);
    free(t);
    return rc;
}

/* StyleTTSDecoder graph + eval, src/stylettsdec.cpp:306-470 */
int zvo_decoder(zvo_ctx *c, const float *hidden, const float *style, int T, float *mel)
{
    apply_threads(c);
    const zvo_tensor *a0w = get(c, "_mel_decoder.asr_res.0.w"), *a0b = get(c, "_mel_decoder.asr_res.0.b");
    const zvo_tensor *a1w = get(c, "_mel_decoder.asr_res.1.w"), *a1b = get(c, "_mel_decoder.asr_res.1.b");
    const zvo_tensor *tow = get(c, "_mel_decoder.to_out.0.w"), *tob = get(c, "_mel_decoder.to_out.0.b");
    if (!a0w || !a0b || !a1w || !a1b || !tow || !tob) return -1;
    const int E = (int)a0w->ne[1], R = (int)a0w->ne[2], M = (int)tow->ne[2];
    const int B = 2 * E, CAT = B + R;

    float *enc = (float *)malloc((size_t)E * T * sizeof(float));
    transpose(hidden, T, E, enc);                                 /* [T][E] -> cf [E][T] */

    float *x0 = (float *)malloc((size_t)B * T * sizeof(float)), *x1 = (float *)malloc((size_t)B * T * sizeof(float));
    float *cat = (float *)malloc((size_t)CAT * T * sizeof(float));
    float *asr = (float *)malloc((size_t)R * T * sizeof(float)), *tmp = (float *)malloc((size_t)R * T * sizeof(float));
    int rc = resblk1d(c, enc, T, 0, E, B, x0);
    if (!rc) rc = resblk1d(c, x0, T, 1, B, B, x1);
    if (!rc) rc = conv_named(c, enc, T, E, a0w, a0b, 0, 1, tmp, NULL);
    if (!rc) instnorm_affine_cf(tmp, R, T, (const float *)a1w->data, (const float *)a1b->data, asr);

    /* decode0..2 on cat([x, asr]) (src/stylettsdec.cpp:398-422), decode3,4 plain */
    const int dims[5][2] = {{CAT, B}, {CAT, B}, {CAT, E}, {E, E}, {E, E}};
    float *cur = x1, *nxt = x0;
    for (int i = 0; i < 5 && !rc; i++)
    {
        const float *in = cur;
        if (dims[i][0] == CAT)
        {
            memcpy(cat, cur, (size_t)B * T * sizeof(float));
            memcpy(cat + (size_t)B * T, asr, (size_t)R * T * sizeof(float));
            in = cat;
        }
        rc = adainresblk1d(c, in, T, style, E, i, dims[i][0], dims[i][1], nxt);
        float *sw = cur; cur = nxt; nxt = sw;
    }
    if (!rc)
    {
        float *o = (float *)malloc((size_t)M * T * sizeof(float));
        rc = conv_named(c, cur, T, E, tow, NULL, 0, 1, o, NULL);
        const float *b = (const float *)tob->data;                /* bias added on the [80,T] frame-major view */
        if (!rc)
            for (int t = 0; t < T; t++)
                for (int m = 0; m < M; m++) mel[(size_t)t * M + m] = o[(size_t)m * T + t] + b[m];
        free(o);
    }
    free(enc); free(x0); free(x1); free(cat); free(asr); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* FastSpeech2 encoder (reference src/fs2encoder.cpp)                                           */

/* ggml_v_expf for __m256, one lane (ggml-cpu.c:1918-1957) */
static inline float v_expf_lane(float x)
{
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = f32_bits(z) << 23;
    const float k = bits_f32(e + f32_bits(1.0f));
    const int cflag = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u, fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)),
                         u, 0x1.ffffecp-1f * b);
    if (!cflag) return fmaf(j, k, k);
    const uint32_t g = (n <= 0.0f) ? 0x82000000u : 0u;
    const float s1 = bits_f32(g + 0x7f000000u);
    const float s2 = bits_f32(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}

/* ggml_compute_forward_soft_max_f32 row (ggml-cpu.c:8857-8879) + ggml_vec_soft_max_f32 AVX2 (:2058-2067) */
static void softmax_row(const float *x, int n, float *y, int order)
{
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) mx = x[i] > mx ? x[i] : mx;
    double sum = 0.0;
    int i = 0;
    if (order == ZVO_ORDER_GGML_AVX2)
        for (; i + 7 < n; i += 8)
        {
            float v[8], h[4];
            for (int l = 0; l < 8; l++) { v[l] = v_expf_lane(x[i + l] - mx); y[i + l] = v[l]; }
            for (int l = 0; l < 4; l++) h[l] = v[l + 4] + v[l];
            h[0] = h[0] + h[2];
            h[1] = h[1] + h[3];
            sum += (double)(h[0] + h[1]);
        }
    for (; i < n; i++)
    {
        float v = expf(x[i] - mx);
        sum += (double)v;
        y[i] = v;
    }
    const float inv = (float)(1.0 / sum);
    for (i = 0; i < n; i++) y[i] *= inv;
}

/* LayerNorm(x + residual) * w + b on tm rows (src/fs2encoder.cpp:132-137) */
static void add_layernorm_tm(const float *x, const float *res, int n, int E, const float *w, const float *b, float *y)
{
    size_t tot = (size_t)n * E;
    float *s = (float *)malloc(tot * sizeof(float));
    for (size_t i = 0; i < tot; i++) s[i] = res ? x[i] + res[i] : x[i];
    zvo_norm_rows(s, n, E, 1e-5f, y);
    for (int i = 0; i < n; i++)
        for (int e = 0; e < E; e++)
        {
            float v = w[e] * y[(size_t)i * E + e];
            y[(size_t)i * E + e] = v + b[e];
        }
    free(s);
}

/* MultiHeadAttention::graph, src/fs2encoder.cpp:71-140 (no mask) */
static int mha(zvo_ctx *c, const float *x, int N, int E, int layer, int H, float *out)
{
    const char *nm[4] = {"w_qs", "w_ks", "w_vs", "fc"};
    const zvo_tensor *W[4], *Bv[4];
    for (int i = 0; i < 4; i++)
    {
        W[i] = get(c, "_pe._enc.laystk.%d.slf_attn.%s.w", layer, nm[i]);
        Bv[i] = get(c, "_pe._enc.laystk.%d.slf_attn.%s.b", layer, nm[i]);
        if (!W[i] || !Bv[i]) return -1;
    }
    const zvo_tensor *lw = get(c, "_pe._enc.laystk.%d.slf_attn.layer_norm.w", layer);
    const zvo_tensor *lb = get(c, "_pe._enc.laystk.%d.slf_attn.layer_norm.b", layer);
    if (!lw || !lb) return -1;
    const int dk = E / H;
    size_t tot = (size_t)N * E;
    float *q = (float *)malloc(tot * 4), *k = (float *)malloc(tot * 4), *v = (float *)malloc(tot * 4), *o = (float *)malloc(tot * 4);
    linear_tm(c, x, N, E, (const float *)W[0]->data, (const float *)Bv[0]->data, E, q);
    linear_tm(c, x, N, E, (const float *)W[1]->data, (const float *)Bv[1]->data, E, k);
    linear_tm(c, x, N, E, (const float *)W[2]->data, (const float *)Bv[2]->data, E, v);

    const float temperature = (float)pow((double)dk, 0.5);        /* src/fs2encoder.cpp:66 */
    const float inv_t = (float)(1.0 / temperature);               /* :107 */
    const int order = c->order;
#pragma omp parallel
    {
        float *qh = (float *)malloc((size_t)dk * 4), *row = (float *)malloc((size_t)N * 4), *p = (float *)malloc((size_t)N * 4);
        float *kh = (float *)malloc((size_t)N * dk * 4), *vt = (float *)malloc((size_t)N * dk * 4);
        for (int h = 0; h < H; h++)
        {
            /* per-thread copies of this head's k (rows) and v transposed ([d][ik]) */
            for (int ik = 0; ik < N; ik++)
                for (int d = 0; d < dk; d++)
                {
                    kh[(size_t)ik * dk + d] = k[(size_t)ik * E + h * dk + d];
                    vt[(size_t)d * N + ik] = v[(size_t)ik * E + h * dk + d];
                }
#pragma omp for schedule(static)
            for (int iq = 0; iq < N; iq++)
            {
                memcpy(qh, q + (size_t)iq * E + h * dk, (size_t)dk * 4);
                for (int ik = 0; ik < N; ik++) row[ik] = dot_f32(qh, kh + (size_t)ik * dk, dk, order) * inv_t;
                softmax_row(row, N, p, order);
                for (int d = 0; d < dk; d++) o[(size_t)iq * E + h * dk + d] = dot_f32(p, vt + (size_t)d * N, N, order);
            }
        }
        free(qh); free(row); free(p); free(kh); free(vt);
    }
    linear_tm(c, o, N, E, (const float *)W[3]->data, (const float *)Bv[3]->data, E, q);
    add_layernorm_tm(q, x, N, E, (const float *)lw->data, (const float *)lb->data, out);
    free(q); free(k); free(v); free(o);
    return 0;
}

/* PositionwiseFeedForward::graph, src/fs2encoder.cpp:174-228 */
static int ffn(zvo_ctx *c, const float *x, int N, int E, int layer, const int ksz[2], float *out)
{
    const zvo_tensor *w1 = get(c, "_pe._enc.laystk.%d.pos_ffn.w_1.w", layer), *b1 = get(c, "_pe._enc.laystk.%d.pos_ffn.w_1.b", layer);
    const zvo_tensor *w2 = get(c, "_pe._enc.laystk.%d.pos_ffn.w_2.w", layer), *b2 = get(c, "_pe._enc.laystk.%d.pos_ffn.w_2.b", layer);
    const zvo_tensor *lw = get(c, "_pe._enc.laystk.%d.pos_ffn.layer_norm.w", layer), *lb = get(c, "_pe._enc.laystk.%d.pos_ffn.layer_norm.b", layer);
    if (!w1 || !b1 || !w2 || !b2 || !lw || !lb) return -1;
    const int F = (int)w1->ne[2];
    float *xc = (float *)malloc((size_t)E * N * 4), *h = (float *)malloc((size_t)F * N * 4), *y = (float *)malloc((size_t)E * N * 4);
    transpose(x, N, E, xc);
    int rc = conv_named(c, xc, N, E, w1, b1, (ksz[0] - 1) / 2, 1, h, NULL);
    if (!rc)
    {
        for (size_t i = 0; i < (size_t)F * N; i++) h[i] = (h[i] > 0.f) ? h[i] : 0.f;
        rc = conv_named(c, h, N, F, w2, b2, (ksz[1] - 1) / 2, 1, y, NULL);
    }
    if (!rc)
    {
        transpose(y, E, N, xc);                                    /* back to tm [N][E] */
        add_layernorm_tm(xc, x, N, E, (const float *)lw->data, (const float *)lb->data, out);
    }
    free(xc); free(h); free(y);
    return rc;
}

/* VariancePredictor::graph, src/fs2encoder.cpp:386-440; second conv pads with the literal 1 (:417) */
static int variance_predictor(zvo_ctx *c, const float *x, int N, int E, const char *prefix, int vk, float *out)
{
    const zvo_tensor *w1 = get(c, "%s.conv_layer.conv1d_1.conv.w", prefix), *b1 = get(c, "%s.conv_layer.conv1d_1.conv.b", prefix);
    const zvo_tensor *w2 = get(c, "%s.conv_layer.conv1d_2.conv.w", prefix), *b2 = get(c, "%s.conv_layer.conv1d_2.conv.b", prefix);
    const zvo_tensor *l1w = get(c, "%s.conv_layer.layer_norm_1.w", prefix), *l1b = get(c, "%s.conv_layer.layer_norm_1.b", prefix);
    const zvo_tensor *l2w = get(c, "%s.conv_layer.layer_norm_2.w", prefix), *l2b = get(c, "%s.conv_layer.layer_norm_2.b", prefix);
    const zvo_tensor *lw = get(c, "%s.linear_layer.w", prefix), *lb = get(c, "%s.linear_layer.b", prefix);
    if (!w1 || !b1 || !w2 || !b2 || !l1w || !l1b || !l2w || !l2b || !lw || !lb) return -1;
    const int V = (int)w1->ne[2];
    float *xc = (float *)malloc((size_t)E * N * 4), *h = (float *)malloc((size_t)V * N * 4), *ht = (float *)malloc((size_t)V * N * 4);
    transpose(x, N, E, xc);
    int rc = conv_named(c, xc, N, E, w1, b1, (vk - 1) / 2, 1, h, NULL);
    if (!rc)
    {
        for (size_t i = 0; i < (size_t)V * N; i++) h[i] = (h[i] > 0.f) ? h[i] : 0.f;
        transpose(h, V, N, ht);                                    /* tm [N][V] */
        add_layernorm_tm(ht, NULL, N, V, (const float *)l1w->data, (const float *)l1b->data, h);
        transpose(h, N, V, ht);                                    /* cf [V][N] */
        rc = conv_named(c, ht, N, V, w2, b2, 1, 1, h, NULL);
    }
    if (!rc)
    {
        for (size_t i = 0; i < (size_t)V * N; i++) h[i] = (h[i] > 0.f) ? h[i] : 0.f;
        transpose(h, V, N, ht);
        add_layernorm_tm(ht, NULL, N, V, (const float *)l2w->data, (const float *)l2b->data, h);
        const float *w = (const float *)lw->data, b = ((const float *)lb->data)[0];
        for (int i = 0; i < N; i++) out[i] = dot_f32(h + (size_t)i * V, w, V, c->order) + b;
    }
    free(xc); free(h); free(ht);
    return rc;
}

/* ggml_zv_mul_clamp_to_i32, src/fs2encoder.cpp:442-474 */
static void bucketize(const float *pred, int N, int nbins, int32_t *out)
{
    const int bin_max = nbins - 1;
    for (int i = 0; i < N; i++)
    {
        float x = pred[i];
        x = x * bin_max;
        int32_t y = (int32_t)(x + 0.5);
        if (y < 0) y = 0;
        if (y > bin_max) y = bin_max;
        out[i] = y;
    }
}

/* host length regulator, src/fs2encoder.cpp:611-654 */
int zvo_length_regulator(const float *features, const float *logdur, int N, int E, int T, float *hidden)
{
    long xoff = 0;
    memset(hidden, 0, (size_t)T * E * sizeof(float));
    for (int i = 0; i < N; i++)
    {
        float dur = exp(logdur[i]) - 1.0;
        int32_t rounded = (int32_t)(dur + 0.5);
        if (rounded < 0) continue;
        for (int32_t r = 0; r < rounded; r++)
        {
            memcpy(hidden + (size_t)xoff * E, features + (size_t)i * E, (size_t)E * sizeof(float));
            xoff += 1;
            if (xoff >= T) break;
        }
        if (xoff >= T) break;
    }
    return (int)xoff;
}

/* Encoder::graph + FS2Encoder graph + eval, src/fs2encoder.cpp:289-336,477-656 */
int zvo_encoder(zvo_ctx *c, const zvo_encoder_params *p, const int32_t *ids, const int32_t *puncts,
                const float *style, float *hidden, int32_t *n_frames, float *features_out, float *logdur_out,
                float *pitch_out, float *energy_out, int32_t *pitch_bucket, int32_t *energy_bucket)
{
    apply_threads(c);
    const int N = p->n_phonemes, E = p->emb_dim + p->punct_emb_dim, T = p->max_seq_len;
    const zvo_tensor *we = get(c, "_pe._enc.src_word_emb.w"), *pe = get(c, "_pe._enc.punct_embed.w");
    const zvo_tensor *st = get(c, "sinusoid_encoding_table");
    const zvo_tensor *pemb = get(c, "_pe._var_adapt.pitch_embedding.w"), *eemb = get(c, "_pe._var_adapt.energy_embedding.w");
    if (!we || !pe || !st || !pemb || !eemb) return -1;
    if (st->ne[1] < N) return fail("sinusoid table has %lld rows < N=%d", (long long)st->ne[1], N);
    for (int i = 0; i < N; i++)
    {
        if (ids[i] < 0 || ids[i] >= we->ne[1]) return fail("phoneme id %d out of range at %d", ids[i], i);
        if (puncts[i] < 0 || puncts[i] >= pe->ne[1]) return fail("punct id %d out of range at %d", puncts[i], i);
    }
    size_t tot = (size_t)N * E;
    float *x = (float *)malloc(tot * 4), *y = (float *)malloc(tot * 4);
    const float *wed = (const float *)we->data, *ped = (const float *)pe->data, *std_ = (const float *)st->data;
    for (int i = 0; i < N; i++)
    {
        memcpy(x + (size_t)i * E, wed + (size_t)ids[i] * p->emb_dim, (size_t)p->emb_dim * 4);
        memcpy(x + (size_t)i * E + p->emb_dim, ped + (size_t)puncts[i] * p->punct_emb_dim, (size_t)p->punct_emb_dim * 4);
        for (int e = 0; e < E; e++) x[(size_t)i * E + e] = x[(size_t)i * E + e] + std_[(size_t)i * E + e];
    }
    int rc = 0;
    for (int l = 0; l < p->n_layers && !rc; l++)
    {
        rc = mha(c, x, N, E, l, p->n_heads, y);
        if (!rc) rc = ffn(c, y, N, E, l, p->ffn_kernel, x);
    }
    float *ld = (float *)malloc((size_t)N * 4), *pp = (float *)malloc((size_t)N * 4), *ep = (float *)malloc((size_t)N * 4);
    int32_t *pb = (int32_t *)malloc((size_t)N * 4), *eb = (int32_t *)malloc((size_t)N * 4);
    if (!rc)
    {
        for (int i = 0; i < N; i++)
            for (int e = 0; e < E; e++) x[(size_t)i * E + e] = x[(size_t)i * E + e] + style[e];
        rc = variance_predictor(c, x, N, E, "_pe._var_adapt.duration_predictor", p->vp_kernel, ld);
    }
    if (!rc) rc = variance_predictor(c, x, N, E, "_pe._var_adapt.pitch_predictor", p->vp_kernel, pp);
    if (!rc)
    {
        bucketize(pp, N, p->ve_n_bins, pb);
        const float *emb = (const float *)pemb->data;
        for (int i = 0; i < N; i++)
            for (int e = 0; e < E; e++) x[(size_t)i * E + e] = x[(size_t)i * E + e] + emb[(size_t)pb[i] * E + e];
        rc = variance_predictor(c, x, N, E, "_pe._var_adapt.engy_pred", p->vp_kernel, ep);   /* sees pitch-augmented features (:569-572) */
    }
    if (!rc)
    {
        bucketize(ep, N, p->ve_n_bins, eb);
        const float *emb = (const float *)eemb->data;
        for (int i = 0; i < N; i++)
            for (int e = 0; e < E; e++) x[(size_t)i * E + e] = x[(size_t)i * E + e] + emb[(size_t)eb[i] * E + e];
        int nf = zvo_length_regulator(x, ld, N, E, T, hidden);
        if (n_frames) *n_frames = nf;
        if (features_out) memcpy(features_out, x, tot * 4);
        if (logdur_out) memcpy(logdur_out, ld, (size_t)N * 4);
        if (pitch_out) memcpy(pitch_out, pp, (size_t)N * 4);
        if (energy_out) memcpy(energy_out, ep, (size_t)N * 4);
        if (pitch_bucket) memcpy(pitch_bucket, pb, (size_t)N * 4);
        if (energy_bucket) memcpy(energy_bucket, eb, (size_t)N * 4);
    }
    free(x); free(y); free(ld); free(pp); free(ep); free(pb); free(eb);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* One layer at a time, for teacher-forced per-layer parity tests (the counterpart of the reference's
 * tensor_dbg, src/utils.cpp:19-44: the GPU runs the same layer on the same input, so nothing compounds).
 * All inputs / outputs here are time-major [rows][channels] like the stage boundaries.           */

int zvo_layer(zvo_ctx *c, int kind, int index, const float *x, int rows, int cols, const float *style, int E, int H,
              const int *ksz, float *out)
{
    apply_threads(c);
    int rc = 0;
    if (kind == ZVO_LAYER_VOC_RESBLOCK)                /* HiFiGANResidualBlock `index` (src/hifigan.cpp:74-185): [L][C] -> [L][C] */
    {
        const int dils[3] = {1, 3, 5};
        float *xc = (float *)malloc((size_t)rows * cols * 4), *yc = (float *)malloc((size_t)rows * cols * 4);
        transpose(x, rows, cols, xc);
        rc = resblock_cf(c, xc, rows, cols, index, dils, 3, yc);
        if (!rc) transpose(yc, cols, rows, out);
        free(xc); free(yc);
    }
    else if (kind == ZVO_LAYER_ENC_FFT)                /* FFTBlock `index` (src/fs2encoder.cpp:71-140,174-228): [N][E] -> [N][E] */
    {
        float *y = (float *)malloc((size_t)rows * cols * 4);
        rc = mha(c, x, rows, cols, index, H, y);
        if (!rc) rc = ffn(c, y, rows, cols, index, ksz, out);
        free(y);
    }
    else if (kind == ZVO_LAYER_ENC_MHA)                /* MultiHeadAttention `index` alone (src/fs2encoder.cpp:71-140, residual + LayerNorm included) */
        rc = mha(c, x, rows, cols, index, H, out);
    else if (kind == ZVO_LAYER_ENC_FFN)                /* PositionwiseFeedForward `index` alone (src/fs2encoder.cpp:174-228, residual + LayerNorm included) */
        rc = ffn(c, x, rows, cols, index, ksz, out);
    else if (kind == ZVO_LAYER_DEC_ADAIN)              /* AdaIN1d alone (src/stylettsdec.cpp:171-200): index = 2 * decode block + (norm - 1): [T][C] -> [T][C] */
    {
        if (index < 0 || index > 9) return fail("AdaIN %d", index);
        float *xc = (float *)malloc((size_t)rows * cols * 4);
        transpose(x, rows, cols, xc);
        rc = adain1d(c, xc, cols, rows, style, E, index / 2, 1 + (index & 1));
        if (!rc) transpose(xc, cols, rows, out);
        free(xc);
    }
    else if (kind == ZVO_LAYER_DEC_BLOCK)              /* 0,1: ResBlk1d encode.{0,1}; 2..6: AdainResBlk1d decode.{0..4}: [T][cin] -> [T][cout] */
    {
        const zvo_tensor *a0w = get(c, "_mel_decoder.asr_res.0.w");
        if (!a0w) return -1;
        const int Ed = (int)a0w->ne[1], R = (int)a0w->ne[2], B = 2 * Ed, CAT = B + R;
        const int dims[7][2] = {{Ed, B}, {B, B}, {CAT, B}, {CAT, B}, {CAT, Ed}, {Ed, Ed}, {Ed, Ed}};
        if (index < 0 || index > 6 || cols != dims[index][0]) return fail("decoder block %d wants %d channels", index, index >= 0 && index <= 6 ? dims[index][0] : -1);
        const int co = dims[index][1];
        float *xc = (float *)malloc((size_t)rows * cols * 4), *yc = (float *)malloc((size_t)rows * co * 4);
        transpose(x, rows, cols, xc);
        if (index < 2) rc = resblk1d(c, xc, rows, index, cols, co, yc);
        else rc = adainresblk1d(c, xc, rows, style, E, index - 2, cols, co, yc);
        if (!rc) transpose(yc, co, rows, out);
        free(xc); free(yc);
    }
    else if (kind == ZVO_LAYER_VAR_PRED)               /* VariancePredictor 0 duration / 1 pitch / 2 energy: [N][E] -> [N] */
    {
        const char *pf[3] = {"_pe._var_adapt.duration_predictor", "_pe._var_adapt.pitch_predictor", "_pe._var_adapt.engy_pred"};
        if (index < 0 || index > 2) return fail("predictor %d", index);
        rc = variance_predictor(c, x, rows, cols, pf[index], ksz[0], out);
    }
    else if (kind == ZVO_LAYER_VOC_UPSAMPLE)           /* x = leaky_relu(x, 0.1); conv_transpose1d(index): src/hifigan.cpp:281-297, 22-71 */
    {
        static const int scales[4] = {5, 5, 4, 3};     /* src/zerovox.cpp:129 */
        if (index < 0 || index > 3) return fail("upsample %d", index);
        float *xc = (float *)malloc((size_t)rows * cols * 4), *up = NULL;
        int OC = 0, OL = 0;
        transpose(x, rows, cols, xc);
        lrelu_inplace(xc, (size_t)rows * cols, 0.1f);
        rc = conv_transpose_cf(c, xc, rows, cols, index, scales[index], &up, &OC, &OL);
        if (!rc && OL != rows * scales[index]) rc = fail("upsample %d: %d rows out of %d", index, OL, rows);
        if (!rc) transpose(up, OC, OL, out);
        free(xc); free(up);
    }
    else if (kind == ZVO_LAYER_VOC_INPUT)              /* src/hifigan.cpp:242-265 */
    {
        const zvo_tensor *mean = get(c, "hifigan.mean"), *scale = get(c, "hifigan.scale");
        const zvo_tensor *iw = get(c, "_meldec.input_conv.w"), *ib = get(c, "_meldec.input_conv.b");
        if (!mean || !scale || !iw || !ib) return -1;
        if (cols != (int)mean->ne[0]) return fail("input conv wants %d mel channels", (int)mean->ne[0]);
        const float *mu = (const float *)mean->data, *sc = (const float *)scale->data;
        const int C0 = (int)iw->ne[2], K = (int)iw->ne[0];
        float *xc = (float *)malloc((size_t)rows * cols * 4), *yc = (float *)malloc((size_t)rows * C0 * 4);
        for (int t = 0; t < rows; t++)
            for (int m = 0; m < cols; m++) xc[(size_t)m * rows + t] = (x[(size_t)t * cols + m] - mu[m]) / sc[m];
        rc = conv_named(c, xc, rows, cols, iw, ib, (K - 1) / 2, 1, yc, NULL);
        if (!rc) transpose(yc, C0, rows, out);
        free(xc); free(yc);
    }
    else if (kind == ZVO_LAYER_VOC_OUTPUT)             /* src/hifigan.cpp:324-345 */
    {
        const zvo_tensor *ow = get(c, "_meldec.output_conv.1.w"), *ob = get(c, "_meldec.output_conv.1.b");
        if (!ow || !ob) return -1;
        const int K = (int)ow->ne[0];
        float *xc = (float *)malloc((size_t)rows * cols * 4), *o = (float *)malloc((size_t)rows * 4);
        transpose(x, rows, cols, xc);
        lrelu_inplace(xc, (size_t)rows * cols, (float)1e-2);
        rc = conv_named(c, xc, rows, cols, ow, ob, (K - 1) / 2, 1, o, NULL);
        if (!rc)
            for (int t = 0; t < rows; t++) out[t] = tanhf(o[t]);
        free(xc); free(o);
    }
    else if (kind == ZVO_LAYER_DEC_ASR_RES)            /* src/stylettsdec.cpp:382-396 */
    {
        const zvo_tensor *a0w = get(c, "_mel_decoder.asr_res.0.w"), *a0b = get(c, "_mel_decoder.asr_res.0.b");
        const zvo_tensor *a1w = get(c, "_mel_decoder.asr_res.1.w"), *a1b = get(c, "_mel_decoder.asr_res.1.b");
        if (!a0w || !a0b || !a1w || !a1b) return -1;
        const int R = (int)a0w->ne[2];
        float *xc = (float *)malloc((size_t)rows * cols * 4), *t0 = (float *)malloc((size_t)rows * R * 4), *t1 = (float *)malloc((size_t)rows * R * 4);
        transpose(x, rows, cols, xc);
        rc = conv_named(c, xc, rows, cols, a0w, a0b, 0, 1, t0, NULL);
        if (!rc)
        {
            instnorm_affine_cf(t0, R, rows, (const float *)a1w->data, (const float *)a1b->data, t1);
            transpose(t1, R, rows, out);
        }
        free(xc); free(t0); free(t1);
    }
    else if (kind == ZVO_LAYER_DEC_TO_OUT)             /* src/stylettsdec.cpp:432-441 */
    {
        const zvo_tensor *tow = get(c, "_mel_decoder.to_out.0.w"), *tob = get(c, "_mel_decoder.to_out.0.b");
        if (!tow || !tob) return -1;
        const int M = (int)tow->ne[2];
        float *xc = (float *)malloc((size_t)rows * cols * 4), *o = (float *)malloc((size_t)rows * M * 4);
        transpose(x, rows, cols, xc);
        rc = conv_named(c, xc, rows, cols, tow, NULL, 0, 1, o, NULL);
        const float *b = (const float *)tob->data;                /* bias added on the [80,T] frame-major view */
        if (!rc)
            for (int t = 0; t < rows; t++)
                for (int m = 0; m < M; m++) out[(size_t)t * M + m] = o[(size_t)m * rows + t] + b[m];
        free(xc); free(o);
    }
    else if (kind == ZVO_LAYER_ENC_EMBED)              /* src/fs2encoder.cpp:306-324; x[n] = (id, punct) as floats */
    {
        const zvo_tensor *we = get(c, "_pe._enc.src_word_emb.w"), *pe = get(c, "_pe._enc.punct_embed.w");
        const zvo_tensor *st = get(c, "sinusoid_encoding_table");
        if (!we || !pe || !st) return -1;
        if (cols != 2) return fail("embedding wants [N][2] (id, punct)");
        const int emb = (int)we->ne[0], pd = (int)pe->ne[0], Ed = emb + pd;
        if (st->ne[1] < rows) return fail("sinusoid table has %lld rows < N=%d", (long long)st->ne[1], rows);
        const float *wed = (const float *)we->data, *ped = (const float *)pe->data, *std_ = (const float *)st->data;
        for (int i = 0; i < rows; i++)
        {
            const int id = (int)x[2 * i], pu = (int)x[2 * i + 1];
            if (id < 0 || id >= we->ne[1] || pu < 0 || pu >= pe->ne[1]) return fail("id / punct out of range at %d", i);
            memcpy(out + (size_t)i * Ed, wed + (size_t)id * emb, (size_t)emb * 4);
            memcpy(out + (size_t)i * Ed + emb, ped + (size_t)pu * pd, (size_t)pd * 4);
            for (int e = 0; e < Ed; e++) out[(size_t)i * Ed + e] = out[(size_t)i * Ed + e] + std_[(size_t)i * Ed + e];
        }
    }
    else
        return fail("unknown layer kind %d", kind);
    return rc;
}

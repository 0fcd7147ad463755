/* oracle/zv_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's hot path (FastSpeech2 encoder -> StyleTTS mel decoder ->
 * HiFi-GAN vocoder) as the reference's ggml CPU backend executes it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (zerovox.cpp_amd/csrc) never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py and tests/golden/ check this
 * restatement against outputs of the reference itself (oracle/_ref/zvref = the unmodified
 * reference stage classes on ggml-CPU, built by oracle/Makefile for x86-64-v3): with
 * ZVO_ORDER_GGML_AVX2 the three stages reproduce the reference bit for bit.
 */
#ifndef ZV_ORACLE_H
#define ZV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zvo_ctx zvo_ctx;

enum { ZVO_F32 = 0, ZVO_F16 = 1 };

/* summation order of every dot product (conv / linear / attention):
 *   GGML_AVX2 : ggml's vec_dot on an AVX2 build — 4 accumulators x 8 lanes over the flattened
 *               (ic*K + k) index in steps of 32, pairwise reduce, scalar tail
 *               (ggml-cpu.c:1352-1393 f32, :1463-1503 f16, macros :655-760).  Bit-exact vs oracle/_ref.
 *   SEQ_F32   : one f32 accumulator, natural index order  (used to measure re-association noise)
 *   SEQ_F64   : one f64 accumulator (closest to exact arithmetic of the same operands)           */
enum { ZVO_ORDER_GGML_AVX2 = 0, ZVO_ORDER_SEQ_F32 = 1, ZVO_ORDER_SEQ_F64 = 2 };

zvo_ctx *zvo_new(void);
void     zvo_free(zvo_ctx *c);
/* register a weight tensor by its GGUF name; data is borrowed (must outlive the ctx).
 * ne[] is the ggml shape (ne[0] fastest), n_dims <= 4.                                           */
int      zvo_set_tensor(zvo_ctx *c, const char *name, const void *data, int dtype, int n_dims, const int64_t *ne);
void     zvo_set_order(zvo_ctx *c, int order);
/* 1 (default): round conv inputs to f16 like ggml's im2col (ggml.c:3776); 0: keep f32 activations */
void     zvo_set_f16_inputs(zvo_ctx *c, int on);
void     zvo_set_threads(zvo_ctx *c, int n);
const char *zvo_last_error(void);

/* ---- stages (I/O layouts are the reference's: hidden[T*E] frame-major, mel[T*80] frame-major) ---- */

/* HiFiGAN::eval, reference src/hifigan.cpp:187-377.  geometry: upsample scales {5,5,4,3} and
 * dilations {1,3,5} are hard-coded like the reference caller (src/zerovox.cpp:127-138).          */
int zvo_vocoder(zvo_ctx *c, const float *mel, int T, float *wav);

/* StyleTTSDecoder::eval, reference src/stylettsdec.cpp:306-470 */
int zvo_decoder(zvo_ctx *c, const float *hidden, const float *style, int T, float *mel);

typedef struct
{
    int n_phonemes;       /* = max_n_phonemes of the reference graph (no mask: SURVEY Appx C-H2) */
    int max_seq_len;      /* T */
    int emb_dim, punct_emb_dim;
    int n_layers, n_heads;
    int ffn_kernel[2];
    int vp_kernel;
    int ve_n_bins;
} zvo_encoder_params;

/* FS2Encoder graph + eval incl. host length regulator, reference src/fs2encoder.cpp:477-656.
 * optional outputs may be NULL: features[E*N], logdur[N], pitch[N], energy[N], buckets i32[N]    */
int zvo_encoder(zvo_ctx *c, const zvo_encoder_params *p, const int32_t *ids, const int32_t *puncts,
                const float *style, float *hidden, int32_t *n_frames,
                float *features, float *logdur, float *pitch, float *energy,
                int32_t *pitch_bucket, int32_t *energy_bucket);

/* length regulator alone (src/fs2encoder.cpp:611-654) — used to teacher-force GPU outputs */
int zvo_length_regulator(const float *features, const float *logdur, int N, int E, int T, float *hidden);

/* ---- primitives exported for kernel-level tests ---- */

/* ggml_conv_1d semantics (ggml.c:3769-3786): x cf [IC][L], w f16 [OC][IC][K], out cf [OC][OL],
 * OL = L + 2*pad - dil*(K-1); bias (may be NULL) added afterwards in f32                          */
int zvo_conv1d(zvo_ctx *c, const float *x, int L, int IC, const uint16_t *w, int OC, int K,
               int pad, int dil, const float *bias, float *out);
/* ggml_norm over the last (contiguous) axis, eps inside the sqrt (ggml-cpu.c:6880-6929) */
void zvo_norm_rows(const float *x, int rows, int n, float eps, float *y);

/* ---- one layer at a time (teacher-forced per-layer parity): x [rows][cols] time-major -> out time-major.
 * style / E: AdaIN style vector (decoder blocks 2..6); H: heads, ksz: FFN kernel sizes (encoder) or {vp_kernel} */
enum { ZVO_LAYER_VOC_RESBLOCK = 0, ZVO_LAYER_ENC_FFT = 1, ZVO_LAYER_DEC_BLOCK = 2, ZVO_LAYER_VAR_PRED = 3,
       ZVO_LAYER_VOC_UPSAMPLE = 4,   /* leaky_relu(0.1) + conv_transpose1d `index`: [L][Cin] -> [L*s][Cout] (src/hifigan.cpp:22-71,281-297) */
       ZVO_LAYER_VOC_INPUT = 5,      /* (mel - mean) / scale + input conv k7: [T][80] -> [T][C0] (src/hifigan.cpp:242-265) */
       ZVO_LAYER_VOC_OUTPUT = 6,     /* leaky_relu(0.01) + output conv k7 + tanh: [L][C] -> [L] (src/hifigan.cpp:324-345) */
       ZVO_LAYER_DEC_ASR_RES = 7,    /* asr_res conv 1x1 + InstanceNorm: [T][E] -> [T][64] (src/stylettsdec.cpp:382-396) */
       ZVO_LAYER_DEC_TO_OUT = 8,     /* to_out conv 1x1 + bias: [T][E] -> [T][80] (src/stylettsdec.cpp:432-441) */
       ZVO_LAYER_ENC_EMBED = 9,      /* word + punctuation embedding + positional encoding: [N][2] (id, punct as floats) -> [N][E] (src/fs2encoder.cpp:306-324) */
       ZVO_LAYER_ENC_MHA = 10,       /* MultiHeadAttention `index` alone, with its residual + LayerNorm: [N][E] -> [N][E] (src/fs2encoder.cpp:71-140) */
       ZVO_LAYER_ENC_FFN = 11,       /* PositionwiseFeedForward `index` alone, with its residual + LayerNorm (src/fs2encoder.cpp:174-228) */
       ZVO_LAYER_DEC_ADAIN = 12 };   /* AdaIN1d alone, index = 2 * decode block + (norm - 1): [T][C] -> [T][C] (src/stylettsdec.cpp:171-200) */
int zvo_layer(zvo_ctx *c, int kind, int index, const float *x, int rows, int cols, const float *style, int E, int H,
              const int *ksz, float *out);

#ifdef __cplusplus
}
#endif
#endif

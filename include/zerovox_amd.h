/* zerovox_amd.h — C-ABI of the MI355X-native ZeroVox hot path (libzerovox_amd.so).
 *
 * This is the drop-in boundary: plain pointers, sizes and status codes, no ggml / HIP / torch
 * types.  Each entry point replaces one piece of the reference's C++ class API
 * (/root/reference/src/zerovox.h, namespace ZeroVOX); the C++ facade with the reference's own
 * class names and method signatures (zerovox.cpp_amd/csrc/zerovox.h) is a thin layer over it.
 *
 *   zv_model_load      <- ZeroVOXModel::ZeroVOXModel(fname)   src/zerovox.cpp:21-179 (GGUF KV ->
 *                         hparams, weight upload, stage construction)
 *   zv_encode          <- FS2Encoder::eval                    src/zerovox.h:191, src/fs2encoder.cpp:594-656
 *   zv_decode          <- StyleTTSDecoder::eval               src/zerovox.h:323, src/stylettsdec.cpp:457-470
 *   zv_vocode          <- HiFiGAN::eval                       src/zerovox.h:378, src/hifigan.cpp:358-377
 *   zv_synthesize      <- ZeroVOXModel::eval                  src/zerovox.cpp:198-335 (three stages back to back)
 *   zv_write_wav       <- ZeroVOXModel::write_wav_file        src/zerovox.cpp:337-391 (PCM16 mono RIFF)
 *   zv_last_error      <- std::runtime_error / die_fmt / GGML_ASSERT messages (src/zerovox.h:435-455)
 *
 * Differences that are deliberate: the number of phonemes N and the number of frames
 * T are run-time arguments (the reference fixes them at graph-build time: MAX_N_PHONEMES, max_seq_len);
 * results for a given (N, T) are those of a reference instance built for exactly that (N, T) — there
 * is no attention mask and InstanceNorm statistics run over all T frames (SURVEY Appx C-H2).
 * Bad ids / sizes return ZV_ERR_ARG instead of aborting.
 *
 * Threading: calls on one zv_model are serialised by the caller (one HIP stream per model);
 * several models per process are allowed (e.g. one per GPU).
 */
#ifndef ZEROVOX_AMD_H
#define ZEROVOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zv_model zv_model;

typedef enum
{
    ZV_OK = 0,
    ZV_ERR_IO = 1,             /* file cannot be opened / read                          */
    ZV_ERR_FORMAT = 2,         /* not a GGUF v3 file, bad KV type, truncated            */
    ZV_ERR_MISSING = 3,        /* required KV key or tensor not found                   */
    ZV_ERR_SHAPE = 4,          /* tensor shape / dtype does not match the contract      */
    ZV_ERR_ARG = 5,            /* bad argument (null pointer, id out of range, T = 0 …) */
    ZV_ERR_DEVICE = 6,         /* HIP runtime error, no gfx950 device, kernel failure   */
    ZV_ERR_OOM = 7
} zv_status;

/* the reference's zerovox_hparams (src/zerovox.h:39-58) + the derived vocoder geometry */
typedef struct
{
    uint32_t max_seq_len;
    uint32_t emb_dim;
    uint32_t punct_emb_dim;
    uint32_t decoder_n_head;
    uint32_t conv_filter_size;
    uint32_t conv_kernel_size[2];
    uint32_t encoder_layer;
    uint32_t encoder_head;
    uint32_t encoder_vp_filter_size;
    uint32_t encoder_vp_kernel_size;
    uint32_t encoder_ve_n_bins;
    uint32_t audio_sampling_rate;
    uint32_t audio_num_mels;
    uint32_t audio_hop_size;
    /* inferred from tensor shapes (the reference hard-codes these, src/zerovox.cpp:127-138) */
    uint32_t voc_channels;
    uint32_t voc_num_upsamples;
    uint32_t voc_upsample_scales[8];
    uint32_t voc_num_resblocks;
    uint32_t voc_resblock_kernels[8];
} zv_hparams;

const char *zv_last_error(void);                 /* thread-local message of the last failure */
const char *zv_version(void);

/* ---- model life cycle ------------------------------------------------------------------- */
zv_status zv_model_load(const char *gguf_path, int device, zv_model **out);
void      zv_model_free(zv_model *m);
zv_status zv_model_get_hparams(const zv_model *m, zv_hparams *out);
/* size the activation arena up-front for the largest (N, T) that will be used (optional:
 * the arena also grows on demand, outside any timed / captured region) */
zv_status zv_model_reserve(zv_model *m, uint32_t max_phonemes, uint32_t max_frames);

/* ---- the hot path: host buffers in / out, synchronous (same protocol as the reference) ---- */
/* ids[n], puncts[n] i32; style[E] f32; hidden[T*E] f32 frame-major, zero-padded tail; returns the
 * regulator's frame count in *n_frames (may be NULL).  Optional taps (NULL to skip): logdur[n],
 * pitch_bucket[n], energy_bucket[n], features[n*E] — the pre-regulator values parity tests need. */
zv_status zv_encode(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style,
                    uint32_t n, uint32_t T, float *hidden, uint32_t *n_frames);
/* n = the encoder's max_n_phonemes: all n ids are embedded and attended to (no mask, src/fs2encoder.cpp:103-110,598-600);
 * num_phonemes <= n = how many of them the length regulator walks (FS2Encoder::eval's argument, :622).  zv_encode is
 * the num_phonemes == n case, which is what the reference's only caller passes (src/zerovox.cpp:200). */
zv_status zv_encode_taps(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style,
                         uint32_t n, uint32_t num_phonemes, uint32_t T, float *hidden, uint32_t *n_frames,
                         float *features, float *logdur, float *pitch, float *energy, int32_t *pitch_bucket,
                         int32_t *energy_bucket);
/* hidden[T*E], style[E] -> mel[T*num_mels] frame-major */
zv_status zv_decode(zv_model *m, const float *hidden, const float *style, uint32_t T, float *mel);
/* mel[T*num_mels] -> wav[T*hop_size] */
zv_status zv_vocode(zv_model *m, const float *mel, uint32_t T, float *wav);

/* ---- "next" row f-3 (streaming): chunked vocoding with halo ------------------------------------
 * Vocodes mel[T][n_mels] in chunks of chunk_frames frames and hands every finished chunk to `sink` (called on the
 * calling thread, in order, with a pointer that is valid only during the call).  Each chunk is computed from its own
 * frames plus zv_vocoder_halo_frames() frames of context on either side, so the samples are bit-identical to those of
 * zv_vocode() on the whole mel (every vocoder kernel sums in an order that does not depend on the tile or on T); the
 * first audio is available after one chunk instead of after the whole utterance. */
typedef void (*zv_wav_sink)(void *user, const float *wav, uint64_t first_sample, uint64_t n_samples);
zv_status zv_vocode_stream(zv_model *m, const float *mel, uint32_t T, uint32_t chunk_frames, zv_wav_sink sink, void *user);
/* frames of context (per side) outside which a mel frame cannot influence a sample: the vocoder's receptive field */
uint32_t  zv_vocoder_halo_frames(zv_model *m);
/* encoder -> decoder -> vocoder with every intermediate kept in HBM; wav[T*hop_size] */
zv_status zv_synthesize(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style,
                        uint32_t n, uint32_t T, float *wav, uint32_t *n_frames);

/* n_utt independent utterances, each with its own (n_phonemes[u], T[u]): bit for bit the result of n_utt zv_synthesize
 * calls (no batch padding: padding would change the numbers, SURVEY Appx C-H2).  Up to 64 utterances / 64 Ki frames go
 * through the three stages as ONE launch per kernel: tensors are row-concatenated over the group and a segment table in
 * HBM tells every kernel where each utterance begins and ends, so the launches have many rounds of workgroups and the
 * whole group — input upload included — is one hipGraph (captured once per capacity bucket when graph mode is on).  For a
 * large group the graph ends before the last vocoder stage's residual blocks: those and the output conv run in
 * utterance sub-groups and a finished sub-group's waveforms are copied to the host (second stream) and into wav[] while
 * the next one computes; wav[u] is complete when the call returns, as before.
 * BASELINE.json configs[3]/[4]; with several GPUs the caller shards the list (one model per GPU, no collective). */
zv_status zv_synthesize_batch(zv_model *m, uint32_t n_utt, const int32_t *const *ids, const int32_t *const *puncts,
                              const float *const *styles, const uint32_t *n_phonemes, const uint32_t *T,
                              float *const *wav, uint32_t *n_frames);

/* The two halves of zv_synthesize_batch for a serving loop that keeps a batch in flight per lane (additive):
 * _begin builds the input block, enqueues the upload, the kernels and the waveform downloads of ONE launch group (at most
 * 64 utterances / 64 Ki frames of capacity, else ZV_ERR_ARG) on lane `lane` (< ZV_BATCH_LANES; each lane owns a stream, an
 * activation arena and a pinned staging block) and returns; _end waits for that batch and fills wav[] / n_frames[], which
 * — like T[] and the pointer arrays' targets — must stay valid until then.  While one lane's last kernels and downloads
 * run, the next lane's upload and first kernels already do.  Results are those of zv_synthesize_batch, bit for bit.
 * Two batches in flight (begin k, end k - 1) keep the GPU busy all the time (zv_batch_timeline: no gap); more lanes work
 * and buy nothing.
 * Every other entry point may be called in between (they use lane 0's stream: not while lane 0 has a batch in flight). */
#define ZV_BATCH_LANES 4
zv_status zv_synthesize_batch_begin(zv_model *m, uint32_t lane, uint32_t n_utt, const int32_t *const *ids,
                                    const int32_t *const *puncts, const float *const *styles, const uint32_t *n_phonemes,
                                    const uint32_t *T, float *const *wav, uint32_t *n_frames);
zv_status zv_synthesize_batch_end(zv_model *m, uint32_t lane);
/* When the last batches ran on the GPU (measurement): for the most recent min(cap, batches begun, 64) batches, oldest first,
 * the times in ms — relative to the first one's start — at which the batch's first operation started and its last kernel
 * ended (HIP events on the lanes' streams; waits for every lane first).  The gaps of the union of [start, end] are the time the
 * GPU had no batch to work on: bench.py reports them as extra.gpu_idle_ms_per_step. */
zv_status zv_batch_timeline(zv_model *m, uint32_t cap, double *start_ms, double *end_ms, uint32_t *n);

/* longest single utterance in frames (buffer descriptors address one utterance with 32-bit byte offsets; at most
 * 32768 frames = 7.4 min of audio).  Longer T returns ZV_ERR_ARG; zv_vocode_stream has no such limit on the total. */
uint32_t  zv_max_frames(const zv_model *m);

/* the utterance the reference's ZeroVOXModel::eval() hard-codes (src/zerovox.cpp:204-314): 120 phoneme ids, 120
 * punctuation ids, a 528-float style vector.  Pointers to static data; any argument may be NULL. */
void      zv_demo_utterance(const int32_t **ids, const int32_t **puncts, const float **style, uint32_t *n_phonemes,
                            uint32_t *style_len);

/* ---- device-resident variants (inputs already in HBM) ------------------------------------------
 * zv_vocode_device / zv_decode_device enqueue on LANE 0's stream and return; zv_memcpy_h2d / _d2h copy on that same stream
 * (so they are ordered with those calls whatever lane a batch was last begun on) and wait for the copy; zv_synchronize waits
 * for EVERY lane.  zv_profile_begin / _end bracket work of the synchronous entry points (lane 0). */
void     *zv_device_alloc(zv_model *m, size_t bytes);
void      zv_device_free(zv_model *m, void *p);
zv_status zv_memcpy_h2d(zv_model *m, void *dst, const void *src, size_t bytes);
zv_status zv_memcpy_d2h(zv_model *m, void *dst, const void *src, size_t bytes);
zv_status zv_vocode_device(zv_model *m, const float *d_mel, uint32_t T, float *d_wav);
zv_status zv_decode_device(zv_model *m, const float *d_hidden, const float *d_style, uint32_t T, float *d_mel);
zv_status zv_synchronize(zv_model *m);
/* capture the vocoder schedule for a given T into a hipGraph and replay it on later calls with the
 * same (T, d_mel, d_wav); 0 turns graph replay off */
zv_status zv_set_graph_mode(zv_model *m, int on);

/* ---- measurement ------------------------------------------------------------------------ */
/* per-kernel-family timing measured with HIP events on the model's stream (eager launches) */
typedef struct
{
    char     name[48];
    uint32_t launches;
    double   total_ms;
    double   algo_bytes;       /* algorithmic bytes moved by those launches */
    double   algo_flops;
} zv_kernel_stat;
zv_status zv_profile_begin(zv_model *m);
/* stops profiling, copies up to cap entries, returns the entry count in *n */
zv_status zv_profile_end(zv_model *m, zv_kernel_stat *stats, uint32_t cap, uint32_t *n);

/* ---- one layer at a time (tests) — the counterpart of the reference's tensor_dbg (src/utils.cpp:19-44): runs ONE layer of
 * the production schedule on a caller-supplied input, so a per-layer comparison against the reference semantics does not
 * compound.  x [rows][cin] and out [rows][cout] are host, time-major, unpadded; rows are rows at the layer's own rate.
 *   ZV_LAYER_VOC_RESBLOCK  index n = stage * num_resblocks + branch: HiFiGANResidualBlock n (src/hifigan.cpp:74-185), fused
 *                          kernels; rows must be a multiple of the stage's samples per frame
 *   ZV_LAYER_ENC_FFT       index l: FFTBlock l = attention sublayer + conv FFN (src/fs2encoder.cpp:71-140,174-228)
 *   ZV_LAYER_DEC_BLOCK     index 0,1: ResBlk1d encode.{0,1}; 2..6: AdainResBlk1d decode.{0..4} with `style`
 *                          (src/stylettsdec.cpp:69-149,242-304); cin as the reference (decode.0..2 take the 2E+64 concat)
 *   ZV_LAYER_VAR_PRED      index 0 duration / 1 pitch / 2 energy: VariancePredictor (src/fs2encoder.cpp:386-440), out [rows]
 *   ZV_LAYER_VOC_UPSAMPLE  index i: leaky_relu(0.1) + conv_transpose1d i (src/hifigan.cpp:22-71,281-297); x [rows][Cin] at the
 *                          stage's INPUT rate (rows a multiple of it), out [rows * scale_i][Cout]
 *   ZV_LAYER_VOC_INPUT     mel normalisation + input conv k7 (src/hifigan.cpp:242-265): x [T][num_mels] -> out [T][channels]
 *   ZV_LAYER_VOC_OUTPUT    leaky_relu(0.01) + output conv k7 + tanh (src/hifigan.cpp:324-345): x [T * hop][C_last] (the MRF mean)
 *                          -> out [T * hop]
 *   ZV_LAYER_DEC_ASR_RES   asr_res: conv 1x1 + InstanceNorm (src/stylettsdec.cpp:382-396): x [T][E] -> out [T][64]
 *   ZV_LAYER_DEC_TO_OUT    to_out: conv 1x1 + bias (src/stylettsdec.cpp:432-441): x [T][E] -> out [T][num_mels]
 *   ZV_LAYER_ENC_EMBED     word + punctuation embedding + positional encoding (src/fs2encoder.cpp:306-324): x [N][2] = (phoneme id,
 *                          punctuation id) as floats -> out [N][E]
 *   ZV_LAYER_ENC_MHA       index l: MultiHeadAttention l alone, with its residual + LayerNorm (src/fs2encoder.cpp:71-140)
 *   ZV_LAYER_ENC_FFN       index l: PositionwiseFeedForward l alone, with its residual + LayerNorm (src/fs2encoder.cpp:174-228)
 *   ZV_LAYER_DEC_ADAIN     index 2 * b + (norm - 1): AdaIN1d norm1 / norm2 of AdainResBlk1d decode.b alone, with `style`
 *                          (src/stylettsdec.cpp:171-200): x [T][C] -> out [T][C], C = the block's cin (norm1) / cout (norm2) */
typedef enum { ZV_LAYER_VOC_RESBLOCK = 0, ZV_LAYER_ENC_FFT = 1, ZV_LAYER_DEC_BLOCK = 2, ZV_LAYER_VAR_PRED = 3,
               ZV_LAYER_VOC_UPSAMPLE = 4, ZV_LAYER_VOC_INPUT = 5, ZV_LAYER_VOC_OUTPUT = 6, ZV_LAYER_DEC_ASR_RES = 7,
               ZV_LAYER_DEC_TO_OUT = 8, ZV_LAYER_ENC_EMBED = 9, ZV_LAYER_ENC_MHA = 10, ZV_LAYER_ENC_FFN = 11,
               ZV_LAYER_DEC_ADAIN = 12 } zv_layer_kind;
zv_status zv_debug_layer(zv_model *m, int kind, int index, const float *x, uint32_t rows, const float *style, float *out);

/* ---- test / measurement switches (none is needed in production; no reference counterpart: the reference's only run-time
 * switch is the thread count, src/zerovox.cpp:86-91).  The shipped library never reads the environment: a switch changes only
 * through this call (the Python test binding forwards ZV_* environment variables through it so that a shell script can A/B a
 * run).  Schedule switches (ZV_NO_FUSE, ZV_NO_TRIPLE, ZV_FUSE256, ZV_NO_MERGE, ZV_TAIL_GROUPS) are sampled when a model is
 * loaded, kernel-regime switches at every launch; a captured hipGraph replays the regime it was captured in, so every call of
 * zv_debug_set makes the models capture anew.  name == NULL resets every switch to its built-in default.  ZV_ERR_ARG for an
 * unknown name.  The list: zerovox.cpp_amd/csrc/knobs.h (timing-only ablation switches that give wrong results exist only in
 * diagnostic builds, -DZV_DIAG).  zv_debug_get reads a switch. */
zv_status zv_debug_set(const char *name, int value);
zv_status zv_debug_get(const char *name, int *value);

/* ---- GGUF inspection without a device (loader half of the boundary; used by the CPU test-suite) ----
 * Parses the file exactly as zv_model_load does and reports the counts; *max_seq_len receives the
 * `<arch>.max_seq_len` KV.  tensor_index >= 0 additionally returns that tensor's name (<= 63 chars + NUL),
 * ggml type code and shape (ne[4], missing dims = 1). */
zv_status zv_gguf_inspect(const char *gguf_path, uint32_t *n_tensors, uint32_t *max_seq_len, int tensor_index,
                          char *name_out, uint32_t *type_out, int64_t *ne_out);

/* ---- "next" row f-2: WAV writer (PCM16 mono, 44-byte RIFF header) --------------------------- */
zv_status zv_write_wav(const char *path, const float *wav, size_t n_samples, uint32_t sampling_rate);

#ifdef __cplusplus
}
#endif
#endif

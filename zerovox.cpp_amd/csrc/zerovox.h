// zerovox.h — host-side mirror of the reference's stage API (namespace ZeroVOX), over the C-ABI.
//
// Same class names, constructor argument positions and eval() signatures as the reference's
// src/zerovox.h:173-223 (FS2Encoder), :310-360 (StyleTTSDecoder), :362-402 (HiFiGAN), :405-430
// (ZeroVOXModel), so that a caller written like the reference's src/zerovox.cpp compiles against this
// header unchanged apart from the two ggml handle types, which cannot travel:
//
//     reference                       here
//     ggml_context  &ctx_w      ->    ZeroVOX::weights_t &ctx_w     (the loaded model: weights in HBM)
//     ggml_backend_t backend    ->    ZeroVOX::backend_t  backend   (the device the model lives on)
//
// With -DZEROVOX_GGML_COMPAT_NAMES the two names are also typedef'ed as ggml_context / ggml_backend_t.
//
// Behavioural notes: the geometry arguments are checked against what the GGUF file
// says (a mismatch throws instead of silently building a different graph); eval() is synchronous;
// errors are std::runtime_error (zv::Error) — never exit() or abort().
#pragma once

#include <cinttypes>
#include <stdexcept>
#include <string>

#include "../../include/zerovox_amd.h"

namespace ZeroVOX
{

// the GGUF KV names, verbatim from the file format contract (reference src/zerovox.h:17-33)
#define HPARAM_MAX_SEQ_LEN            "zerovox-resnet-fs2-styletts.max_seq_len"
#define HPARAM_EMB_DIM                "zerovox-resnet-fs2-styletts.emb_dim"
#define HPARAM_PUNCT_EMB_DIM          "zerovox-resnet-fs2-styletts.punct_emb_dim"
#define HPARAM_DECODER_N_HEAD         "zerovox-resnet-fs2-styletts.decoder.n_head"
#define HPARAM_CONV_FILTER_SIZE       "zerovox-resnet-fs2-styletts.decoder.conv_filter_size"
#define HPARAM_CONV_KERNEL_SIZE_0     "zerovox-resnet-fs2-styletts.decoder.conv_kernel_size.0"
#define HPARAM_CONV_KERNEL_SIZE_1     "zerovox-resnet-fs2-styletts.decoder.conv_kernel_size.1"
#define HPARAM_ENCODER_LAYER          "zerovox-resnet-fs2-styletts.encoder.layer"
#define HPARAM_ENCODER_HEAD           "zerovox-resnet-fs2-styletts.encoder.head"
#define HPARAM_ENCODER_VP_FILTER_SIZE "zerovox-resnet-fs2-styletts.encoder.vp_filter_size"
#define HPARAM_ENCODER_VP_KERNEL_SIZE "zerovox-resnet-fs2-styletts.encoder.vp_kernel_size"
#define HPARAM_ENCODER_VE_N_BINS      "zerovox-resnet-fs2-styletts.encoder.ve_n_bins"
#define HPARAM_AUDIO_NUM_MELS         "zerovox-resnet-fs2-styletts.audio.num_mels"
#define HPARAM_AUDIO_HOP_SIZE         "zerovox-resnet-fs2-styletts.audio.hop_size"
#define HPARAM_AUDIO_SAMPLING_RATE    "zerovox-resnet-fs2-styletts.audio.sampling_rate"

const int NUM_PHONEMES   = 154;
const int NUM_PUNCTS     =   6;
const int MAX_N_PHONEMES = 120;

typedef zv_hparams zerovox_hparams;      // superset of the reference struct (src/zerovox.h:39-58)

typedef zv_model  weights_t;
typedef zv_model *backend_t;

class FS2Encoder
{
  public:
    FS2Encoder(weights_t &ctx_w, backend_t backend, uint32_t max_n_phonemes, uint32_t embed_dim,
               uint32_t punct_embed_dim, uint32_t encoder_layer, uint32_t encoder_head, uint32_t conv_filter_size,
               uint32_t conv_kernel_size[2], uint32_t vp_kernel_size, uint32_t ve_n_bins, uint32_t max_seq_len);
    ~FS2Encoder() = default;

    // x must hold max_seq_len * (embed_dim + punct_embed_dim) floats; returns the regulator's frame count.
    // src_seq_data / puncts_data must hold max_n_phonemes entries, all of which are encoded (no mask).
    uint32_t eval(const int32_t *src_seq_data, const int32_t *puncts_data, const float *style_embed_data,
                  uint32_t num_phonemes, float *x);

  private:
    zv_model *model;
    uint32_t  max_n_phonemes, max_seq_len;
};

class StyleTTSDecoder
{
  public:
    StyleTTSDecoder(weights_t &ctx_w, backend_t backend, uint32_t max_seq_len, uint32_t dim_in, uint32_t style_dim,
                    uint32_t residual_dim, uint32_t dim_out);
    ~StyleTTSDecoder() = default;

    void eval(const float *enc_seq_data, const float *spk_emb_data, float *mel);

  private:
    zv_model *model;
    uint32_t  max_seq_len;
};

class HiFiGAN
{
  public:
    HiFiGAN(weights_t &ctx_w, backend_t backend, uint32_t max_seq_len, uint32_t in_channels, uint32_t hop_size,
            uint32_t kernel_size, int num_upsamples, const int *upsample_scales, int num_resblocks,
            int num_resblock_dilations, const int64_t *resblock_dilations);

    void eval(const float *mel, float *wav);

  private:
    zv_model *model;
    uint32_t  max_seq_len;
};

class ZeroVOXModel
{
  public:
    ZeroVOXModel(const std::string &fname);
    ~ZeroVOXModel();

    // the reference synthesises a hard-coded sentence (src/zerovox.cpp:198-335); same here, plus an
    // overload that takes the utterance
    void eval(void);
    void eval(const int32_t *src_seq, const int32_t *puncts, const float *style_embed, uint32_t num_phonemes);

    bool write_wav_file(const std::string &fname);

    const zerovox_hparams &get_hparams() const { return hparams; }
    const float *get_wav() const { return wav; }
    uint32_t     get_num_frames() const { return n_frames; }

  private:
    void release();
    zerovox_hparams  hparams;
    zv_model        *model;
    FS2Encoder      *encoder;
    StyleTTSDecoder *decoder;
    HiFiGAN         *meldec;
    float           *hidden_state;
    float           *mel;
    float           *wav;
    uint32_t         n_frames;
};

}  // namespace ZeroVOX

#ifdef ZEROVOX_GGML_COMPAT_NAMES
typedef ZeroVOX::weights_t ggml_context;
typedef ZeroVOX::backend_t ggml_backend_t;
#endif

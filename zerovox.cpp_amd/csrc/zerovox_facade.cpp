// zerovox_facade.cpp — ZeroVOX:: classes of csrc/zerovox.h implemented over the C-ABI.
#include "zerovox.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace ZeroVOX
{

static void chk(zv_status st)
{
    if (st != ZV_OK) throw zv::Error(st, zv_last_error());
}

static void expect(bool ok, const char *what)
{
    if (!ok) throw zv::Error(ZV_ERR_SHAPE, std::string("constructor argument does not match the GGUF file: ") + what);
}

FS2Encoder::FS2Encoder(weights_t &ctx_w, backend_t backend, uint32_t max_n_phonemes_, uint32_t embed_dim,
                       uint32_t punct_embed_dim, uint32_t encoder_layer, uint32_t encoder_head, uint32_t conv_filter_size,
                       uint32_t conv_kernel_size[2], uint32_t vp_kernel_size, uint32_t ve_n_bins, uint32_t max_seq_len_)
    : model(&ctx_w), max_n_phonemes(max_n_phonemes_), max_seq_len(max_seq_len_)
{
    (void)backend;
    zv_hparams hp;
    chk(zv_model_get_hparams(model, &hp));
    expect(hp.emb_dim == embed_dim && hp.punct_emb_dim == punct_embed_dim, "embed_dim / punct_embed_dim");
    expect(hp.encoder_layer == encoder_layer && hp.encoder_head == encoder_head, "encoder_layer / encoder_head");
    expect(hp.conv_filter_size == conv_filter_size, "conv_filter_size");
    expect(hp.conv_kernel_size[0] == conv_kernel_size[0] && hp.conv_kernel_size[1] == conv_kernel_size[1], "conv_kernel_size");
    expect(hp.encoder_vp_kernel_size == vp_kernel_size && hp.encoder_ve_n_bins == ve_n_bins, "vp_kernel_size / ve_n_bins");
    if (max_n_phonemes == 0 || max_seq_len == 0) throw zv::Error(ZV_ERR_ARG, "max_n_phonemes and max_seq_len must be > 0");
    chk(zv_model_reserve(model, max_n_phonemes, max_seq_len));
}

uint32_t FS2Encoder::eval(const int32_t *src_seq_data, const int32_t *puncts_data, const float *style_embed_data,
                          uint32_t num_phonemes, float *x)
{
    // The reference graph always encodes max_n_phonemes tokens (no mask, src/fs2encoder.cpp:103-110,598-600: all
    // max_n_phonemes ids are uploaded) and the regulator walks the first num_phonemes of them (:622).
    if (num_phonemes > max_n_phonemes)
        throw zv::Error(ZV_ERR_ARG, "FS2Encoder::eval: num_phonemes exceeds max_n_phonemes");
    uint32_t n_frames = 0;
    chk(zv_encode_taps(model, src_seq_data, puncts_data, style_embed_data, max_n_phonemes, num_phonemes, max_seq_len, x,
                       &n_frames, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    return n_frames;
}

StyleTTSDecoder::StyleTTSDecoder(weights_t &ctx_w, backend_t backend, uint32_t max_seq_len_, uint32_t dim_in,
                                 uint32_t style_dim, uint32_t residual_dim, uint32_t dim_out)
    : model(&ctx_w), max_seq_len(max_seq_len_)
{
    (void)backend;
    (void)residual_dim;        // taken from the asr_res tensor shape; the reference hard-codes 64
    zv_hparams hp;
    chk(zv_model_get_hparams(model, &hp));
    expect(hp.emb_dim + hp.punct_emb_dim == dim_in && dim_in == style_dim, "dim_in / style_dim");
    expect(hp.audio_num_mels == dim_out, "dim_out");
    if (max_seq_len == 0) throw zv::Error(ZV_ERR_ARG, "max_seq_len must be > 0");
    chk(zv_model_reserve(model, 1, max_seq_len));
}

void StyleTTSDecoder::eval(const float *enc_seq_data, const float *spk_emb_data, float *mel)
{
    chk(zv_decode(model, enc_seq_data, spk_emb_data, max_seq_len, mel));
}

HiFiGAN::HiFiGAN(weights_t &ctx_w, backend_t backend, uint32_t max_seq_len_, uint32_t in_channels, uint32_t hop_size,
                 uint32_t kernel_size, int num_upsamples, const int *upsample_scales, int num_resblocks,
                 int num_resblock_dilations, const int64_t *resblock_dilations)
    : model(&ctx_w), max_seq_len(max_seq_len_)
{
    (void)backend;
    zv_hparams hp;
    chk(zv_model_get_hparams(model, &hp));
    expect(hp.audio_num_mels == in_channels && hp.audio_hop_size == hop_size, "in_channels / hop_size");
    expect(kernel_size == 7, "kernel_size (the stored input/output convs are k7)");
    expect((int)hp.voc_num_upsamples == num_upsamples && (int)hp.voc_num_resblocks == num_resblocks, "num_upsamples / num_resblocks");
    for (int i = 0; i < num_upsamples; i++) expect((int)hp.voc_upsample_scales[i] == upsample_scales[i], "upsample_scales");
    expect(num_resblock_dilations == 3, "num_resblock_dilations");
    for (int j = 0; j < num_resblocks; j++)
        expect(resblock_dilations[j * 3] == 1 && resblock_dilations[j * 3 + 1] == 3 && resblock_dilations[j * 3 + 2] == 5, "resblock_dilations {1,3,5}");
    if (max_seq_len == 0) throw zv::Error(ZV_ERR_ARG, "max_seq_len must be > 0");
    chk(zv_model_reserve(model, 1, max_seq_len));
}

void HiFiGAN::eval(const float *mel, float *wav)
{
    chk(zv_vocode(model, mel, max_seq_len, wav));
}

// ---------------------------------------------------------------------------------------------------

ZeroVOXModel::ZeroVOXModel(const std::string &fname)
    : model(nullptr), encoder(nullptr), decoder(nullptr), meldec(nullptr), hidden_state(nullptr), mel(nullptr), wav(nullptr), n_frames(0)
{
    int device = 0;
    if (const char *e = getenv("ZEROVOX_DEVICE")) device = atoi(e);
    chk(zv_model_load(fname.c_str(), device, &model));
    try
    {
        chk(zv_model_get_hparams(model, &hparams));
        const uint32_t emb_size = hparams.emb_dim + hparams.punct_emb_dim;
        hidden_state = new float[(size_t)hparams.max_seq_len * emb_size];
        mel = new float[(size_t)hparams.max_seq_len * hparams.audio_num_mels];
        wav = new float[(size_t)hparams.max_seq_len * hparams.audio_hop_size];
        encoder = new FS2Encoder(*model, model, MAX_N_PHONEMES, hparams.emb_dim, hparams.punct_emb_dim, hparams.encoder_layer,
                                 hparams.encoder_head, hparams.conv_filter_size, hparams.conv_kernel_size,
                                 hparams.encoder_vp_kernel_size, hparams.encoder_ve_n_bins, hparams.max_seq_len);
        decoder = new StyleTTSDecoder(*model, model, hparams.max_seq_len, emb_size, emb_size, 64, hparams.audio_num_mels);
        int scales[8];
        for (uint32_t i = 0; i < hparams.voc_num_upsamples; i++) scales[i] = (int)hparams.voc_upsample_scales[i];
        std::vector<int64_t> dil;
        for (uint32_t j = 0; j < hparams.voc_num_resblocks; j++) { dil.push_back(1); dil.push_back(3); dil.push_back(5); }
        meldec = new HiFiGAN(*model, model, hparams.max_seq_len, hparams.audio_num_mels, hparams.audio_hop_size, 7,
                             (int)hparams.voc_num_upsamples, scales, (int)hparams.voc_num_resblocks, 3, dil.data());
    }
    catch (...)
    {
        release();
        throw;
    }
}

ZeroVOXModel::~ZeroVOXModel() { release(); }

void ZeroVOXModel::release()
{
    delete encoder;
    delete decoder;
    delete meldec;
    delete[] hidden_state;
    delete[] mel;
    delete[] wav;
    encoder = nullptr; decoder = nullptr; meldec = nullptr;
    hidden_state = mel = wav = nullptr;
    if (model) zv_model_free(model);
    model = nullptr;
}

void ZeroVOXModel::eval(const int32_t *src_seq, const int32_t *puncts, const float *style_embed, uint32_t num_phonemes)
{
    if (num_phonemes != (uint32_t)MAX_N_PHONEMES)
    {
        // any length: run the C-ABI path directly with N = num_phonemes (the stage objects are pinned to MAX_N_PHONEMES)
        chk(zv_synthesize(model, src_seq, puncts, style_embed, num_phonemes, hparams.max_seq_len, wav, &n_frames));
        return;
    }
    n_frames = encoder->eval(src_seq, puncts, style_embed, num_phonemes, hidden_state);
    decoder->eval(hidden_state, style_embed, mel);
    meldec->eval(mel, wav);
}

void ZeroVOXModel::eval(void)
{
    // The reference hard-codes one utterance here (src/zerovox.cpp:204-314: 120 phoneme / punctuation ids of a German
    // sentence and a 528-float style vector from its speaker encoder); the same data drives this entry point
    // (zv_demo_utterance).  A checkpoint whose style width differs from 528 gets the first min(E, 528) values, zeros behind.
    const int32_t *ids = nullptr, *puncts = nullptr;
    const float *sty = nullptr;
    uint32_t n = 0, ns = 0;
    zv_demo_utterance(&ids, &puncts, &sty, &n, &ns);
    std::vector<float> style(hparams.emb_dim + hparams.punct_emb_dim, 0.0f);
    for (size_t i = 0; i < style.size() && i < ns; i++) style[i] = sty[i];
    eval(ids, puncts, style.data(), n);
}

bool ZeroVOXModel::write_wav_file(const std::string &fname)
{
    // like the reference, all max_seq_len * hop_size samples are written (src/zerovox.cpp:369)
    return zv_write_wav(fname.c_str(), wav, (size_t)hparams.max_seq_len * hparams.audio_hop_size, hparams.audio_sampling_rate) == ZV_OK;
}

}  // namespace ZeroVOX

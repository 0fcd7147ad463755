// A caller written the way the reference's own driver drives the three stages (construction with a weight
// registry + backend handle, then FS2Encoder::eval -> StyleTTSDecoder::eval -> HiFiGAN::eval on caller-owned
// host buffers; reference src/zerovox.cpp:104-137, 326-334) — compiled against OUR header with the ggml handle
// names mapped by -DZEROVOX_GGML_COMPAT_NAMES.  If this file builds and runs, a reference-style call site is a
// drop-in.  usage: facade_callsite model.gguf n_phonemes out.f32   (writes the waveform as raw float32)
#define ZEROVOX_GGML_COMPAT_NAMES
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../zerovox.cpp_amd/csrc/zerovox.h"

using namespace ZeroVOX;

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: facade_callsite model.gguf n_phonemes out.f32\n"); return 2; }
    try
    {
        zv_model *loaded = nullptr;
        if (zv_model_load(argv[1], 0, &loaded) != ZV_OK) throw std::runtime_error(zv_last_error());
        zerovox_hparams hparams;
        if (zv_model_get_hparams(loaded, &hparams) != ZV_OK) throw std::runtime_error(zv_last_error());

        ggml_context  *ctx_w   = loaded;          // the reference passes *ctx_w and backend to every stage
        ggml_backend_t backend = loaded;
        const uint32_t n_ph = (uint32_t)atoi(argv[2]);

        FS2Encoder *encoder = new FS2Encoder(*ctx_w, backend, n_ph, hparams.emb_dim, hparams.punct_emb_dim,
                                             hparams.encoder_layer, hparams.encoder_head, hparams.conv_filter_size,
                                             hparams.conv_kernel_size, hparams.encoder_vp_kernel_size,
                                             hparams.encoder_ve_n_bins, hparams.max_seq_len);
        uint32_t emb_size = hparams.emb_dim + hparams.punct_emb_dim;
        StyleTTSDecoder *decoder = new StyleTTSDecoder(*ctx_w, backend, hparams.max_seq_len, /*dim_in=*/emb_size,
                                                       /*style_dim=*/emb_size, /*residual_dim=*/64, hparams.audio_num_mels);
        const int kernel_size = 7;
        const int num_upsamples = 4;
        int upsample_scales[num_upsamples] = {5, 5, 4, 3};
        const int num_resblocks = 3;
        const int num_resblock_dilations = 3;
        int64_t resblock_dilations[num_resblocks * num_resblock_dilations] = {1, 3, 5, 1, 3, 5, 1, 3, 5};
        HiFiGAN *meldec = new HiFiGAN(*ctx_w, backend, hparams.max_seq_len, hparams.audio_num_mels, hparams.audio_hop_size,
                                      kernel_size, num_upsamples, upsample_scales, num_resblocks, num_resblock_dilations,
                                      resblock_dilations);

        std::vector<int32_t> src_seq(n_ph), puncts(n_ph);
        for (uint32_t i = 0; i < n_ph; i++) { src_seq[i] = 1 + (int32_t)((i * 37 + 11) % NUM_PHONEMES); puncts[i] = (int32_t)(i % 3); }
        std::vector<float> style(emb_size);
        for (uint32_t i = 0; i < emb_size; i++) style[i] = 0.05f * (float)((int)(i * 2654435761u % 201) - 100) / 100.0f;
        std::vector<float> hidden_state((size_t)hparams.max_seq_len * emb_size), mel((size_t)hparams.max_seq_len * hparams.audio_num_mels),
            wav((size_t)hparams.max_seq_len * hparams.audio_hop_size);

        uint32_t frames = encoder->eval(src_seq.data(), puncts.data(), style.data(), n_ph, hidden_state.data());
        decoder->eval(hidden_state.data(), style.data(), mel.data());
        meldec->eval(mel.data(), wav.data());

        FILE *f = fopen(argv[3], "wb");
        if (!f || fwrite(wav.data(), 4, wav.size(), f) != wav.size()) throw std::runtime_error("cannot write output");
        fclose(f);
        printf("frames %u samples %zu\n", frames, wav.size());
        delete encoder;
        delete decoder;
        delete meldec;
        zv_model_free(loaded);
    }
    catch (const std::exception &e)
    {
        fprintf(stderr, "facade_callsite: %s\n", e.what());
        return 1;
    }
    return 0;
}

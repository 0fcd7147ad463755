// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the *unmodified* reference stage classes (ZeroVOX::FS2Encoder / StyleTTSDecoder / HiFiGAN,
// /root/reference/src/{fs2encoder,stylettsdec,hifigan}.cpp) on the reference's ggml CPU backend, so
// that our CPU restatement (zv_oracle.c) and the golden fixtures under tests/golden/ are pinned to
// outputs of the reference itself.  This file is ours; it restates only the ~60-line load sequence
// of ZeroVOXModel::ZeroVOXModel (reference src/zerovox.cpp:21-179), which itself cannot be built
// here (it needs <sndfile.h> and <format>).
//
// Harness rules taken from SURVEY.md §8c / Appx C:
//   H1  compute arenas must be zero pages        -> mallopt(M_MMAP_THRESHOLD, 4096) first thing
//   H3  inputs are re-uploaded before each eval  -> the stage classes' eval() already does that
//   H10 the vocoder prints two tensors per eval  -> fd 1 is pointed at /dev/null, results on fd 2
//   one instance of each stage class per process (function-local static graph buffers)
//
// usage:
//   zvref <model.gguf> [--threads n] [--reps r] [--N n_phonemes] [--T frames]
//         [--enc ids.i32 puncts.i32 style.f32 out_prefix]
//         [--dec hidden.f32 style.f32 out_mel.f32]
//         [--voc mel.f32 out_wav.f32]
//         [--chain]   decoder input = the encoder's hidden, vocoder input = the decoder's mel (ZeroVOXModel::eval,
//                     reference src/zerovox.cpp:326-334); the --dec / --voc input files are then ignored ("-" is fine)
//         [--num n]   num_phonemes handed to FS2Encoder::eval (default N): the graph still encodes all N tokens
// All files are raw little-endian arrays.  One JSON line with timings goes to stderr.

#include <malloc.h>
#include <unistd.h>
#include <fcntl.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <cinttypes>

#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

// the encoder's graph handles (features, log_duration_prediction, gf) are private members; the
// harness only *reads* them to dump intermediates.
#define private public
#include "zerovox.h"
#undef private

using namespace ZeroVOX;

static std::vector<uint8_t> read_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "zvref: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n);
    if (n && fread(buf.data(), 1, n, f) != (size_t)n) { fprintf(stderr, "zvref: short read %s\n", path); exit(2); }
    fclose(f);
    return buf;
}

static void write_file(const std::string &path, const void *data, size_t nbytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "zvref: cannot write %s\n", path.c_str()); exit(2); }
    if (nbytes && fwrite(data, 1, nbytes, f) != nbytes) { fprintf(stderr, "zvref: short write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

static uint32_t get_u32(struct gguf_context *g, const char *key)
{
    const int kid = gguf_find_key(g, key);
    if (kid < 0) { fprintf(stderr, "zvref: key not found: %s\n", key); exit(2); }
    if (gguf_get_kv_type(g, kid) != GGUF_TYPE_UINT32) { fprintf(stderr, "zvref: key %s: wrong type\n", key); exit(2); }
    return gguf_get_val_u32(g, kid);
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    mallopt(M_MMAP_THRESHOLD, 4096);   // H1: every arena comes from fresh (zero) mmap pages

    if (argc < 2) { fprintf(stderr, "usage: zvref model.gguf [options]\n"); return 2; }
    const char *gguf_path = argv[1];

    int threads = 4, reps = 1;          // 4 = the reference's default (ggml.h GGML_DEFAULT_N_THREADS)
    int N = -1, T = -1, num = -1;
    bool chain = false;
    const char *enc_args[4] = {0}, *dec_args[3] = {0}, *voc_args[2] = {0};
    for (int i = 2; i < argc; i++)
    {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { fprintf(stderr, "zvref: %s needs %d args\n", a.c_str(), n); exit(2); } };
        if      (a == "--threads") { need(1); threads = atoi(argv[++i]); }
        else if (a == "--reps")    { need(1); reps    = atoi(argv[++i]); }
        else if (a == "--N")       { need(1); N       = atoi(argv[++i]); }
        else if (a == "--T")       { need(1); T       = atoi(argv[++i]); }
        else if (a == "--num")     { need(1); num     = atoi(argv[++i]); }
        else if (a == "--chain")   { chain = true; }
        else if (a == "--enc")     { need(4); for (int j = 0; j < 4; j++) enc_args[j] = argv[++i]; }
        else if (a == "--dec")     { need(3); for (int j = 0; j < 3; j++) dec_args[j] = argv[++i]; }
        else if (a == "--voc")     { need(2); for (int j = 0; j < 2; j++) voc_args[j] = argv[++i]; }
        else { fprintf(stderr, "zvref: unknown option %s\n", a.c_str()); return 2; }
    }

    // H10: silence printf output of the stage classes (print_tensor in HiFiGAN::eval)
    fflush(stdout);
    int devnull = open("/dev/null", O_WRONLY);
    dup2(devnull, 1);

    // ---- load sequence, following reference src/zerovox.cpp:28-56,86-91,140-172 ----
    struct ggml_context *ctx_w = nullptr;
    struct gguf_init_params gp = { /*.no_alloc =*/ true, /*.ctx =*/ &ctx_w };
    struct gguf_context *g = gguf_init_from_file(gguf_path, gp);
    if (!g) { fprintf(stderr, "zvref: gguf_init_from_file() failed\n"); return 2; }

    zerovox_hparams hp;
    hp.max_seq_len            = get_u32(g, HPARAM_MAX_SEQ_LEN);
    hp.emb_dim                = get_u32(g, HPARAM_EMB_DIM);
    hp.punct_emb_dim          = get_u32(g, HPARAM_PUNCT_EMB_DIM);
    hp.decoder_n_head         = get_u32(g, HPARAM_DECODER_N_HEAD);
    hp.conv_filter_size       = get_u32(g, HPARAM_CONV_FILTER_SIZE);
    hp.conv_kernel_size[0]    = get_u32(g, HPARAM_CONV_KERNEL_SIZE_0);
    hp.conv_kernel_size[1]    = get_u32(g, HPARAM_CONV_KERNEL_SIZE_1);
    hp.encoder_layer          = get_u32(g, HPARAM_ENCODER_LAYER);
    hp.encoder_head           = get_u32(g, HPARAM_ENCODER_HEAD);
    hp.encoder_vp_filter_size = get_u32(g, HPARAM_ENCODER_VP_FILTER_SIZE);
    hp.encoder_vp_kernel_size = get_u32(g, HPARAM_ENCODER_VP_KERNEL_SIZE);
    hp.encoder_ve_n_bins      = get_u32(g, HPARAM_ENCODER_VE_N_BINS);
    hp.audio_sampling_rate    = get_u32(g, HPARAM_AUDIO_SAMPLING_RATE);
    hp.audio_num_mels         = get_u32(g, HPARAM_AUDIO_NUM_MELS);
    hp.audio_hop_size         = get_u32(g, HPARAM_AUDIO_HOP_SIZE);

    if (T < 0) T = (int)hp.max_seq_len;
    if (N < 0) N = MAX_N_PHONEMES;
    const uint32_t E = hp.emb_dim + hp.punct_emb_dim;

    ggml_backend_t backend = ggml_backend_cpu_init();
    ggml_backend_cpu_set_n_threads(backend, threads);

    ggml_backend_buffer_t buf_w = ggml_backend_alloc_ctx_tensors(ctx_w, backend);
    if (!buf_w) { fprintf(stderr, "zvref: ggml_backend_alloc_ctx_tensors() failed\n"); return 2; }

    // graphs are built before the weight bytes are uploaded, as the reference does
    FS2Encoder      *encoder = nullptr;
    StyleTTSDecoder *decoder = nullptr;
    HiFiGAN         *meldec  = nullptr;
    try
    {
        if (enc_args[0])
            encoder = new FS2Encoder(*ctx_w, backend, (uint32_t)N, hp.emb_dim, hp.punct_emb_dim, hp.encoder_layer,
                                     hp.encoder_head, hp.conv_filter_size, hp.conv_kernel_size,
                                     hp.encoder_vp_kernel_size, hp.encoder_ve_n_bins, (uint32_t)T);
        if (dec_args[0])
            decoder = new StyleTTSDecoder(*ctx_w, backend, (uint32_t)T, E, E, 64, hp.audio_num_mels);
        if (voc_args[0])
        {
            // vocoder geometry is hard-coded in the reference (src/zerovox.cpp:127-138)
            static int     upsample_scales[4]    = {5, 5, 4, 3};
            static int64_t resblock_dilations[9] = {1, 3, 5, 1, 3, 5, 1, 3, 5};
            meldec = new HiFiGAN(*ctx_w, backend, (uint32_t)T, hp.audio_num_mels, hp.audio_hop_size, 7, 4,
                                 upsample_scales, 3, 3, resblock_dilations);
        }
    }
    catch (const std::exception &e) { fprintf(stderr, "zvref: stage construction failed: %s\n", e.what()); return 3; }

    {
        FILE *f = fopen(gguf_path, "rb");
        const int n_tensors = gguf_get_n_tensors(g);
        for (int i = 0; i < n_tensors; i++)
        {
            const char *name = gguf_get_tensor_name(g, i);
            struct ggml_tensor *t = ggml_get_tensor(ctx_w, name);
            size_t offs = gguf_get_data_offset(g) + gguf_get_tensor_offset(g, i);
            std::vector<uint8_t> buf(ggml_nbytes(t));
            if (fseek(f, (long)offs, SEEK_SET) != 0 || fread(buf.data(), 1, buf.size(), f) != buf.size())
            { fprintf(stderr, "zvref: read failed for %s\n", name); return 2; }
            ggml_backend_tensor_set(t, buf.data(), 0, buf.size());
        }
        fclose(f);
    }
    gguf_free(g);

    double t_enc = 0, t_dec = 0, t_voc = 0;
    uint32_t n_frames = 0;
    std::vector<float> chain_hidden, chain_mel;
    if (num < 0) num = N;

    if (encoder)
    {
        auto ids = read_file(enc_args[0]), pun = read_file(enc_args[1]), sty = read_file(enc_args[2]);
        if (ids.size() != (size_t)N * 4 || pun.size() != (size_t)N * 4 || sty.size() != (size_t)E * 4)
        { fprintf(stderr, "zvref: encoder input sizes do not match N=%d E=%u\n", N, E); return 2; }
        std::vector<float> hidden((size_t)T * E);
        for (int r = 0; r < reps; r++)
        {
            double t0 = now_s();
            n_frames = encoder->eval((const int32_t *)ids.data(), (const int32_t *)pun.data(), (const float *)sty.data(),
                                     (uint32_t)num, hidden.data());
            double dt = now_s() - t0;
            if (r == 0 || dt < t_enc) t_enc = dt;
        }
        std::string p = enc_args[3];
        if (chain) chain_hidden = hidden;
        write_file(p + ".hidden.f32", hidden.data(), hidden.size() * 4);
        write_file(p + ".features.f32", ggml_get_data_f32(encoder->features), (size_t)N * E * 4);
        write_file(p + ".logdur.f32", ggml_get_data_f32(encoder->log_duration_prediction), (size_t)N * 4);
        int32_t nf = (int32_t)n_frames;
        write_file(p + ".nframes.i32", &nf, 4);
        // pitch / energy predictions and their buckets: the two MAP_CUSTOM2 nodes, in graph order
        const char *names[2] = {"pitch", "energy"};
        int found = 0;
        for (int i = 0; i < ggml_graph_n_nodes(encoder->gf) && found < 2; i++)
        {
            struct ggml_tensor *nd = ggml_graph_node(encoder->gf, i);
            if (nd->op != GGML_OP_MAP_CUSTOM2) continue;
            write_file(p + "." + names[found] + ".f32", ggml_get_data_f32(nd->src[1]), (size_t)N * 4);
            write_file(p + "." + names[found] + "_bucket.i32", ggml_get_data(nd), (size_t)N * 4);
            found++;
        }
    }

    if (decoder)
    {
        auto sty = read_file(dec_args[1]);
        std::vector<uint8_t> hid;
        if (chain && !chain_hidden.empty())
            hid.assign((const uint8_t *)chain_hidden.data(), (const uint8_t *)chain_hidden.data() + chain_hidden.size() * 4);
        else
            hid = read_file(dec_args[0]);
        if (hid.size() != (size_t)T * E * 4 || sty.size() != (size_t)E * 4)
        { fprintf(stderr, "zvref: decoder input sizes do not match T=%d E=%u\n", T, E); return 2; }
        std::vector<float> mel((size_t)T * hp.audio_num_mels);
        for (int r = 0; r < reps; r++)
        {
            double t0 = now_s();
            decoder->eval((const float *)hid.data(), (const float *)sty.data(), mel.data());
            double dt = now_s() - t0;
            if (r == 0 || dt < t_dec) t_dec = dt;
        }
        if (chain) chain_mel = mel;
        write_file(dec_args[2], mel.data(), mel.size() * 4);
    }

    if (meldec)
    {
        std::vector<uint8_t> mel;
        if (chain && !chain_mel.empty())
            mel.assign((const uint8_t *)chain_mel.data(), (const uint8_t *)chain_mel.data() + chain_mel.size() * 4);
        else
            mel = read_file(voc_args[0]);
        if (mel.size() != (size_t)T * hp.audio_num_mels * 4)
        { fprintf(stderr, "zvref: vocoder input size does not match T=%d\n", T); return 2; }
        std::vector<float> wav((size_t)T * hp.audio_hop_size);
        for (int r = 0; r < reps; r++)
        {
            double t0 = now_s();
            meldec->eval((const float *)mel.data(), wav.data());
            double dt = now_s() - t0;
            if (r == 0 || dt < t_voc) t_voc = dt;
        }
        write_file(voc_args[1], wav.data(), wav.size() * 4);
    }

    fprintf(stderr, "{\"threads\": %d, \"reps\": %d, \"N\": %d, \"T\": %d, \"n_frames\": %u, "
                    "\"enc_s\": %.6f, \"dec_s\": %.6f, \"voc_s\": %.6f}\n",
            threads, reps, N, T, n_frames, t_enc, t_dec, t_voc);
    return 0;
}

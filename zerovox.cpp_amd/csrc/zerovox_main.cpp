// zerovox — command-line driver, the counterpart of the reference's main() (src/zerovox.cpp:396-406):
//     ZeroVOXModel model(g_gguf_filename); model.eval(); model.write_wav_file("foo.wav");
// Run without arguments it does exactly that (model "medium-ldec.gguf", output "foo.wav", built-in utterance).
// The reference compiles its utterance in (src/zerovox.cpp:204-314: phoneme ids, punctuation ids and a style
// vector produced by a speaker encoder that is not part of the repository); here it can also be read from a text
// file so that the binary is usable with any front end:
//
//     line 1: phoneme ids            (N integers, whitespace separated)
//     line 2: punctuation ids        (N integers)
//     line 3: style embedding        (emb_dim + punct_emb_dim floats; a single 0 means the zero vector)
//
// usage: zerovox [-m model.gguf] [-u utterance.txt] [-o out.wav] [--trim] [--info]
//   --trim   write only the frames the length regulator produced (the reference always writes max_seq_len frames,
//            src/zerovox.cpp:369)
//   --info   list the checkpoint's tensors (name, type, shape), then exit (no GPU needed)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "zerovox.h"

static const char *k_default_model = "medium-ldec.gguf";        // reference src/zerovox.cpp:16
static const char *k_default_out = "foo.wav";                   // reference src/zerovox.cpp:403

static void usage(FILE *f)
{
    fprintf(f, "usage: zerovox [-m model.gguf] [-u utterance.txt] [-o out.wav] [--trim] [--info]\n"
               "  defaults: -m %s -o %s, built-in utterance (like the reference's main)\n"
               "  utterance.txt: line 1 phoneme ids, line 2 punctuation ids, line 3 style floats (or a single 0)\n",
            k_default_model, k_default_out);
}

template <typename T> static std::vector<T> parse_line(const std::string &line)
{
    std::vector<T> v;
    std::istringstream is(line);
    T x;
    while (is >> x) v.push_back(x);
    if (!is.eof()) throw std::runtime_error("utterance file: malformed number in '" + line.substr(0, 40) + "'");
    return v;
}

int main(int argc, char **argv)
{
    std::string model_path = k_default_model, out_path = k_default_out, utt_path;
    bool trim = false, info = false;
    for (int i = 1; i < argc; i++)
    {
        const std::string a = argv[i];
        auto need = [&](const char *flag) -> std::string {
            if (i + 1 >= argc) { fprintf(stderr, "zerovox: %s needs a value\n", flag); usage(stderr); exit(2); }
            return argv[++i];
        };
        if (a == "-m") model_path = need("-m");
        else if (a == "-u") utt_path = need("-u");
        else if (a == "-o") out_path = need("-o");
        else if (a == "--trim") trim = true;
        else if (a == "--info") info = true;
        else if (a == "-h" || a == "--help") { usage(stdout); return 0; }
        else { fprintf(stderr, "zerovox: unknown argument '%s'\n", a.c_str()); usage(stderr); return 2; }
    }

    try
    {
        if (info)
        {
            uint32_t n_tensors = 0, max_seq_len = 0;
            if (zv_gguf_inspect(model_path.c_str(), &n_tensors, &max_seq_len, -1, nullptr, nullptr, nullptr) != ZV_OK)
                throw std::runtime_error(zv_last_error());
            printf("%s: %u tensors, max_seq_len %u\n", model_path.c_str(), n_tensors, max_seq_len);
            for (uint32_t i = 0; i < n_tensors; i++)
            {
                char name[64];
                uint32_t type = 0;
                int64_t ne[4];
                if (zv_gguf_inspect(model_path.c_str(), nullptr, nullptr, (int)i, name, &type, ne) != ZV_OK)
                    throw std::runtime_error(zv_last_error());
                printf("  %-48s %s [%lld, %lld, %lld, %lld]\n", name, type == 0 ? "f32" : (type == 1 ? "f16" : "other"),
                       (long long)ne[0], (long long)ne[1], (long long)ne[2], (long long)ne[3]);
            }
            return 0;
        }

        ZeroVOX::ZeroVOXModel model(model_path);
        const ZeroVOX::zerovox_hparams &hp = model.get_hparams();
        if (utt_path.empty())
            model.eval();
        else
        {
            std::ifstream f(utt_path);
            if (!f) throw std::runtime_error("cannot open utterance file '" + utt_path + "'");
            std::string l1, l2, l3;
            if (!std::getline(f, l1) || !std::getline(f, l2) || !std::getline(f, l3))
                throw std::runtime_error("utterance file needs three lines (ids, punctuation ids, style)");
            std::vector<int32_t> ids = parse_line<int32_t>(l1), puncts = parse_line<int32_t>(l2);
            std::vector<float> style = parse_line<float>(l3);
            const size_t E = hp.emb_dim + hp.punct_emb_dim;
            if (ids.empty() || ids.size() != puncts.size())
                throw std::runtime_error("utterance file: need as many punctuation ids as phoneme ids (> 0)");
            if (style.size() == 1 && style[0] == 0.0f) style.assign(E, 0.0f);
            if (style.size() != E)
                throw std::runtime_error("utterance file: style vector has " + std::to_string(style.size()) + " values, model needs " +
                                         std::to_string(E));
            model.eval(ids.data(), puncts.data(), style.data(), (uint32_t)ids.size());
        }

        const uint32_t nf = model.get_num_frames();
        if (trim)
        {
            const size_t n = (size_t)nf * hp.audio_hop_size;
            if (zv_write_wav(out_path.c_str(), model.get_wav(), n, hp.audio_sampling_rate) != ZV_OK)
                throw std::runtime_error(zv_last_error());
            printf("Successfully created %s with %zu samples (%u frames).\n", out_path.c_str(), n, nf);
        }
        else
        {
            if (!model.write_wav_file(out_path)) throw std::runtime_error(zv_last_error());
            printf("Successfully created %s with %zu samples.\n", out_path.c_str(), (size_t)hp.max_seq_len * hp.audio_hop_size);
        }
    }
    catch (const std::exception &e)
    {
        fprintf(stderr, "zerovox: %s\n", e.what());
        return 1;
    }
    return 0;
}

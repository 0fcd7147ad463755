// Host-side robustness driver for the loader half of the boundary (run under ASan/UBSan on the CPU build:
// scripts/asan_host.sh).  Feeds zv_gguf_inspect / zv_model_load truncated and bit-flipped copies of a valid
// checkpoint: every call must return a status code (never crash, never read out of bounds).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/zerovox_amd.h"

static std::vector<unsigned char> slurp(const char *p)
{
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    if (fread(b.data(), 1, b.size(), f) != b.size()) exit(2);
    fclose(f);
    return b;
}

static void spit(const std::string &p, const unsigned char *d, size_t n)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f || fwrite(d, 1, n, f) != n) { perror(p.c_str()); exit(2); }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_fuzz valid.gguf scratch_dir\n"); return 2; }
    const std::vector<unsigned char> good = slurp(argv[1]);
    const std::string tmp = std::string(argv[2]) + "/fuzz.gguf";
    uint32_t nt = 0, msl = 0;
    if (zv_gguf_inspect(argv[1], &nt, &msl, -1, nullptr, nullptr, nullptr) != ZV_OK) { fprintf(stderr, "valid file rejected: %s\n", zv_last_error()); return 1; }
    printf("valid: %u tensors, max_seq_len %u\n", nt, msl);
    for (uint32_t i = 0; i < nt; i++)
    {
        char name[64]; uint32_t ty; int64_t ne[4];
        if (zv_gguf_inspect(argv[1], nullptr, nullptr, (int)i, name, &ty, ne) != ZV_OK) return 1;
    }
    int ok = 0, rejected = 0;
    // header region: the KV + tensor-info block sits in front of the first tensor's data
    size_t header = good.size() < 65536 ? good.size() : 65536;
    for (size_t cut = 0; cut < header; cut += 37)          // truncations
    {
        spit(tmp, good.data(), cut);
        (zv_gguf_inspect(tmp.c_str(), &nt, &msl, 0, nullptr, nullptr, nullptr) == ZV_OK) ? ok++ : rejected++;
    }
    unsigned s = 12345;
    for (int it = 0; it < 400; it++)                        // byte corruptions inside the header
    {
        std::vector<unsigned char> b = good;
        for (int k = 0; k < 3; k++)
        {
            s = s * 1664525u + 1013904223u;
            b[(s >> 8) % header] = (unsigned char)(s >> 24);
        }
        spit(tmp, b.data(), b.size());
        zv_model *m = nullptr;
        char name[64]; uint32_t ty; int64_t ne[4];
        (zv_gguf_inspect(tmp.c_str(), &nt, &msl, 1, name, &ty, ne) == ZV_OK) ? ok++ : rejected++;
        if (zv_model_load(tmp.c_str(), 0, &m) == ZV_OK) zv_model_free(m);      // no GPU here: a status, not a crash
    }
    float wav[8] = {0.f, 0.5f, -0.5f, 2.f, -2.f, 1e-9f, 0.25f, -0.25f};
    if (zv_write_wav((std::string(argv[2]) + "/o.wav").c_str(), wav, 8, 22050) != ZV_OK) return 1;
    if (zv_write_wav("/nonexistent-dir/o.wav", wav, 8, 22050) == ZV_OK) return 1;
    remove(tmp.c_str());
    printf("mutants: %d accepted, %d rejected, no crash\n", ok, rejected);
    return 0;
}

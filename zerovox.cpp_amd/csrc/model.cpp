// model.cpp — loader + fixed kernel schedules (see model.h).
#include "model.h"
#include "knobs.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace zv
{

static const char *KV_PREFIX = "zerovox-resnet-fs2-styletts.";   // reference src/zerovox.h:17-33

// ---------------------------------------------------------------------------------------------------
// weights

void *Model::dev_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) fail(ZV_ERR_OOM, "hipMalloc(%zu) failed", bytes);
    allocs_.push_back(p);
    return p;
}

float *Model::upload_f32(const GgufTensor &t, int pad_to, float pad_value)
{
    if (t.type != GGML_F32) fail(ZV_ERR_SHAPE, "tensor %s: expected f32", t.name.c_str());
    const size_t n = (size_t)t.nelements();
    const size_t np = pad_to > 0 ? (size_t)std::max<int64_t>(pad_to, (int64_t)n) : n;
    std::vector<float> h(np + 64, pad_value);            // 64 floats of slack: prologues read whole float4 groups
    memcpy(h.data(), t.data, n * sizeof(float));
    float *d = (float *)dev_alloc(h.size() * sizeof(float));
    ZV_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

float *Model::upload_vec(const GgufFile &g, const std::string &name, int expect_n, int pad_to, float pad_value)
{
    const GgufTensor &t = g.get(name);
    if (t.nelements() != expect_n) fail(ZV_ERR_SHAPE, "tensor %s: expected %d elements, found %lld", name.c_str(), expect_n, (long long)t.nelements());
    return upload_f32(t, pad_to, pad_value);
}

// GGUF conv weight: ggml ne [K, IC, OC] f16 (k fastest), bias f32 [OC]  (SURVEY.md Appx A)
ConvW Model::load_conv(const GgufFile &g, const std::string &wname, const std::string &bname, int expect_cin, bool gemm_pack)
{
    const GgufTensor &w = g.get(wname);
    if (w.type != GGML_F16) fail(ZV_ERR_SHAPE, "tensor %s: conv weights must be f16", wname.c_str());
    ConvW c;
    c.K = (int)w.ne[0];
    c.Cin = (int)w.ne[1];
    c.Cout = (int)w.ne[2];
    if (expect_cin >= 0 && c.Cin != expect_cin) fail(ZV_ERR_SHAPE, "tensor %s: expected %d input channels, found %d", wname.c_str(), expect_cin, c.Cin);
    if ((c.K & 1) == 0) fail(ZV_ERR_SHAPE, "tensor %s: even kernel size %d is not a 'same' conv", wname.c_str(), c.K);
    c.Cin_p = round_up(c.Cin, 16);
    c.Cout_p = round_up(c.Cout, 16);
    c.ck = conv_pick_ck(c.Cin_p);
    std::vector<uint16_t> packed(packed_conv_weight_halfs(c.Cin_p, c.Cout_p, c.K));
    pack_conv_weight((const uint16_t *)w.data, c.K, c.Cin, c.Cout, c.Cin_p, c.Cout_p, c.ck, packed.data());
    c.w = dev_alloc(packed.size() * 2 + 32768);      // slack: the MFMA loops request up to 16 KiB past the last block
    ZV_HIP(hipMemcpy(c.w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    if (gemm_pack && c.Cin_p >= 256 && conv_gemm_groups(c.Cout_p) >= 1)
    {
        // batches run the wide decoder convs on conv_gemm_kernel: the same weights once more, in its stream order
        std::vector<uint16_t> p8(conv_gemm_weight_halfs(c.Cin_p, c.Cout_p, c.K));
        pack_conv_weight_gemm((const uint16_t *)w.data, c.K, c.Cin, c.Cout, c.Cin_p, c.Cout_p, p8.data());
        c.w8 = dev_alloc(p8.size() * 2);
        ZV_HIP(hipMemcpy(c.w8, p8.data(), p8.size() * 2, hipMemcpyHostToDevice));
    }
    if (!bname.empty())
    {
        const GgufTensor &b = g.get(bname);
        if (b.type != GGML_F32 || b.nelements() != c.Cout) fail(ZV_ERR_SHAPE, "tensor %s: expected f32[%d]", bname.c_str(), c.Cout);
        c.bias = upload_f32(b, round_up(c.Cout_p, 32), 0.f);
    }
    return c;
}

// ConvTranspose1d(stride s, kernel K, padding p = s/2 + s%2, output_padding s%2) as the reference defines it:
// zero-stuff + conv with the stored, already flipped kernel (src/hifigan.cpp:22-71).  Output sample
// t = q*s + r only sees stuffed positions off + i*s, i.e. taps k = off - r + (i - q)*s: per phase r a
// short conv over the *un-stuffed* input.  All s phases become one ordinary conv with s*Cout_p output
// channels (channel r*Cout_p + oc) whose channels-last output [L][s*Cout_p] IS the up-sampled
// sequence [L*s][Cout_p] — no stuffed buffer, no s-fold wasted MACs.
ConvW Model::load_upsample(const GgufFile &g, int idx, int stride, int expect_cin)
{
    char nm[96];
    snprintf(nm, sizeof(nm), "_meldec.upsamples.%d.1.w", idx);
    const GgufTensor &w = g.get(nm);
    if (w.type != GGML_F16) fail(ZV_ERR_SHAPE, "tensor %s: conv weights must be f16", nm);
    const int K = (int)w.ne[0], IC = (int)w.ne[1], OC = (int)w.ne[2];
    if (IC != expect_cin) fail(ZV_ERR_SHAPE, "tensor %s: expected %d input channels, found %d", nm, expect_cin, IC);
    const int s = stride;
    const int p = s / 2 + s % 2, op = s % 2;
    const int off = (K - 1) - p;
    // reference output length: (L-1)*s + 1 + 2*off + op - (K-1) must equal L*s
    if (2 * off + op + 1 - (K - 1) != s) fail(ZV_ERR_SHAPE, "tensor %s: kernel %d / stride %d do not give L*s outputs", nm, K, s);
    // delta = i - q over all (k, r):  k = off - r + delta*s
    int dmin = 0, dmax = 0;
    for (int r = 0; r < s; r++)
        for (int k = 0; k < K; k++)
            if ((k - off + r) % s == 0)
            {
                const int d = (k - off + r) / s;
                dmin = std::min(dmin, d);
                dmax = std::max(dmax, d);
            }
    const int nd = std::max(-dmin, dmax);          // symmetric window so the conv stays a "same" conv
    ConvW c;
    c.K = 2 * nd + 1;
    c.Cin = IC;
    c.Cin_p = round_up(IC, 16);
    const int OCp = round_up(OC, 16);
    c.Cout = s * OCp;
    c.Cout_p = s * OCp;
    // batches run the wide ones (at least one group of 8 output tiles, input channels in 64-channel blocks) on conv_gemm_kernel
    // behind an f16 operand pre-pass; its chains walk 256-channel chunks, so these convs do everywhere (same bits in every regime)
    // (at least 768 products per output element: measured 225 -> 181 + 15 us and 365 -> 192 + 75 + 50 us (kernel + leftover tiles +
    // pre-pass) for the 1 536- and 768-deep ones; the 384-deep one 496 -> 364 + 120 us — conv_gemm_kernel's one workgroup per CU
    // spends a six-unit contraction mostly in its prologue and its 256-KiB epilogue — stays on conv1d_mfma_kernel)
    const bool gemm_pack = c.Cin_p >= 128 && (c.Cin_p & 63) == 0 && conv_gemm_groups(c.Cout_p) >= 1 && c.K * c.Cin_p >= 768;
    c.ck = conv_pick_ck(c.Cin_p, gemm_pack ? 256 : 128);      // measured: the 3-input (MRF mean) prologue of these convs prefers 128-channel chunks
    // virtual weight in GGUF conv layout [OC'][IC][K'] (k fastest)
    std::vector<uint16_t> v((size_t)c.Cout * IC * c.K, 0);
    const uint16_t *src = (const uint16_t *)w.data;
    for (int r = 0; r < s; r++)
        for (int oc = 0; oc < OC; oc++)
            for (int ic = 0; ic < IC; ic++)
                for (int tp = 0; tp < c.K; tp++)
                {
                    const int k = off - r + (tp - nd) * s;
                    if (k >= 0 && k < K) v[((size_t)(r * OCp + oc) * IC + ic) * c.K + tp] = src[((size_t)oc * IC + ic) * K + k];
                }
    std::vector<uint16_t> packed(packed_conv_weight_halfs(c.Cin_p, c.Cout_p, c.K));
    pack_conv_weight(v.data(), c.K, IC, c.Cout, c.Cin_p, c.Cout_p, c.ck, packed.data());
    c.w = dev_alloc(packed.size() * 2 + 32768);      // slack: as in load_conv
    ZV_HIP(hipMemcpy(c.w, packed.data(), packed.size() * 2, hipMemcpyHostToDevice));
    if (gemm_pack)
    {
        std::vector<uint16_t> p8(conv_gemm_weight_halfs(c.Cin_p, c.Cout_p, c.K));
        pack_conv_weight_gemm(v.data(), c.K, IC, c.Cout, c.Cin_p, c.Cout_p, p8.data());
        c.w8 = dev_alloc(p8.size() * 2);
        ZV_HIP(hipMemcpy(c.w8, p8.data(), p8.size() * 2, hipMemcpyHostToDevice));
    }
    snprintf(nm, sizeof(nm), "_meldec.upsamples.%d.1.b", idx);
    const GgufTensor &b = g.get(nm);
    if (b.type != GGML_F32 || b.nelements() != OC) fail(ZV_ERR_SHAPE, "tensor %s: expected f32[%d]", nm, OC);
    std::vector<float> hb(round_up(c.Cout_p, 32) + 64, 0.f);
    for (int r = 0; r < s; r++) memcpy(hb.data() + (size_t)r * OCp, b.data, (size_t)OC * 4);
    c.bias = (float *)dev_alloc(hb.size() * 4);
    ZV_HIP(hipMemcpy(c.bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    return c;
}

Model::Model(const std::string &path, int dev) : device(dev)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(ZV_ERR_DEVICE, "no HIP device available");
    if (dev < 0 || dev >= ndev) fail(ZV_ERR_ARG, "device %d out of range (%d devices)", dev, ndev);
    ZV_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    ZV_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) fail(ZV_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
    n_cu = prop.multiProcessorCount;
    // schedule switches are fixed when the model is built (knobs.h: tests force a regime, measurements A/B one)
    no_fuse_ = knob(ZV_NO_FUSE) != 0;
    no_triple_ = knob(ZV_NO_TRIPLE) != 0;
    force_fuse256_ = knob(ZV_FUSE256) != 0;
    no_merge_ = knob(ZV_NO_MERGE) != 0;
    tail_groups_ = knob(ZV_TAIL_GROUPS);
    ZV_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    lanes_.resize(1);
    lanes_[0].stream = stream;

    GgufFile g;
    g.open(path);
    auto kv = [&](const char *k) { return g.get_u32(std::string(KV_PREFIX) + k); };
    // all 15 keys are required, as in the reference (src/zerovox.cpp:39-56)
    hp.max_seq_len = kv("max_seq_len");
    hp.emb_dim = kv("emb_dim");
    hp.punct_emb_dim = kv("punct_emb_dim");
    hp.decoder_n_head = kv("decoder.n_head");
    hp.conv_filter_size = kv("decoder.conv_filter_size");
    hp.conv_kernel_size[0] = kv("decoder.conv_kernel_size.0");
    hp.conv_kernel_size[1] = kv("decoder.conv_kernel_size.1");
    hp.encoder_layer = kv("encoder.layer");
    hp.encoder_head = kv("encoder.head");
    hp.encoder_vp_filter_size = kv("encoder.vp_filter_size");
    hp.encoder_vp_kernel_size = kv("encoder.vp_kernel_size");
    hp.encoder_ve_n_bins = kv("encoder.ve_n_bins");
    hp.audio_sampling_rate = kv("audio.sampling_rate");
    hp.audio_num_mels = kv("audio.num_mels");
    hp.audio_hop_size = kv("audio.hop_size");

    const int Ed = (int)E();
    if (Ed % 16) fail(ZV_ERR_SHAPE, "emb_dim + punct_emb_dim = %d must be a multiple of 16", Ed);
    if (hp.audio_num_mels % 16) fail(ZV_ERR_SHAPE, "num_mels = %u must be a multiple of 16", hp.audio_num_mels);
    if (hp.encoder_head == 0 || Ed % hp.encoder_head) fail(ZV_ERR_SHAPE, "encoder.head = %u does not divide %d", hp.encoder_head, Ed);
    if (hp.encoder_vp_kernel_size != 3) fail(ZV_ERR_SHAPE, "vp_kernel_size = %u: the reference pads the second predictor conv with a literal 1 (src/fs2encoder.cpp:417), only 3 is a 'same' conv", hp.encoder_vp_kernel_size);
    char nm[128], nb[128];

    // ---------------- vocoder (src/hifigan.cpp:208-218; geometry from tensor shapes) ----------------
    const int M = (int)hp.audio_num_mels;
    voc_.mean = upload_vec(g, "hifigan.mean", M);
    voc_.scale = upload_vec(g, "hifigan.scale", M, 0, 1.f);
    voc_.in_conv = load_conv(g, "_meldec.input_conv.w", "_meldec.input_conv.b", M);
    int C = voc_.in_conv.Cout;
    hp.voc_channels = C;
    int n_up = 0;
    while (n_up < 8)
    {
        snprintf(nm, sizeof(nm), "_meldec.upsamples.%d.1.w", n_up);
        if (!g.find(nm)) break;
        n_up++;
    }
    if (n_up == 0) fail(ZV_ERR_MISSING, "tensor '_meldec.upsamples.0.1.w' not found");
    // the stride is not stored in the file: the reference hard-codes {5,5,4,3} (src/zerovox.cpp:129);
    // every HiFi-GAN config has kernel = 2 * stride, which is what we derive and check against hop_size.
    int hop = 1;
    voc_.n_up = n_up;
    hp.voc_num_upsamples = n_up;
    int n_blocks = 0;
    while (true)
    {
        snprintf(nm, sizeof(nm), "_meldec.blocks.%d.convs1.0.1.w", n_blocks);
        if (!g.find(nm)) break;
        n_blocks++;
    }
    if (n_blocks == 0 || n_blocks % n_up) fail(ZV_ERR_SHAPE, "%d residual blocks do not divide over %d upsample stages", n_blocks, n_up);
    voc_.n_rb = n_blocks / n_up;
    if (voc_.n_rb != 3) fail(ZV_ERR_SHAPE, "num_resblocks = %d: the schedule (like the reference caller) is built for 3", voc_.n_rb);
    hp.voc_num_resblocks = voc_.n_rb;
    for (int i = 0; i < n_up; i++)
    {
        snprintf(nm, sizeof(nm), "_meldec.upsamples.%d.1.w", i);
        const int K = (int)g.get(nm).ne[0];
        if (K % 2) fail(ZV_ERR_SHAPE, "tensor %s: odd transposed-conv kernel %d", nm, K);
        const int s = K / 2;
        voc_.scales[i] = s;
        hp.voc_upsample_scales[i] = s;
        hop *= s;
        voc_.ups[i] = load_upsample(g, i, s, C);
        // the schedule's buffers are sized for channel halving per stage (every HiFi-GAN generator; 512 -> 32 here)
        if ((int)g.get(nm).ne[2] * 2 != C) fail(ZV_ERR_SHAPE, "tensor %s: %lld output channels, expected %d (channels halve per upsample stage)", nm, (long long)g.get(nm).ne[2], C / 2);
        C = (int)g.get(nm).ne[2];
        for (int j = 0; j < voc_.n_rb; j++)
            for (int d = 0; d < voc_.n_dil; d++)
            {
                ResPair rp;
                const int n = i * voc_.n_rb + j;
                snprintf(nm, sizeof(nm), "_meldec.blocks.%d.convs1.%d.1.w", n, d);
                snprintf(nb, sizeof(nb), "_meldec.blocks.%d.convs1.%d.1.b", n, d);
                rp.c1 = load_conv(g, nm, nb, C);
                snprintf(nm, sizeof(nm), "_meldec.blocks.%d.convs2.%d.1.w", n, d);
                snprintf(nb, sizeof(nb), "_meldec.blocks.%d.convs2.%d.1.b", n, d);
                rp.c2 = load_conv(g, nm, nb, C);
                if (rp.c1.Cout != C || rp.c2.Cout != C) fail(ZV_ERR_SHAPE, "residual block %d: channel mismatch", n);
                if (rp.c1.K == rp.c2.K && pair_supported(rp.c1.Cout_p, rp.c1.K))
                {
                    std::vector<uint16_t> pk(pair_weight_halfs(rp.c1.Cout_p, rp.c1.K));
                    void **dst[2] = {&rp.p1, &rp.p2};
                    const char *fmt[2] = {"_meldec.blocks.%d.convs1.%d.1.w", "_meldec.blocks.%d.convs2.%d.1.w"};
                    for (int q = 0; q < 2; q++)
                    {
                        snprintf(nm, sizeof(nm), fmt[q], n, d);
                        pack_pair_weight((const uint16_t *)g.get(nm).data, rp.c1.K, C, rp.c1.Cout_p, pk.data());
                        *dst[q] = dev_alloc(pk.size() * 2 + 8192);
                        ZV_HIP(hipMemcpy(*dst[q], pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
                        {   // the same weights in 16 x 16 x 32 fragment order (resblock_pair_kernel, resblock_block32_kernel): conv1 as the A operand, conv2 as B
                            std::vector<uint16_t> xk(pair_weight16_halfs(rp.c1.Cout_p, rp.c1.K));
                            pack_pair_weight16((const uint16_t *)g.get(nm).data, rp.c1.K, C, rp.c1.Cout_p, xk.data(), q == 1);
                            void **xd = q ? &rp.x2 : &rp.x1;
                            *xd = dev_alloc(xk.size() * 2 + 8192);
                            ZV_HIP(hipMemcpy(*xd, xk.data(), xk.size() * 2, hipMemcpyHostToDevice));
                        }
                        if (rp.c1.Cout_p == 64)
                        {
                            std::vector<uint16_t> rk(pair_ring_weight_halfs(64, rp.c1.K));
                            pack_pair_weight_ring((const uint16_t *)g.get(nm).data, rp.c1.K, C, 64, rk.data(), q == 1);
                            void **rd = q ? &rp.r2 : &rp.r1;
                            *rd = dev_alloc(rk.size() * 2);
                            ZV_HIP(hipMemcpy(*rd, rk.data(), rk.size() * 2, hipMemcpyHostToDevice));
                        }
                    }
                }
                if (i == 0 && d == 0) hp.voc_resblock_kernels[j] = rp.c1.K;
                voc_.pairs.push_back(rp);
            }
    }
    if ((uint32_t)hop != hp.audio_hop_size) fail(ZV_ERR_SHAPE, "product of upsample scales %d != audio.hop_size %u", hop, hp.audio_hop_size);
    {
        const GgufTensor &w = g.get("_meldec.output_conv.1.w");
        const GgufTensor &b = g.get("_meldec.output_conv.1.b");
        if (w.type != GGML_F16 || w.ne[1] != C || w.ne[2] != 1) fail(ZV_ERR_SHAPE, "_meldec.output_conv.1.w: expected f16 [K,%d,1]", C);
        if (b.type != GGML_F32 || b.nelements() != 1) fail(ZV_ERR_SHAPE, "_meldec.output_conv.1.b: expected f32 [1]");
        if ((hp.voc_channels >> n_up) != (uint32_t)C) fail(ZV_ERR_SHAPE, "vocoder channels %u do not halve down to %d over %d stages", hp.voc_channels, C, n_up);
        voc_.out_K = (int)w.ne[0];
        voc_.out_C = C;
        const int Cp = round_up(C, 16);
        std::vector<uint16_t> h((size_t)voc_.out_K * Cp, 0);
        const uint16_t *src = (const uint16_t *)w.data;
        for (int ic = 0; ic < C; ic++)
            for (int k = 0; k < voc_.out_K; k++) h[(size_t)k * Cp + ic] = src[(size_t)ic * voc_.out_K + k];
        voc_.out_w = (uint16_t *)dev_alloc(h.size() * 2);
        ZV_HIP(hipMemcpy(voc_.out_w, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        voc_.out_b = ((const float *)b.data)[0];
    }

    // ---------------- decoder (src/stylettsdec.cpp:33-66,163-168,220-239,334-340) ----------------
    {
        dec_.M = M;
        const GgufTensor &a0 = g.get("_mel_decoder.asr_res.0.w");
        dec_.R = (int)a0.ne[2];
        const int R = dec_.R, B = 2 * Ed, CAT = B + R;
        if (R % 16) fail(ZV_ERR_SHAPE, "residual_dim = %d must be a multiple of 16", R);
        const int edims[2][2] = {{Ed, B}, {B, B}};
        for (int i = 0; i < 2; i++)
        {
            DecBlk &b = dec_.enc[i];
            b.cin = edims[i][0];
            b.cout = edims[i][1];
            b.learned_sc = b.cin != b.cout;
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.conv1.w", i);
            snprintf(nb, sizeof(nb), "_mel_decoder.encode.%d.conv1.b", i);
            b.conv1 = load_conv(g, nm, nb, b.cin, true);
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.conv2.w", i);
            snprintf(nb, sizeof(nb), "_mel_decoder.encode.%d.conv2.b", i);
            b.conv2 = load_conv(g, nm, nb, b.cin, true);
            if (b.conv1.Cout != b.cin || b.conv2.Cout != b.cout) fail(ZV_ERR_SHAPE, "_mel_decoder.encode.%d: channel mismatch", i);
            if (b.learned_sc)
            {
                snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.conv1x1.w", i);
                b.sc = load_conv(g, nm, "", b.cin, true);
            }
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.norm1.w", i); b.n1w = upload_vec(g, nm, b.cin);
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.norm1.b", i); b.n1b = upload_vec(g, nm, b.cin);
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.norm2.w", i); b.n2w = upload_vec(g, nm, b.cin);
            snprintf(nm, sizeof(nm), "_mel_decoder.encode.%d.norm2.b", i); b.n2b = upload_vec(g, nm, b.cin);
        }
        dec_.asr0 = load_conv(g, "_mel_decoder.asr_res.0.w", "_mel_decoder.asr_res.0.b", Ed);
        dec_.asr1w = upload_vec(g, "_mel_decoder.asr_res.1.w", R);
        dec_.asr1b = upload_vec(g, "_mel_decoder.asr_res.1.b", R);
        const int ddims[5][2] = {{CAT, B}, {CAT, B}, {CAT, Ed}, {Ed, Ed}, {Ed, Ed}};
        // all ten AdaIN fc layers (Linear(E -> 2C)) concatenated into one GEMV; `extra` carries the +1 of gamma
        int fc_out = 0;
        for (int i = 0; i < 5; i++) fc_out += 2 * ddims[i][0] + 2 * ddims[i][1];
        std::vector<float> W((size_t)fc_out * Ed), Bv(fc_out + 64, 0.f), Ex(fc_out + 64, 0.f);
        int o = 0;
        for (int i = 0; i < 5; i++)
        {
            DecBlk &b = dec_.dec[i];
            b.cin = ddims[i][0];
            b.cout = ddims[i][1];
            b.learned_sc = b.cin != b.cout;
            snprintf(nm, sizeof(nm), "_mel_decoder.decode.%d.conv1.w", i);
            snprintf(nb, sizeof(nb), "_mel_decoder.decode.%d.conv1.b", i);
            b.conv1 = load_conv(g, nm, nb, b.cin, true);
            snprintf(nm, sizeof(nm), "_mel_decoder.decode.%d.conv2.w", i);
            snprintf(nb, sizeof(nb), "_mel_decoder.decode.%d.conv2.b", i);
            b.conv2 = load_conv(g, nm, nb, b.cout, true);
            if (b.conv1.Cout != b.cout || b.conv2.Cout != b.cout) fail(ZV_ERR_SHAPE, "_mel_decoder.decode.%d: channel mismatch", i);
            if (b.learned_sc)
            {
                snprintf(nm, sizeof(nm), "_mel_decoder.decode.%d.conv1x1.w", i);
                b.sc = load_conv(g, nm, "", b.cin, true);
            }
            for (int k = 1; k <= 2; k++)
            {
                const int Cn = (k == 1) ? b.cin : b.cout;
                snprintf(nm, sizeof(nm), "_mel_decoder.decode.%d.norm%d.fc.w", i, k);
                snprintf(nb, sizeof(nb), "_mel_decoder.decode.%d.norm%d.fc.b", i, k);
                const GgufTensor &fw = g.get(nm), &fb = g.get(nb);
                if (fw.type != GGML_F32 || fw.ne[0] != Ed || fw.ne[1] != 2 * Cn) fail(ZV_ERR_SHAPE, "tensor %s: expected f32 [%d, %d]", nm, Ed, 2 * Cn);
                if (fb.type != GGML_F32 || fb.nelements() != 2 * Cn) fail(ZV_ERR_SHAPE, "tensor %s: expected f32 [%d]", nb, 2 * Cn);
                memcpy(W.data() + (size_t)o * Ed, fw.data, (size_t)2 * Cn * Ed * 4);
                memcpy(Bv.data() + o, fb.data, (size_t)2 * Cn * 4);
                for (int c = 0; c < Cn; c++) Ex[o + c] = 1.0f;
                (k == 1 ? b.g1 : b.g2) = o;
                o += 2 * Cn;
            }
        }
        dec_.fc_out = fc_out;
        dec_.fcW = (float *)dev_alloc(W.size() * 4);
        dec_.fcB = (float *)dev_alloc(Bv.size() * 4);
        dec_.fcExtra = (float *)dev_alloc(Ex.size() * 4);
        ZV_HIP(hipMemcpy(dec_.fcW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
        ZV_HIP(hipMemcpy(dec_.fcB, Bv.data(), Bv.size() * 4, hipMemcpyHostToDevice));
        ZV_HIP(hipMemcpy(dec_.fcExtra, Ex.data(), Ex.size() * 4, hipMemcpyHostToDevice));
        dec_.to_out = load_conv(g, "_mel_decoder.to_out.0.w", "_mel_decoder.to_out.0.b", Ed);
        if (dec_.to_out.Cout != M) fail(ZV_ERR_SHAPE, "_mel_decoder.to_out.0.w: expected %d output channels", M);
    }

    // ---------------- encoder (src/fs2encoder.cpp:29-62,152-171,256-261,344-382,504-505) ----------------
    {
        const GgufTensor &we = g.get("_pe._enc.src_word_emb.w");
        const GgufTensor &pe = g.get("_pe._enc.punct_embed.w");
        const GgufTensor &st = g.get("sinusoid_encoding_table");
        if (we.ne[0] != hp.emb_dim || pe.ne[0] != hp.punct_emb_dim || st.ne[0] != Ed) fail(ZV_ERR_SHAPE, "embedding / position tables do not match emb_dim/punct_emb_dim");
        if (we.ne[1] < 1 || pe.ne[1] < 1 || st.ne[1] < 1) fail(ZV_ERR_SHAPE, "empty embedding / position table");
        enc_.wemb = upload_f32(we);
        enc_.pemb = upload_f32(pe);
        enc_.posenc = upload_f32(st);
        enc_.posenc_rows = (int)st.ne[1];
        enc_.wemb_rows = (int)we.ne[1];          // ids are checked against what the file holds (155 / 7 rows in the
        enc_.pemb_rows = (int)pe.ne[1];          // reference's checkpoints, src/zerovox.h:35-36)
        enc_.layers.resize(hp.encoder_layer);
        for (uint32_t l = 0; l < hp.encoder_layer; l++)
        {
            EncLayer &L = enc_.layers[l];
            std::vector<float> W((size_t)3 * Ed * Ed), Bv(3 * Ed + 64, 0.f);
            const char *names[3] = {"w_qs", "w_ks", "w_vs"};
            for (int i = 0; i < 3; i++)
            {
                snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.slf_attn.%s.w", l, names[i]);
                snprintf(nb, sizeof(nb), "_pe._enc.laystk.%u.slf_attn.%s.b", l, names[i]);
                const GgufTensor &w = g.get(nm), &b = g.get(nb);
                if (w.type != GGML_F32 || w.ne[0] != Ed || w.ne[1] != Ed || b.nelements() != Ed) fail(ZV_ERR_SHAPE, "tensor %s: expected f32 [%d, %d]", nm, Ed, Ed);
                memcpy(W.data() + (size_t)i * Ed * Ed, w.data, (size_t)Ed * Ed * 4);
                memcpy(Bv.data() + (size_t)i * Ed, b.data, (size_t)Ed * 4);
            }
            L.qkvW = (float *)dev_alloc(W.size() * 4);
            L.qkvB = (float *)dev_alloc(Bv.size() * 4);
            ZV_HIP(hipMemcpy(L.qkvW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
            ZV_HIP(hipMemcpy(L.qkvB, Bv.data(), Bv.size() * 4, hipMemcpyHostToDevice));
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.slf_attn.fc.w", l);
            const GgufTensor &fw = g.get(nm);
            if (fw.type != GGML_F32 || fw.ne[0] != Ed || fw.ne[1] != Ed) fail(ZV_ERR_SHAPE, "tensor %s: expected f32 [%d, %d]", nm, Ed, Ed);
            L.fcW = upload_f32(fw);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.slf_attn.fc.b", l); L.fcB = upload_vec(g, nm, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.slf_attn.layer_norm.w", l); L.ln1w = upload_vec(g, nm, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.slf_attn.layer_norm.b", l); L.ln1b = upload_vec(g, nm, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.pos_ffn.layer_norm.w", l); L.ln2w = upload_vec(g, nm, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.pos_ffn.layer_norm.b", l); L.ln2b = upload_vec(g, nm, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.pos_ffn.w_1.w", l);
            snprintf(nb, sizeof(nb), "_pe._enc.laystk.%u.pos_ffn.w_1.b", l);
            L.w1 = load_conv(g, nm, nb, Ed);
            snprintf(nm, sizeof(nm), "_pe._enc.laystk.%u.pos_ffn.w_2.w", l);
            snprintf(nb, sizeof(nb), "_pe._enc.laystk.%u.pos_ffn.w_2.b", l);
            L.w2 = load_conv(g, nm, nb, L.w1.Cout);
            if (L.w2.Cout != Ed) fail(ZV_ERR_SHAPE, "pos_ffn.w_2 must map back to %d channels", Ed);
            if (L.w1.K != (int)hp.conv_kernel_size[0] || L.w2.K != (int)hp.conv_kernel_size[1]) fail(ZV_ERR_SHAPE, "pos_ffn kernel sizes do not match the KV keys");
        }
        auto load_vp = [&](VarPred &v, const char *prefix) {
            snprintf(nm, sizeof(nm), "%s.conv_layer.conv1d_1.conv.w", prefix);
            snprintf(nb, sizeof(nb), "%s.conv_layer.conv1d_1.conv.b", prefix);
            v.c1 = load_conv(g, nm, nb, Ed);
            v.V = v.c1.Cout;
            snprintf(nm, sizeof(nm), "%s.conv_layer.conv1d_2.conv.w", prefix);
            snprintf(nb, sizeof(nb), "%s.conv_layer.conv1d_2.conv.b", prefix);
            v.c2 = load_conv(g, nm, nb, v.V);
            if (v.c1.K != 3 || v.c2.K != 3 || v.c2.Cout != v.V) fail(ZV_ERR_SHAPE, "%s: predictor convs must be k3, %d -> %d", prefix, v.V, v.V);
            snprintf(nm, sizeof(nm), "%s.conv_layer.layer_norm_1.w", prefix); v.l1w = upload_vec(g, nm, v.V);
            snprintf(nm, sizeof(nm), "%s.conv_layer.layer_norm_1.b", prefix); v.l1b = upload_vec(g, nm, v.V);
            snprintf(nm, sizeof(nm), "%s.conv_layer.layer_norm_2.w", prefix); v.l2w = upload_vec(g, nm, v.V);
            snprintf(nm, sizeof(nm), "%s.conv_layer.layer_norm_2.b", prefix); v.l2b = upload_vec(g, nm, v.V);
            snprintf(nm, sizeof(nm), "%s.linear_layer.w", prefix); v.lw = upload_vec(g, nm, v.V);
            snprintf(nm, sizeof(nm), "%s.linear_layer.b", prefix); v.lb = upload_vec(g, nm, 1);
        };
        load_vp(enc_.dur, "_pe._var_adapt.duration_predictor");
        load_vp(enc_.pitch, "_pe._var_adapt.pitch_predictor");
        load_vp(enc_.energy, "_pe._var_adapt.engy_pred");
        const GgufTensor &pemb = g.get("_pe._var_adapt.pitch_embedding.w"), &eemb = g.get("_pe._var_adapt.energy_embedding.w");
        if (pemb.ne[0] != Ed || pemb.ne[1] != hp.encoder_ve_n_bins || eemb.ne[0] != Ed || eemb.ne[1] != hp.encoder_ve_n_bins)
            fail(ZV_ERR_SHAPE, "pitch/energy embedding: expected f32 [%d, %u]", Ed, hp.encoder_ve_n_bins);
        enc_.pitch_emb = upload_f32(pemb);
        enc_.energy_emb = upload_f32(eemb);
    }
    ZV_HIP(hipDeviceSynchronize());
}

Model::~Model()
{
    hipSetDevice(device);
    stash_lane();
    for (Lane &l : lanes_)
        if (l.stream) hipStreamSynchronize(l.stream);
    drop_graphs();
    prof_clear();
    for (void *p : allocs_) hipFree(p);
    if (lanes_.empty())
    {
        if (pinned_) hipHostFree(pinned_);
        for (hipEvent_t e : tail_events_) hipEventDestroy(e);
        if (copy_stream_) hipStreamDestroy(copy_stream_);
    }
    for (hipEvent_t &e : batch_events_)
        if (e)
        {
            hipEventDestroy(e);
            e = nullptr;
        }
    for (Lane &l : lanes_)
    {
        if (l.copy_stream) hipStreamSynchronize(l.copy_stream);
        if (l.arena.base) hipFree(l.arena.base);
        if (l.io) hipFree(l.io);
        if (l.pinned) hipHostFree(l.pinned);
        for (hipEvent_t e : l.tail_events) hipEventDestroy(e);
        if (l.copy_stream) hipStreamDestroy(l.copy_stream);
        if (l.stream) hipStreamDestroy(l.stream);
    }
}

void Model::sync() { ZV_HIP(hipStreamSynchronize(stream)); }

hipStream_t Model::copy_stream()
{
    if (!copy_stream_) ZV_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    return copy_stream_;
}

hipEvent_t Model::batch_event(uint64_t seq, int which)
{
    hipEvent_t &e = batch_events_[2 * (seq % BATCH_RING) + which];
    if (!e) ZV_HIP(hipEventCreate(&e));
    return e;
}

hipEvent_t Model::tail_event(int i)
{
    while ((int)tail_events_.size() <= i)
    {
        hipEvent_t e;
        ZV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        tail_events_.push_back(e);
    }
    return tail_events_[i];
}

uint32_t Model::vocoder_halo_frames() const
{
    double frames = (voc_.in_conv.K - 1) / 2;            // input conv, at the frame rate
    double rate = 1.0;                                   // samples per frame at the current stage
    int sumd = 0;
    for (int d = 0; d < voc_.n_dil; d++) sumd += voc_.dil[d];
    for (int i = 0; i < voc_.n_up; i++)
    {
        // polyphase transposed conv: ups[i].K taps at the INPUT rate of the stage
        frames += (double)voc_.ups[i].K / rate;
        rate *= voc_.scales[i];
        int kmax = 1;
        for (int j = 0; j < voc_.n_rb; j++) kmax = std::max(kmax, voc_.pairs[((size_t)i * voc_.n_rb + j) * voc_.n_dil].c1.K);
        frames += (double)((kmax - 1) / 2) * (sumd + voc_.n_dil) / rate;      // dilated conv + plain conv per dilation
    }
    frames += (double)((voc_.out_K - 1) / 2) / rate;
    return (uint32_t)std::ceil(frames) + 1;
}

void Model::stash_lane()
{
    if (lanes_.empty()) return;
    Lane &l = lanes_[cur_lane_];
    l.stream = stream;
    l.arena = arena_;
    l.io = io_;
    l.io_cap = io_cap_;
    l.pinned = pinned_;
    l.pinned_cap = pinned_cap_;
    l.copy_stream = copy_stream_;
    l.tail_events = tail_events_;
}

void Model::select_lane(int i)
{
    if (i < 0 || i >= 16) fail(ZV_ERR_ARG, "lane %d out of range", i);
    if (i == cur_lane_ && (size_t)i < lanes_.size()) return;
    stash_lane();
    while ((int)lanes_.size() <= i)
    {
        Lane l;
        ZV_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        lanes_.push_back(l);
    }
    cur_lane_ = i;
    stream = lanes_[i].stream;
    arena_ = lanes_[i].arena;
    io_ = lanes_[i].io;
    io_cap_ = lanes_[i].io_cap;
    pinned_ = lanes_[i].pinned;
    pinned_cap_ = lanes_[i].pinned_cap;
    copy_stream_ = lanes_[i].copy_stream;
    tail_events_ = lanes_[i].tail_events;
}

void Model::sync_all_lanes()
{
    stash_lane();
    for (Lane &l : lanes_) ZV_HIP(hipStreamSynchronize(l.stream));
}

void *Model::pinned_scratch(size_t bytes)
{
    if (bytes > pinned_cap_)
    {
        ZV_HIP(hipStreamSynchronize(stream));            // the block belongs to this lane: only its streams use it
        if (copy_stream_) ZV_HIP(hipStreamSynchronize(copy_stream_));
        if (pinned_) hipHostFree(pinned_);
        pinned_ = nullptr;
        pinned_cap_ = 0;
        if (hipHostMalloc(&pinned_, bytes, hipHostMallocDefault) != hipSuccess) fail(ZV_ERR_OOM, "hipHostMalloc(%zu) failed", bytes);
        pinned_cap_ = bytes;
    }
    return pinned_;
}

void *Model::io_scratch(size_t bytes)
{
    if (bytes > io_cap_)
    {
        ZV_HIP(hipStreamSynchronize(stream));
        if (io_) hipFree(io_);
        io_ = nullptr;
        io_cap_ = 0;
        if (hipMalloc(&io_, bytes) != hipSuccess) fail(ZV_ERR_OOM, "hipMalloc(%zu) for I/O scratch failed", bytes);
        io_cap_ = bytes;
    }
    return io_;
}

// ---------------------------------------------------------------------------------------------------
// activation arena

uint32_t Model::max_frames_per_utterance() const
{
    // buffer descriptors address a segment with 32-bit byte offsets: rows * channels * 4 < 2^31 at every stage
    // (rows * channels peaks at the first upsample stages: T * hop * C_last <= T * s0 * C0 / 2 ...)
    uint64_t worst = (uint64_t)round_up(hp.voc_channels, 16);
    uint64_t rate = 1, C = hp.voc_channels;
    for (uint32_t i = 0; i < hp.voc_num_upsamples; i++)
    {
        rate *= hp.voc_upsample_scales[i];
        C >>= 1;
        worst = std::max<uint64_t>(worst, rate * (uint64_t)round_up((int)C, 16));
    }
    worst = std::max<uint64_t>(worst, (uint64_t)(2 * E() + dec_.R));
    const uint64_t lim = ((uint64_t)1 << 31) / (4 * worst) - 64;
    return (uint32_t)std::min<uint64_t>(lim, 32768);
}

size_t Model::arena_bytes_for(size_t n_rows, size_t t_rows, int nseg) const
{
    const size_t Ed = E(), N = n_rows, T = t_rows, S = (size_t)nseg;
    // vocoder: c0 + two ping-pong pools of (up + 3 y + 3 xt) sized for the widest stages
    size_t voc = T * round_up(hp.voc_channels, 16) * 4;
    size_t pool[2] = {0, 0};
    size_t L = T;
    int C = hp.voc_channels;
    for (uint32_t i = 0; i < hp.voc_num_upsamples; i++)
    {
        L *= hp.voc_upsample_scales[i];
        C >>= 1;
        const size_t Cp = round_up(C, 16);
        const size_t need = L * Cp * (4 + 3 * 4 + 3 * 4) + 16 * 256;
        pool[i & 1] = std::max(pool[i & 1], need);
    }
    voc += pool[0] + pool[1] + 4096;
    // decoder: cat + a handful of [T][2E] buffers + per-segment vectors + three sets of statistics partials
    const size_t CAT = 2 * Ed + dec_.R;
    const size_t nblk = T / 32 + S;           // >= sum over segments of ceil(T_u / 32) ... sized per segment below
    (void)nblk;
    size_t dec = T * (CAT + 4 * 2 * Ed + 2 * dec_.R) * 4 + S * (size_t)(dec_.fc_out + 8 * CAT + 512) * 4 +
                 3 * (T / 32 + S) * CAT * 16 + T * (2 * CAT + 2 * Ed) * 2 + 65536;
    // encoder
    const size_t Fp = round_up(hp.conv_filter_size, 16);
    size_t enc = N * (Ed * 8 + 3 * Ed + Fp + 1024) * 4 + 65536;
    return std::max(voc, std::max(dec, enc)) + (1 << 20);
}

void Model::arena_require(size_t bytes)
{
    if (bytes <= arena_.cap) return;
    ZV_HIP(hipStreamSynchronize(stream));
    // captured graphs hold pointers into the arena they were captured on: whichever lane regrows its arena, every graph
    // goes (growth happens a handful of times per process, outside timed regions)
    drop_graphs();
    if (arena_.base) hipFree(arena_.base);
    arena_ = DeviceArena();
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) fail(ZV_ERR_OOM, "hipMalloc(%zu) for the activation arena failed", bytes);
    // on the lane's own stream: the streams are non-blocking, so a memset on the null stream is NOT ordered with the
    // kernels that follow on `stream` and could zero an arena they have already started to fill
    // ZV_ARENA_FILL=255 fills it with NaN patterns instead: no kernel may depend on what a fresh arena holds (test hook)
    const int fill = knob(ZV_ARENA_FILL);
    ZV_HIP(hipMemsetAsync(p, fill, bytes, stream));
    arena_.base = (char *)p;
    arena_.cap = bytes;
}

void Model::reserve(uint32_t max_phonemes, uint32_t max_frames)
{
    arena_require(arena_bytes_for(std::max(1u, max_phonemes), std::max(1u, max_frames), 1));
}

void Model::reserve_batch(const Batch &b) { arena_require(arena_bytes_for(b.n_rows, b.t_rows, b.nseg)); }

// ---------------------------------------------------------------------------------------------------
// launch helpers

void Model::prof_clear()
{
    for (auto &p : prof)
    {
        hipEventDestroy(p.e0);
        hipEventDestroy(p.e1);
    }
    prof.clear();
}

void Model::tick(const char *, double, double, hipEvent_t *e0)
{
    *e0 = nullptr;
    if (!profiling || in_group_) return;
    ZV_HIP(hipEventCreate(e0));
    ZV_HIP(hipEventRecord(*e0, stream));
}

void Model::tock(hipEvent_t e0, const char *name, double bytes, double flops)
{
    if (!profiling) return;
    if (in_group_)
    {
        group_bytes_ += bytes;
        group_flops_ += flops;
        group_n_++;
        return;
    }
    hipEvent_t e1;
    ZV_HIP(hipEventCreate(&e1));
    ZV_HIP(hipEventRecord(e1, stream));
    prof.push_back({name, e0, e1, bytes, flops, group_n_ > 0 ? group_n_ : 1});
}

// the ResBlock launches of one stage run back to back: one event pair brackets the whole run so that the per-launch
// average is not inflated by ~2 us of event-record overhead per launch
void Model::group_begin()
{
    if (!profiling) return;
    ZV_HIP(hipEventCreate(&group_e0_));
    ZV_HIP(hipEventRecord(group_e0_, stream));
    in_group_ = true;
    group_bytes_ = group_flops_ = 0.0;
    group_n_ = 0;
}

void Model::group_end(const char *name)
{
    if (!profiling || !in_group_) return;
    in_group_ = false;
    tock(group_e0_, name, group_bytes_, group_flops_);
    group_n_ = 0;
}

#define ZV_LAUNCH(name, bytes, flops, call)          \
    do                                               \
    {                                                \
        if (skip_launch_) break;                     \
        hipEvent_t _e0;                              \
        tick(name, bytes, flops, &_e0);              \
        ZV_HIP(call);                                \
        tock(_e0, name, bytes, flops);               \
    } while (0)

ConvJob Model::job(const ConvW &w) const
{
    ConvJob j;
    memset(&j, 0, sizeof(j));
    j.Cin_p = w.Cin_p;
    j.Cout_p = w.Cout_p;
    j.K = w.K;
    j.dil = 1;
    j.pad = (w.K - 1) / 2;
    j.ck = w.ck;
    j.w = w.w;
    j.w8 = w.w8;
    j.bias = w.bias;
    j.pro = PRO_ACT;
    j.slope = 1.0f;
    j.pscale = 1.0f;
    j.escale = 1.0f;
    j.ldx = w.Cin_p;
    j.ldo = w.Cout_p;
    return j;
}

void Model::dbg_inject(void *dev, int ld, int cols, size_t rows)
{
    ZV_HIP(hipMemcpy2DAsync(dev, (size_t)ld * 4, dbg_layer.x, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, stream));
}

void Model::dbg_extract(const void *dev, int ld, int cols, size_t rows)
{
    ZV_HIP(hipMemcpy2DAsync(dbg_layer.out, (size_t)cols * 4, dev, (size_t)ld * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost, stream));
    ZV_HIP(hipStreamSynchronize(stream));
    dbg_layer.done = true;
}

int Model::voc_stage_rate(int stage) const
{
    int r = 1;
    for (int i = 0; i <= stage && i < voc_.n_up; i++) r *= voc_.scales[i];
    return r;
}

int Model::voc_stage_channels(int stage) const { return voc_.in_conv.Cout >> (stage + 1); }

void Model::conv(const ConvJob *jobs, int n, const Segs &segs, int rate, const char *name, double bytes, double flops)
{
    ZV_LAUNCH(name, bytes, flops, launch_conv(stream, jobs, n, n_cu, segs, rate));
}

// algorithmic bytes / flops of one conv layer: f32 activations in + out (+ residual),
// f16 weights, f32 bias; 2*L*Cin*Cout*K flops.  L = the batch's capacity rows (exact for a single utterance and for
// batches of equal-length utterances).
static double conv_bytes(double L, int Cin, int Cout, int K, bool res)
{
    return 4.0 * L * Cin + 4.0 * L * Cout + (res ? 4.0 * L * Cout : 0.0) + 2.0 * Cin * Cout * K + 4.0 * Cout;
}
static double conv_flops(double L, int Cin, int Cout, int K) { return 2.0 * L * Cin * Cout * K; }

// ---------------------------------------------------------------------------------------------------
// HiFi-GAN vocoder (reference src/hifigan.cpp:187-377): fixed schedule of 2 + n_up * 7 launches

void Model::vocode_dev(const Batch &bt, const float *d_mel, float *d_wav)
{
    if (bt.t_rows == 0 || bt.t_max <= 0) fail(ZV_ERR_ARG, "T must be > 0");
    vocode_group(bt, d_mel, d_wav);
}

void Model::vocode_tail(const Batch &bt, const float *d_mel, float *d_wav, int g0, int cnt)
{
    if (!bt.d_frm || g0 < 0 || cnt < 1 || g0 + cnt > bt.nseg) fail(ZV_ERR_ARG, "internal: bad segment group");
    Batch sub = bt;
    sub.d_frm = bt.d_frm + g0;
    sub.nseg = cnt;
    vocode_group(sub, d_mel, d_wav, 2);
}

void Model::vocode_group(const Batch &bt, const float *d_mel, float *d_wav, int part)
{
    struct Unskip { bool &f; ~Unskip() { f = false; } } unskip{skip_launch_};
    skip_launch_ = part == 2;
    arena_require(arena_bytes_for(1, bt.t_rows, bt.nseg));
    arena_.used = 0;
    const Segs fr = bt.frames();
    const int M = hp.audio_num_mels;
    size_t L = bt.t_rows;                       // capacity rows at the current stage (buffer sizes)
    double La = std::min((double)bt.t_rows, (double)bt.t_max * bt.nseg);      // rows this call covers (accounting)
    int rate = 1;
    int C = voc_.in_conv.Cout;
    float *c0 = arena_.take_n<float>(L * voc_.in_conv.Cout_p);
    // batches: the first upsample conv runs on conv_gemm_kernel over an f16 operand tensor (see below) — the input conv writes it
    const int upg0 = knob(ZV_UP_GEMM);
    const bool c0_f16 = dbg_layer.kind < 0 && voc_.n_up > 0 && voc_.ups[0].w8 && upg0 && knob(ZV_CONV_GEMM) != 0 &&
                        (upg0 == 2 || (long)L >= 16384) && voc_.ups[0].Cin_p == voc_.in_conv.Cout_p;

    // V0: (mel - mean) / scale -> input conv k7 + bias            (src/hifigan.cpp:242-265)
    {
        ConvJob j = job(voc_.in_conv);
        j.x0 = d_mel;
        j.ldx = M;
        j.pro = PRO_MELNORM;
        j.pa = voc_.mean;
        j.pb = voc_.scale;
        j.out = c0;
        if (c0_f16)
        {   // the only reader is the first upsample conv on conv_gemm_kernel: its operand f16(lrelu(c0, 0.1)) straight from here
            j.eact = 1;
            j.oslope = 0.1f;
            j.out_f16 = 1;
        }
        conv(&j, 1, fr, rate, "voc_input_conv", conv_bytes(La, M, C, j.K, false), conv_flops(La, M, C, j.K));
        if (dbg_layer.kind == ZV_LAYER_VOC_INPUT)
        {
            dbg_extract(c0, voc_.in_conv.Cout_p, C, L);
            return;
        }
    }

    char *pool_base[2];
    size_t pool_sz[2] = {0, 0};
    {
        size_t Ls = bt.t_rows;
        int Cs = C;
        for (int i = 0; i < voc_.n_up; i++)
        {
            Ls *= voc_.scales[i];
            Cs >>= 1;
            const size_t need = Ls * round_up(Cs, 16) * (4 + 3 * 4 + 3 * 4) + 16 * 256;
            pool_sz[i & 1] = std::max(pool_sz[i & 1], need);
        }
        pool_base[0] = (char *)arena_.take(pool_sz[0]);
        pool_base[1] = (char *)arena_.take(pool_sz[1]);
    }

    const float third = (float)(1.0 / (float)voc_.n_rb);            // src/hifigan.cpp:315
    const float *prev_y[3] = {nullptr, nullptr, nullptr};
    const float *prev_merged = nullptr;          // the previous stage stored (y0 + y1) + y2 instead of the three branches
    for (int i = 0; i < voc_.n_up; i++)
    {
        const bool last_stage = i == voc_.n_up - 1;
        skip_launch_ = part == 2;                 // the head runs every upsample conv, the last stage's too (whole batch)
        const int s = voc_.scales[i];
        const ConvW &up = voc_.ups[i];
        const int Cout = C >> 1, Cp = round_up(Cout, 16);
        const size_t Lo = L * s;
        DeviceArena pool;
        pool.base = pool_base[i & 1];
        pool.cap = pool_sz[i & 1];
        float *ub = pool.take_n<float>(Lo * Cp);
        float *y[3];
        _Float16 *xt[3];
        for (int j = 0; j < 3; j++) y[j] = pool.take_n<float>(Lo * Cp);
        for (int j = 0; j < 3; j++) xt[j] = (_Float16 *)pool.take_n<float>(Lo * Cp);   // f16 xt, or f32 ping-pong partner of y (fused path)

        // V1: leaky_relu(0.1) -> transposed conv (polyphase) + bias      (src/hifigan.cpp:281-297, 22-71)
        {
            ConvJob j = job(up);
            j.slope = 0.1f;
            if (i == 0) { j.x0 = c0; j.pro = PRO_ACT; }
            else if (prev_merged) { j.x0 = prev_merged; j.pro = PRO_SCALE_ACT; j.pscale = third; }
            else { j.x0 = prev_y[0]; j.x1 = prev_y[1]; j.x2 = prev_y[2]; j.pro = PRO_SUM3_ACT; j.pscale = third; }
            const bool dbg_up = dbg_layer.kind == ZV_LAYER_VOC_UPSAMPLE && dbg_layer.index == i;
            if (dbg_up)
            {
                // the layer's input is what enters leaky_relu (src/hifigan.cpp:281): the input conv's output / the MRF mean
                float *in = i == 0 ? c0 : const_cast<float *>(prev_merged ? prev_merged : prev_y[0]);
                dbg_inject(in, up.Cin_p, C, L);
                j.x0 = in;
                j.x1 = j.x2 = nullptr;
                if (i > 0) { j.pro = PRO_SCALE_ACT; j.pscale = 1.0f; }
            }
            j.out = ub;
            // batches, wide upsample convs: the prologue as a pass of its own (f16 operand tensor, parked in the stage's last xt
            // buffer — free until the residual blocks run), the conv on conv_gemm_kernel (ZV_UP_GEMM = 0 never, 2 at any length)
            const int upg = knob(ZV_UP_GEMM);
            if (i == 0 && c0_f16)
            {
                j.x0 = c0;
                j.pro = PRO_RAW_F16;
            }
            else if (up.w8 && upg && knob(ZV_CONV_GEMM) != 0 && (upg == 2 || (long)L >= 16384) && (size_t)up.Cin_p * 2 * L <= Lo * Cp * 4)
            {
                ZV_LAUNCH("voc_upsample", 0.0, 0.0, launch_act_f16(stream, (const float *)j.x0, (const float *)j.x1, (const float *)j.x2,
                                                                   j.pro == PRO_ACT ? 1.0f : j.pscale, j.slope, xt[2], (size_t)L * up.Cin_p));
                j.x0 = xt[2];
                j.x1 = j.x2 = nullptr;
                j.pro = PRO_RAW_F16;
                j.pscale = 1.0f;
            }
            // algorithmic: true polyphase MAC count L_in*Cin*Cout*k
            conv(&j, 1, fr, rate, "voc_upsample", 4.0 * La * C * (i == 0 ? 1 : 3) + 4.0 * La * s * Cout + 2.0 * C * Cout * 2 * s,
                 2.0 * La * C * Cout * 2 * s);
        }
        if (dbg_layer.kind == ZV_LAYER_VOC_UPSAMPLE && dbg_layer.index == i)
        {
            dbg_extract(ub, Cp, Cout, Lo);
            return;
        }
        skip_launch_ = (part == 1 && last_stage) || (part == 2 && !last_stage);
        L = Lo;
        La *= s;
        rate *= s;
        C = Cout;
        const bool dbg_here = dbg_layer.kind == 0 && dbg_layer.index / voc_.n_rb == i;
        if (dbg_here) dbg_inject(ub, Cp, Cout, L);
        const long Lbatch = (long)bt.t_max * rate * bt.nseg;       // rows the launches of this stage cover

        // V2: the 3 MRF branches run side by side (one job each).  Fused path: one launch per dilation
        // (conv -> lrelu -> conv -> + residual, xt kept in LDS), y ping-pongs between two buffers because a
        // workgroup's halo rows belong to its neighbours' output tiles.
        const ResPair &rp0 = voc_.pairs[((size_t)i * voc_.n_rb) * voc_.n_dil];
        // 256-channel stage: the fused kernel needs all 256 xt channels in one workgroup, which leaves few workgroups per
        // branch for a short utterance — two unfused launches (480 workgroups at 512 frames) win below about a round
        // of fused ones (round 4, on the 16 x 16 x 32 kernel, whole vocoder under graph replay: 128 frames 0.276 unfused /
        // 0.291 fused ms, 256: 0.320 / 0.333, 512: 0.470 / 0.465, 1 024: 0.852 / 0.814)
        const bool enough_rows = Cp != 256 || force_fuse256_ || (Lbatch / 54) * 3 >= (long)n_cu;
        const bool fused = !no_fuse_ && rp0.p1 != nullptr && enough_rows;
        const float *ycur[3] = {ub, ub, ub};
        const float *merged_sum = nullptr;
        group_begin();
        // narrow stages: the whole residual block (all dilations) of the three branches in ONE launch, y tile kept
        // in registers between the dilation pairs (launch_triple)
        bool whole_block = fused && !no_triple_ && voc_.n_dil <= TRIPLE_MAX_DIL;
        for (int jb = 0; jb < 3 && whole_block; jb++)
        {
            const ResPair &r0 = voc_.pairs[((size_t)i * voc_.n_rb + jb) * voc_.n_dil];
            whole_block = triple_supported(Cp, r0.c1.K, voc_.dil, voc_.n_dil);
            for (int d = 0; d < voc_.n_dil && whole_block; d++)
                whole_block = voc_.pairs[((size_t)i * voc_.n_rb + jb) * voc_.n_dil + d].p1 != nullptr;
        }
        if (whole_block)
        {
            TripleJob tj[3];
            double bb = 0, ff = 0;
            for (int jb = 0; jb < 3; jb++)
            {
                TripleJob &t = tj[jb];
                memset(&t, 0, sizeof(t));
                t.y = ub;
                t.out = y[jb];
                t.n_dil = voc_.n_dil;
                t.Cp = Cp;
                t.slope = 0.1f;
                for (int d = 0; d < voc_.n_dil; d++)
                {
                    const ResPair &rp = voc_.pairs[((size_t)i * voc_.n_rb + jb) * voc_.n_dil + d];
                    t.K = rp.c1.K;
                    t.w1[d] = rp.p1;
                    t.w2[d] = rp.p2;
                    t.w1x[d] = rp.x1;
                    t.w2x[d] = rp.x2;
                    t.b1[d] = rp.c1.bias;
                    t.b2[d] = rp.c2.bias;
                    t.dil[d] = voc_.dil[d];
                    bb += conv_bytes(La, C, C, rp.c1.K, false) + conv_bytes(La, C, C, rp.c2.K, true);
                    ff += conv_flops(La, C, C, rp.c1.K) + conv_flops(La, C, C, rp.c2.K);
                }
                ycur[jb] = y[jb];
            }
            ZV_LAUNCH("voc_resblock_conv", bb, ff, launch_triple(stream, tj, 3, n_cu, fr, rate));
        }
        // 64 channels, batches: the first two dilation pairs of the branches with few taps in ONE launch (resblock_block64_kernel:
        // the branch's tensor crosses HBM once instead of twice; ZV_BLOCK64 = most taps it takes, 0 = never; negative: at any length)
        bool b64[3] = {false, false, false};
        {
            const int k64 = knob(ZV_BLOCK64);
            const int kmax64 = k64 < 0 ? -k64 : k64;
            if (fused && !whole_block && Cp == 64 && voc_.n_dil == 3 && kmax64 >= 3 && (k64 < 0 || Lbatch / 244 >= 4L * n_cu))
            {
                TripleJob tj[3];
                int nj = 0;
                double bb = 0, ff = 0;
                for (int jb = 0; jb < 3; jb++)
                {
                    const ResPair *rp = &voc_.pairs[((size_t)i * voc_.n_rb + jb) * voc_.n_dil];
                    if (rp[0].c1.K > kmax64 || !rp[0].r1 || !rp[0].r2 || !rp[1].r1 || !rp[1].r2 || !block64_supported(Cp, rp[0].c1.K, voc_.dil, 2)) continue;
                    TripleJob &t = tj[nj++];
                    memset(&t, 0, sizeof(t));
                    t.y = ub;
                    t.out = (float *)xt[jb];
                    t.n_dil = 2;
                    t.Cp = Cp;
                    t.K = rp[0].c1.K;
                    t.slope = 0.1f;
                    for (int d = 0; d < t.n_dil; d++)
                    {
                        t.w1[d] = rp[d].r1;
                        t.w2[d] = rp[d].r2;
                        t.b1[d] = rp[d].c1.bias;
                        t.b2[d] = rp[d].c2.bias;
                        t.dil[d] = voc_.dil[d];
                        bb += conv_bytes(La, C, C, rp[d].c1.K, false) + conv_bytes(La, C, C, rp[d].c2.K, true);
                        ff += conv_flops(La, C, C, rp[d].c1.K) + conv_flops(La, C, C, rp[d].c2.K);
                    }
                    b64[jb] = true;
                    ycur[jb] = (float *)xt[jb];
                }
                if (nj) ZV_LAUNCH("voc_resblock_conv", bb, ff, launch_block64(stream, tj, nj, fr, rate));
            }
        }
        for (int d = 0; d < voc_.n_dil && !whole_block; d++)
        {
            ConvJob j1[3], j2[3];
            PairJob pj[3];
            double b1 = 0, f1 = 0, b2 = 0, f2 = 0;
            int npj = 0;                     // pair jobs of this dilation (the branches resblock_block64_kernel has not covered)
            for (int jb = 0; jb < 3; jb++)
            {
                if (b64[jb] && d < 2) continue;
                const ResPair &rp = voc_.pairs[((size_t)i * voc_.n_rb + jb) * voc_.n_dil + d];
                const float *yin = ycur[jb];
                float *yout = fused ? ((d & 1) ? (float *)xt[jb] : y[jb]) : y[jb];
                if (fused && !rp.p1) fail(ZV_ERR_SHAPE, "residual block %d: branches of one stage must all be fusable", i * voc_.n_rb + jb);
                // xt = lrelu(conv(lrelu(y), k, dil) + b)  kept as the f16 operand of the next conv (:108-150)
                ConvJob a = job(rp.c1);
                a.x0 = yin;
                a.pro = PRO_ACT;
                a.slope = 0.1f;
                a.dil = voc_.dil[d];
                a.pad = (rp.c1.K - 1) / 2 * voc_.dil[d];
                a.eact = 1;
                a.oslope = 0.1f;
                a.out_f16 = 1;
                a.out = xt[jb];
                j1[jb] = a;
                // y = y + (conv(xt, k, 1) + b)                                                    (:169-181)
                ConvJob b = job(rp.c2);
                b.x0 = xt[jb];
                b.pro = PRO_RAW_F16;
                b.res = yin;
                b.ldres = Cp;
                b.out = y[jb];
                j2[jb] = b;
                PairJob &p = pj[npj++];
                memset(&p, 0, sizeof(p));
                p.y = yin;
                p.out = yout;
                p.w1 = rp.x1;
                p.w2 = rp.x2;
                p.w1r = rp.r1;
                p.w2r = rp.r2;
                p.b1 = rp.c1.bias;
                p.b2 = rp.c2.bias;
                p.Cp = Cp;
                p.K = rp.c1.K;
                p.dil = voc_.dil[d];
                p.slope = 0.1f;
                ycur[jb] = fused ? yout : y[jb];
                b1 += conv_bytes(La, C, C, rp.c1.K, false);
                f1 += conv_flops(La, C, C, rp.c1.K);
                b2 += conv_bytes(La, C, C, rp.c2.K, true);
                f2 += conv_flops(La, C, C, rp.c2.K);
            }
            // the last pair of the stage: the three branches' outputs are only ever used summed (MRF, :300-315), so the
            // workgroups run all three branches of a tile and store the sum alone
            // ... once the merged launch (a third of the workgroups, each three times as long) still has rounds of workgroups to
            // spare: at one round (a single 512-frame utterance) the merged 128- / 64-channel launches took 45.7 / 37.3 us against
            // 28.4 / 32.8 us for the three branches side by side, more than the upsample conv gains from reading one tensor
            const int merge_tile = Cp >= 256 ? 54 : (Cp == 128 ? 118 : 246);
            const bool merge_pays = knob(ZV_MERGE_ALWAYS) != 0 || (Lbatch / merge_tile >= 4L * n_cu && Cp <= knob(ZV_MERGE_MAXC));
            const bool merge = fused && !no_merge_ && !dbg_here && d == voc_.n_dil - 1 && merge_pays;
            if (merge)
            {
                bool ms_free = true;
                for (int q = 0; q < npj; q++) ms_free = ms_free && pj[0].out != pj[q].y;
                float *ms = ms_free ? pj[0].out : nullptr;
                if (!ms) fail(ZV_ERR_DEVICE, "internal: no free buffer for the merged MRF sum");
                if (Cp >= 256 && knob(ZV_MERGE_SEQ) != 0)
                {
                    // 256 channels: the branches one launch each on the side-by-side kernel (96-row tiles, all staging loads in flight:
                    // 1 020 us for the three against 1 105 us for the three-branches-per-workgroup form; at 128 channels the single-
                    // branch launches' tails cost more than they gain: 1 422 against 1 386 us), every launch adding its term into the
                    // running sum — (y0 + y1) + y2, the merged form's association, hence its bits
                    for (int jb = 0; jb < 3; jb++)
                    {
                        PairJob q = pj[jb];
                        q.sum_out = ms;
                        q.sum_in = jb ? ms : nullptr;
                        ZV_LAUNCH("voc_resblock_conv", (b1 + b2) / 3, (f1 + f2) / 3, launch_pair(stream, &q, 1, n_cu, fr, rate));
                    }
                }
                else
                    ZV_LAUNCH("voc_resblock_conv", b1 + b2, f1 + f2,
                              launch_pair(stream, pj, npj, n_cu, fr, rate, ms));
                merged_sum = ms;
            }
            else if (fused)
            {
                if (npj) ZV_LAUNCH("voc_resblock_conv", b1 + b2, f1 + f2, launch_pair(stream, pj, npj, n_cu, fr, rate));
            }
            else
            {
                conv(j1, 3, fr, rate, "voc_resblock_conv", b1, f1);
                conv(j2, 3, fr, rate, "voc_resblock_conv", b2, f2);
            }
        }
        // one profile entry per stage (bench.py prices every stage against its own binding roof)
        static const char *const rb_names[8] = {"voc_resblock_s0", "voc_resblock_s1", "voc_resblock_s2", "voc_resblock_s3",
                                                "voc_resblock_s4", "voc_resblock_s5", "voc_resblock_s6", "voc_resblock_s7"};
        group_end(rb_names[i < 8 ? i : 7]);
        if (dbg_here)
        {
            dbg_extract(ycur[dbg_layer.index % voc_.n_rb], Cp, Cout, L);
            return;
        }
        for (int jb = 0; jb < 3; jb++) y[jb] = const_cast<float *>(ycur[jb]);
        for (int jb = 0; jb < 3; jb++) prev_y[jb] = y[jb];
        prev_merged = merged_sum;
    }

    // V3: (sum of branches)/3 -> leaky_relu(0.01) -> conv k7 (C -> 1) + b -> tanh          (:315-345)
    skip_launch_ = part == 1;
    {
        OutConvArgs a;
        a.x0 = prev_merged ? prev_merged : prev_y[0];
        a.x1 = prev_merged ? nullptr : prev_y[1];
        a.x2 = prev_merged ? nullptr : prev_y[2];
        a.ldx = round_up(C, 16);
        a.L = 0;
        a.C = C;
        a.K = voc_.out_K;
        a.pscale = third;
        a.slope = (float)1e-2;
        a.w = voc_.out_w;
        a.bias = voc_.out_b;
        a.out = d_wav;
        a.segs = fr;
        a.rate = rate;
        if (dbg_layer.kind == ZV_LAYER_VOC_OUTPUT)
        {
            // the layer's input is the MRF mean that enters leaky_relu(0.01) (src/hifigan.cpp:315-324)
            float *in = const_cast<float *>(a.x0);
            dbg_inject(in, a.ldx, C, L);
            a.x1 = a.x2 = nullptr;
            a.pscale = 1.0f;
        }
        ZV_LAUNCH("voc_output_conv", 12.0 * La * C + 4.0 * La, 2.0 * La * C * a.K, launch_out_conv(stream, a));
        if (dbg_layer.kind == ZV_LAYER_VOC_OUTPUT) dbg_extract(d_wav, 1, 1, L);
    }
}

void Model::drop_graphs()
{
    sync_all_lanes();                            // an exec of another lane may still be running
    for (auto &g : graphs_)
        if (g.exec) hipGraphExecDestroy(g.exec);
    graphs_.clear();
}

// Replays the captured schedule for (kind, capacities, buffers) or captures it first.  The capacities decide grids and
// arena layout; the segment tables are read by the kernels at run time, so a batch graph does not depend on the
// utterances' lengths.
template <typename F> void Model::run_captured(int kind, const Batch &b, const void *const *key, int nkey, F &&enqueue)
{
    const void *kp[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < nkey && i < 8; i++) kp[i] = key[i];
    for (auto &g : graphs_)
        if (g.kind == kind && g.epoch == knob_epoch() && g.b.nseg == b.nseg && g.b.n_max == b.n_max && g.b.t_max == b.t_max && g.b.n_rows == b.n_rows &&
            g.b.t_rows == b.t_rows && g.b.d_tok == b.d_tok && g.b.d_frm == b.d_frm &&
            (b.d_tok || memcmp(&g.b.tok1, &b.tok1, sizeof(Seg)) == 0) && (b.d_frm || memcmp(&g.b.frm1, &b.frm1, sizeof(Seg)) == 0) &&
            memcmp(g.p, kp, sizeof(kp)) == 0)
        {
            ZV_HIP(hipGraphLaunch(g.exec, stream));
            return;
        }
    reserve_batch(b);                            // hipMalloc is not capturable
    hipGraph_t graph = nullptr;
    ZV_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    try
    {
        enqueue();
    }
    catch (...)
    {
        hipStreamEndCapture(stream, &graph);
        if (graph) hipGraphDestroy(graph);
        throw;
    }
    ZV_HIP(hipStreamEndCapture(stream, &graph));
    CapturedGraph cg;
    cg.kind = kind;
    cg.epoch = knob_epoch();       // a graph replays the kernel regime it was captured in: a later zv_debug_set captures anew
    cg.b = b;
    memcpy(cg.p, kp, sizeof(kp));
    hipError_t e = hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) fail(ZV_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    if (graphs_.size() >= 16)
    {
        // the oldest graphs may still be executing: drain the stream before their execs go away
        ZV_HIP(hipStreamSynchronize(stream));
        drop_graphs();
    }
    graphs_.push_back(cg);
    ZV_HIP(hipGraphLaunch(cg.exec, stream));
}

void Model::chain_dev(const Batch &b, const int32_t *d_ids, const int32_t *d_puncts, const float *d_styles, float *d_hidden,
                      float *d_mel, float *d_wav, int32_t *d_nframes, const void *h2d_src, void *h2d_dst, size_t h2d_bytes, int voc_part)
{
    auto run = [&]() {
        if (h2d_bytes) ZV_HIP(hipMemcpyAsync(h2d_dst, h2d_src, h2d_bytes, hipMemcpyHostToDevice, stream));
        encode_dev(b, d_ids, d_puncts, d_styles, d_hidden, d_nframes);
        decode_dev(b, d_hidden, d_styles, d_mel);        // the reference vocodes all T frames (src/zerovox.cpp:326-334)
        if (voc_part == 1) vocode_group(b, d_mel, d_wav, 1);      // the caller runs the last stage in utterance groups
        else vocode_dev(b, d_mel, d_wav);
    };
    if (!graph_mode || profiling)
    {
        run();
        return;
    }
    const void *key[8] = {d_ids, d_puncts, d_styles, d_hidden, d_mel, d_wav, d_nframes, h2d_src};
    run_captured(voc_part == 1 ? 2 : 1, b, key, 8, run);
}

void Model::vocode_dev_graph(const Batch &b, const float *d_mel, float *d_wav)
{
    if (!graph_mode || profiling)
    {
        vocode_dev(b, d_mel, d_wav);
        return;
    }
    const void *key[2] = {d_mel, d_wav};
    run_captured(0, b, key, 2, [&]() { vocode_dev(b, d_mel, d_wav); });
}

// ---------------------------------------------------------------------------------------------------
// StyleTTS mel decoder (reference src/stylettsdec.cpp:306-470)

void Model::decode_dev(const Batch &bt, const float *d_hidden, const float *d_styles, float *d_mel)
{
    if (bt.t_rows == 0 || bt.t_max <= 0) fail(ZV_ERR_ARG, "T must be > 0");
    arena_require(arena_bytes_for(1, bt.t_rows, bt.nseg));
    arena_.used = 0;
    const Segs fr = bt.frames();
    const int Ed = (int)E(), B = 2 * Ed, R = dec_.R, CAT = B + R, S = bt.nseg;
    const size_t L = bt.t_rows;
    const int nblk = (bt.t_max + 31) / 32;                         // statistics blocks per segment
    const int hs = round_up(dec_.fc_out + 64, 64);                 // AdaIN vectors per segment
    const int ss = 2 * CAT + 64;                                   // (mean, rstd) pairs per segment
    float *h = arena_.take_n<float>((size_t)S * hs);
    float *st_x = arena_.take_n<float>((size_t)S * ss), *st_t = arena_.take_n<float>((size_t)S * ss);
    float *st_y = arena_.take_n<float>((size_t)S * ss), *st_a = arena_.take_n<float>((size_t)S * ss);
    double *part_t = arena_.take_n<double>((size_t)S * nblk * CAT * 2), *part_o = arena_.take_n<double>((size_t)S * nblk * CAT * 2);
    float *cat = arena_.take_n<float>(L * CAT);
    float *t1 = arena_.take_n<float>(L * B);
    float *sc = arena_.take_n<float>(L * B);
    float *x0 = arena_.take_n<float>(L * B);
    float *xa = arena_.take_n<float>(L * B);
    float *asr_t = arena_.take_n<float>(L * R);
    _Float16 *xa16 = arena_.take_n<_Float16>(L * CAT), *t16 = arena_.take_n<_Float16>(L * B), *xr16 = arena_.take_n<_Float16>(L * CAT);
    const double Ld = (double)L;
    // Two ways to feed a conv its normalised operand, same bits (tests): (a) the conv normalises while it stages its
    // input tile (PRO_NORM_ACT) — no extra launch, right for a very short utterance where every launch is latency; (b) one
    // pass writes the f16 operand (launch_norm_act_f16) and the conv copies it (PRO_RAW_F16) — right when launches have
    // many rounds of workgroups: a 1 056-wide conv stages every input tile 9 times (once per group of 128 output
    // channels), so (a) repeats the f32 prologue 9 times and reads twice the bytes.
    const int pre_env = knob(ZV_DEC_PREPASS);      // test / A-B hook
    // (round 4: with the single-utterance conv form's loader waves the pass pays from 256 frames on — it takes the statistics launch's
    // place and leaves the loaders a plain copy: one utterance of 128 / 256 / 512 / 1 024 frames 1.32 / 1.345 / 1.62 / 2.16 ms fused,
    // 1.33 / 1.33 / 1.58 / 2.07 ms with the pass)
    const bool prepass = pre_env >= 0 ? pre_env != 0 : (size_t)bt.t_max * bt.nseg >= 256;

    // D2: all ten AdaIN fc layers at once for every utterance's style vector            (src/stylettsdec.cpp:175-189)
    ZV_LAUNCH("dec_adain_fc", 4.0 * dec_.fc_out * (Ed + 2), 2.0 * S * dec_.fc_out * Ed,
              launch_linear(stream, d_styles, Ed, Ed, dec_.fcW, dec_.fcB, dec_.fc_out, h, hs, dec_.fcExtra, segs_single(S)));

    const float rsqrt2 = (float)(1.0 / sqrt(2.0));                       // src/stylettsdec.cpp:146,301

    auto finalize = [&](const double *part, int C, float *stat, int c_off) {
        ZV_LAUNCH("dec_in_stats", 16.0 * S * nblk * C, 4.0 * S * nblk * C,
                  launch_stats_finalize(stream, part, nblk, C, 1e-5f, stat, ss, c_off, fr, 1));
    };
    // make `j` read lrelu(norm(x)) with x's statistics still in `part` (channels [0, Cpart)); stores them in `stat`
    auto norm_input = [&](ConvJob &j, const float *x, int ldx, int C, const double *part, int Cpart, float *stat, const float *g,
                          const float *b, int gb_seg, _Float16 *op16, _Float16 *raw16 = nullptr) {
        if (prepass)
        {
            ZV_LAUNCH("dec_norm_operand", 6.0 * Ld * C, 8.0 * Ld * C,
                      launch_norm_act_f16(stream, x, ldx, C, part, nblk, Cpart, 1e-5f, stat, ss, g, b, gb_seg, 0.2f, op16, C, fr, raw16));
            j.x0 = op16;
            j.ldx = C;
            j.pro = PRO_RAW_F16;
        }
        else
        {
            finalize(part, Cpart, stat, 0);
            j.x0 = x;
            j.ldx = ldx;
            j.pro = PRO_NORM_ACT;
            j.pstat = stat;
            j.pstat_seg = ss;
            j.pa = g;
            j.pb = b;
            j.pab_seg = gb_seg;
            j.slope = 0.2f;
        }
    };

    // InstanceNorm statistics of the stage input (it comes from the encoder or the host, not from a conv of ours)
    ZV_LAUNCH("dec_in_stats", 4.0 * Ld * Ed, 3.0 * Ld * Ed, launch_stats_partial(stream, d_hidden, Ed, Ed, part_o, nblk, fr, 1));

    // one residual block: IN/AdaIN -> lrelu -> conv1 -> IN/AdaIN -> lrelu -> conv2 -> (+ shortcut) / sqrt2.
    // The partial sums of x's statistics are in part_o (channels [0, Cpart) of x; the others are final in st_in already);
    // the block leaves the partial sums of its output in part_o again when want_stats.  gb_seg: per-segment stride of the
    // affine vectors (0 for the encode blocks' shared InstanceNorm weights, hs for the decode blocks' AdaIN vectors).
    int blk_no = 0;              // 0,1: encode blocks; 2..6: decode blocks (dbg_layer.index)
    auto block = [&](const DecBlk &b, const float *x, int ldx, int Cpart, float *st_in, const float *g1, const float *b1,
                     const float *g2, const float *b2, int gb_seg, float *out, int ldo, bool want_stats) {
        const bool dbg_here = dbg_layer.kind == 2 && dbg_layer.index == blk_no && !dbg_layer.done;
        blk_no++;
        if (dbg_layer.done) return;
        if (dbg_here)
        {   // the layer's input comes from the host; its statistics are recomputed for every channel
            dbg_inject(const_cast<float *>(x), ldx, b.cin, L);
            ZV_HIP(launch_stats_partial(stream, x, ldx, b.cin, part_o, nblk, fr, 1));
            Cpart = b.cin;
        }
        const float *res = x;
        int ldres = ldx;
        ConvJob jj[2];
        int nj = 0;
        {
            ConvJob j = job(b.conv1);
            // (a learned shortcut reads f16(x): with the pre-pass on, that operand is written by the same pass)
            norm_input(j, x, ldx, b.cin, part_o, Cpart, st_in, g1, b1, gb_seg, xa16, b.learned_sc ? xr16 : nullptr);
            j.out = t1;
            j.stat_part = part_t;
            j.stat_nblk = nblk;
            j.stat_C = b.conv1.Cout;
            jj[nj++] = j;
        }
        double bytes = conv_bytes(Ld, b.cin, b.conv1.Cout, 3, false), flops = conv_flops(Ld, b.cin, b.conv1.Cout, 3);
        if (b.learned_sc)
        {
            ConvJob j = job(b.sc);
            j.x0 = x;
            j.ldx = ldx;
            if (prepass)
            {
                j.x0 = xr16;
                j.ldx = b.cin;
                j.pro = PRO_RAW_F16;
            }
            j.out = sc;
            res = sc;
            ldres = b.sc.Cout_p;
            const double sb = conv_bytes(Ld, b.cin, b.cout, 1, false), sf = conv_flops(Ld, b.cin, b.cout, 1);
            if (b.sc.Cout_p == b.conv1.Cout_p)
            {   // same output width as conv1: second job of the same launch
                jj[nj++] = j;
                bytes += sb;
                flops += sf;
            }
            else
                conv(&j, 1, fr, 1, "dec_conv", sb, sf);
        }
        conv(jj, nj, fr, 1, "dec_conv", bytes, flops);
        const int Cm = b.conv1.Cout;
        {
            ConvJob j = job(b.conv2);
            norm_input(j, t1, b.conv1.Cout_p, Cm, part_t, Cm, st_t, g2, b2, gb_seg, t16);
            j.res = res;
            j.ldres = ldres;
            j.escale = rsqrt2;
            j.out = out;
            j.ldo = ldo;
            if (want_stats)
            {
                j.stat_part = part_o;
                j.stat_nblk = nblk;
                j.stat_C = b.cout;
            }
            conv(&j, 1, fr, 1, "dec_conv", conv_bytes(Ld, Cm, b.cout, 3, true), conv_flops(Ld, Cm, b.cout, 3));
        }
        if (dbg_here) dbg_extract(out, ldo, b.cout, L);
    };

    // AdaIN1d alone (sub-block tap, ZV_LAYER_DEC_ADAIN; index = 2 * decode block + (norm - 1); reference src/stylettsdec.cpp:171-200):
    // the production fc GEMM above, the production statistics (partial sums + finalise) and the prologue's arithmetic
    // ((x - mean) * rstd) * gamma + beta written out by norm_apply_kernel — no activation, no conv
    if (dbg_layer.kind == ZV_LAYER_DEC_ADAIN)
    {
        const int bi = dbg_layer.index / 2, k = dbg_layer.index & 1;
        if (bi < 0 || bi >= 5) return;
        const DecBlk &b = dec_.dec[bi];
        const int Cn = k ? b.cout : b.cin, go = k ? b.g2 : b.g1;
        float *xin = cat;                                  // [L][Cn] with leading dimension Cn: any buffer of L * CAT floats
        dbg_inject(xin, Cn, Cn, L);
        ZV_HIP(launch_stats_partial(stream, xin, Cn, Cn, part_o, nblk, fr, 1));
        ZV_HIP(launch_stats_finalize(stream, part_o, nblk, Cn, 1e-5f, st_x, ss, 0, fr, 1));
        ZV_HIP(launch_norm_apply(stream, xin, Cn, Cn, st_x, ss, h + go, h + go + Cn, t1, Cn, nullptr, nblk, fr));
        dbg_extract(t1, Cn, Cn, L);
        return;
    }

    // encode0 / encode1: ResBlk1d with affine InstanceNorm                         (src/stylettsdec.cpp:69-149,373-374)
    block(dec_.enc[0], d_hidden, Ed, Ed, st_x, dec_.enc[0].n1w, dec_.enc[0].n1b, dec_.enc[0].n2w, dec_.enc[0].n2b, 0, x0, B, true);
    block(dec_.enc[1], x0, B, B, st_y, dec_.enc[1].n1w, dec_.enc[1].n1b, dec_.enc[1].n2w, dec_.enc[1].n2b, 0, cat, CAT, true);

    // asr_res = IN_affine(conv1x1(enc_seq) + b) written straight into the concat buffer      (:382-404)
    if (dbg_layer.done) return;
    {
        ConvJob j = job(dec_.asr0);
        j.x0 = d_hidden;
        j.ldx = Ed;
        j.out = asr_t;
        j.stat_part = part_t;
        j.stat_nblk = nblk;
        j.stat_C = R;
        conv(&j, 1, fr, 1, "dec_conv", conv_bytes(Ld, Ed, R, 1, false), conv_flops(Ld, Ed, R, 1));
        finalize(part_t, R, st_a, 0);
        ZV_LAUNCH("dec_norm_apply", 8.0 * Ld * R, 3.0 * Ld * R,
                  launch_norm_apply(stream, asr_t, R, R, st_a, ss, dec_.asr1w, dec_.asr1b, cat + B, CAT, part_t, nblk, fr));
        finalize(part_t, R, st_x, B);            // statistics of the concat's asr columns: final for decode0..2
        if (dbg_layer.kind == ZV_LAYER_DEC_ASR_RES)
        {
            dbg_extract(cat + B, CAT, R, L);
            return;
        }
    }

    // decode0..4: AdainResBlk1d; blocks 0..2 read cat([x, asr]) and 0,1 write x back into it   (:406-428).  The x
    // columns' statistics arrive as partial sums from the producing conv2, the asr columns' are already in st_x.
    const float *cur = cat;
    int ldc = CAT;
    float *outs[5] = {cat, cat, xa, x0, xa};
    const int ldos[5] = {CAT, CAT, Ed, Ed, Ed};
    const int cparts[5] = {B, B, B, Ed, Ed};
    float *sts[5] = {st_x, st_x, st_x, st_y, st_y};
    for (int i = 0; i < 5; i++)
    {
        const DecBlk &b = dec_.dec[i];
        block(b, cur, ldc, cparts[i], sts[i], h + b.g1, h + b.g1 + b.cin, h + b.g2, h + b.g2 + b.cout, hs, outs[i], ldos[i], i < 4);
        cur = outs[i];
        ldc = ldos[i];
    }
    // to_out: conv1x1 E -> num_mels + b, emitted frame-major                                       (:432-441)
    if (dbg_layer.done) return;
    {
        ConvJob j = job(dec_.to_out);
        j.x0 = cur;
        j.ldx = ldc;
        j.out = d_mel;
        j.ldo = dec_.M;
        if (dbg_layer.kind == ZV_LAYER_DEC_TO_OUT) dbg_inject(const_cast<float *>(cur), ldc, Ed, L);
        conv(&j, 1, fr, 1, "dec_conv", conv_bytes(Ld, Ed, dec_.M, 1, false), conv_flops(Ld, Ed, dec_.M, 1));
        if (dbg_layer.kind == ZV_LAYER_DEC_TO_OUT) dbg_extract(d_mel, dec_.M, dec_.M, L);
    }
}

// ---------------------------------------------------------------------------------------------------
// FastSpeech2 encoder + variance adaptor + length regulator (reference src/fs2encoder.cpp:289-336,477-656)

Model::EncoderTaps Model::encode_dev(const Batch &bt, const int32_t *d_ids, const int32_t *d_puncts, const float *d_styles,
                                     float *d_hidden, int32_t *d_nframes)
{
    if (bt.n_rows == 0 || bt.t_rows == 0 || bt.n_max <= 0 || bt.t_max <= 0) fail(ZV_ERR_ARG, "N and T must be > 0");
    // the real extents decide (the kernels walk each segment's own rows); n_max is a capacity rounded up for grid sizing
    const int n_longest = bt.n_real > 0 ? bt.n_real : bt.n_max;
    if (n_longest > enc_.posenc_rows) fail(ZV_ERR_ARG, "%d phonemes exceed the %d rows of the sinusoid table", n_longest, enc_.posenc_rows);
    arena_require(arena_bytes_for(bt.n_rows, bt.t_rows, bt.nseg));
    arena_.used = 0;
    const Segs tk = bt.tokens(), fr = bt.frames();
    const Segs tkm = knob(ZV_LINEAR_MERGED) != 0 ? bt.tokens_merged() : tk;      // the per-token layers (linear, 1-tap conv, plain LayerNorm) see one dense segment
    const int Ed = (int)E(), H = hp.encoder_head, dk = Ed / H;
    const size_t n = bt.n_rows;
    const double nd = (double)n;
    const int Fp = round_up(hp.conv_filter_size, 16);
    float *x = arena_.take_n<float>(n * Ed), *y = arena_.take_n<float>(n * Ed);
    float *qkv = arena_.take_n<float>(n * 3 * Ed), *o = arena_.take_n<float>(n * Ed);
    float *f = arena_.take_n<float>(n * Ed);
    _Float16 *hh = arena_.take_n<_Float16>(n * Fp);
    const int Vp = round_up(enc_.dur.V, 16);
    float *va = arena_.take_n<float>(n * Vp), *vb = arena_.take_n<float>(n * Vp);
    // LayerNorm launches that carry tail work (kernels.h: launch_layernorm_tail): the style add, the predictors' linear layer, the
    // bucket + embedding step — 6 launches fewer per call, same operations in the same order (ZV_LN_TAIL = 0: separate launches)
    const bool tails = dbg_layer.kind < 0 && knob(ZV_LN_TAIL) != 0 && layernorm_tail_ok(Ed) && layernorm_tail_ok(enc_.dur.V);
    EncoderTaps t;
    t.features = x;
    t.logdur = arena_.take_n<float>(n);
    t.pitch = arena_.take_n<float>(n);
    t.energy = arena_.take_n<float>(n);
    t.pitch_bucket = arena_.take_n<int32_t>(n);
    t.energy_bucket = arena_.take_n<int32_t>(n);
    t.cum = arena_.take_n<int32_t>(n);

    ZV_LAUNCH("enc_embed", 8.0 * nd * Ed, 1.0 * nd * Ed,
              launch_embed(stream, d_ids, d_puncts, enc_.wemb, hp.emb_dim, enc_.pemb, hp.punct_emb_dim, enc_.posenc, x, Ed, tk));
    if (dbg_layer.kind == ZV_LAYER_ENC_EMBED)
    {
        dbg_extract(x, Ed, Ed, n);
        return t;
    }
    const float temperature = (float)pow((double)dk, 0.5);               // src/fs2encoder.cpp:66
    const float inv_t = (float)(1.0 / temperature);                      // :107
    int layer_no = 0;
    for (const EncLayer &Ly : enc_.layers)
    {
        const bool dbg_here = dbg_layer.kind == 1 && dbg_layer.index == layer_no;
        // sub-block taps (the reference's tensor_dbg taps any node, src/utils.cpp:19-44): the attention sublayer alone
        // (ZV_LAYER_ENC_MHA: x -> y) and the conv feed-forward sublayer alone (ZV_LAYER_ENC_FFN: y -> x)
        const bool dbg_mha = dbg_layer.kind == ZV_LAYER_ENC_MHA && dbg_layer.index == layer_no;
        const bool dbg_ffn = dbg_layer.kind == ZV_LAYER_ENC_FFN && dbg_layer.index == layer_no;
        layer_no++;
        if (dbg_here || dbg_mha) dbg_inject(x, Ed, Ed, n);
        ZV_LAUNCH("enc_linear", 4.0 * (3.0 * Ed * Ed + 4.0 * nd * Ed), 6.0 * nd * Ed * Ed,
                  launch_linear(stream, x, Ed, Ed, Ly.qkvW, Ly.qkvB, 3 * Ed, qkv, 3 * Ed, nullptr, tkm));
        ZV_LAUNCH("enc_attention", 16.0 * nd * Ed, 4.0 * nd * bt.n_max * Ed,
                  launch_attention(stream, qkv, qkv + Ed, qkv + 2 * Ed, 3 * Ed, H, dk, inv_t, o, Ed, tk));
        ZV_LAUNCH("enc_linear", 4.0 * (1.0 * Ed * Ed + 2.0 * nd * Ed), 2.0 * nd * Ed * Ed,
                  launch_linear(stream, o, Ed, Ed, Ly.fcW, Ly.fcB, Ed, f, Ed, nullptr, tkm));
        ZV_LAUNCH("enc_layernorm", 12.0 * nd * Ed, 8.0 * nd * Ed,
                  launch_add_layernorm(stream, f, Ed, x, Ed, Ed, Ed, Ly.ln1w, Ly.ln1b, 1e-5f, y, Ed, tkm));
        if (dbg_mha)
        {
            dbg_extract(y, Ed, Ed, n);
            return t;
        }
        if (dbg_ffn) dbg_inject(y, Ed, Ed, n);
        {   // FFN: conv k9 + b -> relu (kept as f16 operand) -> conv k1 + b            (src/fs2encoder.cpp:190-214)
            ConvJob a = job(Ly.w1);
            a.x0 = y;
            a.eact = 1;
            a.oslope = 0.f;
            a.out_f16 = 1;
            a.out = hh;
            conv(&a, 1, tk, 1, "enc_conv", conv_bytes(nd, Ed, Ly.w1.Cout, Ly.w1.K, false), conv_flops(nd, Ed, Ly.w1.Cout, Ly.w1.K));
            ConvJob b = job(Ly.w2);
            b.x0 = hh;
            b.pro = PRO_RAW_F16;
            b.out = f;
            // (a 1-tap conv is per token: like the linear layers it takes the batch as one dense segment)
            conv(&b, 1, Ly.w2.K == 1 ? tkm : tk, 1, "enc_conv", conv_bytes(nd, Ly.w1.Cout, Ed, Ly.w2.K, false), conv_flops(nd, Ly.w1.Cout, Ed, Ly.w2.K));
        }
        // (the last layer's LayerNorm also adds the style vector: features = encoder output + style_embed, :550-552)
        if (tails && layer_no == (int)enc_.layers.size())
            ZV_LAUNCH("enc_layernorm", 12.0 * nd * Ed, 9.0 * nd * Ed,
                      launch_layernorm_tail(stream, f, Ed, y, Ed, Ed, Ed, Ly.ln2w, Ly.ln2b, 1e-5f, x, Ed, tk, d_styles, Ed, nullptr, nullptr,
                                            nullptr, nullptr, 0, 0, nullptr, 0, nullptr));
        else
            ZV_LAUNCH("enc_layernorm", 12.0 * nd * Ed, 8.0 * nd * Ed,
                      launch_add_layernorm(stream, f, Ed, y, Ed, Ed, Ed, Ly.ln2w, Ly.ln2b, 1e-5f, x, Ed, tkm));
        if (dbg_here || dbg_ffn)
        {
            dbg_extract(x, Ed, Ed, n);
            return t;
        }
    }
    // features = encoder output + style_embed                                             (:550-552)
    if (!(tails && !enc_.layers.empty()))
        ZV_LAUNCH("enc_add_style", 8.0 * nd * Ed, 1.0 * nd * Ed, launch_add_rowvec(stream, x, Ed, Ed, d_styles, Ed, tk));

    int pred_no = 0;
    // VariancePredictor::graph (:386-440): conv + relu, LayerNorm, conv + relu, LayerNorm, linear.  `emb` (pitch / energy): the
    // prediction's bucket and x += embedding[bucket] (:442-474, 565-569) follow.  With `tails` the second LayerNorm's launch also
    // does the linear layer and the bucket / embedding step (5 + 1 launches -> 4).
    auto predictor = [&](const VarPred &v, float *out, const float *emb, int32_t *bucket) {
        const bool dbg_here = dbg_layer.kind == 3 && dbg_layer.index == pred_no && !dbg_layer.done;
        pred_no++;
        if (dbg_layer.done) return;
        if (dbg_here) dbg_inject(x, Ed, Ed, n);
        ConvJob a = job(v.c1);
        a.x0 = x;
        a.eact = 1;
        a.oslope = 0.f;
        a.out = va;
        conv(&a, 1, tk, 1, "enc_conv", conv_bytes(nd, Ed, v.V, 3, false), conv_flops(nd, Ed, v.V, 3));
        ZV_LAUNCH("enc_layernorm", 8.0 * nd * v.V, 8.0 * nd * v.V,
                  launch_add_layernorm(stream, va, Vp, nullptr, 0, v.V, Vp, v.l1w, v.l1b, 1e-5f, vb, Vp, tkm));
        ConvJob b = job(v.c2);
        b.x0 = vb;
        b.pad = 1;                                              // literal 1 in the reference (:417)
        b.eact = 1;
        b.oslope = 0.f;
        b.out = va;
        conv(&b, 1, tk, 1, "enc_conv", conv_bytes(nd, v.V, v.V, 3, false), conv_flops(nd, v.V, v.V, 3));
        if (tails)
        {
            ZV_LAUNCH("enc_layernorm", 8.0 * nd * v.V + (emb ? 12.0 * nd * Ed : 0.0), 10.0 * nd * v.V,
                      launch_layernorm_tail(stream, va, Vp, nullptr, 0, v.V, Vp, v.l2w, v.l2b, 1e-5f, vb, Vp, tk, nullptr, 0, v.lw, v.lb, out,
                                            emb, (int)hp.encoder_ve_n_bins, Ed, x, Ed, bucket));
            return;
        }
        ZV_LAUNCH("enc_layernorm", 8.0 * nd * v.V, 8.0 * nd * v.V,
                  launch_add_layernorm(stream, va, Vp, nullptr, 0, v.V, Vp, v.l2w, v.l2b, 1e-5f, vb, Vp, tkm));
        ZV_LAUNCH("enc_rowdot", 4.0 * nd * v.V, 2.0 * nd * v.V, launch_rowdot(stream, vb, Vp, v.V, v.lw, v.lb, out, tk));
        if (dbg_here) dbg_extract(out, 1, 1, n);
        if (emb && !dbg_layer.done)
            ZV_LAUNCH("enc_bucket_embed", 12.0 * nd * Ed, 1.0 * nd * Ed,
                      launch_bucket_embed_add(stream, out, hp.encoder_ve_n_bins, emb, Ed, x, Ed, bucket, tk));
    };
    predictor(enc_.dur, t.logdur, nullptr, nullptr);
    predictor(enc_.pitch, t.pitch, enc_.pitch_emb, t.pitch_bucket);
    if (dbg_layer.done) return t;
    predictor(enc_.energy, t.energy, enc_.energy_emb, t.energy_bucket);      // sees the pitch-augmented features (:569-572)
    if (dbg_layer.done) return t;
    ZV_LAUNCH("enc_length_regulator", 4.0 * (nd + (double)bt.t_rows) * Ed, 0.0,
              launch_length_regulator(stream, x, Ed, t.logdur, Ed, d_hidden, Ed, t.cum, d_nframes, tk, fr));
    return t;
}

}  // namespace zv

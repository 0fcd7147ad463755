// capi.cpp — the extern "C" boundary declared in include/zerovox_amd.h.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "model.h"
#include "knobs.h"

using zv::Model;

struct PendingBatch;
struct zv_model
{
    Model        *m;
    PendingBatch *pending;       // [ZV_BATCH_LANES], see zv_synthesize_batch_begin
};

static void free_pending(zv_model *m);
static void lane0_select(zv_model *m);   // selects lane 0 without the busy check (copies, waits, profiling)
static void use_lane0(zv_model *m);      // selects lane 0 (the lane of every synchronous entry point); fails when lane 0 has a batch in flight

static thread_local std::string g_last_error;

template <typename F> static zv_status guarded(F &&f)
{
    try
    {
        f();
        return ZV_OK;
    }
    catch (const zv::Error &e)
    {
        g_last_error = e.what();
        return e.status;
    }
    catch (const std::bad_alloc &)
    {
        g_last_error = "out of host memory";
        return ZV_ERR_OOM;
    }
    catch (const std::exception &e)
    {
        g_last_error = e.what();
        return ZV_ERR_DEVICE;
    }
}

#define ZV_NEED(cond, what) \
    if (!(cond)) zv::fail(ZV_ERR_ARG, "%s: %s", __func__, what)

extern "C" {

const char *zv_last_error(void) { return g_last_error.c_str(); }
const char *zv_version(void) { return "zerovox.cpp_amd 0.1 (gfx950)"; }

zv_status zv_model_load(const char *gguf_path, int device, zv_model **out)
{
    return guarded([&] {
        ZV_NEED(gguf_path && out, "null argument");
        *out = nullptr;
        Model *m = new Model(gguf_path, device);
        *out = new zv_model{m, nullptr};
    });
}

void zv_model_free(zv_model *m)
{
    if (!m) return;
    free_pending(m);
    delete m->m;
    delete m;
}

zv_status zv_model_get_hparams(const zv_model *m, zv_hparams *out)
{
    return guarded([&] {
        ZV_NEED(m && out, "null argument");
        *out = m->m->hp;
    });
}

zv_status zv_model_reserve(zv_model *m, uint32_t max_phonemes, uint32_t max_frames)
{
    return guarded([&] {
        ZV_NEED(m, "null model");
        ZV_HIP(hipSetDevice(m->m->device));
        m->m->reserve(max_phonemes, max_frames);
    });
}

// ---- host-buffer entry points -------------------------------------------------------------------

static void check_ids(const Model &M, const int32_t *ids, const int32_t *puncts, uint32_t n)
{
    // the reference aborts inside ggml_get_rows on a bad id (ggml-cpu.c:8456); the tables have 155 / 7 rows in the
    // reference's checkpoints (src/zerovox.h:35-36) — the limit is what the loaded file holds
    const int wmax = M.wemb_rows() - 1, pmax = M.pemb_rows() - 1;
    for (uint32_t i = 0; i < n; i++)
    {
        if (ids[i] < 0 || ids[i] > wmax) zv::fail(ZV_ERR_ARG, "phoneme id %d at position %u is outside [0, %d]", ids[i], i, wmax);
        if (puncts[i] < 0 || puncts[i] > pmax) zv::fail(ZV_ERR_ARG, "punctuation id %d at position %u is outside [0, %d]", puncts[i], i, pmax);
    }
}

#include "demo_utterance.inc"

void zv_demo_utterance(const int32_t **ids, const int32_t **puncts, const float **style, uint32_t *n_phonemes, uint32_t *style_len)
{
    if (ids) *ids = kDemoIds;
    if (puncts) *puncts = kDemoPuncts;
    if (style) *style = kDemoStyle;
    if (n_phonemes) *n_phonemes = 120;
    if (style_len) *style_len = 528;
}

uint32_t zv_max_frames(const zv_model *m) { return m ? m->m->max_frames_per_utterance() : 0; }

static void check_T(const Model &M, uint32_t T)
{
    if (T == 0) zv::fail(ZV_ERR_ARG, "T must be > 0");
    if (T > M.max_frames_per_utterance())
        zv::fail(ZV_ERR_ARG, "T = %u exceeds zv_max_frames() = %u frames per utterance (32-bit offsets inside a segment); use zv_vocode_stream for longer audio",
                 T, M.max_frames_per_utterance());
}

zv_status zv_encode_taps(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style, uint32_t n,
                         uint32_t num_phonemes, uint32_t T, float *hidden, uint32_t *n_frames, float *features, float *logdur,
                         float *pitch, float *energy, int32_t *pitch_bucket, int32_t *energy_bucket)
{
    return guarded([&] {
        ZV_NEED(m && ids && puncts && style && hidden, "null argument");
        ZV_NEED(n > 0, "n must be > 0");
        ZV_NEED(num_phonemes <= n, "num_phonemes exceeds n");
        Model &M = *m->m;
        use_lane0(m);
        check_T(M, T);
        check_ids(M, ids, puncts, n);
        const size_t E = M.E();
        const size_t b_ids = (size_t)n * 4, b_sty = E * 4, b_hid = (size_t)T * E * 4;
        char *io = (char *)M.io_scratch(256 + 2 * b_ids + b_sty + b_hid + 1024);
        int32_t *d_nf = (int32_t *)io;
        io += 256;
        int32_t *d_ids = (int32_t *)io, *d_pun = (int32_t *)(io + b_ids);
        float *d_sty = (float *)(io + 2 * b_ids + 256 - (2 * b_ids) % 256);
        float *d_hid = (float *)((char *)d_sty + ((b_sty + 255) & ~(size_t)255));
        ZV_HIP(hipMemcpyAsync(d_ids, ids, b_ids, hipMemcpyHostToDevice, M.stream));
        ZV_HIP(hipMemcpyAsync(d_pun, puncts, b_ids, hipMemcpyHostToDevice, M.stream));
        ZV_HIP(hipMemcpyAsync(d_sty, style, b_sty, hipMemcpyHostToDevice, M.stream));
        const zv::Batch bt = zv::Batch::single(n, T, num_phonemes);
        Model::EncoderTaps t = M.encode_dev(bt, d_ids, d_pun, d_sty, d_hid, d_nf);
        ZV_HIP(hipMemcpyAsync(hidden, d_hid, b_hid, hipMemcpyDeviceToHost, M.stream));
        int32_t nf = 0;
        ZV_HIP(hipMemcpyAsync(&nf, d_nf, 4, hipMemcpyDeviceToHost, M.stream));
        if (features) ZV_HIP(hipMemcpyAsync(features, t.features, (size_t)n * E * 4, hipMemcpyDeviceToHost, M.stream));
        if (logdur) ZV_HIP(hipMemcpyAsync(logdur, t.logdur, b_ids, hipMemcpyDeviceToHost, M.stream));
        if (pitch) ZV_HIP(hipMemcpyAsync(pitch, t.pitch, b_ids, hipMemcpyDeviceToHost, M.stream));
        if (energy) ZV_HIP(hipMemcpyAsync(energy, t.energy, b_ids, hipMemcpyDeviceToHost, M.stream));
        if (pitch_bucket) ZV_HIP(hipMemcpyAsync(pitch_bucket, t.pitch_bucket, b_ids, hipMemcpyDeviceToHost, M.stream));
        if (energy_bucket) ZV_HIP(hipMemcpyAsync(energy_bucket, t.energy_bucket, b_ids, hipMemcpyDeviceToHost, M.stream));
        M.sync();
        if (n_frames) *n_frames = (uint32_t)nf;
    });
}

zv_status zv_encode(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style, uint32_t n, uint32_t T,
                    float *hidden, uint32_t *n_frames)
{
    return zv_encode_taps(m, ids, puncts, style, n, n, T, hidden, n_frames, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

zv_status zv_decode(zv_model *m, const float *hidden, const float *style, uint32_t T, float *mel)
{
    return guarded([&] {
        ZV_NEED(m && hidden && style && mel, "null argument");
        Model &M = *m->m;
        use_lane0(m);
        check_T(M, T);
        const size_t E = M.E(), Mm = M.hp.audio_num_mels;
        const size_t b_hid = (size_t)T * E * 4, b_sty = (E * 4 + 255) & ~(size_t)255, b_mel = (size_t)T * Mm * 4;
        char *io = (char *)M.io_scratch(b_hid + b_sty + b_mel + 1024);
        float *d_sty = (float *)io, *d_hid = (float *)(io + b_sty), *d_mel = (float *)(io + b_sty + ((b_hid + 255) & ~(size_t)255));
        ZV_HIP(hipMemcpyAsync(d_hid, hidden, b_hid, hipMemcpyHostToDevice, M.stream));
        ZV_HIP(hipMemcpyAsync(d_sty, style, E * 4, hipMemcpyHostToDevice, M.stream));
        M.decode_dev(zv::Batch::single(1, T, 1), d_hid, d_sty, d_mel);
        ZV_HIP(hipMemcpyAsync(mel, d_mel, b_mel, hipMemcpyDeviceToHost, M.stream));
        M.sync();
    });
}

zv_status zv_vocode(zv_model *m, const float *mel, uint32_t T, float *wav)
{
    return guarded([&] {
        ZV_NEED(m && mel && wav, "null argument");
        Model &M = *m->m;
        use_lane0(m);
        check_T(M, T);
        const size_t b_mel = ((size_t)T * M.hp.audio_num_mels * 4 + 255) & ~(size_t)255, b_wav = (size_t)T * M.hp.audio_hop_size * 4;
        char *io = (char *)M.io_scratch(b_mel + b_wav);
        float *d_mel = (float *)io, *d_wav = (float *)(io + b_mel);
        ZV_HIP(hipMemcpyAsync(d_mel, mel, (size_t)T * M.hp.audio_num_mels * 4, hipMemcpyHostToDevice, M.stream));
        M.vocode_dev_graph(zv::Batch::single(1, T, 1), d_mel, d_wav);
        ZV_HIP(hipMemcpyAsync(wav, d_wav, b_wav, hipMemcpyDeviceToHost, M.stream));
        M.sync();
    });
}

uint32_t zv_vocoder_halo_frames(zv_model *m) { return m ? m->m->vocoder_halo_frames() : 0; }

zv_status zv_vocode_stream(zv_model *m, const float *mel, uint32_t T, uint32_t chunk_frames, zv_wav_sink sink, void *user)
{
    return guarded([&] {
        ZV_NEED(m && mel && sink, "null argument");
        ZV_NEED(T > 0 && chunk_frames > 0, "T and chunk_frames must be > 0");
        Model &M = *m->m;
        use_lane0(m);
        const size_t Mm = M.hp.audio_num_mels, hop = M.hp.audio_hop_size;
        const uint32_t H = M.vocoder_halo_frames();
        const uint32_t ctx_max = std::min<uint64_t>(T, (uint64_t)chunk_frames + 2 * H);
        check_T(M, ctx_max);                          // only a chunk plus its context is ever vocoded at once
        const size_t b_mel = ((size_t)T * Mm * 4 + 255) & ~(size_t)255, b_wav = (size_t)ctx_max * hop * 4;
        M.reserve(1, ctx_max);
        char *io = (char *)M.io_scratch(b_mel + b_wav);
        float *d_mel = (float *)io, *d_wav = (float *)(io + b_mel);
        // two pinned slots: chunk c is copied out and delivered while chunk c + 1 is being computed
        const size_t slot = ((size_t)chunk_frames * hop * 4 + 255) & ~(size_t)255;
        char *pin = (char *)M.pinned_scratch(2 * slot);
        hipEvent_t done[2] = {nullptr, nullptr};
        ZV_HIP(hipEventCreateWithFlags(&done[0], hipEventDisableTiming));
        ZV_HIP(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
        struct Pending { bool live; uint64_t first, n; } pend[2] = {{false, 0, 0}, {false, 0, 0}};
        auto deliver = [&](int k) {
            if (!pend[k].live) return;
            ZV_HIP(hipEventSynchronize(done[k]));
            sink(user, (const float *)(pin + k * slot), pend[k].first, pend[k].n);
            pend[k].live = false;
        };
        try
        {
            ZV_HIP(hipMemcpyAsync(d_mel, mel, (size_t)T * Mm * 4, hipMemcpyHostToDevice, M.stream));
            int k = 0;
            for (uint32_t a = 0; a < T; a += chunk_frames, k ^= 1)
            {
                const uint32_t b = std::min<uint64_t>(T, (uint64_t)a + chunk_frames);
                const uint32_t c0 = a > H ? a - H : 0, c1 = std::min<uint64_t>(T, (uint64_t)b + H);
                deliver(k);                                   // the slot we are about to overwrite
                M.vocode_dev(zv::Batch::single(1, c1 - c0, 1), d_mel + (size_t)c0 * Mm, d_wav);
                ZV_HIP(hipMemcpyAsync(pin + k * slot, d_wav + (size_t)(a - c0) * hop, (size_t)(b - a) * hop * 4, hipMemcpyDeviceToHost, M.stream));
                ZV_HIP(hipEventRecord(done[k], M.stream));
                pend[k] = {true, (uint64_t)a * hop, (uint64_t)(b - a) * hop};
                deliver(k ^ 1);                               // the previous chunk, while this one runs
            }
            deliver(k);
            deliver(k ^ 1);
        }
        catch (...)
        {
            hipStreamSynchronize(M.stream);
            hipEventDestroy(done[0]);
            hipEventDestroy(done[1]);
            throw;
        }
        M.sync();
        hipEventDestroy(done[0]);
        hipEventDestroy(done[1]);
    });
}

// one utterance end to end on the currently selected lane; no host synchronisation
static void synthesize_enqueue(Model &M, const int32_t *ids, const int32_t *puncts, const float *style, uint32_t n, uint32_t T,
                               float *wav, int32_t *nf_host)
{
    check_ids(M, ids, puncts, n);
    check_T(M, T);
    const size_t E = M.E(), Mm = M.hp.audio_num_mels, hop = M.hp.audio_hop_size;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_ids = al((size_t)n * 4), b_sty = al(E * 4), b_hid = al((size_t)T * E * 4), b_mel = al((size_t)T * Mm * 4),
                 b_wav = al((size_t)T * hop * 4);
    M.reserve(n, T);
    char *io = (char *)M.io_scratch(256 + 2 * b_ids + b_sty + b_hid + b_mel + b_wav);
    int32_t *d_nf = (int32_t *)io;
    io += 256;
    int32_t *d_ids = (int32_t *)io, *d_pun = (int32_t *)(io + b_ids);
    float *d_sty = (float *)(io + 2 * b_ids), *d_hid = (float *)(io + 2 * b_ids + b_sty);
    float *d_mel = (float *)((char *)d_hid + b_hid), *d_wav = (float *)((char *)d_mel + b_mel);
    ZV_HIP(hipMemcpyAsync(d_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, M.stream));
    ZV_HIP(hipMemcpyAsync(d_pun, puncts, (size_t)n * 4, hipMemcpyHostToDevice, M.stream));
    ZV_HIP(hipMemcpyAsync(d_sty, style, E * 4, hipMemcpyHostToDevice, M.stream));
    M.chain_dev(zv::Batch::single(n, T, n), d_ids, d_pun, d_sty, d_hid, d_mel, d_wav, d_nf);
    ZV_HIP(hipMemcpyAsync(nf_host, d_nf, 4, hipMemcpyDeviceToHost, M.stream));
    ZV_HIP(hipMemcpyAsync(wav, d_wav, (size_t)T * hop * 4, hipMemcpyDeviceToHost, M.stream));
}

zv_status zv_synthesize(zv_model *m, const int32_t *ids, const int32_t *puncts, const float *style, uint32_t n, uint32_t T,
                        float *wav, uint32_t *n_frames)
{
    return guarded([&] {
        ZV_NEED(m && ids && puncts && style && wav, "null argument");
        ZV_NEED(n > 0 && T > 0, "n and T must be > 0");
        Model &M = *m->m;
        use_lane0(m);
        int32_t nf = 0;
        synthesize_enqueue(M, ids, puncts, style, n, T, wav, &nf);
        M.sync();
        if (n_frames) *n_frames = (uint32_t)nf;
    });
}

// copies the finished waveforms out of the pinned staging block, on a few threads when there is enough to move
// (utterances [a, b) of a launch group; off[], wav[] and T[] are indexed from the group's first utterance)
static void scatter_out(const char *pin_wav, const size_t *off, float *const *wav, const uint32_t *T, size_t hop, uint32_t a, uint32_t b)
{
    size_t total = 0;
    for (uint32_t u = a; u < b; u++) total += (size_t)T[u] * hop * 4;
    // (a group of the 8-group tail is 5 MB: with the old 8 MB threshold every group went out on one thread, 4 ms per batch between
    // a batch's end and the next batch's start on that lane)
    const unsigned nth = total > ((size_t)1 << 20) ? 4u : 1u;
    auto work = [&](unsigned k) {
        for (uint32_t u = a + k; u < b; u += nth) memcpy(wav[u], pin_wav + off[u], (size_t)T[u] * hop * 4);
    };
    if (nth == 1)
    {
        work(0);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned k = 1; k < nth; k++) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
}

// ---- batches.  One launch group (up to 64 utterances / 64 Ki frames) is enqueued on a lane — input block built in the
// lane's pinned memory, the chain as one hipGraph, the last vocoder stage in utterance groups with each group's download
// behind it on the lane's copy stream — and finished later: wait for the downloads, copy the waveforms out.  The
// synchronous entry point is enqueue + finish on lane 0; zv_synthesize_batch_begin / _end expose the two halves so that a
// caller can keep a batch in flight per lane (the next batch's upload and kernels run while this one's tail downloads).
struct PendingBatch
{
    bool                  active = false;
    uint32_t              n_utt = 0;
    int                   G = 1;
    std::vector<uint32_t> gb;          // group boundaries (utterance indices), G + 1 entries
    std::vector<size_t>   woff;        // byte offset of each utterance's waveform in the pinned block
    std::vector<uint32_t> T;
    std::vector<float *>  wav;
    uint32_t             *n_frames = nullptr;
    const char           *h_wav = nullptr;
    const int32_t        *h_nf = nullptr;
    size_t                hop = 0;
};
static PendingBatch &pending_slot(zv_model *m, int lane)
{
    if (!m->pending) m->pending = new PendingBatch[ZV_BATCH_LANES];
    return m->pending[lane];
}
static void free_pending(zv_model *m)
{
    delete[] m->pending;
    m->pending = nullptr;
}
static void lane0_select(zv_model *m)
{
    if (hipSetDevice(m->m->device) != hipSuccess) zv::fail(ZV_ERR_DEVICE, "hipSetDevice(%d) failed", m->m->device);
    m->m->select_lane(0);
}
static void use_lane0(zv_model *m)
{
    // the synchronous and device-resident entry points run on lane 0's stream, arena and I/O block whatever lane was
    // touched last: select it FIRST, then refuse the call while a batch of that lane is still reading those buffers
    if (m->pending && m->pending[0].active)
        zv::fail(ZV_ERR_ARG, "lane 0 has a batch in flight (zv_synthesize_batch_begin): finish it with zv_synthesize_batch_end first");
    if (hipSetDevice(m->m->device) != hipSuccess) zv::fail(ZV_ERR_DEVICE, "hipSetDevice(%d) failed", m->m->device);
    m->m->select_lane(0);
}

static void batch_enqueue(zv_model *m, int lane, uint32_t n_utt, const int32_t *const *ids, const int32_t *const *puncts,
                          const float *const *styles, const uint32_t *n_phonemes, const uint32_t *T, float *const *wav,
                          uint32_t *n_frames)
{
    Model &M = *m->m;
    PendingBatch &pb = pending_slot(m, lane);
    if (pb.active) zv::fail(ZV_ERR_ARG, "lane %d already has a batch in flight", lane);
    M.select_lane(lane);
    const size_t E = M.E(), Mm = M.hp.audio_num_mels, hop = M.hp.audio_hop_size;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    uint32_t nmax = 0, tmax = 0;
    for (uint32_t u = 0; u < n_utt; u++)
    {
        nmax = std::max(nmax, n_phonemes[u]);
        tmax = std::max(tmax, T[u]);
    }
    zv::Batch bt;
    bt.nseg = (int)n_utt;
    bt.n_max = zv::round_up((int)nmax, 32);
    bt.n_real = (int)nmax;
    bt.t_max = zv::round_up((int)tmax, 64);
    bt.n_rows = (size_t)bt.nseg * bt.n_max;
    bt.t_rows = (size_t)bt.nseg * bt.t_max;
    // device block: [frame counts][inputs: token table | frame table | ids | puncts | styles][hidden][mel][wav]
    // (tables: one entry per utterance + one that spans all of them, see Batch::tokens_merged)
    const size_t b_tab = al((size_t)(bt.nseg + 1) * sizeof(zv::Seg)), b_ids = al(bt.n_rows * 4), b_sty = al((size_t)bt.nseg * E * 4);
    const size_t b_in = 2 * b_tab + 2 * b_ids + b_sty;
    const size_t b_nf = al((size_t)bt.nseg * 4), b_hid = al(bt.t_rows * E * 4), b_mel = al(bt.t_rows * Mm * 4),
                 b_wav = al(bt.t_rows * hop * 4);
    M.reserve_batch(bt);
    char *io = (char *)M.io_scratch(b_nf + b_in + b_hid + b_mel + b_wav);
    int32_t *d_nf = (int32_t *)io;
    char *d_in = io + b_nf;
    zv::Seg *d_tok = (zv::Seg *)d_in, *d_frm = (zv::Seg *)(d_in + b_tab);
    int32_t *d_ids = (int32_t *)(d_in + 2 * b_tab), *d_pun = (int32_t *)(d_in + 2 * b_tab + b_ids);
    float *d_sty = (float *)(d_in + 2 * b_tab + 2 * b_ids);
    float *d_hid = (float *)(d_in + b_in), *d_mel = (float *)((char *)d_hid + b_hid), *d_wav = (float *)((char *)d_mel + b_mel);
    bt.d_tok = d_tok;
    bt.d_frm = d_frm;
    // pinned mirror of the input block + landing area of the results
    size_t wav_bytes = 0;
    pb.woff.assign(n_utt, 0);
    for (uint32_t u = 0; u < n_utt; u++)
    {
        pb.woff[u] = wav_bytes;
        wav_bytes += (size_t)T[u] * hop * 4;
    }
    char *pin = (char *)M.pinned_scratch(b_in + b_nf + al(wav_bytes));
    {
        zv::Seg *h_tok = (zv::Seg *)pin, *h_frm = (zv::Seg *)(pin + b_tab);
        int32_t *h_ids = (int32_t *)(pin + 2 * b_tab), *h_pun = (int32_t *)(pin + 2 * b_tab + b_ids);
        float *h_sty = (float *)(pin + 2 * b_tab + 2 * b_ids);
        int32_t n0 = 0, t0 = 0;
        for (uint32_t u = 0; u < n_utt; u++)
        {
            const int32_t n = (int32_t)n_phonemes[u], t = (int32_t)T[u];
            h_tok[u] = zv::Seg{n0, n, n, 0};
            h_frm[u] = zv::Seg{t0, t, 0, 0};
            memcpy(h_ids + n0, ids[u], (size_t)n * 4);
            memcpy(h_pun + n0, puncts[u], (size_t)n * 4);
            memcpy(h_sty + (size_t)u * E, styles[u], E * 4);
            n0 += n;
            t0 += t;
        }
        h_tok[n_utt] = zv::Seg{0, n0, n0, 0};
        h_frm[n_utt] = zv::Seg{0, t0, 0, 0};
    }
    int32_t *h_nf = (int32_t *)(pin + b_in);
    char *h_wav = pin + b_in + b_nf;
    // Large batches: the last vocoder stage (two thirds of a waveform's bytes are produced there) runs in G groups of
    // utterances; a finished group's waveforms travel to the host on the lane's copy stream while the next group's kernels
    // run.  Same kernels on the same rows: same bits.
    // (as many groups as the switch asks for, of at least two utterances each: a group of one utterance leaves the whole-block
    // kernel two rounds of workgroups — measured 21.4 / 21.1 / 21.1 / 22.7 ms per batch with 4 / 8 / 16 / 32 groups of 32 utterances)
    const int G = (M.tail_groups() > 1 && bt.nseg >= 4 && wav_bytes >= ((size_t)16 << 20) && !M.profiling && M.dbg_layer.kind < 0)
                      ? std::min(M.tail_groups(), (int)bt.nseg / 2) : 1;
    // from here on work is queued that reads the lane's pinned input block and writes its I/O block: if anything fails the
    // lane's streams are drained before the error leaves, so an idle-looking lane never has work in flight
    try
    {
        hipStream_t cs = M.copy_stream();
        pb.gb.assign(G + 1, n_utt);
        pb.gb[0] = 0;
        // Batches in flight on different lanes share the GPU kernel by kernel (worth 1.4 ms per batch: the latency-bound encoder /
        // decoder launches of one fill the other's vocoder).  Every batch owns a (start, done) pair of timing events on its lane's
        // stream — zv_batch_timeline reports from them when the GPU had no batch to work on.  (Round 4 measured that number
        // without a profiler: 0.00 ms per step with two batches in flight; a device-side limit of two concurrent batches with a
        // third queued behind them, built to close gaps a kernel trace had shown, changed nothing and was removed again.)
        const uint64_t seq = M.next_batch_seq();
        ZV_HIP(hipEventRecord(M.batch_event(seq, 0), M.stream));
        if (G <= 1)
        {
            M.chain_dev(bt, d_ids, d_pun, d_sty, d_hid, d_mel, d_wav, d_nf, pin, d_in, b_in);
            ZV_HIP(hipEventRecord(M.batch_event(seq, 1), M.stream));
            ZV_HIP(hipMemcpyAsync(h_nf, d_nf, (size_t)bt.nseg * 4, hipMemcpyDeviceToHost, M.stream));
            ZV_HIP(hipMemcpyAsync(h_wav, d_wav, wav_bytes, hipMemcpyDeviceToHost, M.stream));
            ZV_HIP(hipEventRecord(M.tail_event(1), M.stream));
        }
        else
        {
            M.chain_dev(bt, d_ids, d_pun, d_sty, d_hid, d_mel, d_wav, d_nf, pin, d_in, b_in, 1);
            ZV_HIP(hipMemcpyAsync(h_nf, d_nf, (size_t)bt.nseg * 4, hipMemcpyDeviceToHost, M.stream));
            for (int g = 1; g < G; g++)               // contiguous groups of about wav_bytes / G each
            {
                uint32_t u = pb.gb[g - 1] + 1;
                while (u < n_utt && pb.woff[u] < wav_bytes * g / G) u++;
                pb.gb[g] = std::min(u, n_utt - (uint32_t)(G - g));
            }
            for (int g = 0; g < G; g++)
            {
                const uint32_t u0 = pb.gb[g], u1 = pb.gb[g + 1];
                M.vocode_tail(bt, d_mel, d_wav, (int)u0, (int)(u1 - u0));
                const size_t o0 = pb.woff[u0], o1 = u1 < n_utt ? pb.woff[u1] : wav_bytes;
                ZV_HIP(hipEventRecord(M.tail_event(2 * g), M.stream));
                ZV_HIP(hipStreamWaitEvent(cs, M.tail_event(2 * g), 0));
                ZV_HIP(hipMemcpyAsync(h_wav + o0, (const char *)d_wav + o0, o1 - o0, hipMemcpyDeviceToHost, cs));
                ZV_HIP(hipEventRecord(M.tail_event(2 * g + 1), cs));
            }
            ZV_HIP(hipEventRecord(M.batch_event(seq, 1), M.stream));
        }
    }
    catch (...)
    {
        hipStreamSynchronize(M.stream);
        if (M.copy_stream()) hipStreamSynchronize(M.copy_stream());
        throw;
    }
    pb.active = true;
    pb.n_utt = n_utt;
    pb.G = G;
    pb.T.assign(T, T + n_utt);
    pb.wav.assign(wav, wav + n_utt);
    pb.n_frames = n_frames;
    pb.h_wav = h_wav;
    pb.h_nf = h_nf;
    pb.hop = hop;
}

static void batch_finish(zv_model *m, int lane)
{
    Model &M = *m->m;
    PendingBatch &pb = pending_slot(m, lane);
    if (!pb.active) zv::fail(ZV_ERR_ARG, "lane %d has no batch in flight", lane);
    M.select_lane(lane);
    try
    {
        for (int g = 0; g < pb.G; g++)
        {
            ZV_HIP(hipEventSynchronize(M.tail_event(2 * g + 1)));
            scatter_out(pb.h_wav, pb.woff.data(), pb.wav.data(), pb.T.data(), pb.hop, pb.gb[g], pb.gb[g + 1]);
        }
        M.sync();                                 // the frame counts (and, unsplit, the waveforms) travel on the lane's stream
    }
    catch (...)
    {
        // a failed wait: drain what can be drained, then give the lane up as idle (its buffers are no longer in use)
        hipStreamSynchronize(M.stream);
        if (M.copy_stream()) hipStreamSynchronize(M.copy_stream());
        pb.active = false;
        throw;
    }
    pb.active = false;
    if (pb.n_frames)
        for (uint32_t u = 0; u < pb.n_utt; u++) pb.n_frames[u] = (uint32_t)pb.h_nf[u];
}

static void batch_check(Model &M, uint32_t n_utt, const int32_t *const *ids, const int32_t *const *puncts, const float *const *styles,
                        const uint32_t *n_phonemes, const uint32_t *T, float *const *wav)
{
    for (uint32_t u = 0; u < n_utt; u++)
    {
        ZV_NEED(ids[u] && puncts[u] && styles[u] && wav[u], "null utterance pointer");
        ZV_NEED(n_phonemes[u] > 0 && T[u] > 0, "n and T must be > 0");
        check_T(M, T[u]);
        if (n_phonemes[u] > M.max_phonemes())
            zv::fail(ZV_ERR_ARG, "utterance %u: %u phonemes exceed the %u rows of the sinusoid table", u, n_phonemes[u], M.max_phonemes());
        check_ids(M, ids[u], puncts[u], n_phonemes[u]);
    }
}

// how many utterances from `a` on form one launch group: up to 64 utterances / 64 Ki frames of capacity
static uint32_t batch_group_end(uint32_t a, uint32_t n_utt, const uint32_t *T)
{
    uint32_t b = a, tmax = 0;
    while (b < n_utt && b - a < 64)
    {
        const uint32_t tm = std::max(tmax, T[b]);
        if (b > a && (uint64_t)(b - a + 1) * zv::round_up((int)tm, 64) > 65536) break;
        tmax = tm;
        b++;
    }
    return b;
}

zv_status zv_synthesize_batch(zv_model *m, uint32_t n_utt, const int32_t *const *ids, const int32_t *const *puncts,
                              const float *const *styles, const uint32_t *n_phonemes, const uint32_t *T, float *const *wav,
                              uint32_t *n_frames)
{
    return guarded([&] {
        ZV_NEED(m && ids && puncts && styles && n_phonemes && T && wav, "null argument");
        Model &M = *m->m;
        ZV_HIP(hipSetDevice(M.device));
        batch_check(M, n_utt, ids, puncts, styles, n_phonemes, T, wav);
        // Groups of up to 64 utterances / 64 Ki frames go through the chain as ONE launch per kernel: every tensor is
        // the row concatenation of the group, the segment tables tell the kernels where each utterance starts and ends.
        // Capacities are rounded up so that batches of similar shape replay the same captured graph.
        uint32_t a = 0;
        while (a < n_utt)
        {
            const uint32_t b = batch_group_end(a, n_utt, T);
            batch_enqueue(m, 0, b - a, ids + a, puncts + a, styles + a, n_phonemes + a, T + a, wav + a, n_frames ? n_frames + a : nullptr);
            batch_finish(m, 0);
            a = b;
        }
    });
}

zv_status zv_synthesize_batch_begin(zv_model *m, uint32_t lane, uint32_t n_utt, const int32_t *const *ids, const int32_t *const *puncts,
                                    const float *const *styles, const uint32_t *n_phonemes, const uint32_t *T, float *const *wav,
                                    uint32_t *n_frames)
{
    return guarded([&] {
        ZV_NEED(m && ids && puncts && styles && n_phonemes && T && wav, "null argument");
        ZV_NEED(lane < ZV_BATCH_LANES, "lane out of range");
        ZV_NEED(n_utt > 0, "empty batch");
        Model &M = *m->m;
        ZV_HIP(hipSetDevice(M.device));
        batch_check(M, n_utt, ids, puncts, styles, n_phonemes, T, wav);
        ZV_NEED(batch_group_end(0, n_utt, T) == n_utt, "an asynchronous batch must fit one launch group (64 utterances, 64 Ki frames of capacity)");
        batch_enqueue(m, (int)lane, n_utt, ids, puncts, styles, n_phonemes, T, wav, n_frames);
    });
}

zv_status zv_synthesize_batch_end(zv_model *m, uint32_t lane)
{
    return guarded([&] {
        ZV_NEED(m, "null argument");
        ZV_NEED(lane < ZV_BATCH_LANES, "lane out of range");
        ZV_HIP(hipSetDevice(m->m->device));
        batch_finish(m, (int)lane);
    });
}

zv_status zv_batch_timeline(zv_model *m, uint32_t cap, double *start_ms, double *end_ms, uint32_t *n)
{
    return guarded([&] {
        ZV_NEED(m && start_ms && end_ms && n, "null argument");
        Model &M = *m->m;
        ZV_HIP(hipSetDevice(M.device));
        M.sync_all_lanes();
        const uint64_t total = M.batch_seq();
        const uint64_t cnt = std::min<uint64_t>(std::min<uint64_t>(cap, total), (uint64_t)Model::BATCH_RING);
        *n = (uint32_t)cnt;
        if (!cnt) return;
        const uint64_t first = total - cnt;
        hipEvent_t base = M.batch_event(first, 0);
        for (uint64_t i = 0; i < cnt; i++)
        {
            float a = 0.f, b = 0.f;
            ZV_HIP(hipEventElapsedTime(&a, base, M.batch_event(first + i, 0)));
            ZV_HIP(hipEventElapsedTime(&b, base, M.batch_event(first + i, 1)));
            start_ms[i] = a;
            end_ms[i] = b;
        }
    });
}

// ---- one layer at a time (tests: teacher-forced per-layer parity) -------------------------------------------------

zv_status zv_debug_layer(zv_model *m, int kind, int index, const float *x, uint32_t rows, const float *style, float *out)
{
    return guarded([&] {
        ZV_NEED(m && x && out, "null argument");
        ZV_NEED(rows > 0, "rows must be > 0");
        Model &M = *m->m;
        use_lane0(m);
        ZV_NEED(!M.graph_mode, "zv_debug_layer runs eagerly: turn graph mode off");
        const size_t E = M.E(), Mm = M.hp.audio_num_mels, hop = M.hp.audio_hop_size;
        std::vector<float> zsty(E, 0.f);
        const float *sty = style ? style : zsty.data();
        M.dbg_layer = Model::DebugLayer();
        M.dbg_layer.kind = kind;
        M.dbg_layer.index = index;
        M.dbg_layer.x = x;
        M.dbg_layer.out = out;
        struct Reset { Model &M; ~Reset() { M.dbg_layer = Model::DebugLayer(); } } reset{M};
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        if (kind == ZV_LAYER_VOC_RESBLOCK || kind == ZV_LAYER_VOC_UPSAMPLE || kind == ZV_LAYER_VOC_INPUT || kind == ZV_LAYER_VOC_OUTPUT)
        {
            // rows are rows at the layer's INPUT rate: frames x samples per frame of the stage the layer reads
            uint32_t rate = 1;
            if (kind == ZV_LAYER_VOC_RESBLOCK)
            {
                const int stage = index / (int)M.hp.voc_num_resblocks;
                ZV_NEED(index >= 0 && stage < (int)M.hp.voc_num_upsamples, "residual block index out of range");
                rate = (uint32_t)M.voc_stage_rate(stage);
            }
            else if (kind == ZV_LAYER_VOC_UPSAMPLE)
            {
                ZV_NEED(index >= 0 && index < (int)M.hp.voc_num_upsamples, "upsample index out of range");
                rate = index == 0 ? 1u : (uint32_t)M.voc_stage_rate(index - 1);
            }
            else if (kind == ZV_LAYER_VOC_OUTPUT)
                rate = (uint32_t)hop;
            ZV_NEED(rows % rate == 0, "rows must be a multiple of the layer's samples per frame");
            const uint32_t T = rows / rate;
            check_T(M, T);
            char *io = (char *)M.io_scratch(al((size_t)T * Mm * 4) + (size_t)T * hop * 4);
            if (kind == ZV_LAYER_VOC_INPUT)
                ZV_HIP(hipMemcpyAsync(io, x, (size_t)T * Mm * 4, hipMemcpyHostToDevice, M.stream));      // the layer's input IS the mel
            else
                ZV_HIP(hipMemsetAsync(io, 0, (size_t)T * Mm * 4, M.stream));
            M.vocode_dev(zv::Batch::single(1, T, 1), (const float *)io, (float *)(io + al((size_t)T * Mm * 4)));
        }
        else if (kind == ZV_LAYER_ENC_FFT || kind == ZV_LAYER_VAR_PRED || kind == ZV_LAYER_ENC_EMBED || kind == ZV_LAYER_ENC_MHA ||
                 kind == ZV_LAYER_ENC_FFN)
        {
            const uint32_t n = rows, T = 8;
            char *io = (char *)M.io_scratch(256 + 2 * al((size_t)n * 4) + al(E * 4) + (size_t)T * E * 4);
            int32_t *d_nf = (int32_t *)io;
            int32_t *d_ids = (int32_t *)(io + 256), *d_pun = (int32_t *)(io + 256 + al((size_t)n * 4));
            float *d_sty = (float *)(io + 256 + 2 * al((size_t)n * 4)), *d_hid = (float *)((char *)d_sty + al(E * 4));
            ZV_HIP(hipMemsetAsync(io, 0, 256 + 2 * al((size_t)n * 4), M.stream));
            std::vector<int32_t> hid, hpu;
            if (kind == ZV_LAYER_ENC_EMBED)
            {
                // x[n] = (phoneme id, punctuation id) as floats
                hid.resize(n);
                hpu.resize(n);
                for (uint32_t i = 0; i < n; i++)
                {
                    hid[i] = (int32_t)x[2 * i];
                    hpu[i] = (int32_t)x[2 * i + 1];
                }
                check_ids(M, hid.data(), hpu.data(), n);
                ZV_HIP(hipMemcpyAsync(d_ids, hid.data(), (size_t)n * 4, hipMemcpyHostToDevice, M.stream));
                ZV_HIP(hipMemcpyAsync(d_pun, hpu.data(), (size_t)n * 4, hipMemcpyHostToDevice, M.stream));
                ZV_HIP(hipStreamSynchronize(M.stream));            // the staging vectors go out of scope below
            }
            ZV_HIP(hipMemcpyAsync(d_sty, sty, E * 4, hipMemcpyHostToDevice, M.stream));
            M.encode_dev(zv::Batch::single(n, T, n), d_ids, d_pun, d_sty, d_hid, d_nf);
        }
        else if (kind == ZV_LAYER_DEC_BLOCK || kind == ZV_LAYER_DEC_ASR_RES || kind == ZV_LAYER_DEC_TO_OUT || kind == ZV_LAYER_DEC_ADAIN)
        {
            const uint32_t T = rows;
            check_T(M, T);
            const size_t b_hid = al((size_t)T * E * 4), b_sty = al(E * 4);
            char *io = (char *)M.io_scratch(b_hid + b_sty + (size_t)T * Mm * 4);
            float *d_sty = (float *)io, *d_hid = (float *)(io + b_sty), *d_mel = (float *)(io + b_sty + b_hid);
            if (kind == ZV_LAYER_DEC_ASR_RES)
                ZV_HIP(hipMemcpyAsync(d_hid, x, (size_t)T * E * 4, hipMemcpyHostToDevice, M.stream));     // asr_res reads the stage input itself
            else
                ZV_HIP(hipMemsetAsync(d_hid, 0, b_hid, M.stream));
            ZV_HIP(hipMemcpyAsync(d_sty, sty, E * 4, hipMemcpyHostToDevice, M.stream));
            M.decode_dev(zv::Batch::single(1, T, 1), d_hid, d_sty, d_mel);
        }
        else
            zv::fail(ZV_ERR_ARG, "unknown layer kind %d", kind);
        M.sync();
        if (!M.dbg_layer.done) zv::fail(ZV_ERR_ARG, "layer (%d, %d) does not exist in this model", kind, index);
    });
}

zv_status zv_debug_set(const char *name, int value)
{
    return guarded([&] {
        if (!name)
        {
            zv::knob_reset();
            return;
        }
        if (!zv::knob_set(name, value)) zv::fail(ZV_ERR_ARG, "unknown switch '%s'", name);
    });
}

zv_status zv_debug_get(const char *name, int *value)
{
    return guarded([&] {
        ZV_NEED(name && value, "null argument");
        if (!zv::knob_get(name, value)) zv::fail(ZV_ERR_ARG, "unknown switch '%s'", name);
    });
}

// ---- device-resident entry points ---------------------------------------------------------------

void *zv_device_alloc(zv_model *m, size_t bytes)
{
    if (!m) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(m->m->device) != hipSuccess) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess)
    {
        g_last_error = "hipMalloc failed";
        return nullptr;
    }
    return p;
}

void zv_device_free(zv_model *m, void *p)
{
    if (!m || !p) return;
    hipSetDevice(m->m->device);
    try { m->m->sync_all_lanes(); } catch (...) {}
    hipFree(p);
}

zv_status zv_memcpy_h2d(zv_model *m, void *dst, const void *src, size_t bytes)
{
    return guarded([&] {
        ZV_NEED(m && dst && src, "null argument");
        // lane 0's stream, the one zv_vocode_device / zv_decode_device run on, whatever lane was touched last (the lanes' streams
        // are not ordered with each other); no busy check: the copy only queues behind what lane 0 already holds
        lane0_select(m);
        ZV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, m->m->stream));
        m->m->sync();
    });
}

zv_status zv_memcpy_d2h(zv_model *m, void *dst, const void *src, size_t bytes)
{
    return guarded([&] {
        ZV_NEED(m && dst && src, "null argument");
        lane0_select(m);
        ZV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, m->m->stream));
        m->m->sync();
    });
}

zv_status zv_vocode_device(zv_model *m, const float *d_mel, uint32_t T, float *d_wav)
{
    return guarded([&] {
        ZV_NEED(m && d_mel && d_wav, "null argument");
        check_T(*m->m, T);
        use_lane0(m);
        m->m->vocode_dev_graph(zv::Batch::single(1, T, 1), d_mel, d_wav);
    });
}

zv_status zv_decode_device(zv_model *m, const float *d_hidden, const float *d_style, uint32_t T, float *d_mel)
{
    return guarded([&] {
        ZV_NEED(m && d_hidden && d_style && d_mel, "null argument");
        check_T(*m->m, T);
        use_lane0(m);
        m->m->decode_dev(zv::Batch::single(1, T, 1), d_hidden, d_style, d_mel);
    });
}

zv_status zv_synchronize(zv_model *m)
{
    return guarded([&] {
        ZV_NEED(m, "null model");
        ZV_HIP(hipSetDevice(m->m->device));
        // every lane: whatever entry point enqueued it, the work is done when this returns (a batch begun with
        // zv_synthesize_batch_begin still needs its _end: that is where its waveforms leave the staging block)
        m->m->sync_all_lanes();
    });
}

zv_status zv_set_graph_mode(zv_model *m, int on)
{
    return guarded([&] {
        ZV_NEED(m, "null model");
        m->m->graph_mode = on != 0;
    });
}

// ---- measurement -----------------------------------------------------------------------------

zv_status zv_profile_begin(zv_model *m)
{
    return guarded([&] {
        ZV_NEED(m, "null model");
        lane0_select(m);                      // the synchronous entry points, which are what gets profiled, run on lane 0
        m->m->sync();
        m->m->prof_clear();
        m->m->profiling = true;
    });
}

zv_status zv_profile_end(zv_model *m, zv_kernel_stat *stats, uint32_t cap, uint32_t *n)
{
    return guarded([&] {
        ZV_NEED(m && n, "null argument");
        Model &M = *m->m;
        lane0_select(m);
        M.sync();
        M.profiling = false;
        std::map<std::string, zv_kernel_stat> agg;
        std::vector<std::string> order;
        for (auto &p : M.prof)
        {
            float ms = 0.f;
            ZV_HIP(hipEventElapsedTime(&ms, p.e0, p.e1));
            auto it = agg.find(p.name);
            if (it == agg.end())
            {
                zv_kernel_stat s;
                memset(&s, 0, sizeof(s));
                strncpy(s.name, p.name, sizeof(s.name) - 1);
                it = agg.emplace(p.name, s).first;
                order.push_back(p.name);
            }
            it->second.launches += (uint32_t)p.launches;
            it->second.total_ms += ms;
            it->second.algo_bytes += p.bytes;
            it->second.algo_flops += p.flops;
        }
        M.prof_clear();
        uint32_t k = 0;
        for (auto &name : order)
        {
            if (stats && k < cap) stats[k] = agg[name];
            k++;
        }
        *n = k;
    });
}

zv_status zv_gguf_inspect(const char *gguf_path, uint32_t *n_tensors, uint32_t *max_seq_len, int tensor_index,
                          char *name_out, uint32_t *type_out, int64_t *ne_out)
{
    return guarded([&] {
        ZV_NEED(gguf_path, "null path");
        zv::GgufFile g;
        g.open(gguf_path);
        if (n_tensors) *n_tensors = (uint32_t)g.tensors().size();
        if (max_seq_len) *max_seq_len = g.get_u32("zerovox-resnet-fs2-styletts.max_seq_len");
        if (tensor_index >= 0)
        {
            if ((size_t)tensor_index >= g.tensors().size()) zv::fail(ZV_ERR_ARG, "tensor index %d out of range", tensor_index);
            const zv::GgufTensor &t = g.tensors()[tensor_index];
            if (name_out) { strncpy(name_out, t.name.c_str(), 63); name_out[63] = 0; }
            if (type_out) *type_out = t.type;
            if (ne_out) for (int i = 0; i < 4; i++) ne_out[i] = t.ne[i];
        }
    });
}

// ---- WAV writer (reference src/zerovox.cpp:337-391 uses libsndfile SF_FORMAT_WAV | SF_FORMAT_PCM_16) ----

zv_status zv_write_wav(const char *path, const float *wav, size_t n_samples, uint32_t sampling_rate)
{
    return guarded([&] {
        ZV_NEED(path && wav, "null argument");
        FILE *f = fopen(path, "wb");
        if (!f) zv::fail(ZV_ERR_IO, "cannot open '%s' for writing", path);
        if (n_samples > (size_t)0x7FFFFFE0u / 2) zv::fail(ZV_ERR_ARG, "%zu samples do not fit a RIFF/WAVE file (32-bit sizes)", n_samples);
        const uint32_t data_bytes = (uint32_t)(n_samples * 2);
        uint8_t hdr[44];
        auto put32 = [&](int o, uint32_t v) { for (int i = 0; i < 4; i++) hdr[o + i] = (uint8_t)(v >> (8 * i)); };
        auto put16 = [&](int o, uint16_t v) { hdr[o] = (uint8_t)v; hdr[o + 1] = (uint8_t)(v >> 8); };
        memcpy(hdr, "RIFF", 4);
        put32(4, 36 + data_bytes);
        memcpy(hdr + 8, "WAVEfmt ", 8);
        put32(16, 16);
        put16(20, 1);                 // PCM
        put16(22, 1);                 // mono
        put32(24, sampling_rate);
        put32(28, sampling_rate * 2);
        put16(32, 2);
        put16(34, 16);
        memcpy(hdr + 36, "data", 4);
        put32(40, data_bytes);
        bool ok = fwrite(hdr, 1, 44, f) == 44;
        std::vector<int16_t> pcm(n_samples);
        for (size_t i = 0; i < n_samples; i++)
        {
            // libsndfile float -> PCM16: scale by 0x7FFF (normalisation on), round to nearest, clip
            float v = wav[i] * 32767.0f;
            long q = lrintf(v);
            if (q > 32767) q = 32767;
            if (q < -32768) q = -32768;
            pcm[i] = (int16_t)q;
        }
        ok = ok && fwrite(pcm.data(), 2, n_samples, f) == n_samples;
        ok = (fclose(f) == 0) && ok;
        if (!ok) zv::fail(ZV_ERR_IO, "short write to '%s'", path);
    });
}

}  // extern "C"

// model.h — the model object behind zv_model: GGUF weights re-laid-out in HBM, a static activation
// arena and the fixed kernel schedule of the three stages.
//
// Replaces ZeroVOXModel's loader (reference src/zerovox.cpp:21-179) and the three ggml graphs built in
// the stage constructors (src/fs2encoder.cpp:477-586, src/stylettsdec.cpp:306-449, src/hifigan.cpp:187-356).
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "gguf_reader.h"
#include "kernels.h"

namespace zv
{

// one Conv1d layer resident in HBM: weights in MFMA fragment order, bias padded with zeros
struct ConvW
{
    void  *w = nullptr;
    void  *w8 = nullptr;        // conv_gemm_kernel's stream order (wide decoder convs), or null
    float *bias = nullptr;
    int    K = 0, Cin = 0, Cout = 0, Cin_p = 0, Cout_p = 0, ck = 0;
};

struct DeviceArena
{
    char  *base = nullptr;
    size_t cap = 0, used = 0;
    void  *take(size_t bytes)
    {
        const size_t a = (used + 255) & ~(size_t)255;
        if (a + bytes > cap) fail(ZV_ERR_OOM, "device arena overflow (%zu + %zu > %zu)", a, bytes, cap);
        used = a + bytes;
        return base + a;
    }
    template <typename T> T *take_n(size_t n) { return (T *)take(n * sizeof(T)); }
};

struct ProfEntry
{
    const char *name;
    hipEvent_t  e0, e1;
    double      bytes, flops;
    int         launches;       // consecutive launches of the same family bracketed by this event pair
};

// One call's utterances (host side).  Tensors of a batch are row-concatenated: utterance u owns the token rows /
// frame rows its table entries name (kernels.h: Seg / Segs).  `nseg`, `n_max`, `t_max`, `n_rows`, `t_rows` are
// CAPACITIES: they size grids and the arena, the real extents are read from the tables on the device, so one captured
// graph serves every batch that fits.  A single utterance needs no table (inline segment).
struct Batch
{
    int        nseg = 1;
    int        n_max = 0, t_max = 0;        // >= every utterance's phonemes / frames
    int        n_real = 0;                  // the longest utterance's real phoneme count (checks; not part of a graph's key)
    size_t     n_rows = 0, t_rows = 0;      // >= sum of phonemes / frames (rows of the concatenated buffers)
    const Seg *d_tok = nullptr, *d_frm = nullptr;
    Seg        tok1{0, 0, 0, 0}, frm1{0, 0, 0, 0};

    static Batch single(uint32_t N, uint32_t T, uint32_t num_phonemes)
    {
        Batch b;
        b.n_max = (int)N;
        b.n_real = (int)N;
        b.t_max = (int)T;
        b.n_rows = N;
        b.t_rows = T;
        b.tok1 = Seg{0, (int32_t)N, (int32_t)num_phonemes, 0};
        b.frm1 = Seg{0, (int32_t)T, 0, 0};
        return b;
    }
    Segs tokens() const { return Segs{d_tok, nseg, n_max, tok1}; }
    // every token row of the batch as ONE segment (table entry nseg: rows [0, sum of phonemes)) — for the per-token layers, whose row
    // tiles then pack the utterances densely instead of padding every utterance to a tile
    Segs tokens_merged() const { return d_tok ? Segs{d_tok + nseg, 1, (int)n_rows, tok1} : tokens(); }
    Segs frames() const { return Segs{d_frm, nseg, t_max, frm1}; }
};

// a captured schedule: replayed when the same entry point is called with the same capacities and buffers
struct CapturedGraph
{
    int            kind = 0;               // 0 vocoder, 1 chain
    unsigned       epoch = 0;              // knob_epoch() at capture
    Batch          b;
    const void    *p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t exec = nullptr;
};

class Model
{
  public:
    Model(const std::string &gguf_path, int device);
    ~Model();

    zv_hparams hp{};
    int        device = 0;
    int        n_cu = 256;
    hipStream_t stream = nullptr;

    // ---- stages (device pointers in, device pointers out; everything enqueued on `stream`) ----
    // every tensor is the row concatenation over the batch: mel [t_rows][M], wav [t_rows * hop], hidden [t_rows][E],
    // ids / puncts [n_rows], styles [nseg][E]
    void vocode_dev(const Batch &b, const float *d_mel, float *d_wav);
    // part: 0 = the whole schedule, 1 = everything up to and including the last stage's upsample conv, 2 = the last
    // stage's residual blocks + the output conv
    void vocode_group(const Batch &b, const float *d_mel, float *d_wav, int part = 0);
    // the tail (part 2) of segments [g0, g0 + cnt) of a batch whose head chain_dev(..., voc_part = 1) has enqueued: same
    // arena layout, same row ranges, same bits as the unsplit schedule
    void vocode_tail(const Batch &b, const float *d_mel, float *d_wav, int g0, int cnt);
    // second stream + events for the waveform download of a finished group under the next group's kernels
    hipStream_t copy_stream();
    hipEvent_t  tail_event(int i);
    // Batches in flight (zv_synthesize_batch_begin) are numbered in the order they are enqueued; each owns a (start, done) pair
    // of timing events on its lane's stream: `start` ahead of its upload, `done` behind its last kernel (zv_batch_timeline).
    static constexpr int BATCH_RING = 64;
    uint64_t    next_batch_seq() { return batch_seq_++; }
    uint64_t    batch_seq() const { return batch_seq_; }
    hipEvent_t  batch_event(uint64_t seq, int which);          // which: 0 start, 1 done
    int         tail_groups() const { return tail_groups_; }
    void decode_dev(const Batch &b, const float *d_hidden, const float *d_styles, float *d_mel);
    // taps are device pointers inside the arena (token rows as in ids), valid until the next call;
    // n_frames [nseg] is written to d_nframes (outside the arena)
    struct EncoderTaps
    {
        float   *features = nullptr, *logdur = nullptr, *pitch = nullptr, *energy = nullptr;
        int32_t *pitch_bucket = nullptr, *energy_bucket = nullptr, *cum = nullptr;
    };
    EncoderTaps encode_dev(const Batch &b, const int32_t *d_ids, const int32_t *d_puncts, const float *d_styles,
                           float *d_hidden, int32_t *d_nframes);

    // One layer at a time (the counterpart of the reference's tensor_dbg, src/utils.cpp:19-44; tests only): while
    // `dbg_layer.kind >= 0` the stage that owns the layer replaces the layer's input with dbg_layer.x (host, time-major,
    // unpadded) right before it runs, copies the layer's output to dbg_layer.out right after it and returns.
    // kinds (include/zerovox_amd.h zv_layer_kind): 0 HiFi-GAN residual block n, 1 encoder FFT block l,
    // 2 decoder residual block b (0,1 encode; 2..6 decode), 3 variance predictor p (0 duration, 1 pitch, 2 energy)
    struct DebugLayer
    {
        int          kind = -1, index = 0;
        const float *x = nullptr;
        float       *out = nullptr;
        bool         done = false;
    } dbg_layer;

    // rows of the sinusoid table = the longest utterance the encoder takes (reference src/fs2encoder.cpp:306-324)
    uint32_t max_phonemes() const { return (uint32_t)enc_.posenc_rows; }
    void reserve(uint32_t max_phonemes, uint32_t max_frames);
    void reserve_batch(const Batch &b);
    int  voc_stage_rate(int stage) const;       // samples per frame after upsample stage `stage`
    int  voc_stage_channels(int stage) const;
    // largest frame count one segment may have: byte offsets inside a segment are 32-bit in the buffer descriptors
    uint32_t max_frames_per_utterance() const;
    int wemb_rows() const { return enc_.wemb_rows; }
    int pemb_rows() const { return enc_.pemb_rows; }
    // receptive field of the vocoder in mel frames per side (input conv + per stage: transposed-conv taps and the widest
    // residual block, converted from the stage's sample rate), rounded up, + 1
    uint32_t vocoder_halo_frames() const;
    void sync();

    // Lanes: independent (stream, activation arena, I/O scratch) triples so that several utterances are in flight at
    // once (zv_synthesize_batch): a single short utterance cannot fill 256 CUs in its narrow stages, four can.
    // `stream` / the arena below always refer to the selected lane; lane 0 is the default.
    void select_lane(int i);
    int  current_lane() const { return cur_lane_; }
    void sync_all_lanes();

    // scratch for host-buffer entry points (grows on demand)
    void *io_scratch(size_t bytes);
    // pinned host staging for batched D2H copies (an async copy into pageable memory blocks the host and would
    // serialise the lanes)
    void *pinned_scratch(size_t bytes);

    // graph replay of the vocoder schedule
    bool graph_mode = false;
    void vocode_dev_graph(const Batch &b, const float *d_mel, float *d_wav);
    // encoder -> decoder -> vocoder back to back (graph replay keyed by (capacities, buffers) when graph_mode is on).
    // h2d_src / h2d_dst / h2d_bytes (optional): an input upload that becomes the first node of the schedule
    void chain_dev(const Batch &b, const int32_t *d_ids, const int32_t *d_puncts, const float *d_styles, float *d_hidden,
                   float *d_mel, float *d_wav, int32_t *d_nframes, const void *h2d_src = nullptr, void *h2d_dst = nullptr,
                   size_t h2d_bytes = 0, int voc_part = 0);

    // profiling (HIP events around every launch while enabled)
    bool profiling = false;
    std::vector<ProfEntry> prof;
    void prof_clear();

    uint32_t E() const { return hp.emb_dim + hp.punct_emb_dim; }

  private:
    // ---- weights ----
    std::vector<void *> allocs_;
    void  *dev_alloc(size_t bytes);
    float *upload_f32(const GgufTensor &t, int pad_to = 0, float pad_value = 0.f);
    float *upload_vec(const GgufFile &g, const std::string &name, int expect_n, int pad_to = 0, float pad_value = 0.f);
    ConvW  load_conv(const GgufFile &g, const std::string &wname, const std::string &bname, int expect_cin = -1, bool gemm_pack = false);
    ConvW  load_upsample(const GgufFile &g, int idx, int stride, int expect_cin);

    // p1/p2: 32 x 32 x 16 fragment order (resblock_triple_kernel), x1/x2: 16 x 16 x 32 order (pair kernel, block32), r1/r2: LDS-ring stream (64 channels)
    struct ResPair { ConvW c1, c2; void *p1 = nullptr, *p2 = nullptr, *r1 = nullptr, *r2 = nullptr, *x1 = nullptr, *x2 = nullptr; };
    struct Voc
    {
        float *mean = nullptr, *scale = nullptr;
        ConvW  in_conv;
        int    n_up = 0;
        int    scales[8] = {0};
        ConvW  ups[8];
        int    n_rb = 0, n_dil = 3;
        int    dil[8] = {1, 3, 5};
        std::vector<ResPair> pairs;       // [(stage*n_rb + j)*n_dil + d]
        uint16_t *out_w = nullptr;        // f16 [K][Cp]
        float  out_b = 0.f;
        int    out_K = 0, out_C = 0;
    } voc_;

    struct DecBlk
    {
        ConvW  conv1, conv2, sc;
        bool   learned_sc = false;
        int    cin = 0, cout = 0;
        float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr;   // encode blocks: affine IN
        int    g1 = 0, g2 = 0;                                                   // decode blocks: offsets into adain h
    };
    struct Dec
    {
        DecBlk enc[2], dec[5];
        ConvW  asr0, to_out;
        float *asr1w = nullptr, *asr1b = nullptr;
        float *fcW = nullptr, *fcB = nullptr, *fcExtra = nullptr;   // all 10 AdaIN fc layers concatenated
        int    fc_out = 0;
        int    R = 64, M = 80;
    } dec_;

    struct EncLayer
    {
        float *qkvW = nullptr, *qkvB = nullptr, *fcW = nullptr, *fcB = nullptr;
        float *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr;
        ConvW  w1, w2;
    };
    struct VarPred
    {
        ConvW  c1, c2;
        float *l1w = nullptr, *l1b = nullptr, *l2w = nullptr, *l2b = nullptr, *lw = nullptr, *lb = nullptr;
        int    V = 0;
    };
    struct Enc
    {
        float *wemb = nullptr, *pemb = nullptr, *posenc = nullptr, *pitch_emb = nullptr, *energy_emb = nullptr;
        int    posenc_rows = 0, wemb_rows = 0, pemb_rows = 0;
        std::vector<EncLayer> layers;
        VarPred dur, pitch, energy;
    } enc_;

    // ---- activations ----
    DeviceArena arena_;
    void  arena_require(size_t bytes);
    size_t arena_bytes_for(size_t n_rows, size_t t_rows, int nseg) const;
    void *io_ = nullptr;
    size_t io_cap_ = 0;

    // ---- launch helpers ----
    void conv(const ConvJob *jobs, int n, const Segs &segs, int rate, const char *name, double bytes, double flops);
    void dbg_inject(void *dev, int ld, int cols, size_t rows);
    void dbg_extract(const void *dev, int ld, int cols, size_t rows);
    ConvJob job(const ConvW &w) const;
    void tick(const char *name, double bytes, double flops, hipEvent_t *e0);
    void tock(hipEvent_t e0, const char *name, double bytes, double flops);
    void group_begin();
    void group_end(const char *name);
    bool       in_group_ = false;
    hipEvent_t group_e0_ = nullptr;
    double     group_bytes_ = 0.0, group_flops_ = 0.0;
    int        group_n_ = 0;

    struct Lane
    {
        hipStream_t stream = nullptr;
        DeviceArena arena;
        void       *io = nullptr;
        size_t      io_cap = 0;
        // host staging block, download stream and events of the lane (a batch in flight per lane: zv_synthesize_batch_begin)
        void       *pinned = nullptr;
        size_t      pinned_cap = 0;
        hipStream_t copy_stream = nullptr;
        std::vector<hipEvent_t> tail_events;
    };
    std::vector<Lane> lanes_;
    void  *pinned_ = nullptr;
    size_t pinned_cap_ = 0;
    int  cur_lane_ = 0;
    uint64_t   batch_seq_ = 0;
    hipEvent_t batch_events_[2 * BATCH_RING] = {};
    void stash_lane();

    bool no_fuse_ = false;        // ZV_NO_FUSE=1: two launches per dilation pair (A/B measurement)
    bool no_triple_ = false;      // ZV_NO_TRIPLE=1: one launch per dilation pair also on the narrow stages (A/B measurement)
    int  tail_groups_ = 8;        // ZV_TAIL_GROUPS=G: utterance groups of a batch's last vocoder stage (0 / 1 = no split)
    bool skip_launch_ = false;    // vocode_group: the launches of the part that is not asked for are skipped
    hipStream_t copy_stream_ = nullptr;
    std::vector<hipEvent_t> tail_events_;
    bool no_merge_ = false;       // ZV_NO_MERGE=1: the last dilation pair of a stage stores its three branch outputs instead of their sum (A/B, tests)
    bool force_fuse256_ = false;  // ZV_FUSE256=1: fused kernel for the 256-channel stage at any length (tests: the path
                                  // long / batched utterances take, exercised at sizes the CPU oracle can check)
    std::vector<CapturedGraph> graphs_;
    void drop_graphs();
    template <typename F> void run_captured(int kind, const Batch &b, const void *const *key, int nkey, F &&enqueue);
};

}  // namespace zv

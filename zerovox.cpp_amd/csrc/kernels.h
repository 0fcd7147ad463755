// kernels.h — host-visible descriptors and launchers of the HIP kernels (gfx950 only).
//
// Activation layout in HBM ("tl" = time-major / channels-last): x[t * ld + c], c < Cp, where
// Cp = C rounded up to 16 and the pad channels hold zeros.  This is the layout of the stage
// boundaries themselves (hidden[t*E+e], mel[t*80+m] — reference src/fs2encoder.cpp:634,
// src/stylettsdec.cpp:432-441), so no transposes exist anywhere in the schedule, and it makes an
// MFMA operand fragment (8 consecutive input channels at one time step) one 16-byte LDS read.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing-only ablation bits (`dbg` fields of the job structs: skip staging / MFMA loops / stores ... — WRONG results) exist only
// in diagnostic builds (-DZV_DIAG, see knobs.h).  In the shipped library every read of them is the constant 0 and the code
// they select is compiled out.
#ifdef ZV_DIAG
#define ZV_DBGBITS(x) (x)
#else
#define ZV_DBGBITS(x) 0
#endif

namespace zv
{

__host__ __device__ static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

// ---- segments: many utterances per launch ---------------------------------------------------------
// A batch is laid out as ONE row-concatenated buffer per tensor: utterance u owns rows [row0, row0 + rows) (frames
// for the decoder / vocoder, phonemes for the encoder).  Every kernel maps a workgroup to (segment, tile inside the
// segment) and treats the segment's ends as the sequence ends (zero padding of the convs, InstanceNorm / attention
// extents): each utterance keeps its own (N, T), exactly as if it had been run alone (reference
// src/fs2encoder.cpp:103-110, src/stylettsdec.cpp:359: no masks, no batch padding).  The table lives in HBM, so one
// captured hipGraph serves every batch that fits its capacity; a single utterance travels inline in the kernel
// arguments (`one`) and needs no table.  `rate` (per launch) converts base rows to the rows of a stage
// (HiFi-GAN stages: 1, 5, 25, 100, 300 samples per frame).
struct Seg
{
    int32_t row0, rows;      // base units (frames or phonemes)
    int32_t aux;             // encoder: num_phonemes the length regulator walks (reference src/fs2encoder.cpp:622)
    int32_t pad;
};
struct Segs
{
    const Seg *tab;          // device table [nseg] or null
    int        nseg;         // table entries; the grid covers all of them, empty ones (rows = 0) exit at once
    int        max_rows;     // >= rows of every entry (grid sizing), base units
    Seg        one;          // the segment when tab == null
};
__host__ __device__ static inline Seg seg_at(const Segs &s, int u)
{
    Seg g = s.tab ? s.tab[u] : s.one;
#if defined(__HIP_DEVICE_COMPILE__)
    // the table entry is the same for every lane of a workgroup: keep it (and every pointer / buffer descriptor derived
    // from it) in scalar registers — a descriptor in vector registers costs a waterfall loop per buffer instruction
    g.row0 = __builtin_amdgcn_readfirstlane(g.row0);
    g.rows = __builtin_amdgcn_readfirstlane(g.rows);
    g.aux = __builtin_amdgcn_readfirstlane(g.aux);
#endif
    return g;
}
static inline Segs segs_single(int rows, int aux = 0)
{
    Segs s;
    s.tab = nullptr;
    s.nseg = 1;
    s.max_rows = rows;
    s.one.row0 = 0;
    s.one.rows = rows;
    s.one.aux = aux;
    s.one.pad = 0;
    return s;
}

// ---- fused Conv1d ("same" length, stride 1) as implicit GEMM on v_mfma_f32_32x32x16_f16 -----------
//
//   out[t][oc] = epilogue( sum_{tap, ic} f16( prologue(x)[t + tap*dil - pad][ic] ) * w[tap][ic][oc] )
//
// which is ggml_conv_1d = im2col(F16) + mul_mat (reference ggml/src/ggml.c:3769-3786): operands f16,
// products exact in f32, f32 accumulation.  Out-of-range taps are zeros *after* the prologue.

enum ConvPrologue : int
{
    PRO_RAW_F16 = 0,      // x is already the f16 operand (written by an EPI f16 store)
    PRO_ACT = 1,          // f16(lrelu(x, slope))            (slope 1 = identity, 0 = relu)
    PRO_NORM_ACT = 2,     // f16(lrelu(((x - mean_c) * rstd_c) * g_c + b_c, slope))   InstanceNorm/AdaIN
    PRO_MELNORM = 3,      // f16((x - a_c) / b_c)            (src/hifigan.cpp:242-243)
    PRO_SUM3_ACT = 4,     // f16(lrelu(((x0 + x1) + x2) * pscale, slope))   MRF mean (src/hifigan.cpp:300-315)
    PRO_SCALE_ACT = 5     // f16(lrelu(x * pscale, slope))                  MRF mean whose sum the producer already formed
};

struct ConvJob
{
    // input
    const void  *x0, *x1, *x2;
    int          ldx;
    int          pro;
    float        slope, pscale;
    const float *pa, *pb;        // per-channel: NORM g,b | MELNORM mean,scale
    const float *pstat;          // per-channel (mean, rstd) pairs
    // geometry
    int          L, Cin_p, Cout_p, K, dil, pad;
    int          ck;             // input-channel chunk staged per LDS pass (set at pack time)
    // weights packed in MFMA-fragment order (see pack_conv_weight), bias padded to 32*ntiles
    const void  *w;
    const void  *w8;             // the same weights in conv_gemm_kernel's stream order (pack_conv_weight_gemm: 16 x 16 x 32 fragments), or null
    const float *bias;
    // epilogue: v = acc + bias; v += res; v *= escale; v = lrelu(v, oslope) if eact; store f32 | f16
    const float *res;
    int          ldres;
    float        escale;
    int          eact;
    float        oslope;
    int          out_f16;
    void        *out;
    int          ldo;
    int          dbg;            // timing-only ablation bits (ZV_DBG env): 1 no staging loads, 2 no MFMA, 4 no epilogue
    // per-segment strides (floats) of the prologue's per-channel vectors: 0 = one vector for every segment
    int          pstat_seg, pab_seg;
    // InstanceNorm statistics of the OUTPUT, produced by the epilogue: per (segment, 32-row block, channel) the f64
    // pair (sum, sum of squares) of the stored values, combined later in a fixed order by launch_stats_finalize
    double      *stat_part;      // [nseg][stat_nblk][stat_C][2] or null
    int          stat_nblk, stat_C;
};

constexpr int CONV_MAX_JOBS = 4;
struct ConvJobs
{
    ConvJob j[CONV_MAX_JOBS];
    Segs    segs;
    int     rate;                // rows per base row of this launch
    int     tps;                 // row tiles per segment (grid.x = tps * nseg)
    int     nt_begin;            // first output tile of this launch (the tiles before it belong to conv_gemm_kernel)
    int     order;               // conv_gemm_kernel: workgroup order (ZV_GEMM_ORDER: 0 plain, 1 one group per XCD, 2 the 9-tile group first)
    int     tile_bytes;          // single-utterance form of conv1d_mfma_kernel: bytes of one of its two LDS tiles (set by the launcher)
    int     xcd_ny, xcd_nx;      // single-utterance form: channel groups / row tiles of the launch when the groups are dealt over the XCDs (0: plain grid)
    int     warm;                // single-utterance forms: the row tiles of a channel group warm their XCD's L2 with the group's weights first
#ifdef ZV_STAMPS
    int     stamp;               // diagnostic build: this launch writes phase stamps
#endif
};

// bytes of one packed conv weight: [ntile32][chunk][tap][kc][lane 64][8 halfs]
size_t packed_conv_weight_halfs(int Cin_p, int Cout_p, int K);
// input-channel chunk per LDS pass: min(Cin_p, ck_max); ck_max <= 0 means the kernel's maximum (256)
int    conv_pick_ck(int Cin_p, int ck_max = 0);
// host-side repack of a GGUF conv weight (ggml ne [K, IC, OC], f16, k fastest) into fragment order
void   pack_conv_weight(const uint16_t *w, int K, int IC, int OC, int Cin_p, int Cout_p, int ck, uint16_t *dst);
// conv_gemm_kernel (batches of wide f16-operand convs): [group of 8 output tiles][unit = (256-chunk, tap, 64-channel block)][k16 step 4]
// [tile 8][lane][8 halfs]: one unit = 32 KiB contiguous = what one workgroup moves into LDS per step of its K loop.  Only whole
// groups of 8 tiles are packed (conv_gemm_groups); the remaining tiles run on conv1d_mfma_kernel.
int    conv_gemm_groups(int Cout_p);
int    conv_gemm_tiles(int Cout_p);         // output tiles conv_gemm_kernel covers (whole groups of 8, + one leftover tile)
int    conv_gemm_units(int Cin_p, int K);
size_t conv_gemm_weight_halfs(int Cin_p, int Cout_p, int K);
void   pack_conv_weight_gemm(const uint16_t *w, int K, int IC, int OC, int Cin_p, int Cout_p, uint16_t *dst);
// all jobs of one launch share the segments, Cout_p and the tile configuration; job.L is ignored (rows come from segs)
hipError_t launch_conv(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu, const Segs &segs, int rate);

// ---- fused HiFi-GAN dilation pair (reference src/hifigan.cpp:99-182, one loop iteration):
//   out = y + ( conv(lrelu(conv(lrelu(y), k, dil) + b1), k, 1) + b2 )
// in ONE launch: xt never leaves LDS, y is read once (+ halo) and written once -> 8 B/element of HBM traffic
// instead of the 20 B/element the two separate convs are accounted for algorithmically.
struct PairJob
{
    const float *y;          // [L][Cp] f32
    float       *out;        // [L][Cp] f32, must not alias y (neighbour workgroups read y's halo)
    const void  *w1, *w2;    // packed by pack_pair_weight16 (16 x 16 x 32 fragments; conv1: A-operand form, conv2: B-operand form)
    const void  *w1r, *w2r;  // the same as the stream resblock_pair64_kernel moves through its LDS ring (pack_pair_weight_ring), or null
    const float *b1, *b2;
    int          L, Cp, K, dil;
    float        slope;
    int          dbg;
    // the running MRF sum, branch by branch (reference src/hifigan.cpp:300-315: (y0 + y1) + y2): when sum_out is set the pair's
    // result v goes there instead of `out`, as sum_in + v (or v itself for the first branch, sum_in null); may be the same buffer
    const float *sum_in;
    float       *sum_out;
};
constexpr int PAIR_MAX_JOBS = 3;
struct PairJobs
{
    PairJob j[PAIR_MAX_JOBS];
    Segs    segs;
    int     rate;
    int     njobs, kmax;
#ifdef ZV_STAMPS
    int     stamp;               // diagnostic build: this launch writes phase stamps
#endif
    float  *merge_out;           // non-null: store (out_0 + out_1) + out_2 here instead of the jobs' own outputs
    int     ring_off;            // resblock_pair64_kernel: byte offset of the weight ring in LDS (set by the launcher)
};
// true when a ResBlock conv pair with Cp (padded) channels and K taps can run on the fused kernel
bool       pair_supported(int Cp, int K);
size_t     pair_weight_halfs(int Cp, int K);
// GGUF conv weight (ggml ne [K, C, C], f16) -> fused-kernel layout
void       pack_pair_weight(const uint16_t *w, int K, int C, int Cp, uint16_t *dst);
// the same weight in v_mfma_f32_16x16x32_f16 fragment order (resblock_pair_kernel, resblock_block32_kernel); conv2_layout: the
// pair's second conv (B-operand form: column c of tile wt = channel 2c + wt), else the first (A-operand form: row r = channel 16 wt + r)
size_t     pair_weight16_halfs(int Cp, int K);
void       pack_pair_weight16(const uint16_t *w, int K, int C, int Cp, uint16_t *dst, bool conv2_layout);
// the same weight as the stream resblock_pair64_kernel / resblock_block64_kernel move through their LDS ring: [tap][step of 32 channels][ntile][wt][lane][8 halfs]
size_t     pair_ring_weight_halfs(int Cp, int K);
void       pack_pair_weight_ring(const uint16_t *w, int K, int C, int Cp, uint16_t *dst, bool conv2_layout);
// merge_out (may be null): the jobs share every time tile and only the sum of their outputs, (out_0 + out_1) + out_2, is
// stored there (the MRF sum of a stage's last dilation pair); the jobs' own `out` pointers are then unused
hipError_t launch_pair(hipStream_t s, const PairJob *jobs, int njobs, int n_cu, const Segs &segs, int rate, float *merge_out = nullptr);

// ---- a whole HiFi-GAN residual block (reference src/hifigan.cpp:74-185: the loop over all dilations) in ONE launch:
// a workgroup keeps a 256-row f32 tile of y in LDS, runs the n_dil fused pairs on it and writes the centre rows once.
// HBM traffic per element: 4 B in (x halo factor) + 4 B out for the whole block instead of once per dilation pair.
constexpr int TRIPLE_MAX_DIL = 3;
struct TripleJob
{
    const float *y;                       // [L][Cp] f32 block input (may be shared by several jobs)
    float       *out;                     // [L][Cp] f32 block output, must not alias y
    // launch_triple: w1 / w2 in pack_pair_weight layout (resblock_triple_kernel: 32 x 32 x 16 fragments) and w1x / w2x in
    // pack_pair_weight16 layout (resblock_block32_kernel); launch_block64: w1 / w2 the ring stream (pack_pair_weight_ring)
    const void  *w1[TRIPLE_MAX_DIL], *w2[TRIPLE_MAX_DIL];
    const void  *w1x[TRIPLE_MAX_DIL], *w2x[TRIPLE_MAX_DIL];
    const float *b1[TRIPLE_MAX_DIL], *b2[TRIPLE_MAX_DIL];
    int          dil[TRIPLE_MAX_DIL];
    int          n_dil;
    int          L, Cp, K;
    float        slope;
    int          dbg;
};
struct TripleJobs
{
    TripleJob j[PAIR_MAX_JOBS];
    Segs      segs;
    int       rate;
    int       interleave;        // resblock_block32_kernel: > 1 = that many jobs share grid.x, interleaved per XCD
    int       db_mask;           // resblock_block32_kernel: bit j = job j keeps two weight buffers in LDS (set by the launcher)
    int       ring_off;          // resblock_block64_kernel: byte offset of the weight ring in LDS (set by the launcher)
#ifdef ZV_STAMPS
    int       stamp;
#endif
};
// true when a ResBlock (Cp channels, K taps, these dilations) fits the whole-block kernel
bool       triple_supported(int Cp, int K, const int *dil, int n_dil);
// several dilation pairs of a 64-channel block in one launch (resblock_block64_kernel): w1 / w2 in pack_pair_weight_ring layout
bool       block64_supported(int Cp, int K, const int *dil, int n_dil);
hipError_t launch_block64(hipStream_t s, const TripleJob *jobs, int njobs, const Segs &segs, int rate);
hipError_t launch_triple(hipStream_t s, const TripleJob *jobs, int njobs, int n_cu, const Segs &segs, int rate);

// ---- vocoder tail: lrelu(0.01) -> conv k7 (C -> 1) + b -> tanh (src/hifigan.cpp:324-345) ----------
struct OutConvArgs
{
    const float *x0, *x1, *x2;   // MRF branches of the last stage, summed in the prologue
    int          ldx, L, C, K;
    float        pscale, slope;
    const uint16_t *w;           // f16 [K][Cp]
    float        bias;
    float       *out;            // wav[L]
    Segs         segs;
    int          rate;
};
hipError_t launch_out_conv(hipStream_t s, const OutConvArgs &a);

// ---- InstanceNorm statistics over time (ggml_norm, ggml-cpu.c:6880-6929: mean and biased variance in f64, --------
// scale = 1/sqrtf(var + eps)).  Two steps: per (segment, 32-row block, channel) partial sums (sum, sum of squares, both
// f64; written by the producing conv's epilogue, by launch_stats_partial for tensors that no conv of ours produced,
// or by launch_norm_apply for its own output), then launch_stats_finalize adds the blocks of a segment in block order
// and stores (mean, rstd).  The 32-row blocks do not depend on any tile shape, so neither do the statistics.
//   part[((u * nblk) + blk) * C + c] = double2(sum, sumsq);  nblk >= ceil(max_rows * rate / 32)
//   stat[u * stat_seg + 2 * (c_off + c) + {0, 1}] = mean, rstd
hipError_t launch_stats_partial(hipStream_t s, const float *x, int ldx, int C, double *part, int nblk, const Segs &segs, int rate);
hipError_t launch_stats_finalize(hipStream_t s, const double *part, int nblk, int C, float eps, float *stat, int stat_seg,
                                 int c_off, const Segs &segs, int rate);
// y[t][c] = ((x - mean) * rstd) * g[c] + b[c]; `part` (may be null) receives the partial sums of y
hipError_t launch_norm_apply(hipStream_t s, const float *x, int ldx, int C, const float *stat, int stat_seg, const float *g,
                             const float *b, float *y, int ldy, double *part, int nblk, const Segs &segs);

// out[i] = f16(lrelu(((x0[i] + x1[i]) + x2[i]) * pscale, slope)) over n contiguous elements (x1 = x2 = null: x0[i] * pscale): the
// operand pre-pass of the upsample convs that run on conv_gemm_kernel
hipError_t launch_act_f16(hipStream_t s, const float *x0, const float *x1, const float *x2, float pscale, float slope, void *out, size_t n);
// y = f16(lrelu(((x - mean) * rstd) * g + b, slope)) for C channels (a multiple of 4): the PRO_NORM_ACT operand written out
// once (PRO_RAW_F16 consumers).  Channels below Cpart get their statistics from `part` (and store them in `stat`), the
// others read `stat`.  g / b: per-segment stride gb_seg (0 = shared).  yraw (may be null, same layout as y): f16(x) itself — the
// operand of the block's 1x1 shortcut conv (reference src/stylettsdec.cpp:132-140,287-296 converts x to f16 in its im2col).
hipError_t launch_norm_act_f16(hipStream_t s, const float *x, int ldx, int C, const double *part, int nblk, int Cpart, float eps,
                               float *stat, int stat_seg, const float *ga, const float *be, int gb_seg, float slope, void *y,
                               int ldy, const Segs &segs, void *yraw = nullptr);

// ---- f32 linear layers: y[n][o] = dot(W[o][:], x[n][:]) + b[o] (ggml_mul_mat on f32 weights) --------
// `extra` (may be null) is a second per-output addend applied after the bias: (acc + b[o]) + extra[o]
// (AdaIN: gamma = h[:C] + 1, reference src/stylettsdec.cpp:186-189)
hipError_t launch_linear(hipStream_t s, const float *x, int ldx, int in, const float *W, const float *b, int out, float *y,
                         int ldy, const float *extra, const Segs &segs);

// ---- encoder pieces (reference src/fs2encoder.cpp) -------------------------------------------------
hipError_t launch_embed(hipStream_t s, const int32_t *ids, const int32_t *puncts, const float *wemb, int emb,
                        const float *pemb, int pdim, const float *posenc, float *x, int ld, const Segs &segs);
// softmax(q k^T * inv_temp) v per (segment, head); q, k, v [rows][ld] token-major, heads side by side
hipError_t launch_attention(hipStream_t s, const float *q, const float *k, const float *v, int ld, int H, int dk,
                            float inv_temp, float *o, int ldo, const Segs &segs);
// y = LayerNorm(x + res) * w + b  over the C real channels (res may be null); channels [C, Cp) are zeroed
hipError_t launch_add_layernorm(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int C, int Cp,
                                const float *w, const float *b, float eps, float *y, int ldy, const Segs &segs);
// the same LayerNorm with a tail in the same launch (short utterances: every launch is latency), each part optional:
//   y += post[segment][:]                                   (post_seg: floats between segments' vectors)
//   pred[row] = dot(y[row][:C], dot_w) + dot_b[0]           (launch_rowdot's chain)
//   bucket[row] = clamp((int)(pred * (nbins - 1) + 0.5)); feat[row][:embC] += emb[bucket][:]   (launch_bucket_embed_add; needs dot_w)
// layernorm_tail_ok(C): the tail exists in the rows-in-registers form only (C <= 768)
bool       layernorm_tail_ok(int C);
hipError_t launch_layernorm_tail(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int C, int Cp, const float *w,
                                 const float *b, float eps, float *y, int ldy, const Segs &segs, const float *post, int post_seg,
                                 const float *dot_w, const float *dot_b, float *pred, const float *emb, int nbins, int embC, float *feat,
                                 int ldf, int32_t *bucket);
// x[row][:] += v[segment][:]
hipError_t launch_add_rowvec(hipStream_t s, float *x, int ld, int C, const float *v, int v_seg, const Segs &segs);
// pred[n] = dot(x[n][:], w) + b
hipError_t launch_rowdot(hipStream_t s, const float *x, int ld, int C, const float *w, const float *b, float *y, const Segs &segs);
// bucket[n] = clamp((int)(pred*(nbins-1) + 0.5), 0, nbins-1); x[n][:] += emb[bucket[n]][:]
hipError_t launch_bucket_embed_add(hipStream_t s, const float *pred, int nbins, const float *emb, int C, float *x, int ld,
                                   int32_t *bucket, const Segs &segs);
// device length regulator: rounded durations of the first `aux` tokens of a segment -> inclusive scan (cum[], one int
// per token row) -> gather into the segment's frames, zero tail; n_frames[segment] = frames
hipError_t launch_length_regulator(hipStream_t s, const float *feat, int ld, const float *logdur, int C, float *hidden,
                                   int ldh, int32_t *cum, int32_t *n_frames, const Segs &tokens, const Segs &frames);

}  // namespace zv

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

namespace zv
{

__host__ __device__ static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

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
    PRO_SUM3_ACT = 4      // f16(lrelu(((x0 + x1) + x2) * pscale, slope))   MRF mean (src/hifigan.cpp:300-315)
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
    int          sk_lg_nw;       // split-K kernel: log2(output tiles per workgroup), filled in by launch_conv
    int          allow_splitk;   // the caller accepts a sequence-length-dependent summation order (decoder / encoder
                                 // convs: InstanceNorm / attention make those stages length-dependent anyway); the
                                 // vocoder never sets it: its output bits must not depend on T (zv_vocode_stream)
};

constexpr int CONV_MAX_JOBS = 4;
struct ConvJobs
{
    ConvJob j[CONV_MAX_JOBS];
};

// bytes of one packed conv weight: [ntile32][chunk][tap][kc][lane 64][8 halfs]
size_t packed_conv_weight_halfs(int Cin_p, int Cout_p, int K);
// input-channel chunk per LDS pass: min(Cin_p, ck_max); ck_max <= 0 means the kernel's maximum (256)
int    conv_pick_ck(int Cin_p, int ck_max = 0);
// host-side repack of a GGUF conv weight (ggml ne [K, IC, OC], f16, k fastest) into fragment order
void   pack_conv_weight(const uint16_t *w, int K, int IC, int OC, int Cin_p, int Cout_p, int ck, uint16_t *dst);
// all jobs of one launch share L-extent class, Cout_p and tile configuration
hipError_t launch_conv(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu);

// ---- fused HiFi-GAN dilation pair (reference src/hifigan.cpp:99-182, one loop iteration):
//   out = y + ( conv(lrelu(conv(lrelu(y), k, dil) + b1), k, 1) + b2 )
// in ONE launch: xt never leaves LDS, y is read once (+ halo) and written once -> 8 B/element of HBM traffic
// instead of the 20 B/element the two separate convs are accounted for algorithmically.
struct PairJob
{
    const float *y;          // [L][Cp] f32
    float       *out;        // [L][Cp] f32, must not alias y (neighbour workgroups read y's halo)
    const void  *w1, *w2;    // packed by pack_pair_weight (zero-padded per-tile segments)
    const float *b1, *b2;
    int          L, Cp, K, dil;
    float        slope;
    int          dbg;
};
constexpr int PAIR_MAX_JOBS = 3;
struct PairJobs
{
    PairJob j[PAIR_MAX_JOBS];
};
// true when a ResBlock with Cp (padded) channels can run on the fused kernel
bool       pair_supported(int Cp);
size_t     pair_weight_halfs(int Cp, int K);
// GGUF conv weight (ggml ne [K, C, C], f16) -> fused-kernel layout
void       pack_pair_weight(const uint16_t *w, int K, int C, int Cp, uint16_t *dst);
hipError_t launch_pair(hipStream_t s, const PairJob *jobs, int njobs, int n_cu);

// ---- a whole HiFi-GAN residual block (reference src/hifigan.cpp:74-185: the loop over all dilations) in ONE launch:
// a workgroup keeps a 256-row f32 tile of y in LDS, runs the n_dil fused pairs on it and writes the centre rows once.
// HBM traffic per element: 4 B in (x halo factor) + 4 B out for the whole block instead of once per dilation pair.
constexpr int TRIPLE_MAX_DIL = 3;
struct TripleJob
{
    const float *y;                       // [L][Cp] f32 block input (may be shared by several jobs)
    float       *out;                     // [L][Cp] f32 block output, must not alias y
    const void  *w1[TRIPLE_MAX_DIL], *w2[TRIPLE_MAX_DIL];    // pack_pair_weight layout
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
};
// true when a ResBlock (Cp channels, K taps, these dilations) fits the whole-block kernel
bool       triple_supported(int Cp, int K, const int *dil, int n_dil);
hipError_t launch_triple(hipStream_t s, const TripleJob *jobs, int njobs, int n_cu);

// ---- vocoder tail: lrelu(0.01) -> conv k7 (C -> 1) + b -> tanh (src/hifigan.cpp:324-345) ----------
struct OutConvArgs
{
    const float *x0, *x1, *x2;   // MRF branches of the last stage, summed in the prologue
    int          ldx, L, C, K;
    float        pscale, slope;
    const uint16_t *w;           // f16 [K][Cp]
    float        bias;
    float       *out;            // wav[L]
};
hipError_t launch_out_conv(hipStream_t s, const OutConvArgs &a);

// ---- InstanceNorm statistics over time (ggml_norm semantics, ggml-cpu.c:6880-6929) ----------------
// stat[c] = (mean, 1/sqrtf(var + eps)); sums accumulated in f64
hipError_t launch_in_stats(hipStream_t s, const float *x, int ld, int L, int C, float eps, float *stat);
// y[t][c] = ((x - mean) * rstd) * g[c] + b[c]
hipError_t launch_norm_apply(hipStream_t s, const float *x, int ldx, int L, int C, const float *stat,
                             const float *g, const float *b, float *y, int ldy);

// ---- f32 linear layers: y[n][o] = dot(W[o][:], x[n][:]) + b[o] (ggml_mul_mat on f32 weights) --------
// `extra` (may be null) is a second per-output addend applied after the bias: (acc + b[o]) + extra[o]
// (AdaIN: gamma = h[:C] + 1, reference src/stylettsdec.cpp:186-189)
hipError_t launch_linear(hipStream_t s, const float *x, int ldx, int n, int in, const float *W, const float *b,
                         int out, float *y, int ldy, const float *extra);

// ---- encoder pieces (reference src/fs2encoder.cpp) -------------------------------------------------
hipError_t launch_embed(hipStream_t s, const int32_t *ids, const int32_t *puncts, const float *wemb, int emb,
                        const float *pemb, int pdim, const float *posenc, int n, float *x, int ld);
hipError_t launch_attention(hipStream_t s, const float *q, const float *k, const float *v, int ld, int n, int H,
                            int dk, float inv_temp, float *o, int ldo);
// y = LayerNorm(x + res) * w + b  over the C real channels (res may be null); channels [C, Cp) are zeroed
hipError_t launch_add_layernorm(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int n, int C,
                                int Cp, const float *w, const float *b, float eps, float *y, int ldy);
hipError_t launch_add_rowvec(hipStream_t s, float *x, int ld, int n, int C, const float *v);
// pred[n] = dot(x[n][:], w) + b
hipError_t launch_rowdot(hipStream_t s, const float *x, int ld, int n, int C, const float *w, const float *b, float *y);
// bucket[n] = clamp((int)(pred*(nbins-1) + 0.5), 0, nbins-1); x[n][:] += emb[bucket[n]][:]
hipError_t launch_bucket_embed_add(hipStream_t s, const float *pred, int n, int nbins, const float *emb, int C,
                                   float *x, int ld, int32_t *bucket);
// device length regulator: rounded durations -> exclusive scan -> gather, zero tail; n_frames[0] = frames
hipError_t launch_length_regulator(hipStream_t s, const float *feat, int ld, const float *logdur, int n, int C,
                                   int T, float *hidden, int ldh, int32_t *n_frames);

}  // namespace zv

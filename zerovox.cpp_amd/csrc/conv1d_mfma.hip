// conv1d_mfma.hip — the hot kernel family: fused Conv1d as implicit GEMM on gfx950 matrix cores.
//
// Replaces, per conv, the reference's 8-node ggml pattern
//     im2col(F16) -> mul_mat -> reshape -> cont(transpose) -> repeat(bias) -> add -> cont(transpose)
// (reference src/hifigan.cpp:132-140, ggml/src/ggml.c:3769-3786, ggml-cpu.c:9890-9961,7377-7554) plus
// the element-wise nodes around it (leaky_relu / norm affine / residual add / scale) with ONE launch:
//
//   stage   : a (BM + (K-1)*dil) x ck tile of the input is read once from HBM (coalesced float4 rows of
//             the channels-last layout), run through the prologue, rounded to f16 (RNE, as ggml's
//             im2col does) and parked in LDS.  No im2col matrix ever exists.
//   compute : v_mfma_f32_32x32x16_f16.  M = time, N = output channel, K = (tap, input channel).
//             A fragments are one ds_read_b128 each (8 consecutive channels of one time step; row
//             stride ck*2+16 B makes the 16-lane read groups bank-conflict free); B fragments stream
//             straight from L2 into registers — weights were re-laid-out at load time so that one
//             fragment is 1 KiB contiguous — with a 4-step register prefetch.  Each wave owns a
//             (32*MT) x 32 output tile so one B fragment feeds MT MFMAs.
//   epilogue: bias, residual add, scale, activation, f32 or f16 store (128-B segments per half-wave).
//
// Several independent convs that share a tile configuration (the three MRF branches of a HiFi-GAN
// stage) ride in one launch as "jobs" (blockIdx.z) so that a 512-frame utterance still fills 256 CUs.
#include "kernels.h"
#include "knobs.h"

#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace zv
{

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

static constexpr int CK_MAX = 256;
// cache policy of the fused kernels' output stores (aux operand of the buffer store; 0 = default, 2 = non-temporal)
#ifndef ZV_ST_AUX
#define ZV_ST_AUX 2       // measured on the batch: -1.1 % (256 / 128 channels), -3.5 % (64), -1.3 % (32) against the default policy
#endif
// timing-only ablation build (-DZV_ABL_A): only the first row tile's A fragment is read from LDS, the others copy it
#ifdef ZV_ABL_A
#define ZV_ABL_LD(arr, p, mt) ((mt) == 0 ? *(const half8 *)(p) : arr[0])
#else
#define ZV_ABL_LD(arr, p, mt) (*(const half8 *)((p) + (mt) * 32 * RS))
#endif
// 16-byte pieces a thread keeps in flight while it stages a tile (10 — one round trip for every small tile — measured no
// faster on the batch's upsample convs and costs the MT = 1 kernels a wave of occupancy)
#ifndef ZV_STAGE_U
#define ZV_STAGE_U 4
#endif
// ... in the single-utterance form of the generic kernel (one workgroup per CU at most: occupancy is not the price there): the
// 34-row x 256-channel f32 tile of a chunk in one round trip instead of three
#ifndef ZV_STAGE_US
#define ZV_STAGE_US 9
#endif
// ... and its loader waves (see conv1d_mfma_kernel): waves that only stage — chunk c + 1 into the second LDS tile while the four
// MFMA waves walk chunk c.  They have their own vector-memory counters: the MFMA waves' counted waits on the weight stream never
// queue behind a tile's loads.  0 = the round-3 form (every wave stages, then every wave multiplies).
#ifndef ZV_SINGLE_LW
#define ZV_SINGLE_LW 4
#endif
#ifndef ZV_STAGE_ULW
#define ZV_STAGE_ULW 10
#endif
#ifndef ZV_STAGE_U128
#define ZV_STAGE_U128 12
#endif
#ifndef ZV_STAGE_U256
#define ZV_STAGE_U256 19
#endif
// waves per SIMD the 64 x 64 wave-tile instantiation of the generic conv kernel is compiled for (3: 168 registers)
#ifndef ZV_NT2_OCC
#define ZV_NT2_OCC 3
#endif

int conv_pick_ck(int Cin_p, int ck_max)
{
    if (ck_max <= 0 || ck_max > CK_MAX) ck_max = CK_MAX;
    // full 256-channel chunks (they run on the immediate-address MFMA loop; fewer stage/barrier rounds per conv)
    // + one remainder chunk
    return Cin_p < ck_max ? Cin_p : ck_max;
}

size_t packed_conv_weight_halfs(int Cin_p, int Cout_p, int K)
{
    return (size_t)((Cout_p + 31) / 32) * K * (Cin_p / 16) * 512;
}

// dst[(((nt*K*nicb) + K*(c0/16) + tap*nkc_chunk + kc) * 64 + lane) * 8 + j]
//   = w[oc = nt*32 + (lane&31)][ic = c0 + kc*16 + 8*(lane>>5) + j][tap]      (0 outside IC/OC)
// i.e. the B-operand fragment of v_mfma_f32_32x32x16_f16: lane l holds B[k = 8*(l>>5) + j][col = l&31].
void pack_conv_weight(const uint16_t *w, int K, int IC, int OC, int Cin_p, int Cout_p, int ck, uint16_t *dst)
{
    const int ntiles = (Cout_p + 31) / 32, nicb = Cin_p / 16;
    for (int nt = 0; nt < ntiles; nt++)
        for (int c0 = 0; c0 < Cin_p; c0 += ck)
        {
            const int nkc = ((Cin_p - c0 < ck) ? (Cin_p - c0) : ck) / 16;
            for (int tap = 0; tap < K; tap++)
                for (int kc = 0; kc < nkc; kc++)
                {
                    size_t blk = (size_t)nt * K * nicb + (size_t)K * (c0 / 16) + (size_t)tap * nkc + kc;
                    uint16_t *d = dst + blk * 512;
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++)
                        {
                            int oc = nt * 32 + (lane & 31);
                            int ic = c0 + kc * 16 + 8 * (lane >> 5) + j;
                            d[lane * 8 + j] = (oc < OC && ic < IC) ? w[((size_t)oc * IC + ic) * K + tap] : (uint16_t)0;
                        }
                }
        }
}

__device__ __forceinline__ float lrelu(float x, float s) { return x > 0.f ? x : x * s; }

// ---- stage: HBM -> prologue -> f16 -> LDS.  U independent 16-byte loads per thread are issued before any of
// them is consumed (hipcc otherwise waits vmcnt(0) after every load and the tile fill becomes a chain of
// full HBM latencies).  LDS row r holds input time row_t0 + r; out-of-range rows are zeros.
// what the staging loop reads: the job's input with every pointer already advanced to the workgroup's segment
struct StageSrc
{
    const void  *x0, *x1, *x2;
    const float *pa, *pb, *pstat;
    int          ldx, L;
    float        slope, pscale;
};

template <int U, int PRO, int NTH = 256>
__device__ __forceinline__ void stage_tile_p(const StageSrc &J, char *smem, int RS, int c0, int ck, int row_t0, int rows,
                                             int tid)
{
    const int cols = ck >> 2;
    const int total = rows * cols;
    const int L = J.L;
    constexpr int pro = PRO;
    int r = tid / cols, c4 = tid - r * cols;
    const int dr = NTH / cols, dc = NTH - dr * cols;
    // cols divides NTH (every 256- / 128- / 64- / 32- / 16-channel chunk): all of a thread's pieces are one column group, its
    // per-channel vectors are loaded once (9 pieces x 4 vectors were 36 more loads per thread and chunk)
    const bool hoist = dc == 0;
    float4 hp0 = {0, 0, 0, 0}, hp1 = hp0, hp2 = hp0, hp3 = hp0;
    if (hoist)
    {
        const int c = c0 + c4 * 4;
        if constexpr (pro == PRO_NORM_ACT)
        {
            hp0 = *(const float4 *)(J.pstat + 2 * c);
            hp1 = *(const float4 *)(J.pstat + 2 * c + 4);
            hp2 = *(const float4 *)(J.pa + c);
            hp3 = *(const float4 *)(J.pb + c);
        }
        else if constexpr (pro == PRO_MELNORM)
        {
            hp2 = *(const float4 *)(J.pa + c);
            hp3 = *(const float4 *)(J.pb + c);
        }
    }
    for (int base = tid; base < total; base += NTH * U)
    {
        float4 v[U], v1[U], v2[U];
        half4 hraw[U];
        int lofs[U];
        bool live[U], inr[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            live[u] = base + u * NTH < total;
            const int t = row_t0 + r;
            inr[u] = live[u] && t >= 0 && t < L;
            lofs[u] = r * RS + c4 * 8;
            const size_t off = (size_t)(inr[u] ? t : 0) * J.ldx + c0 + c4 * 4;
            if constexpr (pro == PRO_RAW_F16)
                hraw[u] = *(const half4 *)((const _Float16 *)J.x0 + off);
            else
            {
                v[u] = *(const float4 *)((const float *)J.x0 + off);
                if constexpr (pro == PRO_SUM3_ACT)
                {
                    v1[u] = *(const float4 *)((const float *)J.x1 + off);
                    v2[u] = *(const float4 *)((const float *)J.x2 + off);
                }
            }
            r += dr;
            c4 += dc;
            if (c4 >= cols) { c4 -= cols; r++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            if (!live[u]) continue;
            half4 h = {0, 0, 0, 0};
            if (inr[u])
            {
                if constexpr (pro == PRO_RAW_F16)
                    h = hraw[u];
                else
                {
                    float4 x = v[u];
                    const int c = c0 + ((lofs[u] % RS) >> 1);
                    (void)c;
                    if constexpr (pro == PRO_SUM3_ACT)
                    {
                        const float sc = J.pscale;
                        x.x = ((x.x + v1[u].x) + v2[u].x) * sc;
                        x.y = ((x.y + v1[u].y) + v2[u].y) * sc;
                        x.z = ((x.z + v1[u].z) + v2[u].z) * sc;
                        x.w = ((x.w + v1[u].w) + v2[u].w) * sc;
                    }
                    else if constexpr (pro == PRO_SCALE_ACT)
                    {
                        const float sc = J.pscale;
                        x.x = x.x * sc;
                        x.y = x.y * sc;
                        x.z = x.z * sc;
                        x.w = x.w * sc;
                    }
                    else if constexpr (pro == PRO_NORM_ACT)
                    {
                        float4 st0 = hp0, st1 = hp1, g = hp2, b = hp3;               // st: mean,rstd,mean,rstd
                        if (!hoist)
                        {
                            st0 = *(const float4 *)(J.pstat + 2 * c);
                            st1 = *(const float4 *)(J.pstat + 2 * c + 4);
                            g = *(const float4 *)(J.pa + c);
                            b = *(const float4 *)(J.pb + c);
                        }
                        x.x = ((x.x - st0.x) * st0.y) * g.x + b.x;
                        x.y = ((x.y - st0.z) * st0.w) * g.y + b.y;
                        x.z = ((x.z - st1.x) * st1.y) * g.z + b.z;
                        x.w = ((x.w - st1.z) * st1.w) * g.w + b.w;
                    }
                    else if constexpr (pro == PRO_MELNORM)
                    {
                        float4 a = hp2, b = hp3;
                        if (!hoist)
                        {
                            a = *(const float4 *)(J.pa + c);
                            b = *(const float4 *)(J.pb + c);
                        }
                        x.x = (x.x - a.x) / b.x;
                        x.y = (x.y - a.y) / b.y;
                        x.z = (x.z - a.z) / b.z;
                        x.w = (x.w - a.w) / b.w;
                    }
                    if constexpr (pro != PRO_MELNORM)
                    {
                        const float sl = J.slope;
                        x.x = lrelu(x.x, sl);
                        x.y = lrelu(x.y, sl);
                        x.z = lrelu(x.z, sl);
                        x.w = lrelu(x.w, sl);
                    }
                    h[0] = (_Float16)x.x;      // v_cvt_f16_f32: round-to-nearest-even, like _cvtss_sh(x, 0)
                    h[1] = (_Float16)x.y;
                    h[2] = (_Float16)x.z;
                    h[3] = (_Float16)x.w;
                }
            }
            *(half4 *)(smem + lofs[u]) = h;
        }
    }
}

// The same fill in two halves for the loader waves of the single-utterance kernel: stage_load_p requests a whole tile (at most
// NTH * U pieces) into registers, stage_store_p applies the prologue and writes LDS — a barrier may sit between the two.  Same
// operations per element as stage_tile_p.
template <int U>
struct StageRegs
{
    float4 v[U];
    float4 p0, p1, p2, p3;      // the thread's per-channel vectors (when all its pieces are one column group)
};
template <int U, int PRO, int NTH>
__device__ __forceinline__ void stage_load_p(const StageSrc &J, int c0, int ck, int row_t0, int rows, int tid, StageRegs<U> &R)
{
    // (an f16 operand tensor travels in 16-byte pieces of 8 channels, everything else in pieces of 4 channels)
    constexpr int PW = PRO == PRO_RAW_F16 ? 8 : 4;
    const int cols = ck / PW, total = rows * cols, L = J.L;
    int r = tid / cols, c4 = tid - r * cols;
    const int dr = NTH / cols, dc = NTH - dr * cols;
    if (dc == 0)
    {
        const int c = c0 + c4 * 4;
        if constexpr (PRO == PRO_NORM_ACT)
        {
            R.p0 = *(const float4 *)(J.pstat + 2 * c);
            R.p1 = *(const float4 *)(J.pstat + 2 * c + 4);
        }
        if constexpr (PRO == PRO_NORM_ACT || PRO == PRO_MELNORM)
        {
            R.p2 = *(const float4 *)(J.pa + c);
            R.p3 = *(const float4 *)(J.pb + c);
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
    {
        const int t = row_t0 + r;
        const bool inr = tid + u * NTH < total && t >= 0 && t < L;
        const size_t off = (size_t)(inr ? t : 0) * J.ldx + c0 + c4 * PW;
        if constexpr (PRO == PRO_RAW_F16)
            R.v[u] = *(const float4 *)((const _Float16 *)J.x0 + off);
        else
            R.v[u] = *(const float4 *)((const float *)J.x0 + off);
        r += dr;
        c4 += dc;
        if (c4 >= cols) { c4 -= cols; r++; }
    }
}
template <int U, int PRO, int NTH>
__device__ __forceinline__ void stage_store_p(const StageSrc &J, char *smem, int RS, int c0, int ck, int row_t0, int rows, int tid,
                                              const StageRegs<U> &R)
{
    constexpr int PW = PRO == PRO_RAW_F16 ? 8 : 4;
    const int cols = ck / PW, total = rows * cols, L = J.L;
    int r = tid / cols, c4 = tid - r * cols;
    const int dr = NTH / cols, dc = NTH - dr * cols;
    const bool hoist = dc == 0;
    const float4 hp0 = R.p0, hp1 = R.p1, hp2 = R.p2, hp3 = R.p3;
#pragma unroll
    for (int u = 0; u < U; u++)
    {
        const int t = row_t0 + r;
        const bool live = tid + u * NTH < total;
        const bool inr = live && t >= 0 && t < L;
        const int lofs = r * RS + c4 * 2 * PW;
        if constexpr (PRO == PRO_RAW_F16)
        {
            if (live) *(float4 *)(smem + lofs) = inr ? R.v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        else if (live)
        {
            half4 h = {0, 0, 0, 0};
            if (inr)
            {
                if constexpr (PRO == PRO_RAW_F16)
                {
                }
                else
                {
                    float4 x = R.v[u];
                    const int c = c0 + c4 * 4;
                    (void)c;
                    if constexpr (PRO == PRO_SCALE_ACT)
                    {
                        const float sc = J.pscale;
                        x.x = x.x * sc;
                        x.y = x.y * sc;
                        x.z = x.z * sc;
                        x.w = x.w * sc;
                    }
                    else if constexpr (PRO == PRO_NORM_ACT)
                    {
                        float4 st0 = hp0, st1 = hp1, g = hp2, b = hp3;               // st: mean,rstd,mean,rstd
                        if (!hoist)
                        {
                            st0 = *(const float4 *)(J.pstat + 2 * c);
                            st1 = *(const float4 *)(J.pstat + 2 * c + 4);
                            g = *(const float4 *)(J.pa + c);
                            b = *(const float4 *)(J.pb + c);
                        }
                        x.x = ((x.x - st0.x) * st0.y) * g.x + b.x;
                        x.y = ((x.y - st0.z) * st0.w) * g.y + b.y;
                        x.z = ((x.z - st1.x) * st1.y) * g.z + b.z;
                        x.w = ((x.w - st1.z) * st1.w) * g.w + b.w;
                    }
                    else if constexpr (PRO == PRO_MELNORM)
                    {
                        float4 a = hp2, b = hp3;
                        if (!hoist)
                        {
                            a = *(const float4 *)(J.pa + c);
                            b = *(const float4 *)(J.pb + c);
                        }
                        x.x = (x.x - a.x) / b.x;
                        x.y = (x.y - a.y) / b.y;
                        x.z = (x.z - a.z) / b.z;
                        x.w = (x.w - a.w) / b.w;
                    }
                    if constexpr (PRO != PRO_MELNORM)
                    {
                        const float sl = J.slope;
                        x.x = lrelu(x.x, sl);
                        x.y = lrelu(x.y, sl);
                        x.z = lrelu(x.z, sl);
                        x.w = lrelu(x.w, sl);
                    }
                    h[0] = (_Float16)x.x;
                    h[1] = (_Float16)x.y;
                    h[2] = (_Float16)x.z;
                    h[3] = (_Float16)x.w;
                }
            }
            *(half4 *)(smem + lofs) = h;
        }
        r += dr;
        c4 += dc;
        if (c4 >= cols) { c4 -= cols; r++; }
    }
}
#define ZV_STAGE_SPLIT_SWITCH(pro, CALL)                       \
    switch (pro)                                               \
    {                                                          \
        case PRO_RAW_F16: CALL(PRO_RAW_F16); break;            \
        case PRO_ACT: CALL(PRO_ACT); break;                    \
        case PRO_NORM_ACT: CALL(PRO_NORM_ACT); break;          \
        case PRO_MELNORM: CALL(PRO_MELNORM); break;            \
        default: CALL(PRO_SCALE_ACT); break;                   \
    }

template <int U, int NTH = 256>
__device__ __forceinline__ void stage_tile(int pro, const StageSrc &J, char *smem, int RS, int c0, int ck, int row_t0, int rows,
                                           int tid)
{
    switch (pro)        // wave-uniform; each case is a straight-line batched fill
    {
        case PRO_RAW_F16: stage_tile_p<U, PRO_RAW_F16, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;
        case PRO_ACT: stage_tile_p<U, PRO_ACT, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;
        case PRO_NORM_ACT: stage_tile_p<U, PRO_NORM_ACT, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;
        case PRO_MELNORM: stage_tile_p<U, PRO_MELNORM, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;
        case PRO_SCALE_ACT: stage_tile_p<U, PRO_SCALE_ACT, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;
        default: stage_tile_p<(U > 4 ? 4 : U), PRO_SUM3_ACT, NTH>(J, smem, RS, c0, ck, row_t0, rows, tid); break;   // three tensors per piece
    }
}

// PRO_RAW_F16 in 16-byte pieces (8 channels) for NTH threads: every load of a round is in flight before the first is
// stored (U x 16 B per thread), so a tile of <= NTH * U pieces costs one round trip.
template <int U, int NTH>
__device__ __forceinline__ void stage_raw16(const StageSrc &J, char *smem, int RS, int c0, int ck, int row_t0, int rows, int tid)
{
    const int cols = ck >> 3;
    const int total = rows * cols;
    const int L = J.L;
    int r = tid / cols, c8 = tid - r * cols;
    const int dr = NTH / cols, dc = NTH - dr * cols;
    for (int base = tid; base < total; base += NTH * U)
    {
        uint4 v[U];
        int lofs[U];
        bool live[U], inr[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            live[u] = base + u * NTH < total;
            const int t = row_t0 + r;
            inr[u] = live[u] && t >= 0 && t < L;
            lofs[u] = r * RS + c8 * 16;
            v[u] = *(const uint4 *)((const _Float16 *)J.x0 + (size_t)(inr[u] ? t : 0) * J.ldx + c0 + c8 * 8);
            r += dr;
            c8 += dc;
            if (c8 >= cols) { c8 -= cols; r++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (live[u]) *(uint4 *)(smem + lofs[u]) = inr[u] ? v[u] : make_uint4(0, 0, 0, 0);
    }
}

// ---- compute: S = K * nkc MFMA steps over one staged chunk.  A fragments are double-buffered in registers
// (the ds_reads of step s+1 are in flight while the MFMAs of step s run), B fragments come from L2 through a
// 4-deep register ring.
template <int MT, int NT>
__device__ __forceinline__ void mfma_chunk(floatx16 (&acc)[MT][NT], const char *abase, int RS, int dil, const half8 *wp,
                                           size_t wseg, int K, int nkc)
{
    // Branch-free, 4 steps per iteration with static register slots so that hipcc can count its waits: the B
    // fragment consumed in slot u was requested four steps earlier (s_waitcnt vmcnt(3)), the A fragments one step
    // earlier.  Steps S..round_up(S,4)-1 do not exist: they run with B = 0 (adds nothing) on a clamped A address.
    const int S = K * nkc;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 b[4][NT];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[u][nt] = wp[nt * wseg + (size_t)((u < S) ? u : S - 1) * 64];
    half8 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a[mt] = *(const half8 *)(abase + mt * 32 * RS);
    int tap = 0, kc = 0;
    for (int s0 = 0; s0 < S; s0 += 4)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            if (++kc == nkc) { kc = 0; tap++; }
            if (tap >= K) tap = 0;                   // past the last step: any valid address
            const char *ap = abase + tap * dil * RS + kc * 32;
            half8 an[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) an[mt] = *(const half8 *)(ap + mt * 32 * RS);
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
            {
                const half8 bu = (s0 + u < S) ? b[u][nt] : zero8;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bu, acc[mt][nt], 0, 0, 0);
            }
            const int sn = s0 + u + 4;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b[u][nt] = wp[nt * wseg + (size_t)((sn < S) ? sn : S - 1) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) a[mt] = an[mt];
        }
    }
}

// ZERO: the first step of a contraction — the accumulator operand is the constant 0 (an inline constant of the
// instruction: no 16 x MT x NT register moves to clear the accumulators first)
template <int MT, int NT, bool SWAP, bool ZERO = false>
__device__ __forceinline__ void mfma_step(floatx16 (&acc)[MT][NT], const half8 (&a)[MT], const half8 (&b)[NT])
{
    const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
        {
            if constexpr (SWAP)      // weights as the A operand -> D[oc][time]
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[nt], a[mt], ZERO ? z : acc[mt][nt], 0, 0, 0);
            else
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], ZERO ? z : acc[mt][nt], 0, 0, 0);
        }
}

// One body = 8 MFMA steps = 8 / (CP/16) taps (half a tap for CP = 256).  Two static B register sets ping-pong
// (b0: steps 0-3, b1: steps 4-7), each refilled for the next body right after its last use; A fragments ping-pong
// one step ahead.  All LDS and weight addresses inside a body are immediates off the body's base.  A wave covers
// NT output tiles of 32 channels (their weight segments are `wseg` half8 apart) and MT row tiles.
template <int CP, int MT, int NT, bool SWAP>
__device__ __forceinline__ void mfma_taps(floatx16 (&acc)[MT][NT], const char *ap, int dilRS, const half8 *wq, size_t wseg, int K)
{
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    constexpr int TPB = (NKC >= 8) ? 1 : 8 / NKC;        // taps per body: 4 / 2 / 1 / (1/2) for CP = 32 / 64 / 128 / 256
    constexpr bool HALF = NKC == 16;                     // CP = 256: a tap is two bodies (channels 0-127, 128-255)
    const int nsb = (K * NKC + 3) >> 2;                  // 4-step sub-blocks (the last one may run partly on zero weights)
    const int nb = nsb >> 1;
    half8 b0[4][NT], b1[4][NT];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b0[u][nt] = wq[nt * wseg + u * 64];
    wq += 4 * 64;
    half8 a0[MT], a1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a0[mt] = *(const half8 *)(ap + mt * 32 * RS);

#define ZV_A_ADDR(un) ((un) == 8 ? apn : tb[((un) / NKC) % 4] + ((un) % NKC) * 32)
#define ZV_LOAD_A(dst, un)                                                                   \
    {                                                                                        \
        const char *np_ = ZV_A_ADDR(un);                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) dst[mt] = *(const half8 *)(np_ + mt * 32 * RS); \
    }
    for (int ib = 0; ib < nb; ib++)
    {
        const char *tb[4];
        tb[0] = ap;
#pragma unroll
        for (int x = 1; x < 4; x++) tb[x] = (x < TPB) ? ap + x * dilRS : ap;
        const char *apn = HALF ? ((ib & 1) ? ap + (dilRS - 256) : ap + 256) : ap + TPB * dilRS;   // next body
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b1[u][nt] = wq[nt * wseg + u * 64];      // steps 4..7 of this body
        __builtin_amdgcn_sched_barrier(0);      // keep the requests here: hipcc otherwise sinks them next to their use
        ZV_LOAD_A(a1, 1) mfma_step<MT, NT, SWAP>(acc, a0, b0[0]);
        ZV_LOAD_A(a0, 2) mfma_step<MT, NT, SWAP>(acc, a1, b0[1]);
        ZV_LOAD_A(a1, 3) mfma_step<MT, NT, SWAP>(acc, a0, b0[2]);
        ZV_LOAD_A(a0, 4) mfma_step<MT, NT, SWAP>(acc, a1, b0[3]);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b0[u][nt] = wq[nt * wseg + (4 + u) * 64];   // steps 0..3 of the next body
        __builtin_amdgcn_sched_barrier(0);
        ZV_LOAD_A(a1, 5) mfma_step<MT, NT, SWAP>(acc, a0, b1[0]);
        ZV_LOAD_A(a0, 6) mfma_step<MT, NT, SWAP>(acc, a1, b1[1]);
        ZV_LOAD_A(a1, 7) mfma_step<MT, NT, SWAP>(acc, a0, b1[2]);
        ZV_LOAD_A(a0, 8) mfma_step<MT, NT, SWAP>(acc, a1, b1[3]);
        ap = apn;
        wq += 8 * 64;
    }
    if (nsb & 1)                                 // odd sub-block count (CP = 64): one more tap on b0
    {
        const char *tb[4] = {ap, ap, ap, ap};
        const char *apn = ap;
        (void)apn;
        ZV_LOAD_A(a1, 1) mfma_step<MT, NT, SWAP>(acc, a0, b0[0]);
        ZV_LOAD_A(a0, 2) mfma_step<MT, NT, SWAP>(acc, a1, b0[1]);
        ZV_LOAD_A(a1, 3) mfma_step<MT, NT, SWAP>(acc, a0, b0[2]);
        mfma_step<MT, NT, SWAP>(acc, a1, b0[3]);
    }
#undef ZV_LOAD_A
#undef ZV_A_ADDR
}


// mfma_taps with ONE set of four weight-fragment slots: the slot a step has consumed is refilled at once with the fragment
// of four steps later (same prefetch distance as the two ping-pong sets above, half their registers: the 64 x 64 wave tile
// then fits 168 registers = three workgroups per CU instead of two).  Same step order, same bits.
template <int CP, int MT, int NT>
__device__ __forceinline__ void mfma_taps_ring4(floatx16 (&acc)[MT][NT], const char *ap, int dilRS, const half8 *wq, size_t wseg, int K)
{
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    static_assert(NKC >= 8, "whole 8-step bodies per tap");
    constexpr bool HALF = NKC == 16;
    const int nb = (K * NKC) >> 3;
    half8 b[4][NT];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[u][nt] = wq[nt * wseg + u * 64];
    half8 a0[MT], a1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a0[mt] = *(const half8 *)(ap + mt * 32 * RS);
#define ZV_R4_LOADA(dst, un)                                                                  \
    {                                                                                         \
        const char *np_ = (un) == 8 ? apn : ap + (un) * 32;                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) dst[mt] = *(const half8 *)(np_ + mt * 32 * RS); \
    }
#define ZV_R4_STEP(u, acur, anext)                                                            \
    ZV_R4_LOADA(anext, (u) + 1)                                                               \
    mfma_step<MT, NT, false>(acc, acur, b[(u) & 3]);                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; nt++) b[(u) & 3][nt] = wq[nt * wseg + ((u) + 4) * 64]; \
    __builtin_amdgcn_sched_barrier(0);
    for (int ib = 0; ib < nb; ib++)
    {
        const char *apn = HALF ? ((ib & 1) ? ap + (dilRS - 256) : ap + 256) : ap + dilRS;   // next body
        ZV_R4_STEP(0, a0, a1) ZV_R4_STEP(1, a1, a0) ZV_R4_STEP(2, a0, a1) ZV_R4_STEP(3, a1, a0)
        ZV_R4_STEP(4, a0, a1) ZV_R4_STEP(5, a1, a0) ZV_R4_STEP(6, a0, a1) ZV_R4_STEP(7, a1, a0)
        ap = apn;
        wq += 8 * 64;
    }
#undef ZV_R4_STEP
#undef ZV_R4_LOADA
}

// The same loop with the A fragments TWO steps ahead (four register sets in rotation) and a scheduling fence after
// every step.  In the loop above hipcc moves each A read down next to the MFMA that consumes it (an lgkmcnt wait right
// behind the read: LDS latency exposed on every step); a fence per step pins the reads where they are written, and two
// steps (>= 4 MFMAs at MT = 2) cover the ds_read_b128 latency.  Used by the fused ResBlock kernels, which have the
// registers to spare; in the generic kernel's widest instantiations the fences cost more registers than they gain.
// The accumulators need no clearing before the call: the first step starts them from the constant 0.  The contraction is
// walked in bodies of 8 steps; CP = 64 with K = 3 mod 4 taps leaves half a body, every other supported shape a whole
// number (pair_shape_ok).
// the first four weight fragments of a contraction: requested by the caller ahead of a phase that does not need them (the
// staging wait, the xt pack) so that the MFMA loop does not start with an exposed L2 round trip
template <int NT>
__device__ __forceinline__ void deep_preload_b(half8 (&b0)[4][NT], const half8 *wq, size_t wseg)
{
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b0[u][nt] = wq[nt * wseg + u * 64];
}

template <int CP, int MT, int NT, bool SWAP, int DEPTH = 2>
__device__ __forceinline__ void mfma_taps_deep(floatx16 (&acc)[MT][NT], const char *ap, int dilRS, const half8 *wq, size_t wseg, int K,
                                               half8 (&b0)[4][NT], int wstep = 8 * 64)
{
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    constexpr int TPB = (NKC >= 8) ? 1 : 8 / NKC;
    constexpr bool HALF = NKC == 16;
    const int nsb = (K * NKC + 3) >> 2;
    const int nb = nsb >> 1;
    half8 b1[4][NT];
    wq += 4 * 64;
    static_assert(DEPTH >= 1 && DEPTH <= 7, "A fragments travel DEPTH steps ahead through a ring of 8 register sets");
    half8 a[8][MT];
#define ZV_UN8(un) ((un) >= 8 ? (un) - 8 : 0)
#define ZV_A_ADDR2(un) ((un) >= 8 ? tbn[(ZV_UN8(un) / NKC) % 4] + (ZV_UN8(un) % NKC) * 32 : tb[((un) / NKC) % 4] + ((un) % NKC) * 32)
#define ZV_LOAD_A2(un)                                                                        \
    {                                                                                         \
        const char *np_ = ZV_A_ADDR2(un);                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) a[(un) % 8][mt] = ZV_ABL_LD(a[(un) % 8], np_, mt); \
    }
#define ZV_STEP2(u, bset) ZV_LOAD_A2((u) + DEPTH) mfma_step<MT, NT, SWAP>(acc, a[(u) % 8], bset[(u) % 4]); __builtin_amdgcn_sched_barrier(0);
#define ZV_STEP2Z(u, bset) ZV_LOAD_A2((u) + DEPTH) mfma_step<MT, NT, SWAP, true>(acc, a[(u) % 8], bset[(u) % 4]); __builtin_amdgcn_sched_barrier(0);
#define ZV_BODY(FIRSTSTEP)                                                                                       \
    {                                                                                                            \
        const char *tb[4], *tbn[4];                                                                              \
        tb[0] = ap;                                                                                              \
        _Pragma("unroll") for (int x = 1; x < 4; x++) tb[x] = (x < TPB) ? ap + x * dilRS : ap;                   \
        const char *apn = HALF ? ((ib & 1) ? ap + (dilRS - 256) : ap + 256) : ap + TPB * dilRS; /* next body */  \
        tbn[0] = apn;                                                                                            \
        _Pragma("unroll") for (int x = 1; x < 4; x++) tbn[x] = (x < TPB) ? apn + x * dilRS : apn;                \
        _Pragma("unroll") for (int u = 0; u < 4; u++)                                                            \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) b1[u][nt] = wq[nt * wseg + u * 64];                \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        FIRSTSTEP(0, b0) ZV_STEP2(1, b0) ZV_STEP2(2, b0) ZV_STEP2(3, b0)                                         \
        _Pragma("unroll") for (int u = 0; u < 4; u++)                                                            \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) b0[u][nt] = wq[nt * wseg + (4 + u) * 64];          \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        ZV_STEP2(4, b1) ZV_STEP2(5, b1) ZV_STEP2(6, b1) ZV_STEP2(7, b1)                                          \
        ap = apn;                                                                                                \
        wq += wstep;                                                                                             \
    }
    {
        const char *tb[4], *tbn[4];
        tb[0] = ap;
#pragma unroll
        for (int x = 1; x < 4; x++) tb[x] = (x < TPB) ? ap + x * dilRS : ap;
#pragma unroll
        for (int x = 0; x < 4; x++) tbn[x] = ap;
        ZV_LOAD_A2(0)
        if constexpr (DEPTH > 1) ZV_LOAD_A2(1)
        if constexpr (DEPTH > 2) ZV_LOAD_A2(2)
        if constexpr (DEPTH > 3) ZV_LOAD_A2(3)
        if constexpr (DEPTH > 4) ZV_LOAD_A2(4)
        if constexpr (DEPTH > 5) ZV_LOAD_A2(5)
        if constexpr (DEPTH > 6) ZV_LOAD_A2(6)
    }
    {
        const int ib = 0;
        ZV_BODY(ZV_STEP2Z)
    }
    for (int ib = 1; ib < nb; ib++) ZV_BODY(ZV_STEP2)
    if constexpr (CP == 64)
        if (nsb & 1)                             // odd sub-block count (K = 3 mod 4 taps): one more tap on b0
        {
            const char *tb[4] = {ap, ap, ap, ap};
            const char *tbn[4] = {ap, ap, ap, ap};
            (void)tbn;
            ZV_STEP2(0, b0) ZV_STEP2(1, b0) ZV_STEP2(2, b0) ZV_STEP2(3, b0)
        }
#undef ZV_BODY
#undef ZV_STEP2Z
#undef ZV_STEP2
#undef ZV_LOAD_A2
#undef ZV_A_ADDR2
#undef ZV_UN8
}

// The MFMA loop of a full 256-channel chunk for single-utterance launches (MT row tiles x ONE output tile per wave, one
// wave per SIMD, at most a round of workgroups: registers are free, latency is everything).  A step is MT MFMAs — 32
// cycles at MT = 1 — so the loops above, whose A fragment is requested one step ahead (an LDS round trip per step) and
// whose weight fragments 4-8 steps ahead (an L2 round trip per 4 steps), take 75 ns per step (phase stamps: 3.6 us per
// chunk for 0.8 us of matrix work).  Here one body = one tap = 16 steps, every address an immediate off the body's base,
// the A fragments travel 6 steps ahead through a ring of 8 register sets and the weight fragments 16 steps ahead through
// a ring of 16 that is carried from chunk to chunk (the generic layout puts the next chunk's first tap right behind this
// chunk's last: the caller preloads the ring once, before the first tile is even staged; the last tap's requests run up
// to 16 KiB past the chunk — load_conv / load_upsample allocate that slack).  Same step order as mfma_taps<256> /
// mfma_chunk — tap-major, 16 channels per step — hence the same bits.  (Measured: 2.8 us per chunk, 1.7 us where the
// weights hit L2; a conv of 33 us becomes 29.6 us.  What is left is the cold weights — every row tile of a channel group
// misses on them together — and the staging round trips between the loops.)
template <int MT>
__device__ __forceinline__ void preload_ring16(half8 (&b)[16], const half8 *wq)
{
#pragma unroll
    for (int u = 0; u < 16; u++) b[u] = wq[u * 64];
}

template <int MT>
__device__ __forceinline__ void mfma_taps_single256(floatx16 (&acc)[MT][1], const char *ap, int dilRS, const half8 *wq, int K,
                                                    half8 (&b)[16])
{
    constexpr int RS = 256 * 2 + 16;
    half8 a[8][MT];
#define ZV_SA(un) (((un) >= 16 ? apn : ap) + ((un) & 15) * 32)
#define ZV_SLOAD(un)                                                                              \
    {                                                                                             \
        const char *np_ = ZV_SA(un);                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) a[(un) & 7][mt] = *(const half8 *)(np_ + mt * 32 * RS); \
    }
#define ZV_SSTEP(u)                                                                               \
    {                                                                                             \
        half8 ac_[MT];                                                                            \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) ac_[mt] = a[(u) & 7][mt];               \
        ZV_SLOAD((u) + 6)                                                                         \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                         \
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac_[mt], b[u], acc[mt][0], 0, 0, 0); \
        b[u] = wq[(16 + (u)) * 64];                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
    }
    {
        const char *apn = ap;
        ZV_SLOAD(0) ZV_SLOAD(1) ZV_SLOAD(2) ZV_SLOAD(3) ZV_SLOAD(4) ZV_SLOAD(5)
    }
    for (int tap = 0; tap < K; tap++)
    {
        const char *apn = ap + dilRS;
        ZV_SSTEP(0) ZV_SSTEP(1) ZV_SSTEP(2) ZV_SSTEP(3) ZV_SSTEP(4) ZV_SSTEP(5) ZV_SSTEP(6) ZV_SSTEP(7)
        ZV_SSTEP(8) ZV_SSTEP(9) ZV_SSTEP(10) ZV_SSTEP(11) ZV_SSTEP(12) ZV_SSTEP(13) ZV_SSTEP(14) ZV_SSTEP(15)
        ap = apn;
        wq += 16 * 64;
    }
#undef ZV_SSTEP
#undef ZV_SLOAD
#undef ZV_SA
}

// ---- diagnostic build only (-DZV_STAMPS, never the shipped library): wave 0 of a workgroup of the fused pair kernel
// stamps the clock at its phase boundaries into a buffer of its own (cdna_hip_programming.md §7, in-kernel stamps)
#ifdef ZV_STAMPS
constexpr int ZV_STAMP_WGS = 1 << 17, ZV_STAMP_N = 12;
__device__ unsigned long long zv_stamp_buf[(size_t)ZV_STAMP_WGS * ZV_STAMP_N];
#ifdef ZV_STAMPS_LOADER
#define ZV_STAMP_TID 256
#else
#define ZV_STAMP_TID 0
#endif
#define ZV_STAMP(k)                                                                                   \
    if (jobs.stamp && threadIdx.x == ZV_STAMP_TID && stamp_wg < ZV_STAMP_WGS)                         \
    {                                                                                                 \
        zv_stamp_buf[(size_t)stamp_wg * ZV_STAMP_N + (k)] = __builtin_amdgcn_s_memrealtime();         \
    }
#else
#define ZV_STAMP(k)
#endif

// f64 partial sums of a wave's 32 x 32 output tile per channel (= lane & 31): the lane's 16 rows in register order, then
// the two half-waves (rows 4*(lane>>5) + ...); rows at or past L do not count.  See launch_stats_finalize.
__device__ __forceinline__ void tile_stats_store(const float (&v)[16], int t_first, int L, double *dst)
{
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        const int t = t_first + (r & 3) + 8 * (r >> 2);
        const double x = (t < L) ? (double)v[r] : 0.0;
        s1 += x;
        s2 += x * x;
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if ((threadIdx.x & 63) < 32) *(double2 *)dst = make_double2(s1, s2);
}

// Each wave owns (32*MT) rows x (32*NT) output channels: NT = 2 halves the LDS reads per MFMA (an A fragment feeds two
// MFMAs) and lets a workgroup cover 256 output channels, so a wide conv stages its input half as often.
// (Measured dead end, round 2: the staged tile double-buffered in LDS and filled by LDS-DMA while the MFMA loop of the
// previous chunk runs.  hipcc answers an LDS-DMA in flight with vmcnt(0) waits on the B-fragment stream of the MFMA
// loop — the counted waits that keep eight fragments in flight are gone — and the wide decoder convs ran 3 % slower.)
// (Measured dead end, round 3: two extra "loader" waves staging chunk c + 1 into a second LDS tile under the MFMA loop of chunk c:
// decoder convs 311 -> 378 us; removed in round 4.)
// Single-utterance convs: every row tile of a channel group walks the same weight stream in step, so each fragment is an L2 miss
// for all of them together (a memory-side round trip per ring refill).  The group's row tiles sit on ONE XCD (see the kernels), so they
// warm its L2 together first: workgroup `part` of `nparts` touches its slice of the group's weight bytes, one 4-byte load per
// 128-byte line (64 lines = 8 KiB per wave instruction, 256 bytes returned).  The result is discarded; nothing waits for it.
__device__ __forceinline__ void l2_warm(const void *base, size_t bytes, int part, int nparts, int tid, int nth)
{
    const size_t lines = (bytes + 127) >> 7;
    const size_t per = (lines + nparts - 1) / nparts, l0 = (size_t)part * per;
    const size_t l1 = l0 + per < lines ? l0 + per : lines;
    unsigned sink = 0;
    for (size_t l = l0 + tid; l < l1; l += nth) sink ^= *(const volatile unsigned *)((const char *)base + (l << 7));
    asm volatile("" ::"v"(sink));
}

template <int MT, int WN, int NT, bool SINGLE = false>
__global__ __launch_bounds__(SINGLE ? 256 + 64 * ZV_SINGLE_LW : 256, (NT == 2 && ZV_NT2_OCC == 3) ? 3 : 2) void conv1d_mfma_kernel(const ConvJobs jobs)
{
    constexpr int WM = 4 / WN;
    constexpr int LW = SINGLE ? ZV_SINGLE_LW : 0;          // loader waves (waves 4 .. 4 + LW - 1)
    constexpr int BM = 32 * MT * WM;
    const ConvJob &J = jobs.j[blockIdx.z];
    // workgroup -> (row tile bx, channel group by of ny).  Single-utterance launches deal the channel groups over the XCDs (the
    // hardware hands workgroup i to XCD i % 8): all row tiles of a channel group run on ONE XCD, whose L2 then holds that group's
    // weight fragments (a 1 056 x 1 056 x 3 conv's 6.7 MB do not fit one XCD's 4 MB; with row tiles dealt over the XCDs every XCD
    // streamed all of them from the memory side: scripts/frag_stream_bw.hip, 7-10 TB/s over the chip)
    int bx = blockIdx.x, by = blockIdx.y, ny = gridDim.y;
    if constexpr (SINGLE)
        if (jobs.xcd_ny)
        {
            const int q = blockIdx.x >> 3, g = q / jobs.xcd_nx;
            by = (blockIdx.x & 7) + 8 * g;
            bx = q - g * jobs.xcd_nx;
            ny = jobs.xcd_ny;
            if (by >= ny) return;
        }

    // workgroup -> (segment, row tile inside the segment)
    const int useg = bx / jobs.tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int m0 = (bx - useg * jobs.tps) * BM;
    if (m0 >= L) return;
    const size_t row0 = (size_t)sg.row0 * jobs.rate;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int K = J.K, dil = J.dil, Cin_p = J.Cin_p, Cout_p = J.Cout_p;
    const int nicb = Cin_p >> 4;
    const int ntiles = (Cout_p + 31) >> 5;
    // the output tiles are dealt evenly over the gridDim.y channel groups (33 tiles over 5 groups: 7 7 7 6 6, not 8 8 8 8 1 —
    // every workgroup stages its input tile for every chunk, however few of its waves have work)
    const int nt_span = ntiles - jobs.nt_begin;     // (tiles before nt_begin belong to conv_gemm_kernel)
    const int gt0 = jobs.nt_begin + (int)((long)by * nt_span / ny), gt1 = jobs.nt_begin + (int)((long)(by + 1) * nt_span / ny);
    const int nt0 = gt0 + wn * NT;
    const bool n_ok = nt0 < gt1;
    // a wave whose second tile does not exist computes the tile before it twice and stores it once
    const int ntl = nt0 + NT <= gt1 ? nt0 : (gt1 - NT > 0 ? gt1 - NT : 0);
    const int rows = BM + (K - 1) * dil;
    const int RS = J.ck * 2 + 16;            // LDS row stride in bytes
    if constexpr (SINGLE)
        if (jobs.xcd_ny && jobs.warm)
            l2_warm((const char *)J.w + (size_t)gt0 * K * nicb * 1024, (size_t)(gt1 - gt0) * K * nicb * 1024, bx, jobs.xcd_nx, tid, 256 + 64 * LW);

    StageSrc S;
    {
        const size_t xo = row0 * J.ldx * (J.pro == PRO_RAW_F16 ? 2 : 4);
        S.x0 = (const char *)J.x0 + xo;
        S.x1 = J.x1 ? (const char *)J.x1 + xo : nullptr;
        S.x2 = J.x2 ? (const char *)J.x2 + xo : nullptr;
        S.pa = J.pa ? J.pa + (size_t)useg * J.pab_seg : nullptr;
        S.pb = J.pb ? J.pb + (size_t)useg * J.pab_seg : nullptr;
        S.pstat = J.pstat ? J.pstat + (size_t)useg * J.pstat_seg : nullptr;
        S.ldx = J.ldx;
        S.L = L;
        S.slope = J.slope;
        S.pscale = J.pscale;
    }

    floatx16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][n][r] = 0.f;

    const char *abase = smem + (wm * 32 * MT + (lane & 31)) * RS + (lane >> 5) * 16;

    // SINGLE: the weight ring of the full 256-channel chunks, requested before the first tile is staged
    half8 bring[SINGLE ? 16 : 1];
    const bool single256 = SINGLE && NT == 1 && J.ck == 256 && Cin_p >= 256 && n_ok;
#ifdef ZV_STAMPS
    const int stamp_wg = bx + jobs.tps * jobs.segs.nseg * by;
    int stamp_k = 1;
#endif
    ZV_STAMP(0)
    // Loader waves: tile c is staged into LDS tile c & 1; barrier #c = "tile c is staged and the MFMA loop of chunk c - 1 is done", so
    // the loaders fill tile c + 1 (last read by chunk c - 1) while the MFMA waves walk chunk c.  One barrier per chunk for every wave.
    if constexpr (LW > 0)
        if (wave >= 4)
        {
            constexpr int NL = 64 * (LW > 0 ? LW : 1), UL = ZV_STAGE_ULW;
            const int ltid = tid - 256;
            // a tile that fits the loaders' registers (and is one tensor) travels in two halves, TWO tiles ahead: tile c + 2 is requested
            // before barrier #c (into the register set tile c left), tile c + 1 — requested a whole chunk earlier — is written after
            // it.  (One tile ahead, the request had only the loaders' wait at the barrier to land in: behind the MFMA waves' weight stream
            // on the CU's vector-memory path a round trip is ~2.5 us and the MFMA waves waited 2.1 us per chunk for the loaders.)
            // The loaders' raw barrier does not wait for the vector-memory counter.
            const bool split = J.pro != PRO_SUM3_ACT && rows * (J.ck >> 2) <= NL * UL && !(ZV_DBGBITS(J.dbg) & 1);
            StageRegs<UL> Ra, Rb;
            const int nck = J.ck;
            auto ckof = [&](int c) { return (Cin_p - c < nck) ? (Cin_p - c) : nck; };
#define ZV_LOAD_(P) stage_load_p<UL, P, NL>(S, cn_, ckn_, m0 - J.pad, rows, ltid, R_)
#define ZV_STORE_(P) stage_store_p<UL, P, NL>(S, dst_, RS, cn_, ckn_, m0 - J.pad, rows, ltid, R_)
#define ZV_LD_TILE(REGS, c)                                        \
    if ((c) < Cin_p)                                               \
    {                                                              \
        StageRegs<UL> &R_ = REGS;                                  \
        const int cn_ = (c), ckn_ = ckof(c);                       \
        ZV_STAGE_SPLIT_SWITCH(J.pro, ZV_LOAD_)                     \
    }
#define ZV_ST_TILE(REGS, c, buf)                                   \
    if ((c) < Cin_p)                                               \
    {                                                              \
        const StageRegs<UL> &R_ = REGS;                            \
        const int cn_ = (c), ckn_ = ckof(c);                       \
        char *dst_ = smem + (buf) * jobs.tile_bytes;               \
        ZV_STAGE_SPLIT_SWITCH(J.pro, ZV_STORE_)                    \
    }
#define ZV_RAW_BARRIER()                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("" ::: "memory");
            if (split)
            {
                ZV_LD_TILE(Ra, 0)
                ZV_LD_TILE(Rb, nck)
                ZV_ST_TILE(Ra, 0, 0)
                // chunks in pairs: tile c lives in Ra for even chunk indices, in Rb for odd ones
#ifdef ZV_STAMPS_LOADER
                const int stamp_wg = bx + jobs.tps * jobs.segs.nseg * by;
                int lk = 1;
                ZV_STAMP(0)
#define ZV_LSTAMP() if (lk < 11) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ZV_STAMP(lk) lk++; }
#else
#define ZV_LSTAMP()
#endif
                for (int c0 = 0; c0 < Cin_p; c0 += 2 * nck)
                {
                    ZV_LD_TILE(Ra, c0 + 2 * nck)
                    ZV_LSTAMP()
                    ZV_RAW_BARRIER()                   // barrier of chunk c0
                    ZV_LSTAMP()
                    ZV_ST_TILE(Rb, c0 + nck, 1)
                    ZV_LSTAMP()
                    if (c0 + nck >= Cin_p) break;
                    ZV_LD_TILE(Rb, c0 + 3 * nck)
                    ZV_LSTAMP()
                    ZV_RAW_BARRIER()                   // barrier of chunk c0 + ck
                    ZV_LSTAMP()
                    ZV_ST_TILE(Ra, c0 + 2 * nck, 0)
                    ZV_LSTAMP()
                }
#ifdef ZV_STAMPS_LOADER
                ZV_STAMP(11)
#endif
#undef ZV_LSTAMP
            }
            else
            {
                int par = 0;
                for (int c0 = 0; c0 < Cin_p; c0 += J.ck)
                {
                    const int ck = ckof(c0);
                    if (!(ZV_DBGBITS(J.dbg) & 1))
                        stage_tile<ZV_STAGE_ULW, NL>(J.pro, S, smem + par * jobs.tile_bytes, RS, c0, ck, m0 - J.pad, rows, ltid);
                    __syncthreads();
                    par ^= 1;
                }
            }
#undef ZV_RAW_BARRIER
#undef ZV_ST_TILE
#undef ZV_LD_TILE
#undef ZV_LOAD_
#undef ZV_STORE_
            return;
        }
    // (behind the loaders' branch: the ring's registers and the loaders' never live side by side)
    if constexpr (SINGLE)
        if (single256) preload_ring16<MT>(bring, (const half8 *)J.w + (size_t)ntl * K * nicb * 64 + lane);
    int par = 0;
    for (int c0 = 0; c0 < Cin_p; c0 += J.ck)
    {
        const int ck = (Cin_p - c0 < J.ck) ? (Cin_p - c0) : J.ck;
        if (LW == 0 && c0) __syncthreads();
        if (LW == 0 && !(ZV_DBGBITS(J.dbg) & 1))
        {
            // an f16 operand tensor (the decoder's pre-pass output) in 16-byte pieces, a 64-row x 256-channel tile in ONE round
            // trip (9 pieces per thread in flight); the 8-byte pieces of stage_tile took four (phase stamps: 5.5-6.2 us per chunk)
            if (J.pro == PRO_RAW_F16 && (ck & 7) == 0 && !SINGLE)
                stage_raw16<(BM >= 64 ? 9 : 5), 256>(S, smem, RS, c0, ck, m0 - J.pad, rows, tid);
            else
                // (the big wave tiles have the registers — dead before the accumulators live — for 8 pieces in flight: a
                // 64-row x 256-channel f32 tile in three round trips instead of five)
                stage_tile<(SINGLE ? ZV_STAGE_US : (MT * NT >= 4 ? 8 : ZV_STAGE_U))>(J.pro, S, smem, RS, c0, ck, m0 - J.pad, rows, tid);
        }
        if constexpr (LW > 0)
        {
            // barrier #c, raw: the weight ring's requests stay in flight across it (__syncthreads would drain them: 1.1 us per chunk)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        else
            __syncthreads();
#ifdef ZV_STAMPS
        if (stamp_k < 10) { ZV_STAMP(stamp_k) stamp_k++; }
#endif
        if (n_ok && !(ZV_DBGBITS(J.dbg) & 2))
        {
            const char *ab_ = abase + (LW > 0 ? par * jobs.tile_bytes : 0);
            const half8 *wp = (const half8 *)J.w + ((size_t)ntl * K * nicb + (size_t)K * (c0 >> 4)) * 64 + lane;
            const size_t wseg = (size_t)K * nicb * 64;             // half8 units between consecutive output tiles
            // full chunks of 256 / 128 / 64 channels take the immediate-address loop (S = K*nkc is a multiple of 4 there
            // and the blocks of a chunk are contiguous [tap][kc]: exactly the order mfma_taps walks)
            if constexpr (SINGLE && NT == 1)
            {
                if (ck == 256 && single256)
                    mfma_taps_single256<MT>(acc, ab_, dil * RS, wp, K, bring);
                else
                    mfma_chunk<MT, NT>(acc, ab_, RS, dil, wp, wseg, K, ck >> 4);
            }
            else if constexpr (NT == 2)
            {
                // the 64 x 64 wave tile keeps to loops with ONE set of four weight-fragment slots (168 registers: three
                // workgroups per CU)
                if (ck == 256 && J.ck == 256)
                    mfma_taps_ring4<256, MT, NT>(acc, ab_, dil * RS, wp, wseg, K);
                else
                    mfma_chunk<MT, NT>(acc, ab_, RS, dil, wp, wseg, K, ck >> 4);
            }
            else if (ck == 256 && J.ck == 256)
                mfma_taps<256, MT, NT, false>(acc, ab_, dil * RS, wp, wseg, K);
            else if (ck == 128 && J.ck == 128)
                mfma_taps<128, MT, NT, false>(acc, ab_, dil * RS, wp, wseg, K);
            else if (ck == 64 && J.ck == 64)
                mfma_taps<64, MT, NT, false>(acc, ab_, dil * RS, wp, wseg, K);
            else
                mfma_chunk<MT, NT>(acc, ab_, RS, dil, wp, wseg, K, ck >> 4);
        }
#ifdef ZV_STAMPS
        if (stamp_k < 11) { ZV_STAMP(stamp_k) stamp_k++; }
#endif
        par ^= 1;
    }

    // ---------------- epilogue ----------------
    if (!n_ok || (ZV_DBGBITS(J.dbg) & 4)) return;
    const float escale = J.escale;
    const int tbase = m0 + wm * 32 * MT + 4 * (lane >> 5);
    const bool has_res = J.res != nullptr;
    const float *res = has_res ? J.res + row0 * J.ldres : nullptr;
    const size_t out0 = row0 * J.ldo;
#pragma unroll
    for (int n = 0; n < NT; n++)
    {
        const int nt = ntl + n;
        if (nt < nt0) continue;                       // the duplicate of a clamped pair
        const int oc = nt * 32 + (lane & 31);
        if (oc >= Cout_p) continue;
        const float bias = J.bias ? J.bias[oc] : 0.f;
        // every residual load of the output tile column in flight before the first use (one round trip per 32 output
        // channels, not one per 32 x 32 tile: the weight-fragment registers of the MFMA loop are free by now)
        float resv[MT][16];
        if (has_res)
        {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    const int t = tbase + mt * 32 + (r & 3) + 8 * (r >> 2);
                    resv[mt][r] = res[(size_t)(t < L ? t : L - 1) * J.ldres + oc];
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
        {
            float outv[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int t = tbase + mt * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[mt][n][r] + bias;
                if (has_res) v = v + resv[mt][r];
                v = v * escale;
                if (J.eact) v = lrelu(v, J.oslope);
                outv[r] = v;
                if (t < L)
                {
                    // non-temporal stores: a conv's output is read by the NEXT launch, long after it has left the caches
                    // (measured on the batch: -1 ... -6.5 % per conv kernel, nothing slower)
                    // ... single-utterance launches store plainly: their small outputs are still in the memory-side cache when the next
                    // launch stages them (configs[2]: 1.711 -> 1.693 ms over three interleaved rounds)
                    if constexpr (SINGLE)
                    {
                        if (J.out_f16) ((_Float16 *)J.out)[out0 + (size_t)t * J.ldo + oc] = (_Float16)v;
                        else ((float *)J.out)[out0 + (size_t)t * J.ldo + oc] = v;
                    }
                    else
                    if (J.out_f16)
                        __builtin_nontemporal_store((_Float16)v, (_Float16 *)J.out + out0 + (size_t)t * J.ldo + oc);
                    else
                        __builtin_nontemporal_store(v, (float *)J.out + out0 + (size_t)t * J.ldo + oc);
                }
            }
            if (J.stat_part && oc < J.stat_C)
            {
                const int blk = (m0 >> 5) + wm * MT + mt;                  // 32-row block of the segment
                if (blk * 32 < L)
                    tile_stats_store(outv, tbase + mt * 32, L, J.stat_part + (((size_t)useg * J.stat_nblk + blk) * J.stat_C + oc) * 2);
            }
        }
    }
#ifdef ZV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZV_STAMP(11)
#endif
}

template <int MT, int WN, int NT, bool SINGLE = false>
static hipError_t launch_cfg(hipStream_t s, ConvJobs &jobs, int njobs, int Lmax, int Cout_p, int halo, int ck, int dmax_)
{
    constexpr int WM = 4 / WN;
    constexpr int BM = 32 * MT * WM;
    const int ntiles = (Cout_p + 31) / 32 - jobs.nt_begin;
    jobs.tps = (Lmax + BM - 1) / BM;
    dim3 grid(jobs.tps * jobs.segs.nseg, (ntiles + WN * NT - 1) / (WN * NT), njobs);
#ifdef ZV_STAMPS
    jobs.stamp = knob(ZV_STAMP_CONV) && knob(ZV_STAMP_CONV) == (int)grid.y && njobs == 1 &&
                 (knob(ZV_STAMP_CIN) ? jobs.j[0].Cin_p == knob(ZV_STAMP_CIN) : jobs.j[0].Cin_p >= 1024);
#endif
    // + dil rows: mfma_taps prefetches one tap past the end
    constexpr int LW = SINGLE ? ZV_SINGLE_LW : 0;
    jobs.xcd_ny = 0;
    jobs.warm = knob(ZV_CONV_WARM) != 0;
    if (SINGLE && knob(ZV_CONV_XCD) != 0)
    {
        // (see the kernel) grid.x = 8 XCDs x slots; slot q of XCD k = (channel group k + 8 (q / nx), row tile q % nx)
        jobs.xcd_ny = grid.y;
        jobs.xcd_nx = grid.x;
        grid = dim3(8 * grid.x * ((grid.y + 7) / 8), 1, njobs);
    }
    jobs.tile_bytes = round_up((BM + halo + dmax_) * (ck * 2 + 16), 16);
    const size_t lds = (size_t)jobs.tile_bytes * (LW > 0 ? 2 : 1);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = conv1d_mfma_kernel<MT, WN, NT, SINGLE>;
    if (lds > 64 * 1024)
    {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256 + 64 * LW), lds, s, jobs);
    return hipGetLastError();
}

static hipError_t launch_conv_gemm(hipStream_t s, const ConvJob &job, const Segs &segs, int rate);

// ---------------------------------------------------------------------------------------------------
// conv_stream_kernel — the memory-bound polyphase transposed convs of a batch (the last two upsample convs, reference
// src/hifigan.cpp:281-297 + 22-71: a few hundred MACs per output element against 8-12 bytes moved) as a stream.
// In conv1d_mfma_kernel every 64-row workgroup of such a conv is a chain of round trips — stage the tile, fetch 36-98 KiB of
// weight fragments from L2, store, wait for the stores to drain — and the launch's rate is workgroups in flight over that chain
// (ablations: with its MFMA loop off the last upsample conv takes 429 of its 610 us, stores alone 280).  Here a workgroup
//   * keeps the weight fragments of its four output tiles in REGISTERS (one 32-channel tile per wave, K * Cin/16 = 12 / 24
//     fragments) and walks a strip of up to 8 consecutive 64-row tiles with them;
//   * requests tile i + 1's rows (f32, one tensor: PRO_ACT / PRO_SCALE_ACT) into registers ahead of tile i's MFMAs, converts and
//     writes them into the other half of a double-buffered LDS tile behind them: one barrier per tile, no staging wait after
//     the first tile, the stores of tile i drain under tile i + 1.
// Same prologue arithmetic, same (tap, channel) chain per output element, same epilogue as conv1d_mfma_kernel: same bits.
template <int NKC>
__global__ __launch_bounds__(256, NKC == 8 ? 2 : 3) void conv_stream_kernel(const ConvJobs jobs, const int strip)
{
    constexpr int K = 3, CIN = 16 * NKC, RS = CIN * 2 + 16, NF = K * NKC, TROWS = 64 + K - 1;
    constexpr int C4 = CIN / 4, NP = (TROWS * C4 + 255) / 256, TILE_B = TROWS * RS;
    const ConvJob &J = jobs.j[0];
    const int useg = blockIdx.x / jobs.tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int s0 = (blockIdx.x - useg * jobs.tps) * strip * 64;
    if (s0 >= L) return;
    const size_t row0 = (size_t)sg.row0 * jobs.rate;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (J.Cout_p + 31) >> 5;
    const int nt_w = blockIdx.y * 4 + wave;
    const bool n_ok = nt_w < ntiles;
    const int nt = n_ok ? nt_w : ntiles - 1;                 // (a wave without a tile computes the last one again and stores nothing)
    half8 wf[NF];
    {
        const half8 *wp = (const half8 *)J.w + (size_t)nt * NF * 64 + lane;
#pragma unroll
        for (int i = 0; i < NF; i++) wf[i] = wp[i * 64];
    }
    const int oc = nt * 32 + (lane & 31);
    const float bias = J.bias ? J.bias[oc] : 0.f;
    const float *xs = (const float *)J.x0 + row0 * J.ldx;
    float *outp = (float *)J.out + row0 * J.ldo;
    const float sc = J.pro == PRO_ACT ? 1.0f : J.pscale, sl = J.slope;
    const int pad = J.pad;

    float4 v[NP];
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const int idx = tid + p * 256 < TROWS * C4 ? tid + p * 256 : TROWS * C4 - 1;
            const int r = idx / C4, c4 = idx % C4;
            const int t = m0 - pad + r;
            v[p] = *(const float4 *)(xs + (size_t)(t < 0 ? 0 : (t < L ? t : L - 1)) * J.ldx + c4 * 4);
        }
    };
    auto write_tile = [&](int buf, int m0) {
#pragma unroll
        for (int p = 0; p < NP; p++)
        {
            const int idx = tid + p * 256;
            if (idx >= TROWS * C4) continue;
            const int r = idx / C4, c4 = idx % C4;
            const int t = m0 - pad + r;
            float4 x = v[p];
            x.x = x.x * sc;
            x.y = x.y * sc;
            x.z = x.z * sc;
            x.w = x.w * sc;
            half4 h;
            h[0] = (_Float16)lrelu(x.x, sl);
            h[1] = (_Float16)lrelu(x.y, sl);
            h[2] = (_Float16)lrelu(x.z, sl);
            h[3] = (_Float16)lrelu(x.w, sl);
            uint2 pk = *(uint2 *)&h;
            const bool in = t >= 0 && t < L;
            pk.x = in ? pk.x : 0u;
            pk.y = in ? pk.y : 0u;
            *(uint2 *)(smem + buf * TILE_B + r * RS + c4 * 8) = pk;
        }
    };

    int m0 = s0;
    load_tile(m0);
    write_tile(0, m0);
    __syncthreads();
    const char *abase = smem + (lane & 31) * RS + (lane >> 5) * 16;
    for (int i = 0; i < strip && m0 < L; i++, m0 += 64)
    {
        const bool more = i + 1 < strip && m0 + 64 < L;
        if (more) load_tile(m0 + 64);
        __builtin_amdgcn_sched_barrier(0);          // the requests stay ahead of the MFMAs
        floatx16 acc[2];
        const floatx16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char *ab = abase + (i & 1) * TILE_B;
#pragma unroll
        for (int tap = 0; tap < K; tap++)
#pragma unroll
            for (int kc = 0; kc < NKC; kc++)
            {
                const half8 a0 = *(const half8 *)(ab + tap * RS + kc * 32);
                const half8 a1 = *(const half8 *)(ab + (32 + tap) * RS + kc * 32);
                const bool first = tap == 0 && kc == 0;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, wf[tap * NKC + kc], first ? z : acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[tap * NKC + kc], first ? z : acc[1], 0, 0, 0);
            }
        if (n_ok)
        {
            const int tbase = m0 + 4 * (lane >> 5);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    const int t = tbase + mt * 32 + (r & 3) + 8 * (r >> 2);
                    const float vv = (acc[mt][r] + bias) * J.escale;
                    if (t < L) __builtin_nontemporal_store(vv, outp + (size_t)t * J.ldo + oc);
                }
        }
        if (more) write_tile((i + 1) & 1, m0 + 64);
        __syncthreads();
    }
}

// the convs conv_stream_kernel takes: one job, 3 taps, one chunk of 64 / 128 input channels read from ONE f32 tensor, bias-only epilogue
static bool conv_stream_ok(const ConvJob &j, int njobs, int nt_begin)
{
    return njobs == 1 && nt_begin == 0 && (j.pro == PRO_ACT || j.pro == PRO_SCALE_ACT) && j.K == 3 && j.dil == 1 && j.pad == 1 &&
           (j.Cin_p == 64 || j.Cin_p == 128) && j.ck == j.Cin_p && (j.ldx & 3) == 0 && !j.res && !j.stat_part && !j.eact && !j.out_f16 &&
           !j.x1 && !j.x2;
}

static hipError_t launch_conv_stream(hipStream_t s, const ConvJob &job, int n_cu, const Segs &segs, int rate)
{
    ConvJobs js;
    js.j[0] = job;
    js.j[0].dbg = diag_bits();
    for (int i = 1; i < CONV_MAX_JOBS; i++) js.j[i] = js.j[0];
    js.segs = segs;
    js.rate = rate;
    js.nt_begin = 0;
    js.order = 0;
    const int Lmax = segs.max_rows * rate;
    const int ntiles = (job.Cout_p + 31) / 32, gy = (ntiles + 3) / 4;
    const int occ = job.Cin_p == 128 ? 2 : 3;
    // strips of 8 tiles while that still leaves about twelve rounds of workgroups, else 4, 2
    int strip = 8;
    while (strip > 2 && (long)((Lmax + 64 * strip - 1) / (64 * strip)) * segs.nseg * gy < 12L * occ * n_cu) strip >>= 1;
    js.tps = (Lmax + 64 * strip - 1) / (64 * strip);
    const dim3 grid(js.tps * segs.nseg, gy, 1);
    const size_t lds = (size_t)2 * 66 * (job.Cin_p * 2 + 16);
    if (job.Cin_p == 128)
        hipLaunchKernelGGL(conv_stream_kernel<8>, grid, dim3(256), lds, s, js, strip);
    else
        hipLaunchKernelGGL(conv_stream_kernel<4>, grid, dim3(256), lds, s, js, strip);
    return hipGetLastError();
}

static hipError_t launch_conv_from(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu, const Segs &segs, int rate, int nt_begin);

hipError_t launch_conv(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu, const Segs &segs, int rate)
{
    if (njobs < 1 || njobs > CONV_MAX_JOBS || segs.nseg < 1 || segs.max_rows < 1) return hipErrorInvalidValue;
    // batches of wide convs over an f16 operand tensor (the decoder's, behind its pre-pass): whole groups of 8 output tiles
    // on conv_gemm_kernel, job by job; the tiles left over (1 056 channels = 4 groups + 1 tile) on the kernel below
    {
        const int g_env = knob(ZV_CONV_GEMM);
        const long rows = (long)segs.max_rows * rate * segs.nseg;
        ConvJob rest[CONV_MAX_JOBS];
        int nrest = 0, ngemm = 0;
        for (int i = 0; i < njobs; i++)
        {
            const ConvJob &j = jobs[i];
            const bool ok = g_env != 0 && (g_env == 2 || rows >= 16384) && j.w8 && j.pro == PRO_RAW_F16 && j.Cin_p >= 128 &&
                            conv_gemm_groups(j.Cout_p) >= 1 && !j.out_f16 && (j.ldx & 7) == 0;
            if (!ok)
            {
                rest[nrest++] = j;
                continue;
            }
            ngemm++;
            hipError_t e = launch_conv_gemm(s, j, segs, rate);
            if (e != hipSuccess) return e;
            const int done = conv_gemm_tiles(j.Cout_p);
            if (done * 32 < j.Cout_p)
            {
                e = launch_conv_from(s, &j, 1, n_cu, segs, rate, done);
                if (e != hipSuccess) return e;
            }
        }
        if (ngemm) return nrest ? launch_conv_from(s, rest, nrest, n_cu, segs, rate, 0) : hipSuccess;
    }
    return launch_conv_from(s, jobs, njobs, n_cu, segs, rate, 0);
}

static hipError_t launch_conv_from(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu, const Segs &segs, int rate, int nt_begin)
{
    const int dbg = diag_bits();
    ConvJobs js;
    js.segs = segs;
    js.rate = rate;
    js.tps = 0;
    js.nt_begin = nt_begin;
    js.order = 0;
    const int Lmax = segs.max_rows * rate;
    int halo = 0, ck = 0, dmax = 1;
    for (int i = 0; i < njobs; i++)
    {
        js.j[i] = jobs[i];
        dmax = jobs[i].dil > dmax ? jobs[i].dil : dmax;
        js.j[i].dbg = dbg;
        if (jobs[i].Cout_p != jobs[0].Cout_p) return hipErrorInvalidValue;
        const int h = (jobs[i].K - 1) * jobs[i].dil;
        if (h > halo) halo = h;
        ck = jobs[i].ck > ck ? jobs[i].ck : ck;
        if (jobs[i].stat_part && jobs[i].stat_nblk * 32 < Lmax) return hipErrorInvalidValue;
    }
    for (int i = njobs; i < CONV_MAX_JOBS; i++) js.j[i] = js.j[0];
    const int Cout_p = jobs[0].Cout_p;
    const int ntiles = (Cout_p + 31) / 32 - nt_begin;
    if (ntiles < 1) return hipErrorInvalidValue;
    // three output tiles already take four waves (one idles): the input tile is staged once instead of twice
    // (two row tiles x two output tiles per workgroup instead, so that row pairs share weight fragments: a single utterance 1.71 -> 1.79 ms)
    const int WN = ntiles >= 3 ? 4 : (ntiles >= 2 ? 2 : 1);
    // pick the tallest wave tile (most B-fragment reuse) that still gives every CU about two workgroups; the tile
    // shape never changes an output bit: every output element is one accumulator chain over (chunk, tap, channel)
    auto wgs = [&](int MT, int NT) {
        const int BM = 32 * MT * (4 / WN);
        return (long)((Lmax + BM - 1) / BM) * segs.nseg * ((ntiles + WN * NT - 1) / (WN * NT)) * njobs;
    };
    int MT = 4;
    while (MT > 1 && wgs(MT, 1) < 2L * n_cu) MT >>= 1;
    while (MT > 1 && (size_t)(32 * MT * (4 / WN) + halo + dmax) * (ck * 2 + 16) > 80 * 1024) MT >>= 1;   // keep >= 2 workgroups per CU in LDS
    {
        // memory-bound convs (the polyphase transposed convs of the narrow HiFi-GAN stages: a few hundred MACs per output
        // element against 8 bytes moved) want workgroups in flight, not weight reuse: measured on the batch, the last
        // three upsample convs take 897 / 595 / 452 us with the tall tiles and 636 / 569 / 416 us with these
        const double ai = 2.0 * jobs[0].K * jobs[0].Cin_p * Cout_p / (4.0 * (jobs[0].Cin_p + Cout_p));
        // ... and the ones conv_stream_kernel takes run there (ZV_CONV_STREAM = 0 never, 2 at any length)
        const int st_env = knob(ZV_CONV_STREAM);
        if (st_env && conv_stream_ok(jobs[0], njobs, nt_begin) && (st_env == 2 || (ai < 200.0 && wgs(1, 1) >= 16L * n_cu)))
            return launch_conv_stream(s, jobs[0], n_cu, segs, rate);
        // (round 3: 64-row tiles for all of them — the 128 -> 4 x 64 channel one 573 -> 501 us: half the weight stream per row)
        if (ai < 200.0 && wgs(1, 1) >= 16L * n_cu) MT = std::min(MT, 2);
    }
    if (MT < knob(ZV_CONV_MT)) MT = knob(ZV_CONV_MT);      // measurement hook: minimum MT
    // two output tiles per wave once a conv is wide and the launch still has rounds of workgroups to spare
    const int nt_env = knob(ZV_CONV_NT);
    // ... and deep (>= 2 048 products per output element: the decoder's; the first two upsample convs, 1 536 / 768 deep, measured
    // 265 / 417 us on 64 x 64 wave tiles and 245 / 395 us on 128 x 32 ones)
    int NT = (WN == 4 && ntiles >= 8 && MT >= 2 && wgs(MT, 2) >= 4L * n_cu && jobs[0].K * jobs[0].Cin_p >= 2048) ? 2 : 1;
    if (nt_env == 1 || (nt_env == 2 && WN == 4 && ntiles >= 2 && MT >= 2)) NT = nt_env;
    if (NT == 2 && MT == 4) MT = 2;        // 64 x 64 per wave: the 128 x 64 shape does not fit 256 registers
    {
        // single-utterance launches (at most a round of workgroups, one wave per SIMD): the deep-lookahead loop for the
        // 256-channel chunks
        // ... of convs with SEVERAL such chunks (the decoder's): measured per launch at 512 frames, the one-chunk 256-channel
        // convs of HiFi-GAN stage 1 take 23.0 us on this loop against 20.0 us on mfma_taps (profiles/r02_v2_single_utterance_kernel_trace.txt
        // vs round 1's trace), the five-chunk decoder convs 29.6 against 33
        if (knob(ZV_CONV_SINGLE) != 0 && MT == 1 && NT == 1 && ck == 256 && wgs(1, 1) <= 2L * n_cu &&
            (jobs[0].Cin_p > 256 || knob(ZV_CONV_SINGLE) == 2))
        {
            if (WN == 4) return launch_cfg<1, 4, 1, true>(s, js, njobs, Lmax, Cout_p, halo, ck, dmax);
            if (WN == 2) return launch_cfg<1, 2, 1, true>(s, js, njobs, Lmax, Cout_p, halo, ck, dmax);
        }
    }
#define ZV_CASE(mt, wn, nt) \
    if (MT == mt && WN == wn && NT == nt) return launch_cfg<mt, wn, nt>(s, js, njobs, Lmax, Cout_p, halo, ck, dmax);
    ZV_CASE(4, 4, 1) ZV_CASE(2, 4, 1) ZV_CASE(1, 4, 1) ZV_CASE(2, 4, 2)
    ZV_CASE(4, 2, 1) ZV_CASE(2, 2, 1) ZV_CASE(1, 2, 1)
    ZV_CASE(4, 1, 1) ZV_CASE(2, 1, 1) ZV_CASE(1, 1, 1)
#undef ZV_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// Fused dilation pair of a HiFi-GAN residual block (reference src/hifigan.cpp:99-182):
//     xt = lrelu(conv(lrelu(y), k, dil) + b1);  out = y + (conv(xt, k, 1) + b2)
// One workgroup produces TM = BM - (k-1) output rows: it stages f16(lrelu(y)) for BM + (k-1)*dil rows, runs
// conv1 over BM rows (the k-1 extra rows are conv2's halo) with the weights as the MFMA A operand so that a
// lane ends up with 4 consecutive channels per register quad, packs xt = f16(lrelu(. + b1)) straight back
// into the same LDS region (rows outside [0, L) are conv2's zero padding), runs conv2 from there and adds
// bias + residual in the epilogue.  xt never touches HBM.
//
// The MFMA loop is specialised on the channel count so that every LDS / weight address inside an iteration is
// an immediate: one body = 8 steps = 8 / (CP/16) taps, the per-body bookkeeping is a handful of vector adds.  The first version of this loop carried ~15 scalar/vector instructions of
// tap/channel bookkeeping per MFMA and was instruction-issue bound (SQ_ACTIVE_INST_ANY ~ 74 % of the kernel with
// the MFMA pipe 19 % busy, profiles/r01_v2_pmc.txt).  Weights for this path are packed per 32-channel output
// tile as round_up(K*CP/16, 4) + 8 blocks, zero blocks behind the real ones: the loop needs no tail handling
// and its prefetch never leaves the tile's segment.
size_t pair_weight_halfs(int Cp, int K)
{
    const int nkc = Cp / 16;
    return (size_t)(Cp / 32) * (round_up(K * nkc, 4) + 8) * 512;
}

void pack_pair_weight(const uint16_t *w, int K, int C, int Cp, uint16_t *dst)
{
    const int nkc = Cp / 16;
    const size_t seg = (size_t)(round_up(K * nkc, 4) + 8) * 512;
    memset(dst, 0, pair_weight_halfs(Cp, K) * 2);
    for (int nt = 0; nt < Cp / 32; nt++)
        for (int tap = 0; tap < K; tap++)
            for (int kc = 0; kc < nkc; kc++)
            {
                uint16_t *d = dst + nt * seg + (size_t)(tap * nkc + kc) * 512;
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++)
                    {
                        const int oc = nt * 32 + (lane & 31), ic = kc * 16 + 8 * (lane >> 5) + j;
                        d[lane * 8 + j] = (oc < C && ic < C) ? w[((size_t)oc * C + ic) * K + tap] : (uint16_t)0;
                    }
            }
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (observed, not promised: only the speed
// depends on it) and each XCD has its own L2.  With grid.x a multiple of 8, workgroup b runs on XCD b % 8; giving XCD
// x the x-th contiguous eighth of a job's time tiles keeps neighbouring tiles — which share their halo rows — behind the
// same L2 instead of spreading every halo over two XCDs.
__device__ __forceinline__ int zv_xcd_tile(int b, int ntiles)
{
#ifdef ZV_NO_XCD_MAP
    return b;
#else
    // per job: its own tile count decides the eighths (jobs of one launch have different tile heights), so that
    // every XCD gets an equal share of every job; workgroups beyond the job's tiles return a tile index >= ntiles
    const int per = (ntiles + 7) >> 3, idx = b >> 3;
    return idx < per ? (b & 7) * per + idx : ntiles;
#endif
}

// leaky-ReLU for 0 <= slope <= 1 as max(x, x*slope): same bits as (x > 0 ? x : x*slope), one instruction less and
// no compare/select pair; raw v_max_f32 keeps hipcc from adding a canonicalising multiply in front of fmaxf
__device__ __forceinline__ float lrelu_max(float x, float s)
{
    float r;
    const float xs = x * s;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(xs));
    return r;
}

// max(x, xs) with xs = x * slope already formed (packed multiplies)
__device__ __forceinline__ float lrelu_max_pre(float x, float xs)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(xs));
    return r;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// f16(lrelu(x + b)) of four values: packed add / multiply / convert (v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32: the
// same IEEE results as the scalar forms), max(x, x*slope) as in lrelu_max
__device__ __forceinline__ uint2 lrelu4_f16(float x0, float x1, float x2, float x3, float sl)
{
    const float2v a = {x0, x1}, c = {x2, x3};
    const float2v as = a * sl, cs = c * sl;
    const float2v ra = {lrelu_max_pre(a[0], as[0]), lrelu_max_pre(a[1], as[1])};
    const float2v rc = {lrelu_max_pre(c[0], cs[0]), lrelu_max_pre(c[1], cs[1])};
    const half2v ha = __builtin_convertvector(ra, half2v), hc = __builtin_convertvector(rc, half2v);
    uint2 pk;
    pk.x = *(const unsigned int *)&ha;
    pk.y = *(const unsigned int *)&hc;
    return pk;
}


// Stage f16(lrelu(y)) rows through a buffer descriptor: rows outside [0, L) are out of the descriptor's range and
// read as 0 (= the conv's zero padding, lrelu(0) = 0) with no per-row predicate; CP is a power of two so the
// row / column split of the flat index is a shift and a mask.
template <int U, int CP>
__device__ __forceinline__ void stage_act_buf(__amdgpu_buffer_rsrc_t rsrc, char *smem, int row_t0, int rows, int tid, float slope)
{
    constexpr int RS = CP * 2 + 16, COLS = CP / 4, SH = (CP == 32) ? 3 : (CP == 64 ? 4 : (CP == 128 ? 5 : 6));
    const int total = rows * COLS;
    for (int base = tid; base < total; base += 256 * U)
    {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int idx = base + u * 256;
            const int r = idx >> SH, c4 = idx & (COLS - 1);
            const int voff = (idx < total) ? ((row_t0 + r) * CP + c4 * 4) * 4 : -16;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
        {
            const int idx = base + u * 256;
            if (idx < total)
            {
                const int r = idx >> SH, c4 = idx & (COLS - 1);
                half4 h;
                h[0] = (_Float16)lrelu_max(__uint_as_float(v[u].x), slope);
                h[1] = (_Float16)lrelu_max(__uint_as_float(v[u].y), slope);
                h[2] = (_Float16)lrelu_max(__uint_as_float(v[u].z), slope);
                h[3] = (_Float16)lrelu_max(__uint_as_float(v[u].w), slope);
                *(half4 *)(smem + r * RS + c4 * 8) = h;
            }
        }
    }
}

// MERGE: one workgroup runs the SAME time tile of all the launch's jobs (the MRF branches of a stage) one after the other
// and stores only (out_0 + out_1) + out_2 — the sum the next layer starts with (reference src/hifigan.cpp:300-315) — so the
// branch outputs of a stage's last dilation pair never reach HBM and the consumer reads one tensor instead of three.  The
// tile height is that of the job with the most taps (a few rows of extra halo for the others).
//
// The kernel runs on v_mfma_f32_16x16x32_f16 (round 4; rounds 1-3: 32x32x16, whose MFMA loop — mfma_taps_deep above — the
// single-utterance whole-block kernel resblock_triple_kernel still uses).
//
// The two f16 MFMA shapes give the SAME BITS for one k-ordered accumulation chain: a chain walked 16 products per instruction
// (32x32x16) equals the same chain walked 32 per instruction (16x16x32) — measured on 204 800 random elements at three scales,
// scripts/mfma_shape_bits.hip: 0 differ (neither equals a sequential fma chain nor an f64 sum rounded once: the matrix core has
// its own order, but one order for both shapes).  So a kernel may change its shape without leaving the family's "every output
// element is ONE chain over (chunk, tap, channel)" contract, and the chip holds a higher clock under the 16 x 16 shape
// (MI355X_MICROARCH.md, DVFS give-back item 7).  One step here = 32 channels of one tap = two steps of the kernel above:
//   LDS side   l[mt][lt]: the two 16-row tiles of 32-row block mt.  Fragment row c of tile lt is block row 2c + lt (even / odd
//              interleave): with the tile's row stride 2 CP + 16 bytes the ds_read_b128 of lane (c, g = k group) then falls on
//              16 distinct 16-byte slots per lane group (rows c and 16 + c would share a slot: 2-way);
//   weights    w[nt][wt]: the two 16-channel tiles of output tile nt, packed in fragment order by pack_pair_weight16 (conv1: the
//              A operand, row r of tile wt = channel 16 wt + r; conv2: the B operand, column c of tile wt = channel 2c + wt, so a
//              lane's two tiles are NEIGHBOURING channels and the epilogue moves 8 bytes per lane: four rows x 128 bytes per
//              instruction, half the instructions of the 4-byte column accesses above).
// acc[mt][nt][wt][lt] is a 16 x 16 tile: conv1 (weights as A) lane (c, g) holds channels 16 wt + 4g + i of time row 2c + lt;
// conv2 holds time rows 8g + 2i + lt of channel 2c + wt.
typedef float floatx4 __attribute__((ext_vector_type(4)));

size_t pair_weight16_halfs(int Cp, int K) { return pair_weight_halfs(Cp, K); }

// conv2_layout: the B-operand form (channel 2c + wt); else the A-operand form (channel 16 wt + r)
void pack_pair_weight16(const uint16_t *w, int K, int C, int Cp, uint16_t *dst, bool conv2_layout)
{
    const int nk2 = Cp / 32;
    const size_t seg = (size_t)(round_up(K * (Cp / 16), 4) + 8) * 512;
    memset(dst, 0, pair_weight16_halfs(Cp, K) * 2);
    for (int nt = 0; nt < Cp / 32; nt++)
        for (int tap = 0; tap < K; tap++)
            for (int k2 = 0; k2 < nk2; k2++)
                for (int wt = 0; wt < 2; wt++)
                {
                    uint16_t *d = dst + nt * seg + ((size_t)(tap * nk2 + k2) * 2 + wt) * 512;
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++)
                        {
                            const int r = lane & 15, g = lane >> 4;
                            const int oc = nt * 32 + (conv2_layout ? 2 * r + wt : 16 * wt + r), ic = k2 * 32 + 8 * g + j;
                            d[lane * 8 + j] = (oc < C && ic < C) ? w[((size_t)oc * C + ic) * K + tap] : (uint16_t)0;
                        }
                }
}

template <int MT, int NT, bool SWAP, bool ZERO = false>
__device__ __forceinline__ void mfma16_step(floatx4 (&acc)[MT][NT][2][2], const half8 (&l)[MT][2], const half8 (&w)[NT][2])
{
    const floatx4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int wt = 0; wt < 2; wt++)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
                {
                    if constexpr (SWAP)      // weights as the A operand -> D[oc][time]
                        acc[mt][nt][wt][lt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[nt][wt], l[mt][lt], ZERO ? z : acc[mt][nt][wt][lt], 0, 0, 0);
                    else
                        acc[mt][nt][wt][lt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l[mt][lt], w[nt][wt], ZERO ? z : acc[mt][nt][wt][lt], 0, 0, 0);
                }
}

// the first two steps' weight fragments of a contraction (see deep_preload_b)
template <int NT>
__device__ __forceinline__ void deep16_preload_b(half8 (&b0)[2][NT][2], const half8 *wq, size_t wseg)
{
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int wt = 0; wt < 2; wt++) b0[u][nt][wt] = wq[nt * wseg + (u * 2 + wt) * 64];
}

// mfma_taps_deep's structure in steps of 32 channels: a body = 4 steps (= its 8), the weight fragments in two ping-pong sets of
// two steps each (requested a set ahead), the LDS fragments one step ahead through a ring of four register sets, a scheduling
// fence per step.  Same (tap, channel) order: the same chain per output element.
template <int CP, int MT, int NT, bool SWAP>
__device__ __forceinline__ void mfma16_taps_deep(floatx4 (&acc)[MT][NT][2][2], const char *ap, int dilRS, const half8 *wq, size_t wseg, int K,
                                                 half8 (&b0)[2][NT][2])
{
    constexpr int RS = CP * 2 + 16, NK2 = CP / 32;
    constexpr int TPB = (NK2 >= 4) ? 1 : 4 / NK2;        // taps per body: 4 / 2 / 1 / (1/2) for CP = 32 / 64 / 128 / 256
    constexpr bool HALF = NK2 == 8;                      // CP = 256: a tap is two bodies
    const int nsb = (K * NK2 + 1) >> 1;                  // 2-step sub-blocks (the last one may run partly on zero weights)
    const int nb = nsb >> 1;
    half8 b1[2][NT][2];
    wq += 4 * 64;
    half8 a[4][MT][2];
#define ZV16_UN(un) ((un) >= 4 ? (un) - 4 : 0)
#define ZV16_A_ADDR(un) ((un) >= 4 ? tbn[(ZV16_UN(un) / NK2) % 4] + (ZV16_UN(un) % NK2) * 64 : tb[((un) / NK2) % 4] + ((un) % NK2) * 64)
#define ZV16_LOAD_A(un)                                                                                          \
    {                                                                                                            \
        const char *np_ = ZV16_A_ADDR(un);                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                                        \
            _Pragma("unroll") for (int lt = 0; lt < 2; lt++) a[(un) % 4][mt][lt] = *(const half8 *)(np_ + (mt * 32 + lt) * RS); \
    }
#define ZV16_STEP(u, bset) ZV16_LOAD_A((u) + 1) mfma16_step<MT, NT, SWAP>(acc, a[(u) % 4], bset[(u) % 2]); __builtin_amdgcn_sched_barrier(0);
#define ZV16_STEPZ(u, bset) ZV16_LOAD_A((u) + 1) mfma16_step<MT, NT, SWAP, true>(acc, a[(u) % 4], bset[(u) % 2]); __builtin_amdgcn_sched_barrier(0);
#define ZV16_BODY(FIRSTSTEP)                                                                                     \
    {                                                                                                            \
        const char *tb[4], *tbn[4];                                                                              \
        tb[0] = ap;                                                                                              \
        _Pragma("unroll") for (int x = 1; x < 4; x++) tb[x] = (x < TPB) ? ap + x * dilRS : ap;                   \
        const char *apn = HALF ? ((ib & 1) ? ap + (dilRS - 256) : ap + 256) : ap + TPB * dilRS; /* next body */  \
        tbn[0] = apn;                                                                                            \
        _Pragma("unroll") for (int x = 1; x < 4; x++) tbn[x] = (x < TPB) ? apn + x * dilRS : apn;                \
        _Pragma("unroll") for (int u = 0; u < 2; u++)                                                            \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                    \
                _Pragma("unroll") for (int wt = 0; wt < 2; wt++) b1[u][nt][wt] = wq[nt * wseg + (u * 2 + wt) * 64];        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        FIRSTSTEP(0, b0) ZV16_STEP(1, b0)                                                                        \
        _Pragma("unroll") for (int u = 0; u < 2; u++)                                                            \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                    \
                _Pragma("unroll") for (int wt = 0; wt < 2; wt++) b0[u][nt][wt] = wq[nt * wseg + (4 + u * 2 + wt) * 64];    \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        ZV16_STEP(2, b1) ZV16_STEP(3, b1)                                                                        \
        ap = apn;                                                                                                \
        wq += 8 * 64;                                                                                            \
    }
    {
        const char *tb[4], *tbn[4];
        tb[0] = ap;
#pragma unroll
        for (int x = 1; x < 4; x++) tb[x] = (x < TPB) ? ap + x * dilRS : ap;
#pragma unroll
        for (int x = 0; x < 4; x++) tbn[x] = ap;
        ZV16_LOAD_A(0)
    }
    {
        const int ib = 0;
        ZV16_BODY(ZV16_STEPZ)
    }
    for (int ib = 1; ib < nb; ib++) ZV16_BODY(ZV16_STEP)
    if constexpr (CP == 64)
        if (nsb & 1)                             // odd sub-block count (K = 3 mod 4 taps): one more tap on b0
        {
            const char *tb[4] = {ap, ap, ap, ap};
            const char *tbn[4] = {ap, ap, ap, ap};
            (void)tbn;
            ZV16_STEP(0, b0) ZV16_STEP(1, b0)
        }
#undef ZV16_BODY
#undef ZV16_STEPZ
#undef ZV16_STEP
#undef ZV16_LOAD_A
#undef ZV16_A_ADDR
#undef ZV16_UN
}

template <int CP, int MT, bool MERGE>
__global__ __launch_bounds__(256) void resblock_pair_kernel(const PairJobs jobs)
{
    constexpr int NT = (CP == 256) ? 2 : 1;            // output tiles of 32 channels per wave
    constexpr int WN = CP / 32 / NT, WM = 4 / WN;
    constexpr int BM = 32 * MT * WM;
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    const int jz = (int)blockIdx.z, bx = (int)blockIdx.x;
    const int TM = BM - (MERGE ? jobs.kmax - 1 : jobs.j[jz].K - 1);
    const int tps = (jobs.segs.max_rows * jobs.rate + TM - 1) / TM;
    const int vt = zv_xcd_tile(bx, tps * jobs.segs.nseg);
    if (vt >= tps * jobs.segs.nseg) return;
    const int useg = vt / tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int t0 = (vt - useg * tps) * TM;
    if (t0 >= L) return;
    floatx4 msum[MERGE ? MT : 1][MERGE ? NT : 1][2][2];
    for (int jb = MERGE ? 0 : jz; jb < (MERGE ? jobs.njobs : jz + 1); jb++)
    {
    const PairJob &P = jobs.j[jb];
    const int K = P.K, dil = P.dil;
    const int h2 = (K - 1) / 2, h1 = h2 * dil;
    const float *y_seg = P.y + (size_t)sg.row0 * jobs.rate * CP;
    float *out_seg = (MERGE ? jobs.merge_out : P.out) + (size_t)sg.row0 * jobs.rate * CP;
    if (MERGE && jb) __syncthreads();               // the previous job's conv2 is done reading the LDS tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lc = lane & 15, lg = lane >> 4;
    const size_t wseg = (size_t)(round_up(K * NKC, 4) + 8) * 64;        // half8 units per n-tile segment

    // ---- stage X: LDS row r <-> time t0 - h2 - h1 + r   (+ dil rows: the zero-weight tap of CP = 32 must read finite data)
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y_seg, 0, L * CP * 4, 0x00020000);
    constexpr int STAGE_U = MERGE ? 4 : ((CP == 32) ? 10 : (CP == 64 ? 12 : (CP == 128 ? ZV_STAGE_U128 : ZV_STAGE_U256)));
    constexpr bool EARLY_B = CP <= 64 && !MERGE;
    half8 bw[2][NT][2];
    if constexpr (EARLY_B) deep16_preload_b<NT>(bw, (const half8 *)P.w1 + wn * NT * wseg + lane, wseg);
    stage_act_buf<STAGE_U, CP>(rs_y, smem, t0 - h2 - h1, BM + 2 * h1 + dil, tid, P.slope);
    __syncthreads();

    // this lane's fragment rows: block row 2c (+ lt) of the wave's 32-row blocks, k group g
    const char *abase = smem + (wm * 32 * MT + 2 * lc) * RS + lg * 16;
    floatx4 acc[MT][NT][2][2];

    // ---- conv1 (dilated), transposed product: acc[mt][nt][wt][lt][i] = xt_pre[time = mt*32 + 2c + lt][oc = 16 wt + 4g + i]
    if constexpr (!EARLY_B) deep16_preload_b<NT>(bw, (const half8 *)P.w1 + wn * NT * wseg + lane, wseg);
    mfma16_taps_deep<CP, MT, NT, true>(acc, abase, dil * RS, (const half8 *)P.w1 + wn * NT * wseg + lane, wseg, K, bw);
    if constexpr (EARLY_B) deep16_preload_b<NT>(bw, (const half8 *)P.w2 + wn * NT * wseg + lane, wseg);     // conv2's first fragments travel under the xt pack
    __syncthreads();                       // every wave is done reading X: its LDS region becomes XT

    // ---- xt = f16(lrelu(conv1 + b1)), zero outside [0, L); XT row i <-> time t0 - h2 + i
    {
        const float sl = P.slope;
        const bool edge = t0 - h2 < 0 || t0 - h2 + BM > L;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
        {
            const int ocb = (wn * NT + nt) * 32;
            float4 bq[2];
#pragma unroll
            for (int wt = 0; wt < 2; wt++) bq[wt] = *(const float4 *)(P.b1 + ocb + 16 * wt + 4 * lg);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
                {
                    const int i = wm * 32 * MT + mt * 32 + 2 * lc + lt;
                    const int t = t0 - h2 + i;
                    const bool in = !edge || (t >= 0 && t < L);
#pragma unroll
                    for (int wt = 0; wt < 2; wt++)
                    {
                        half4 h;
                        h[0] = (_Float16)lrelu_max(acc[mt][nt][wt][lt][0] + bq[wt].x, sl);
                        h[1] = (_Float16)lrelu_max(acc[mt][nt][wt][lt][1] + bq[wt].y, sl);
                        h[2] = (_Float16)lrelu_max(acc[mt][nt][wt][lt][2] + bq[wt].z, sl);
                        h[3] = (_Float16)lrelu_max(acc[mt][nt][wt][lt][3] + bq[wt].w, sl);
                        uint2 pk = *(uint2 *)&h;
                        if (edge)
                        {
                            pk.x = in ? pk.x : 0u;
                            pk.y = in ? pk.y : 0u;
                        }
                        *(uint2 *)(smem + i * RS + (ocb + 16 * wt + 4 * lg) * 2) = pk;
                    }
                }
        }
    }
    __syncthreads();

    // ---- conv2 (dil 1): output row j <-> time t0 + j reads XT rows j .. j + 2*h2; rows j >= TM are discarded
    if constexpr (!EARLY_B) deep16_preload_b<NT>(bw, (const half8 *)P.w2 + wn * NT * wseg + lane, wseg);
    mfma16_taps_deep<CP, MT, NT, false>(acc, abase, RS, (const half8 *)P.w2 + wn * NT * wseg + lane, wseg, K, bw);

    // ---- epilogue: out = y + (conv2 + b2).  acc[mt][nt][wt][lt][i] = time row mt*32 + 8g + 2i + lt, channel 2c + wt: the lane's two
    // weight tiles are neighbouring channels -> 8-byte accesses, four rows x 128 bytes per instruction.  Buffer descriptors over
    // exactly this tile's valid rows: row >= TM or time >= L is out of range (loads give 0, stores are dropped).
    const int nrows = (L - t0 < TM) ? (L - t0) : TM;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(y_seg + (size_t)t0 * CP), 0, nrows * CP * 4, 0x00020000);
    float *const outp = (!MERGE && P.sum_out) ? P.sum_out + (size_t)sg.row0 * jobs.rate * CP : out_seg;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(outp + (size_t)t0 * CP), 0, nrows * CP * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sum = __builtin_amdgcn_make_buffer_rsrc(
        (void *)((P.sum_in ? P.sum_in : y_seg) + (P.sum_in ? (size_t)sg.row0 * jobs.rate * CP : 0) + (size_t)t0 * CP), 0, nrows * CP * 4, 0x00020000);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
    {
        const int oc = (wn * NT + nt) * 32 + 2 * lc;
        const float2 bias = *(const float2 *)(P.b2 + oc);
        const int voff = ((wm * 32 * MT + 8 * lg) * CP + oc) * 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
        {
            u32x2 resv[2][4];
#pragma unroll
            for (int lt = 0; lt < 2; lt++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    resv[lt][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff, (mt * 32 + 2 * i + lt) * CP * 4, 0);
            u32x2 sumv[2][4];
            if (!MERGE && P.sum_out && P.sum_in)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        sumv[lt][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_sum, voff, (mt * 32 + 2 * i + lt) * CP * 4, 0);
#pragma unroll
            for (int lt = 0; lt < 2; lt++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    float v0 = (acc[mt][nt][0][lt][i] + bias.x) + __uint_as_float(resv[lt][i][0]);
                    float v1 = (acc[mt][nt][1][lt][i] + bias.y) + __uint_as_float(resv[lt][i][1]);
                    if constexpr (MERGE)
                    {
                        msum[mt][nt][0][lt][i] = jb == 0 ? v0 : msum[mt][nt][0][lt][i] + v0;
                        msum[mt][nt][1][lt][i] = jb == 0 ? v1 : msum[mt][nt][1][lt][i] + v1;
                        if (jb != jobs.njobs - 1) continue;
                        v0 = msum[mt][nt][0][lt][i];
                        v1 = msum[mt][nt][1][lt][i];
                    }
                    else if (P.sum_out && P.sum_in)
                    {
                        // this branch's term of the MRF sum: sum_out = sum_in + v (the first branch stores v itself)
                        v0 = __uint_as_float(sumv[lt][i][0]) + v0;
                        v1 = __uint_as_float(sumv[lt][i][1]) + v1;
                    }
                    const u32x2 o = {__float_as_uint(v0), __float_as_uint(v1)};
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, (mt * 32 + 2 * i + lt) * CP * 4, ZV_ST_AUX);
                }
        }
    }
    }
}

#ifdef ZV_STAMPS
extern "C" int zv_debug_read_stamps(unsigned long long *out, size_t n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(zv_stamp_buf), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// ---------------------------------------------------------------------------------------------------
// The fused dilation pair of the 64-channel stage for batches: weights through an LDS ring.
// In resblock_pair_kernel<64> every wave streams its own copy of every B fragment from L1 (1 KiB per two MFMAs per wave,
// the two row halves of a workgroup fetching the same fragments twice): at the matrix pipe's full rate that alone is the
// whole 64 B/clk of the CU's vector-memory path, which the staging loads, the residual loads and the stores share — the
// MFMA loops run at 40 % of the pipe's rate (measured: 0.87 ms per launch against 0.36 ms of MFMA time).  Here
//   * a workgroup is 4 waves x (64 rows x all 64 channels) = 256 rows (halo recompute 1.04 instead of 1.08 at 11 taps),
//   * the weights of both convs travel global -> LDS once per workgroup as one stream of 2K chunks (one tap = 8 fragments
//     = 8 KiB each) through a ring of four slots, by LDS-DMA, two chunks ahead of the MFMAs (one workgroup barrier per
//     tap; the stream keeps running under the xt pack),
//   * both MFMA operands are ds_read_b128s two steps ahead; one A fragment feeds two MFMAs and so does one B fragment.
// Vector-memory bytes per output row fall 5x.  Same operations in the same order per output element as
// resblock_pair_kernel (tap-major, 16 channels per step), hence the same bits.
// Weight layout (pack_pair_weight_ring): [tap][kc][ntile][lane][8 halfs].
size_t pair_ring_weight_halfs(int Cp, int K) { return (size_t)K * (Cp / 16) * (Cp / 32) * 512; }

// the ring stream in v_mfma_f32_16x16x32_f16 fragment order (resblock_pair64_kernel / resblock_block64_kernel): [tap][step of 32
// channels][ntile][wt][lane][8 halfs]; conv2_layout as in pack_pair_weight16
void pack_pair_weight_ring(const uint16_t *w, int K, int C, int Cp, uint16_t *dst, bool conv2_layout)
{
    const int nk2 = Cp / 32, nnt = Cp / 32;
    for (int tap = 0; tap < K; tap++)
        for (int k2 = 0; k2 < nk2; k2++)
            for (int nt = 0; nt < nnt; nt++)
                for (int wt = 0; wt < 2; wt++)
                {
                    uint16_t *d = dst + ((((size_t)(tap * nk2 + k2) * nnt + nt) * 2) + wt) * 512;
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++)
                        {
                            const int r = lane & 15, g = lane >> 4;
                            const int oc = nt * 32 + (conv2_layout ? 2 * r + wt : 16 * wt + r), ic = k2 * 32 + 8 * g + j;
                            d[lane * 8 + j] = (oc < C && ic < C) ? w[((size_t)oc * C + ic) * K + tap] : (uint16_t)0;
                        }
                }
}

template <bool MERGE>
__global__ __launch_bounds__(256, 2) void resblock_pair64_kernel(const PairJobs jobs)
{
    constexpr int CP = 64, MT = 2, NT = 2, BM = 256, RS = CP * 2 + 16;
    constexpr int CHUNK = 8 * 1024;                  // one tap: 4 channel steps x 2 output tiles
    const int jz = (int)blockIdx.z, bx = (int)blockIdx.x;
    const int TM = BM - (MERGE ? jobs.kmax - 1 : jobs.j[jz].K - 1);
    const int tps = (jobs.segs.max_rows * jobs.rate + TM - 1) / TM;
    const int vt = zv_xcd_tile(bx, tps * jobs.segs.nseg);
    if (vt >= tps * jobs.segs.nseg) return;
    const int useg = vt / tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int t0 = (vt - useg * tps) * TM;
    if (t0 >= L) return;

#ifdef ZV_STAMPS
    const int stamp_wg = (int)(blockIdx.x + gridDim.x * blockIdx.z);
    if (jobs.stamp && threadIdx.x == 0 && stamp_wg < ZV_STAMP_WGS)
    {
        zv_stamp_buf[(size_t)stamp_wg * ZV_STAMP_N + 8] = __builtin_amdgcn_s_getreg(63492);      // HW_ID
        zv_stamp_buf[(size_t)stamp_wg * ZV_STAMP_N + 9] = __builtin_amdgcn_s_getreg(63508);      // XCC_ID
    }
#endif
    ZV_STAMP(0)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char *ring = smem + jobs.ring_off;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lg = lane >> 4;
    const char *abase = smem + (wave * 32 * MT + 2 * lc) * RS + lg * 16;      // block row 2c (+ lt), k group g (see resblock_pair_kernel)
    const char *bl = ring + lane * 16;

    floatx4 msum[MERGE ? MT : 1][MERGE ? NT : 1][2][2];
    for (int jb = MERGE ? 0 : jz; jb < (MERGE ? jobs.njobs : jz + 1); jb++)
    {
        const PairJob &P = jobs.j[jb];
        const int K = P.K, dil = P.dil;
        const int h2 = (K - 1) / 2, h1 = h2 * dil;
        const float *y_seg = P.y + (size_t)sg.row0 * jobs.rate * CP;
        float *out_seg = (MERGE ? jobs.merge_out : P.out) + (size_t)sg.row0 * jobs.rate * CP;
        const int nchunk = 2 * K;
        // chunk g of the pair's weight stream (conv1's taps, then conv2's) -> ring slot g & 3; a wave moves 2 of its 8 fragments
        // (the stream's base pointers pinned in scalar registers: re-reading them from the kernel arguments at every request
        // would put an lgkmcnt(0) wait — which also waits for the LDS reads in flight — into every tap)
        const uint64_t w1a = (uint64_t)P.w1r + wave * 2048, w2a = (uint64_t)P.w2r + wave * 2048 - (uint64_t)K * CHUNK;
        const uint32_t w1lo = __builtin_amdgcn_readfirstlane((uint32_t)w1a), w1hi = __builtin_amdgcn_readfirstlane((uint32_t)(w1a >> 32));
        const uint32_t w2lo = __builtin_amdgcn_readfirstlane((uint32_t)w2a), w2hi = __builtin_amdgcn_readfirstlane((uint32_t)(w2a >> 32));
        auto issue = [&](int g) {
            const uint64_t base = g < K ? ((uint64_t)w1hi << 32 | w1lo) : ((uint64_t)w2hi << 32 | w2lo);
            const char *src = (const char *)base + (size_t)g * CHUNK + lane * 16;
            char *dst = ring + (g & 3) * CHUNK + wave * 2048;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 1024), (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
        };
        if (MERGE && jb) __syncthreads();               // the previous job's conv2 is done reading the tile and the ring
        issue(0);
        issue(1);
        issue(2);

        // ---- stage X: LDS row r <-> time t0 - h2 - h1 + r
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y_seg, 0, L * CP * 4, 0x00020000);
        // (two workgroups per CU whatever the register count — LDS decides — so every staging load of the tile is in flight at once)
        stage_act_buf<20, CP>(rs_y, smem, t0 - h2 - h1, BM + 2 * h1 + dil, tid, P.slope);
        // the residual operand (the tile's centre rows again, in the accumulator layout) is requested right behind the
        // staging loads, while their lines are still in L2, and waits in registers until the epilogue
        const int nrows = (L - t0 < TM) ? (L - t0) : TM;
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(y_seg + (size_t)t0 * CP), 0, nrows * CP * 4, 0x00020000);
        // conv2's accumulator layout: [mt][nt][wt][lt][i] = time row mt*32 + 8g + 2i + lt, channel nt*32 + 2c + wt — 8 bytes per lane
        const int voff0 = ((wave * 32 * MT + 8 * lg) * CP + 2 * lc) * 4;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 resv[MT][NT][2][4];
        auto load_res = [&]() {
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            resv[mt][nt][lt][i] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, voff0 + nt * 128, (mt * 32 + 2 * i + lt) * CP * 4, 0);
        };
        if constexpr (!MERGE) load_res();      // (the merged form holds the branches' running sum: it loads in the epilogue)
        ZV_STAMP(1)
        __syncthreads();                                // X complete; the barrier drains the first three chunks too
        ZV_STAMP(2)

        floatx4 acc[MT][NT][2][2];
        half8 a[2][MT][2], b[2][NT][2];
        int g = 0;                                      // chunk = tap of the stream
        // one step = 32 channels: the operand's two 16-row tiles per 32-row block (rows 2c + lt), the weights' two 16-channel tiles per
        // output tile, both from LDS; a chunk = [step 2][nt 2][wt 2] fragments
#define ZV_LDR(slot, aptr, boff)                                                                                      \
    {                                                                                                                 \
        const char *ap_ = (aptr);                                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                                             \
            _Pragma("unroll") for (int lt = 0; lt < 2; lt++) a[slot][mt][lt] = *(const half8 *)(ap_ + (mt * 32 + lt) * RS);    \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                             \
            _Pragma("unroll") for (int wt = 0; wt < 2; wt++) b[slot][nt][wt] = *(const half8 *)(bp_ + (boff) + (nt * 2 + wt) * 1024); \
    }
#define ZV_MF(slot, SW, Z)                                \
    mfma16_step<MT, NT, SW, Z>(acc, a[slot], b[slot]);    \
    __builtin_amdgcn_sched_barrier(0);
        // one tap: step 0 | wait for the next chunk, barrier, request the chunk three ahead | step 1 (which already reads the
        // next tap's first fragments)
#define ZV_TAP(SW, Z0, tapstride)                                                                         \
    {                                                                                                     \
        const char *bp_ = bl + (g & 3) * CHUNK, *bn_ = bl + ((g + 1) & 3) * CHUNK;                        \
        ZV_LDR(1, ap + 64, 4 * 1024) ZV_MF(0, SW, Z0)                                                     \
        if (g + 2 < nchunk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                              \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                     \
        if (g + 3 < nchunk) issue(g + 3);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        ap += (tapstride);                                                                                \
        bp_ = bn_;                                                                                        \
        ZV_LDR(0, ap, 0) ZV_MF(1, SW, false)                                                              \
        g++;                                                                                              \
    }
        // ---- conv1 (dilated), transposed product
        {
            const char *ap = abase;
            {
                const char *bp_ = bl;
                ZV_LDR(0, ap, 0)
            }
            ZV_TAP(true, true, dil * RS)
            for (int tap = 1; tap < K; tap++) ZV_TAP(true, false, dil * RS)
        }
        ZV_STAMP(3)
        // every wave is done reading X: its LDS region becomes XT (raw barriers here: __syncthreads would drain the weight
        // stream, whose next chunks are in flight under the pack)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        ZV_STAMP(4)
        // ---- xt = f16(lrelu(conv1 + b1)), zero outside [0, L); XT row i <-> time t0 - h2 + i.  acc[mt][nt][wt][lt][i]: time row
        // mt*32 + 2c + lt, channel nt*32 + 16 wt + 4g + i
        {
            const float sl = P.slope;
            const bool edge = t0 - h2 < 0 || t0 - h2 + BM > L;
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
            {
                float4 bq[2];
#pragma unroll
                for (int wt = 0; wt < 2; wt++) bq[wt] = *(const float4 *)(P.b1 + nt * 32 + 16 * wt + 4 * lg);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
                    {
                        const int i = wave * 32 * MT + mt * 32 + 2 * lc + lt;
                        const int t = t0 - h2 + i;
                        const bool in = !edge || (t >= 0 && t < L);
#pragma unroll
                        for (int wt = 0; wt < 2; wt++)
                        {
                            uint2 pk = lrelu4_f16(acc[mt][nt][wt][lt][0] + bq[wt].x, acc[mt][nt][wt][lt][1] + bq[wt].y,
                                                  acc[mt][nt][wt][lt][2] + bq[wt].z, acc[mt][nt][wt][lt][3] + bq[wt].w, sl);
                            if (edge)
                            {
                                pk.x = in ? pk.x : 0u;
                                pk.y = in ? pk.y : 0u;
                            }
                            *(uint2 *)(smem + i * RS + (nt * 32 + 16 * wt + 4 * lg) * 2) = pk;
                        }
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        ZV_STAMP(5)
        // ---- conv2 (dil 1): output row j <-> time t0 + j reads XT rows j .. j + 2*h2; rows j >= TM are discarded
        {
            const char *ap = abase;
            {
                const char *bp_ = bl + (g & 3) * CHUNK;
                ZV_LDR(0, ap, 0)
            }
            ZV_TAP(false, true, RS)
            for (int tap = 1; tap < K; tap++) ZV_TAP(false, false, RS)
        }
#undef ZV_TAP
#undef ZV_MF
#undef ZV_LDR

        // ---- epilogue: out = y + (conv2 + b2); descriptors over exactly this tile's valid rows (see resblock_pair_kernel)
        ZV_STAMP(6)
        if constexpr (MERGE) load_res();
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)(out_seg + (size_t)t0 * CP), 0, nrows * CP * 4, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
        {
            const float2 bias = *(const float2 *)(P.b2 + nt * 32 + 2 * lc);
            const int voff = voff0 + nt * 128;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        float v0 = (acc[mt][nt][0][lt][i] + bias.x) + __uint_as_float(resv[mt][nt][lt][i][0]);
                        float v1 = (acc[mt][nt][1][lt][i] + bias.y) + __uint_as_float(resv[mt][nt][lt][i][1]);
                        if constexpr (MERGE)
                        {
                            msum[mt][nt][0][lt][i] = jb == 0 ? v0 : msum[mt][nt][0][lt][i] + v0;
                            msum[mt][nt][1][lt][i] = jb == 0 ? v1 : msum[mt][nt][1][lt][i] + v1;
                            if (jb != jobs.njobs - 1) continue;
                            v0 = msum[mt][nt][0][lt][i];
                            v1 = msum[mt][nt][1][lt][i];
                        }
                        const u32x2 o = {__float_as_uint(v0), __float_as_uint(v1)};
                        __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, (mt * 32 + 2 * i + lt) * CP * 4, ZV_ST_AUX);
                    }
        }
#ifdef ZV_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        ZV_STAMP(7)
    }
}

template <bool MERGE>
static hipError_t launch_pair64_ring(hipStream_t s, PairJobs &js, int njobs, int Lmax, int Kmax, int dmax)
{
    constexpr int BM = 256;
    const int TMmin = BM - (Kmax - 1);
    dim3 grid(round_up(((Lmax + TMmin - 1) / TMmin) * js.segs.nseg, 8), 1, MERGE ? 1 : njobs);
    // operand rows: BM + (K - 1) * dil, + dil: the last tap's prefetch reads one tap past the end
    js.ring_off = round_up((BM + Kmax * dmax) * (64 * 2 + 16), 1024);
    const size_t lds = (size_t)js.ring_off + 4 * 8192;
    if (lds > 80 * 1024) return hipErrorInvalidValue;
    auto kern = resblock_pair64_kernel<MERGE>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, js);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// resblock_block64_kernel — SEVERAL dilation pairs of a 64-channel residual block in one launch, for the branches whose halo is
// small (3 taps: 2 rows per pair and dilation step): resblock_pair64_kernel's machinery (weights of every conv through the
// four-slot LDS ring as ONE stream over all the pairs, both operands from LDS, 64 x 64 per wave) around resblock_block32_kernel's
// data flow (the f32 tile stays in registers in the accumulator layout — it is the residual operand of every pair —, the f16
// operand tile is regenerated in LDS per pair, every conv runs over the full 256-row tile as a zero-padded sequence and only
// the TM = 256 - 2H centre rows are stored).  The branch's tensor crosses HBM once per block instead of once per pair.
// Same operations in the same order per output element as the pair kernels: same bits.
__global__ __launch_bounds__(256, 2) void resblock_block64_kernel(const TripleJobs jobs)
{
    constexpr int CP = 64, MT = 2, NT = 2, BM = 256, RS = CP * 2 + 16;
    constexpr int CHUNK = 8 * 1024;
    const TripleJob &P = jobs.j[blockIdx.z];
    const int K = P.K, nd = P.n_dil;
    const int h2 = (K - 1) / 2;
    int sumd = 0, dmax = 1;
    for (int d = 0; d < nd; d++) { sumd += P.dil[d]; dmax = P.dil[d] > dmax ? P.dil[d] : dmax; }
    const int H = h2 * (sumd + nd);
    const int TM = BM - 2 * H;
    const int tps = (jobs.segs.max_rows * jobs.rate + TM - 1) / TM;
    const int vt = zv_xcd_tile(blockIdx.x, tps * jobs.segs.nseg);
    if (vt >= tps * jobs.segs.nseg) return;
    const int useg = vt / tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int t0 = (vt - useg * tps) * TM;
    if (t0 >= L) return;
    const float *y_seg = P.y + (size_t)sg.row0 * jobs.rate * CP;
    float *out_seg = P.out + (size_t)sg.row0 * jobs.rate * CP;
    const int XM = h2 * dmax;                         // zero margin of the operand region on either side of the tile
    const int xrows = BM + 2 * XM + 2 * dmax;         // + slack: the last tap's look-ahead reads one tap past the end
    const bool edge = t0 - H < 0 || t0 - H + BM > L;

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char *ring = smem + jobs.ring_off;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lg = lane >> 4;
    const char *abase = smem + (wave * 32 * MT + 2 * lc) * RS + lg * 16;      // block row 2c (+ lt), k group g (see resblock_pair_kernel)
    const char *bl = ring + lane * 16;
    const int nchunk = 2 * K * nd;                    // the block's weight stream: per pair conv1's taps, then conv2's
    // chunk g -> ring slot g & 3; base pointers of the (at most six) convs in scalar registers
    uint32_t wlo[2 * TRIPLE_MAX_DIL], whi[2 * TRIPLE_MAX_DIL];
#pragma unroll
    for (int c = 0; c < 2 * TRIPLE_MAX_DIL; c++)
    {
        const int d = c >> 1 < nd ? c >> 1 : 0;
        const uint64_t a = (uint64_t)((c & 1) ? P.w2[d] : P.w1[d]) + wave * 2048;
        wlo[c] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        whi[c] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    }
    auto issue = [&](int g) {
        const int c = g / K, tap = g - c * K;         // conv index (2 * pair + conv), tap
        uint32_t lo = wlo[0], hi = whi[0];
#pragma unroll
        for (int q = 1; q < 2 * TRIPLE_MAX_DIL; q++)
            if (c == q) { lo = wlo[q]; hi = whi[q]; }
        const char *src = (const char *)((uint64_t)hi << 32 | lo) + (size_t)tap * CHUNK + lane * 16;
        char *dst = ring + (g & 3) * CHUNK + wave * 2048;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 1024), (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
    };
    issue(0);
    issue(1);
    issue(2);
    // the whole operand region starts as zeros (margins stay zero for the whole kernel; nothing in it is ever uninitialised)
    for (int i = tid; i < xrows * RS / 16; i += 256) ((uint4 *)smem)[i] = make_uint4(0, 0, 0, 0);
    // tile row i <-> time t0 - H + i; register [mt][nt][wt][lt][i] (conv2's accumulator layout): row wave*64 + mt*32 + 8g + 2i + lt,
    // channel nt*32 + 2c + wt — the lane's two weight tiles are neighbouring channels: 8-byte loads / stores, 4-byte operand writes
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    float yreg[MT][NT][2][2][4];
    {
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y_seg, 0, L * CP * 4, 0x00020000);
        const int voff = ((t0 - H + wave * 32 * MT + 8 * lg) * CP + 2 * lc) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_y, voff + nt * 128 + (mt * 32 + 2 * i + lt) * CP * 4, 0, 0);
                        yreg[mt][nt][0][lt][i] = __uint_as_float(v[0]);
                        yreg[mt][nt][1][lt][i] = __uint_as_float(v[1]);
                    }
    }
    __syncthreads();                                  // zeros written (and the first chunks landed)
    const float sl = P.slope;
    int g = 0;
    floatx4 acc[MT][NT][2][2];
    half8 a[2][MT][2], b[2][NT][2];
#define ZV_LDR(slot, aptr, boff)                                                                                      \
    {                                                                                                                 \
        const char *ap_ = (aptr);                                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                                             \
            _Pragma("unroll") for (int lt = 0; lt < 2; lt++) a[slot][mt][lt] = *(const half8 *)(ap_ + (mt * 32 + lt) * RS);    \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                                                             \
            _Pragma("unroll") for (int wt = 0; wt < 2; wt++) b[slot][nt][wt] = *(const half8 *)(bp_ + (boff) + (nt * 2 + wt) * 1024); \
    }
#define ZV_MF(slot, SW, Z)                                \
    mfma16_step<MT, NT, SW, Z>(acc, a[slot], b[slot]);    \
    __builtin_amdgcn_sched_barrier(0);
#define ZV_TAP(SW, Z0, tapstride)                                                                         \
    {                                                                                                     \
        const char *bp_ = bl + (g & 3) * CHUNK, *bn_ = bl + ((g + 1) & 3) * CHUNK;                        \
        ZV_LDR(1, ap + 64, 4 * 1024) ZV_MF(0, SW, Z0)                                                     \
        if (g + 2 < nchunk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                              \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                     \
        if (g + 3 < nchunk) issue(g + 3);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        ap += (tapstride);                                                                                \
        bp_ = bn_;                                                                                        \
        ZV_LDR(0, ap, 0) ZV_MF(1, SW, false)                                                              \
        g++;                                                                                              \
    }
    for (int d = 0; d < nd; d++)
    {
        const int dil = P.dil[d], h1 = h2 * dil;
        // ---- X = f16(lrelu(Y)) into region rows XM .. XM + BM - 1 (Y is zero outside [0, L): so is X)
        {
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            const int row = wave * 32 * MT + mt * 32 + 8 * lg + 2 * i + lt;
                            half2v h;
                            h[0] = (_Float16)lrelu_max(yreg[mt][nt][0][lt][i], sl);
                            h[1] = (_Float16)lrelu_max(yreg[mt][nt][1][lt][i], sl);
                            *(half2v *)(smem + (XM + row) * RS + (nt * 32 + 2 * lc) * 2) = h;
                        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // (raw: the weight stream stays in flight)
        // ---- conv1 (dilated), transposed product; output tile row i reads region rows XM + i - h1 + tap * dil
        {
            const char *ap = abase + (XM - h1) * RS;
            {
                const char *bp_ = bl + (g & 3) * CHUNK;
                ZV_LDR(0, ap, 0)
            }
            ZV_TAP(true, true, dil * RS)
            for (int tap = 1; tap < K; tap++) ZV_TAP(true, false, dil * RS)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // every wave is done reading X: its rows become XT
        // ---- xt = f16(lrelu(conv1 + b1)), zero outside [0, L): acc[mt][nt][wt][lt][i] = time row mt*32 + 2c + lt, channel nt*32 + 16 wt + 4g + i
        {
            const float *b1 = P.b1[d];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
            {
                float4 bq[2];
#pragma unroll
                for (int wt = 0; wt < 2; wt++) bq[wt] = *(const float4 *)(b1 + nt * 32 + 16 * wt + 4 * lg);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
                    {
                        const int i = wave * 32 * MT + mt * 32 + 2 * lc + lt;
                        const int t = t0 - H + i;
                        const bool in = !edge || (t >= 0 && t < L);
#pragma unroll
                        for (int wt = 0; wt < 2; wt++)
                        {
                            uint2 pk = lrelu4_f16(acc[mt][nt][wt][lt][0] + bq[wt].x, acc[mt][nt][wt][lt][1] + bq[wt].y,
                                                  acc[mt][nt][wt][lt][2] + bq[wt].z, acc[mt][nt][wt][lt][3] + bq[wt].w, sl);
                            if (edge)
                            {
                                pk.x = in ? pk.x : 0u;
                                pk.y = in ? pk.y : 0u;
                            }
                            *(uint2 *)(smem + (XM + i) * RS + (nt * 32 + 16 * wt + 4 * lg) * 2) = pk;
                        }
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- conv2 (dil 1): output tile row i reads region rows XM + i - h2 + tap;  Y = Y + (conv2 + b2), 0 outside [0, L)
        {
            const char *ap = abase + (XM - h2) * RS;
            {
                const char *bp_ = bl + (g & 3) * CHUNK;
                ZV_LDR(0, ap, 0)
            }
            ZV_TAP(false, true, RS)
            for (int tap = 1; tap < K; tap++) ZV_TAP(false, false, RS)
        }
        {
            const float *b2 = P.b2[d];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
            {
                const float2 bias = *(const float2 *)(b2 + nt * 32 + 2 * lc);
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            const float v0 = (acc[mt][nt][0][lt][i] + bias.x) + yreg[mt][nt][0][lt][i];
                            const float v1 = (acc[mt][nt][1][lt][i] + bias.y) + yreg[mt][nt][1][lt][i];
                            bool in = true;
                            if (edge)
                            {
                                const int t = t0 - H + wave * 32 * MT + mt * 32 + 8 * lg + 2 * i + lt;
                                in = t >= 0 && t < L;
                            }
                            yreg[mt][nt][0][lt][i] = in ? v0 : 0.f;
                            yreg[mt][nt][1][lt][i] = in ? v1 : 0.f;
                        }
            }
        }
        if (d + 1 < nd)
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // every wave is done reading XT before the next X goes over it
        }
    }
#undef ZV_TAP
#undef ZV_MF
#undef ZV_LDR
    // ---- store the centre rows (tile rows H .. H + TM - 1, time < L): anything else gets an out-of-range offset
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)out_seg, 0, L * CP * 4, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int lt = 0; lt < 2; lt++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int row = wave * 32 * MT + mt * 32 + 8 * lg + 2 * i + lt;
                    const int t = t0 - H + row;
                    const int voff = (row >= H && row < H + TM && t >= 0) ? (t * CP + nt * 32 + 2 * lc) * 4 : -8;
                    const u32x2 o = {__float_as_uint(yreg[mt][nt][0][lt][i]), __float_as_uint(yreg[mt][nt][1][lt][i])};
                    __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, 0, ZV_ST_AUX);
                }
}

// the blocks resblock_block64_kernel takes: 64 channels, few taps (the halo of n_dil pairs leaves most of the 256-row tile)
bool block64_supported(int Cp, int K, const int *dil, int n_dil)
{
    if (Cp != 64 || K < 3 || (K & 1) == 0 || n_dil < 1 || n_dil > TRIPLE_MAX_DIL) return false;
    int sumd = 0;
    for (int d = 0; d < n_dil; d++) sumd += dil[d];
    return 256 - (K - 1) * (sumd + n_dil) >= 192;          // at least three quarters of the tile's rows are output
}

hipError_t launch_block64(hipStream_t s, const TripleJob *jobs, int njobs, const Segs &segs, int rate)
{
    if (njobs < 1 || njobs > PAIR_MAX_JOBS || segs.nseg < 1 || segs.max_rows < 1) return hipErrorInvalidValue;
    TripleJobs js;
    js.segs = segs;
    js.rate = rate;
    js.interleave = 1;
    js.db_mask = 0;
    const int Lmax = segs.max_rows * rate;
    int gx = 1, rows_max = 0;
    for (int i = 0; i < njobs; i++)
    {
        js.j[i] = jobs[i];
        js.j[i].dbg = 0;
        const TripleJob &P = jobs[i];
        if (!block64_supported(P.Cp, P.K, P.dil, P.n_dil)) return hipErrorInvalidValue;
        int sumd = 0, dmax = 1;
        for (int d = 0; d < P.n_dil; d++) { sumd += P.dil[d]; dmax = P.dil[d] > dmax ? P.dil[d] : dmax; }
        const int h2 = (P.K - 1) / 2, TM = 256 - 2 * h2 * (sumd + P.n_dil);
        gx = std::max(gx, ((Lmax + TM - 1) / TM) * segs.nseg);
        rows_max = std::max(rows_max, 256 + 2 * h2 * dmax + 2 * dmax);
    }
    for (int i = njobs; i < PAIR_MAX_JOBS; i++) js.j[i] = js.j[0];
    js.ring_off = round_up(rows_max * (64 * 2 + 16), 1024);
    const size_t lds = (size_t)js.ring_off + 4 * 8192;
    if (lds > 80 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)resblock_block64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(resblock_block64_kernel, dim3(round_up(gx, 8), 1, njobs), dim3(256), lds, s, js);
    return hipGetLastError();
}

// the MFMA loop of the fused kernels walks whole 8-step bodies (CP = 64: also half a body at the end) and at least one
bool pair_supported(int Cp, int K)
{
    if (!(Cp == 32 || Cp == 64 || Cp == 128 || Cp == 256) || K < 1 || (K & 1) == 0) return false;
    const int nsb = (K * (Cp / 16) + 3) >> 2;
    return nsb >= 2 && (Cp == 64 || (nsb & 1) == 0);
}

template <int CP, int MT, bool MERGE>
static hipError_t launch_pair_cfg(hipStream_t s, const PairJobs &js, int njobs, int Lmax, int Kmax, int dmax)
{
    constexpr int WNc = (CP == 256) ? 4 : CP / 32;
    constexpr int BM = 32 * MT * (4 / WNc);
    const int TMmin = BM - (Kmax - 1);
    if (TMmin < 32) return hipErrorInvalidValue;
    // jobs differ in K: grid.x is sized for the smallest TM, workgroups beyond a job's extent exit at once
    dim3 grid(round_up(((Lmax + TMmin - 1) / TMmin) * js.segs.nseg, 8), 1, MERGE ? 1 : njobs);      // multiple of 8: zv_xcd_tile
    // rows touched: BM + taps (K rounded up to the loop's granularity, + 1 for the last prefetch) * dil.  The loop walks
    // whole taps once a tap is at least a body (CP >= 128): K + 1 taps (51 KB for the 128-channel stage: room for three
    // workgroups per CU instead of two — measured worth 0.6 %)
    // (CP = 256: exactly K taps — the prefetch one tap past the end reads rows that exist but are never used — so that the
    // 96-row tile of MT = 3 stays under 80 KB: two workgroups per CU)
    size_t lds = (size_t)(BM + (Kmax + (CP == 256 ? 0 : (CP >= 128 ? 1 : 4))) * dmax) * (CP * 2 + 16);
#ifdef ZV_DIAG
    lds += (size_t)knob(ZV_LDS_PAD);       // diagnostic build: extra bytes of LDS per workgroup = a lower occupancy on purpose
#endif
    auto kern = resblock_pair_kernel<CP, MT, MERGE>;
    if (lds > 64 * 1024)
    {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, js);
    return hipGetLastError();
}

hipError_t launch_pair(hipStream_t s, const PairJob *jobs, int njobs, int n_cu, const Segs &segs, int rate, float *merge_out)
{
    if (njobs < 1 || njobs > PAIR_MAX_JOBS || segs.nseg < 1 || segs.max_rows < 1) return hipErrorInvalidValue;
    const int dbg = diag_bits(), mt_env = knob(ZV_PAIR_MT);
    PairJobs js;
    js.segs = segs;
    js.rate = rate;
    js.njobs = njobs;
    js.merge_out = merge_out;
#ifdef ZV_STAMPS
    js.stamp = knob(ZV_STAMP_CP) && knob(ZV_STAMP_CP) == jobs[0].Cp && !merge_out;
#endif
    const int Lmax = segs.max_rows * rate;
    int Kmax = 0, dmax = 0;
    for (int i = 0; i < njobs; i++)
    {
        js.j[i] = jobs[i];
        js.j[i].dbg = dbg;
        if (jobs[i].Cp != jobs[0].Cp) return hipErrorInvalidValue;
        Kmax = jobs[i].K > Kmax ? jobs[i].K : Kmax;
        dmax = jobs[i].dil > dmax ? jobs[i].dil : dmax;
    }
    for (int i = njobs; i < PAIR_MAX_JOBS; i++) js.j[i] = js.j[0];
    // the running MRF sum (PairJob::sum_in / sum_out) exists only in resblock_pair_kernel<CP, MT, false>: not together with the
    // merged form, and not on the 64-channel ring kernel (which would silently write P.out instead)
    bool any_sum = false;
    for (int i = 0; i < njobs; i++) any_sum = any_sum || jobs[i].sum_out || jobs[i].sum_in;
    if (any_sum && merge_out) return hipErrorInvalidValue;
    const int Cp = jobs[0].Cp;
    const int WN = Cp == 256 ? 4 : Cp / 32;
    auto wgs = [&](int MT) {
        const int BM = 32 * MT * (4 / WN);
        const int TM = BM - (Kmax - 1);
        return TM < 32 ? 0L : (long)((Lmax + TM - 1) / TM) * segs.nseg * njobs;
    };
    // tallest tile that still gives every CU about three workgroups, but never a BM so small that the
    // (k-1)-row halo dominates (MT >= 2: BM >= 64 / 128 / 256 for 128 / 64 / 32 channels)
    // measured (512 frames, batch 1): BM = 64/128/256 rows (MT = 2) beats MT = 4 at every channel count — three
    // to four workgroups per CU hide the staging / epilogue phases better than taller tiles save weight traffic
    // measured (batch of 32 x 1 024 frames): once a launch has many rounds of workgroups the 128-channel stage is
    // bound by the weight stream from L2 (1 KiB of B fragment per 2 MFMAs per wave at MT = 2) and BM = 128 is 14 % faster;
    // the 64-channel stage does not care (-1 %).  The tile height never changes an output bit.
    // 64 channels, batches: the form with the weights through an LDS ring (ZV_PAIR64_RING = 0 never, 2 whenever it fits)
    {
        const int ring_env = knob(ZV_PAIR64_RING);
        bool ok = Cp == 64 && !any_sum && ring_env != 0 && Kmax >= 3 && (256 + Kmax * dmax) * 144 + 4 * 8192 + 1024 <= 80 * 1024;
        for (int i = 0; i < njobs && ok; i++) ok = jobs[i].w1r && jobs[i].w2r;
        const long rwgs = (long)((Lmax + 256 - Kmax) / (257 - Kmax)) * segs.nseg * njobs;
        if (ok && (ring_env == 2 || rwgs >= 6L * n_cu))
        {
            js.kmax = Kmax;
            return merge_out ? launch_pair64_ring<true>(s, js, njobs, Lmax, Kmax, dmax) : launch_pair64_ring<false>(s, js, njobs, Lmax, Kmax, dmax);
        }
    }
    int MT = (Cp == 128 && wgs(4) >= 8L * n_cu) ? 4 : 2;
    // 256 channels, batches: 96-row tiles (two thirds of the weight-fragment traffic per row, 10 instead of 16 % of conv2 spent
    // on halo rows at 11 taps; 80 KB of LDS and 234 registers still give two workgroups per CU): 1 010 -> 897 us per launch.
    // (The merged form would need 330 registers.)
    if (Cp == 256 && !merge_out && wgs(3) >= 4L * n_cu) MT = 3;
    if (mt_env == 2 || mt_env == 4 || (mt_env == 3 && Cp == 256 && !merge_out)) MT = mt_env;      // (the merged form of MT = 3 needs 330 registers)
    js.kmax = Kmax;
#define ZV_PCASE(cp, mt)                                                                                 \
    if (Cp == cp && MT == mt)                                                                            \
        return merge_out ? launch_pair_cfg<cp, mt, true>(s, js, njobs, Lmax, Kmax, dmax)                 \
                         : launch_pair_cfg<cp, mt, false>(s, js, njobs, Lmax, Kmax, dmax);
    ZV_PCASE(32, 4) ZV_PCASE(32, 2) ZV_PCASE(64, 4) ZV_PCASE(64, 2) ZV_PCASE(128, 4) ZV_PCASE(128, 2) ZV_PCASE(256, 2) ZV_PCASE(256, 3) ZV_PCASE(256, 4)
#undef ZV_PCASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// Whole residual block in one launch (reference src/hifigan.cpp:99-182, all iterations of the dilation loop).
// The narrow stages are HBM-bound: per dilation pair the fused kernel above reads y (+ halo) and writes y, i.e. the
// block's three pairs move the tile through HBM three times.  Here a workgroup loads a tile of R = 256 rows once
// (f32, in registers) and for every dilation d
//     X  = f16(lrelu(Y))                    (registers -> LDS)
//     xt = f16(lrelu(conv(X, k, d) + b1))   (MFMA, weights as the A operand; packed over X like in the pair kernel)
//     Y += conv(xt, k, 1) + b2              (MFMA; accumulated into the f32 tile)
// All convs run over the full tile as if it were a zero-padded sequence of R rows, so after pair d the rows within
// the cumulative halo of a tile edge are wrong — they never reach the TM = R - 2*H centre rows that are stored
// (H = h2 * (sum(dil) + n_dil), h2 = (k-1)/2).  Rows outside [0, L) are forced to zero after every pair: they are the
// reference's zero padding, not computed values.  Same operations in the same order per output element as the pair
// kernel, hence the same bits.
// LDS: X/XT region rows [0, R + 2*XM + slack), row XM + i <-> tile row i (XM = h2 * max(dil): zero margins, written
// once).  The f32 tile itself stays in registers (see below).
template <int CP, int MT, int R>
__global__ __launch_bounds__(64 * (R / 32 / MT)) void resblock_triple_kernel(const TripleJobs jobs)
{
    // 8 / MT waves, each owns 32*MT tile rows x all CP (= 32) channels; its slice of the f32 tile Y lives in 16*MT
    // registers per lane in the MFMA accumulator layout (row = (r&3) + 8*(r>>2) + 4*(lane>>5), channel = lane & 31)
    // for the whole kernel, so LDS only holds the f16 operand tile (26 KiB: several workgroups per CU).  Every wave
    // streams the same weight fragments (one output tile), MT row tiles per fragment.
    constexpr int NWV = R / 32 / MT, NTH = 64 * NWV;
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    static_assert(CP == 32, "one 32-channel output tile per wave");
    const TripleJob &P = jobs.j[blockIdx.z];
    const int K = P.K, nd = P.n_dil;
    const int h2 = (K - 1) / 2;
    int sumd = 0, dmax = 1;
    for (int d = 0; d < nd; d++) { sumd += P.dil[d]; dmax = P.dil[d] > dmax ? P.dil[d] : dmax; }
    const int H = h2 * (sumd + nd);
    const int TM = R - 2 * H;
    const int tps = (jobs.segs.max_rows * jobs.rate + TM - 1) / TM;      // (segment, tile) as in resblock_pair_kernel
    const int vt = zv_xcd_tile(blockIdx.x, tps * jobs.segs.nseg);
    if (vt >= tps * jobs.segs.nseg) return;
    const int useg = vt / tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int t0 = (vt - useg * tps) * TM;
    if (t0 >= L) return;
    const float *y_seg = P.y + (size_t)sg.row0 * jobs.rate * CP;
    float *out_seg = P.out + (size_t)sg.row0 * jobs.rate * CP;
    const int XM = h2 * dmax;
    // only the tiles at a segment's ends hold rows outside [0, L): the others skip every range mask
    const bool edge = t0 - H < 0 || t0 - H + R > L;
    const int xrows = R + 2 * XM + 5 * dmax;          // + slack: zero-weight taps and the last A prefetch read past the margin

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t wseg = (size_t)(round_up(K * NKC, 4) + 8) * 64;
    const int col = lane & 31;
    const int irow0 = wave * 32 * MT + 4 * (lane >> 5);   // tile row of register [mt][r]: irow0 + mt*32 + (r&3) + 8*(r>>2)

    // ---- zero the X region (its margins stay zero for the whole kernel); load this lane's slice of the tile:
    // tile row i <-> time t0 - H + i, rows outside [0, L) are out of the descriptor's range and read as 0
    for (int i = tid; i < xrows * RS / 16; i += NTH) ((uint4 *)smem)[i] = make_uint4(0, 0, 0, 0);
    float yreg[MT][16];
    {
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y_seg, 0, L * CP * 4, 0x00020000);
        const int voff = ((t0 - H + irow0) * CP + col) * 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                yreg[mt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_y, voff + (mt * 32 + (r & 3) + 8 * (r >> 2)) * CP * 4, 0, 0));
    }

    const char *abase = smem + (wave * 32 * MT + (lane & 31)) * RS + (lane >> 5) * 16;
    const float sl = P.slope;
    half8 bw[4][1];
    deep_preload_b<1>(bw, (const half8 *)P.w1[0] + lane, wseg);            // the first conv's fragments travel under the tile load
    for (int d = 0; d < nd; d++)
    {
        const int dil = P.dil[d], h1 = h2 * dil;
        __syncthreads();                       // zero fill done / previous conv2 done reading XT
        // ---- X = f16(lrelu(Y)) into region rows XM .. XM + R - 1
        {
            char *xp = smem + (XM + irow0) * RS + col * 2;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    *(_Float16 *)(xp + (mt * 32 + (r & 3) + 8 * (r >> 2)) * RS) = (_Float16)lrelu_max(yreg[mt][r], sl);
        }
        __syncthreads();

        // ---- conv1 (dilated), transposed product; output tile row i reads region rows XM + i - h1 + tap*dil
        floatx16 acc[MT][1];
        if (ZV_DBGBITS(P.dbg) & 2)      // timing ablation only: the MFMA loops (which start the accumulators from 0) are skipped
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][0][r] = 0.f;
        if (!(ZV_DBGBITS(P.dbg) & 2))
            mfma_taps_deep<CP, MT, 1, true>(acc, abase + (XM - h1) * RS, dil * RS, (const half8 *)P.w1[d] + lane, wseg, K, bw);
        deep_preload_b<1>(bw, (const half8 *)P.w2[d] + lane, wseg);                 // under the xt pack
        __syncthreads();                       // every wave is done reading X: the region becomes XT
        {
            const int hh = lane >> 5;
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *(const float4 *)(P.b1[d] + 8 * q + 4 * hh);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
            {
                const int i = wave * 32 * MT + mt * 32 + (lane & 31);
                const int t = t0 - H + i;
                const bool in = !edge || (t >= 0 && t < L);
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    half4 h;
                    h[0] = (_Float16)lrelu_max(acc[mt][0][4 * q + 0] + bq[q].x, sl);
                    h[1] = (_Float16)lrelu_max(acc[mt][0][4 * q + 1] + bq[q].y, sl);
                    h[2] = (_Float16)lrelu_max(acc[mt][0][4 * q + 2] + bq[q].z, sl);
                    h[3] = (_Float16)lrelu_max(acc[mt][0][4 * q + 3] + bq[q].w, sl);
                    uint2 pk = *(uint2 *)&h;
                    if (edge)
                    {
                        pk.x = in ? pk.x : 0u;
                        pk.y = in ? pk.y : 0u;
                    }
                    *(uint2 *)(smem + (XM + i) * RS + (8 * q + 4 * hh) * 2) = pk;
                }
            }
        }
        __syncthreads();

        // ---- conv2 (dil 1): output tile row i reads region rows XM + i - h2 + tap;  Y = Y + (conv2 + b2), 0 outside [0, L)
        if (!(ZV_DBGBITS(P.dbg) & 2))
            mfma_taps_deep<CP, MT, 1, false>(acc, abase + (XM - h2) * RS, RS, (const half8 *)P.w2[d] + lane, wseg, K, bw);
        if (d + 1 < nd) deep_preload_b<1>(bw, (const half8 *)P.w1[d + 1] + lane, wseg);      // under the epilogue and the next X write
        {
            const float bias = P.b2[d][col];
            if (edge)
            {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                    {
                        const int t = t0 - H + irow0 + mt * 32 + (r & 3) + 8 * (r >> 2);
                        const float v = (acc[mt][0][r] + bias) + yreg[mt][r];
                        yreg[mt][r] = (t >= 0 && t < L) ? v : 0.f;
                    }
            }
            else
            {
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) yreg[mt][r] = (acc[mt][0][r] + bias) + yreg[mt][r];
            }
        }
    }

    // ---- store the centre rows (tile rows H .. H + TM - 1, time < L): anything else gets an out-of-range offset
    if (ZV_DBGBITS(P.dbg) & 4) return;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)out_seg, 0, L * CP * 4, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int i = irow0 + mt * 32 + (r & 3) + 8 * (r >> 2);
            const int t = t0 - H + i;
            const int voff = (i >= H && i < H + TM && t >= 0) ? (t * CP + col) * 4 : -4;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yreg[mt][r]), rs_out, voff, 0, ZV_ST_AUX);
        }
}

// ---------------------------------------------------------------------------------------------------
// The whole-block kernel, second form (batches): the conv weights go through LDS.
// In the kernel above every wave streams its own copy of every B fragment from L1/L2 (1 KiB per two MFMAs per wave: the
// CU's vector-memory path is the busiest unit of the launch) through a 32-register ring, which puts the kernel at 150
// registers: ONE 8-wave workgroup per CU, whose waves sit in the same phase between the same barriers, so the matrix
// pipe idles during every pack / epilogue phase and the vector ALUs during every MFMA loop.  Here the 8 ... 24 KiB of a
// conv's fragments are copied global -> LDS once per workgroup (LDS-DMA, one 1-KiB fragment per wave-instruction, no
// registers) while the waves write the operand tile, both MFMA operands are ds_read_b128s two steps ahead, and the
// kernel fits 128 registers: two workgroups per CU, one in its MFMA loop while the other packs.  Same operations in the
// same order per output element as resblock_triple_kernel / resblock_pair_kernel, hence the same bits.
__device__ __forceinline__ void dma_weights32(const void *wsrc, char *wlds, int nblk, int wave, int lane, int nwv)
{
    for (int blk = wave; blk < nblk; blk += nwv)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)wsrc + blk * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(wlds + blk * 1024), 16, 0, 0);
}

// (A variant with Y in the transposed accumulator layout — both convs with the weights as the A operand, 8-byte operand
// writes, 16-byte tile loads / stores: half the vector instructions — measured 7 % SLOWER: a 16-byte access per lane in
// that layout touches 32 rows x 32 bytes per instruction, against 2 rows x 128 bytes for the 4-byte column accesses.)
// LDS stores the compiler does not see as LDS stores: hipcc orders every visible LDS access behind an LDS-DMA in flight with
// s_waitcnt vmcnt(0), which would drain the weight stream at the first operand write of every phase.  The hazards these
// stores do have (against the other waves' reads) are covered by the kernel's own barriers and lgkmcnt waits.
template <int OFF>
__device__ __forceinline__ void lds_st_b64(unsigned addr, uint2 v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// ---- the whole-block kernel on v_mfma_f32_16x16x32_f16 (round 4; layouts as in resblock_pair_kernel, same bits) ----
// one conv of the 32-channel block: one step per tap (32 channels), operand rows (two 16-row tiles per 32-row block: rows 2c + lt)
// and weight fragments (two 16-channel tiles) from LDS, one tap ahead of the MFMAs that consume them
template <int MT, bool SWAP>
__device__ __forceinline__ void mfma32x_ldsw(floatx4 (&acc)[MT][1][2][2], const char *ap, int dilRS, const char *wl, int K)
{
    constexpr int RS = 32 * 2 + 16;
    half8 a[2][MT][2], b[2][1][2];
#define ZV_LD(slot, aptr, woff)                                                                                       \
    {                                                                                                                 \
        const char *ap_ = (aptr);                                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++)                                                             \
            _Pragma("unroll") for (int lt = 0; lt < 2; lt++) a[slot][mt][lt] = *(const half8 *)(ap_ + (mt * 32 + lt) * RS);    \
        _Pragma("unroll") for (int wt = 0; wt < 2; wt++) b[slot][0][wt] = *(const half8 *)(wl + (woff) + wt * 1024);   \
    }
#define ZV_ST(slot, Z)                                     \
    mfma16_step<MT, 1, SWAP, Z>(acc, a[slot], b[slot]);    \
    __builtin_amdgcn_sched_barrier(0);
    // tap 0 starts the accumulators from the constant 0; the other K - 1 taps (K odd) go two at a time.  The last iteration's
    // look-ahead reads one tap past the end (operand rows and weight-buffer slack that exist) and is never used.
    const char *t1 = ap + dilRS;
    ZV_LD(0, ap, 0)
    ZV_LD(1, t1, 2 * 1024) ZV_ST(0, true)
    for (int it = (K - 1) >> 1; it > 0; it--)
    {
        const char *t2 = t1 + dilRS, *t3 = t2 + dilRS;
        ZV_LD(0, t2, 4 * 1024) ZV_ST(1, false)
        ZV_LD(1, t3, 6 * 1024) ZV_ST(0, false)
        t1 = t3;
        wl += 4 * 1024;
    }
#undef ZV_ST
#undef ZV_LD
}

template <int OFF>
__device__ __forceinline__ void lds_st_b32(unsigned addr, unsigned v)
{
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// the operand writes (two neighbouring channels per lane and row: 4-byte stores) and the xt pack, unrolled at compile time
// (y[mt][wt][lt] holds four rows of one channel: the slope multiply goes two rows at a time, v_pk_mul_f32)
template <int MT_, int I = 0>
__device__ __forceinline__ void xwrite_all16(unsigned xa, const floatx4 (&y)[MT_][2][2], float sl)
{
    constexpr int mt = I / 2, lt = I % 2;
    const floatx4 s0 = y[mt][0][lt] * sl, s1 = y[mt][1][lt] * sl;
#define ZV_XW(i)                                                                                                  \
    {                                                                                                             \
        const float2v r = {lrelu_max_pre(y[mt][0][lt][i], s0[i]), lrelu_max_pre(y[mt][1][lt][i], s1[i])};      \
        const half2v h = __builtin_convertvector(r, half2v);                                                     \
        lds_st_b32<(mt * 32 + 2 * i + lt) * 80>(xa, *(const unsigned *)&h);                                       \
    }
    ZV_XW(0) ZV_XW(1) ZV_XW(2) ZV_XW(3)
#undef ZV_XW
    if constexpr (I + 1 < MT_ * 2) xwrite_all16<MT_, I + 1>(xa, y, sl);
}
template <int MT_, int I = 0>
__device__ __forceinline__ void pack_all16(unsigned pa, const uint2 (&pk)[MT_][2][2])
{
    constexpr int mt = I / 4, lt = (I / 2) % 2, wt = I % 2;
    lds_st_b64<(mt * 32 + lt) * 80 + 32 * wt>(pa, pk[mt][lt][wt]);
    if constexpr (I + 1 < MT_ * 4) pack_all16<MT_, I + 1>(pa, pk);
}

template <int MT, int R>
__global__ __launch_bounds__(64 * (R / 32 / MT), (MT >= 4 ? 2 : 4)) void resblock_block32_kernel(const TripleJobs jobs)
{
    constexpr int CP = 32, NWV = R / 32 / MT, NTH = 64 * NWV;
    constexpr int RS = CP * 2 + 16, NKC = CP / 16;
    // workgroup -> (job, tile): with `il` jobs interleaved the MRF branches of one stretch of the sequence run next to each
    // other on the same XCD (blockIdx.x & 7 picks the XCD), so only the first of them fetches the shared input from HBM.
    const int il = jobs.interleave;
    const int bx = il > 1 ? (int)(((blockIdx.x >> 3) / il) << 3 | (blockIdx.x & 7)) : (int)blockIdx.x;
    const int jb = il > 1 ? (int)((blockIdx.x >> 3) % il) : (int)blockIdx.z;
    const TripleJob &P = jobs.j[jb];
    int H;
    {
        int sumd0 = 0;
        for (int d = 0; d < P.n_dil; d++) sumd0 += P.dil[d];
        H = ((P.K - 1) / 2) * (sumd0 + P.n_dil);
    }
    const int TM = R - 2 * H;
    const int tps = (jobs.segs.max_rows * jobs.rate + TM - 1) / TM;
    const int vt = zv_xcd_tile(bx, tps * jobs.segs.nseg);
    if (vt >= tps * jobs.segs.nseg) return;
    const int useg = vt / tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int t0 = (vt - useg * tps) * TM;
    if (t0 >= L) return;
    const bool edge = t0 - H < 0 || t0 - H + R > L;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 16 x 16 x 32 layouts (see resblock_pair_kernel): lane (c, g); the f32 tile in conv2's accumulator layout
    // y[mt][wt][lt][i] = tile row wave*32*MT + mt*32 + 8g + 2i + lt, channel 2c + wt
    const int lc = lane & 15, lg = lane >> 4;
    const int irow0 = wave * 32 * MT + 8 * lg;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#ifdef ZV_STAMPS
    const int stamp_wg = blockIdx.x;
    int stamp_k = 1;
#endif
    ZV_STAMP(0)

    const int K = P.K, nd = P.n_dil;
    const int h2 = (K - 1) / 2;
    int dmax = 1;
    for (int d = 0; d < nd; d++) dmax = P.dil[d] > dmax ? P.dil[d] : dmax;
    const float *y_seg = P.y + (size_t)sg.row0 * jobs.rate * CP;
    const int XM = h2 * dmax;
    const int xrows = R + 2 * XM + 5 * dmax;          // + slack: zero-weight taps and the last A prefetch read past the margin
    const int nb = ((K * NKC + 3) >> 2) >> 1;         // 8-step bodies per conv
    const int nblk = 8 * nb;                          // weight fragments per conv (real ones first, zero blocks behind)

    // Two weight buffers where they fit (jobs.db_mask: 3- and 7-tap branches): a conv's fragments are requested while the
    // conv BEFORE it runs and have a whole MFMA loop plus a pack / epilogue phase to land.  With one buffer (11 taps) the
    // request can only follow the barrier that ends the previous conv and the next barrier waits for it: phase stamps of
    // that form show 2 us of DMA latency in each of a block's six pack / epilogue phases (27.6 us per workgroup).
    // The biases of the block's six convs sit in LDS, so nothing in the loop below waits on the vector-memory counter but
    // the barriers that are meant to.
    const bool db = (jobs.db_mask >> jb) & 1;
    char *wlds = smem + round_up(xrows * RS, 1024);
    char *wlds2 = db ? wlds + nblk * 1024 : wlds;                  // conv2's weights
    float *blds = (float *)(wlds + (db ? 2 : 1) * nblk * 1024 + 2048);      // behind the last-prefetch slack: [d][conv][32]
    dma_weights32(P.w1[0], wlds, nblk, wave, lane, NWV);
    if (tid < 64 * nd)
    {
        const int d_ = tid >> 6, c_ = tid & 31;
        blds[tid] = (tid & 32) ? P.b2[d_][c_] : P.b1[d_][c_];
    }
    // the margins of the operand region stay zero for the whole kernel: rows [0, XM) and [XM + R, xrows)
    {
        const int lo = XM * RS / 16, hi0 = (XM + R) * RS / 16, hi = xrows * RS / 16;
        for (int i = tid; i < lo; i += NTH) ((uint4 *)smem)[i] = make_uint4(0, 0, 0, 0);
        for (int i = hi0 + tid; i < hi; i += NTH) ((uint4 *)smem)[i] = make_uint4(0, 0, 0, 0);
    }
    // tile row i <-> time t0 - H + i; rows outside [0, L) are out of the descriptor's range and read as 0
    floatx4 yreg[MT][2][2];          // [mt][wt][lt][i]
    {
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)y_seg, 0, L * CP * 4, 0x00020000);
        const int voff = ((t0 - H + irow0) * CP + 2 * lc) * 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int lt = 0; lt < 2; lt++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_y, voff + (mt * 32 + 2 * i + lt) * CP * 4, 0, 0);
                    yreg[mt][0][lt][i] = __uint_as_float(v[0]);
                    yreg[mt][1][lt][i] = __uint_as_float(v[1]);
                }
    }

    const char *abase = smem + (wave * 32 * MT + 2 * lc) * RS + lg * 16;
    const char *wl = wlds + lane * 16, *wl2 = wlds2 + lane * 16;
    const float sl = P.slope;
    for (int d = 0; d < nd; d++)
    {
        const int dil = P.dil[d], h1 = h2 * dil;
        if (d)
        {
            // the previous conv2 is done reading XT and its weights (raw barrier: conv1's weights may be in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!db) dma_weights32(P.w1[d], wlds, nblk, wave, lane, NWV);
        }
        // ---- X = f16(lrelu(Y)) into region rows XM .. XM + R - 1
        {
            static_assert(RS == 80, "xwrite_all / pack_all carry the row stride");
            xwrite_all16<MT>((unsigned)(uintptr_t)(smem + (XM + irow0) * RS + 2 * lc * 2), yreg, sl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();                       // X complete, conv1's weights landed (the barrier drains the DMA)
        if (db) dma_weights32(P.w2[d], wlds2, nblk, wave, lane, NWV);      // under conv1 and the pack
#ifdef ZV_STAMPS
        if (stamp_k < 11) { ZV_STAMP(stamp_k) stamp_k++; }
#endif

        // ---- conv1 (dilated), transposed product; output tile row i reads region rows XM + i - h1 + tap*dil
        floatx4 acc[MT][1][2][2];
        mfma32x_ldsw<MT, true>(acc, abase + (XM - h1) * RS, dil * RS, wl, K);
#ifdef ZV_STAMPS
        if (stamp_k < 11) { ZV_STAMP(stamp_k) stamp_k++; }
#endif
        // every wave is done reading X and conv1's weights (raw barrier: conv2's weights may be in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (the biases are read before any request goes out: hipcc orders an LDS read behind a pending LDS-DMA with vmcnt(0))
        // (raw ds_reads: hipcc would order a visible LDS read behind the weight DMA in flight with vmcnt(0) and drain it here)
        float4 bq[2];
        float2v bias2;
        {
            // conv1: channels 16 wt + 4g + i; conv2: channels 2c + wt
            const unsigned ba = (unsigned)(uintptr_t)(blds + d * 64 + 4 * lg), bb = (unsigned)(uintptr_t)(blds + d * 64 + 32 + 2 * lc);
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:64\n\tds_read_b64 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bias2)
                         : "v"(ba), "v"(bb)
                         : "memory");
        }
        if (!db) dma_weights32(P.w2[d], wlds, nblk, wave, lane, NWV);
        // ---- xt = f16(lrelu(conv1 + b1)), zero outside [0, L): acc[mt][0][wt][lt][i] = tile row mt*32 + 2c + lt, channel 16 wt + 4g + i
        {
            const unsigned pa = (unsigned)(uintptr_t)(smem + (XM + wave * 32 * MT + 2 * lc) * RS + 4 * lg * 2);
            uint2 pkv[MT][2][2];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int lt = 0; lt < 2; lt++)
                {
#pragma unroll
                    for (int wt = 0; wt < 2; wt++)
                        pkv[mt][lt][wt] = lrelu4_f16(acc[mt][0][wt][lt][0] + bq[wt].x, acc[mt][0][wt][lt][1] + bq[wt].y,
                                                     acc[mt][0][wt][lt][2] + bq[wt].z, acc[mt][0][wt][lt][3] + bq[wt].w, sl);
                }
            if (edge)
            {
                // (a branch the compiler keeps: interior tiles skip the selects)
                asm volatile("; edge tile: rows outside [0, L) are zero");
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
                    {
                        const int t = t0 - H + wave * 32 * MT + mt * 32 + 2 * lc + lt;
                        const bool in = t >= 0 && t < L;
#pragma unroll
                        for (int wt = 0; wt < 2; wt++)
                        {
                            pkv[mt][lt][wt].x = in ? pkv[mt][lt][wt].x : 0u;
                            pkv[mt][lt][wt].y = in ? pkv[mt][lt][wt].y : 0u;
                        }
                    }
            }
            pack_all16<MT>(pa, pkv);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __syncthreads();                       // XT complete, conv2's weights landed
        if (db && d + 1 < nd) dma_weights32(P.w1[d + 1], wlds, nblk, wave, lane, NWV);      // under conv2, the update and the next X write
#ifdef ZV_STAMPS
        if (stamp_k < 11) { ZV_STAMP(stamp_k) stamp_k++; }
#endif

        // ---- conv2 (dil 1): output tile row i reads region rows XM + i - h2 + tap;  Y = Y + (conv2 + b2), 0 outside [0, L)
        mfma32x_ldsw<MT, false>(acc, abase + (XM - h2) * RS, RS, wl2, K);
        {
            // (four rows of a channel per accumulator: packed adds, v_pk_add_f32, two rows at a time)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int wt = 0; wt < 2; wt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++) yreg[mt][wt][lt] = (acc[mt][0][wt][lt] + bias2[wt]) + yreg[mt][wt][lt];
            if (edge)
            {
                asm volatile("; edge tile: rows outside [0, L) are zero");
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int lt = 0; lt < 2; lt++)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            const int t = t0 - H + irow0 + mt * 32 + 2 * i + lt;
                            const bool in = t >= 0 && t < L;
                            yreg[mt][0][lt][i] = in ? yreg[mt][0][lt][i] : 0.f;
                            yreg[mt][1][lt][i] = in ? yreg[mt][1][lt][i] : 0.f;
                        }
            }
        }
    }

    // ---- store the centre rows (tile rows H .. H + TM - 1, time < L): anything else gets an out-of-range offset
    float *const out_seg = P.out + (size_t)sg.row0 * jobs.rate * CP;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)out_seg, 0, L * CP * 4, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int lt = 0; lt < 2; lt++)
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                // (row >= H gives t = t0 + row - H >= 0; t < L is the descriptor's range)
                const int ro = irow0 - H + mt * 32 + 2 * i + lt;
                const int voff = (unsigned)ro < (unsigned)TM ? ((t0 + ro) * CP + 2 * lc) * 4 : -8;
                const u32x2 o = {__float_as_uint(yreg[mt][0][lt][i]), __float_as_uint(yreg[mt][1][lt][i])};
                __builtin_amdgcn_raw_buffer_store_b64(o, rs_out, voff, 0, ZV_ST_AUX);
            }
#ifdef ZV_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZV_STAMP(11)
#endif
}

bool triple_supported(int Cp, int K, const int *dil, int n_dil)
{
    if (Cp != 32 || n_dil < 1 || n_dil > TRIPLE_MAX_DIL || !pair_supported(Cp, K) || (K & 1) == 0) return false;
    int sumd = 0;
    for (int d = 0; d < n_dil; d++) sumd += dil[d];
    return 256 - (K - 1) * (sumd + n_dil) >= 96;          // at least 3/8 of the tile's rows are output
}

// the tile height launch_triple picks: 512 rows (the halo recompute of the 11-tap branch falls from 1.9x to 1.3x) once there are
// enough rows for about eight rounds of such workgroups, else 256 (measured at 512 frames: 100 vs 104 us)
static void triple_tile(int njobs, int n_cu, const Segs &segs, int rate, int &MT, int &R)
{
    const int Lmax = segs.max_rows * rate;
    R = ((long)Lmax * segs.nseg * njobs >= 7000L * n_cu || knob(ZV_TRIPLE_V2) == 3) ? 512 : 256;      // (ZV_TRIPLE_V2 = 3: tests force the batches' tile)
    MT = 2;
}

hipError_t launch_triple(hipStream_t s, const TripleJob *jobs, int njobs, int n_cu, const Segs &segs, int rate)
{
    if (njobs < 1 || njobs > PAIR_MAX_JOBS || segs.nseg < 1 || segs.max_rows < 1) return hipErrorInvalidValue;
    const int dbg = diag_bits();
    TripleJobs js;
    js.segs = segs;
    js.rate = rate;
    js.interleave = 1;
    js.db_mask = 0;
#ifdef ZV_STAMPS
    js.stamp = knob(ZV_STAMP_CP) == 32;
#endif
    const int Lmax = segs.max_rows * rate;
    int R, MT;
    triple_tile(njobs, n_cu, segs, rate, MT, R);
    int gx = 1;
    size_t lds = 0;
    for (int i = 0; i < njobs; i++)
    {
        js.j[i] = jobs[i];
        js.j[i].dbg = dbg;
        const TripleJob &P = jobs[i];
        if (P.Cp != jobs[0].Cp || !triple_supported(P.Cp, P.K, P.dil, P.n_dil)) return hipErrorInvalidValue;
        int sumd = 0, dmax = 1;
        for (int d = 0; d < P.n_dil; d++) { sumd += P.dil[d]; dmax = P.dil[d] > dmax ? P.dil[d] : dmax; }
        const int h2 = (P.K - 1) / 2, TM = R - 2 * h2 * (sumd + P.n_dil);
        gx = std::max(gx, ((Lmax + TM - 1) / TM) * segs.nseg);
        const size_t rows = R + 2 * h2 * dmax + 5 * dmax;
        lds = std::max(lds, rows * (P.Cp * 2 + 16));
    }
    for (int i = njobs; i < PAIR_MAX_JOBS; i++) js.j[i] = js.j[0];
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const dim3 grid(round_up(gx, 8), 1, njobs);      // multiple of 8: zv_xcd_tile
    // batches: the form with the weights in LDS (two workgroups per CU); ZV_TRIPLE_V2 = 0 never, 2 always, 3 always and on 512-row tiles (A/B, tests)
    const int v2_env = knob(ZV_TRIPLE_V2);
    if (v2_env && (R == 512 || v2_env >= 2))
    {
        size_t lds2 = 0;
        js.db_mask = 0;
        for (int i = 0; i < njobs; i++)
        {
            const TripleJob &P = jobs[i];
            int dmax = 1;
            for (int d = 0; d < P.n_dil; d++) dmax = P.dil[d] > dmax ? P.dil[d] : dmax;
            const size_t rows = R + 2 * ((P.K - 1) / 2) * dmax + 5 * dmax;
            const int nb = ((P.K * 2 + 3) >> 2) >> 1;
            // operand tile + weight buffer(s) + 2 fragments (the last B prefetch) + the block's biases
            const size_t one = (size_t)round_up((int)(rows * 80), 1024) + (size_t)(8 * nb + 2) * 1024 + 1024;
            const bool db = knob(ZV_TRIPLE_DB) != 0 && one + (size_t)8 * nb * 1024 <= 80 * 1024;
            if (db) js.db_mask |= 1 << i;
            lds2 = std::max(lds2, one + (db ? (size_t)8 * nb * 1024 : 0));
        }
        if (lds2 <= 80 * 1024)
        {
            js.interleave = knob(ZV_TRIPLE_INTERLEAVE) != 0 ? njobs : 1;
            const dim3 grid2 = js.interleave > 1 ? dim3(round_up(gx, 8) * njobs, 1, 1) : grid;
            auto launch = [&](auto kern, int nth) {
                hipError_t e = lds2 > 64 * 1024 ? hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) : hipSuccess;
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, grid2, dim3(nth), lds2, s, js);
                return hipGetLastError();
            };
            // (the whole-block kernel of batches reads the 16 x 16 x 32 fragment order: TripleJob::w1x / w2x)
            for (int i = 0; i < njobs; i++)
                for (int d = 0; d < jobs[i].n_dil; d++)
                    if (!jobs[i].w1x[d] || !jobs[i].w2x[d]) return hipErrorInvalidValue;
            for (int i = 0; i < PAIR_MAX_JOBS; i++)
                for (int d = 0; d < TRIPLE_MAX_DIL; d++)
                {
                    js.j[i].w1[d] = js.j[i].w1x[d];
                    js.j[i].w2[d] = js.j[i].w2x[d];
                }
            return R == 512 ? launch(resblock_block32_kernel<2, 512>, 512) : launch(resblock_block32_kernel<2, 256>, 256);
        }
    }
#define ZV_TCASE(mt, r) \
    if (MT == mt && R == r) { hipLaunchKernelGGL((resblock_triple_kernel<32, mt, r>), grid, dim3(64 * (r / 32 / mt)), lds, s, js); return hipGetLastError(); }
    ZV_TCASE(2, 256) ZV_TCASE(2, 512)
#undef ZV_TCASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// conv_gemm_kernel — the wide convs of a batch over an f16 operand tensor (StyleTTS decoder behind its operand pre-pass,
// reference src/stylettsdec.cpp:69-149,242-304: ggml_conv_1d = im2col(F16) + mul_mat) as a 256 x 256-tile GEMM.
//
// conv1d_mfma_kernel streams every weight fragment global -> registers once per wave (1 KiB per two MFMAs at its 64-row wave
// tiles: the CU's whole 64 B/clk vector-memory path at the matrix pipe's full rate) and restages its input tile for every
// 256-channel chunk between its MFMA loops.  Here
//   * a workgroup is 8 waves (2 along time x 4 along channels), each owns 128 rows x 64 output channels = 4 x 2 MFMA tiles:
//     128 accumulator registers, one B fragment feeds four MFMAs, one A fragment two;
//   * BOTH operands reach LDS by LDS-DMA (buffer_load ... lds, no registers, no VALU): per step of the K loop ("unit" =
//     (256-channel chunk, tap, 64-channel block)) the workgroup moves a 256-row x 64-channel slice of the operand tensor — the
//     rows shifted by tap * dil: the im2col is the DMA's address arithmetic, rows outside the utterance are outside the
//     buffer descriptor and arrive as zeros — and the 64 x 256 block of weights, 32 KiB each, into one of two 64-KiB buffers,
//     while the MFMAs of the previous unit run (one barrier per unit);
//   * the operand tile's 16-byte pieces are XOR-swizzled by row (slot = piece ^ ((row >> 1) & 7)) on the SOURCE side of the DMA
//     and on the ds_read_b128 side, so fragment reads are bank-conflict free without padding.
// Units walk (chunk, tap, channel) in conv1d_mfma_kernel's order — 64-channel blocks of a chunk inside a tap — so every
// output element is the same accumulation chain: same bits as every other regime.  Channels past Cin_p inside the last block
// read finite neighbours (the next row) against zero weights.
int conv_gemm_groups(int Cout_p) { return ((Cout_p + 31) / 32) / 8; }
// tiles the kernel covers: whole groups of 8, plus ONE leftover tile (1 056 channels = 33 tiles, 528 = 17) that the last group's
// workgroups compute on the side (one extra 32 x 32 tile per wave); more leftovers stay with conv1d_mfma_kernel
int conv_gemm_tiles(int Cout_p)
{
    const int nt = (Cout_p + 31) / 32, ng = nt / 8;
    return ng * 8 + ((nt - ng * 8 == 1 && ng >= 1) ? 1 : 0);
}

__device__ __forceinline__ int gemm_groups_dev(int Cout_p) { return ((Cout_p + 31) >> 5) >> 3; }

__device__ __forceinline__ int conv_gemm_units_dev(int Cin_p, int K)
{
    const int full = Cin_p >> 8, rem = Cin_p & 255;
    return K * (full * 4 + ((rem + 63) >> 6));
}

int conv_gemm_units(int Cin_p, int K)
{
    int n = 0;
    for (int c0 = 0; c0 < Cin_p; c0 += 256) n += K * ((std::min(256, Cin_p - c0) + 63) / 64);
    return n;
}

size_t conv_gemm_weight_halfs(int Cin_p, int Cout_p, int K)
{
    return (size_t)conv_gemm_tiles(Cout_p) * conv_gemm_units(Cin_p, K) * 2048;      // 4 fragments of 512 halfs per (unit, tile)
}

// one 1-KiB piece global -> LDS: lane i lands at lds_base + 16 * i.  The DMA is invisible to hipcc (which would answer a
// visible one with vmcnt(0) in front of every LDS read): the kernel counts it by hand.  M0 carries the LDS address and is
// restored (cdna_hip_programming.md §5.7); s_nop 4: the descriptor / M0 may have been written by the instructions just before.
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(u32x4s rsrc, unsigned lds_base, unsigned voff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voff), "s"(rsrc)
                 : "memory");
}

__device__ __forceinline__ u32x4s make_rsrc(const void *base, unsigned bytes)
{
    const uint64_t a = (uint64_t)base;
    u32x4s r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);      // stride 0
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

// ---- the kernel runs on v_mfma_f32_16x16x32_f16 (round 4; round 3: 32x32x16): one half unit = ONE step of 32 channels — the
// same chain per output element as two k16 steps (scripts/mfma_shape_bits.hip) —, 8 x 4 tiles of 16 x 16 per wave.
// EXTRA: the last group's workgroups, which also own the conv's ninth leftover tile (one more 32 x 32 block per wave).  Two
// instantiations behind one launch: the 8-tile path keeps its own schedule and registers.
//   * operand image: row r's four 16-byte pieces at slots piece ^ ((r >> 1) & 3) — the swizzle under which the 16 x 16 x 32 A
//     fragment reads (lane = row c of a 16-row tile, k group g = piece) fall on 16 distinct slots per lane group;
//   * weights (pack_conv_weight_gemm): per half unit [tile32][wt][lane][8 halfs], column c of tile wt = channel 2c + wt: a lane's
//     two tiles of a 32-channel group are NEIGHBOURING channels, the epilogue moves 8 bytes per lane (four rows x 128 bytes per
//     instruction: half the instructions of the 4-byte form);
//   * the InstanceNorm partial sums keep tile_stats_store's summation order (rows 0-3, 8-11, 16-19, 24-27 then the other half,
//     each a sequential f64 chain): the chain is handed from lane group to lane group.
void pack_conv_weight_gemm(const uint16_t *w, int K, int IC, int OC, int Cin_p, int Cout_p, uint16_t *dst)
{
    const int ng = conv_gemm_groups(Cout_p), nu = conv_gemm_units(Cin_p, K), ntk = conv_gemm_tiles(Cout_p);
    size_t gbase = 0;                                // halfs before group g
    for (int g = 0; g < ng; g++)
    {
        const int ntg = (g == ng - 1) ? ntk - 8 * g : 8;          // 8, or 9 in the last group
        int u = 0;
        for (int c0 = 0; c0 < Cin_p; c0 += 256)
        {
            const int nsub = (std::min(256, Cin_p - c0) + 63) / 64;
            for (int tap = 0; tap < K; tap++)
                for (int sub = 0; sub < nsub; sub++, u++)
                    for (int hf = 0; hf < 2; hf++)
                        for (int nt = 0; nt < ntg; nt++)
                            for (int wt = 0; wt < 2; wt++)
                            {
                                uint16_t *d = dst + gbase + ((((size_t)u * 2 + hf) * ntg + nt) * 2 + wt) * 512;
                                for (int lane = 0; lane < 64; lane++)
                                    for (int j = 0; j < 8; j++)
                                    {
                                        const int oc = (g * 8 + nt) * 32 + 2 * (lane & 15) + wt;
                                        const int ic = c0 + sub * 64 + hf * 32 + 8 * (lane >> 4) + j;
                                        d[lane * 8 + j] = (oc < OC && ic < IC) ? w[((size_t)oc * IC + ic) * K + tap] : (uint16_t)0;
                                    }
                            }
        }
        gbase += (size_t)nu * 4 * ntg * 512;
    }
}

template <bool EXTRA>
__device__ __forceinline__ void conv_gemm_body(const ConvJobs &jobs, const int g, const int rt)
{
    constexpr int BM = 256, UNIT = 32768;            // bytes of an 8-tile unit's weight block
    constexpr int ntg = EXTRA ? 9 : 8;
    constexpr int AH = 16384, BH = ntg * 2048;       // bytes of a half unit's operand slice / weight fragments
    constexpr int SLOT = AH + 18432;                 // one ring slot (room for 9 tiles)
    const ConvJob &J = jobs.j[0];
    const int useg = rt / jobs.tps;
    const Seg sg = seg_at(jobs.segs, useg);
    const int L = sg.rows * jobs.rate;
    const int m0 = (rt - useg * jobs.tps) * BM;
    if (m0 >= L) return;
    const size_t row0 = (size_t)sg.row0 * jobs.rate;

    extern __shared__ __attribute__((aligned(1024))) char smem[];     // [4 slots][A 16 KiB | B 16 (18) KiB]
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lc = lane & 15, lg = lane >> 4;
    const int K = J.K, dil = J.dil, Cin_p = J.Cin_p, ldx = J.ldx;
    const int nunits = conv_gemm_units_dev(Cin_p, K), nhalf = 2 * nunits;
    constexpr int bunit = ntg * 4096;                // bytes of one unit's weight block

    const u32x4s rs_a = make_rsrc((const _Float16 *)J.x0 + row0 * ldx, (unsigned)((size_t)L * ldx * 2));
    const u32x4s rs_b = make_rsrc((const char *)J.w8 + (size_t)g * nunits * UNIT, (unsigned)((size_t)nunits * bunit));

    // this wave's two operand pieces of a half unit: piece j = wave * 2 + i covers tile rows 16 j .. 16 j + 15; lane -> (row,
    // slot); the slot holds the 16-byte piece (slot ^ ((row >> 1) & 3)) of the row's 64 bytes
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 1) & 3);
        a_voff[i] = ((m0 - J.pad + row) * ldx + piece * 8) * 2;
    }

    // half-unit walk: (chunk, tap, 64-channel block of the chunk, half)
    int uc0 = 0, utap = 0, usub = 0, unsub = (min(256, Cin_p) + 63) >> 6, uh = 0;
    unsigned i_abase = 0;
    int i_aoff = 0, i_boff = 0;
    auto issue_begin = [&](int h) {
        i_abase = lds0 + (h & 3) * SLOT;
        i_aoff = (utap * dil * ldx + uc0 + usub * 64 + uh * 32) * 2;
        i_boff = (h >> 1) * bunit + (h & 1) * BH;
        if (++uh == 2)
        {
            uh = 0;
            if (++usub == unsub)
            {
                usub = 0;
                if (++utap == K)
                {
                    utap = 0;
                    uc0 += 256;
                    unsub = (min(256, Cin_p - uc0) + 63) >> 6;
                }
            }
        }
    };
    auto issue_pa = [&](int i) { dma_piece(rs_a, i_abase + (wave * 2 + i) * 1024, (unsigned)(a_voff[i] + i_aoff)); };
    auto issue_pb = [&](int i) { dma_piece(rs_b, i_abase + AH + (wave * 2 + i) * 1024, (unsigned)(i_boff + (wave * 2 + i) * 1024 + lane * 16)); };
    auto issue_part = [&](int i) {
        issue_pa(i);
        issue_pb(i);
    };
    auto issue_x = [&]() {
        if (EXTRA && wave < 2) dma_piece(rs_b, i_abase + AH + (16 + wave) * 1024, (unsigned)(i_boff + (16 + wave) * 1024 + lane * 16));
    };
    const bool five = EXTRA && wave < 2;             // this wave requests five pieces per half

    // fragment addresses inside a slot: A tile tm (16 rows): row wm*128 + tm*16 + c, piece g, swizzled; B tile tn: fragment wn*4 + tn
    const int swz = (lc >> 1) & 3;
    const int a_rd = (wm * 128 + lc) * 64 + ((lg ^ swz) << 4);
    const int ax_rd = (wave * 32 + lc) * 64 + ((lg ^ swz) << 4);      // the ninth tile: rows wave * 32 ...
    const int b_rd = AH + wn * 4096 + lane * 16;

    floatx4 acc[8][4];
    floatx4 accx[2][2];
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // prologue: halves 0, 1, 2 requested
#pragma unroll
    for (int h = 0; h < 3; h++)
        if (h < nhalf)
        {
            issue_begin(h);
            issue_part(0);
            issue_part(1);
            issue_x();
        }
    // the half about to be consumed: its weight fragments and the first four row tiles' operand fragments are read at the END of the
    // previous iteration (that half has landed for every wave one barrier earlier), so that the MFMAs start right behind the barrier
    half8 aA[4], bb[4], axA, bxb[2];
#define ZV_G16_READ0(hn)                                                                                          \
    {                                                                                                             \
        const char *nb_ = smem + ((hn) & 3) * SLOT;                                                               \
        _Pragma("unroll") for (int tm = 0; tm < 4; tm++) aA[tm] = *(const half8 *)(nb_ + a_rd + tm * 1024);       \
        _Pragma("unroll") for (int tn = 0; tn < 4; tn++) bb[tn] = *(const half8 *)(nb_ + b_rd + tn * 1024);       \
        if constexpr (EXTRA)                                                                                      \
        {                                                                                                         \
            axA = *(const half8 *)(nb_ + ax_rd);                                                                  \
            bxb[0] = *(const half8 *)(nb_ + AH + 16 * 1024 + lane * 16);                                          \
            bxb[1] = *(const half8 *)(nb_ + AH + 17 * 1024 + lane * 16);                                          \
        }                                                                                                         \
    }
    {
        if (nhalf > 2)
        {
            if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ZV_G16_READ0(0)
    }
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nhalf; h++)
    {
        if (h > 0)
        {
            if (h + 2 < nhalf)
            {
                if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const bool more = h + 3 < nhalf;
        if (more) issue_begin(h + 3);
        const char *buf = smem + (h & 3) * SLOT;
        // this half's other four row tiles (+ the ninth tile's second row tile), then the MFMAs of the first four
        half8 aB[4], axB, bcur[4], bxc[2];
#pragma unroll
        for (int tm = 0; tm < 4; tm++) aB[tm] = *(const half8 *)(buf + a_rd + (4 + tm) * 1024);
        if constexpr (EXTRA) axB = *(const half8 *)(buf + ax_rd + 1024);
#pragma unroll
        for (int tn = 0; tn < 4; tn++) bcur[tn] = bb[tn];
        if constexpr (EXTRA)
        {
            bxc[0] = bxb[0];
            bxc[1] = bxb[1];
        }
        __builtin_amdgcn_sched_barrier(0);
#define ZV_G16_COL(av, tmb, tn, ZERO)                                                                                   \
    _Pragma("unroll") for (int tm = 0; tm < 4; tm++)                                                                    \
        acc[(tmb) + tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[tm], bcur[tn], (ZERO) ? zero4 : acc[(tmb) + tm][tn], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
        if (h == 0)
        {
            ZV_G16_COL(aA, 0, 0, true)
            if (more) issue_pa(0);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aA, 0, 1, true)
            ZV_G16_COL(aA, 0, 2, true)
            if (more) issue_pb(0);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aA, 0, 3, true)
            if constexpr (EXTRA)
            {
                accx[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axA, bxc[0], zero4, 0, 0, 0);
                accx[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axA, bxc[1], zero4, 0, 0, 0);
            }
        }
        else
        {
            ZV_G16_COL(aA, 0, 0, false)
            if (more) issue_pa(0);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aA, 0, 1, false)
            ZV_G16_COL(aA, 0, 2, false)
            if (more) issue_pb(0);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aA, 0, 3, false)
            if constexpr (EXTRA)
            {
                accx[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axA, bxc[0], accx[0][0], 0, 0, 0);
                accx[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axA, bxc[1], accx[0][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next half's first fragments (visible since this iteration's barrier), under the last sixteen MFMAs
        if (h + 1 < nhalf) ZV_G16_READ0(h + 1)
        __builtin_amdgcn_sched_barrier(0);
        if (h == 0)
        {
            ZV_G16_COL(aB, 4, 0, true)
            if (more) issue_pa(1);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aB, 4, 1, true)
            ZV_G16_COL(aB, 4, 2, true)
            ZV_G16_COL(aB, 4, 3, true)
            if constexpr (EXTRA)
            {
                accx[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axB, bxc[0], zero4, 0, 0, 0);
                accx[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axB, bxc[1], zero4, 0, 0, 0);
            }
        }
        else
        {
            ZV_G16_COL(aB, 4, 0, false)
            if (more) issue_pa(1);
            __builtin_amdgcn_sched_barrier(0);
            ZV_G16_COL(aB, 4, 1, false)
            ZV_G16_COL(aB, 4, 2, false)
            ZV_G16_COL(aB, 4, 3, false)
            if constexpr (EXTRA)
            {
                accx[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axB, bxc[0], accx[1][0], 0, 0, 0);
                accx[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(axB, bxc[1], accx[1][1], 0, 0, 0);
            }
        }
        if (more)
        {
            issue_pb(1);
            issue_x();
        }
#undef ZV_G16_COL
    }
#undef ZV_G16_READ0

    // ---- epilogue: bias, residual, scale, activation, f32 store, InstanceNorm partial sums (as conv1d_mfma_kernel), on a
    // 32-row x 32-channel block = row tiles (tA, tB) x the channel pair (2c, 2c + 1) of the block's two column tiles
    const int Cout_p = J.Cout_p;
    const float escale = J.escale;
    const bool has_res = J.res != nullptr;
    const float *res = has_res ? J.res + row0 * J.ldres : nullptr;
    float *out = (float *)J.out + row0 * J.ldo;
    // c0[i], c1[i]: rows 4g + i of the block's first 16-row tile; d0, d1: of its second; channels oc, oc + 1
    auto finish_block = [&](const floatx4 &c0, const floatx4 &c1, const floatx4 &d0, const floatx4 &d1, int t_first, int oc, int blk) {
        if (oc >= Cout_p) return;
        const float2 bias = J.bias ? *(const float2 *)(J.bias + oc) : make_float2(0.f, 0.f);
        float2v resv[2][4];
        if (has_res)
        {
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int t = t_first + 16 * hf + 4 * lg + i;
                    resv[hf][i] = *(const float2v *)(res + (size_t)(t < L ? t : L - 1) * J.ldres + oc);
                }
        }
        float outv[2][2][4];        // [channel][row half][i]
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int t = t_first + 16 * hf + 4 * lg + i;
                float v0 = (hf ? d0[i] : c0[i]) + bias.x, v1 = (hf ? d1[i] : c1[i]) + bias.y;
                if (has_res)
                {
                    v0 = v0 + resv[hf][i][0];
                    v1 = v1 + resv[hf][i][1];
                }
                v0 = v0 * escale;
                v1 = v1 * escale;
                if (J.eact)
                {
                    v0 = lrelu(v0, J.oslope);
                    v1 = lrelu(v1, J.oslope);
                }
                outv[0][hf][i] = v0;
                outv[1][hf][i] = v1;
                if (t < L)
                {
                    const float2v o = {v0, v1};
                    __builtin_nontemporal_store(o, (float2v *)(out + (size_t)t * J.ldo + oc));
                }
            }
        if (J.stat_part && oc < J.stat_C && blk * 32 < L)
        {
            // tile_stats_store's order: per channel two sequential f64 chains — rows 0-3, 8-11, 16-19, 24-27 (lane groups 0, 2, 0, 2)
            // and rows 4-7, 12-15, 20-23, 28-31 (groups 1, 3, 1, 3) — added at the end.  A chain walks from group g to g + 2 and back.
#pragma unroll
            for (int ch = 0; ch < 2; ch++)
            {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int st = 0; st < 4; st++)          // stage: row half st >> 1, lane groups {0, 1} (even stages) / {2, 3} (odd)
                {
                    if (st)
                    {
                        s1 = __shfl_xor(s1, 32, 64);
                        s2 = __shfl_xor(s2, 32, 64);
                    }
                    if ((lg >> 1) == (st & 1))
#pragma unroll
                        for (int i = 0; i < 4; i++)
                        {
                            const int t = t_first + 16 * (st >> 1) + 4 * lg + i;
                            const double x = (t < L) ? (double)outv[ch][st >> 1][i] : 0.0;
                            s1 += x;
                            s2 += x * x;
                        }
                }
                // lane groups 2 and 3 hold the two chains
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                if (lg == 2) *(double2 *)(J.stat_part + (((size_t)useg * J.stat_nblk + blk) * J.stat_C + oc + ch) * 2) = make_double2(s1, s2);
            }
        }
    };
    const int tbase = m0 + wm * 128;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int b4 = 0; b4 < 4; b4++)
            finish_block(acc[2 * b4][2 * p], acc[2 * b4][2 * p + 1], acc[2 * b4 + 1][2 * p], acc[2 * b4 + 1][2 * p + 1], tbase + b4 * 32,
                         ((g * 8 + wn * 2 + p) << 5) + 2 * lc, (m0 >> 5) + wm * 4 + b4);
    if constexpr (EXTRA)
        finish_block(accx[0][0], accx[0][1], accx[1][0], accx[1][1], m0 + wave * 32, ((g * 8 + 8) << 5) + 2 * lc, (m0 >> 5) + wave);
}

__global__ __launch_bounds__(512, 2) void conv_gemm_kernel(const ConvJobs jobs)
{
    const ConvJob &J = jobs.j[0];
    const int ng = (gemm_groups_dev(J.Cout_p));
    const int rts = jobs.tps * jobs.segs.nseg;
    int g, rt;
    if (jobs.order == 2 && ng > 1)
    {
        // longest jobs first (see conv_gemm_kernel)
        if ((int)blockIdx.x < rts) { g = ng - 1; rt = blockIdx.x; }
        else { const int b2 = blockIdx.x - rts; g = b2 % (ng - 1); rt = b2 / (ng - 1); }
    }
    else
    {
        g = blockIdx.x % ng;
        rt = blockIdx.x / ng;
    }
    if (rt >= rts) return;
    const bool ninth = g == ng - 1 && ((J.Cout_p + 31) >> 5) - 8 * ng == 1;
    if (ninth)
        conv_gemm_body<true>(jobs, g, rt);
    else
        conv_gemm_body<false>(jobs, g, rt);
}

static hipError_t launch_conv_gemm(hipStream_t s, const ConvJob &job, const Segs &segs, int rate)
{
    ConvJobs js;
    js.j[0] = job;
    js.j[0].dbg = diag_bits();
    for (int i = 1; i < CONV_MAX_JOBS; i++) js.j[i] = js.j[0];
    js.segs = segs;
    js.rate = rate;
    js.nt_begin = 0;
    js.order = knob(ZV_GEMM_ORDER);
    const int Lmax = segs.max_rows * rate;
    js.tps = (Lmax + 255) / 256;
    if (job.stat_part && job.stat_nblk * 32 < Lmax) return hipErrorInvalidValue;
    const int ng = conv_gemm_groups(job.Cout_p), rts = js.tps * segs.nseg;
    // (grid.x covers (row tile, group) in the kernel's XCD-aware order: 8 / ng XCDs per group)
    const dim3 grid((js.order == 1 && (ng == 1 || ng == 2 || ng == 4)) ? round_up(rts, 8 / ng) * ng : rts * ng, 1, 1);
    const int lds = 4 * (16384 + 18432);
    // (per launch, like every other launcher here: the attribute is stored per device)
    hipError_t e = hipFuncSetAttribute((const void *)conv_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_gemm_kernel, grid, dim3(512), lds, s, js);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// vocoder tail: lrelu -> conv (C -> 1, K taps) + bias -> tanh.  Cout = 1 has no GEMM shape: each lane owns
// one output sample and walks its K x C window in LDS (f16 operands, f32 accumulate).

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p)
{
    const f32x4 v = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void out_conv_tanh_kernel(const OutConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cp = round_up(a.C, 16);
    const int RS = Cp * 2 + 16;
    const int K = a.K, pad = (K - 1) / 2;
    const int rows = 256 + K - 1;
    const int tps = (a.segs.max_rows * a.rate + 255) >> 8;
    const int useg = blockIdx.x / tps;
    const Seg sg = seg_at(a.segs, useg);
    const int L = sg.rows * a.rate;
    const int m0 = (blockIdx.x - useg * tps) * 256;
    if (m0 >= L) return;
    const size_t row0 = (size_t)sg.row0 * a.rate;
    const float *x0 = a.x0 + row0 * a.ldx, *x1 = a.x1 ? a.x1 + row0 * a.ldx : nullptr, *x2 = a.x2 ? a.x2 + row0 * a.ldx : nullptr;
    const int tid = threadIdx.x;
    _Float16 *wl = (_Float16 *)(smem + rows * RS);
    for (int i = tid; i < K * Cp; i += 256) wl[i] = ((const _Float16 *)a.w)[i];

    const int cols = Cp >> 2;
    for (int idx = tid; idx < rows * cols; idx += 256)
    {
        const int r = idx / cols, c4 = idx - r * cols;
        const int t = m0 - pad + r;
        half4 h = {0, 0, 0, 0};
        if (t >= 0 && t < L)
        {
            const size_t off = (size_t)t * a.ldx + c4 * 4;
            // non-temporal loads: the branch outputs are read exactly once, here (measured 620 -> 577 us per batch launch;
            // the same policy on the ResBlock kernels' staging loads costs them 4 ... 9 %: their residual re-read wants L2)
            float4 v = nt_load4(x0 + off);
            if (x1)
            {
                const float4 b = nt_load4(x1 + off);
                const float4 d = nt_load4(x2 + off);
                v.x = ((v.x + b.x) + d.x) * a.pscale;
                v.y = ((v.y + b.y) + d.y) * a.pscale;
                v.z = ((v.z + b.z) + d.z) * a.pscale;
                v.w = ((v.w + b.w) + d.w) * a.pscale;
            }
            else
            {   // x0 already is the branches' sum (merged last pair of the stage)
                v.x = v.x * a.pscale;
                v.y = v.y * a.pscale;
                v.z = v.z * a.pscale;
                v.w = v.w * a.pscale;
            }
            h[0] = (_Float16)lrelu(v.x, a.slope);
            h[1] = (_Float16)lrelu(v.y, a.slope);
            h[2] = (_Float16)lrelu(v.z, a.slope);
            h[3] = (_Float16)lrelu(v.w, a.slope);
        }
        *(half4 *)(smem + r * RS + c4 * 8) = h;
    }
    __syncthreads();
    const int t = m0 + tid;
    if (t >= L) return;
    float acc = 0.f;
    for (int tap = 0; tap < K; tap++)
    {
        const char *row = smem + (tid + tap) * RS;
        for (int c = 0; c < Cp; c += 8)
        {
            const half8 x = *(const half8 *)(row + c * 2);
            const half8 w = *(const half8 *)(wl + tap * Cp + c);
#pragma unroll
            for (int j = 0; j < 8; j++) acc = fmaf((float)x[j], (float)w[j], acc);
        }
    }
    a.out[row0 + t] = tanhf(acc + a.bias);
}

hipError_t launch_out_conv(hipStream_t s, const OutConvArgs &a)
{
    const int Cp = round_up(a.C, 16);
    const size_t lds = (size_t)(256 + a.K - 1) * (Cp * 2 + 16) + (size_t)a.K * Cp * 2;
    if (lds > 64 * 1024 || a.segs.nseg < 1) return hipErrorInvalidValue;
    const int tps = (a.segs.max_rows * a.rate + 255) / 256;
    hipLaunchKernelGGL(out_conv_tanh_kernel, dim3(tps * a.segs.nseg), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace zv

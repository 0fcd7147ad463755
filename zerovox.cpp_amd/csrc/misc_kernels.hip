// misc_kernels.hip — the non-conv kernels of the fixed schedule: InstanceNorm statistics, f32 linear
// layers, attention, LayerNorm, variance-adaptor bucketing and the device length regulator.
// Every kernel maps a workgroup to (segment, tile inside the segment) — see `Segs` in kernels.h — so one launch
// covers all utterances of a batch while each utterance keeps its own extents.
#include "kernels.h"
#include "knobs.h"

#include <cstring>

namespace zv
{

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef float floatx16m __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// InstanceNorm1d statistics over time (ggml_norm: mean and biased variance accumulated in f64, scale =
// 1/sqrtf(var + eps); reference ggml-cpu.c:6906-6923), in two steps so that no kernel ever has to walk a whole
// sequence: (1) per 32-row block and channel the f64 pair (sum x, sum x^2) — written by the conv epilogue that
// produced the tensor (conv1d_mfma.hip: tile_stats_store), by stats_partial_kernel below for tensors that come from
// elsewhere, or by norm_apply_kernel for its own output; (2) stats_finalize_kernel adds a segment's blocks in block
// order.  The reference subtracts the f32 mean before squaring (two passes); sum x^2 - (sum x)^2 / L in f64 is the
// same quantity to ~1e-12 relative, far inside the f32 rounding of the reference's own terms.
__global__ __launch_bounds__(256) void stats_partial_kernel(const float *__restrict__ x, int ldx, int C,
                                                            double *__restrict__ part, int nblk, const Segs segs, int rate)
{
    __shared__ double red[2][4][64];
    const int useg = blockIdx.z, blk = blockIdx.y;
    const Seg sg = seg_at(segs, useg);
    const int L = sg.rows * rate;
    if (blk * 32 >= L) return;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float *xs = x + (size_t)sg.row0 * rate * ldx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int t = blk * 32 + rg * 8 + i;
            const double v = (t < L) ? (double)xs[(size_t)t * ldx + c] : 0.0;
            s1 += v;
            s2 += v * v;
        }
    red[0][rg][cl] = s1;
    red[1][rg][cl] = s2;
    __syncthreads();
    if (rg == 0 && c < C)
    {
        const double a = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        const double b = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        *(double2 *)(part + (((size_t)useg * nblk + blk) * C + c) * 2) = make_double2(a, b);
    }
}

hipError_t launch_stats_partial(hipStream_t s, const float *x, int ldx, int C, double *part, int nblk, const Segs &segs, int rate)
{
    const int nb = (segs.max_rows * rate + 31) / 32;
    if (nb > nblk || segs.nseg < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stats_partial_kernel, dim3((C + 63) / 64, nb, segs.nseg), dim3(256), 0, s, x, ldx, C, part, nblk, segs, rate);
    return hipGetLastError();
}

// (mean, rstd) of one channel of one segment from its blocks' partial sums, added in block order
__device__ __forceinline__ float2 stats_from_partials(const double *__restrict__ p, int C, int nb, int L, float eps)
{
    double s1 = 0.0, s2 = 0.0;
    for (int b0 = 0; b0 < nb; b0 += 8)          // eight loads in flight, added in block order
    {
        double2 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = *(const double2 *)(p + (size_t)(b0 + i < nb ? b0 + i : nb - 1) * C * 2);
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (b0 + i < nb)
            {
                s1 += v[i].x;
                s2 += v[i].y;
            }
    }
    const double mean = s1 / (double)L;
    double var = s2 / (double)L - mean * mean;
    var = var > 0.0 ? var : 0.0;
    return make_float2((float)mean, 1.0f / sqrtf((float)var + eps));
}

__global__ __launch_bounds__(64) void stats_finalize_kernel(const double *__restrict__ part, int nblk, int C, float eps,
                                                            float *__restrict__ stat, int stat_seg, int c_off,
                                                            const Segs segs, int rate)
{
    const int useg = blockIdx.y;
    const Seg sg = seg_at(segs, useg);
    const int L = sg.rows * rate;
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C || L <= 0) return;
    const float2 mr = stats_from_partials(part + ((size_t)useg * nblk * C + c) * 2, C, (L + 31) >> 5, L, eps);
    float *o = stat + (size_t)useg * stat_seg + 2 * (c_off + c);
    o[0] = mr.x;
    o[1] = mr.y;
}

hipError_t launch_stats_finalize(hipStream_t s, const double *part, int nblk, int C, float eps, float *stat, int stat_seg,
                                 int c_off, const Segs &segs, int rate)
{
    if (segs.nseg < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((C + 63) / 64, segs.nseg), dim3(64), 0, s, part, nblk, C, eps, stat, stat_seg,
                       c_off, segs, rate);
    return hipGetLastError();
}

// y = ((x - mean) * rstd) * g + b  (affine InstanceNorm written out: the decoder's asr_res branch, reference
// src/stylettsdec.cpp:382-404) + the partial sums of y for the InstanceNorm that follows
__global__ __launch_bounds__(256) void norm_apply_kernel(const float *__restrict__ x, int ldx, int C,
                                                         const float *__restrict__ stat, int stat_seg,
                                                         const float *__restrict__ g, const float *__restrict__ b,
                                                         float *__restrict__ y, int ldy, double *__restrict__ part, int nblk,
                                                         const Segs segs)
{
    __shared__ double red[2][4][64];
    const int useg = blockIdx.z, blk = blockIdx.y;
    const Seg sg = seg_at(segs, useg);
    const int L = sg.rows;
    if (blk * 32 >= L) return;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float *xs = x + (size_t)sg.row0 * ldx;
    float *ys = y + (size_t)sg.row0 * ldy;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
    {
        const float mean = stat[(size_t)useg * stat_seg + 2 * c], rstd = stat[(size_t)useg * stat_seg + 2 * c + 1];
        const float gc = g[c], bc = b[c];
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int t = blk * 32 + rg * 8 + i;
            if (t < L)
            {
                float v = (xs[(size_t)t * ldx + c] - mean) * rstd;
                v = v * gc;
                v = v + bc;
                ys[(size_t)t * ldy + c] = v;
                s1 += (double)v;
                s2 += (double)v * (double)v;
            }
        }
    }
    if (!part) return;
    red[0][rg][cl] = s1;
    red[1][rg][cl] = s2;
    __syncthreads();
    if (rg == 0 && c < C)
    {
        const double a = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        const double q = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        *(double2 *)(part + (((size_t)useg * nblk + blk) * C + c) * 2) = make_double2(a, q);
    }
}

hipError_t launch_norm_apply(hipStream_t s, const float *x, int ldx, int C, const float *stat, int stat_seg, const float *g,
                             const float *b, float *y, int ldy, double *part, int nblk, const Segs &segs)
{
    const int nb = (segs.max_rows + 31) / 32;
    if ((part && nb > nblk) || segs.nseg < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(norm_apply_kernel, dim3((C + 63) / 64, nb, segs.nseg), dim3(256), 0, s, x, ldx, C, stat, stat_seg, g, b, y,
                       ldy, part, nblk, segs);
    return hipGetLastError();
}

// The conv operand of an InstanceNorm / AdaIN layer, written once: y = f16(lrelu(((x - mean) * rstd) * g + b, slope)) —
// the same operations in the same order as the conv kernel's PRO_NORM_ACT prologue, so a conv that reads y (PRO_RAW_F16)
// gives the bits of a conv that normalises on the fly.  Worth its launch when a conv re-stages its input once per group
// of output channels (the decoder's 1 056-wide convs: 9 times) — the f32 prologue then runs once instead of 9 times and the
// staging reads half the bytes.  One workgroup = 64 channels x all rows of one segment; the statistics of channels below
// Cpart are first finalised from the partial sums (exactly stats_finalize_kernel's arithmetic) and also stored in `stat`.
__global__ __launch_bounds__(256) void norm_act_f16_kernel(const float *__restrict__ x, int ldx, int C,
                                                           const double *__restrict__ part, int nblk, int Cpart, float eps,
                                                           float *__restrict__ stat, int stat_seg,
                                                           const float *__restrict__ ga, const float *__restrict__ be, int gb_seg,
                                                           float slope, _Float16 *__restrict__ y, int ldy, _Float16 *__restrict__ yraw,
                                                           const Segs segs)
{
    __shared__ float sm[4][64];                  // mean, rstd, gamma, beta
    const int useg = blockIdx.y;
    const Seg sg = seg_at(segs, useg);
    const int L = sg.rows;
    if (L <= 0) return;
    const int tid = threadIdx.x;
    if (tid < 64)
    {
        const int c = blockIdx.x * 64 + tid;
        float2 mr = make_float2(0.f, 0.f);
        float g = 0.f, b = 0.f;
        if (c < C)
        {
            float *st = stat + (size_t)useg * stat_seg + 2 * c;
            if (c < Cpart)
            {
                mr = stats_from_partials(part + ((size_t)useg * nblk * Cpart + c) * 2, Cpart, (L + 31) >> 5, L, eps);
                st[0] = mr.x;
                st[1] = mr.y;
            }
            else
                mr = make_float2(st[0], st[1]);
            g = ga[(size_t)useg * gb_seg + c];
            b = be[(size_t)useg * gb_seg + c];
        }
        sm[0][tid] = mr.x;
        sm[1][tid] = mr.y;
        sm[2][tid] = g;
        sm[3][tid] = b;
    }
    __syncthreads();
    const int c4 = (tid & 15) * 4, rl = tid >> 4;
    const int c = blockIdx.x * 64 + c4;
    if (c >= C) return;                          // C is a multiple of 4
    const float4 mean = *(const float4 *)&sm[0][c4], rstd = *(const float4 *)&sm[1][c4];
    const float4 g = *(const float4 *)&sm[2][c4], b = *(const float4 *)&sm[3][c4];
    const float *xs = x + (size_t)sg.row0 * ldx + c;
    _Float16 *ys = y + (size_t)sg.row0 * ldy + c;
    _Float16 *yr = yraw ? yraw + (size_t)sg.row0 * ldy + c : nullptr;      // f16(x): the operand of a block's 1x1 shortcut conv
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    // the rows of a segment are dealt over gridDim.z workgroups in chunks of 128 (every one of them finalises the statistics of
    // its 64 channels for itself: the same arithmetic, the same values); eight 16-byte loads per thread in flight
    for (int t0 = blockIdx.z * 128 + rl; t0 < L; t0 += 128 * gridDim.z)
    {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const int t = t0 + 16 * u;
            v[u] = *(const float4 *)(xs + (size_t)(t < L ? t : L - 1) * ldx);
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const int t = t0 + 16 * u;
            if (t >= L) continue;
            float4 r;
            r.x = ((v[u].x - mean.x) * rstd.x) * g.x + b.x;
            r.y = ((v[u].y - mean.y) * rstd.y) * g.y + b.y;
            r.z = ((v[u].z - mean.z) * rstd.z) * g.z + b.z;
            r.w = ((v[u].w - mean.w) * rstd.w) * g.w + b.w;
            half4v h;
            h[0] = (_Float16)(r.x > 0.f ? r.x : r.x * slope);
            h[1] = (_Float16)(r.y > 0.f ? r.y : r.y * slope);
            h[2] = (_Float16)(r.z > 0.f ? r.z : r.z * slope);
            h[3] = (_Float16)(r.w > 0.f ? r.w : r.w * slope);
            *(half4v *)(ys + (size_t)t * ldy) = h;
            if (yr)
            {
                half4v hr;
                hr[0] = (_Float16)v[u].x;
                hr[1] = (_Float16)v[u].y;
                hr[2] = (_Float16)v[u].z;
                hr[3] = (_Float16)v[u].w;
                *(half4v *)(yr + (size_t)t * ldy) = hr;
            }
        }
    }
}

hipError_t launch_norm_act_f16(hipStream_t s, const float *x, int ldx, int C, const double *part, int nblk, int Cpart, float eps,
                               float *stat, int stat_seg, const float *ga, const float *be, int gb_seg, float slope, void *y,
                               int ldy, const Segs &segs, void *yraw)
{
    if ((C & 3) || (ldx & 3) || (ldy & 3) || segs.nseg < 1 || Cpart > C) return hipErrorInvalidValue;
    if (Cpart > 0 && (segs.max_rows + 31) / 32 > nblk) return hipErrorInvalidValue;
    // (row chunks per segment: enough workgroups for about eight per CU, at most one per 128 rows)
    int gz = 1;
    while (gz < 8 && (long)((C + 63) / 64) * segs.nseg * gz < 2048 && gz * 128 < segs.max_rows) gz <<= 1;
    hipLaunchKernelGGL(norm_act_f16_kernel, dim3((C + 63) / 64, segs.nseg, gz), dim3(256), 0, s, x, ldx, C, part, nblk, Cpart, eps, stat,
                       stat_seg, ga, be, gb_seg, slope, (_Float16 *)y, ldy, (_Float16 *)yraw, segs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// f16 operand pre-pass of the wide polyphase transposed convs of a batch (reference src/hifigan.cpp:281-297: leaky_relu in
// front of every upsample conv; the MRF mean of the previous stage, :315, rides in the same pass):
//     out[i] = f16(lrelu(((x0[i] + x1[i]) + x2[i]) * pscale, slope))          (x1 = x2 = null: x0[i] * pscale)
// — the prologue conv1d_mfma_kernel applies while it stages (PRO_ACT / PRO_SCALE_ACT / PRO_SUM3_ACT), element for element,
// so that conv_gemm_kernel can move the operand global -> LDS by DMA.
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void act_f16_kernel(const float4 *__restrict__ x0, const float4 *__restrict__ x1,
                                                      const float4 *__restrict__ x2, float pscale, float slope,
                                                      half4v *__restrict__ out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride)
    {
        float4 v[4], a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const size_t i = i0 + u * stride;
            if (i >= n4) continue;
            v[u] = x0[i];
            if (x1)
            {
                a[u] = x1[i];
                b[u] = x2[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const size_t i = i0 + u * stride;
            if (i >= n4) continue;
            float4 x = v[u];
            if (x1)
            {
                x.x = ((x.x + a[u].x) + b[u].x) * pscale;
                x.y = ((x.y + a[u].y) + b[u].y) * pscale;
                x.z = ((x.z + a[u].z) + b[u].z) * pscale;
                x.w = ((x.w + a[u].w) + b[u].w) * pscale;
            }
            else
            {
                x.x = x.x * pscale;
                x.y = x.y * pscale;
                x.z = x.z * pscale;
                x.w = x.w * pscale;
            }
            half4v h;
            h[0] = (_Float16)(x.x > 0.f ? x.x : x.x * slope);
            h[1] = (_Float16)(x.y > 0.f ? x.y : x.y * slope);
            h[2] = (_Float16)(x.z > 0.f ? x.z : x.z * slope);
            h[3] = (_Float16)(x.w > 0.f ? x.w : x.w * slope);
            __builtin_nontemporal_store(h, out + i);
        }
    }
}

hipError_t launch_act_f16(hipStream_t s, const float *x0, const float *x1, const float *x2, float pscale, float slope, void *out,
                          size_t n)
{
    if ((n & 3) || !x0 || (!x1) != (!x2)) return hipErrorInvalidValue;
    const size_t n4 = n >> 2;
    const size_t wgs = std::min<size_t>((n4 + 1023) / 1024, (size_t)1 << 20);
    if (wgs == 0) return hipSuccess;
    hipLaunchKernelGGL(act_f16_kernel, dim3((unsigned)wgs), dim3(256), 0, s, (const float4 *)x0, (const float4 *)x1, (const float4 *)x2,
                       pscale, slope, (half4v *)out, n4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// y[n][o] = dot(W[o][:], x[n][:]) + b[o]   (f32 weights: ggml_mul_mat + ggml_add, reference
// src/fs2encoder.cpp:77-89,127-128 and src/stylettsdec.cpp:178-179) on the f32-input matrix cores:
// v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain (exact f32, the reference's own FMA accumulation).
// A workgroup of 4 waves owns 64 rows x 128 outputs: wave w the outputs [32w, 32w + 32) of both 32-row halves, so a
// W operand feeds two MFMAs.  Lane l holds A[row l&31][k + (l>>5)] / B[k + (l>>5)][col l&31].
// `extra[o]` is added after the bias (AdaIN: gamma = h[:C] + 1, src/stylettsdec.cpp:186-189).
template <int RT>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float *__restrict__ x, int ldx, int in,
                                                          const float *__restrict__ W, const float *__restrict__ b, int out,
                                                          float *__restrict__ y, int ldy, const float *__restrict__ extra,
                                                          const Segs segs, int tps)
{
    // 32-wide k-chunks of both operands go through LDS: global rows are read as coalesced 128-B segments, the MFMA
    // operand (one f32 per lane, rows across lanes) comes back from LDS with a 33-float row stride (conflict-free)
    __shared__ float Xs[32 * RT][33];
    __shared__ float Ws[128][33];
    const int useg = blockIdx.y / tps;
    const Seg sg = seg_at(segs, useg);
    const int n = sg.rows;
    const int n0 = (blockIdx.y - useg * tps) * 32 * RT;
    if (n0 >= n) return;
    const float *xs = x + (size_t)sg.row0 * ldx;
    float *ys = y + (size_t)sg.row0 * ldy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = blockIdx.x * 128;
    const int i = lane & 31, kk = lane >> 5;
    floatx16m acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0.f;
    float4 xv[RT], wv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < RT; u++)
        {
            const int idx = tid + u * 256, r = idx >> 3, c = (idx & 7) * 4;
            const bool kin = k0 + c < in;                      // `in` is a multiple of 4: a float4 is all in or all out
            const int row = n0 + r;
            xv[u] = (kin && row < n) ? *(const float4 *)(xs + (size_t)row * ldx + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int idx = tid + u * 256, r = idx >> 3, c = (idx & 7) * 4;
            const bool kin = k0 + c < in;
            const int col = o0 + r;
            wv[u] = (kin && col < out) ? *(const float4 *)(W + (size_t)col * in + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < in; k0 += 32)
    {
        __syncthreads();                                       // previous chunk fully consumed
#pragma unroll
        for (int u = 0; u < RT; u++)
        {
            const int idx = tid + u * 256, r = idx >> 3, c = (idx & 7) * 4;
            Xs[r][c] = xv[u].x; Xs[r][c + 1] = xv[u].y; Xs[r][c + 2] = xv[u].z; Xs[r][c + 3] = xv[u].w;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int idx = tid + u * 256, r = idx >> 3, c = (idx & 7) * 4;
            Ws[r][c] = wv[u].x; Ws[r][c + 1] = wv[u].y; Ws[r][c + 2] = wv[u].z; Ws[r][c + 3] = wv[u].w;
        }
        __syncthreads();
        if (k0 + 32 < in) fetch(k0 + 32);                      // next chunk's rows are in flight under this chunk's MFMAs
#pragma unroll
        for (int k = 0; k < 32; k += 2)
        {
            const float wk = Ws[wave * 32 + i][k + kk];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[i][k + kk], wk, acc0, 0, 0, 0);
            if constexpr (RT == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[32 + i][k + kk], wk, acc1, 0, 0, 0);
        }
    }
    // D: col = lane&31 (output), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (token)
    const int col = o0 + wave * 32 + i;
    if (col >= out) return;
    const float bias = b ? b[col] : 0.f;
    const float ex = extra ? extra[col] : 0.f;
#pragma unroll
    for (int hf = 0; hf < RT; hf++)
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int t = n0 + hf * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (t < n)
            {
                float v = (hf ? acc1[r] : acc0[r]) + bias;
                if (extra) v = v + ex;
                ys[(size_t)t * ldy + col] = v;
            }
        }
}

hipError_t launch_linear(hipStream_t s, const float *x, int ldx, int in, const float *W, const float *b, int out, float *y,
                         int ldy, const float *extra, const Segs &segs)
{
    if ((in & 3) || (ldx & 3) || segs.nseg < 1) return hipErrorInvalidValue;          // float4 row loads
    // 64-row workgroups (a W operand feeds two MFMAs) once they fill the chip, 32-row ones for short inputs; the shape
    // never changes a bit: every output element is one k-ordered chain
    const long wg64 = (long)((segs.max_rows + 63) / 64) * segs.nseg * ((out + 127) / 128);
    if (wg64 >= 256)
    {
        const int tps = (segs.max_rows + 63) / 64;
        hipLaunchKernelGGL(linear_mfma_kernel<2>, dim3((out + 127) / 128, tps * segs.nseg), dim3(256), 0, s, x, ldx, in, W, b, out, y,
                           ldy, extra, segs, tps);
    }
    else
    {
        const int tps = (segs.max_rows + 31) / 32;
        hipLaunchKernelGGL(linear_mfma_kernel<1>, dim3((out + 127) / 128, tps * segs.nseg), dim3(256), 0, s, x, ldx, in, W, b, out, y,
                           ldy, extra, segs, tps);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Encoder::graph prologue (reference src/fs2encoder.cpp:306-324): x[n] = cat(word_emb[id], punct_emb[p]) + posenc[n]
__global__ void embed_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ puncts,
                             const float *__restrict__ wemb, int emb, const float *__restrict__ pemb, int pdim,
                             const float *__restrict__ posenc, float *__restrict__ x, int ld, const Segs segs)
{
    const Seg sg = seg_at(segs, blockIdx.y);
    const int n = blockIdx.x;                       // position inside the utterance
    if (n >= sg.rows) return;
    const size_t row = (size_t)sg.row0 + n;
    const int E = emb + pdim;
    const int id = ids[row], p = puncts[row];
    for (int e = threadIdx.x; e < E; e += blockDim.x)
    {
        const float v = (e < emb) ? wemb[(size_t)id * emb + e] : pemb[(size_t)p * pdim + (e - emb)];
        x[row * ld + e] = v + posenc[(size_t)n * E + e];
    }
}

hipError_t launch_embed(hipStream_t s, const int32_t *ids, const int32_t *puncts, const float *wemb, int emb,
                        const float *pemb, int pdim, const float *posenc, float *x, int ld, const Segs &segs)
{
    hipLaunchKernelGGL(embed_kernel, dim3(segs.max_rows, segs.nseg), dim3(256), 0, s, ids, puncts, wemb, emb, pemb, pdim, posenc, x,
                       ld, segs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// ScaledDotProductAttention without mask (reference src/fs2encoder.cpp:103-123) on the f32 matrix cores.
// One workgroup (4 waves) = 64 queries of one (utterance, head):
//   scores : wave (qt = w & 1, kh = w >> 1) computes the transposed tile S^T[32 keys][32 queries] = K_tile . Q_tile^T
//            for its query half qt and every second key tile (K tiles staged through LDS two at a time, the wave's
//            Q fragment lives in registers), scales by 1/temperature as a separate multiply (src/fs2encoder.cpp:107)
//            and parks it in LDS as S[key][query];
//   softmax: per query over all keys, max-subtracted, sum in f64, 1/sum applied as an f32 multiply (ggml soft_max);
//   P . V  : wave (qt, dh) accumulates its share of the 32-wide d tiles over the keys in ascending order; the P operand
//            comes straight from LDS, V rows straight from L2 (128-B segments).
// v_mfma_f32_32x32x2_f32 is an exact k-ordered f32 fma chain, so every dot product is accumulated in index order.
// 64 rows x dk floats of one head, global -> LDS: eight 16-byte loads per thread in flight (clamped addresses, rows past the
// utterance become zeros), then the writes — as four scalars when the destination stride is odd
template <bool VEC>
__device__ __forceinline__ void att_stage64(float *dst, int dstride, const float *src, int ld, int key0, int n, int c4n, int tid)
{
    const int total = 64 * c4n;
    for (int base = tid; base < total; base += 256 * 8)
    {
        float4 t[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            const int idx = base + u * 256 < total ? base + u * 256 : total - 1;
            const int r = idx / c4n, c4 = idx - r * c4n;
            const int key = key0 + r;
            t[u] = *(const float4 *)(src + (size_t)(key < n ? key : n - 1) * ld + c4 * 4);
            if (key >= n) t[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            off[u] = r * dstride + c4 * 4;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (base + u * 256 >= total) continue;
            float *d = dst + off[u];
            if constexpr (VEC)
                *(float4 *)d = t[u];
            else
            {
                d[0] = t[u].x; d[1] = t[u].y; d[2] = t[u].z; d[3] = t[u].w;
            }
        }
    }
}

// P . V over the (up to) 64 keys of one staged V block for ND d tiles of a wave: four key pairs per round.  Every LDS read of a
// round is issued before the first is waited for — as raw ds_read instructions: hipcc sinks a plain (or volatile) read under the
// select that follows it, a branch and a wait per read, 2 000 cycles per round of 20 MFMAs — at clamped rows; the MFMAs are
// unguarded (a tile the wave does not own multiplies zeros).
__device__ __forceinline__ float lds_read_f32(const float *p)
{
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p));
    return v;
}
template <int ND, int NDMAX>
__device__ __forceinline__ void att_pv_block(floatx16m (&oacc)[NDMAX], const float *pkb, const float *Vs, int dk, int hh, int nk,
                                             const int (&doff)[NDMAX], const bool (&dok)[NDMAX])
{
    const int npb = (nk + 1) >> 1;
    float a[2][4], bv[2][4][ND];
    // round = four key pairs.  A wave issues in order: the reads and address arithmetic of round r + 1 and the selects of pair
    // u + 1 sit BETWEEN the MFMAs of pair u, where they issue under the matrix pipe's 64 cycles per MFMA — grouped in front of the
    // round's 20 MFMAs (this loop's first form) they ran with the pipe idle: 2 000 cycles per round for 1 280 of matrix work
    auto issue_pair = [&](float &ar, float (&br)[ND], int s0, int u) {
        const int kl = 2 * (s0 + u) + hh;
        const int klc = kl < nk ? kl : 0;
        ar = lds_read_f32(pkb + (size_t)klc * 64);
        const float *vr = Vs + klc * dk;
#pragma unroll
        for (int i = 0; i < ND; i++) br[i] = lds_read_f32(vr + doff[i]);
    };
    auto arrive = [&](float (&ar)[4], float (&br)[4][ND]) {
        // one wait for the round; every value passes through an asm statement behind it, so no use can move ahead of it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0]), "+v"(ar[1]), "+v"(ar[2]), "+v"(ar[3]));
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < ND; i++) asm volatile("" : "+v"(br[u][i]));
    };
    auto select_pair = [&](float &ar, float (&br)[ND], int s0, int u) {
        const bool kin = 2 * (s0 + u) + hh < nk;
        ar = kin ? ar : 0.f;
#pragma unroll
        for (int i = 0; i < ND; i++) br[i] = (kin && dok[i]) ? br[i] : 0.f;
    };
    auto round = [&](float (&ac)[4], float (&bc)[4][ND], float (&an)[4], float (&bn)[4][ND], int s0) {
        const bool more = s0 + 4 < npb;
        arrive(ac, bc);
        select_pair(ac[0], bc[0], s0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            if (more) issue_pair(an[u], bn[u], s0 + 4, u);
            if (u < 3) select_pair(ac[u + 1], bc[u + 1], s0, u + 1);
#pragma unroll
            for (int i = 0; i < ND; i++) oacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], bc[u][i], oacc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int u = 0; u < 4; u++) issue_pair(a[0][u], bv[0][u], 0, u);
    for (int s0 = 0; s0 < npb; s0 += 8)
    {
        round(a[0], bv[0], a[1], bv[1], s0);
        if (s0 + 4 >= npb) break;
        round(a[1], bv[1], a[0], bv[0], s0 + 4);
    }
}

constexpr int ATT_NS_MAX = 144;          // dk <= 288
constexpr int ATT_NDT_MAX = 5;           // d tiles per wave: dk <= 320
constexpr int ATT_LDS_MAX = 160 * 1024 - 4096;

__global__ __launch_bounds__(256) void attention_mfma_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                             const float *__restrict__ v, int ld, int dk, float inv_temp,
                                                             float *__restrict__ o, int ldo, const Segs segs, int dbg_arg)
{
    const int dbg = ZV_DBGBITS(dbg_arg);        // timing-only ablation bits: diagnostic builds only (kernels.h)
    extern __shared__ __attribute__((aligned(16))) float att_sm[];
    __shared__ float redm[4][64];
    __shared__ double redd[4][64];
    const Seg sg = seg_at(segs, blockIdx.z);
    const int n = sg.rows;
    const int q0 = blockIdx.x * 64;
    if (q0 >= n) return;
    const int h = blockIdx.y;
    const size_t rb = (size_t)sg.row0;
    const float *qs = q + rb * ld + h * dk, *ks = k + rb * ld + h * dk, *vs = v + rb * ld + h * dk;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qt = wave & 1, kh = wave >> 1;
    const int ql = lane & 31, hh = lane >> 5;
    const int ns = dk >> 1, KSTR = dk | 1;                 // odd row stride: conflict-free column reads
    float *Ks = att_sm;                                    // [64][KSTR]
    float *S = att_sm + 64 * KSTR + ((64 * KSTR) & 1);     // [n_pad][64]
    const int nkt = (n + 31) >> 5;

    // ---- this lane's Q fragment: B[k = 2s + hh][col = query ql].  The 64 query rows go through the K region first (coalesced
    // 16-byte loads, all in flight at once): read straight from memory, a lane's fragment is 66 loads of 16 bytes at a row's stride,
    // 64 separate segments per instruction.  (Rows past the utterance are zeros; their scores are never used.)
    float qreg[ATT_NS_MAX];
    const int c4n = dk >> 2;
    // a 64-row block of K or V in registers: requested a phase ahead of the barrier pair that lets it into LDS
    constexpr int ATT_NPV = (64 * (2 * ATT_NS_MAX / 4) + 255) / 256;
    float4 vpre[ATT_NPV];
    const int vtotal = 64 * c4n;
    auto blk_load = [&](const float *src, int kb) {
#pragma unroll
        for (int p = 0; p < ATT_NPV; p++)
        {
            const int idx = tid + p * 256 < vtotal ? tid + p * 256 : vtotal - 1;
            const int r = idx / c4n, c4 = idx - r * c4n;
            const int key = kb + r;
            vpre[p] = *(const float4 *)(src + (size_t)(key < n ? key : n - 1) * ld + c4 * 4);
        }
    };
    blk_load(ks, 0);            // the first K block travels under the Q rows
    {
        att_stage64<false>(Ks, KSTR, qs, ld, q0, n, c4n, tid);
        __syncthreads();
        const float *qp = Ks + (qt * 32 + ql) * KSTR + hh;
#pragma unroll
        for (int j = 0; j < ATT_NS_MAX; j++)
        {
            const int jc = j < ns ? j : 0;
            qreg[j] = lds_read_f32(qp + 2 * jc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < ATT_NS_MAX; j++)
        {
            asm volatile("" : "+v"(qreg[j]));
            qreg[j] = j < ns ? qreg[j] : 0.f;
        }
    }

    // ---- scores (the loop's first barrier also ends the reads of the Q rows above)
    for (int kt0 = 0; kt0 < nkt; kt0 += 2)
    {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < ATT_NPV; p++)
        {
            const int idx = tid + p * 256;
            if (idx >= vtotal) continue;
            const int r = idx / c4n, c4 = idx - r * c4n;
            const float4 t = kt0 * 32 + r < n ? vpre[p] : make_float4(0.f, 0.f, 0.f, 0.f);
            float *d = Ks + r * KSTR + c4 * 4;
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        }
        __syncthreads();
        if (kt0 + 2 < nkt) blk_load(ks, (kt0 + 2) * 32);      // the next block under this block's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const int kt = kt0 + kh;
        if (kt < nkt && !(dbg & 1))
        {
            floatx16m acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            const float *kr = Ks + (kh * 32 + ql) * KSTR + hh;
            // groups of four steps (ns is even: a last group of two), the next group's K operands requested ahead of this
            // group's MFMAs (round 2 guarded every step: a branch and an exposed LDS round trip per MFMA)
            float kc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) kc[j] = kr[2 * j];
#pragma unroll
            for (int s = 0; s < ATT_NS_MAX; s += 4)
            {
                if (s + 4 <= ns)
                {
                    float kn[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) kn[j] = kr[2 * (s + 4 + j)];       // (past the last group: read, never used)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[j], qreg[s + j], acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; j++) kc[j] = kn[j];
                }
                else if (s + 2 <= ns)
                {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[0], qreg[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[1], qreg[s + 1], acc, 0, 0, 0);
                }
            }
            // D[row = key (r&3) + 8*(r>>2) + 4*hh][col = query ql]
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key < n) S[(size_t)key * 64 + qt * 32 + ql] = acc[r] * inv_temp;
            }
        }
    }
    __syncthreads();

    // the V rows of the first 64 keys are requested now and travel under the softmax; every later block under the MFMAs of the
    // block before it
    auto v_write = [&](int kb) {
#pragma unroll
        for (int p = 0; p < ATT_NPV; p++)
        {
            const int idx = tid + p * 256;
            if (idx >= vtotal) continue;
            const int r = idx / c4n, c4 = idx - r * c4n;
            *(float4 *)(att_sm + r * dk + c4 * 4) = kb + r < n ? vpre[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (!(dbg & (4 | 16))) blk_load(vs, 0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- softmax over keys: thread (query = tid & 63, part = tid >> 6) walks keys part, part + 4, ...
    if (!(dbg & 2))
    {
        const int qi = tid & 63, part = tid >> 6;
        float mx = -INFINITY;
        for (int key = part; key < n; key += 4) mx = fmaxf(mx, S[(size_t)key * 64 + qi]);
        redm[part][qi] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(redm[0][qi], redm[1][qi]), fmaxf(redm[2][qi], redm[3][qi]));
        // the sum of the exponentials: four interleaved partial sums (key mod 4), each in key order, then
        // (p0 + p1) + (p2 + p3) — the scalar kernel below adds in exactly this order, so the two kernels give the same bits
        double sum = 0.0;
        for (int key = part; key < n; key += 4)
        {
            const float e = expf(S[(size_t)key * 64 + qi] - mx);
            S[(size_t)key * 64 + qi] = e;
            sum += (double)e;
        }
        redd[part][qi] = sum;
        __syncthreads();
        const float inv = (float)(1.0 / ((redd[0][qi] + redd[1][qi]) + (redd[2][qi] + redd[3][qi])));
        for (int key = part; key < n; key += 4) S[(size_t)key * 64 + qi] = S[(size_t)key * 64 + qi] * inv;
    }
    __syncthreads();

    // ---- P . V: wave (qt, dh = kh) owns d tiles dh, dh + 2, ...; the V rows of 64 keys at a time go through the LDS region the
    // K tiles used (coalesced 16-byte loads, every one of a block in flight at once — round 2 read V straight from L2, 4 bytes
    // per lane inside the MFMA loop: a round trip per four key pairs, half of the kernel's time), keys in ascending order as before
    const int ndt = (dk + 31) >> 5;
    floatx16m oacc[ATT_NDT_MAX];
#pragma unroll
    for (int i = 0; i < ATT_NDT_MAX; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[i][r] = 0.f;
    const float *pp = S + qt * 32 + ql;
    float *Vs = att_sm;                                    // [64][dk]
    int doff[ATT_NDT_MAX];
    bool dok[ATT_NDT_MAX];
#pragma unroll
    for (int i = 0; i < ATT_NDT_MAX; i++)
    {
        const int d = (kh + 2 * i) * 32 + ql;
        dok[i] = kh + 2 * i < ndt && d < dk;
        doff[i] = dok[i] ? d : 0;
    }
    for (int kb = 0; kb < ((dbg & 4) ? 0 : n); kb += 64)
    {
        if (kb) __syncthreads();                           // the previous block is consumed (the first: S is complete)
        if (!(dbg & 16)) v_write(kb);
        __syncthreads();
        if (!(dbg & 16) && kb + 64 < n) blk_load(vs, kb + 64);
        __builtin_amdgcn_sched_barrier(0);
        const int nk = n - kb < 64 ? n - kb : 64;
        const float *pkb = pp + (size_t)kb * 64;
        if (dbg & 8) continue;
        switch ((ndt + 1) >> 1)                            // d tiles of the wider wave (kh = 0): straight-line code per count
        {
        case 1: att_pv_block<1, ATT_NDT_MAX>(oacc, pkb, Vs, dk, hh, nk, doff, dok); break;
        case 2: att_pv_block<2, ATT_NDT_MAX>(oacc, pkb, Vs, dk, hh, nk, doff, dok); break;
        case 3: att_pv_block<3, ATT_NDT_MAX>(oacc, pkb, Vs, dk, hh, nk, doff, dok); break;
        case 4: att_pv_block<4, ATT_NDT_MAX>(oacc, pkb, Vs, dk, hh, nk, doff, dok); break;
        default: att_pv_block<5, ATT_NDT_MAX>(oacc, pkb, Vs, dk, hh, nk, doff, dok); break;
        }
    }
    // D[row = query (r&3) + 8*(r>>2) + 4*hh][col = d ql]
    float *os = o + rb * ldo + h * dk;
#pragma unroll
    for (int i = 0; i < ATT_NDT_MAX; i++)
    {
        const int d = (kh + 2 * i) * 32 + ql;
        if (kh + 2 * i < ndt && d < dk)
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int qi = q0 + qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (qi < n) os[(size_t)qi * ldo + d] = oacc[i][r];
            }
    }
}

// The same operation with scalar fma chains: one block per (query, head) — the same bits as the matrix-core kernel (every
// dot product is the same k-ordered fma chain, the softmax sum is associated the same way).  Used where that kernel does
// not fit (head width not a multiple of 4 or above 288, more keys than its LDS score tile holds) and for a single short
// utterance, where (queries x heads) blocks fill the chip and a handful of 64-query workgroups does not.
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, int ld, int dk, float inv_temp,
                                                        float *__restrict__ o, int ldo, const Segs segs)
{
    extern __shared__ float sm[];
    const Seg sg = seg_at(segs, blockIdx.z);
    const int n = sg.rows;
    const int iq = blockIdx.x, h = blockIdx.y;
    if (iq >= n) return;
    float *qs = sm;                 // dk
    float *p = sm + dk;             // n
    __shared__ double redd[4];
    __shared__ float redf[4];
    const size_t rb = (size_t)sg.row0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *qrow = q + (rb + iq) * ld + h * dk;
    for (int d = tid; d < dk; d += 256) qs[d] = qrow[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int ik = tid; ik < n; ik += 256)
    {
        const float *krow = k + (rb + ik) * ld + h * dk;
        float acc = 0.f;
        for (int d = 0; d < dk; d++) acc = fmaf(qs[d], krow[d], acc);
        acc = acc * inv_temp;
        p[ik] = acc;
        mx = fmaxf(mx, acc);
    }
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    for (int ik = tid; ik < n; ik += 256) p[ik] = expf(p[ik] - mx);
    __syncthreads();
    if (tid < 4)            // the association of attention_mfma_kernel: four interleaved partial sums in key order
    {
        double sum = 0.0;
        for (int ik = tid; ik < n; ik += 4) sum += (double)p[ik];
        redd[tid] = sum;
    }
    __syncthreads();
    const float inv = (float)(1.0 / ((redd[0] + redd[1]) + (redd[2] + redd[3])));
    for (int ik = tid; ik < n; ik += 256) p[ik] = p[ik] * inv;
    __syncthreads();
    for (int d = tid; d < dk; d += 256)
    {
        const float *vcol = v + rb * ld + h * dk + d;
        float acc = 0.f;
        for (int ik = 0; ik < n; ik++) acc = fmaf(p[ik], vcol[(size_t)ik * ld], acc);
        o[(rb + iq) * ldo + h * dk + d] = acc;
    }
}

hipError_t launch_attention(hipStream_t s, const float *q, const float *k, const float *v, int ld, int H, int dk,
                            float inv_temp, float *o, int ldo, const Segs &segs)
{
    if (segs.nseg < 1) return hipErrorInvalidValue;
    const int n = segs.max_rows;
    const int KSTR = dk | 1;
    const size_t lds_mfma = ((size_t)64 * KSTR + 1 + (size_t)((n + 31) & ~31) * 64) * sizeof(float);
    const bool force_scalar = knob(ZV_ATT_SCALAR) != 0, force_mfma = knob(ZV_ATT_MFMA) != 0;     // test hooks
    const long wgs_mfma = (long)((n + 63) / 64) * H * segs.nseg;
    if (!force_scalar && (dk & 3) == 0 && dk <= 2 * ATT_NS_MAX && (ld & 3) == 0 && lds_mfma <= (size_t)ATT_LDS_MAX &&
        (wgs_mfma >= 48 || force_mfma))
    {
        auto kern = attention_mfma_kernel;
        if (lds_mfma > 48 * 1024)
        {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mfma);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((n + 63) / 64, H, segs.nseg), dim3(256), lds_mfma, s, q, k, v, ld, dk, inv_temp, o, ldo, segs, diag_bits() >> 8);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(dk + n) * sizeof(float);
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attention_kernel, dim3(n, H, segs.nseg), dim3(256), lds, s, q, k, v, ld, dk, inv_temp, o, ldo, segs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// y = LayerNorm(x + res) * w + b over channels: one wave per row (reference src/fs2encoder.cpp:132-137,
// ggml_norm: mean and biased variance of (x - mean) accumulated in f64)
// Optional tail (LnTail, all launch-count savers for short utterances — the same operations in the same order as the separate
// kernels they replace, hence the same bits):
//   post   : y += post[segment][c]            (features = encoder output + style_embed, reference src/fs2encoder.cpp:550-552)
//   dot_w  : pred[row] = dot(y[row], dot_w) + dot_b[0]   (VariancePredictor linear_layer, :434-435; lane l sums channels l, l + 64, ...
//            in that order and the wave sum follows: rowdot_kernel's chain)
//   emb    : bucket[row] = clamp((int)((double)(pred * (nbins - 1)) + 0.5)); feat[row][:] += emb[bucket][:]   (:442-474, 565-569)
struct LnTail
{
    const float *post;
    int          post_seg;
    const float *dot_w, *dot_b;
    float       *pred;
    const float *emb;
    int          nbins, embC;
    float       *feat;
    int          ldf;
    int32_t     *bucket;
};

template <bool TAIL>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ res, int ldr, int C, int Cp,
                                                            const float *__restrict__ w, const float *__restrict__ b,
                                                            float eps, float *__restrict__ y, int ldy, const Segs segs, const LnTail tail)
{
    const Seg sg = seg_at(segs, blockIdx.y);
    const int rloc = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rloc >= sg.rows) return;
    const size_t row = (size_t)sg.row0 + rloc;
    const float *xr = x + row * ldx;
    const float *rr = res ? res + row * ldr : nullptr;
    // rows of up to 64 * LN_MAXE channels are read once and kept in registers for the three passes (same per-lane
    // summation order as the plain three-pass form below, so the same bits)
    constexpr int LN_MAXE = 12;
    if (C <= 64 * LN_MAXE)
    {
        // (every load unconditional at a clamped column, then selects: a load under a run-time condition becomes a branch with a
        // wait behind it — 24 + 24 exposed round trips per row in round 2's form of this loop, 12.5 us per launch)
        float v[LN_MAXE], rv[LN_MAXE], wv[LN_MAXE], bv[LN_MAXE];
        float pv[TAIL ? LN_MAXE : 1], dv[TAIL ? LN_MAXE : 1];
        // (the tail's vectors come with the other loads: unconditional, from a valid dummy vector where a part is absent)
        const float *post = nullptr, *dotw = nullptr;
        if constexpr (TAIL)
        {
            post = tail.post ? tail.post + (size_t)blockIdx.y * tail.post_seg : w;
            dotw = tail.dot_w ? tail.dot_w : w;
        }
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
        {
            const int c = lane + 64 * e, cc = c < C ? c : C - 1;
            v[e] = xr[cc];
            rv[e] = rr ? rr[cc] : 0.f;
            wv[e] = w[cc];
            bv[e] = b[cc];
            if constexpr (TAIL)
            {
                pv[e] = post[cc];
                dv[e] = dotw[cc];
            }
        }
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
        {
            const int c = lane + 64 * e;
            v[e] = (c < C) ? (rr ? v[e] + rv[e] : v[e]) : 0.f;
        }
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
            if (lane + 64 * e < C) s += (double)v[e];
        s = wave_sum(s);
        const float mean = (float)(s / (double)C);
        double s2 = 0.0;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
            if (lane + 64 * e < C)
            {
                const float d = v[e] - mean;
                s2 += (double)(d * d);
            }
        s2 = wave_sum(s2);
        const float var = (float)(s2 / (double)C);
        const float scale = 1.0f / sqrtf(var + eps);
        float dacc = 0.f;
        const bool has_post = TAIL && tail.post != nullptr;
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
        {
            const int c = lane + 64 * e;
            if (c < C)
            {
                float t = (v[e] - mean) * scale;
                t = wv[e] * t;
                t = t + bv[e];
                if constexpr (TAIL)
                {
                    if (has_post) t = t + pv[e];
                    dacc = fmaf(t, dv[e], dacc);
                }
                y[row * ldy + c] = t;
            }
        }
        for (int c = C + lane; c < Cp; c += 64) y[row * ldy + c] = 0.f;
        if (TAIL && tail.dot_w)
        {
            dacc = wave_sum_f(dacc);
            const float pr = dacc + tail.dot_b[0];
            if (lane == 0) tail.pred[row] = pr;
            if (tail.emb)
            {
                const int bin_max = tail.nbins - 1;
                const float p = pr * (float)bin_max;
                int yb = (int)((double)p + 0.5);          // truncating cast of x + 0.5 (double), not round-half-even
                yb = yb < 0 ? 0 : (yb > bin_max ? bin_max : yb);
                if (lane == 0) tail.bucket[row] = yb;
                float *fr = tail.feat + row * tail.ldf;
                const float *er = tail.emb + (size_t)yb * tail.embC;
                for (int c = lane; c < tail.embC; c += 64) fr[c] = fr[c] + er[c];
            }
        }
        return;
    }
    double s = 0.0;
    for (int c = lane; c < C; c += 64) s += (double)(rr ? xr[c] + rr[c] : xr[c]);
    s = wave_sum(s);
    const float mean = (float)(s / (double)C);
    double s2 = 0.0;
    for (int c = lane; c < C; c += 64)
    {
        const float v = (rr ? xr[c] + rr[c] : xr[c]) - mean;
        s2 += (double)(v * v);
    }
    s2 = wave_sum(s2);
    const float var = (float)(s2 / (double)C);
    const float scale = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < C; c += 64)
    {
        float v = ((rr ? xr[c] + rr[c] : xr[c]) - mean) * scale;
        v = w[c] * v;
        y[row * ldy + c] = v + b[c];
    }
    for (int c = C + lane; c < Cp; c += 64) y[row * ldy + c] = 0.f;
}

hipError_t launch_add_layernorm(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int C, int Cp,
                                const float *w, const float *b, float eps, float *y, int ldy, const Segs &segs)
{
    LnTail tail;
    memset(&tail, 0, sizeof(tail));
    hipLaunchKernelGGL(add_layernorm_kernel<false>, dim3((segs.max_rows + 3) / 4, segs.nseg), dim3(256), 0, s, x, ldx, res, ldr, C, Cp, w, b,
                       eps, y, ldy, segs, tail);
    return hipGetLastError();
}

bool layernorm_tail_ok(int C) { return C <= 64 * 12; }

hipError_t launch_layernorm_tail(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int C, int Cp, const float *w,
                                 const float *b, float eps, float *y, int ldy, const Segs &segs, const float *post, int post_seg,
                                 const float *dot_w, const float *dot_b, float *pred, const float *emb, int nbins, int embC, float *feat,
                                 int ldf, int32_t *bucket)
{
    if (!layernorm_tail_ok(C)) return hipErrorInvalidValue;        // the tail lives in the rows-in-registers form
    if (emb && (!dot_w || !feat || !bucket || nbins < 1)) return hipErrorInvalidValue;
    if (dot_w && (!dot_b || !pred)) return hipErrorInvalidValue;
    LnTail tail;
    tail.post = post;
    tail.post_seg = post_seg;
    tail.dot_w = dot_w;
    tail.dot_b = dot_b;
    tail.pred = pred;
    tail.emb = emb;
    tail.nbins = nbins;
    tail.embC = embC;
    tail.feat = feat;
    tail.ldf = ldf;
    tail.bucket = bucket;
    hipLaunchKernelGGL(add_layernorm_kernel<true>, dim3((segs.max_rows + 3) / 4, segs.nseg), dim3(256), 0, s, x, ldx, res, ldr, C, Cp, w, b,
                       eps, y, ldy, segs, tail);
    return hipGetLastError();
}

__global__ void add_rowvec_kernel(float *__restrict__ x, int ld, int C, const float *__restrict__ v, int v_seg, const Segs segs)
{
    const Seg sg = seg_at(segs, blockIdx.y);
    if ((int)blockIdx.x >= sg.rows) return;
    const size_t row = (size_t)sg.row0 + blockIdx.x;
    const float *vv = v + (size_t)blockIdx.y * v_seg;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[row * ld + c] = x[row * ld + c] + vv[c];
}

hipError_t launch_add_rowvec(hipStream_t s, float *x, int ld, int C, const float *v, int v_seg, const Segs &segs)
{
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(segs.max_rows, segs.nseg), dim3(256), 0, s, x, ld, C, v, v_seg, segs);
    return hipGetLastError();
}

// VariancePredictor linear_layer (reference src/fs2encoder.cpp:434-435): one wave per token
__global__ __launch_bounds__(256) void rowdot_kernel(const float *__restrict__ x, int ld, int C, const float *__restrict__ w,
                                                     const float *__restrict__ b, float *__restrict__ y, const Segs segs)
{
    const Seg sg = seg_at(segs, blockIdx.y);
    const int rloc = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (rloc >= sg.rows) return;
    const size_t row = (size_t)sg.row0 + rloc;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(x[row * ld + c], w[c], acc);
    acc = wave_sum_f(acc);
    if (lane == 0) y[row] = acc + b[0];
}

hipError_t launch_rowdot(hipStream_t s, const float *x, int ld, int C, const float *w, const float *b, float *y, const Segs &segs)
{
    hipLaunchKernelGGL(rowdot_kernel, dim3((segs.max_rows + 3) / 4, segs.nseg), dim3(256), 0, s, x, ld, C, w, b, y, segs);
    return hipGetLastError();
}

// ggml_zv_mul_clamp_to_i32 + get_rows + add (reference src/fs2encoder.cpp:442-474,565-569)
__global__ void bucket_embed_add_kernel(const float *__restrict__ pred, int nbins, const float *__restrict__ emb, int C,
                                        float *__restrict__ x, int ld, int32_t *__restrict__ bucket, const Segs segs)
{
    const Seg sg = seg_at(segs, blockIdx.y);
    if ((int)blockIdx.x >= sg.rows) return;
    const size_t n = (size_t)sg.row0 + blockIdx.x;
    const int bin_max = nbins - 1;
    float p = pred[n];
    p = p * (float)bin_max;
    int y = (int)((double)p + 0.5);          // truncating cast of x + 0.5 (double), not round-half-even
    y = y < 0 ? 0 : (y > bin_max ? bin_max : y);
    if (threadIdx.x == 0) bucket[n] = y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[n * ld + c] = x[n * ld + c] + emb[(size_t)y * C + c];
}

hipError_t launch_bucket_embed_add(hipStream_t s, const float *pred, int nbins, const float *emb, int C, float *x, int ld,
                                   int32_t *bucket, const Segs &segs)
{
    hipLaunchKernelGGL(bucket_embed_add_kernel, dim3(segs.max_rows, segs.nseg), dim3(256), 0, s, pred, nbins, emb, C, x, ld, bucket,
                       segs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Length regulator on the device (the reference does it on the host, src/fs2encoder.cpp:611-654):
//   dur_i = (int)((float)(exp(logdur_i) - 1.0) + 0.5) for the first num_phonemes (= aux) tokens of the utterance;
//   frame f belongs to the token whose cumulative duration first exceeds f; frames past the total (or past T) are zero.
__global__ __launch_bounds__(1024) void lr_scan_kernel(const float *__restrict__ logdur, int32_t *__restrict__ cum,
                                                       int32_t *__restrict__ n_frames, const Segs tokens, const Segs frames)
{
    __shared__ int buf[1024];
    const Seg tk = seg_at(tokens, blockIdx.x), fr = seg_at(frames, blockIdx.x);
    const int n = tk.rows, T = fr.rows;
    if (n <= 0) return;
    const int nwalk = tk.aux < n ? tk.aux : n;
    const float *ld_ = logdur + tk.row0;
    int32_t *cm = cum + tk.row0;
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n; base += 1024)
    {
        const int i = base + tid;
        int d = 0;
        if (i < nwalk)
        {
            const float dur = (float)(exp((double)ld_[i]) - 1.0);
            d = (int)((double)dur + 0.5);
            if (d < 0) d = 0;
            if (d > T) d = T;          // keeps the running sum far from int overflow; frames stop at T anyway
        }
        buf[tid] = d;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1)
        {
            const int v = (tid >= o) ? buf[tid - o] : 0;
            __syncthreads();
            buf[tid] += v;
            __syncthreads();
        }
        if (i < n) cm[i] = carry + buf[tid];
        carry += buf[1023];
        __syncthreads();
    }
    if (tid == 0) n_frames[blockIdx.x] = carry < T ? carry : T;
}

__global__ void lr_gather_kernel(const float *__restrict__ feat, int ld, const int32_t *__restrict__ cum, int C,
                                 float *__restrict__ hidden, int ldh, const Segs tokens, const Segs frames)
{
    const Seg tk = seg_at(tokens, blockIdx.y), fr = seg_at(frames, blockIdx.y);
    const int f = blockIdx.x;
    if (f >= fr.rows) return;
    const int n = tk.rows;
    const int32_t *cm = cum + tk.row0;
    // first token i with cum[i] > f
    int lo = 0, hi = n;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (cm[mid] > f) hi = mid; else lo = mid + 1;
    }
    const bool live = lo < n;
    const float *src = feat + ((size_t)tk.row0 + lo) * ld;
    float *dst = hidden + ((size_t)fr.row0 + f) * ldh;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dst[c] = live ? src[c] : 0.f;
}

// the same gather, 16 frames per workgroup: the utterance's cumulative durations are read into LDS once (coalesced) and searched
// there — the kernel above walks them in global memory, eight dependent round trips per frame — and the rows move in 16-byte pieces
__global__ __launch_bounds__(256) void lr_gather16_kernel(const float *__restrict__ feat, int ld, const int32_t *__restrict__ cum, int C,
                                                          float *__restrict__ hidden, int ldh, const Segs tokens, const Segs frames)
{
    extern __shared__ int32_t lr_cs[];
    const Seg tk = seg_at(tokens, blockIdx.y), fr = seg_at(frames, blockIdx.y);
    const int f0 = blockIdx.x * 16;
    if (f0 >= fr.rows) return;
    const int n = tk.rows;
    for (int i = threadIdx.x; i < n; i += 256) lr_cs[i] = cum[tk.row0 + i];
    __syncthreads();
    const int f = f0 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
    if (f >= fr.rows) return;
    int lo = 0, hi = n;                  // first token i with cum[i] > f
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (lr_cs[mid] > f) hi = mid; else lo = mid + 1;
    }
    const bool live = lo < n;
    const float4 *src = (const float4 *)(feat + ((size_t)tk.row0 + (live ? lo : 0)) * ld);
    float4 *dst = (float4 *)(hidden + ((size_t)fr.row0 + f) * ldh);
    for (int c4 = l16; c4 < (C >> 2); c4 += 16)
    {
        const float4 v = src[c4];
        dst[c4] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Scan and gather in ONE launch for utterances of at most 1 024 tokens: every workgroup (16 frames) recomputes its utterance's
// rounded durations and their inclusive scan in LDS — a few hundred integer operations against a launch boundary — and the first
// workgroup of a segment also stores the scan (the `cum` tap) and the frame count.  Integer sums: any scan order, the same values.
__global__ __launch_bounds__(256) void lr_fused16_kernel(const float *__restrict__ feat, int ld, const float *__restrict__ logdur, int C,
                                                         float *__restrict__ hidden, int ldh, int32_t *__restrict__ cum,
                                                         int32_t *__restrict__ n_frames, const Segs tokens, const Segs frames)
{
    __shared__ int32_t lr_cs[1024];
    __shared__ int32_t lr_part[256];
    const Seg tk = seg_at(tokens, blockIdx.y), fr = seg_at(frames, blockIdx.y);
    const int f0 = blockIdx.x * 16;
    const int n = tk.rows, T = fr.rows;
    if (f0 >= T) return;
    const int nwalk = tk.aux < n ? tk.aux : n;
    const int tid = threadIdx.x;
    // four consecutive tokens per thread: durations (lr_scan_kernel's arithmetic), their running sums, the thread's total
    int d[4], tot = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const int i = tid * 4 + j;
        int v = 0;
        if (i < nwalk)
        {
            const float dur = (float)(exp((double)logdur[tk.row0 + i]) - 1.0);
            v = (int)((double)dur + 0.5);
            if (v < 0) v = 0;
            if (v > T) v = T;
        }
        tot += v;
        d[j] = tot;
    }
    lr_part[tid] = tot;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1)
    {
        const int v = tid >= o ? lr_part[tid - o] : 0;
        __syncthreads();
        lr_part[tid] += v;
        __syncthreads();
    }
    const int base = tid ? lr_part[tid - 1] : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) lr_cs[tid * 4 + j] = base + d[j];
    __syncthreads();
    if (blockIdx.x == 0)
    {
        for (int i = tid; i < n; i += 256) cum[tk.row0 + i] = lr_cs[i];
        if (tid == 0)
        {
            const int total = lr_part[255];
            n_frames[blockIdx.y] = total < T ? total : T;
        }
    }
    const int f = f0 + (tid >> 4), l16 = tid & 15;
    if (f >= T) return;
    int lo = 0, hi = n;                  // first token i with cum[i] > f
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (lr_cs[mid] > f) hi = mid; else lo = mid + 1;
    }
    const bool live = lo < n;
    const float4 *src = (const float4 *)(feat + ((size_t)tk.row0 + (live ? lo : 0)) * ld);
    float4 *dst = (float4 *)(hidden + ((size_t)fr.row0 + f) * ldh);
    for (int c4 = l16; c4 < (C >> 2); c4 += 16)
    {
        const float4 v = src[c4];
        dst[c4] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

hipError_t launch_length_regulator(hipStream_t s, const float *feat, int ld, const float *logdur, int C, float *hidden,
                                   int ldh, int32_t *cum, int32_t *n_frames, const Segs &tokens, const Segs &frames)
{
    if (tokens.nseg != frames.nseg || tokens.nseg < 1) return hipErrorInvalidValue;
    if ((C & 3) == 0 && (ld & 3) == 0 && (ldh & 3) == 0 && tokens.max_rows >= 1 && tokens.max_rows <= 1024 && frames.max_rows >= 1)
    {
        hipLaunchKernelGGL(lr_fused16_kernel, dim3((frames.max_rows + 15) / 16, frames.nseg), dim3(256), 0, s, feat, ld, logdur, C, hidden,
                           ldh, cum, n_frames, tokens, frames);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(lr_scan_kernel, dim3(tokens.nseg), dim3(1024), 0, s, logdur, cum, n_frames, tokens, frames);
    if ((C & 3) == 0 && (ld & 3) == 0 && (ldh & 3) == 0 && (size_t)tokens.max_rows * 4 <= 48 * 1024 && tokens.max_rows >= 1)
        hipLaunchKernelGGL(lr_gather16_kernel, dim3((frames.max_rows + 15) / 16, frames.nseg), dim3(256), (size_t)tokens.max_rows * 4, s,
                           feat, ld, cum, C, hidden, ldh, tokens, frames);
    else
        hipLaunchKernelGGL(lr_gather_kernel, dim3(frames.max_rows, frames.nseg), dim3(256), 0, s, feat, ld, cum, C, hidden, ldh, tokens,
                           frames);
    return hipGetLastError();
}

}  // namespace zv

// misc_kernels.hip — the non-conv kernels of the fixed schedule: InstanceNorm statistics, f32 linear
// layers, attention, LayerNorm, variance-adaptor bucketing and the device length regulator.
// All of them are latency/bandwidth-trivial next to the convs (convs are > 90 % of the
// reference's time); they exist so that no stage ever goes back to the host between kernels.
#include "kernels.h"

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
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm1d statistics over time (ggml_norm: mean and biased variance of (x - mean) accumulated in f64,
// scale = 1/sqrtf(var + eps); reference ggml-cpu.c:6906-6923).  1024 threads = 16 channels (64-B row segments)
// x 64 time lanes: the kernel is pure latency (a few MB), so it is cut into many short strided loops.
__global__ __launch_bounds__(1024) void in_stats_kernel(const float *__restrict__ x, int ld, int L, int C, float eps,
                                                        float *__restrict__ stat)
{
    __shared__ double red[64][17];
    __shared__ float meanv[16];
    const int cl = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < C;
    double s = 0.0;
    if (ok)
        for (int t = ty; t < L; t += 64) s += (double)x[(size_t)t * ld + c];
    red[ty][cl] = s;
    __syncthreads();
    if (ty == 0)
    {
        double tot = 0.0;
#pragma unroll 8
        for (int i = 0; i < 64; i++) tot += red[i][cl];
        meanv[cl] = (float)(tot / (double)L);
    }
    __syncthreads();
    const float mean = meanv[cl];
    double s2 = 0.0;
    if (ok)
        for (int t = ty; t < L; t += 64)
        {
            const float v = x[(size_t)t * ld + c] - mean;
            s2 += (double)(v * v);
        }
    __syncthreads();
    red[ty][cl] = s2;
    __syncthreads();
    if (ty == 0 && ok)
    {
        double tot = 0.0;
#pragma unroll 8
        for (int i = 0; i < 64; i++) tot += red[i][cl];
        const float var = (float)(tot / (double)L);
        stat[2 * c] = mean;
        stat[2 * c + 1] = 1.0f / sqrtf(var + eps);
    }
}

hipError_t launch_in_stats(hipStream_t s, const float *x, int ld, int L, int C, float eps, float *stat)
{
    hipLaunchKernelGGL(in_stats_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, x, ld, L, C, eps, stat);
    return hipGetLastError();
}

__global__ void norm_apply_kernel(const float *__restrict__ x, int ldx, int L, int C, const float *__restrict__ stat,
                                  const float *__restrict__ g, const float *__restrict__ b, float *__restrict__ y, int ldy)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (c >= C) return;
    float v = (x[(size_t)t * ldx + c] - stat[2 * c]) * stat[2 * c + 1];
    v = v * g[c];
    v = v + b[c];
    y[(size_t)t * ldy + c] = v;
}

hipError_t launch_norm_apply(hipStream_t s, const float *x, int ldx, int L, int C, const float *stat, const float *g,
                             const float *b, float *y, int ldy)
{
    hipLaunchKernelGGL(norm_apply_kernel, dim3((C + 63) / 64, L), dim3(64), 0, s, x, ldx, L, C, stat, g, b, y, ldy);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// y[n][o] = dot(W[o][:], x[n][:]) + b[o]   (f32 weights: ggml_mul_mat + ggml_add, reference
// src/fs2encoder.cpp:77-89,127-128 and src/stylettsdec.cpp:178-179) on the f32-input matrix cores:
// v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain (exact f32, the reference's own FMA accumulation),
// one wave per 32 rows x 32 outputs.  Lane l holds A[row l&31][k + (l>>5)] / B[k + (l>>5)][col l&31].
// `extra[o]` is added after the bias (AdaIN: gamma = h[:C] + 1, src/stylettsdec.cpp:186-189).
typedef float floatx16m __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void linear_mfma_kernel(const float *__restrict__ x, int ldx, int n, int in,
                                                         const float *__restrict__ W, const float *__restrict__ b, int out,
                                                         float *__restrict__ y, int ldy, const float *__restrict__ extra)
{
    // 32 x 32 k-chunks of both operands go through LDS: global rows are read as coalesced 128-B segments, the MFMA
    // operand (one f32 per lane, rows across lanes) comes back from LDS with a 33-float row stride (conflict-free)
    __shared__ float Xs[32][33];
    __shared__ float Ws[32][33];
    const int lane = threadIdx.x;
    const int o0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int i = lane & 31, kk = lane >> 5;
    floatx16m acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float4 xv[4], wv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int idx = lane + u * 64, r = idx >> 3, c = (idx & 7) * 4;
            const bool kin = k0 + c < in;                      // `in` is a multiple of 4: a float4 is all in or all out
            const int row = n0 + r, col = o0 + r;
            xv[u] = (kin && row < n) ? *(const float4 *)(x + (size_t)row * ldx + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[u] = (kin && col < out) ? *(const float4 *)(W + (size_t)col * in + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < in; k0 += 32)
    {
        __syncthreads();                                       // previous chunk fully consumed
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int idx = lane + u * 64, r = idx >> 3, c = (idx & 7) * 4;
            Xs[r][c] = xv[u].x; Xs[r][c + 1] = xv[u].y; Xs[r][c + 2] = xv[u].z; Xs[r][c + 3] = xv[u].w;
            Ws[r][c] = wv[u].x; Ws[r][c + 1] = wv[u].y; Ws[r][c + 2] = wv[u].z; Ws[r][c + 3] = wv[u].w;
        }
        __syncthreads();
        if (k0 + 32 < in) fetch(k0 + 32);                      // next chunk's rows are in flight under this chunk's MFMAs
#pragma unroll
        for (int k = 0; k < 32; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[i][k + kk], Ws[i][k + kk], acc, 0, 0, 0);
    }
    // D: col = lane&31 (output), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (token)
    const int col = o0 + i;
    if (col >= out) return;
    const float bias = b ? b[col] : 0.f;
    const float ex = extra ? extra[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        const int t = n0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (t < n)
        {
            float v = acc[r] + bias;
            if (extra) v = v + ex;
            y[(size_t)t * ldy + col] = v;
        }
    }
}

hipError_t launch_linear(hipStream_t s, const float *x, int ldx, int n, int in, const float *W, const float *b, int out,
                         float *y, int ldy, const float *extra)
{
    if ((in & 3) || (ldx & 3)) return hipErrorInvalidValue;          // float4 row loads
    hipLaunchKernelGGL(linear_mfma_kernel, dim3((out + 31) / 32, (n + 31) / 32), dim3(64), 0, s, x, ldx, n, in, W, b, out, y,
                       ldy, extra);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Encoder::graph prologue (reference src/fs2encoder.cpp:306-324): x[n] = cat(word_emb[id], punct_emb[p]) + posenc[n]
__global__ void embed_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ puncts,
                             const float *__restrict__ wemb, int emb, const float *__restrict__ pemb, int pdim,
                             const float *__restrict__ posenc, float *__restrict__ x, int ld)
{
    const int n = blockIdx.x;
    const int E = emb + pdim;
    const int id = ids[n], p = puncts[n];
    for (int e = threadIdx.x; e < E; e += blockDim.x)
    {
        const float v = (e < emb) ? wemb[(size_t)id * emb + e] : pemb[(size_t)p * pdim + (e - emb)];
        x[(size_t)n * ld + e] = v + posenc[(size_t)n * E + e];
    }
}

hipError_t launch_embed(hipStream_t s, const int32_t *ids, const int32_t *puncts, const float *wemb, int emb,
                        const float *pemb, int pdim, const float *posenc, int n, float *x, int ld)
{
    hipLaunchKernelGGL(embed_kernel, dim3(n), dim3(256), 0, s, ids, puncts, wemb, emb, pemb, pdim, posenc, x, ld);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// ScaledDotProductAttention without mask (reference src/fs2encoder.cpp:103-123): one block per (query, head).
// scores = (q.k) * inv_temp -> softmax over keys (max-subtracted, sum in f64) -> out[d] = sum_k p[k] v[k][d].
// q,k,v are [n][H*dk] token-major; the head-major concat of the reference is the same memory layout.
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                        const float *__restrict__ v, int ld, int n, int dk, float inv_temp,
                                                        float *__restrict__ o, int ldo)
{
    extern __shared__ float sm[];
    float *qs = sm;                 // dk
    float *p = sm + dk;             // n
    __shared__ double redd[4];
    __shared__ float redf[4];
    const int iq = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *qrow = q + (size_t)iq * ld + h * dk;
    for (int d = tid; d < dk; d += 256) qs[d] = qrow[d];
    __syncthreads();
    float mx = -INFINITY;
    for (int ik = tid; ik < n; ik += 256)
    {
        const float *krow = k + (size_t)ik * ld + h * dk;
        float acc = 0.f;
        for (int d = 0; d < dk; d++) acc = fmaf(qs[d], krow[d], acc);
        acc = acc * inv_temp;
        p[ik] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max_f(mx);
    if (lane == 0) redf[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double sum = 0.0;
    for (int ik = tid; ik < n; ik += 256)
    {
        const float e = expf(p[ik] - mx);
        p[ik] = e;
        sum += (double)e;
    }
    sum = wave_sum(sum);
    if (lane == 0) redd[wv] = sum;
    __syncthreads();
    const float inv = (float)(1.0 / (redd[0] + redd[1] + redd[2] + redd[3]));
    for (int ik = tid; ik < n; ik += 256) p[ik] = p[ik] * inv;
    __syncthreads();
    for (int d = tid; d < dk; d += 256)
    {
        const float *vcol = v + h * dk + d;
        float acc = 0.f;
        for (int ik = 0; ik < n; ik++) acc = fmaf(p[ik], vcol[(size_t)ik * ld], acc);
        o[(size_t)iq * ldo + h * dk + d] = acc;
    }
}

hipError_t launch_attention(hipStream_t s, const float *q, const float *k, const float *v, int ld, int n, int H, int dk,
                            float inv_temp, float *o, int ldo)
{
    const size_t lds = (size_t)(dk + n) * sizeof(float);
    hipLaunchKernelGGL(attention_kernel, dim3(n, H), dim3(256), lds, s, q, k, v, ld, n, dk, inv_temp, o, ldo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// y = LayerNorm(x + res) * w + b over channels: one wave per row (reference src/fs2encoder.cpp:132-137,
// ggml_norm semantics as in in_stats_kernel)
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ res, int ldr, int n, int C, int Cp,
                                                            const float *__restrict__ w, const float *__restrict__ b,
                                                            float eps, float *__restrict__ y, int ldy)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const float *xr = x + (size_t)row * ldx;
    const float *rr = res ? res + (size_t)row * ldr : nullptr;
    // rows of up to 64 * LN_MAXE channels are read once and kept in registers for the three passes (same per-lane
    // summation order as the plain three-pass form below, so the same bits)
    constexpr int LN_MAXE = 12;
    if (C <= 64 * LN_MAXE)
    {
        float v[LN_MAXE];
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
        {
            const int c = lane + 64 * e;
            v[e] = (c < C) ? (rr ? xr[c] + rr[c] : xr[c]) : 0.f;
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
#pragma unroll
        for (int e = 0; e < LN_MAXE; e++)
        {
            const int c = lane + 64 * e;
            if (c < C)
            {
                float t = (v[e] - mean) * scale;
                t = w[c] * t;
                y[(size_t)row * ldy + c] = t + b[c];
            }
        }
        for (int c = C + lane; c < Cp; c += 64) y[(size_t)row * ldy + c] = 0.f;
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
        y[(size_t)row * ldy + c] = v + b[c];
    }
    for (int c = C + lane; c < Cp; c += 64) y[(size_t)row * ldy + c] = 0.f;
}

hipError_t launch_add_layernorm(hipStream_t s, const float *x, int ldx, const float *res, int ldr, int n, int C, int Cp,
                                const float *w, const float *b, float eps, float *y, int ldy)
{
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, ldx, res, ldr, n, C, Cp, w, b, eps, y, ldy);
    return hipGetLastError();
}

__global__ void add_rowvec_kernel(float *__restrict__ x, int ld, int C, const float *__restrict__ v)
{
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(size_t)n * ld + c] = x[(size_t)n * ld + c] + v[c];
}

hipError_t launch_add_rowvec(hipStream_t s, float *x, int ld, int n, int C, const float *v)
{
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(n), dim3(256), 0, s, x, ld, C, v);
    return hipGetLastError();
}

// VariancePredictor linear_layer (reference src/fs2encoder.cpp:434-435): one wave per token
__global__ __launch_bounds__(256) void rowdot_kernel(const float *__restrict__ x, int ld, int n, int C,
                                                     const float *__restrict__ w, const float *__restrict__ b,
                                                     float *__restrict__ y)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(x[(size_t)row * ld + c], w[c], acc);
    acc = wave_sum_f(acc);
    if (lane == 0) y[row] = acc + b[0];
}

hipError_t launch_rowdot(hipStream_t s, const float *x, int ld, int n, int C, const float *w, const float *b, float *y)
{
    hipLaunchKernelGGL(rowdot_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, ld, n, C, w, b, y);
    return hipGetLastError();
}

// ggml_zv_mul_clamp_to_i32 + get_rows + add (reference src/fs2encoder.cpp:442-474,565-569)
__global__ void bucket_embed_add_kernel(const float *__restrict__ pred, int nbins, const float *__restrict__ emb, int C,
                                        float *__restrict__ x, int ld, int32_t *__restrict__ bucket)
{
    const int n = blockIdx.x;
    const int bin_max = nbins - 1;
    float p = pred[n];
    p = p * (float)bin_max;
    int y = (int)((double)p + 0.5);          // truncating cast of x + 0.5 (double), not round-half-even
    y = y < 0 ? 0 : (y > bin_max ? bin_max : y);
    if (threadIdx.x == 0) bucket[n] = y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[(size_t)n * ld + c] = x[(size_t)n * ld + c] + emb[(size_t)y * C + c];
}

hipError_t launch_bucket_embed_add(hipStream_t s, const float *pred, int n, int nbins, const float *emb, int C, float *x,
                                   int ld, int32_t *bucket)
{
    hipLaunchKernelGGL(bucket_embed_add_kernel, dim3(n), dim3(256), 0, s, pred, nbins, emb, C, x, ld, bucket);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Length regulator on the device (the reference does it on the host, src/fs2encoder.cpp:611-654):
//   dur_i = (int)((float)(exp(logdur_i) - 1.0) + 0.5);  frame f belongs to the token whose cumulative
//   duration first exceeds f; frames past the total (or past T) are zero.
__global__ __launch_bounds__(1024) void lr_scan_kernel(const float *__restrict__ logdur, int n, int T,
                                                       int32_t *__restrict__ cum, int32_t *__restrict__ n_frames)
{
    __shared__ int buf[1024];
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < n; base += 1024)
    {
        const int i = base + tid;
        int d = 0;
        if (i < n)
        {
            const float dur = (float)(exp((double)logdur[i]) - 1.0);
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
        if (i < n) cum[i] = carry + buf[tid];
        carry += buf[1023];
        __syncthreads();
    }
    if (tid == 0) n_frames[0] = carry < T ? carry : T;
}

__global__ void lr_gather_kernel(const float *__restrict__ feat, int ld, const int32_t *__restrict__ cum, int n, int C,
                                 float *__restrict__ hidden, int ldh)
{
    const int f = blockIdx.x;
    // first token i with cum[i] > f
    int lo = 0, hi = n;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > f) hi = mid; else lo = mid + 1;
    }
    const bool live = lo < n;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        hidden[(size_t)f * ldh + c] = live ? feat[(size_t)lo * ld + c] : 0.f;
}

hipError_t launch_length_regulator(hipStream_t s, const float *feat, int ld, const float *logdur, int n, int C, int T,
                                   float *hidden, int ldh, int32_t *n_frames)
{
    // cum[] lives right behind n_frames (caller reserves 1 + n ints)
    int32_t *cum = n_frames + 1;
    hipLaunchKernelGGL(lr_scan_kernel, dim3(1), dim3(1024), 0, s, logdur, n, T, cum, n_frames);
    hipLaunchKernelGGL(lr_gather_kernel, dim3(T), dim3(256), 0, s, feat, ld, cum, n, C, hidden, ldh);
    return hipGetLastError();
}

}  // namespace zv

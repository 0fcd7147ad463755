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

#include <hip/hip_fp16.h>

namespace zv
{

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

static constexpr int CK_MAX = 128;

int conv_pick_ck(int Cin_p)
{
    int nchunks = (Cin_p + CK_MAX - 1) / CK_MAX;
    int ck = round_up((Cin_p + nchunks - 1) / nchunks, 16);
    return ck;
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

template <int MT, int WN>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(const ConvJobs jobs)
{
    constexpr int WM = 4 / WN;
    constexpr int BM = 32 * MT * WM;
    const ConvJob &J = jobs.j[blockIdx.z];

    const int L = J.L;
    const int m0 = blockIdx.x * BM;
    if (m0 >= L) return;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int K = J.K, dil = J.dil, Cin_p = J.Cin_p, Cout_p = J.Cout_p;
    const int nicb = Cin_p >> 4;
    const int ntiles = (Cout_p + 31) >> 5;
    const int nt = blockIdx.y * WN + wn;
    const bool n_ok = nt < ntiles;
    const int rows = BM + (K - 1) * dil;
    const int RS = J.ck * 2 + 16;            // LDS row stride in bytes

    floatx16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    const char *abase = smem + (wm * 32 * MT + (lane & 31)) * RS + (lane >> 5) * 16;

    for (int c0 = 0; c0 < Cin_p; c0 += J.ck)
    {
        const int ck = (Cin_p - c0 < J.ck) ? (Cin_p - c0) : J.ck;
        if (c0) __syncthreads();

        // ---------------- stage: HBM -> prologue -> f16 -> LDS ----------------
        {
            const int cols = ck >> 2;
            int r = tid / cols, c4 = tid - r * cols;
            const int dr = 256 / cols, dc = 256 - dr * cols;
            const int pro = J.pro;
            while (r < rows)
            {
                const int t = m0 - J.pad + r;
                half4 h = {0, 0, 0, 0};
                if (t >= 0 && t < L)
                {
                    const int c = c0 + c4 * 4;
                    const size_t off = (size_t)t * J.ldx + c;
                    if (pro == PRO_RAW_F16)
                    {
                        h = *(const half4 *)((const _Float16 *)J.x0 + off);
                    }
                    else
                    {
                        float4 v = *(const float4 *)((const float *)J.x0 + off);
                        if (pro == PRO_SUM3_ACT)
                        {
                            const float4 b = *(const float4 *)((const float *)J.x1 + off);
                            const float4 d = *(const float4 *)((const float *)J.x2 + off);
                            const float s = J.pscale;
                            v.x = ((v.x + b.x) + d.x) * s;
                            v.y = ((v.y + b.y) + d.y) * s;
                            v.z = ((v.z + b.z) + d.z) * s;
                            v.w = ((v.w + b.w) + d.w) * s;
                        }
                        else if (pro == PRO_NORM_ACT)
                        {
                            const float4 st0 = *(const float4 *)(J.pstat + 2 * c);       // mean,rstd,mean,rstd
                            const float4 st1 = *(const float4 *)(J.pstat + 2 * c + 4);
                            const float4 g = *(const float4 *)(J.pa + c);
                            const float4 b = *(const float4 *)(J.pb + c);
                            v.x = ((v.x - st0.x) * st0.y) * g.x + b.x;
                            v.y = ((v.y - st0.z) * st0.w) * g.y + b.y;
                            v.z = ((v.z - st1.x) * st1.y) * g.z + b.z;
                            v.w = ((v.w - st1.z) * st1.w) * g.w + b.w;
                        }
                        else if (pro == PRO_MELNORM)
                        {
                            const float4 a = *(const float4 *)(J.pa + c);
                            const float4 b = *(const float4 *)(J.pb + c);
                            v.x = (v.x - a.x) / b.x;
                            v.y = (v.y - a.y) / b.y;
                            v.z = (v.z - a.z) / b.z;
                            v.w = (v.w - a.w) / b.w;
                        }
                        if (pro != PRO_MELNORM)
                        {
                            const float s = J.slope;
                            v.x = lrelu(v.x, s);
                            v.y = lrelu(v.y, s);
                            v.z = lrelu(v.z, s);
                            v.w = lrelu(v.w, s);
                        }
                        h[0] = (_Float16)v.x;      // v_cvt_f16_f32: round-to-nearest-even, like _cvtss_sh(x, 0)
                        h[1] = (_Float16)v.y;
                        h[2] = (_Float16)v.z;
                        h[3] = (_Float16)v.w;
                    }
                }
                *(half4 *)(smem + r * RS + c4 * 8) = h;
                r += dr;
                c4 += dc;
                if (c4 >= cols) { c4 -= cols; r++; }
            }
        }
        __syncthreads();

        // ---------------- compute: S = K * nkc MFMA steps over this chunk ----------------
        if (n_ok)
        {
            const int nkc = ck >> 4;
            const int S = K * nkc;
            const half8 *wp = (const half8 *)J.w + ((size_t)nt * K * nicb + (size_t)K * (c0 >> 4)) * 64 + lane;
            half8 bq[4];
#pragma unroll
            for (int u = 0; u < 4; u++) bq[u] = wp[(size_t)((u < S) ? u : S - 1) * 64];
            int tap = 0, kc = 0;
            for (int s0 = 0; s0 < S; s0 += 4)
            {
                half8 bn[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int sn = s0 + 4 + u;
                    bn[u] = wp[(size_t)((sn < S) ? sn : S - 1) * 64];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    if (s0 + u < S)
                    {
                        const char *ap = abase + tap * dil * RS + kc * 32;
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
                        {
                            const half8 a = *(const half8 *)(ap + mt * 32 * RS);
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[u], acc[mt], 0, 0, 0);
                        }
                        if (++kc == nkc) { kc = 0; tap++; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) bq[u] = bn[u];
            }
        }
    }

    // ---------------- epilogue ----------------
    if (!n_ok) return;
    const int oc = nt * 32 + (lane & 31);
    if (oc >= Cout_p) return;
    const float bias = J.bias ? J.bias[oc] : 0.f;
    const float escale = J.escale;
    const int tbase = m0 + wm * 32 * MT + 4 * (lane >> 5);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int t = tbase + mt * 32 + (r & 3) + 8 * (r >> 2);
            if (t < L)
            {
                float v = acc[mt][r] + bias;
                if (J.res) v = v + J.res[(size_t)t * J.ldres + oc];
                v = v * escale;
                if (J.eact) v = lrelu(v, J.oslope);
                if (J.out_f16)
                    ((_Float16 *)J.out)[(size_t)t * J.ldo + oc] = (_Float16)v;
                else
                    ((float *)J.out)[(size_t)t * J.ldo + oc] = v;
            }
        }
}

template <int MT, int WN>
static hipError_t launch_cfg(hipStream_t s, const ConvJobs &jobs, int njobs, int Lmax, int Cout_p, int K, int dil, int ck)
{
    constexpr int WM = 4 / WN;
    constexpr int BM = 32 * MT * WM;
    const int ntiles = (Cout_p + 31) / 32;
    dim3 grid((Lmax + BM - 1) / BM, (ntiles + WN - 1) / WN, njobs);
    const size_t lds = (size_t)(BM + (K - 1) * dil) * (ck * 2 + 16);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = conv1d_mfma_kernel<MT, WN>;
    if (lds > 64 * 1024)
    {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, jobs);
    return hipGetLastError();
}

hipError_t launch_conv(hipStream_t s, const ConvJob *jobs, int njobs, int n_cu)
{
    if (njobs < 1 || njobs > CONV_MAX_JOBS) return hipErrorInvalidValue;
    ConvJobs js;
    int Lmax = 0, Kmax = 0, halo = 0, ck = 0;
    for (int i = 0; i < njobs; i++)
    {
        js.j[i] = jobs[i];
        if (jobs[i].Cout_p != jobs[0].Cout_p) return hipErrorInvalidValue;
        Lmax = jobs[i].L > Lmax ? jobs[i].L : Lmax;
        const int h = (jobs[i].K - 1) * jobs[i].dil;
        if (h > halo) { halo = h; Kmax = jobs[i].K; }
        ck = jobs[i].ck > ck ? jobs[i].ck : ck;
    }
    for (int i = njobs; i < CONV_MAX_JOBS; i++) js.j[i] = jobs[0];
    // launch_cfg sizes LDS from (K-1)*dil: pass the job with the largest halo as (K, dil) = (halo+1, 1)
    (void)Kmax;
    const int Kh = halo + 1;
    const int Cout_p = jobs[0].Cout_p;
    const int ntiles = (Cout_p + 31) / 32;
    const int WN = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
    // pick the tallest wave tile (most B-fragment reuse) that still gives every CU about two workgroups
    auto wgs = [&](int MT) {
        const int BM = 32 * MT * (4 / WN);
        return (long)((Lmax + BM - 1) / BM) * ((ntiles + WN - 1) / WN) * njobs;
    };
    int MT = 4;
    while (MT > 1 && wgs(MT) < 2L * n_cu) MT >>= 1;
#define ZV_CASE(mt, wn) \
    if (MT == mt && WN == wn) return launch_cfg<mt, wn>(s, js, njobs, Lmax, Cout_p, Kh, 1, ck);
    ZV_CASE(4, 4) ZV_CASE(2, 4) ZV_CASE(1, 4)
    ZV_CASE(4, 2) ZV_CASE(2, 2) ZV_CASE(1, 2)
    ZV_CASE(4, 1) ZV_CASE(2, 1) ZV_CASE(1, 1)
#undef ZV_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// vocoder tail: lrelu -> conv (C -> 1, K taps) + bias -> tanh.  Cout = 1 has no GEMM shape: each lane owns
// one output sample and walks its K x C window in LDS (f16 operands, f32 accumulate).

__global__ __launch_bounds__(256) void out_conv_tanh_kernel(const OutConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cp = round_up(a.C, 16);
    const int RS = Cp * 2 + 16;
    const int K = a.K, pad = (K - 1) / 2;
    const int rows = 256 + K - 1;
    const int m0 = blockIdx.x * 256;
    const int tid = threadIdx.x;
    _Float16 *wl = (_Float16 *)(smem + rows * RS);
    for (int i = tid; i < K * Cp; i += 256) wl[i] = ((const _Float16 *)a.w)[i];

    const int cols = Cp >> 2;
    for (int idx = tid; idx < rows * cols; idx += 256)
    {
        const int r = idx / cols, c4 = idx - r * cols;
        const int t = m0 - pad + r;
        half4 h = {0, 0, 0, 0};
        if (t >= 0 && t < a.L)
        {
            const size_t off = (size_t)t * a.ldx + c4 * 4;
            float4 v = *(const float4 *)(a.x0 + off);
            if (a.x1)
            {
                const float4 b = *(const float4 *)(a.x1 + off);
                const float4 d = *(const float4 *)(a.x2 + off);
                v.x = ((v.x + b.x) + d.x) * a.pscale;
                v.y = ((v.y + b.y) + d.y) * a.pscale;
                v.z = ((v.z + b.z) + d.z) * a.pscale;
                v.w = ((v.w + b.w) + d.w) * a.pscale;
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
    if (t >= a.L) return;
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
    a.out[t] = tanhf(acc + a.bias);
}

hipError_t launch_out_conv(hipStream_t s, const OutConvArgs &a)
{
    const int Cp = round_up(a.C, 16);
    const size_t lds = (size_t)(256 + a.K - 1) * (Cp * 2 + 16) + (size_t)a.K * Cp * 2;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(out_conv_tanh_kernel, dim3((a.L + 255) / 256), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace zv

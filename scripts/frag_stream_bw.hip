// Diagnostic (not part of the product): what does ONE CU get out of L2 when its waves stream 1-KiB weight fragments (64 lanes x 16 bytes,
// contiguous), as the single-utterance conv loops do — as a function of the fragments each wave keeps in flight?
//   hipcc --offload-arch=gfx950 -O2 scripts/frag_stream_bw.hip -o /tmp/fsb && /tmp/fsb
// One workgroup of W waves per CU (256 workgroups), every wave walks its own 192-KiB stretch of a 64-MiB buffer REPS times (L2-resident after
// the first pass: 256 CUs x 4 waves x 192 KiB = 192 MiB does not fit, so the stretches are shared 16 ways like the row tiles of a conv share
// their column's weights), D requests in flight per wave (a register ring), consumed by a dependent xor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void stream(const u4 *__restrict__ w, unsigned *out, int frags, int reps, int nstretch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // nstretch > 0: stretches shared chip-wide (the working set, 64 MiB, lives in the memory-side cache); nstretch < 0: every XCD (blockIdx.x & 7)
    // owns -nstretch stretches of its own (L2-resident: the single-utterance convs after their channel groups were dealt over the XCDs)
    const int stretch = nstretch > 0 ? (blockIdx.x * nw + wave) % nstretch : (blockIdx.x & 7) * (-nstretch) + ((blockIdx.x >> 3) * nw + wave) % (-nstretch);
    const u4 *p = w + (size_t)stretch * frags * 64 + lane;
    u4 ring[D];
    u4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++)
    {
#pragma unroll
        for (int u = 0; u < D; u++) ring[u] = p[u * 64];
        for (int f = 0; f + D <= frags; f += D)
        {
#pragma unroll
            for (int u = 0; u < D; u++)
            {
                acc ^= ring[u];
                ring[u] = p[((f + D + u) % frags) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <int D>
static void run(int W, const u4 *w, unsigned *out, bool l2)
{
    const int frags = 192, reps = 20, nstretch = l2 ? -4 : 64 * 1024 * 1024 / (frags * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(stream<D>, dim3(256), dim3(64 * W), 0, 0, w, out, frags, 2, nstretch);
    hipEventRecord(a);
    hipLaunchKernelGGL(stream<D>, dim3(256), dim3(64 * W), 0, 0, w, out, frags, reps, nstretch);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes_cu = (double)W * frags * 1024 * reps;
    printf("%s waves/CU %d  in flight per wave %2d (%3d KiB per CU): %.1f GB/s per CU = %.1f B/clk at 2.1 GHz, %.2f TB/s over 256 CUs, %.0f ns per fragment and wave\n", l2 ? "L2-resident " : "memory-side ", W, D,
           W * D, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu / (ms * 1e-3) / 2.1e9, 256 * bytes_cu / (ms * 1e-3) / 1e12, ms * 1e6 / (frags * reps));
}

int main()
{
    u4 *w;
    unsigned *out;
    hipMalloc(&w, 64u << 20);
    hipMalloc(&out, 64);
    hipMemset(w, 1, 64u << 20);
    for (int l2 = 0; l2 < 2; l2++)
        for (int W : {4, 8})
        {
            run<4>(W, w, out, l2);
            run<8>(W, w, out, l2);
            run<16>(W, w, out, l2);
            run<32>(W, w, out, l2);
            run<48>(W, w, out, l2);
        }
    return 0;
}

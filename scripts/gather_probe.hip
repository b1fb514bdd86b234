// What does a scattered small read cost in HBM bytes on gfx950, and what does rocprofv3's FETCH_SIZE say about it? (diagnostic)
// Random aligned chunks of 16, 64 or 128 bytes (1, 4 or 8 neighbouring lanes x 16 B) are gathered from a table far
// larger than the 256 MiB Infinity Cache.  Printed: chunks per second and the bandwidth that rate would mean if a chunk
// cost 64 B or 128 B of HBM traffic.  If a 128-byte chunk is gathered at about the rate of a 64-byte one, a miss
// fetches a whole 128-byte line whatever it needs; if at about half the rate, memory is read in 64-byte sectors.
// Under `rocprofv3 --pmc FETCH_SIZE` (one kernel per chunk size, named by it) FETCH_SIZE / chunks gives the counter's
// bytes per chunk to set beside that.
// Build: hipcc --offload-arch=gfx950 -O2 -o gather_probe gather_probe.hip      usage: gather_probe [table GiB = 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long mix(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// W lanes share one chunk of W * 16 bytes; every group gathers `iters` chunks, eight requests in flight.
template <int W> __global__ __launch_bounds__(256) void gather(const uint4 *table, unsigned long long chunks_in_table, int iters, unsigned *sink)
{
    const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long group = gid / W;
    const unsigned lane = (unsigned)(gid % W);
    unsigned acc = 0;
    for (int it = 0; it < iters; it += 8) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned long long c = mix(group * 0x9e3779b97f4a7c15ull + (unsigned long long)(it + k)) % chunks_in_table;
            v[k] = table[c * W + lane];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int W> void run(const uint4 *table, size_t bytes, unsigned *sink, const char *name)
{
    const int blocks = 256 * 16, threads = 256, iters = 256;
    const unsigned long long chunks_in_table = bytes / (16ull * W);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(gather<W>, dim3(blocks), dim3(threads), 0, 0, table, chunks_in_table, 8, sink);      // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(gather<W>, dim3(blocks), dim3(threads), 0, 0, table, chunks_in_table, iters, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double chunks = (double)blocks * threads / W * iters;
    const double rate = chunks / (ms * 1e-3);
    std::printf("%-10s %d-byte chunks: %.3e chunks in %.3f ms = %.3e chunks/s; at 64 B each %.2f TB/s, at 128 B each %.2f TB/s, "
                "useful bytes %.2f TB/s\n", name, 16 * W, chunks, ms, rate, rate * 64 / 1e12, rate * 128 / 1e12, rate * 16 * W / 1e12);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)std::atoi(argv[1]) : 8;
    const size_t bytes = gib << 30;
    uint4 *table = nullptr;
    unsigned *sink = nullptr;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { std::printf("allocation failed\n"); return 1; }
    (void)hipMemset(table, 1, bytes);
    (void)hipDeviceSynchronize();
    run<1>(table, bytes, sink, "gather16");
    run<4>(table, bytes, sink, "gather64");
    run<8>(table, bytes, sink, "gather128");
    run<16>(table, bytes, sink, "gather256");
    (void)hipFree(table); (void)hipFree(sink);
    return 0;
}

// How fast can ONE compute unit stream from HBM on gfx950, and does it matter which CUs stream together? (diagnostic)
// Workgroups of 768 threads each read a private region (far beyond the caches in total) with K 16-byte loads per
// thread in flight; only the workgroups picked by `mode` stream, the others leave at once:
//   one: workgroup 0 alone;  xcd: 16 workgroups with blockIdx % 8 == 0 (one XCD as dispatch deals them);
//   spread: workgroups 0..15 (two per XCD);  half: blockIdx % 8 < 4, all of them;  all: every workgroup.
// Printed: GB/s per streaming workgroup and in total.
// Build: hipcc --offload-arch=gfx950 -O2 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

template <int K> __global__ __launch_bounds__(768) void stream(const uint4 *base, size_t region16, int reps, int mode, unsigned *sink)
{
    const int b = blockIdx.x;
    bool on = true;
    if (mode == 0) on = (b == 0);
    else if (mode == 1) on = (b % 8 == 0) && (b / 8 < 16);
    else if (mode == 2) on = (b < 16);
    else if (mode == 3) on = (b % 8 < 4);
    if (!on) return;
    const uint4 *p = base + (size_t)b * region16;
    unsigned acc = 0;
    for (int r = 0; r < reps; r++) {
        for (size_t i = threadIdx.x; i + (size_t)(K - 1) * 768 < region16; i += (size_t)K * 768) {
            uint4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = p[i + (size_t)k * 768];
#pragma unroll
            for (int k = 0; k < K; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int K> void run(const uint4 *base, size_t region16, int mode, const char *mname, unsigned *sink)
{
    const int reps = 1;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(stream<K>, dim3(256), dim3(768), 0, 0, base, region16, reps, mode, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(stream<K>, dim3(256), dim3(768), 0, 0, base, region16, reps, mode, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const int nwg = mode == 0 ? 1 : (mode == 1 || mode == 2) ? 16 : mode == 3 ? 128 : 256;
    const double bytes = (double)nwg * region16 * 16.0 * reps;
    std::printf("%-7s K=%2d: %3d workgroups, %.3f ms, %.1f GB/s each, %.2f TB/s in total\n", mname, K, nwg, ms,
                bytes / nwg / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

int main()
{
    const size_t region = (size_t)96 << 20;             // 96 MiB per workgroup, 24 GiB in all
    const size_t region16 = region / 16;
    uint4 *base = nullptr; unsigned *sink = nullptr;
    if (hipMalloc(&base, region * 256) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { std::printf("allocation failed\n"); return 1; }
    (void)hipMemset(base, 1, region * 256);
    (void)hipDeviceSynchronize();
    const char *names[5] = {"one", "xcd", "spread", "half", "all"};
    for (int mode = 0; mode < 5; mode++) {
        run<2>(base, region16, mode, names[mode], sink);
        run<6>(base, region16, mode, names[mode], sink);
        run<12>(base, region16, mode, names[mode], sink);
    }
    (void)hipFree(base); (void)hipFree(sink);
    return 0;
}

// Do the waves of a SIMD overlap on code shaped like the node passes of the step kernel? (diagnostic)
// One workgroup of W waves; every wave runs REPS times a pass over 14 register-held values with the arithmetic of
// pass 2 (dependent fp64 chain per value, a uniform branch between values).  Prints s_memtime ticks per pass for the
// first, a middle and the last wave.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pass_probe pass_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REPS 2000
template <int MODE> __global__ void k(double *out, unsigned long long *cyc, float seed, int chunk, double avg)
{
    float d[16];
#pragma unroll
    for (int j = 0; j < 16; j++) d[j] = seed + threadIdx.x * 0.001f + j;
    double bin1 = 0.0, bin2 = 0.0;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int r = 0; r < REPS; r++) {
        asm volatile("" : "+v"(avg));
        int c = chunk; asm volatile("" : "+v"(c)); c = __builtin_amdgcn_readfirstlane(c);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MODE != 1) { if (j >= c) break; }
            const double dev = (double)d[j] - avg;
            const double t = dev * dev * 0.01;
            const double up = t + 0x1.8p36;
            const double hi = up - 0x1.8p36;
            bin1 += hi;
            bin2 += floor((t - hi) * 0x1p64);
        }
        if (MODE == 2) __syncthreads();
    }
    unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    out[threadIdx.x] = bin1 + bin2;
}

template <int MODE> void run(const char *name)
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int waves : {1, 4, 8, 12, 16}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.5f, MODE == 1 ? 16 : 14, 3.25);
        (void)hipDeviceSynchronize();
        unsigned long long c[16]; (void)hipMemcpy(c, cyc, 16 * 8, hipMemcpyDeviceToHost);
        std::printf("%-34s waves=%2d  ticks per pass: first %.0f  middle %.0f  last %.0f\n", name, waves,
                    (double)c[0] / REPS, (double)c[waves / 2] / REPS, (double)c[waves - 1] / REPS);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("14 values, uniform break per value");
    run<1>("16 values, straight line");
    run<2>("14 values + barrier per pass");
    return 0;
}

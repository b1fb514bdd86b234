// Does straight-line code larger than the instruction cache slow a CU's waves down? (diagnostic)
// One workgroup of W waves runs a loop whose body is K independent-enough v_fma_f32 (8 bytes each), for bodies of
// 8 KB .. 256 KB; ticks per wave-instruction per SIMD follow from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O2 -o icache_probe icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K> __global__ void k(float *out, unsigned long long *cyc, float seed, int outer)
{
    float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int o = 0; o < outer; o++) {
#pragma unroll
        for (int r = 0; r < K / 4; r++) {
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f0) : "v"(f1)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f2) : "v"(f3));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f1) : "v"(f0)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f3) : "v"(f2));
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = f0 + f1 + f2 + f3;
}

template <int K> void run()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&cyc, 8);
    const int outer = (1 << 20) / K;
    for (int waves : {4, 8, 12, 16}) {
        hipLaunchKernelGGL(k<K>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.5f, outer);
        (void)hipDeviceSynchronize();
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        std::printf("body %4d KB  waves=%2d  %.2f ticks per wave-instruction per SIMD\n", K * 8 / 1024, waves,
                    (double)c / ((double)K * outer * (waves / 4)));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() { run<1024>(); run<4096>(); run<8192>(); run<16384>(); run<32768>(); return 0; }

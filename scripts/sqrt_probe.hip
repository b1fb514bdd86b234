// Is a correctly rounded fp32 square root to be had from the reciprocal-root approximation and fused multiply-adds alone? (diagnostic)
// For EVERY float in [2^-20, 2^24) (the window the node passes vouch for) candidates are compared with (float)sqrt((double)x),
// which is the correctly rounded root (a double carries more than 2 * 24 + 2 bits):
//   A: y = v_rsq_f32(x); g = x y; h = y / 2; g += (x - g g) h           (one correction)
//   B: A, then once more g += (x - g g) h                               (two corrections)
//   C: the shipped form: v_sqrt_f32(x) moved to a neighbour when one of two exact residuals asks for it
// Printed: the number of inputs on which each differs from the reference, and the first such input.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o sqrt_probe sqrt_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__device__ __forceinline__ float cand_a(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    return g;
}
__device__ __forceinline__ float cand_b(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    const float h = 0.5f * y;
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    return g;
}
__device__ __forceinline__ float cand_c(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}

__global__ void check(unsigned lo, unsigned hi, unsigned long long *bad, unsigned *first)
{
    const unsigned long long n = (unsigned long long)hi - lo;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = lo + (unsigned)i;
        const float x = __uint_as_float(bits);
        const float ref = (float)sqrt((double)x);
        const float a = cand_a(x), b = cand_b(x), c = cand_c(x);
        if (a != ref) { atomicAdd(&bad[0], 1ull); atomicMin(&first[0], bits); }
        if (b != ref) { atomicAdd(&bad[1], 1ull); atomicMin(&first[1], bits); }
        if (c != ref) { atomicAdd(&bad[2], 1ull); atomicMin(&first[2], bits); }
    }
}

int main()
{
    unsigned long long *bad; unsigned *first;
    (void)hipMalloc(&bad, 3 * sizeof(unsigned long long)); (void)hipMalloc(&first, 3 * sizeof(unsigned));
    (void)hipMemset(bad, 0, 3 * sizeof(unsigned long long)); (void)hipMemset(first, 0xff, 3 * sizeof(unsigned));
    const float flo = 0x1p-20f, fhi = 0x1p24f;
    unsigned lo, hi;
    __builtin_memcpy(&lo, &flo, 4); __builtin_memcpy(&hi, &fhi, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi, bad, first);
    (void)hipDeviceSynchronize();
    unsigned long long hb[3]; unsigned hf[3];
    (void)hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, sizeof hf, hipMemcpyDeviceToHost);
    std::printf("%llu inputs in [2^-20, 2^24)\n", (unsigned long long)hi - lo);
    const char *names[3] = {"A rsq + one correction", "B rsq + two corrections", "C sqrt + neighbour test (shipped)"};
    for (int k = 0; k < 3; k++) std::printf("%-36s wrong on %llu inputs, first 0x%08x\n", names[k], hb[k], hf[k]);
    return 0;
}

// Round-trip time of a flag hand-off between two workgroups on different CUs (diagnostic).
// usage: pingpong  -> prints ns per round trip for pairs of workgroups (0, other) and three access kinds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p)
{
    if (SCOPE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SCOPE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int SCOPE>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v)
{
    if (SCOPE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SCOPE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else asm volatile("global_store_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// blocks a and b bounce a counter `rounds` times; everyone else exits at once
template <int SCOPE>
__global__ void pingpong(unsigned long long *flags, int a, int b, int rounds, unsigned long long *out, unsigned *xcc)
{
    if (threadIdx.x != 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if ((int)blockIdx.x == a) xcc[0] = id;
    if ((int)blockIdx.x == b) xcc[1] = id;
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    const bool first = (int)blockIdx.x == a;
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        if (first) {
            st<SCOPE>(&flags[0], (unsigned long long)r);
            unsigned long long spins = 0;
            while (ld<SCOPE>(&flags[16]) != (unsigned long long)r) { if (++spins > 2000000ull) { out[1] = 1; return; } }
        } else {
            unsigned long long spins = 0;
            while (ld<SCOPE>(&flags[0]) != (unsigned long long)r) { if (++spins > 2000000ull) { out[1] = 1; return; } }
            st<SCOPE>(&flags[16], (unsigned long long)r);
        }
    }
    if (first) out[0] = wall_clock64() - t0;
}

int main()
{
    unsigned long long *flags, *out;
    unsigned *xcc;
    CHECK(hipMalloc(&flags, 4096)); CHECK(hipMalloc(&out, 16)); CHECK(hipMalloc(&xcc, 8));
    const int rounds = 2000;
    const int others[] = {8, 1, 16, 3, 64, 129};
    for (int scope = 0; scope < 3; scope++)
        for (int other : others) {
            CHECK(hipMemset(flags, 0, 4096)); CHECK(hipMemset(out, 0, 16));
            if (scope == 0) hipLaunchKernelGGL(pingpong<0>, dim3(256), dim3(64), 0, 0, flags, 0, other, rounds, out, xcc);
            if (scope == 1) hipLaunchKernelGGL(pingpong<1>, dim3(256), dim3(64), 0, 0, flags, 0, other, rounds, out, xcc);
            if (scope == 2) hipLaunchKernelGGL(pingpong<2>, dim3(256), dim3(64), 0, 0, flags, 0, other, rounds, out, xcc);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2]; unsigned x[2];
            CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
            printf("%s blocks 0,%d (xcc %u,%u): %s %.0f ns per round trip\n",
                   scope == 0 ? "agent-scope atomics" : scope == 1 ? "workgroup-scope atomics" : "sc0 load/store",
                   other, x[0], x[1], h[1] ? "TIMED OUT" : "ok", h[1] ? 0.0 : (double)h[0] * 10.0 / rounds);
        }
    return 0;
}

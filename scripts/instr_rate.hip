// Issue cost of a few gfx950 vector instructions (diagnostic): one workgroup of W waves on one CU
// runs a long stream of one instruction; s_memtime ticks per wave-instruction per SIMD follow from the time at
// which the LAST wave is done (the oldest wave of a SIMD runs at single-wave speed whatever the others do).  Build: hipcc --offload-arch=gfx950 -O2 -o instr_rate instr_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define OUTER 200

template <int OP> __global__ void k(double *out, unsigned long long *cyc, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int o = 0; o < OUTER; o++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (OP == 0) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a2) : "v"(a3));
                           asm volatile("v_add_f64 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(a3) : "v"(a2)); }
            if (OP == 1) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a2) : "v"(a3));
                           asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a3) : "v"(a2)); }
            if (OP == 2) { asm volatile("v_trunc_f64 %0, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_trunc_f64 %0, %1" : "=v"(a2) : "v"(a3));
                           asm volatile("v_trunc_f64 %0, %1" : "=v"(a1) : "v"(a0)); asm volatile("v_trunc_f64 %0, %1" : "=v"(a3) : "v"(a2)); }
            if (OP == 3) { asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u0) : "v"(a1)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u1) : "v"(a3));
                           asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u2) : "v"(a0)); asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u3) : "v"(a2)); }
            if (OP == 4) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a0) : "v"(u1)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a2) : "v"(u3));
                           asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a1) : "v"(u0)); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a3) : "v"(u2)); }
            if (OP == 5) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a0) : "v"(f1)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a2) : "v"(f3));
                           asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a1) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a3) : "v"(f2)); }
            if (OP == 6) { asm volatile("v_sqrt_f32 %0, %1" : "=v"(f0) : "v"(f1)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f2) : "v"(f3));
                           asm volatile("v_sqrt_f32 %0, %1" : "=v"(f1) : "v"(f0)); asm volatile("v_sqrt_f32 %0, %1" : "=v"(f3) : "v"(f2)); }
            if (OP == 7) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f2) : "v"(f3));
                           asm volatile("v_add_f32 %0, %0, %1" : "+v"(f1) : "v"(f0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f3) : "v"(f2)); }
            if (OP == 8) { asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q0) : "v"(u1)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q2) : "v"(u3));
                           asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q1) : "v"(u0)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q3) : "v"(u2)); }
            if (OP == 9) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q0) : "v"(q1)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q2) : "v"(q3));
                           asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q1) : "v"(q0)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q3) : "v"(q2)); }
            if (OP == 10) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a2) : "v"(a3));
                            asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a3) : "v"(a2)); }
            if (OP == 11) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));
                            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(a2)); }
            if (OP == 12) { asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a2), "v"(a3) : "vcc");
                            asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a1), "v"(a0) : "vcc"); asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a3), "v"(a2) : "vcc"); }
            if (OP == 13) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u0) : "v"(u1)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u2) : "v"(u3));
                            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u1) : "v"(u0)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u3) : "v"(u2)); }
            if (OP == 15) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a0) : "v"(a1)); asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a2) : "v"(a3));
                            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a3) : "v"(a2)); }
            if (OP == 16) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f0) : "v"(f1)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f2) : "v"(f3));
                            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f1) : "v"(f0)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f3) : "v"(f2)); }
            if (OP == 17) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));
                            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a1) : "v"(a0)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a3) : "v"(a2)); }
            if (OP == 18) { asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2)); asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u2) : "v"(u3), "v"(u0));
                            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u1) : "v"(u0), "v"(u3)); asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u3) : "v"(u2), "v"(u1)); }
            if (OP == 19) { asm volatile("v_ldexp_f64 %0, %0, 32" : "+v"(a0)); asm volatile("v_ldexp_f64 %0, %0, 32" : "+v"(a2));
                            asm volatile("v_ldexp_f64 %0, %0, 32" : "+v"(a1)); asm volatile("v_ldexp_f64 %0, %0, 32" : "+v"(a3)); }
            if (OP == 20) { asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q0), "v"(q1) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q2), "v"(q3) : "vcc");
                            asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q1), "v"(q0) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q3), "v"(q2) : "vcc"); }
            if (OP == 21) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u0) : "v"(u1) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u2) : "v"(u3) : "vcc");
                            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u1) : "v"(u0) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u3) : "v"(u2) : "vcc"); }
            if (OP == 22) { asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(f0), "v"(f1) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f2) : "v"(f3) : "vcc");
                            asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(f1), "v"(f0) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f3) : "v"(f2) : "vcc"); }
            if (OP == 23) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(a2), "v"(a3), "v"(a0));
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a2) : "v"(a3), "v"(a0), "v"(a1)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a3) : "v"(a0), "v"(a1), "v"(a2)); }
            if (OP == 24) { asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3)); asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a1) : "v"(a2), "v"(a3), "v"(a0));
                            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a2) : "v"(a3), "v"(a0), "v"(a1)); asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a3) : "v"(a0), "v"(a1), "v"(a2)); }
            if (OP == 25) { asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(u0) : "s20"); asm volatile("v_readlane_b32 s21, %0, 5" :: "v"(u1) : "s21");
                            asm volatile("v_readlane_b32 s22, %0, 7" :: "v"(u2) : "s22"); asm volatile("v_readlane_b32 s23, %0, 9" :: "v"(u3) : "s23"); }
            if (OP == 26) { asm volatile("v_pk_mul_f32 %0, %0, %2\n v_add_f64 %1, %1, %3" : "+v"(a0), "+v"(a2) : "v"(a1), "v"(a3)); asm volatile("v_sqrt_f32 %0, %2\n v_cvt_f64_f32 %1, %3" : "=v"(f0), "=v"(a1) : "v"(f1), "v"(f2)); }
            if (OP == 27) { asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u2) : "v"(u3));
                            asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u1) : "v"(u0)); asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u3) : "v"(u2)); }
            if (OP == 28) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u2) : "v"(u3));
                            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u1) : "v"(u0)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u3) : "v"(u2)); }
            if (OP == 29) { asm volatile("v_xor_b32 %0, %2, %1\n v_pk_min_u16 %3, %3, %0" : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3));
                            asm volatile("v_xor_b32 %0, %2, %1\n v_pk_min_u16 %3, %3, %0" : "+v"(u0), "+v"(u1) : "v"(u2), "v"(u3)); }
            if (OP == 30) { asm volatile("v_xor_b32 %0, s20, %1\n v_pk_min_u16 %2, %2, %0" : "+v"(u0), "+v"(u1), "+v"(u3) :: "s20");
                            asm volatile("v_xor_b32 %0, s20, %1\n v_pk_min_u16 %2, %2, %0" : "+v"(u0), "+v"(u2), "+v"(u3) :: "s20"); }
            if (OP == 14) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u2) : "v"(u3));
                            asm volatile("v_min_u32 %0, %0, %1" : "+v"(u1) : "v"(u0)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u3) : "v"(u2)); }
        }
    }
    unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;      // every wave: the last one to finish is what counts
    out[threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3 + (double)(q0 + q1 + q2 + q3);
}

template <int OP> void run(const char *name)
{
    double *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 16 * 8);
    for (int waves : {4, 8, 12, 16}) {
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.5);
        hipDeviceSynchronize();
        unsigned long long c[16]; (void)hipMemcpy(c, cyc, 16 * 8, hipMemcpyDeviceToHost);
        unsigned long long first = c[0], last = 0;
        for (int w = 0; w < waves; w++) last = c[w] > last ? c[w] : last;
        std::printf("%-20s waves=%2d  first wave %.2f ticks per instruction, all waves done %.2f ticks per wave-instruction per SIMD\n", name, waves,
                    (double)first / ((double)REP * OUTER), (double)last / ((double)REP * OUTER * ((waves + 3) / 4)));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<7>("v_add_f32"); run<14>("v_min_u32"); run<11>("v_pk_mul_f32"); run<0>("v_add_f64"); run<1>("v_mul_f64"); run<10>("v_fma_f64");
    run<12>("v_cmp_gt_f64"); run<2>("v_trunc_f64"); run<3>("v_cvt_u32_f64"); run<4>("v_cvt_f64_u32"); run<5>("v_cvt_f64_f32");
    run<15>("v_pk_fma_f32"); run<17>("v_pk_add_f32"); run<16>("v_fma_f32"); run<18>("v_min3_u32");
    run<19>("v_ldexp_f64"); run<20>("v_cmp_lt_u64"); run<21>("v_add(c)_co_u32"); run<22>("v_cmp+cndmask"); run<23>("v_pk_fma 3src"); run<24>("v_fma_f64 3src");
    run<27>("v_pk_min_u16"); run<28>("v_xor_b32"); run<29>("xor+pk_min chain"); run<30>("xor(sgpr)+pk_min");
    run<25>("v_readlane"); run<26>("mix pk/f64/sqrt/cvt");
    run<6>("v_sqrt_f32"); run<8>("v_lshlrev_b64"); run<9>("v_lshl_add_u64"); run<13>("v_mov_b32_dpp");
    return 0;
}

// Device-side building blocks shared by the graph and voxel engines:
// exact (order-independent) fixed-point sums, Philox4x32-10, wave/block helpers.
// gfx950 only: wavefront = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_WAVE 64

// ---------------------------------------------------------------------------
// fx128: signed 128-bit fixed point, unit 2^-64.  Every cost sum is an integer
// sum of terms truncated (toward zero) to this unit, so the result does not
// depend on the order or grouping of the additions, and removing a term is the
// exact inverse of adding it.
// ---------------------------------------------------------------------------
struct fx128 {
    unsigned long long lo;
    long long hi;
};

__device__ __forceinline__ fx128 fx_zero() { return fx128{0ull, 0ll}; }

__device__ __forceinline__ void fx_add(fx128 &a, const fx128 &b)
{
    unsigned long long l = a.lo + b.lo;
    a.hi += b.hi + (long long)(l < b.lo);
    a.lo = l;
}

__device__ __forceinline__ fx128 fx_neg(fx128 a)
{
    fx128 r;
    r.lo = ~a.lo + 1ull;
    r.hi = (long long)(~(unsigned long long)a.hi + (unsigned long long)(r.lo == 0ull));
    return r;
}

// trunc(|t| * 2^64) with the sign of t.  |t| must be < 2^62; otherwise (NaN and
// infinities included) *err is set and the term counts as zero.
// Below 2^32 (every term of a sane cost) the integer part and the fraction are split in fp64, both
// exactly: trunc(|t|) fits 32 bits, |t| - trunc(|t|) is exact, and so is its product with 2^64.
__device__ __forceinline__ fx128 fx_from_double(double t, unsigned &err)
{
    const double a = fabs(t);
    fx128 r;
    if (__builtin_expect(a < 4294967296.0, 1)) {
        const double ti = trunc(a);
        const double s = (a - ti) * 0x1p64;                         // < 2^64, exact
        const unsigned h32 = __double2uint_rz(s * 0x1p-32);
        const unsigned l32 = __double2uint_rz(fma((double)h32, -0x1p32, s));
        r.lo = ((unsigned long long)h32 << 32) | (unsigned long long)l32;
        r.hi = (long long)(unsigned long long)__double2uint_rz(ti);
    } else {
        unsigned long long bits = (unsigned long long)__double_as_longlong(t);
        int e = (int)((bits >> 52) & 0x7ffu);
        unsigned long long m = (bits & ((1ull << 52) - 1ull)) | (1ull << 52);
        r = fx_zero();
        if (e >= 1023 + 62) { err = 1u; return r; }
        int sh = e - 1011;              // value * 2^64 = m * 2^sh, sh in [44, 73]
        if (sh >= 64) r.hi = (long long)(m << (sh - 64));
        else { r.lo = m << sh; r.hi = (long long)(m >> (64 - sh)); }
    }
    if (__double_as_longlong(t) < 0) r = fx_neg(r);
    return r;
}

// acc += trunc(t * 2^64) for 0 <= t < 2^32, without a branch (the caller vouches for the range).
__device__ __forceinline__ void fx_add_small_nonneg(fx128 &acc, double t)
{
    const double ti = trunc(t);
    const double p = (t - ti) * 0x1p32;                             // fraction * 2^32, exact
    const unsigned h32 = __double2uint_rz(p);
    const unsigned l32 = __double2uint_rz((p - (double)h32) * 0x1p32);
    const unsigned long long lo = ((unsigned long long)h32 << 32) | (unsigned long long)l32;
    const unsigned long long l = acc.lo + lo;
    acc.hi += (long long)(unsigned long long)__double2uint_rz(ti) + (long long)(l < lo);
    acc.lo = l;
}

// trunc(t * 2^64) for 0 <= t < 2^32, without a branch (the caller vouches for the range)
__device__ __forceinline__ fx128 fx_from_small_nonneg(double t)
{
    fx128 r = fx_zero();
    fx_add_small_nonneg(r, t);
    return r;
}

// sqrtf for a normal x well inside the exponent range, correctly rounded, from the reciprocal-root approximation y:
// g = x y is the root to about two units in the last place, x - g g is its exact residual (one fused multiply-add) and
// g + (x - g g) y / 2 rounds to the correctly rounded root.  That it does so for EVERY float in [2^-20, 2^24), the only
// inputs the callers pass, is not taken from an error analysis: pt_debug_sqrt_check compares all 369 098 752 of them
// with (float)sqrt((double)x) on the device (the parity suite runs it).  One quarter-rate instruction and four
// full-rate ones (two nodes' worth fit the packed forms), against one and thirteen for moving v_sqrt_f32 to a
// neighbour by two residual tests.
__device__ __forceinline__ float sqrt_rn_midrange(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// the same for two numbers side by side (each half is the scalar sequence above: same bits)
typedef float pt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pt_f2 sqrt_rn_midrange2(pt_f2 x)
{
    const pt_f2 y = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
    const pt_f2 g = x * y, h = y * 0.5f;
    const pt_f2 d = __builtin_elementwise_fma(-g, g, x);
    return __builtin_elementwise_fma(d, h, g);
}

// exp as a real call.  Inlined into a long step loop, the eleven polynomial coefficients of the library's exp
// become loop-invariant register pairs, get spilled under register pressure, and every evaluation then waits for
// nine scratch loads one after the other; behind a call they are immediates again.
__device__ __attribute__((noinline)) double exp_outlined(double x) { return exp(x); }

// nearest double (ties to even) of a * 2^-64
__device__ __forceinline__ double fx_to_double(fx128 a)
{
    bool neg = a.hi < 0;
    if (neg) a = fx_neg(a);
    unsigned long long hi = (unsigned long long)a.hi, lo = a.lo;
    double r;
    if (__builtin_expect(hi < (1ull << 21), 1)) {
        // below 2^21 (every sum of a sane cost): the integer part and the upper half of the fraction fill at most 53
        // bits, so their sum is exact, and adding the lower half rounds the exact value once, to nearest even
        const double top = (double)(unsigned)hi + (double)(unsigned)(lo >> 32) * 0x1p-32;
        r = top + (double)(unsigned)lo * 0x1p-64;
    } else {
        int lz = __clzll((long long)hi);
        unsigned long long top = lz ? ((hi << lz) | (lo >> (64 - lz))) : hi;
        unsigned long long rest = lz ? (lo << lz) : lo;
        top |= (unsigned long long)(rest != 0ull);      // sticky bit, far below the rounding position
        r = ldexp((double)top, -lz);
    }
    return neg ? -r : r;
}

// Sum of a 32-bit value over the wavefront by data-parallel-primitive adds (no LDS traffic):
// row shifts 1, 2, 3, then 4 and 8 build the inclusive prefix sum of every row of 16 lanes,
// two row broadcasts carry the row totals on; lane 63 ends with the total of all 64 lanes.
__device__ __forceinline__ unsigned wave_sum_u32_lane63(unsigned v)
{
    unsigned t = v;
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);      // row_shr:3
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1, 3
    t += (unsigned)__builtin_amdgcn_update_dpp(0, (int)t, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2, 3
    return t;
}

// Sum of v over the wavefront, valid in lane 63 only.  The 128 bits are cut into five limbs of
// 26 bits; a limb summed over 64 lanes stays below 2^32, so the limbs add without carries and
// are put together again (modulo 2^128: two's complement negatives come out right).
__device__ __forceinline__ fx128 fx_wave_reduce_lane63(fx128 v)
{
    const unsigned long long hi = (unsigned long long)v.hi;
    const unsigned m = (1u << 26) - 1u;
    unsigned l0 = (unsigned)v.lo & m;
    unsigned l1 = (unsigned)(v.lo >> 26) & m;
    unsigned l2 = (unsigned)((v.lo >> 52) | (hi << 12)) & m;
    unsigned l3 = (unsigned)(hi >> 14) & m;
    unsigned l4 = (unsigned)(hi >> 40);
    l0 = wave_sum_u32_lane63(l0); l1 = wave_sum_u32_lane63(l1); l2 = wave_sum_u32_lane63(l2);
    l3 = wave_sum_u32_lane63(l3); l4 = wave_sum_u32_lane63(l4);
    fx128 r;
    r.lo = (unsigned long long)l0; r.hi = 0ll;
    fx128 t;
    t.lo = (unsigned long long)l1 << 26; t.hi = 0ll; fx_add(r, t);
    t.lo = (unsigned long long)l2 << 52; t.hi = (long long)((unsigned long long)l2 >> 12); fx_add(r, t);
    t.lo = 0ull; t.hi = (long long)((unsigned long long)l3 << 14); fx_add(r, t);
    t.lo = 0ull; t.hi = (long long)((unsigned long long)l4 << 40); fx_add(r, t);
    return r;
}

// Add v into a 128-bit accumulator in LDS with two 64-bit atomics; the carry of
// each low-word add is derived from the value that add returned, so any
// interleaving of adders gives the exact total.
__device__ __forceinline__ void fx_atomic_add_lds(fx128 *acc, const fx128 &v)
{
    unsigned long long old = atomicAdd(&acc->lo, v.lo);
    unsigned long long carry = ((old + v.lo) < old) ? 1ull : 0ull;
    atomicAdd((unsigned long long *)&acc->hi, (unsigned long long)v.hi + carry);
}

// every thread calls; the last lane of each wave adds its wave's total into *acc
__device__ __forceinline__ void fx_block_accumulate(fx128 *acc, fx128 v)
{
    if (__ballot((v.lo | (unsigned long long)v.hi) != 0ull) == 0ull) return;    // nothing in this wave
    v = fx_wave_reduce_lane63(v);
    if ((threadIdx.x & (PT_WAVE - 1)) == PT_WAVE - 1) fx_atomic_add_lds(acc, v);
}

// ---------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi)
{
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// the p-th double of stream (stream_id, domain)
__device__ __forceinline__ double philox_double(uint32_t k0, uint32_t k1, uint32_t stream_id, uint32_t domain,
                                                unsigned long long p)
{
    unsigned long long q = p >> 1;
    uint32_t w[4];
    philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream_id, domain, k0, k1, w);
    return (p & 1ull) ? u53(w[2], w[3]) : u53(w[0], w[1]);
}

// ---------------------------------------------------------------------------
// block-wide exclusive scan of an int array in place (any address space),
// len arbitrary.  tmp: LDS scratch of >= 17 ints.  All threads must call.
// Returns the total.
// ---------------------------------------------------------------------------
__device__ inline int block_exclusive_scan(int *arr, int len, int *tmp)
{
    const int B = blockDim.x, tid = threadIdx.x;
    const int chunk = (len + B - 1) / B;
    const int beg = min(tid * chunk, len), end = min(beg + chunk, len);
    int s = 0;
    for (int i = beg; i < end; i++) s += arr[i];
    // inclusive wave scan of s
    int lane = tid & (PT_WAVE - 1), wv = tid / PT_WAVE, nw = (B + PT_WAVE - 1) / PT_WAVE;
    int inc = s;
#pragma unroll
    for (int off = 1; off < PT_WAVE; off <<= 1) {
        int o = __shfl_up(inc, off, PT_WAVE);
        if (lane >= off) inc += o;
    }
    if (lane == PT_WAVE - 1) tmp[wv] = inc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < nw; w++) { int t = tmp[w]; tmp[w] = run; run += t; }
        tmp[16] = run;
    }
    __syncthreads();
    int base = tmp[wv] + inc - s;
    int total = tmp[16];
    for (int i = beg; i < end; i++) { int t = arr[i]; arr[i] = base; base += t; }
    __syncthreads();
    return total;
}

// same over an unsigned short array (values and total must stay below 65536)
__device__ inline int block_exclusive_scan_u16(unsigned short *arr, int len, int *tmp)
{
    const int B = blockDim.x, tid = threadIdx.x;
    const int chunk = (len + B - 1) / B;
    const int beg = min(tid * chunk, len), end = min(beg + chunk, len);
    int s = 0;
    for (int i = beg; i < end; i++) s += arr[i];
    int lane = tid & (PT_WAVE - 1), wv = tid / PT_WAVE, nw = (B + PT_WAVE - 1) / PT_WAVE;
    int inc = s;
#pragma unroll
    for (int off = 1; off < PT_WAVE; off <<= 1) {
        int o = __shfl_up(inc, off, PT_WAVE);
        if (lane >= off) inc += o;
    }
    if (lane == PT_WAVE - 1) tmp[wv] = inc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < nw; w++) { int t = tmp[w]; tmp[w] = run; run += t; }
        tmp[16] = run;
    }
    __syncthreads();
    int base = tmp[wv] + inc - s;
    int total = tmp[16];
    for (int i = beg; i < end; i++) { int t = arr[i]; arr[i] = (unsigned short)base; base += t; }
    __syncthreads();
    return total;
}

// Error slot, SHA-256 and the scalar helpers of the C ABI.
#include "pt_host.h"
#include "../../include/procgen_pt.h"

#include <cmath>
#include <cstring>

namespace pt {

static thread_local int g_err_code = PT_OK;
static thread_local std::string g_err_msg;

int set_error(int code, const std::string &msg)
{
    g_err_code = code;
    g_err_msg = msg;
    return code;
}

int clear_error()
{
    g_err_code = PT_OK;
    g_err_msg.clear();
    return PT_OK;
}

static inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

void sha256(const uint8_t *data, size_t len, uint8_t out[32])
{
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    // message schedule over the padded message, one 64-byte block at a time
    const uint64_t bitlen = (uint64_t)len * 8u;
    const size_t padded = ((len + 9 + 63) / 64) * 64;
    for (size_t off = 0; off < padded; off += 64) {
        uint8_t blk[64];
        for (size_t i = 0; i < 64; i++) {
            size_t p = off + i;
            uint8_t b = 0;
            if (p < len) b = data[p];
            else if (p == len) b = 0x80;
            else if (p >= padded - 8) b = (uint8_t)(bitlen >> (8 * (padded - 1 - p)));
            blk[i] = b;
        }
        uint32_t w[64];
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t v[8];
        std::memcpy(v, h, sizeof v);
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = v[7] + (rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[i] + w[i];
            uint32_t t2 = (rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1;
            v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) h[i] += v[i];
    }
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
    }
}

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // namespace pt

extern "C" {

int pt_last_error(void) { return pt::g_err_code; }
const char *pt_last_error_string(void) { return pt::g_err_msg.c_str(); }

// reference src/ParallelTempering.Core.fs:88-95
int pt_geometric_ladder(double t_min, double t_max, int32_t n, double *out)
{
    if (!out) return 0;
    if (n < 2 || t_min <= 0.0 || t_max <= 0.0 || t_min > t_max) { out[0] = t_max; return 1; }
    const double ratio = t_max / t_min;
    for (int i = 0; i < n; i++) out[i] = t_max * std::pow(ratio, -(double)i / (double)(n - 1));
    return n;
}

// reference src/ParallelTempering.Core.fs:138-139
int32_t pt_swap_frequency(int32_t node_count)
{
    int f = (int)(std::sqrt((double)node_count) * 2.0);
    return f > 5 ? f : 5;
}

// reference src/Deterministic.fs:94-102 (`uint64 x` of a negative int sign-extends)
uint64_t pt_hash_coordinates(uint64_t base_seed, int32_t x, int32_t y, int32_t z)
{
    uint64_t h = base_seed;
    h ^= (uint64_t)(int64_t)x * 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h ^= (uint64_t)(int64_t)y * 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    h ^= (uint64_t)(int64_t)z * 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    return h;
}

// little-endian u64 of SHA-256(utf8(seed))[0..8]: the same eight bytes
// HashRNG.bytesToFloat reads (reference src/Deterministic.fs:14-22)
uint64_t pt_seed_key(const char *seed)
{
    uint8_t dg[32];
    pt::sha256((const uint8_t *)seed, std::strlen(seed), dg);
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | dg[i];
    return v;
}

}  // extern "C"

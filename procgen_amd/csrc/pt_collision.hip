// collision_detection: the second entry point of libgpu_compute the reference host calls from product
// code (reference src/GPU.Compute.fs:42-49, wrapper src/GPU.Wrappers.fs:347-416).  Semantics are those of
// the host's own CPU routine, SpatialHash.detectCollisions (reference src/SpatialHash.fs:89-146): the
// set of pairs i < j with  eps < (double)|p_j - p_i| < minDistance,  the distance taken in fp32 as
// sqrt((dx*dx + dy*dy) + dz*dz) without fused multiply-add.  (The reference's CUDA kernel tests
// dist^2 < min^2 over all n^2 thread pairs and has no lower guard.)  Pairs come back sorted by (i, j).
#include "pt_device.h"
#include "pt_host.h"
#include "../../include/procgen_pt.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

namespace {

struct CollParams {
    int n, mbits;               // 2^mbits buckets
    double inv_cell;
    float d2_hi, d2_lo;
    const float *pos;           // [n][3]
    int *cx;                    // [n][3] cell coordinates
    int *bucket_of;             // [n]
    int *start;                 // [M + 1] bucket offsets
    int *sorted;                // [n] node ids by bucket
    int2 *pairs;                // output, or null while counting
    unsigned long long *count;
    unsigned long long capacity;
};

__device__ __forceinline__ int cd_bucket(int x, int y, int z, int mbits)
{
    const unsigned h = (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u;
    return (int)((h ^ (h >> 15)) & ((1u << mbits) - 1u));
}

// Cells are a hair wider than minDistance, so two points closer than minDistance always sit in the same
// or in adjacent cells whatever the rounding of the scaling.
__global__ void cd_cells_kernel(CollParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v = floor((double)P.pos[3 * i + k] * P.inv_cell);
        v = fmin(fmax(v, -1073741824.0), 1073741824.0);
        c[k] = (int)v;
        P.cx[3 * i + k] = c[k];
    }
    const int b = cd_bucket(c[0], c[1], c[2], P.mbits);
    P.bucket_of[i] = b;
    atomicAdd(&P.start[b + 1], 1);
}

__global__ void cd_fill_kernel(CollParams P, int *cursor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const int b = P.bucket_of[i];
    P.sorted[P.start[b] + atomicAdd(&cursor[b], 1)] = i;
}

// every node scans the 27 cells around its own; a candidate counts only if it really lies in the cell
// being scanned (other cells can share the bucket), which also keeps each pair from being seen twice
__global__ void cd_pairs_kernel(CollParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const float xi = P.pos[3 * i], yi = P.pos[3 * i + 1], zi = P.pos[3 * i + 2];
    const int cx = P.cx[3 * i], cy = P.cx[3 * i + 1], cz = P.cx[3 * i + 2];
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
        const int b = cd_bucket(nx, ny, nz, P.mbits);
        for (int k = P.start[b]; k < P.start[b + 1]; k++) {
            const int j = P.sorted[k];
            if (j <= i) continue;
            if (P.cx[3 * j] != nx || P.cx[3 * j + 1] != ny || P.cx[3 * j + 2] != nz) continue;
            const float ddx = P.pos[3 * j] - xi, ddy = P.pos[3 * j + 1] - yi, ddz = P.pos[3 * j + 2] - zi;
            float s2 = ddx * ddx + ddy * ddy;
            s2 = s2 + ddz * ddz;
            if (s2 < P.d2_hi && s2 >= P.d2_lo) {
                const unsigned long long slot = atomicAdd(P.count, 1ull);
                if (P.pairs && slot < P.capacity) P.pairs[slot] = make_int2(i, j);
            }
        }
    }
}

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <class T> T *as() const { return (T *)p; }
};

template <class Pred> float first_float_where(Pred pred)
{
    uint32_t lo = 0u, hi = 0x7f800000u;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        float f; std::memcpy(&f, &mid, 4);
        if (pred(f)) hi = mid; else lo = mid + 1;
    }
    float f; std::memcpy(&f, &lo, 4);
    return f;
}

// pairs_out holds at most `capacity` pairs; *count receives the true number found
int detect(const float *pos, int n, double min_distance, double eps, int *pairs_out, long long capacity, long long *count)
{
    *count = 0;
    if (n < 2) return PT_OK;
    if (pt::device_count() < 1) return pt::set_error(PT_ERR_NO_DEVICE, "no HIP device");
    int mbits = 6;
    while ((1ll << mbits) < 2ll * n && mbits < 26) mbits++;
    const size_t M = (size_t)1 << mbits;
    Buf dpos, dcx, dbucket, dstart, dcursor, dsorted, dcount, dpairs;
    PT_HIP_TRY(dpos.alloc(sizeof(float) * 3 * (size_t)n)); PT_HIP_TRY(dcx.alloc(sizeof(int) * 3 * (size_t)n));
    PT_HIP_TRY(dbucket.alloc(sizeof(int) * (size_t)n)); PT_HIP_TRY(dstart.alloc(sizeof(int) * (M + 1)));
    PT_HIP_TRY(dcursor.alloc(sizeof(int) * M)); PT_HIP_TRY(dsorted.alloc(sizeof(int) * (size_t)n));
    PT_HIP_TRY(dcount.alloc(sizeof(unsigned long long)));
    PT_HIP_TRY(hipMemcpy(dpos.p, pos, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemset(dstart.p, 0, sizeof(int) * (M + 1)));
    PT_HIP_TRY(hipMemset(dcursor.p, 0, sizeof(int) * M));
    PT_HIP_TRY(hipMemset(dcount.p, 0, sizeof(unsigned long long)));
    CollParams P;
    std::memset(&P, 0, sizeof P);
    P.n = n; P.mbits = mbits;
    P.inv_cell = 1.0 / (min_distance * (1.0 + 0x1p-20));
    P.d2_hi = first_float_where([&](float s) { return (double)sqrtf(s) >= min_distance; });
    P.d2_lo = first_float_where([&](float s) { return (double)sqrtf(s) > eps; });
    P.pos = dpos.as<float>(); P.cx = dcx.as<int>(); P.bucket_of = dbucket.as<int>(); P.start = dstart.as<int>();
    P.sorted = dsorted.as<int>(); P.count = dcount.as<unsigned long long>();
    const int threads = 256, blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(cd_cells_kernel, dim3(blocks), dim3(threads), 0, 0, P);
    PT_HIP_TRY(hipGetLastError());
    // bucket offsets: an exclusive scan of M + 1 counters, done on the host (one small round trip)
    std::vector<int> start(M + 1);
    PT_HIP_TRY(hipMemcpy(start.data(), dstart.p, sizeof(int) * (M + 1), hipMemcpyDeviceToHost));
    { int run = 0; for (size_t b = 0; b <= M; b++) { run += start[b]; start[b] = run; } }
    PT_HIP_TRY(hipMemcpy(dstart.p, start.data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cd_fill_kernel, dim3(blocks), dim3(threads), 0, 0, P, dcursor.as<int>());
    PT_HIP_TRY(hipGetLastError());
    // first pass counts, second pass writes
    hipLaunchKernelGGL(cd_pairs_kernel, dim3(blocks), dim3(threads), 0, 0, P);
    PT_HIP_TRY(hipGetLastError());
    unsigned long long found = 0;
    PT_HIP_TRY(hipMemcpy(&found, dcount.p, sizeof found, hipMemcpyDeviceToHost));
    *count = (long long)found;
    if (found == 0 || !pairs_out) return PT_OK;
    PT_HIP_TRY(dpairs.alloc(sizeof(int2) * (size_t)found));
    PT_HIP_TRY(hipMemset(dcount.p, 0, sizeof(unsigned long long)));
    P.pairs = dpairs.as<int2>(); P.capacity = found;
    hipLaunchKernelGGL(cd_pairs_kernel, dim3(blocks), dim3(threads), 0, 0, P);
    PT_HIP_TRY(hipGetLastError());
    std::vector<int2> h((size_t)found);
    PT_HIP_TRY(hipMemcpy(h.data(), dpairs.p, sizeof(int2) * (size_t)found, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end(), [](const int2 &a, const int2 &b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    const long long m = std::min<long long>((long long)found, capacity);
    for (long long k = 0; k < m; k++) { pairs_out[2 * k] = h[(size_t)k].x; pairs_out[2 * k + 1] = h[(size_t)k].y; }
    return PT_OK;
}

template <class Real>
void entry(const Real *positions, int *collision_pairs, int node_count, Real min_distance, int *out_count)
{
    pt::clear_error();
    if (node_count <= 0 || !positions || !out_count) { if (out_count) *out_count = 0; return; }
    if (!(min_distance > 0) || !std::isfinite((double)min_distance)) {     // reference src/GPU.Wrappers.fs:382-383
        pt::set_error(PT_ERR_INVALID_ARGUMENT, "minDistance must be positive and finite");
        return;
    }
    std::vector<float> pos(3 * (size_t)node_count);
    for (size_t k = 0; k < pos.size(); k++) pos[k] = (float)positions[k];
    // the host sizes its buffer for nodeCount * nodeCount / 2 pairs (reference src/GPU.Wrappers.fs:354-359)
    const long long capacity = (long long)node_count * node_count / 2;
    long long found = 0;
    if (detect(pos.data(), node_count, (double)min_distance, 0.001, collision_pairs, capacity, &found) != PT_OK) return;   // outputs untouched
    *out_count = (int)std::min<long long>(found, capacity);
}

}  // namespace

extern "C" {

void collision_detection(const double *positions, int *collision_pairs, int node_count, double min_distance, int *out_count)
{
    entry<double>(positions, collision_pairs, node_count, min_distance, out_count);
}

void collision_detection_f32(const float *positions, int *collision_pairs, int node_count, float min_distance, int *out_count)
{
    entry<float>(positions, collision_pairs, node_count, min_distance, out_count);
}

}  // extern "C"

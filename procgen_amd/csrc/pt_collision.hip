// collision_detection: the second entry point of libgpu_compute the reference host calls from product
// code (reference src/GPU.Compute.fs:42-49, wrapper src/GPU.Wrappers.fs:347-416).  Semantics are those of
// the host's own CPU routine, SpatialHash.detectCollisions (reference src/SpatialHash.fs:89-146): the
// set of pairs i < j with  eps < (double)|p_j - p_i| < minDistance,  the distance taken in fp32 as
// sqrt((dx*dx + dy*dy) + dz*dz) without fused multiply-add.  (The reference's CUDA kernel tests
// dist^2 < min^2 over all n^2 thread pairs and has no lower guard.)  Pairs come back sorted by (i, j).
//
// Everything between the upload of the positions and the download of the pairs runs on the device:
//   cells    cell coordinates and bucket of every node, bucket sizes (atomics)
//   scan     bucket sizes -> bucket offsets (one workgroup)
//   fill     node ids ordered by bucket
//   count    every node counts its partners j > i over the 27 cells around it
//   scan     counts -> offsets into the pair list (one workgroup; the last offset is the total)
//   emit     every node writes its partners at its offset and sorts them: node i's run follows node
//            i-1's, so the list is in (i, j) order with no sort over the whole list
// The work arrays are either a grow-only set the library allocates (the reference entry points) or a
// fixed array in device memory (pt_collision_detect_device, which allocates nothing).
#include "pt_device.h"
#include "pt_host.h"
#include "../../include/procgen_pt.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
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
    int *start;                 // [M + 1] bucket sizes at [b + 1], then bucket offsets at [b]
    int *cursor;                // [M]
    int *sorted;                // [n] node ids by bucket
    unsigned long long *off;    // [n + 1] partner counts at [i + 1], then offsets at [i]; [n] = number of pairs
    int2 *pairs;                // output
    unsigned long long capacity;    // pairs the output holds
    int *count_out;             // min(number of pairs, capacity), or null
};

// ints of work memory for n nodes and 2^mbits buckets (the off array counts double)
__host__ __device__ constexpr size_t cd_work_ints(size_t n, size_t M) { return 3 * n + n + (M + 1) + M + n + 2 * (n + 1) + 2; }

int cd_mbits(int n)
{
    int mbits = 6;
    while ((1ll << mbits) < 2ll * n && mbits < 26) mbits++;
    return mbits;
}

// carve the work arrays out of one block of ints (8-byte aligned base)
void cd_carve(CollParams &P, int *base)
{
    const size_t n = (size_t)P.n, M = (size_t)1 << P.mbits;
    P.off = (unsigned long long *)base; base += 2 * (n + 1);
    P.cx = base; base += 3 * n;
    P.bucket_of = base; base += n;
    P.sorted = base; base += n;
    P.start = base; base += M + 1;
    P.cursor = base;
}

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

// In-place inclusive scan of arr[0 .. len) by one workgroup: thread t owns a contiguous run.  With the
// sizes stored one place up (arr[0] = 0) the result reads as the exclusive offsets plus the total.
template <class T>
__global__ __launch_bounds__(1024) void cd_scan_kernel(T *arr, long long len)
{
    __shared__ T wave_tot[17];
    const int B = blockDim.x, tid = threadIdx.x;
    const long long chunk = (len + B - 1) / B;
    const long long beg = min((long long)tid * chunk, len), end = min(beg + chunk, len);
    T s = 0;
    for (long long i = beg; i < end; i++) s += arr[i];
    const int lane = tid & (PT_WAVE - 1), wv = tid / PT_WAVE, nw = (B + PT_WAVE - 1) / PT_WAVE;
    T inc = s;
#pragma unroll
    for (int o = 1; o < PT_WAVE; o <<= 1) {
        const T up = __shfl_up(inc, o, PT_WAVE);
        if (lane >= o) inc += up;
    }
    if (lane == PT_WAVE - 1) wave_tot[wv] = inc;
    __syncthreads();
    if (tid == 0) {
        T run = 0;
        for (int w = 0; w < nw; w++) { const T t = wave_tot[w]; wave_tot[w] = run; run += t; }
    }
    __syncthreads();
    T run = wave_tot[wv] + inc - s;
    for (long long i = beg; i < end; i++) { run += arr[i]; arr[i] = run; }
}

__global__ void cd_fill_kernel(CollParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const int b = P.bucket_of[i];
    P.sorted[P.start[b] + atomicAdd(&P.cursor[b], 1)] = i;
}

// Node i's partners: every node scans the 27 cells around its own; a candidate counts only if it really
// lies in the cell being scanned (other cells can share the bucket, and the same bucket can come up for
// two of the 27 cells), which keeps each pair from being seen twice.  visit(j) is called per partner.
template <class Visit>
__device__ __forceinline__ void cd_partners(const CollParams &P, int i, Visit visit)
{
    const float xi = P.pos[3 * i], yi = P.pos[3 * i + 1], zi = P.pos[3 * i + 2];
    const int cx = P.cx[3 * i], cy = P.cx[3 * i + 1], cz = P.cx[3 * i + 2];
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
        const int b = cd_bucket(nx, ny, nz, P.mbits);
        const int k1 = P.start[b + 1];
        for (int k = P.start[b]; k < k1; k++) {
            const int j = P.sorted[k];
            if (j <= i) continue;
            if (P.cx[3 * j] != nx || P.cx[3 * j + 1] != ny || P.cx[3 * j + 2] != nz) continue;
            const float ddx = P.pos[3 * j] - xi, ddy = P.pos[3 * j + 1] - yi, ddz = P.pos[3 * j + 2] - zi;
            float s2 = ddx * ddx + ddy * ddy;
            s2 = s2 + ddz * ddz;
            if (s2 < P.d2_hi && s2 >= P.d2_lo) visit(j);
        }
    }
}

__global__ void cd_count_kernel(CollParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) P.off[0] = 0ull;
    if (i >= P.n) return;
    unsigned long long c = 0;
    cd_partners(P, i, [&](int) { c++; });
    P.off[i + 1] = c;
}

// heap sort of the y members of a[0 .. m) (all x members are equal)
__device__ void cd_sort_run(int2 *a, long long m)
{
    auto sift = [&](long long root, long long end) {
        const int v = a[root].y;
        for (;;) {
            long long ch = 2 * root + 1;
            if (ch >= end) break;
            if (ch + 1 < end && a[ch + 1].y > a[ch].y) ch++;
            if (a[ch].y <= v) break;
            a[root].y = a[ch].y;
            root = ch;
        }
        a[root].y = v;
    };
    for (long long k = m / 2 - 1; k >= 0; k--) sift(k, m);
    for (long long end = m - 1; end > 0; end--) {
        const int t = a[0].y; a[0].y = a[end].y; a[end].y = t;
        sift(0, end);
    }
}

__global__ void cd_emit_kernel(CollParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && P.count_out) *P.count_out = (int)min(P.off[P.n], P.capacity);
    if (i >= P.n) return;
    const unsigned long long at = P.off[i];
    if (at >= P.capacity || P.off[i + 1] == at) return;
    // a run cut short by the capacity is sorted first where it was counted in full: the pairs kept are
    // then the first of the (i, j) order.  That needs the whole run in memory, so such a run (at most
    // one per call) goes through the slow path: repeated selection of the next smallest partner.
    const unsigned long long want = P.off[i + 1] - at, room = P.capacity - at;
    if (want <= room) {
        unsigned long long k = at;
        cd_partners(P, i, [&](int j) { P.pairs[k++] = make_int2(i, j); });
        if (want > 1) cd_sort_run(P.pairs + at, (long long)want);
    } else {
        int last = i;
        for (unsigned long long k = 0; k < room; k++) {
            int best = 0x7fffffff;
            cd_partners(P, i, [&](int j) { if (j > last && j < best) best = j; });
            P.pairs[at + k] = make_int2(i, best);
            last = best;
        }
    }
}

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

void cd_thresholds(CollParams &P, double min_distance, double eps)
{
    P.inv_cell = 1.0 / (min_distance * (1.0 + 0x1p-20));
    P.d2_hi = first_float_where([&](float s) { return (double)sqrtf(s) >= min_distance; });
    P.d2_lo = first_float_where([&](float s) { return (double)sqrtf(s) > eps; });
}

// the passes up to the pair offsets; P.off[n] then holds the number of pairs
int cd_launch_count(const CollParams &P, hipStream_t st)
{
    const size_t M = (size_t)1 << P.mbits;
    const int threads = 256, blocks = (P.n + threads - 1) / threads;
    PT_HIP_TRY(hipMemsetAsync(P.start, 0, sizeof(int) * (2 * M + 1), st));      // start[] and cursor[] are adjacent
    hipLaunchKernelGGL(cd_cells_kernel, dim3(blocks), dim3(threads), 0, st, P);
    hipLaunchKernelGGL(cd_scan_kernel<int>, dim3(1), dim3(1024), 0, st, P.start, (long long)(M + 1));
    hipLaunchKernelGGL(cd_fill_kernel, dim3(blocks), dim3(threads), 0, st, P);
    hipLaunchKernelGGL(cd_count_kernel, dim3(blocks), dim3(threads), 0, st, P);
    hipLaunchKernelGGL(cd_scan_kernel<unsigned long long>, dim3(1), dim3(1024), 0, st, P.off, (long long)P.n + 1);
    PT_HIP_TRY(hipGetLastError());
    return PT_OK;
}

int cd_launch_emit(const CollParams &P, hipStream_t st)
{
    const int threads = 256, blocks = (P.n + threads - 1) / threads;
    hipLaunchKernelGGL(cd_emit_kernel, dim3(blocks), dim3(threads), 0, st, P);
    PT_HIP_TRY(hipGetLastError());
    return PT_OK;
}

// ---- work memory of the reference entry points: allocated on first use, grown when a call needs more ----
struct GrowBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t need(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        const size_t want = n + n / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
};
struct Workspace {
    std::mutex mu;
    int device = -1;
    GrowBuf pos, work, pairs;
};
Workspace g_ws;
constexpr size_t RETAIN_PAIR_BYTES = (size_t)64 << 20;      // a larger pair list is freed when its call ends

// pairs_out holds at most `capacity` pairs; *count receives the true number found
int detect(const float *pos, int n, double min_distance, double eps, int *pairs_out, long long capacity, long long *count)
{
    *count = 0;
    if (n < 2) return PT_OK;
    if (pt::device_count() < 1) return pt::set_error(PT_ERR_NO_DEVICE, "no HIP device");
    std::lock_guard<std::mutex> lock(g_ws.mu);
    int dev = 0;
    PT_HIP_TRY(hipGetDevice(&dev));
    if (dev != g_ws.device) {       // the work memory belongs to the device it was allocated on
        if (g_ws.device >= 0) {
            int prev = dev;
            (void)hipSetDevice(g_ws.device);
            for (GrowBuf *b : {&g_ws.pos, &g_ws.work, &g_ws.pairs}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->bytes = 0; }
            (void)hipSetDevice(prev);
        }
        g_ws.device = dev;
    }
    CollParams P;
    std::memset(&P, 0, sizeof P);
    P.n = n; P.mbits = cd_mbits(n);
    const size_t M = (size_t)1 << P.mbits;
    PT_HIP_TRY(g_ws.pos.need(sizeof(float) * 3 * (size_t)n));
    PT_HIP_TRY(g_ws.work.need(sizeof(int) * cd_work_ints((size_t)n, M)));
    cd_thresholds(P, min_distance, eps);
    P.pos = (const float *)g_ws.pos.p;
    cd_carve(P, (int *)g_ws.work.p);
    hipStream_t st = nullptr;
    PT_HIP_TRY(hipMemcpyAsync(g_ws.pos.p, pos, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    { int rc = cd_launch_count(P, st); if (rc) return rc; }
    unsigned long long found = 0;
    PT_HIP_TRY(hipMemcpyAsync(&found, P.off + n, sizeof found, hipMemcpyDeviceToHost, st));
    PT_HIP_TRY(hipStreamSynchronize(st));
    *count = (long long)found;
    const unsigned long long keep = std::min<unsigned long long>(found, (unsigned long long)std::max<long long>(capacity, 0));
    if (keep == 0 || !pairs_out) return PT_OK;
    PT_HIP_TRY(g_ws.pairs.need(sizeof(int2) * (size_t)keep));
    P.pairs = (int2 *)g_ws.pairs.p; P.capacity = keep; P.count_out = nullptr;
    { int rc = cd_launch_emit(P, st); if (rc) return rc; }
    PT_HIP_TRY(hipMemcpyAsync(pairs_out, g_ws.pairs.p, sizeof(int2) * (size_t)keep, hipMemcpyDeviceToHost, st));
    PT_HIP_TRY(hipStreamSynchronize(st));
    // one dense call does not pin its pair list for the life of the process
    if (g_ws.pairs.bytes > RETAIN_PAIR_BYTES) { (void)hipFree(g_ws.pairs.p); g_ws.pairs.p = nullptr; g_ws.pairs.bytes = 0; }
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

// ---- work memory of pt_collision_detect_device: a fixed array in device memory, one copy per GPU ----
constexpr int NOALLOC_MAX_NODES = PT_COLLISION_DEVICE_MAX_NODES;
constexpr size_t NOALLOC_MAX_BUCKETS = (size_t)2 * NOALLOC_MAX_NODES;      // cd_mbits(max nodes) buckets
constexpr size_t NOALLOC_INTS = cd_work_ints(NOALLOC_MAX_NODES, NOALLOC_MAX_BUCKETS);
static_assert((NOALLOC_MAX_NODES & (NOALLOC_MAX_NODES - 1)) == 0, "the bucket bound assumes a power of two");

}  // namespace

static __device__ __align__(256) int scratch[NOALLOC_INTS];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
int *g_scratch_ptr[MAX_DEVICES];        // device address of `scratch` on each GPU, looked up once per GPU

// The name `scratch` is a host-side handle, not an address a kernel can use: the device address of
// this GPU's copy comes from the runtime.
int scratch_address(int dev, int **out)
{
    if (dev < 0 || dev >= MAX_DEVICES) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        int *p = nullptr;
        PT_HIP_TRY(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) return pt::set_error(PT_ERR_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return PT_OK;
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

void pt_collision_release(void)
{
    std::lock_guard<std::mutex> lock(g_ws.mu);
    if (g_ws.device < 0) return;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    if (hipSetDevice(g_ws.device) == hipSuccess)
        for (GrowBuf *b : {&g_ws.pos, &g_ws.work, &g_ws.pairs}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->bytes = 0; }
    if (have_prev) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    g_ws.device = -1;
}

int pt_collision_detect_device(const float *d_positions, int node_count, float min_distance, int *d_pairs,
                               long long pair_capacity, int *d_count, void *stream)
{
    pt::clear_error();
    if (!d_positions || !d_count || node_count < 0 || pair_capacity < 0 || (pair_capacity > 0 && !d_pairs))
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null pointer or negative size");
    if (pair_capacity > 0x7fffffffll)       // *d_count is an int
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "pair_capacity above INT_MAX: the count is reported as an int");
    if (node_count > NOALLOC_MAX_NODES)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "pt_collision_detect_device holds at most " + std::to_string(NOALLOC_MAX_NODES) +
                                                       " nodes in its fixed work memory; use collision_detection for more");
    if (!(min_distance > 0) || !std::isfinite((double)min_distance))
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "minDistance must be positive and finite");
    hipStream_t st = (hipStream_t)stream;
    if (node_count < 2) {
        PT_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int), st));
        return PT_OK;
    }
    int dev = 0;
    PT_HIP_TRY(hipGetDevice(&dev));
    int *work = nullptr;
    { int rc = scratch_address(dev, &work); if (rc) return rc; }
    CollParams P;
    std::memset(&P, 0, sizeof P);
    P.n = node_count; P.mbits = cd_mbits(node_count);
    cd_thresholds(P, (double)min_distance, 0.001);
    P.pos = d_positions;
    cd_carve(P, work);
    P.pairs = (int2 *)d_pairs; P.capacity = (unsigned long long)pair_capacity; P.count_out = d_count;
    { int rc = cd_launch_count(P, st); if (rc) return rc; }
    return cd_launch_emit(P, st);
}

}  // extern "C"

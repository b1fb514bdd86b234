// Host side of the batched spatial-graph engine (C ABI in include/procgen_pt.h).
#include "pt_graph_kernels.h"
#include "pt_host.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "pt_graph_kernels.hip.inc"
#include "pt_graph_delta.hip.inc"
#include "pt_graph_big.hip.inc"
#include "pt_graph_bigc.hip.inc"

namespace {

constexpr size_t LDS_BUDGET = 160 * 1024 - 512;     // gfx950: 160 KiB per workgroup

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n ? n : 16;
        return hipMalloc(&p, bytes);
    }
    template <class T> T *as() const { return (T *)p; }
};

}  // namespace

struct pt_graph_engine {
    pt_graph_config cfg;
    int N = 0, Ecap = 0, T = 0, G = 0, R = 0, Tl = 0, f = 0;
    int device = 0, block = 0;
    bool lds_mode = false;
    bool delta_mode = false;    // incremental evaluation (pt_graph_delta_kernel)
    bool delta_lists = false;   // ... with per-node lists of incident edges (global memory) and their fill counts (LDS)
    int walk[3] = {DELTA_WALK_MEMBERS, DELTA_CAND, DELTA_WALK_ROUNDS};  // bounds of the subtree walk through those lists: nodes, entering edges, round trips
    bool big_mode = false;      // incremental evaluation in global memory (pt_graph_big_kernel)
    int CB = 1;                 // ... by that many workgroups per replica (pt_graph_bigc_kernel when > 1)
    int ept = 24;               // edges each thread of that kernel holds in registers (24: any workgroup size, 48: at most 512 threads)
    int C = 1;                  // workgroups per replica
    size_t lds_bytes = 0, delta_lds_bytes = 0;
    size_t bigc_lds_bytes = 0;  // pt_graph_bigc_kernel: its header plus the room for distances between the two distribution passes
    hipStream_t stream = nullptr;
    GraphParams P;
    SwapParams SP;
    DevBuf px, py, pz, ea, eb, et, ebuf, ecount, kind, value, origin, terminal, rowend, adj, adj4, cost, temps,
        slot_of_rep, rep_of_slot, cells, sorted, queue, bitmap, collbuf, comm, chain_u, swap_u, trace, swap_trace,
        stats, swap_stats, status, terms, group_keys, legacy_in, legacy_out, legacy_flag, dacc, dreach, dvis, dparent, dparscr, dlabel, dlabscr, epk,
        dparent32, dparscr32, bvis, bjmp, bdist, bcomm, binc, bdeg,
        conv_hist, conv_swaps, conv_swap_total, conv_out;
    // the drop-in iteration: what the caller's position array holds if nobody touched it since the last call
    std::vector<unsigned char> legacy_shadow;
    bool legacy_shadow_valid = false;
    int conv_window = 0;            // 0: convergence monitoring off
    long long conv_first_step = 0;  // first step the history holds
    long long step = 0;
    long long steps_since_init = 0, relabel_every = 0;     // incremental kernels: see launch_steps
    bool relabel_forced = false;                            // test hook: rebuild at every interval, whatever the labels reached
    long long trace_steps = 0, trace_passes = 0;
    bool uploaded = false;
    bool stepped = false;                      // a step kernel has run since the last upload (ladder keys are fixed from then on)
    std::mutex mu;
    // timing of the step kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double timed_ms = 0.0;
    long long timed_launches = 0;
    bool timing = true;

    ~pt_graph_engine()
    {
        for (auto &e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Smallest non-negative float s for which pred(s) holds; pred must be monotone (false ... true).
// Positive floats order like their bit patterns, so this is a binary search over the bits.
template <class Pred> float first_float_where(Pred pred)
{
    uint32_t lo = 0u, hi = 0x7f800000u;             // +0 .. +inf
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        float f; std::memcpy(&f, &mid, 4);
        if (pred(f)) hi = mid; else lo = mid + 1;
    }
    float f; std::memcpy(&f, &lo, 4);
    return f;
}

int drain_events(pt_graph_engine *e)
{
    for (size_t i = 0; i < e->events_used; i++) {
        float ms = 0.f;
        PT_HIP_TRY(hipEventSynchronize(e->events[i].second));
        PT_HIP_TRY(hipEventElapsedTime(&ms, e->events[i].first, e->events[i].second));
        e->timed_ms += ms;
        e->timed_launches++;
    }
    e->events_used = 0;
    return PT_OK;
}

int launch_init(pt_graph_engine *e, const GraphParams &P, int write_cost);

// What a kernel can only flag reaches the caller when the stream is next waited for: a cost term
// outside the exact sums' range (the tracked cost is then wrong), injected draws that ran out, a
// cluster of workgroups that lost a member.  The bits stay set until the next upload.  (A full edge
// list, PT_ERR_CAPACITY, is a defined no-op move and only shows in the statistics.)
int device_status(const void *d_status)
{
    unsigned bits = 0;
    PT_HIP_TRY(hipMemcpy(&bits, d_status, sizeof bits, hipMemcpyDeviceToHost));
    if (bits & (1u << PT_ERR_CLUSTER_TIMEOUT))
        return pt::set_error(PT_ERR_CLUSTER_TIMEOUT, "workgroups sharing one replica lost each other (the device was not theirs alone); the run stopped early");
    if (bits & (1u << PT_ERR_NUMERIC_RANGE))
        return pt::set_error(PT_ERR_NUMERIC_RANGE, "a cost term was not finite or reached 2^62: the tracked costs are not valid");
    if (bits & (1u << PT_ERR_STREAM_EXHAUSTED))
        return pt::set_error(PT_ERR_STREAM_EXHAUSTED, "the injected draws are shorter than the run");
    return PT_OK;
}

int launch_steps(pt_graph_engine *e, long long step0, int nsteps)
{
    if (nsteps > 0) e->stepped = true;
    if (nsteps <= 0) return PT_OK;
    // The incremental kernels hand out discovery labels from a 32-bit counter that only grows (by at
    // most one per edge looked at in a step).  Long before it can wrap, the state they continue from is
    // rebuilt from scratch by the init kernel: same cost, same sums, same reached sets bit for bit, a
    // fresh search tree with labels below N.  No launch runs more than relabel_every steps on one set
    // of labels: a longer request is cut into pieces with a rebuild between them.
    // relabel_every is the worst case (every step re-reaching through every edge).  Real steps use a
    // handful of labels, and a rebuild is a full evaluation by one workgroup per replica (tens of
    // milliseconds at a million nodes), so when the worst-case budget is spent the host first reads how
    // far the counters really got and, if at least a quarter of the budget is still there, goes on.
    if (e->delta_mode || e->big_mode) {
        if (e->steps_since_init >= e->relabel_every) {
            long long room = 0;
            if (!e->relabel_forced) {
                std::vector<int> reach((size_t)e->R * 4);
                PT_HIP_TRY(hipStreamSynchronize(e->stream));
                PT_HIP_TRY(hipMemcpy(reach.data(), e->dreach.p, reach.size() * sizeof(int), hipMemcpyDeviceToHost));
                unsigned top = 0;                               // the largest label in use by any replica
                for (int r = 0; r < e->R; r++) top = std::max(top, (unsigned)reach[(size_t)r * 4 + 2]);
                room = (long long)((0xfffffffeull - (unsigned long long)top) / ((unsigned long long)e->Ecap + 2ull));
                room = std::min(room, e->relabel_every);
            }
            if (room > 0 && room * 4 >= e->relabel_every) e->steps_since_init = e->relabel_every - room;
            else {
                int rc = launch_init(e, e->P, 1);
                if (rc) return rc;
                e->steps_since_init = 0;
            }
        }
        const long long room = e->relabel_every - e->steps_since_init;
        if (nsteps > room) {
            int rc = launch_steps(e, step0, (int)room);
            if (rc) return rc;
            return launch_steps(e, step0 + room, nsteps - (int)room);
        }
    }
    e->steps_since_init += nsteps;
    hipEvent_t a = nullptr, b = nullptr;
    if (e->timing) {
        if (e->events_used == e->events.size()) {
            if (e->events.size() >= 4096) { int rc = drain_events(e); if (rc) return rc; }
            else {
                PT_HIP_TRY(hipEventCreate(&a)); PT_HIP_TRY(hipEventCreate(&b));
                e->events.emplace_back(a, b);
            }
        }
        a = e->events[e->events_used].first; b = e->events[e->events_used].second;
        e->events_used++;
        PT_HIP_TRY(hipEventRecord(a, e->stream));
    }
    const dim3 cgrid((unsigned)((e->R + 7) / 8 * 8 * e->C));
    if (e->big_mode && e->CB > 1) {
        PT_HIP_TRY(hipMemsetAsync(e->bcomm.p, 0, e->bcomm.bytes, e->stream));
        const unsigned blocks = (unsigned)((e->R + 7) / 8 * 8 * e->CB);
        if (e->block <= 512)
            hipLaunchKernelGGL(pt_graph_bigc_kernel<512>, dim3(blocks), dim3(e->block), e->bigc_lds_bytes, e->stream, e->P, step0, nsteps);
        else if (e->block <= 768)
            hipLaunchKernelGGL(pt_graph_bigc_kernel<768>, dim3(blocks), dim3(e->block), e->bigc_lds_bytes, e->stream, e->P, step0, nsteps);
        else
            hipLaunchKernelGGL(pt_graph_bigc_kernel<1024>, dim3(blocks), dim3(e->block), e->bigc_lds_bytes, e->stream, e->P, step0, nsteps);
    } else if (e->big_mode)
        hipLaunchKernelGGL(pt_graph_big_kernel, dim3(e->R), dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->delta_mode && e->ept == 48)
        hipLaunchKernelGGL((pt_graph_delta_kernel<48, 32, 512>), dim3(e->R), dim3(e->block), e->delta_lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->delta_mode && e->ept == 32)
        hipLaunchKernelGGL((pt_graph_delta_kernel<32, 16, 768>), dim3(e->R), dim3(e->block), e->delta_lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->delta_mode)
        hipLaunchKernelGGL((pt_graph_delta_kernel<24, 16, 1024>), dim3(e->R), dim3(e->block), e->delta_lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->C == 4)
        hipLaunchKernelGGL(pt_graph_cluster_kernel<4>, cgrid, dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->C == 3)
        hipLaunchKernelGGL(pt_graph_cluster_kernel<3>, cgrid, dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->C == 2)
        hipLaunchKernelGGL(pt_graph_cluster_kernel<2>, cgrid, dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    else if (e->lds_mode)
        hipLaunchKernelGGL(pt_graph_steps_kernel<true>, dim3(e->R), dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    else
        hipLaunchKernelGGL(pt_graph_steps_kernel<false>, dim3(e->R), dim3(e->block), e->lds_bytes, e->stream, e->P, step0, nsteps);
    PT_HIP_TRY(hipGetLastError());
    if (e->timing) PT_HIP_TRY(hipEventRecord(b, e->stream));
    return PT_OK;
}

int launch_init(pt_graph_engine *e, const GraphParams &P, int write_cost)
{
    if (e->lds_mode)
        hipLaunchKernelGGL(pt_graph_init_kernel<true>, dim3(e->R), dim3(e->block), e->lds_bytes, e->stream, P, write_cost);
    else
        hipLaunchKernelGGL(pt_graph_init_kernel<false>, dim3(e->R), dim3(e->block), e->lds_bytes, e->stream, P, write_cost);
    PT_HIP_TRY(hipGetLastError());
    return PT_OK;
}

int launch_swap(pt_graph_engine *e, long long pass, const double *d_energy)
{
    SwapParams sp = e->SP;
    sp.energy = d_energy;
    const int threads = PT_WAVE, blocks = e->G;                 // one wave per ladder
    hipLaunchKernelGGL(pt_graph_swap_kernel, dim3(blocks), dim3(threads), 0, e->stream, sp, pass);
    PT_HIP_TRY(hipGetLastError());
    return PT_OK;
}

// Steps [step, step+n) of the loop at reference src/ParallelTempering.Core.fs:162-177: the swap
// pass follows every step whose index is a multiple of f.  The caller holds the engine's lock.
int run_steps(pt_graph_engine *e, long long steps)
{
    long long s = e->step, end = e->step + steps;
    const long long f = e->f;
    while (s < end) {
        long long next_swap_step = ((s + f - 1) / f) * f;         // first multiple of f that is >= s
        long long last = std::min(end - 1, next_swap_step);       // run through `last` inclusive
        int n = (int)(last - s + 1);
        int rc = launch_steps(e, s, n);
        if (rc) return rc;
        s = last + 1;
        if (last == next_swap_step && e->Tl > 1) {
            rc = launch_swap(e, last / f, e->cost.as<double>());
            if (rc) return rc;
        }
    }
    e->step = end;
    return PT_OK;
}

int enable_convergence(pt_graph_engine *e, int window)
{
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    PT_HIP_TRY(e->conv_hist.alloc(sizeof(double) * (size_t)e->G * e->Tl * window));
    PT_HIP_TRY(e->conv_swaps.alloc((size_t)e->G * window));
    PT_HIP_TRY(e->conv_swap_total.alloc(sizeof(unsigned long long) * (size_t)e->G));
    PT_HIP_TRY(e->conv_out.alloc(sizeof(pt_convergence_metrics) * (size_t)e->G));
    PT_HIP_TRY(hipMemset(e->conv_hist.p, 0, e->conv_hist.bytes));
    PT_HIP_TRY(hipMemset(e->conv_swaps.p, 0, e->conv_swaps.bytes));
    PT_HIP_TRY(hipMemset(e->conv_swap_total.p, 0, e->conv_swap_total.bytes));
    e->conv_window = window;
    e->conv_first_step = e->step;
    e->P.conv_hist = e->conv_hist.as<double>(); e->P.conv_window = window;
    e->SP.conv_hist = e->conv_hist.as<double>(); e->SP.conv_window = window;
    e->SP.conv_swaps = e->conv_swaps.as<uint8_t>();
    e->SP.conv_swap_total = e->conv_swap_total.as<unsigned long long>();
    e->SP.swap_frequency = e->f;
    return PT_OK;
}

int launch_convergence(hipStream_t stream, const ConvParams &cp)
{
    const size_t lds = sizeof(double) * (2 * (size_t)cp.chains + PT_WAVE);
    if (lds > 60000) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "too many chains for the convergence check");
    hipLaunchKernelGGL(pt_graph_convergence_kernel, dim3((unsigned)cp.ladders), dim3(PT_WAVE), lds, stream, cp);
    PT_HIP_TRY(hipGetLastError());
    return PT_OK;
}

int check_convergence(pt_graph_engine *e, double strictness, int iteration, pt_convergence_metrics *out)
{
    const long long recorded = e->step - e->conv_first_step;
    ConvParams cp;
    std::memset(&cp, 0, sizeof cp);
    cp.hist = e->conv_hist.as<double>();
    cp.ring = e->conv_window;
    cp.chain_stride = e->conv_window;
    cp.ladder_stride = (long long)e->conv_window * e->Tl;
    cp.n = (int)std::min<long long>(recorded, e->conv_window);
    cp.first = (int)((e->step - cp.n) % e->conv_window);
    cp.chains = e->Tl; cp.window = e->conv_window; cp.ladders = e->G;
    cp.swaps = e->conv_swaps.as<uint8_t>();
    cp.swap_total = e->conv_swap_total.as<unsigned long long>();
    cp.swap_ring = e->conv_window;
    cp.strictness = strictness; cp.iteration = iteration;
    cp.out = e->conv_out.as<pt_convergence_metrics>();
    int rc = launch_convergence(e->stream, cp);
    if (rc) return rc;
    PT_HIP_TRY(hipMemcpyAsync(out, e->conv_out.p, sizeof(pt_convergence_metrics) * (size_t)e->G, hipMemcpyDeviceToHost, e->stream));
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    return device_status(e->status.p);
}

}  // namespace

extern "C" {

void pt_graph_default_config(pt_graph_config *cfg, int32_t node_count, int32_t edge_capacity,
                             int32_t temp_count, int32_t group_count)
{
    std::memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = (int32_t)sizeof *cfg;
    cfg->node_count = node_count;
    cfg->edge_capacity = edge_capacity;
    cfg->temp_count = temp_count;
    cfg->group_count = group_count;
    cfg->swap_frequency = 0;
    cfg->mutation_mode = PT_MUT_ADAPTIVE;
    cfg->rng_mode = PT_RNG_PHILOX;
    // reference src/ProceduralGeneration.fs:100-107
    const double w[6] = {1.0, 1.0, 0.1, 0.5, 0.05, 0.1};
    std::memcpy(cfg->weights, w, sizeof w);
    cfg->target_edge_length = 2.0;
    cfg->min_distance = 0.5;
    cfg->min_distance_eps = 0.001;
    cfg->philox_key = 0;
    cfg->seed = nullptr;
    cfg->device = -1;
}

int pt_graph_create(const pt_graph_config *cfg, pt_graph_engine **out)
{
    pt::clear_error();
    if (!cfg || !out) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null config or output pointer");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(pt_graph_config))
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "pt_graph_config.struct_size mismatch");
    if (cfg->node_count < 1 || cfg->temp_count < 1 || cfg->group_count < 1 || cfg->edge_capacity < 0)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "node_count, temp_count, group_count must be >= 1");
    if ((long long)cfg->group_count * cfg->temp_count > (1ll << 24))
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "too many replicas");
    int Tl = cfg->ladder_temp_count > 0 ? cfg->ladder_temp_count : cfg->temp_count;
    if (Tl != cfg->temp_count && cfg->group_count != 1)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "a split ladder needs group_count == 1");
    if (cfg->ladder_slot_base < 0 || cfg->ladder_slot_base + cfg->temp_count > Tl)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "ladder_slot_base + temp_count exceeds ladder_temp_count");
    if (!(cfg->min_distance > 0.0) || !std::isfinite(cfg->min_distance))
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "min_distance must be positive and finite");
    if (cfg->mutation_mode != PT_MUT_ADAPTIVE && cfg->mutation_mode != PT_MUT_PERTURB_ONLY)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "unknown mutation_mode");
    if (cfg->rng_mode != PT_RNG_PHILOX && cfg->rng_mode != PT_RNG_INJECTED)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "unknown rng_mode");
    if (pt::device_count() < 1) return pt::set_error(PT_ERR_NO_DEVICE, "no HIP device");

    int dev = cfg->device;
    if (dev < 0) PT_HIP_TRY(hipGetDevice(&dev));
    if (dev >= pt::device_count()) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    DeviceGuard guard(dev);

    pt_graph_engine *e = new (std::nothrow) pt_graph_engine();
    if (!e) return pt::set_error(PT_ERR_OUT_OF_MEMORY, "host allocation failed");
    e->cfg = *cfg;
    e->cfg.seed = nullptr;
    e->device = dev;
    e->N = cfg->node_count; e->Ecap = std::max(cfg->edge_capacity, 1); e->T = cfg->temp_count; e->G = cfg->group_count;
    e->R = e->G * e->T; e->Tl = Tl;
    e->f = cfg->swap_frequency > 0 ? cfg->swap_frequency : pt_swap_frequency(cfg->node_count);
    uint64_t key = cfg->seed ? pt_seed_key(cfg->seed) : cfg->philox_key;

    const int N = e->N, R = e->R;
    // hash grid: M = 2^mb buckets >= 1.25 N, split over three axes, every axis >= 4 cells
    int mb = 6;
    while ((1ll << mb) < (long long)N + N / 4) mb++;
    int gx = (mb + 2) / 3, gy = (mb - gx + 1) / 2, gz = mb - gx - gy;
    const size_t M = (size_t)1 << mb;
    const int nwords = (N + 31) / 32;

    // workgroup size: enough lanes to cover max(N, E) a few elements each
    int block = cfg->block_threads;
    if (block <= 0) {
        int work = std::max(N, e->Ecap);
        block = 64;
        while (block < 1024 && block * 8 < work) block *= 2;
        // Twelve waves with 168 registers each run the incremental step faster than sixteen with 128
        // (10k nodes x 64 replicas: 14.0 against 14.7 us per step) when a replica fits 32 edges and 16
        // nodes per thread.
        if (block == 1024 && N <= 16 * 768 && e->Ecap <= 32 * 768) block = 768;
        // Many more replicas than CUs: the device is kept busy by the number of workgroups, and every
        // wave of a workgroup repeats the per-step fixed work, so the smallest workgroup that still
        // holds a replica's edges (32 per thread in the variant such workgroups get) and caches its nodes (16 per
        // thread) runs fastest (1k-node graphs x 8192 replicas: 256 threads 89 M proposals/s, 192 threads 112 M,
        // 128 threads 142 M, 64 threads -- one wave per replica -- 163 M).
        hipDeviceProp_t prop;
        int cus = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount; else (void)hipGetLastError();
        if (cus > 0 && (long long)e->R >= 2ll * cus) {
            const int small = std::max((N + 15) / 16, (e->Ecap + 31) / 32);
            block = std::min(block, std::max(64, (small + 63) / 64 * 64));
        }
    }
    block = std::min(1024, std::max(64, (block + 63) / 64 * 64));
    e->block = block;

    // Incremental evaluation: the default whenever its LDS layout fits (positions, parents, five
    // visited sets), the edge list fits the workgroup's registers and no cluster was asked for.
    // Kernel variant (edges per thread / registers per thread): 24 / 128 for workgroups above 768 threads,
    // 32 / 168 for smaller ones where the list fits -- the larger register budget is what counts (fewer
    // spills: 14.2 against 15.5 us per step at the headline shape, 117 M against 105 M proposals/s for
    // 1k-node graphs x 8192 replicas) --, else 48 / 256 (more spills again, the slowest of the three).
    {
        const long long cap = e->Ecap;
        if (block > 768) e->ept = 24;
        else if (cap <= 32ll * block) e->ept = 32;
        else if (block <= 512 && cap <= 48ll * block) e->ept = 48;
        else e->ept = 24;
        if (const char *v = std::getenv("PROCGEN_PT_EPT")) {        // diagnostic override
            const int k = std::atoi(v);
            if ((k == 24 || (k == 32 && block <= 768) || (k == 48 && block <= 512)) && cap <= (long long)k * block) e->ept = k;
        }
    }
    if (!(cfg->flags & (PT_FLAG_GLOBAL_MEMORY_PLACEMENT | PT_FLAG_FULL_EVALUATION | PT_FLAG_STREAMING_INCREMENTAL)) && N < 65535 && cfg->cluster_size <= 1 &&
        (long long)e->Ecap <= (long long)e->ept * block) {
        const size_t Npad = ((size_t)N + 3) & ~(size_t)3;
        const size_t need = sizeof(Shared) + Npad * 12 + (((size_t)N * 2 + 15) / 16) * 16 + 5 * ((((size_t)nwords * 4 + 15) / 16) * 16) +
                            sizeof(unsigned) * DELTA_CAND;
        if (need <= LDS_BUDGET) { e->delta_mode = true; e->delta_lds_bytes = need; }
        // a byte per node for the fill counts of the per-node lists of incident edges, where that still fits
        // (PROCGEN_PT_DELTA_LISTS=0 turns the lists off: the parity tests run both)
        const size_t with_lists = need + (((size_t)N + 15) / 16) * 16;
        const char *lv = std::getenv("PROCGEN_PT_DELTA_LISTS");
        if (e->delta_mode && with_lists <= LDS_BUDGET && !(lv && std::atoi(lv) == 0) &&
            sizeof(unsigned) * (size_t)R * N * DELTA_LIST <= ((size_t)8 << 30)) {
            e->delta_lists = true; e->delta_lds_bytes = with_lists;
        }
        // PROCGEN_PT_DELTA_WALK=members,entering,rounds narrows (or widens, up to the buffers' sizes) the bounds within
        // which a cut-off subtree is walked through the lists; the parity tests use it to drive the fall-back paths
        if (const char *wv = std::getenv("PROCGEN_PT_DELTA_WALK")) {
            int m = 0, c = 0, k = 0;
            if (std::sscanf(wv, "%d,%d,%d", &m, &c, &k) == 3) {
                e->walk[0] = std::min(std::max(m, 0), DELTA_SUBTREE);
                e->walk[1] = std::min(std::max(c, 0), DELTA_CAND);
                e->walk[2] = std::max(k, 0);
            }
        }
    }

    // When the LDS-resident incremental kernel cannot take the replica (too many nodes or edges) and
    // nothing asks for the full evaluation, the replica goes to global memory and the streaming
    // incremental kernel: far faster than re-evaluating the whole cost per proposal in LDS.
    const bool stream = (cfg->flags & PT_FLAG_STREAMING_INCREMENTAL) ||
                        (!e->delta_mode && cfg->cluster_size <= 1 &&
                         !(cfg->flags & (PT_FLAG_GLOBAL_MEMORY_PLACEMENT | PT_FLAG_FULL_EVALUATION)));

    // Placement.  LDS-resident: positions (8N + 4N), u16 bucket table (2 M'), u16 sorted ids (2N)
    // and the visited bitmap all in LDS; M' is the largest power of two <= M that fits.
    size_t lds = sizeof(Shared);
    int bitmap_in_lds = 0, cells_in_lds = 0;
    int lgx = 0, lgy = 0, lgz = 0;
    bool lds_mode = false;
    if (!(cfg->flags & PT_FLAG_GLOBAL_MEMORY_PLACEMENT) && !stream && N < 65536) {
        const size_t Npad = ((size_t)N + 3) & ~(size_t)3;
        const size_t fixed = sizeof(Shared) + Npad * 14 + (((size_t)nwords * 4 + 15) / 16) * 16;
        int lmb = mb;
        while (lmb > 6 && fixed + ((size_t)2 << lmb) > LDS_BUDGET) lmb--;
        // accept up to ~2.5 nodes per bucket; beyond that the global placement's bigger table wins
        if (fixed + ((size_t)2 << lmb) <= LDS_BUDGET && ((size_t)1 << lmb) * 5 >= (size_t)N * 2) {
            lds_mode = true;
            lgx = (lmb + 2) / 3; lgy = (lmb - lgx + 1) / 2; lgz = lmb - lgx - lgy;
            lds = fixed + ((size_t)2 << lmb);
            bitmap_in_lds = 1;
        }
    }
    if (!lds_mode) {
        // header, then the visited bitmap, then the bucket array when they fit
        size_t bm_bytes = (((size_t)nwords * 4 + 15) / 16) * 16;
        if (lds + bm_bytes <= LDS_BUDGET) { bitmap_in_lds = 1; lds += bm_bytes; }
        // several small replicas share a CU: keep each workgroup's LDS modest unless the graph is big
        size_t cells_bytes = (M + 1) * 4;
        size_t cells_cap = (block >= 1024) ? LDS_BUDGET : (LDS_BUDGET * block / 1024);
        if (lds + cells_bytes <= cells_cap) { cells_in_lds = 1; lds += cells_bytes; }
    }
    e->lds_bytes = lds;
    e->lds_mode = lds_mode;

    // Replicas that got the global placement for their size (or by the streaming flag) are evaluated
    // incrementally too, by the kernel that keeps everything in global memory.
    if (!e->delta_mode && !lds_mode && !(cfg->flags & (PT_FLAG_GLOBAL_MEMORY_PLACEMENT | PT_FLAG_FULL_EVALUATION)))
        e->big_mode = true;

    {
        // a step raises the counter by at most one per edge (an edge draws a ticket once, when it first
        // finds its head unvisited) plus one; labels restart below N
        e->relabel_every = std::max<long long>(1, (3ll << 30) / ((long long)e->Ecap + 2));
        if (const char *v = std::getenv("PROCGEN_PT_RELABEL_EVERY")) {                   // test hook
            const long long k = std::atoll(v);
            if (k > 0) { e->relabel_every = k; e->relabel_forced = true; }
        }
        if (const char *v = std::getenv("PROCGEN_PT_RELABEL_BUDGET")) {                  // test hook: a small worst-case budget, not forced
            const long long k = std::atoll(v);
            if (k > 0) e->relabel_every = k;
        }
    }
    // Cluster size: as many workgroups per replica (up to 4) as keeps the whole grid co-resident at
    // one workgroup per CU.  Only the LDS-resident placement has a cluster kernel.
    int C = 1;
    if (cfg->cluster_size < 0 || cfg->cluster_size > (e->big_mode ? 16 : 4)) {
        delete e;
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "cluster_size must be 0..4 (0..16 for replicas held in global memory)");
    }
    // A census launch: are `blocks` workgroups with this LDS footprint resident all at once?  It runs on the
    // engine's own stream and waits for that stream only, so it does not stall other engines of the process;
    // what it observes is the device as it is now -- a clustered engine needs the CUs it was promised while it
    // runs (another engine's long kernel can leave a clustered grid partly resident, and the launch then ends
    // with PT_ERR_CLUSTER_TIMEOUT, reported by the next call that waits for the stream).
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete e;
        return pt::set_error(PT_ERR_HIP, "cannot create the engine's stream");
    }
    auto census = [&](unsigned blocks) -> bool {
        DevBuf probe;
        unsigned init[2] = {0u, 1u}, res[2] = {0u, 0u};
        if (probe.alloc(sizeof init) != hipSuccess) { (void)hipGetLastError(); return false; }
        bool ok = hipMemcpyAsync(probe.p, init, sizeof init, hipMemcpyHostToDevice, e->stream) == hipSuccess;
        // (the clustered streaming kernel takes a CU's whole LDS)
        const size_t foot = e->big_mode ? std::max(lds, LDS_BUDGET) : lds;
        if (ok && foot > 64 * 1024)
            ok = hipFuncSetAttribute((const void *)pt_census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)foot) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(pt_census_kernel, dim3(blocks), dim3(64), foot, e->stream, probe.as<unsigned>(), probe.as<unsigned>() + 1, blocks);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpyAsync(res, probe.p, sizeof res, hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
                 hipStreamSynchronize(e->stream) == hipSuccess;
        }
        if (!ok) (void)hipGetLastError();
        return ok && res[0] == blocks && res[1] == 1u;
    };
    if (e->big_mode && cfg->cluster_size != 1) {
        // Several workgroups per replica pay once a member has thousands of nodes to stream.
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); prop.multiProcessorCount = 0; }
        const long long rows = ((long long)R + 7) / 8 * 8;
        int want = cfg->cluster_size;
        if (want == 0) {
            want = 1;
            // measured: worth it from about 6 000 nodes per member (30k nodes: 40 -> 27 us per step with 4)
            while (want < 16 && rows * want * 2 <= prop.multiProcessorCount && (long long)N >= 6000ll * want * 2) want *= 2;
        }
        while (want > 1 && (rows * want > prop.multiProcessorCount || !census((unsigned)(rows * want)))) {
            if (cfg->cluster_size > 1) {
                delete e;
                return pt::set_error(PT_ERR_INVALID_ARGUMENT, "cluster_size: that many workgroups are not co-resident on this device");
            }
            want /= 2;
        }
        e->CB = std::max(1, want);
        // Twelve waves with 168 registers each stream faster than sixteen with 128 and their spills
        // (10^6 nodes x 16 replicas: 240 k against 223 k proposals/s; 512 threads with 256 registers: 215 k).
        if (e->CB > 1 && cfg->block_threads <= 0 && e->block == 1024) e->block = 768;
    }
    if (lds_mode && cfg->cluster_size != 1 && !e->delta_mode) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); prop.multiProcessorCount = 0; }
        const long long slots = (long long)prop.multiProcessorCount * std::max<size_t>(1, LDS_BUDGET / std::max<size_t>(lds, 1));
        const long long rows = ((long long)R + 7) / 8 * 8;
        int want = cfg->cluster_size ? cfg->cluster_size : 4;
        while (want > 1 && rows * want > slots) want--;
        if (cfg->cluster_size > 1 && want != cfg->cluster_size) {
            delete e;
            return pt::set_error(PT_ERR_INVALID_ARGUMENT, "cluster_size needs more co-resident workgroups than the device has CUs");
        }
        C = want;
        // census: shrink the cluster until a grid of that many workgroups (same LDS footprint) is
        // observed to be resident all at once
        while (C > 1 && !census((unsigned)(rows * C))) {
            if (cfg->cluster_size > 1) {
                delete e;
                return pt::set_error(PT_ERR_INVALID_ARGUMENT, "cluster_size: that many workgroups are not co-resident on this device");
            }
            C--;
        }
    }
    e->C = C;

    hipError_t herr = hipSuccess;
    auto A = [&](DevBuf &b, size_t bytes) { if (herr == hipSuccess) herr = b.alloc(bytes); };
    A(e->px, sizeof(float) * R * N); A(e->py, sizeof(float) * R * N); A(e->pz, sizeof(float) * R * N);
    A(e->ea, sizeof(int) * (size_t)R * C * 2 * e->Ecap); A(e->eb, sizeof(int) * (size_t)R * C * 2 * e->Ecap);
    A(e->et, sizeof(double) * (size_t)R * C * 2 * e->Ecap);
    A(e->comm, sizeof(ClusterComm) * (size_t)R);
    A(e->ebuf, sizeof(int) * R); A(e->ecount, sizeof(int) * R);
    A(e->kind, (size_t)e->G * N); A(e->value, sizeof(double) * (size_t)e->G * N);
    A(e->origin, sizeof(int) * e->G); A(e->terminal, sizeof(int) * e->G);
    A(e->rowend, sizeof(int) * (size_t)R * N); A(e->adj, sizeof(int) * (size_t)R * e->Ecap);
    A(e->adj4, sizeof(int4) * (size_t)R * N);
    A(e->cost, sizeof(double) * R);
    A(e->temps, sizeof(double) * (size_t)e->G * Tl);
    A(e->slot_of_rep, sizeof(int) * (size_t)e->G * Tl); A(e->rep_of_slot, sizeof(int) * (size_t)e->G * Tl);
    A(e->cells, (cells_in_lds || lds_mode) ? 16 : sizeof(int) * (size_t)R * (M + 1));
    A(e->sorted, sizeof(int) * (size_t)R * N);
    A(e->queue, sizeof(int) * (size_t)R * 2 * N);
    A(e->bitmap, bitmap_in_lds ? 16 : sizeof(unsigned) * (size_t)R * nwords);
    const int collcap = 64;     // deferred collisions per thread; more are priced on the spot
    A(e->collbuf, sizeof(float) * (size_t)R * C * collcap * block);
    A(e->stats, sizeof(unsigned long long) * (size_t)R * 4);
    A(e->swap_stats, sizeof(unsigned long long) * (size_t)e->G * 2);
    A(e->status, sizeof(unsigned));
    A(e->terms, sizeof(double) * (size_t)R * 8);
    if (e->delta_mode) {
        A(e->dacc, sizeof(fx128) * (size_t)R * ACC_COUNT);
        A(e->dreach, sizeof(int) * (size_t)R * 4);
        A(e->dvis, sizeof(unsigned) * (size_t)R * 2 * nwords);
        A(e->dparent, sizeof(unsigned short) * (size_t)R * N);
        A(e->dparscr, sizeof(unsigned short) * (size_t)R * N);
        A(e->dlabel, sizeof(unsigned) * (size_t)R * N); A(e->dlabscr, sizeof(unsigned) * (size_t)R * N);
        A(e->epk, sizeof(unsigned) * (size_t)R * e->ept * block);
    }
    if (e->delta_lists) A(e->binc, sizeof(unsigned) * (size_t)R * N * DELTA_LIST);
    if (e->big_mode) {
        A(e->dacc, sizeof(fx128) * (size_t)R * ACC_COUNT);
        A(e->dreach, sizeof(int) * (size_t)R * 4);
        A(e->dvis, sizeof(unsigned) * (size_t)R * 2 * nwords);
        A(e->dlabel, sizeof(unsigned) * (size_t)R * N); A(e->dlabscr, sizeof(unsigned) * (size_t)R * N);
        A(e->dparent32, sizeof(unsigned) * (size_t)R * N); A(e->dparscr32, sizeof(unsigned) * (size_t)R * N);
        A(e->bvis, sizeof(unsigned) * (size_t)R * 3 * nwords);
        A(e->bjmp, sizeof(unsigned) * (size_t)R * N);
        A(e->bdist, sizeof(float) * (size_t)R * N);
        if (e->CB > 1) {
            A(e->bcomm, sizeof(BigComm) * (size_t)R);
            A(e->binc, sizeof(unsigned) * (size_t)R * N * BIGC_INC);
            A(e->bdeg, sizeof(int) * (size_t)R * N);
        }
    }
    if (herr == hipSuccess && e->delta_mode && e->delta_lds_bytes > 64 * 1024)
        herr = hipFuncSetAttribute(e->ept == 48 ? (const void *)pt_graph_delta_kernel<48, 32, 512>
                                   : e->ept == 32 ? (const void *)pt_graph_delta_kernel<32, 16, 768>
                                                     : (const void *)pt_graph_delta_kernel<24, 16, 1024>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->delta_lds_bytes);
    if (herr == hipSuccess && lds > 64 * 1024) {
        const void *ks[5] = {lds_mode ? (const void *)pt_graph_steps_kernel<true> : (const void *)pt_graph_steps_kernel<false>,
                             lds_mode ? (const void *)pt_graph_init_kernel<true> : (const void *)pt_graph_init_kernel<false>,
                             (const void *)pt_graph_cluster_kernel<2>, (const void *)pt_graph_cluster_kernel<3>,
                             (const void *)pt_graph_cluster_kernel<4>};
        for (int i = 0; i < (lds_mode ? 5 : 2) && herr == hipSuccess; i++)
            herr = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (herr == hipSuccess && e->big_mode && lds > 64 * 1024) {
        herr = hipFuncSetAttribute((const void *)pt_graph_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (herr == hipSuccess && e->big_mode && e->CB > 1) {
        e->bigc_lds_bytes = LDS_BUDGET;
        herr = hipFuncSetAttribute(e->block <= 512 ? (const void *)pt_graph_bigc_kernel<512>
                                   : e->block <= 768 ? (const void *)pt_graph_bigc_kernel<768> : (const void *)pt_graph_bigc_kernel<1024>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->bigc_lds_bytes);
    }
    if (herr != hipSuccess) {
        int code = herr == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP;
        std::string msg = std::string("engine allocation: ") + hipGetErrorString(herr);
        (void)hipGetLastError();
        delete e;
        return pt::set_error(code, msg);
    }

    GraphParams &P = e->P;
    std::memset(&P, 0, sizeof P);
    P.N = N; P.Ecap = e->Ecap; P.T = e->T; P.G = e->G; P.Tl = Tl;
    P.slot_base = cfg->ladder_slot_base; P.group_base = cfg->group_base;
    P.C = C;
    P.mb = mb; P.gx = gx; P.gy = gy; P.gz = gz; P.nwords = nwords;
    P.lgx = lgx; P.lgy = lgy; P.lgz = lgz;
    P.mutation_mode = cfg->mutation_mode; P.rng_mode = cfg->rng_mode;
    P.cells_in_lds = cells_in_lds; P.bitmap_in_lds = bitmap_in_lds;
    std::memcpy(P.w, cfg->weights, sizeof P.w);
    P.target_len = cfg->target_edge_length; P.min_dist = cfg->min_distance; P.eps = cfg->min_distance_eps;
    // cells a hair wider than min_distance: two points closer than min_distance are then always in
    // adjacent cells whatever the rounding of the coordinate scaling
    P.inv_cell = 1.0 / (cfg->min_distance * (1.0 + 0x1p-20));
    {
        int ex = 0;
        const double mant = std::frexp(cfg->min_distance, &ex);
        P.cell_pow2 = (mant == 0.5 && ex > -100 && ex < 100) ? 1 : 0;
        P.inv_cell_f = P.cell_pow2 ? (float)(1.0 / cfg->min_distance) : 0.0f;
        P.d2_hi = first_float_where([&](float s) { return (double)sqrtf(s) >= cfg->min_distance; });
        P.d2_lo = first_float_where([&](float s) { return (double)sqrtf(s) > cfg->min_distance_eps; });
        P.collcap = collcap; P.collstride = block;
    }
    P.key0 = (uint32_t)key; P.key1 = (uint32_t)(key >> 32);
    P.px = e->px.as<float>(); P.py = e->py.as<float>(); P.pz = e->pz.as<float>();
    P.ea = e->ea.as<int>(); P.eb = e->eb.as<int>(); P.et = e->et.as<double>(); P.comm = e->comm.as<ClusterComm>(); P.ebuf = e->ebuf.as<int>(); P.ecount = e->ecount.as<int>();
    P.kind = e->kind.as<uint8_t>(); P.value = e->value.as<double>();
    P.origin_idx = e->origin.as<int>(); P.terminal_idx = e->terminal.as<int>();
    P.rowend = e->rowend.as<int>(); P.adj = e->adj.as<int>(); P.adj4 = e->adj4.as<int4>(); P.cost = e->cost.as<double>();
    P.temps = e->temps.as<double>(); P.slot_of_rep = e->slot_of_rep.as<int>();
    P.cells = e->cells.as<int>(); P.sorted = e->sorted.as<int>(); P.queue = e->queue.as<int>();
    P.bitmap = e->bitmap.as<unsigned>(); P.collbuf = e->collbuf.as<float>();
    P.stats = e->stats.as<unsigned long long>(); P.status = e->status.as<unsigned>();
    P.terms = nullptr;
    if (e->delta_mode) {
        P.dacc = e->dacc.as<fx128>(); P.dreach = e->dreach.as<int>(); P.dvis = e->dvis.as<unsigned>();
        P.dparent = e->dparent.as<unsigned short>(); P.dparscr = e->dparscr.as<unsigned short>();
        P.epk = e->epk.as<unsigned>(); P.epk_stride = e->ept * block;
        P.binc = e->delta_lists ? e->binc.as<unsigned>() : nullptr;
        P.walk_members = e->walk[0]; P.walk_entering = e->walk[1]; P.walk_rounds = e->walk[2];
        P.dlabel = e->dlabel.as<unsigned>(); P.dlabscr = e->dlabscr.as<unsigned>();
    }

    if (e->big_mode) {
        P.dacc = e->dacc.as<fx128>(); P.dreach = e->dreach.as<int>(); P.dvis = e->dvis.as<unsigned>();
        P.dlabel = e->dlabel.as<unsigned>(); P.dlabscr = e->dlabscr.as<unsigned>();
        P.dparent32 = e->dparent32.as<unsigned>(); P.dparscr32 = e->dparscr32.as<unsigned>();
        P.bvis = e->bvis.as<unsigned>(); P.bjmp = e->bjmp.as<unsigned>(); P.bdist = e->bdist.as<float>();
        P.CB = e->CB; P.R = e->R; P.bcomm = e->bcomm.as<BigComm>();
        if (e->CB > 1) {
            P.binc = e->binc.as<unsigned>(); P.bdeg = e->bdeg.as<int>();
            long long runs = (long long)((e->bigc_lds_bytes - ((sizeof(Shared) + 15) & ~(size_t)15)) / 16);
            if (const char *v = std::getenv("PROCGEN_PT_BIGC_CACHE_RUNS")) runs = std::min(runs, std::max(0ll, std::atoll(v)));     // test hook
            P.bigc_cache_runs = (int)runs;
        }
    }

    SwapParams &S = e->SP;
    std::memset(&S, 0, sizeof S);
    S.G = e->G; S.Tl = Tl; S.group_base = cfg->group_base; S.rng_mode = cfg->rng_mode;
    S.key0 = P.key0; S.key1 = P.key1;
    S.temps = P.temps; S.slot_of_rep = e->slot_of_rep.as<int>(); S.rep_of_slot = e->rep_of_slot.as<int>();
    S.swap_stats = e->swap_stats.as<unsigned long long>(); S.status = P.status;
    *out = e;
    return PT_OK;
}

void pt_graph_destroy(pt_graph_engine *e)
{
    if (!e) return;
    DeviceGuard guard(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    delete e;
}

int pt_graph_upload(pt_graph_engine *e, const float *positions, const int32_t *edges, int32_t edge_stride,
                    const int32_t *edge_counts, const uint8_t *node_kind, const double *node_value,
                    const double *temperatures)
{
    pt::clear_error();
    if (!e || !positions || !temperatures) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine, positions or temperatures");
    if (!edge_counts && e->Ecap > 0 && edges) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "edges given without edge_counts");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    const int N = e->N, R = e->R, G = e->G, T = e->T, Tl = e->Tl, Ecap = e->Ecap;

    for (int i = 0; i < G * Tl; i++)
        if (!(temperatures[i] > 0.0) || !std::isfinite(temperatures[i]))
            return pt::set_error(PT_ERR_INVALID_ARGUMENT, "temperatures must be positive and finite");   // reference GPU.Wrappers.fs:278-279

    std::vector<float> sx((size_t)R * N), sy((size_t)R * N), sz((size_t)R * N);
    for (size_t i = 0; i < (size_t)R * N; i++) { sx[i] = positions[3 * i]; sy[i] = positions[3 * i + 1]; sz[i] = positions[3 * i + 2]; }
    const int C = e->C;
    std::vector<int> ha((size_t)R * C * 2 * Ecap, 0), hb((size_t)R * C * 2 * Ecap, 0), hcount(R, 0), hbuf(R, 0);
    for (int r = 0; r < R; r++) {
        int ec = (edges && edge_counts) ? edge_counts[r] : 0;
        if (ec < 0 || ec > Ecap || ec > edge_stride) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "edge count exceeds edge_capacity or edge_stride");
        hcount[r] = ec;
        const int32_t *src = edges + (size_t)r * edge_stride * 2;
        for (int k = 0; k < ec; k++) {
            int a = src[2 * k], b = src[2 * k + 1];
            if (a < 0 || b < 0 || a >= N || b >= N) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "edge index out of bounds");  // reference GPU.Wrappers.fs:253-254
            // list index k (0 = head) -> array index ec-1-k, in every cluster member's copy
            for (int c = 0; c < C; c++) {
                ha[((size_t)r * C + c) * 2 * Ecap + (ec - 1 - k)] = a;
                hb[((size_t)r * C + c) * 2 * Ecap + (ec - 1 - k)] = b;
            }
        }
    }
    std::vector<uint8_t> hk((size_t)G * N);
    std::vector<double> hv((size_t)G * N, 0.0);
    std::vector<int> horigin(G, -1), hterm(G, -1);
    for (int g = 0; g < G; g++) {
        for (int i = 0; i < N; i++) {
            uint8_t k = node_kind ? node_kind[(size_t)g * N + i]
                                  : (uint8_t)(i == 0 ? PT_NODE_ORIGIN : (i == N - 1 ? PT_NODE_TERMINAL : PT_NODE_STANDARD));
            if (k > PT_NODE_STANDARD) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "unknown node kind");
            hk[(size_t)g * N + i] = k;
            if (node_value) hv[(size_t)g * N + i] = node_value[(size_t)g * N + i];
            // first Origin / first Terminal (reference ProceduralGeneration.fs:50)
            if (k == PT_NODE_ORIGIN && horigin[g] < 0) horigin[g] = i;
            if (k == PT_NODE_TERMINAL && hterm[g] < 0) hterm[g] = i;
        }
    }
    std::vector<int> sor((size_t)G * Tl), ros((size_t)G * Tl);
    for (int g = 0; g < G; g++) for (int s = 0; s < Tl; s++) { sor[(size_t)g * Tl + s] = s; ros[(size_t)g * Tl + s] = s; }

    hipStream_t st = e->stream;
    PT_HIP_TRY(hipStreamSynchronize(st));
    PT_HIP_TRY(hipMemcpy(e->px.p, sx.data(), sizeof(float) * sx.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->py.p, sy.data(), sizeof(float) * sy.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->pz.p, sz.data(), sizeof(float) * sz.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->ea.p, ha.data(), sizeof(int) * ha.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->eb.p, hb.data(), sizeof(int) * hb.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->ebuf.p, hbuf.data(), sizeof(int) * R, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->ecount.p, hcount.data(), sizeof(int) * R, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->kind.p, hk.data(), hk.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->value.p, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->origin.p, horigin.data(), sizeof(int) * G, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->terminal.p, hterm.data(), sizeof(int) * G, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->temps.p, temperatures, sizeof(double) * (size_t)G * Tl, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->slot_of_rep.p, sor.data(), sizeof(int) * sor.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemcpy(e->rep_of_slot.p, ros.data(), sizeof(int) * ros.size(), hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemset(e->stats.p, 0, e->stats.bytes));
    PT_HIP_TRY(hipMemset(e->swap_stats.p, 0, e->swap_stats.bytes));
    PT_HIP_TRY(hipMemset(e->status.p, 0, sizeof(unsigned)));
    PT_HIP_TRY(hipMemset(e->comm.p, 0, e->comm.bytes));
    (void)T;

    e->step = 0;
    e->steps_since_init = 0;
    if (e->conv_window) { int rc = enable_convergence(e, e->conv_window); if (rc) return rc; }
    { int rc = launch_init(e, e->P, 1); if (rc) return rc; }
    PT_HIP_TRY(hipStreamSynchronize(st));
    e->uploaded = true;
    e->stepped = false;
    return device_status(e->status.p);      // an initial cost outside the exact sums' range
}

int pt_graph_set_draws(pt_graph_engine *e, const double *chain_u, int64_t steps, const double *swap_u, int64_t passes)
{
    pt::clear_error();
    if (!e || steps < 0 || passes < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "bad draws arguments");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    size_t nc = (size_t)e->G * e->Tl * (size_t)steps * PT_DRAWS_PER_STEP;
    size_t ns = (size_t)e->G * (size_t)passes * (size_t)std::max(e->Tl - 1, 0);
    PT_HIP_TRY(e->chain_u.alloc(sizeof(double) * nc));
    PT_HIP_TRY(e->swap_u.alloc(sizeof(double) * ns));
    if (nc) PT_HIP_TRY(hipMemcpy(e->chain_u.p, chain_u, sizeof(double) * nc, hipMemcpyHostToDevice));
    if (ns) PT_HIP_TRY(hipMemcpy(e->swap_u.p, swap_u, sizeof(double) * ns, hipMemcpyHostToDevice));
    e->P.chain_u = e->chain_u.as<double>(); e->P.draw_steps = steps;
    e->SP.swap_u = e->swap_u.as<double>(); e->SP.draw_passes = passes;
    return PT_OK;
}

int pt_graph_set_ladder_keys(pt_graph_engine *e, const uint64_t *keys)
{
    pt::clear_error();
    if (!e || !keys) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine or keys");
    std::lock_guard<std::mutex> lock(e->mu);
    if (e->stepped) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "ladder keys must be set before the first step of a run (upload again to start a new one)");
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    PT_HIP_TRY(e->group_keys.alloc(sizeof(uint64_t) * (size_t)e->G));
    PT_HIP_TRY(hipMemcpy(e->group_keys.p, keys, sizeof(uint64_t) * (size_t)e->G, hipMemcpyHostToDevice));
    e->P.group_keys = e->group_keys.as<unsigned long long>();
    e->SP.group_keys = e->P.group_keys;
    return PT_OK;
}

int pt_graph_enable_trace(pt_graph_engine *e, int64_t max_steps, int64_t max_passes)
{
    pt::clear_error();
    if (!e || max_steps < 0 || max_passes < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "bad trace arguments");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    size_t nt = (size_t)max_steps * e->G * e->Tl;
    size_t ns = (size_t)max_passes * e->G * (size_t)std::max(e->Tl - 1, 0);
    PT_HIP_TRY(e->trace.alloc(sizeof(pt_trace_rec) * nt));
    PT_HIP_TRY(e->swap_trace.alloc(ns));
    PT_HIP_TRY(hipMemset(e->trace.p, 0, e->trace.bytes));
    PT_HIP_TRY(hipMemset(e->swap_trace.p, 0, e->swap_trace.bytes));
    e->trace_steps = max_steps; e->trace_passes = max_passes;
    e->P.trace = max_steps ? e->trace.as<pt_trace_rec>() : nullptr; e->P.trace_steps = max_steps;
    e->SP.swap_trace = max_passes ? e->swap_trace.as<uint8_t>() : nullptr; e->SP.trace_passes = max_passes;
    return PT_OK;
}

int pt_graph_read_trace(pt_graph_engine *e, pt_trace_rec *steps_out, int64_t n_steps, uint8_t *swaps_out, int64_t n_passes)
{
    pt::clear_error();
    if (!e) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine");
    if (n_steps > e->trace_steps || n_passes > e->trace_passes) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "trace request exceeds enabled size");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    if (steps_out && n_steps)
        PT_HIP_TRY(hipMemcpy(steps_out, e->trace.p, sizeof(pt_trace_rec) * (size_t)n_steps * e->G * e->Tl, hipMemcpyDeviceToHost));
    if (swaps_out && n_passes && e->Tl > 1)
        PT_HIP_TRY(hipMemcpy(swaps_out, e->swap_trace.p, (size_t)n_passes * e->G * (e->Tl - 1), hipMemcpyDeviceToHost));
    return PT_OK;
}

// Steps [step, step+n) of the loop at reference src/ParallelTempering.Core.fs:162-177: the swap
// pass follows every step whose index is a multiple of f.
int pt_graph_run(pt_graph_engine *e, int64_t steps)
{
    pt::clear_error();
    if (!e || steps < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine or negative steps");
    if (!e->uploaded) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "pt_graph_run before pt_graph_upload");
    if (e->Tl != e->T) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "split ladder: use pt_graph_run_to_swap_point / pt_graph_swap_pass");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    return run_steps(e, steps);
}

// Set the index of the next step (the swap schedule and the draw addresses follow it).
int pt_graph_set_step(pt_graph_engine *e, int64_t step)
{
    pt::clear_error();
    if (!e || step < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine or negative step");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    PT_HIP_TRY(hipMemset(e->comm.p, 0, e->comm.bytes));     // hand-off tags restart with the step counter
    e->step = step;
    if (e->conv_window) return enable_convergence(e, e->conv_window);   // the history does not span the jump
    return PT_OK;
}

int pt_graph_enable_convergence(pt_graph_engine *e, int32_t window_size)
{
    pt::clear_error();
    if (!e || window_size < 1) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine or window_size < 1");
    if (e->Tl != e->T) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "convergence monitoring needs the full ladder in one engine");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    return enable_convergence(e, window_size);
}

int pt_graph_check_convergence(pt_graph_engine *e, double strictness, int32_t iteration, pt_convergence_metrics *out)
{
    pt::clear_error();
    if (!e || !out) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (!e->conv_window) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "pt_graph_check_convergence before pt_graph_enable_convergence");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    return check_convergence(e, strictness, iteration, out);
}

int pt_graph_run_until_converged(pt_graph_engine *e, int64_t max_iterations, int32_t check_freq, double strictness,
                                 int64_t *steps_done, pt_convergence_metrics *out)
{
    pt::clear_error();
    if (!e || !out || max_iterations < 0 || check_freq < 1)
        return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument, negative max_iterations or check_freq < 1");
    if (!e->uploaded) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "run before upload");
    if (e->Tl != e->T) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "convergence monitoring needs the full ladder in one engine");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    if (!e->conv_window) {
        // reference src/ParallelTempering.Core.fs:512
        const long long w = std::max<long long>(200, max_iterations / 10);
        if (w > 0x3fffffff) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "history window too large");
        int rc = enable_convergence(e, (int)w);
        if (rc) return rc;
    }
    const long long start = e->step, half = e->conv_window / 2;
    bool stopped = false;
    while (e->step < max_iterations && !stopped) {
        // first step index c >= step with c % check_freq == 0 and c > window/2 (:539)
        long long c = std::max(e->step, half + 1);
        c = (c + check_freq - 1) / check_freq * check_freq;
        const long long last = std::min<long long>(c, max_iterations - 1);
        int rc = run_steps(e, last - e->step + 1);
        if (rc) return rc;
        if (last == c) {
            rc = check_convergence(e, strictness, (int)c, out);
            if (rc) return rc;
            stopped = true;
            for (int g = 0; g < e->G; g++) stopped = stopped && out[g].has_converged;
        }
    }
    if (!stopped) {     // :516-519
        int rc = check_convergence(e, strictness, (int)std::max<long long>(e->step, max_iterations), out);
        if (rc) return rc;
    }
    if (steps_done) *steps_done = e->step - start;
    return PT_OK;
}

int pt_graph_run_to_swap_point(pt_graph_engine *e, int64_t max_steps, int64_t *steps_done, int32_t *swap_due)
{
    pt::clear_error();
    if (!e || max_steps < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine or negative steps");
    if (!e->uploaded) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "run before upload");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    long long s = e->step;
    const long long f = e->f;
    long long next_swap_step = ((s + f - 1) / f) * f;
    long long last = std::min(s + max_steps - 1, next_swap_step);
    int n = (int)(last - s + 1);
    if (n > 0) { int rc = launch_steps(e, s, n); if (rc) return rc; e->step = last + 1; }
    if (steps_done) *steps_done = n > 0 ? n : 0;
    if (swap_due) *swap_due = (n > 0 && last == next_swap_step) ? 1 : 0;
    return PT_OK;
}

int pt_graph_local_energy_ptr(pt_graph_engine *e, double **d_energy_local, int32_t *count)
{
    pt::clear_error();
    if (!e || !d_energy_local) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    *d_energy_local = e->cost.as<double>();
    if (count) *count = e->R;
    return PT_OK;
}

int pt_graph_read_energies(pt_graph_engine *e, double *dst)
{
    pt::clear_error();
    if (!e || !dst) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipMemcpyAsync(dst, e->cost.p, sizeof(double) * (size_t)e->R, hipMemcpyDefault, e->stream));
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    return device_status(e->status.p);
}

int pt_graph_swap_pass(pt_graph_engine *e, const double *d_energy_all_replicas)
{
    pt::clear_error();
    if (!e || !d_energy_all_replicas) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    if (e->step <= 0 || (e->step - 1) % e->f != 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "no swap pass is due at this step");
    if (e->Tl < 2) return PT_OK;
    return launch_swap(e, (e->step - 1) / e->f, d_energy_all_replicas);
}

int pt_graph_sync(pt_graph_engine *e)
{
    pt::clear_error();
    if (!e) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine");
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    return device_status(e->status.p);
}

int pt_graph_replica_slots(pt_graph_engine *e, int32_t *slot_of_replica_all)
{
    pt::clear_error();
    if (!e || !slot_of_replica_all) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    PT_HIP_TRY(hipMemcpy(slot_of_replica_all, e->slot_of_rep.p, sizeof(int) * (size_t)e->G * e->Tl, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Output row of local replica (g, t): its slot minus the engine's first slot when the whole
// ladder is local, else its replica position (a split ladder's caller reorders by slot itself).
static int output_row(const pt_graph_engine *e, const std::vector<int> &sor, int g, int t)
{
    if (e->Tl == e->T) return g * e->T + sor[(size_t)g * e->Tl + t];
    return g * e->T + t;
}

int pt_graph_download(pt_graph_engine *e, float *positions, int32_t *edges, int32_t edge_stride,
                      int32_t *edge_counts, double *costs)
{
    pt::clear_error();
    if (!e) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    const int N = e->N, R = e->R, G = e->G, T = e->T, Ecap = e->Ecap;
    std::vector<int> sor((size_t)G * e->Tl);
    PT_HIP_TRY(hipMemcpy(sor.data(), e->slot_of_rep.p, sizeof(int) * sor.size(), hipMemcpyDeviceToHost));
    if (positions) {
        std::vector<float> sx((size_t)R * N), sy((size_t)R * N), sz((size_t)R * N);
        PT_HIP_TRY(hipMemcpy(sx.data(), e->px.p, sizeof(float) * sx.size(), hipMemcpyDeviceToHost));
        PT_HIP_TRY(hipMemcpy(sy.data(), e->py.p, sizeof(float) * sy.size(), hipMemcpyDeviceToHost));
        PT_HIP_TRY(hipMemcpy(sz.data(), e->pz.p, sizeof(float) * sz.size(), hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++) for (int t = 0; t < T; t++) {
            size_t src = ((size_t)g * T + t) * N, dst = (size_t)output_row(e, sor, g, t) * N;
            for (int i = 0; i < N; i++) {
                positions[3 * (dst + i)] = sx[src + i]; positions[3 * (dst + i) + 1] = sy[src + i]; positions[3 * (dst + i) + 2] = sz[src + i];
            }
        }
    }
    std::vector<int> hcount(R);
    PT_HIP_TRY(hipMemcpy(hcount.data(), e->ecount.p, sizeof(int) * R, hipMemcpyDeviceToHost));
    if (edge_counts)
        for (int g = 0; g < G; g++) for (int t = 0; t < T; t++) edge_counts[output_row(e, sor, g, t)] = hcount[g * T + t];
    if (edges) {
        const int C = e->C;
        std::vector<int> ha((size_t)R * C * 2 * Ecap), hb((size_t)R * C * 2 * Ecap), hbuf(R);
        PT_HIP_TRY(hipMemcpy(ha.data(), e->ea.p, sizeof(int) * ha.size(), hipMemcpyDeviceToHost));
        PT_HIP_TRY(hipMemcpy(hb.data(), e->eb.p, sizeof(int) * hb.size(), hipMemcpyDeviceToHost));
        PT_HIP_TRY(hipMemcpy(hbuf.data(), e->ebuf.p, sizeof(int) * R, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++) for (int t = 0; t < T; t++) {
            int r = g * T + t, ec = hcount[r];
            if (ec > edge_stride) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "edge_stride smaller than a replica's edge count");
            size_t src = (((size_t)r * C) * 2 + hbuf[r]) * Ecap;      // member 0's copy
            int32_t *dst = edges + (size_t)output_row(e, sor, g, t) * edge_stride * 2;
            for (int k = 0; k < ec; k++) { dst[2 * k] = ha[src + (ec - 1 - k)]; dst[2 * k + 1] = hb[src + (ec - 1 - k)]; }
        }
    }
    if (costs) {
        std::vector<double> hc(R);
        PT_HIP_TRY(hipMemcpy(hc.data(), e->cost.p, sizeof(double) * R, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; g++) for (int t = 0; t < T; t++) costs[output_row(e, sor, g, t)] = hc[g * T + t];
    }
    return device_status(e->status.p);      // the arrays are filled either way; the code says whether to trust them
}

int pt_graph_cost_terms(pt_graph_engine *e, double *terms)
{
    pt::clear_error();
    if (!e || !terms) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    if (!e->uploaded) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "cost_terms before upload");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    GraphParams P = e->P;
    P.terms = e->terms.as<double>();
    { int rc = launch_init(e, P, 0); if (rc) return rc; }
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<double> h((size_t)e->R * 8);
    std::vector<int> sor((size_t)e->G * e->Tl);
    PT_HIP_TRY(hipMemcpy(h.data(), e->terms.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(sor.data(), e->slot_of_rep.p, sizeof(int) * sor.size(), hipMemcpyDeviceToHost));
    for (int g = 0; g < e->G; g++) for (int t = 0; t < e->T; t++)
        std::memcpy(terms + (size_t)output_row(e, sor, g, t) * 8, h.data() + ((size_t)g * e->T + t) * 8, sizeof(double) * 8);
    return PT_OK;
}

int pt_graph_get_stats(pt_graph_engine *e, pt_graph_stats *out)
{
    pt::clear_error();
    if (!e || !out) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<unsigned long long> st((size_t)e->R * 4), sw((size_t)e->G * 2);
    unsigned status = 0;
    PT_HIP_TRY(hipMemcpy(st.data(), e->stats.p, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(sw.data(), e->swap_stats.p, sizeof(unsigned long long) * sw.size(), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(&status, e->status.p, sizeof status, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->steps = e->step;
    unsigned long long early = 0;
    for (int r = 0; r < e->R; r++) {
        out->proposals += (int64_t)st[(size_t)r * 4]; out->accepted += (int64_t)st[(size_t)r * 4 + 1];
        early += st[(size_t)r * 4 + 2];
    }
    out->early_rejects = (uint32_t)std::min<unsigned long long>(early, 0xffffffffull);
    for (int g = 0; g < e->G; g++) { out->swap_attempts += (int64_t)sw[(size_t)g * 2]; out->swaps += (int64_t)sw[(size_t)g * 2 + 1]; }
    out->status_bits = status;
    return PT_OK;
}

int pt_graph_kernel_time(pt_graph_engine *e, double *ms, int64_t *launches, int32_t reset)
{
    pt::clear_error();
    if (!e) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    PT_HIP_TRY(hipStreamSynchronize(e->stream));
    int rc = drain_events(e);
    if (rc) return rc;
    if (ms) *ms = e->timed_ms;
    if (launches) *launches = e->timed_launches;
    if (reset) { e->timed_ms = 0.0; e->timed_launches = 0; }
    return PT_OK;
}

void *pt_graph_stream(pt_graph_engine *e) { return e ? (void *)e->stream : nullptr; }

int pt_graph_info(pt_graph_engine *e, pt_graph_info_rec *out)
{
    pt::clear_error();
    if (!e || !out) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(out, 0, sizeof *out);
    out->incremental = (e->delta_mode || e->big_mode) ? 1 : 0;
    out->lds_placement = (e->delta_mode || e->lds_mode) ? 1 : 0;
    out->cluster_size = e->big_mode ? e->CB : e->C;
    out->block_threads = e->block;
    out->lds_bytes = (int32_t)(e->delta_mode ? e->delta_lds_bytes : e->lds_bytes);
    out->swap_frequency = e->f;
    return PT_OK;
}

}  // extern "C"

// ---- the drop-in entry points' iteration on resident chains ----------------------------------------------
// The unchanged host calls parallel_tempering_step once per iteration and passes the array the previous
// call filled (reference src/GPU.Wrappers.fs:314-323).  Uploading it, re-evaluating every chain's cost and
// downloading everything per call is what the call pattern suggests and costs milliseconds; instead the
// chains stay resident, the caller's array is copied to the device as it is (no host-side conversion), one
// kernel checks that it still equals the resident chains (slot order, after the float conversion a fresh
// upload would apply) and only a caller that edited its positions pays for a reload and a fresh cost
// evaluation.  Everything the iteration does is addressed by temperature slot, so resident chains in
// permuted order give the same bits as freshly uploaded ones in slot order.
// The usual caller does not touch the array at all between two calls: the library keeps a host copy of what it
// left there, and when the array still equals that copy byte for byte (a threaded memcmp, a tenth of the time the
// upload takes) nothing is uploaded or checked on the device at all.
namespace {

bool legacy_same_bytes(const void *a, const void *b, size_t bytes)
{
    const unsigned char *x = (const unsigned char *)a, *y = (const unsigned char *)b;
    const int parts = bytes >= (4u << 20) ? 4 : 1;
    if (parts == 1) return std::memcmp(x, y, bytes) == 0;
    const size_t per = (bytes / parts + 63) & ~(size_t)63;
    int differ[4] = {0, 0, 0, 0};
    std::thread th[3];
    auto part = [&](int p) {
        const size_t lo = std::min(bytes, per * (size_t)p), hi = (p == parts - 1) ? bytes : std::min(bytes, lo + per);
        differ[p] = (hi > lo && std::memcmp(x + lo, y + lo, hi - lo) != 0) ? 1 : 0;
    };
    for (int p = 1; p < parts; p++) th[p - 1] = std::thread(part, p);
    part(0);
    for (int p = 1; p < parts; p++) th[p - 1].join();
    return !(differ[0] | differ[1] | differ[2] | differ[3]);
}

template <class Real>
__global__ void legacy_compare_kernel(const Real *in, const float *px, const float *py, const float *pz, const int *rep_of_slot,
                                      int T, int N, int *mismatch)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)T * N) return;
    const int s = (int)(k / N), i = (int)(k - (long long)s * N);
    const size_t at = (size_t)rep_of_slot[s] * N + i;
    // bit patterns: a NaN never equals the resident value and so forces the reload that reports it
    const bool same = __float_as_uint((float)in[3 * k]) == __float_as_uint(px[at]) &&
                      __float_as_uint((float)in[3 * k + 1]) == __float_as_uint(py[at]) &&
                      __float_as_uint((float)in[3 * k + 2]) == __float_as_uint(pz[at]);
    if (!same) *mismatch = 1;
}

template <class Real>
__global__ void legacy_load_kernel(const Real *in, float *px, float *py, float *pz, int T, int N)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)T * N) return;
    px[k] = (float)in[3 * k]; py[k] = (float)in[3 * k + 1]; pz[k] = (float)in[3 * k + 2];     // chain t = replica t = slot t
}

// The chains in slot order in the caller's width, plus the list of the nodes whose three values differ (as
// bits) from what the caller's array holds now: after an ordinary iteration that is at most one node per
// chain, and only those need to travel back.  changed[0] counts them (entries beyond `cap` are dropped:
// the host then takes the whole array).  The device copy of the caller's array takes the new values too, so
// that it still mirrors the caller's array at the next call.
template <class Real> struct LegacyChange { long long k; Real x, y, z; };
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }
template <class Real>
__global__ void legacy_store_kernel(Real *out, Real *cost_out, Real *in, const float *px, const float *py, const float *pz,
                                    const double *cost, const int *rep_of_slot, int T, int N, int *changed,
                                    LegacyChange<Real> *list, int cap)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (long long)T * N) return;
    const int s = (int)(k / N), i = (int)(k - (long long)s * N);
    const size_t at = (size_t)rep_of_slot[s] * N + i;
    const Real x = (Real)px[at], y = (Real)py[at], z = (Real)pz[at];
    out[3 * k] = x; out[3 * k + 1] = y; out[3 * k + 2] = z;
    if (i == 0) cost_out[s] = (Real)cost[rep_of_slot[s]];
    if (!same_bits(x, in[3 * k]) || !same_bits(y, in[3 * k + 1]) || !same_bits(z, in[3 * k + 2])) {
        const int slot = atomicAdd(changed, 1);
        if (slot < cap) { list[slot].k = k; list[slot].x = x; list[slot].y = y; list[slot].z = z; }
        in[3 * k] = x; in[3 * k + 1] = y; in[3 * k + 2] = z;         // `in` follows the caller's array
    }
}

template <class Real>
int legacy_iteration(pt_graph_engine *e, Real *positions, Real *costs, long long call_index, bool positions_are_new)
{
    const int T = e->T, N = e->N;
    const size_t cells = (size_t)T * N, bytes = sizeof(Real) * 3 * cells;
    hipStream_t st = e->stream;
    if (e->legacy_in.bytes < bytes) PT_HIP_TRY(e->legacy_in.alloc(bytes));
    if (e->legacy_out.bytes < bytes + sizeof(Real) * (size_t)T) PT_HIP_TRY(e->legacy_out.alloc(bytes + sizeof(Real) * (size_t)T));
    constexpr int CAP = 4096;               // changed nodes sent back one by one; more than that and the whole array goes
    if (!e->legacy_flag.p) PT_HIP_TRY(e->legacy_flag.alloc(2 * sizeof(int) + sizeof(LegacyChange<double>) * CAP));
    int *d_changed = e->legacy_flag.as<int>() + 1;
    LegacyChange<Real> *d_list = (LegacyChange<Real> *)(e->legacy_flag.as<int>() + 2);
    const unsigned threads = 256, blocks = (unsigned)((cells + threads - 1) / threads);
    Real *d_in = e->legacy_in.as<Real>(), *d_out = e->legacy_out.as<Real>();
    // untouched since the last call (then the device copy `d_in` mirrors it as well)?
    const bool untouched = !positions_are_new && e->legacy_shadow_valid && e->legacy_shadow.size() == bytes &&
                           legacy_same_bytes(positions, e->legacy_shadow.data(), bytes);
    e->legacy_shadow_valid = false;         // until this call has ended well
    if (!untouched) PT_HIP_TRY(hipMemcpyAsync(d_in, positions, bytes, hipMemcpyHostToDevice, st));
    int reload = positions_are_new ? 1 : 0;
    if (!reload && !untouched) {
        PT_HIP_TRY(hipMemsetAsync(e->legacy_flag.p, 0, sizeof(int), st));
        hipLaunchKernelGGL(legacy_compare_kernel<Real>, dim3(blocks), dim3(threads), 0, st, d_in, e->px.as<float>(), e->py.as<float>(),
                           e->pz.as<float>(), e->rep_of_slot.as<int>(), T, N, e->legacy_flag.as<int>());
        PT_HIP_TRY(hipGetLastError());
        PT_HIP_TRY(hipMemcpyAsync(&reload, e->legacy_flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        PT_HIP_TRY(hipStreamSynchronize(st));
    }
    if (reload) {
        // the caller's chains, as a fresh upload would place them: chain t in replica t in slot t, costs from scratch
        std::vector<int> ident((size_t)T);
        for (int t = 0; t < T; t++) ident[(size_t)t] = t;
        hipLaunchKernelGGL(legacy_load_kernel<Real>, dim3(blocks), dim3(threads), 0, st, d_in, e->px.as<float>(), e->py.as<float>(),
                           e->pz.as<float>(), T, N);
        PT_HIP_TRY(hipGetLastError());
        PT_HIP_TRY(hipMemcpyAsync(e->slot_of_rep.p, ident.data(), sizeof(int) * (size_t)T, hipMemcpyHostToDevice, st));
        PT_HIP_TRY(hipMemcpyAsync(e->rep_of_slot.p, ident.data(), sizeof(int) * (size_t)T, hipMemcpyHostToDevice, st));
        PT_HIP_TRY(hipStreamSynchronize(st));           // `ident` leaves scope
        int rc = launch_init(e, e->P, 1);
        if (rc) return rc;
        e->steps_since_init = 0;
    }
    e->step = call_index;
    { int rc = run_steps(e, 1); if (rc) return rc; }
    PT_HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof(int), st));
    hipLaunchKernelGGL(legacy_store_kernel<Real>, dim3(blocks), dim3(threads), 0, st, d_out, d_out + 3 * cells, d_in,
                       e->px.as<float>(), e->py.as<float>(), e->pz.as<float>(), e->cost.as<double>(), e->rep_of_slot.as<int>(), T, N,
                       d_changed, d_list, CAP);
    PT_HIP_TRY(hipGetLastError());
    // the outputs are written only once the device reports no failure: first the costs, the count and the status ...
    std::vector<Real> hc((size_t)T);
    int changed = 0;
    PT_HIP_TRY(hipMemcpyAsync(hc.data(), d_out + 3 * cells, sizeof(Real) * (size_t)T, hipMemcpyDeviceToHost, st));
    PT_HIP_TRY(hipMemcpyAsync(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, st));
    PT_HIP_TRY(hipStreamSynchronize(st));
    { int rc = device_status(e->status.p); if (rc) return rc; }
    // ... then the positions: the few nodes that changed, or (first call of a job, swap passes) the whole array
    if (changed <= CAP) {
        std::vector<LegacyChange<Real>> list((size_t)changed);
        if (changed) {
            PT_HIP_TRY(hipMemcpyAsync(list.data(), d_list, sizeof(LegacyChange<Real>) * (size_t)changed, hipMemcpyDeviceToHost, st));
            PT_HIP_TRY(hipStreamSynchronize(st));
        }
        for (const LegacyChange<Real> &c : list) { positions[3 * c.k] = c.x; positions[3 * c.k + 1] = c.y; positions[3 * c.k + 2] = c.z; }
        if (untouched) {                    // the host copy takes the same few values
            Real *sh = (Real *)e->legacy_shadow.data();
            for (const LegacyChange<Real> &c : list) { sh[3 * c.k] = c.x; sh[3 * c.k + 1] = c.y; sh[3 * c.k + 2] = c.z; }
        }
    } else {
        PT_HIP_TRY(hipMemcpyAsync(positions, d_out, bytes, hipMemcpyDeviceToHost, st));
        PT_HIP_TRY(hipStreamSynchronize(st));
    }
    if (!untouched || changed > CAP) {      // the host copy is taken anew
        e->legacy_shadow.resize(bytes);
        std::memcpy(e->legacy_shadow.data(), positions, bytes);
    }
    e->legacy_shadow_valid = true;
    for (int t = 0; t < T; t++) costs[t] = hc[(size_t)t];
    return PT_OK;
}

}  // namespace

int pt_graph_legacy_iteration(pt_graph_engine *e, void *positions, void *costs, int real_bytes, long long call_index,
                              bool positions_are_new)
{
    if (!e || !e->uploaded || e->G != 1 || e->Tl != e->T) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "not a whole single ladder");
    std::lock_guard<std::mutex> lock(e->mu);
    DeviceGuard guard(e->device);
    if (real_bytes == 8) return legacy_iteration<double>(e, (double *)positions, (double *)costs, call_index, positions_are_new);
    return legacy_iteration<float>(e, (float *)positions, (float *)costs, call_index, positions_are_new);
}

extern "C" {

#if defined(PT_STAMPS) || defined(PT_STAMPS_LIGHT)
// diagnostic build only: read (and clear) the per-phase cycle totals, [4096][16]
int pt_debug_read_stamps(unsigned long long *out, int clear)
{
    PT_HIP_TRY(hipDeviceSynchronize());
    PT_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 4096 * PT_NSTAMP));
    if (clear) {
        static std::vector<unsigned long long> z(4096 * PT_NSTAMP, 0ull);
        PT_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z.data(), sizeof(unsigned long long) * z.size()));
    }
    return PT_OK;
}
#endif

// unit-test hook: on how many floats x with lo <= x < hi (both positive and normal) the kernels' square root is not
// the correctly rounded one
int pt_debug_sqrt_check(float lo, float hi, uint64_t *mismatches)
{
    pt::clear_error();
    if (!mismatches || !(lo > 0.0f) || !(hi >= lo) || !std::isfinite(hi)) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "bad arguments");
    if (pt::device_count() < 1) return pt::set_error(PT_ERR_NO_DEVICE, "no HIP device");
    unsigned blo, bhi;
    std::memcpy(&blo, &lo, 4); std::memcpy(&bhi, &hi, 4);
    DevBuf bad;
    PT_HIP_TRY(bad.alloc(sizeof(unsigned long long)));
    PT_HIP_TRY(hipMemset(bad.p, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(pt_debug_sqrt_kernel, dim3(4096), dim3(256), 0, 0, blo, bhi, bad.as<unsigned long long>());
    PT_HIP_TRY(hipGetLastError());
    PT_HIP_TRY(hipDeviceSynchronize());
    unsigned long long h = 0;
    PT_HIP_TRY(hipMemcpy(&h, bad.p, sizeof h, hipMemcpyDeviceToHost));
    *mismatches = (uint64_t)h;
    return PT_OK;
}

// unit-test hook: exact fx128 sum of n doubles on the device
int pt_debug_canonical_sum(const double *terms, int64_t n, double *out, int32_t *range_error)
{
    pt::clear_error();
    if (!terms || !out || n < 0) return pt::set_error(PT_ERR_INVALID_ARGUMENT, "bad arguments");
    if (pt::device_count() < 1) return pt::set_error(PT_ERR_NO_DEVICE, "no HIP device");
    DevBuf x, o, er;
    PT_HIP_TRY(x.alloc(sizeof(double) * (size_t)n)); PT_HIP_TRY(o.alloc(sizeof(double))); PT_HIP_TRY(er.alloc(sizeof(unsigned)));
    if (n) PT_HIP_TRY(hipMemcpy(x.p, terms, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    PT_HIP_TRY(hipMemset(er.p, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(pt_debug_sum_kernel, dim3(1), dim3(256), 0, 0, x.as<double>(), (long long)n, o.as<double>(), er.as<unsigned>());
    PT_HIP_TRY(hipGetLastError());
    PT_HIP_TRY(hipDeviceSynchronize());
    unsigned herr = 0;
    PT_HIP_TRY(hipMemcpy(out, o.p, sizeof(double), hipMemcpyDeviceToHost));
    PT_HIP_TRY(hipMemcpy(&herr, er.p, sizeof herr, hipMemcpyDeviceToHost));
    if (range_error) *range_error = (int32_t)herr;
    return PT_OK;
}

// Replaces the extern at reference src/GPU.Compute.fs:98-109; the statistics follow the host
// routine (src/ParallelTempering.Core.fs:442-492).
void check_convergence_async(const double *costHistory, const int32_t *swapHistory, int numChains, int historyLength,
                             int swapHistoryLength, double strictness, int iteration, ConvergenceMetrics *outMetrics,
                             void *stream)
{
    (void)stream;
    pt::clear_error();
    if (!outMetrics) { pt::set_error(PT_ERR_INVALID_ARGUMENT, "null outMetrics"); return; }
    // what the CUDA routine leaves when it has too little history (src/gpu/convergence.cu:293-300)
    outMetrics->GelmanRubin = 2.0f; outMetrics->EffectiveSampleSize = 1.0f; outMetrics->CostStabilization = 1.0f;
    outMetrics->AvgSwapAcceptance = 0.0f; outMetrics->Iteration = iteration; outMetrics->HasConverged = 0;
    if (numChains < 0 || historyLength < 0 || swapHistoryLength < 0 || (numChains > 0 && historyLength > 0 && !costHistory) ||
        (swapHistoryLength > 0 && !swapHistory)) {
        pt::set_error(PT_ERR_INVALID_ARGUMENT, "negative length or null history");
        return;
    }
    auto run = [&]() -> int {
        DevBuf hist, swaps, out;
        const size_t nh = (size_t)numChains * (size_t)historyLength;
        PT_HIP_TRY(hist.alloc(sizeof(double) * std::max<size_t>(nh, 1)));
        PT_HIP_TRY(swaps.alloc(std::max<size_t>((size_t)swapHistoryLength, 1)));
        PT_HIP_TRY(out.alloc(sizeof(pt_convergence_metrics)));
        if (nh) PT_HIP_TRY(hipMemcpy(hist.p, costHistory, sizeof(double) * nh, hipMemcpyHostToDevice));
        std::vector<uint8_t> sw((size_t)swapHistoryLength);
        for (int i = 0; i < swapHistoryLength; i++) sw[(size_t)i] = swapHistory[i] != 0;
        if (swapHistoryLength) PT_HIP_TRY(hipMemcpy(swaps.p, sw.data(), sw.size(), hipMemcpyHostToDevice));
        ConvParams cp;
        std::memset(&cp, 0, sizeof cp);
        cp.hist = hist.as<double>();
        cp.ring = std::max(historyLength, 1); cp.chain_stride = historyLength; cp.ladder_stride = 0;
        cp.first = 0; cp.n = historyLength; cp.chains = numChains; cp.window = historyLength; cp.ladders = 1;
        cp.swaps = swaps.as<uint8_t>(); cp.swap_total = nullptr; cp.swap_ring = std::max(swapHistoryLength, 1); cp.swap_n = swapHistoryLength;
        cp.strictness = strictness; cp.iteration = iteration;
        cp.out = out.as<pt_convergence_metrics>();
        int rc = launch_convergence(nullptr, cp);
        if (rc) return rc;
        pt_convergence_metrics m;
        PT_HIP_TRY(hipMemcpy(&m, out.p, sizeof m, hipMemcpyDeviceToHost));
        if (m.has_gelman_rubin) outMetrics->GelmanRubin = (float)m.gelman_rubin;
        if (m.has_ess) outMetrics->EffectiveSampleSize = (float)m.effective_sample_size;
        outMetrics->CostStabilization = (float)m.cost_stabilization;
        outMetrics->AvgSwapAcceptance = (float)m.swap_acceptance_rate;
        outMetrics->HasConverged = m.has_converged;
        return PT_OK;
    };
    (void)run();
}

}  // extern "C"

// ---- voxel-chunk engine (same translation unit: it shares the swap kernel and the host helpers) ----
#include "pt_chunk_kernels.hip.inc"
#include "pt_chunk_host.hip.inc"

// The entry points the unchanged F# host binds (reference src/GPU.Compute.fs:31-40),
// as thin wrappers over the batched engine.
#include "pt_host.h"
#include "../../include/procgen_pt.h"

#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

struct LegacyState {
    std::mutex mu;
    pt_graph_engine *engine = nullptr;
    int n = 0, e = 0, t = 0;
    long long calls = 0;
    std::vector<float> pos;
    std::vector<double> temps;
    std::vector<int32_t> ecount, edges;     // what the engine was last uploaded with
    bool uploaded = false;
    // Opt-in (PROCGEN_PT_PIN_CALLER_ARRAY=1): the caller's position array is page-locked once the same array has
    // come twice in a row (the host's loop passes one GC-pinned array for the whole job, reference
    // src/GPU.Wrappers.fs:312-323), so that whole-array copies run at the bus rate.  The registration outlives the
    // call, which is only safe when the caller keeps the array alive until parallel_tempering_reset() or until it
    // passes a different array; by default nothing of the caller's memory is retained after a call returns.
    void *seen_ptr = nullptr;
    size_t seen_bytes = 0;
    int seen_count = 0;
    bool registered = false;
};

LegacyState &state()
{
    static LegacyState s;
    return s;
}

void forget_array(LegacyState &s)
{
    if (s.registered) { if (hipHostUnregister(s.seen_ptr) != hipSuccess) (void)hipGetLastError(); }
    s.registered = false; s.seen_ptr = nullptr; s.seen_bytes = 0; s.seen_count = 0;
}

bool pin_caller_array()
{
    static const bool on = [] { const char *v = std::getenv("PROCGEN_PT_PIN_CALLER_ARRAY"); return v && std::atoi(v) != 0; }();
    return on;
}

void note_array(LegacyState &s, void *p, size_t bytes)
{
    if (!pin_caller_array()) return;
    if (p != s.seen_ptr || bytes != s.seen_bytes) { forget_array(s); s.seen_ptr = p; s.seen_bytes = bytes; }
    s.seen_count++;
    if (!s.registered && s.seen_count == 2) {
        if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) s.registered = true;
        else (void)hipGetLastError();           // not fatal: the copies just take the pageable path
    }
}

void drop(LegacyState &s)
{
    forget_array(s);
    if (s.engine) pt_graph_destroy(s.engine);
    s.engine = nullptr;
    s.n = s.e = s.t = 0;
    s.calls = 0;
    s.uploaded = false;
}

// One iteration of ParallelTempering.run's loop body (reference
// src/ParallelTempering.Core.fs:162-177) over the caller's arrays.
template <class Real>
void step_impl(Real *positions, int *edges, Real *costs, Real *temperatures, int nodeCount, int edgeCount, int tempCount)
{
    pt::clear_error();
    // availability probe (reference src/GPU.Wrappers.fs:112): nothing to compute and no GPU state touched.  The
    // ABI is void, so all the host can learn is through pt_last_error(): it says PT_ERR_NO_DEVICE when the HIP
    // runtime sees no usable GPU (asked once per process).
    if (nodeCount == 0 && edgeCount == 0 && tempCount == 0) {
        static const int devices = pt::device_count();
        if (devices < 1) pt::set_error(PT_ERR_NO_DEVICE, "no usable HIP device: every later call will fail");
        return;
    }
    if (nodeCount <= 0 || tempCount <= 0 || edgeCount < 0 || !positions || !costs || !temperatures || (edgeCount > 0 && !edges)) {
        pt::set_error(PT_ERR_INVALID_ARGUMENT, "parallel_tempering_step: bad counts or null arrays");
        return;
    }
    LegacyState &s = state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!s.engine || s.n != nodeCount || s.e != edgeCount || s.t != tempCount) {
        drop(s);
        pt_graph_config cfg;
        pt_graph_default_config(&cfg, nodeCount, edgeCount, tempCount, 1);
        cfg.mutation_mode = PT_MUT_PERTURB_ONLY;    // the shared topology is an input only at this boundary
        cfg.seed = "libgpu_compute/parallel_tempering_step";
        if (pt_graph_create(&cfg, &s.engine) != PT_OK) { s.engine = nullptr; return; }
        s.n = nodeCount; s.e = edgeCount; s.t = tempCount; s.calls = 0;
        s.pos.resize((size_t)tempCount * nodeCount * 3);
        s.temps.assign(tempCount, 0.0);
        s.ecount.assign(tempCount, edgeCount);
    }
    const size_t np = (size_t)tempCount * nodeCount * 3;
    // The topology and the ladder go to the device when they change (normally once per job); the chains
    // stay resident between calls and the engine itself checks that the caller's array still holds them.
    bool same_inputs = s.uploaded && s.edges.size() == (size_t)edgeCount * 2;
    for (int i = 0; same_inputs && i < tempCount; i++) same_inputs = s.temps[i] == (double)temperatures[i];
    for (size_t k = 0; same_inputs && k < s.edges.size(); k++) same_inputs = s.edges[k] == edges[k];
    if (!same_inputs) {
        s.uploaded = false;
        for (size_t i = 0; i < np; i++) s.pos[i] = (float)positions[i];
        for (int i = 0; i < tempCount; i++) s.temps[i] = (double)temperatures[i];
        s.edges.assign(edges, edges + (size_t)edgeCount * 2);
        // every chain shares graph 0's topology (reference src/GPU.Wrappers.fs:307-308)
        std::vector<int32_t> ed((size_t)tempCount * edgeCount * 2);
        for (int t = 0; t < tempCount; t++)
            for (int k = 0; k < edgeCount * 2; k++) ed[(size_t)t * edgeCount * 2 + k] = edges[k];
        if (pt_graph_upload(s.engine, s.pos.data(), ed.data(), edgeCount, s.ecount.data(), nullptr, nullptr, s.temps.data()) != PT_OK) {
            s.edges.clear();
            return;
        }
        s.uploaded = true;
    }
    note_array(s, positions, sizeof(Real) * np);
    // on failure (bad values found on the device, ...) the outputs stay untouched and the next call starts afresh
    if (pt_graph_legacy_iteration(s.engine, positions, costs, (int)sizeof(Real), s.calls, false) != PT_OK) { s.uploaded = false; return; }
    s.calls++;
}

}  // namespace

extern "C" {

void parallel_tempering_step(double *positions, int *edges, double *costs, double *temperatures,
                             int nodeCount, int edgeCount, int tempCount)
{
    step_impl<double>(positions, edges, costs, temperatures, nodeCount, edgeCount, tempCount);
}

void parallel_tempering_step_f32(float *positions, int *edges, float *costs, float *temperatures,
                                 int nodeCount, int edgeCount, int tempCount)
{
    step_impl<float>(positions, edges, costs, temperatures, nodeCount, edgeCount, tempCount);
}

void parallel_tempering_reset(void)
{
    LegacyState &s = state();
    {
        std::lock_guard<std::mutex> lock(s.mu);
        drop(s);
    }
    pt_collision_release();
}

}  // extern "C"

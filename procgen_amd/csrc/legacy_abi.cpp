// The entry points the unchanged F# host binds (reference src/GPU.Compute.fs:31-40),
// as thin wrappers over the batched engine.
#include "pt_host.h"
#include "../../include/procgen_pt.h"

#include <cmath>
#include <mutex>
#include <vector>

extern "C" int pt_graph_set_step(pt_graph_engine *e, int64_t step);

namespace {

struct LegacyState {
    std::mutex mu;
    pt_graph_engine *engine = nullptr;
    int n = 0, e = 0, t = 0;
    long long calls = 0;
    std::vector<float> pos;
    std::vector<double> temps, costs;
    std::vector<int32_t> ecount;
};

LegacyState &state()
{
    static LegacyState s;
    return s;
}

void drop(LegacyState &s)
{
    if (s.engine) pt_graph_destroy(s.engine);
    s.engine = nullptr;
    s.n = s.e = s.t = 0;
    s.calls = 0;
}

// One iteration of ParallelTempering.run's loop body (reference
// src/ParallelTempering.Core.fs:162-177) over the caller's arrays.
template <class Real>
void step_impl(Real *positions, int *edges, Real *costs, Real *temperatures, int nodeCount, int edgeCount, int tempCount)
{
    pt::clear_error();
    // availability probe (reference src/GPU.Wrappers.fs:112): nothing to do, touch nothing
    if (nodeCount == 0 && edgeCount == 0 && tempCount == 0) return;
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
        s.temps.resize(tempCount); s.costs.resize(tempCount);
        s.ecount.assign(tempCount, edgeCount);
    }
    const size_t np = (size_t)tempCount * nodeCount * 3;
    for (size_t i = 0; i < np; i++) s.pos[i] = (float)positions[i];
    for (int i = 0; i < tempCount; i++) s.temps[i] = (double)temperatures[i];
    // every chain shares graph 0's topology (reference src/GPU.Wrappers.fs:307-308): edge_stride 0 re-reads it
    std::vector<int32_t> ed((size_t)tempCount * edgeCount * 2);
    for (int t = 0; t < tempCount; t++)
        for (int k = 0; k < edgeCount * 2; k++) ed[(size_t)t * edgeCount * 2 + k] = edges[k];
    if (pt_graph_upload(s.engine, s.pos.data(), ed.data(), edgeCount, s.ecount.data(), nullptr, nullptr, s.temps.data()) != PT_OK) return;
    if (pt_graph_set_step(s.engine, s.calls) != PT_OK) return;
    if (pt_graph_run(s.engine, 1) != PT_OK) return;
    // a device-side failure (a cost term out of range, ...) comes back from the download: outputs stay untouched
    if (pt_graph_download(s.engine, s.pos.data(), nullptr, 0, nullptr, s.costs.data()) != PT_OK) return;
    s.calls++;
    for (size_t i = 0; i < np; i++) positions[i] = (Real)s.pos[i];
    for (int i = 0; i < tempCount; i++) costs[i] = (Real)s.costs[i];
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
    std::lock_guard<std::mutex> lock(s.mu);
    drop(s);
}

}  // extern "C"

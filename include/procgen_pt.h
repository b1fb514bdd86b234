/*
 * procgen_pt.h -- C ABI of libgpu_compute.so, the MI355X (gfx950) parallel-
 * tempering library that takes the place of the reference's CUDA library of
 * the same file name.
 *
 * Two layers:
 *   1. The legacy entry point the unchanged F# host binds by P/Invoke
 *      (reference src/GPU.Compute.fs:31-40), in the width that host really
 *      marshals (double arrays; F# `float` is System.Double), plus an _f32
 *      twin for callers that follow the CUDA source's `float*` signature
 *      (reference src/gpu/parallel_tempering.cu:130-138).
 *   2. A batched engine API (create / upload / run(steps) / download) that
 *      keeps all replica state resident in HBM and runs whole swap intervals
 *      per launch.  The legacy entry point is a thin wrapper over it.
 *
 * All pointers are plain host pointers unless a name starts with d_.
 * No function throws or aborts across this boundary; every engine function
 * returns a PT_* code and records a message for pt_last_error_string().
 */
#ifndef PROCGEN_PT_H
#define PROCGEN_PT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes -------------------------------------------------------- */
#define PT_OK                    0
#define PT_ERR_INVALID_ARGUMENT  1
#define PT_ERR_NO_DEVICE         2
#define PT_ERR_HIP               3
#define PT_ERR_NUMERIC_RANGE     4   /* a cost term was NaN/Inf or >= 2^62 */
#define PT_ERR_CAPACITY          5   /* addEdge hit edge_capacity (treated as a no-op move) */
#define PT_ERR_STREAM_EXHAUSTED  6   /* injected draw arrays shorter than the run */
#define PT_ERR_OUT_OF_MEMORY     7
#define PT_ERR_CLUSTER_TIMEOUT   8   /* workgroups cooperating on one replica lost each other (not co-resident) */

/* Last error of the calling thread's most recent library call
 * (additive exports; the reference library reports nothing at all:
 * reference src/gpu/parallel_tempering.cu:130-248 returns void and checks no
 * CUDA status). */
int pt_last_error(void);
const char *pt_last_error_string(void);

/* ---- legacy entry points ------------------------------------------------ */

/*
 * Replaces reference src/gpu/parallel_tempering.cu:130-248 behind the extern
 * declared at reference src/GPU.Compute.fs:31-40 and called from
 * src/GPU.Wrappers.fs:112 (availability probe, all counts 0) and :315-323
 * (once per iteration).
 *
 *   positions     [tempCount][nodeCount][3], in/out, chain-major
 *   edges         [edgeCount][2] (from,to) of the shared topology, in
 *   costs         [tempCount], out
 *   temperatures  [tempCount], in, index 0 hottest
 *   edgeCount     number of EDGES (as GPU.Wrappers.fs:293,321 passes it)
 *
 * One call is one iteration of ParallelTempering.run's loop body
 * (reference src/ParallelTempering.Core.fs:162-177): every chain makes one
 * Metropolis proposal (position perturbation; the shared topology cannot
 * change through this ABI), and a sequential adjacent swap pass runs when the
 * call counter is a multiple of swapFrequency(nodeCount).  The call counter
 * and the random streams persist across calls with equal
 * (nodeCount, edgeCount, tempCount) and restart when they change.
 * Node 0 is the origin, node nodeCount-1 the terminal, every other node
 * Standard 0.0 (node values do not cross this ABI).
 * With all counts 0 (the host's availability probe) it returns at once without touching the GPU state;
 * pt_last_error() is then PT_ERR_NO_DEVICE when the process has no usable GPU (the host's available()
 * can only see an exception, so a host that wants the CPU fallback in that case checks it).
 * On failure the outputs are left untouched; see pt_last_error().
 */
void parallel_tempering_step(double *positions, int *edges, double *costs, double *temperatures,
                             int nodeCount, int edgeCount, int tempCount);

/* Same contract with 32-bit floats, the width written in the CUDA source
 * (reference src/gpu/parallel_tempering.cu:131-137). */
void parallel_tempering_step_f32(float *positions, int *edges, float *costs, float *temperatures,
                                 int nodeCount, int edgeCount, int tempCount);

/* Drop the legacy entry points' persistent state, collision_detection's work memory included (the reference
 * never frees its static device buffers, src/gpu/parallel_tempering.cu:139-146).
 * Nothing of the caller's arrays is retained after parallel_tempering_step returns.  The one exception is
 * opt-in: with PROCGEN_PT_PIN_CALLER_ARRAY=1 in the environment the positions array is page-locked
 * (hipHostRegister) once the same array has come twice in a row and stays so until a different array comes or
 * this function is called; a host that sets it must keep the array alive until then and call this function
 * before freeing it. */
void parallel_tempering_reset(void);

/* Free the device work memory collision_detection / collision_detection_f32 keep between calls (it grows to the
 * largest call's needs; a pair list above 64 MiB is freed when its call ends anyway).  No counterpart in the
 * reference, which allocates and frees per call (src/gpu/collision_detection.cu:60-117). */
void pt_collision_release(void);

/* Replaces the reference export of the same name (reference src/GPU.Compute.fs:42-49, caller
 * src/GPU.Wrappers.fs:347-416): all pairs i < j with 0.001 < |p_j - p_i| < minDistance (fp32 distance,
 * the predicate of the host's CPU routine, reference src/SpatialHash.fs:89-146), written as
 * i0,j0,i1,j1,... sorted by (i, j); out_count[0] = number of pairs.  F# `float` is System.Double, so the
 * unchanged host marshals double*: collision_detection takes doubles, collision_detection_f32 floats.
 * (The fp32 entry's bound is the fp32 value it receives: a pair whose distance lies between that and the
 * double the caller rounded it from is judged against the fp32 value.)
 * The host's buffer holds nodeCount * nodeCount / 2 pairs.  On failure the outputs are left untouched
 * (pt_last_error tells why). */
void collision_detection(const double *positions, int *collision_pairs, int node_count, double min_distance, int *out_count);
void collision_detection_f32(const float *positions, int *collision_pairs, int node_count, float min_distance, int *out_count);

/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the same pair search as
 * collision_detection_f32 on arrays already on the device, with all work memory in a fixed array that
 * is part of the library's device image (about 1.4 MB per GPU, held until the process ends).  It calls
 * no hipMalloc / hipFree; the first call on a GPU looks that array's address up once.
 *   d_positions   [node_count][3] float, device
 *   d_pairs       [pair_capacity][2] int, device: the first min(found, pair_capacity) pairs of the
 *                 (i, j) order
 *   d_count       one int, device: min(found, pair_capacity)
 *   stream        a hipStream_t (NULL: the default stream); the call returns once the work is queued
 * node_count above PT_COLLISION_DEVICE_MAX_NODES and pair_capacity above INT_MAX are refused with
 * PT_ERR_INVALID_ARGUMENT.
 * All calls on one GPU share the work array: calls that overlap in time on one GPU (two streams, two
 * threads, a captured graph replayed beside any other call) race; queue them on one stream or order them
 * with events.
 * Under stream capture: the first call on a GPU takes a host lock and asks the runtime for the array's address
 * (hipGetSymbolAddress, which may load the code object) -- make one eager call per GPU, any size, before
 * capturing. */
#define PT_COLLISION_DEVICE_MAX_NODES 32768
int pt_collision_detect_device(const float *d_positions, int node_count, float min_distance, int *d_pairs,
                               long long pair_capacity, int *d_count, void *stream);

/* ---- batched spatial-graph engine --------------------------------------- */

typedef struct pt_graph_engine pt_graph_engine;

enum { PT_MUT_ADAPTIVE = 0, PT_MUT_PERTURB_ONLY = 1 };
enum { PT_RNG_PHILOX = 0, PT_RNG_INJECTED = 1 };
enum { PT_NODE_ORIGIN = 0, PT_NODE_TERMINAL = 1, PT_NODE_STANDARD = 2 };

/* Draws per chain per step: address k = 0 selector, 1 node / i / list index,
 * 2..4 x,y,z offsets (addEdge: 2 = j), 5 accept.  Unused addresses are skipped. */
#define PT_DRAWS_PER_STEP 6

typedef struct pt_graph_config {
    int32_t struct_size;        /* = sizeof(pt_graph_config) */
    int32_t node_count;         /* N >= 1 */
    int32_t edge_capacity;      /* per-replica capacity of the edge list */
    int32_t temp_count;         /* T: temperatures per ladder held by THIS engine */
    int32_t group_count;        /* G independent ladders; this engine runs G*T replicas */
    int32_t swap_frequency;     /* f; <= 0 selects max(5, int(sqrt(N)*2)) (reference ParallelTempering.Core.fs:138-139) */
    int32_t mutation_mode;      /* PT_MUT_* (reference Mutations.adaptive, ParallelTempering.Core.fs:67-79) */
    int32_t rng_mode;           /* PT_RNG_* */
    double weights[6];          /* connectivity, endpoint, edgeLength, collision, distribution, transitions
                                   (reference ProceduralGeneration.fs:100-107) */
    double target_edge_length;  /* 2.0  (reference ProceduralGeneration.fs:77) */
    double min_distance;        /* 0.5  (reference ProceduralGeneration.fs:78) */
    double min_distance_eps;    /* 0.001 (reference SpatialHash.fs:18) */
    uint64_t philox_key;        /* used when seed is NULL */
    const char *seed;           /* UTF-8; key = little-endian u64 of SHA-256(seed)[0..8], the bytes
                                   HashRNG turns into a float (reference Deterministic.fs:14-22) */
    int32_t device;             /* HIP device ordinal, < 0 = current device */
    int32_t group_base;         /* global index of this engine's first ladder (multi-GPU sharding by ladder) */
    int32_t ladder_temp_count;  /* global ladder length when ONE ladder is split over several engines
                                   (then group_count must be 1); 0 = temp_count */
    int32_t ladder_slot_base;   /* first temperature slot initially held by this engine's replicas */
    int32_t block_threads;      /* workgroup size, 0 = default */
    int32_t flags;              /* PT_FLAG_* */
    int32_t cluster_size;       /* workgroups (CUs) cooperating on one replica: 0 = choose, 1..4 */
    int32_t reserved;
} pt_graph_config;

/* Keep every array in global memory even when the replica would fit in LDS
 * (same results; exists so tests can run both placements on one input). */
#define PT_FLAG_GLOBAL_MEMORY_PLACEMENT 1
/* Re-evaluate the full cost for every proposal (one workgroup, or a cluster of them, per replica)
 * instead of updating the exact sums incrementally.  Same results bit for bit; a cluster_size
 * above 1 implies it.  Incremental evaluation is the default whenever the replica fits in LDS. */
#define PT_FLAG_FULL_EVALUATION 2
/* Incremental evaluation with every array in global memory and 32-bit node ids (what replicas too
 * large for a CU's LDS get by default); set on a small graph it exists so tests can run that kernel. */
#define PT_FLAG_STREAMING_INCREMENTAL 4

/* Fill a config with the reference's defaults for N nodes, T temperatures. */
void pt_graph_default_config(pt_graph_config *cfg, int32_t node_count, int32_t edge_capacity,
                             int32_t temp_count, int32_t group_count);

int pt_graph_create(const pt_graph_config *cfg, pt_graph_engine **out);
void pt_graph_destroy(pt_graph_engine *e);

/*
 * Upload initial state.  Replica (g, t) starts in temperature slot t of ladder g.
 *   positions    [G][T][N][3] float
 *   edges        [G][T][edge_stride][2] int32 in the reference's LIST order
 *                (index 0 = list head = most recently added edge, the order
 *                Marshaling.flattenEdges emits, reference GPU.Wrappers.fs:208-209)
 *   edge_counts  [G][T]
 *   node_kind    [G][N] PT_NODE_*           (NULL: node 0 origin, N-1 terminal, rest standard)
 *   node_value   [G][N] Standard values     (NULL: 0.0)
 *   temperatures [G][Tl] where Tl = ladder_temp_count or temp_count; index 0 hottest
 * Resets the step counter to 0 and computes every replica's initial cost
 * (reference ParallelTempering.Core.fs:157-160).
 */
int pt_graph_upload(pt_graph_engine *e, const float *positions, const int32_t *edges, int32_t edge_stride,
                    const int32_t *edge_counts, const uint8_t *node_kind, const double *node_value,
                    const double *temperatures);

/* Injected draws (PT_RNG_INJECTED): chain_u [G][T][steps][PT_DRAWS_PER_STEP],
 * swap_u [G][passes][Tl-1]; both doubles in [0,1), copied to the device. */
int pt_graph_set_draws(pt_graph_engine *e, const double *chain_u, int64_t steps,
                       const double *swap_u, int64_t passes);

/* Per-ladder Philox keys, keys [G]: ladder g then draws from keys[g] with the temperature slot as the
 * stream id (and swap stream 0), so that a ladder's trajectory does not depend on its index in the
 * batch (e.g. keys[g] = pt_hash_coordinates(pt_seed_key(seed), x, y, z)).  Without this call all ladders
 * share the engine's key and differ by stream id.  Call before the first step of a run: once a step has
 * run since the last upload the call fails with PT_ERR_INVALID_ARGUMENT (streams do not change in mid-run). */
int pt_graph_set_ladder_keys(pt_graph_engine *e, const uint64_t *keys);

/* Optional per-step trace for parity tests: records [max_steps][G*T] of
 * pt_trace_rec in slot order, and swap decisions [max_passes][G][Tl-1] bytes. */
typedef struct pt_trace_rec {
    int32_t kind;       /* 0 perturb, 1 addEdge, 2 removeEdge */
    int32_t a, b;       /* perturb: node; add: i, j; remove: list index; -1 when not drawn */
    int32_t flags;      /* bit0 no-op move, bit1 accept draw used, bit2 accepted, bit3 capacity hit */
    int32_t reach0;     /* nodes reachable from node 0 in the proposed state */
    int32_t reach_origin;
    double delta;
    double u;           /* accept draw or -1 */
    double cost_after;
} pt_trace_rec;
int pt_graph_enable_trace(pt_graph_engine *e, int64_t max_steps, int64_t max_passes);
int pt_graph_read_trace(pt_graph_engine *e, pt_trace_rec *steps_out, int64_t n_steps,
                        uint8_t *swaps_out, int64_t n_passes);

/*
 * Advance every replica by `steps` iterations of ParallelTempering.run's loop
 * (reference ParallelTempering.Core.fs:162-177), swap passes included
 * (after every step whose index is a multiple of f, step 0 included).
 * Asynchronous on the engine's stream; pt_graph_sync() waits.
 */
int pt_graph_run(pt_graph_engine *e, int64_t steps);
int pt_graph_sync(pt_graph_engine *e);
/* Index of the next step; the swap schedule and the draw addresses follow it
 * (upload resets it to 0; resuming a run sets it back). */
int pt_graph_set_step(pt_graph_engine *e, int64_t step);

/* Split-ladder operation (one ladder over several engines/GPUs):
 * pt_graph_run_interval advances to just before the next swap pass and
 * returns; the caller all-gathers the energies and calls pt_graph_swap_pass
 * with the full ladder's energies in REPLICA order (device pointer), which
 * every engine applies identically.  d_energy_local exposes this engine's
 * energies for the gather. */
int pt_graph_run_to_swap_point(pt_graph_engine *e, int64_t max_steps, int64_t *steps_done, int32_t *swap_due);
int pt_graph_local_energy_ptr(pt_graph_engine *e, double **d_energy_local, int32_t *count);
int pt_graph_swap_pass(pt_graph_engine *e, const double *d_energy_all_replicas);
/* Copy this engine's energies (replica order, temp_count * group_count doubles) to dst, a host or a
 * device pointer; waits for the steps launched so far. */
int pt_graph_read_energies(pt_graph_engine *e, double *dst);

/* Download in SLOT order (chain i of ladder g = temperature slot i, as the
 * reference's chains array).  Any pointer may be NULL.  Synchronises.
 *   positions [G][T][N][3], edges [G][T][edge_stride][2] list order,
 *   edge_counts [G][T], costs [G][T], slots [G][T] = slot held by local replica t. */
int pt_graph_download(pt_graph_engine *e, float *positions, int32_t *edges, int32_t edge_stride,
                      int32_t *edge_counts, double *costs);
int pt_graph_replica_slots(pt_graph_engine *e, int32_t *slot_of_replica_all);

/* Cost terms of the current state of every local replica, slot order:
 * [G*T][8] = conn, endpoint, edge, coll, dist, trans, total, reach0. */
int pt_graph_cost_terms(pt_graph_engine *e, double *terms);

/* Counters since upload: proposals made, accepted, swap attempts, swaps done,
 * status bits (PT_ERR_* seen on device, as a bitmask 1<<code), and how many removals of a search-tree
 * edge the incremental kernels rejected from a lower bound of the cost alone (only when no trace is
 * recorded; the decision is the one the full count gives). */
typedef struct pt_graph_stats {
    int64_t steps, proposals, accepted, swap_attempts, swaps;
    uint32_t status_bits;
    uint32_t early_rejects;
} pt_graph_stats;
int pt_graph_get_stats(pt_graph_engine *e, pt_graph_stats *out);

/* Timing of the step kernel: milliseconds and launches accumulated by HIP
 * events on the engine's stream since the last reset (for the benchmark). */
int pt_graph_kernel_time(pt_graph_engine *e, double *ms, int64_t *launches, int32_t reset);
void *pt_graph_stream(pt_graph_engine *e);

/* How the engine decided to run (diagnostics, benchmark records). */
typedef struct pt_graph_info_rec {
    int32_t incremental;        /* 1: incremental evaluation, 0: full evaluation per proposal */
    int32_t lds_placement;      /* 1: replica state resident in LDS */
    int32_t cluster_size;       /* workgroups per replica */
    int32_t block_threads;
    int32_t lds_bytes;          /* dynamic LDS per workgroup of the step kernel */
    int32_t swap_frequency;
    int32_t reserved[2];
} pt_graph_info_rec;
int pt_graph_info(pt_graph_engine *e, pt_graph_info_rec *out);

/* ---- convergence monitoring ---------------------------------------------- */
/* Convergence.ConvergenceMetrics (reference src/ParallelTempering.Core.fs:315-322); the two
 * options are a flag and a value. */
typedef struct pt_convergence_metrics {
    double gelman_rubin;            /* valid when has_gelman_rubin */
    double effective_sample_size;   /* valid when has_ess; of chain 0, as the reference takes it */
    double cost_stabilization;
    double swap_acceptance_rate;
    int32_t has_gelman_rubin, has_ess, has_converged, iteration;
} pt_convergence_metrics;

/* Replaces Convergence.initHistory / updateHistory (reference src/ParallelTempering.Core.fs:333-356):
 * from the next step on the device keeps, per ladder, the last window_size costs of every chain
 * (taken after the swap pass on swap steps) and the last window_size swap outcomes.  Calling it
 * again empties the history.  Full ladders only (temp_offset 0, temp_local = temp_count). */
int pt_graph_enable_convergence(pt_graph_engine *e, int32_t window_size);
/* Convergence.checkConvergence (reference src/ParallelTempering.Core.fs:442-492) on the device,
 * one record per ladder; `iteration` is copied into the records.  Synchronises. */
int pt_graph_check_convergence(pt_graph_engine *e, double strictness, int32_t iteration, pt_convergence_metrics *out);
/* Convergence.runUntilConvergence (reference src/ParallelTempering.Core.fs:496-550): steps from
 * the engine's current step up to max_iterations, checking after every step s with
 * s % check_freq == 0 and s > window/2, and stops as soon as EVERY ladder has converged (the
 * reference runs one ladder).  The history window is max(200, max_iterations/10) unless
 * pt_graph_enable_convergence was called before.  out[group_count] holds the last check (the
 * final one at max_iterations when no earlier check stopped the run); *steps_done the steps run. */
int pt_graph_run_until_converged(pt_graph_engine *e, int64_t max_iterations, int32_t check_freq, double strictness,
                                 int64_t *steps_done, pt_convergence_metrics *out);

/* Replaces the extern declared at reference src/GPU.Compute.fs:98-109 (implemented by
 * src/gpu/convergence.cu:255-363; no host code calls it).  The argument types are the ones the
 * F# declaration marshals: float[] and float are doubles, bool[] arrives as 4-byte elements, the
 * record is four floats and two ints (:88-96).  costHistory is [numChains][historyLength], oldest
 * sample first; historyLength doubles as the window size.  The statistics are the host routine's
 * (src/ParallelTempering.Core.fs:442-492), evaluated in fp64 and rounded into the record; an
 * absent statistic reads as the CUDA routine's placeholder (2.0 for R-hat, 1.0 for ESS).
 * The call completes before it returns (the record is host memory); `stream` is ignored. */
typedef struct ConvergenceMetrics {
    float GelmanRubin, EffectiveSampleSize, CostStabilization, AvgSwapAcceptance;
    int32_t Iteration, HasConverged;
} ConvergenceMetrics;
void check_convergence_async(const double *costHistory, const int32_t *swapHistory, int numChains, int historyLength,
                             int swapHistoryLength, double strictness, int iteration, ConvergenceMetrics *outMetrics,
                             void *stream);

/* ======================================================================================
 * Voxel-chunk engine: parallel tempering over 32^3 terrain chunks.
 *
 * Replaces, for whole batches of chunks, what the reference runs per chunk on the CPU:
 * Terrain.Generation.generate (reference src/Terrain.Generation.fs:172-199) =
 * ParallelTempering.run with mutateVoxels (:123-169) and the cost built by
 * Terrain.Constraints.build (reference src/Terrain.Constraints.fs:213-247).
 * group = one chunk's ladder; every temperature of a ladder holds its own copy of the chunk.
 * Voxel v = x + 32 y + 1024 z (reference src/Voxel.Core.fs:113-114).
 * Draw addresses of a step (8 per step): 0 voxel index, 1 refinement selector, 2 density single,
 * 3 temperature selector, 4 temperature single, 5 moisture selector, 6 moisture single, 7 accept.
 * ====================================================================================== */
typedef struct pt_chunk_engine pt_chunk_engine;

#define PT_CHUNK_RES 32
#define PT_CHUNK_VOXELS (32 * 32 * 32)
#define PT_CHUNK_DRAWS_PER_STEP 8
#define PT_CHUNK_TERMS 12   /* traversability, geological, overhang, erosion, aesthetics, caves, performance,
                               biomeTransition, total, cave reach, underground air voxels, solid voxels */

typedef struct pt_chunk_config {
    int32_t struct_size;
    int32_t temp_count;             /* temperatures per chunk ladder (reference default 6) */
    int32_t group_count;            /* chunks */
    int32_t swap_frequency;         /* <= 0: 50 (reference src/Terrain.Generation.fs:18-23) */
    int32_t rng_mode;               /* PT_RNG_PHILOX / PT_RNG_INJECTED */
    int32_t device;                 /* -1: current */
    int32_t group_base;             /* added to the ladder index in stream ids (distinct ranks, distinct streams) */
    int32_t flags;
    double weights[8];              /* traversability, geological, overhang, erosion, aesthetics, caves, performance, biomeTransition */
    float density_mutation_scale;   /* 0.1 */
    float solid_to_air_threshold;   /* 0.5 */
    double refinement_probability;  /* 0.7 */
    uint64_t philox_key;
    const char *seed;               /* when not NULL: key = first 8 bytes of SHA-256(seed) */
} pt_chunk_config;

void pt_chunk_default_config(pt_chunk_config *cfg, int32_t temp_count, int32_t group_count);
int pt_chunk_create(const pt_chunk_config *cfg, pt_chunk_engine **out);
void pt_chunk_destroy(pt_chunk_engine *e);
/* density, moisture, temperature [G][T][32768] f32; material [G][32768] u8 (shared by a ladder, never
 * mutated); temperatures [G][T] fp64, index 0 hottest.  Computes the initial costs. */
int pt_chunk_upload(pt_chunk_engine *e, const float *density, const float *moisture, const float *temperature,
                    const uint8_t *material, const double *temperatures);
/* Chunk positions, xyz [G][3]: chunk g then draws from the key pt_hash_coordinates(engine key, x, y, z)
 * (the chunk seed of reference src/Terrain.Generation.fs:172-176) with its temperature slot as the stream
 * id, so its trajectory depends on (seed, position) and not on where in the batch it sits.  Without this
 * call all chunks share the engine's key and chunk g draws from stream (group_base + g) * T + slot.
 * pt_chunk_set_ladder_keys gives the per-chunk keys directly.  Call before the first step of a run: after
 * steps have run since the last upload both calls fail with PT_ERR_INVALID_ARGUMENT. */
int pt_chunk_set_positions(pt_chunk_engine *e, const int32_t *xyz);
int pt_chunk_set_ladder_keys(pt_chunk_engine *e, const uint64_t *keys);
/* injected draws: chain_u [G][T][steps][8], swap_u [G][passes][T-1] */
int pt_chunk_set_draws(pt_chunk_engine *e, const double *chain_u, int64_t steps, const double *swap_u, int64_t passes);
int pt_chunk_enable_trace(pt_chunk_engine *e, int64_t max_steps, int64_t max_passes);
/* records as pt_trace_rec: kind 0 refinement / 1 flip, a = voxel, b = bit0 temperature mutated | bit1
 * moisture mutated, reach0 = cave reach, reach_origin = underground air voxels (proposed state) */
int pt_chunk_read_trace(pt_chunk_engine *e, pt_trace_rec *steps_out, int64_t n_steps, uint8_t *swaps_out, int64_t n_passes);
int pt_chunk_run(pt_chunk_engine *e, int64_t steps);
int pt_chunk_sync(pt_chunk_engine *e);
/* slot order; any pointer may be NULL */
int pt_chunk_download(pt_chunk_engine *e, float *density, float *moisture, float *temperature, double *costs);
int pt_chunk_replica_slots(pt_chunk_engine *e, int32_t *slot_of_replica);
/* from-scratch evaluation of the current states, slot order: terms [G][T][PT_CHUNK_TERMS] */
int pt_chunk_cost_terms(pt_chunk_engine *e, double *terms);
int pt_chunk_get_stats(pt_chunk_engine *e, pt_graph_stats *out);
int pt_chunk_kernel_time(pt_chunk_engine *e, double *ms, int64_t *launches, int32_t reset);
/* How the step kernel is launched for this engine's replica count: lanes per workgroup (64 or 256) and how many
 * of them hold a replica (the other lanes of a 64-lane workgroup assist: shared draws, wave-wide cave search).
 * The host picks it from the replica count and the device's compute units; PROCGEN_PT_CHUNK_RPB (a power of two,
 * the fewest replicas per 64-lane workgroup) and PROCGEN_PT_CHUNK_WIDE=1 (256-lane workgroups), read at create
 * time, let the parity tests run every route at small sizes.  Either pointer may be NULL. */
int pt_chunk_launch_shape(pt_chunk_engine *e, int32_t *block_threads, int32_t *replicas_per_workgroup);

/* Unit-test hook: the number of floats x in [lo, hi) (positive, normal) whose square root as the node passes compute it
 * differs from the correctly rounded one.  The passes vouch for [2^-20, 2^24) before they use it. */
int pt_debug_sqrt_check(float lo, float hi, uint64_t *mismatches);
/* Unit-test hook: the exact fixed-point sum the engine uses, over n doubles. */
int pt_debug_canonical_sum(const double *terms, int64_t n, double *out, int32_t *range_error);

/* Scalar helpers mirroring the reference (ParallelTempering.Core.fs:88-95, :138-139). */
int pt_geometric_ladder(double t_min, double t_max, int32_t n, double *out);
int32_t pt_swap_frequency(int32_t node_count);
uint64_t pt_seed_key(const char *seed);
/* DeterministicGeneration.hashCoordinates (reference src/Deterministic.fs:94-102): the seed of the chunk at
 * (x, y, z) as Terrain.Generation.generate derives it (reference src/Terrain.Generation.fs:172-176). */
uint64_t pt_hash_coordinates(uint64_t base_seed, int32_t x, int32_t y, int32_t z);

#ifdef __cplusplus
}
#endif
#endif

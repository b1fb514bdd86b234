// Spatial-graph parallel-tempering kernels for gfx950: one workgroup per
// (ladder, temperature) replica, whole swap intervals per launch.
//
// What one proposal does, following the reference CPU path
// (reference src/ParallelTempering.Core.fs:122-133 optimizationStep):
//   Mutations.adaptive (:67-79) -> full weighted cost of the proposed state
//   (reference src/ProceduralGeneration.fs:73-81) -> Acceptance.acceptMove (:32-34).
// The proposed state is never materialised: the cost is evaluated over the
// current arrays plus a one-element overlay (moved node / added edge / removed
// edge) and the arrays are edited only when the move is accepted.
#pragma once

#include "pt_device.h"
#include "../../include/procgen_pt.h"

// One record per cluster member (its own 128-byte line) and the leader's decision word.
struct __align__(128) MemberSlot {
    unsigned long long w[8];    // exact partial sums / BFS counts / error bit
    unsigned long long flag;    // step + 1 once w[] is complete
    unsigned long long pad[7];
};
struct __align__(128) ClusterComm {
    MemberSlot m[4];
    unsigned long long decision;    // (step + 1) << 8 | accept
    unsigned long long pad[15];
};

struct BigComm;

struct GraphParams {
    // sizes
    int N, Ecap, T, G, Tl, slot_base, group_base;
    int C;                      // workgroups per replica (cluster size); edge lists exist once per member
    int mb, gx, gy, gz;         // hash grid: M = 1<<mb buckets = (1<<gx)*(1<<gy)*(1<<gz)
    int lgx, lgy, lgz;          // hash grid of the LDS-resident placement
    int nwords;                 // visited bitmap words
    int mutation_mode, rng_mode;
    int cells_in_lds, bitmap_in_lds;
    // cost
    double w[6], target_len, min_dist, eps, inv_cell;
    float inv_cell_f, d2_hi, d2_lo;     // see cell_coord() and the collision test
    int cell_pow2, collcap, collstride;     // collstride: threads per workgroup (rows of the deferred-collision lists)
    uint32_t key0, key1;
    // When not null: ladder g draws from key group_keys[g] with its temperature slot as the stream id, so
    // that its trajectory does not depend on where in the batch it sits (otherwise: one key, stream id
    // (group_base + g) * Tl + slot).
    const unsigned long long *group_keys;
    // state, replica-major
    float *px, *py, *pz;        // [R][N]
    int *ea, *eb;               // [R][C][2][Ecap]  ARRAY order (reverse of the reference's list order), double buffered
    double *et;                 // [R][C][2][Ecap]  per-edge smoothTransitions term
    ClusterComm *comm;          // [R]
    int *ebuf;                  // [R] which half is current
    int *ecount;                // [R]
    const uint8_t *kind;        // [G][N]
    const double *value;        // [G][N]
    const int *origin_idx, *terminal_idx;   // [G]
    int *rowend, *adj;          // CSR of the current edges: row u = adj[rowend[u-1] .. rowend[u])
    int4 *adj4;                 // [R][N] out-degree + first three out-neighbours of every node
    double *cost;               // [R]
    const double *temps;        // [G][Tl]
    const int *slot_of_rep;     // [G][Tl]
    // scratch, replica-major
    int *cells;                 // [R][M+1]  (when not in LDS)
    int *sorted;                // [R][N]
    int *queue;                 // [R][2][N]
    unsigned *bitmap;           // [R][nwords] (when not in LDS)
    float *collbuf;             // [R][C][collcap][collstride]
    // draws
    const double *chain_u;      // [G][Tl][draw_steps][6]
    long long draw_steps;
    // outputs
    pt_trace_rec *trace;        // [trace_steps][G*Tl]
    long long trace_steps;
    unsigned long long *stats;  // [R][4] proposals, accepted, -, -
    unsigned *status;           // bitmask of PT_ERR_* seen
    double *terms;              // [R][8] or null
    // state of the incremental evaluation (null when it is not available), kept between launches
    fx128 *dacc;                // [R][8]  exact sums of the committed state, indexed like the LDS accumulators
    int *dreach;                // [R][4]  reach from node 0, reach from the origin, largest discovery label in use, -
    unsigned *dvis;             // [R][2][nwords]  visited sets of the two searches
    unsigned short *dparent;    // [R][N]  search from node 0: the node every reached node was discovered from
    unsigned short *dparscr;    // [R][N]  scratch: parents found while a proposal is evaluated
    unsigned *dlabel;           // [R][N]  discovery labels: strictly increasing from parent to child
    unsigned *dlabscr;          // [R][N]  scratch: labels found while a proposal is evaluated
    unsigned *epk;              // [R][epk_stride]  scratch: the edge list packed as a | b << 16
    int epk_stride;             // edges-per-thread bound times the workgroup size
    // incremental evaluation with every array in global memory (pt_graph_big_kernel; null otherwise)
    unsigned *dparent32;        // [R][N]  discovery parents, 32-bit ids
    unsigned *dparscr32;        // [R][N]  scratch: parents found while a proposal is evaluated
    unsigned *bvis;             // [R][3][nwords]  scratch visited sets (two searches) and the cut-off subtree
    unsigned *bjmp;             // [R][N]  scratch of the subtree search
    float *bdist;               // [R][N]  distances to the proposed centroid between the two distribution passes
    int CB, R;                  // pt_graph_bigc_kernel: workgroups per replica; replicas of this engine
    int bigc_cache_runs;        // that kernel: runs of four distances a workgroup keeps in LDS between the two distribution passes
    struct BigComm *bcomm;      // [R] cluster records of that kernel
    unsigned *binc;             // [R][N][16]  that kernel's per-node lists of incident edges: (other end << 1) | (1 when the edge leaves the node)
    int *bdeg;                  // [R][N]  entries in use; above 16: the node's list overflowed and is not used
    int walk_members, walk_entering, walk_rounds;   // pt_graph_delta_kernel: bounds of the walk of a cut-off subtree through the lists
    // convergence monitoring (null when off): cost of every chain after every step, a ring per chain
    double *conv_hist;          // [G][Tl][conv_window], step s at index s % conv_window
    int conv_window;
};

struct SwapParams {
    int G, Tl, group_base, rng_mode;
    uint32_t key0, key1;
    const unsigned long long *group_keys;   // see GraphParams
    const double *temps;        // [G][Tl]
    int *slot_of_rep;           // [G][Tl]
    int *rep_of_slot;           // [G][Tl]
    const double *energy;       // [G][Tl] replica order
    const double *swap_u;       // [G][passes][Tl-1]
    long long draw_passes;
    uint8_t *swap_trace;        // [trace_passes][G][Tl-1]
    long long trace_passes;
    unsigned long long *swap_stats; // [G][2] attempts, swaps
    unsigned *status;
    // convergence monitoring (null when off)
    double *conv_hist;          // [G][Tl][conv_window]: rewritten for the swap step with the costs after the pass
    uint8_t *conv_swaps;        // [G][conv_window] ring of swap outcomes in the order the reference appends them
    unsigned long long *conv_swap_total;    // [G] outcomes appended so far
    int conv_window, swap_frequency;
};

// One convergence check per ladder (pt_graph_convergence_kernel).
struct ConvParams {
    const double *hist;         // chain c of ladder g, sample k (0 = oldest): hist[g*ladder_stride + c*chain_stride + (first + k) % ring]
    long long ladder_stride, chain_stride;
    int ring, first, n;         // ring size, ring index of the oldest sample, samples per chain
    int chains, window, ladders;
    const uint8_t *swaps;       // [ladders][swap_ring]
    const unsigned long long *swap_total;   // [ladders] outcomes appended so far, or null: swap_n entries each
    int swap_ring, swap_n;
    double strictness;
    int iteration;
    pt_convergence_metrics *out;    // [ladders]
};

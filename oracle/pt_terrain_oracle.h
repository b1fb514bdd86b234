/*
 * pt_terrain_oracle.h -- CPU restatement of the reference's voxel-chunk parallel tempering
 * (terrain constraints, voxel mutation, the shared PT loop).
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h): nothing that ships may include, link or call it.
 * The reference is F#/.NET 8 and cannot run here; its tests hold no numeric vector for the
 * terrain cost, so this restatement is "parity unpinned": it is checked against values derived
 * by hand from the reference's formulas (tests/test_terrain_oracle.py), nothing more.
 *
 * Voxel v = x + y*res + z*res*res (reference src/Voxel.Core.fs:113-114), position
 * (x, y, z) * 0.5 (:73-76, 106-111); edges directed to the +x, +y, +z neighbours (:147-164).
 */
#ifndef PT_TERRAIN_ORACLE_H
#define PT_TERRAIN_ORACLE_H

#include "pt_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { OT_W_TRAVERSABILITY = 0, OT_W_GEOLOGICAL, OT_W_OVERHANG, OT_W_EROSION, OT_W_AESTHETICS, OT_W_CAVES,
       OT_W_PERFORMANCE, OT_W_BIOME, OT_W_COUNT };

typedef struct ot_config {
    double w[OT_W_COUNT];           /* reference src/Terrain.Constraints.fs:238-247 */
    float density_mutation_scale;   /* 0.1  reference src/Terrain.Generation.fs:68-72 */
    double refinement_probability;  /* 0.7 */
    float solid_to_air_threshold;   /* 0.5 */
    int32_t sum_mode;               /* OG_SUM_FSHARP: the reference's sequential sums; OG_SUM_CANONICAL: exact sums */
    int32_t res;                    /* 32 */
} ot_config;

void ot_default_config(ot_config *c);

typedef struct ot_terms {
    double trav, geo, overhang, erosion, aesthetics, caves, performance, biome, total;
    int32_t reached;            /* voxels the cave search reaches (1 = only voxel 0) */
    int32_t underground_air;    /* air voxels with y <= res/2 */
    int32_t solid, surface;     /* solid voxels; interior solid voxels with an air neighbour */
    int32_t range_error;
} ot_terms;

/* Terrain.Constraints.build (reference src/Terrain.Constraints.fs:213-235) */
double ot_cost(const float *density, const float *moisture, const uint8_t *material, const ot_config *c, ot_terms *out);

/* The generic path the reference takes for the traversability term: filter the lattice edges by
 * slope, then the directed BFS from voxel 0 (:12-31).  Returns the number of voxels reached. */
int ot_traversable_reach(int res);

typedef struct ot_trace_rec {
    int32_t kind;       /* 0 refinement, 1 flip */
    int32_t a;          /* voxel index */
    int32_t b;          /* bit0: temperature mutated, bit1: moisture mutated */
    int32_t flags;      /* bit1 drew accept u, bit2 accepted */
    int32_t reach0;     /* cave search reach of the proposed state */
    int32_t reach_origin; /* underground air voxels of the proposed state */
    double delta;
    double u;
    double cost_after;
} ot_trace_rec;

/* ParallelTempering.run over voxel chunks (reference src/Terrain.Generation.fs:172-199 with
 * mutateVoxels :123-169): one ladder.  Arrays [T][V] in slot order, material [V] shared.
 * Draw addresses within a step (stride 8): 0 voxel index, 1 refinement selector, 2 density single,
 * 3 temperature selector, 4 temperature single, 5 moisture selector, 6 moisture single, 7 accept. */
typedef struct ot_run_args {
    int32_t temps, iters, swap_freq, threads;
    float *density, *moisture, *temperature;    /* in/out [T][V] */
    const uint8_t *material;                    /* [V] */
    const double *ladder;                       /* [T] */
    double *cost;                               /* out [T] */
    og_stream *chain_streams;                   /* [T] */
    og_stream *swap_stream;
    ot_trace_rec *trace;                        /* [iters][T] or NULL */
    uint8_t *swap_trace;                        /* [passes][T-1] or NULL */
    int64_t *n_swap_passes;
    int64_t first_step;
} ot_run_args;

int ot_run(const ot_run_args *a, const ot_config *c);

#ifdef __cplusplus
}
#endif
#endif

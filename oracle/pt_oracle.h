/*
 * pt_oracle.h -- CPU restatement of the reference's parallel-tempering CPU path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing that ships (the HIP library, its host
 * layer, the Python bindings of the product) may include, link or call this.
 * It exists so the tests and the CPU-baseline leg of the benchmark have an
 * independent statement of what the reference computes.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  The reference is F#/.NET 8 and cannot run here, so
 * this is a restatement, not a build of the reference.  It is pinned by the
 * reference's own known-answer values (tests/TestData.fs:8-19 and the values
 * quoted in doc comments); whole-trajectory behaviour beyond those is
 * "parity unpinned": .NET's System.Random, Math.Exp and Vector3 are not
 * available, so random draws are injected as streams of doubles in [0,1).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- random streams ---------------------------------------------------- */

/* A stream hands out doubles u in [0,1).  mode 0: from an array (injected
 * draws); mode 1: Philox4x32-10, the p-th double of stream (stream_id, domain)
 * under a 64-bit key.
 * stride == 0: draws are consumed strictly in call order.
 * stride  > 0: addressed draws -- step s of a chain owns positions
 *   s*stride .. s*stride+stride-1 and the k-th KIND of draw of a step always
 *   reads position s*stride+k (k = 0 selector, 1 node / i / list index,
 *   2..4 x,y,z or j at 2, 5 accept); draws the step does not make leave their
 *   position unread.  Within a step the reference's call order is kept
 *   (selector, move draws, accept draw); addressing by (chain, step, k) makes
 *   the stream a pure function of a counter, which is what a parallel
 *   implementation needs. */
typedef struct og_stream {
    int32_t mode;
    int32_t exhausted;      /* set when an array stream runs dry (returns 0.0) */
    const double *u;
    int64_t len;
    uint32_t key[2];
    uint32_t stream_id;
    uint32_t domain;        /* 0 = per-chain proposal stream, 1 = swap stream */
    int64_t pos;
    int64_t stride;
    int64_t step_base;      /* = step*stride, set by og_run at each step */
} og_stream;

void og_stream_array(og_stream *s, const double *u, int64_t len);
void og_stream_philox(og_stream *s, uint64_t key, uint32_t stream_id, uint32_t domain);
double og_stream_next(og_stream *s);
double og_stream_at(og_stream *s, int64_t p);

void og_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void og_sha256(const uint8_t *data, size_t len, uint8_t out[32]);
/* Deterministic.fs:14-27: u = LE_u64(SHA256(utf8(seed + decimal(index)))[0..8]) / 2^64-ish */
double og_hashrng_next_float(const char *seed, int index);
int og_hashrng_next_int(const char *seed, int index, int max);
uint64_t og_hash_coordinates(uint64_t base_seed, int x, int y, int z);
/* first 8 bytes (little endian) of SHA-256(utf8(seed)) */
uint64_t og_seed_key(const char *seed);

/* ---- graphs ------------------------------------------------------------ */

enum { OG_ORIGIN = 0, OG_TERMINAL = 1, OG_STANDARD = 2 };

typedef struct og_graph {
    int32_t n;
    int32_t e;
    int32_t ecap;
    float *pos;             /* 3*n */
    int32_t *edges;         /* 2*ecap, LIST order: index 0 is the list head (newest) */
    const uint8_t *kind;    /* n, shared, never mutated */
    const double *value;    /* n, Standard value, shared */
} og_graph;

enum { OG_SUM_FSHARP = 0, OG_SUM_CANONICAL = 1 };
enum { OG_ALGO_FAITHFUL = 0, OG_ALGO_OPTIMISED = 1 };
enum { OG_MUT_ADAPTIVE = 0, OG_MUT_PERTURB_ONLY = 1 };

typedef struct og_config {
    double w[6];            /* connectivity, endpoint, edgeLength, collision, distribution, transitions */
    double target_len;      /* 2.0 */
    double min_dist;        /* 0.5 */
    double eps;             /* 0.001 */
    int32_t sum_mode;
    int32_t algo;
    int32_t mutation_mode;
    int32_t brute_threshold;/* 200 */
} og_config;

void og_default_config(og_config *c);

typedef struct og_terms {
    double conn, endpoint, edge, coll, dist, trans, total;
    int32_t reach0, reach_origin;
    int64_t npairs;
    int32_t range_error;
} og_terms;

double og_cost(const og_graph *g, const og_config *c, og_terms *out);

int og_count_reachable(int n, const int32_t *edges, int e, int start, int algo);
double og_collision_penalty(const float *pos, int n, const og_config *c, int64_t *npairs, int *err);
/* SpatialHash.detectCollisions (SpatialHash.fs:89-101 predicate) as a list of pairs i < j in (i, j) order */
int64_t og_collision_pairs(const float *pos, int n, double min_dist, double eps, int32_t *out, int64_t cap);

/* ---- Convergence (reference src/ParallelTempering.Core.fs:312-492) --------- */
typedef struct og_conv_history og_conv_history;
typedef struct og_conv_metrics {
    double gelman_rubin, effective_sample_size, cost_stabilization, swap_acceptance_rate;
    int32_t has_gelman_rubin, has_ess, has_converged, iteration;
} og_conv_metrics;
og_conv_history *og_conv_new(int chains, int window);                  /* initHistory :333-339 */
void og_conv_free(og_conv_history *h);
/* updateHistory :342-356 as runUntilConvergence calls it (:523-536): chain_costs[chains] after the
 * step (after the swap pass when there was one); outcomes[n_outcomes] in PAIR order 0..T-2, which
 * the reference prepends one by one and so appends to the history last pair first. */
void og_conv_update(og_conv_history *h, const double *chain_costs, const uint8_t *outcomes, int n_outcomes);
void og_conv_check(const og_conv_history *h, int iteration, double strictness, og_conv_metrics *out);  /* :442-492 */

/* ---- scalar pieces ------------------------------------------------------ */
void og_geometric_ladder(double tmin, double tmax, int n, double *out, int *out_n);
int og_swap_frequency(int node_count);
double og_accept_probability(double current, double proposed, double temperature);
double og_swap_probability(double c1, double t1, double c2, double t2);
void og_uniform_sphere(int count, float *pos);
/* Deterministic.fs:44-135: the seeded workload generators */
void og_fibonacci_sphere(const char *seed, int count, double perturbation, float *pos);
void og_seeded_uniform_sphere(const char *seed, int count, double radius, float *pos);
void og_create_graph(const char *seed, int n, double perturbation, float *pos, uint8_t *kind, double *value);
int og_generate_edges(const char *seed, const float *pos, int n, double avg_degree, int32_t *edges, double *weights, int cap);

/* ---- stepping ----------------------------------------------------------- */

typedef struct og_trace_rec {
    int32_t kind;       /* 0 perturb, 1 addEdge, 2 removeEdge */
    int32_t a;          /* perturb: node index; add: i; remove: list index (or -1) */
    int32_t b;          /* add: j; else -1 */
    int32_t flags;      /* bit0 no-op proposal, bit1 drew accept u, bit2 accepted */
    int32_t reach0;
    int32_t reach_origin;
    double delta;
    double u;           /* accept draw, or -1 when none was drawn */
    double cost_after;  /* chain cost after the step */
} og_trace_rec;

typedef struct og_chain {
    og_graph g;         /* current state */
    og_graph scratch;   /* proposal buffer */
    double cost;
    double temperature;
} og_chain;

/* one optimizationStep (ParallelTempering.Core.fs:122-133) */
void og_step(og_chain *ch, const og_config *c, og_stream *s, og_trace_rec *tr);

/* ParallelTempering.run (ParallelTempering.Core.fs:149-179) for one ladder.
 * pos: [T][N][3], edges: [T][ecap][2] list order with e0[t] live entries.
 * trace may be NULL, else [iters][T]; swap_trace may be NULL, else
 * [n_swap_passes][T-1] bytes (1 = swapped).  threads>1 steps chains in
 * parallel (one thread per chain, capped), mirroring Execution.fs:31-39. */
typedef struct og_run_args {
    int32_t n, ecap, temps, iters, swap_freq, threads;
    float *pos;             /* in/out, slot order */
    int32_t *edges;         /* in/out */
    int32_t *ecount;        /* in/out [T] */
    const uint8_t *kind;
    const double *value;
    const double *temperature;
    double *cost;           /* out [T] */
    og_stream *chain_streams;   /* [T] */
    og_stream *swap_stream;     /* 1 */
    og_trace_rec *trace;
    uint8_t *swap_trace;
    int64_t *n_swap_passes;     /* out, may be NULL */
    int64_t first_step;         /* index of the first step: the steps run are first_step .. first_step+iters-1 */
} og_run_args;

int og_run(const og_run_args *a, const og_config *c);

#ifdef __cplusplus
}
#endif
#endif

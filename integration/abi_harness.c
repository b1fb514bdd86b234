/*
 * abi_harness.c -- what the unchanged F# host does at the native boundary, replayed from plain C.
 *
 * The reference's host reaches libgpu_compute.so through P/Invoke (reference src/GPU.Compute.fs:11-109): a
 * resolver loads the file by path, the externs are bound by name, arrays arrive as pinned pointers (F# `float`
 * is System.Double, so `float[]` is double*; `bool[]` marshals as 4-byte elements), structs as sequential
 * records.  No .NET runtime is at hand where this repository is built, so this program stands in for it:
 *   - it loads the library by path (dlopen, as NativeLibrary.Load does) and binds by name;
 *   - it checks at compile time that the records the glue module (integration/GPU.Batched.fs) declares have
 *     the size and field offsets of the C structs (.NET sequential layout: every field at its natural
 *     alignment, ByValArray inline, nativeint 8 bytes);
 *   - it replays the call sequences of GPU.Wrappers.fs: the availability probe (:109-120), the
 *     per-iteration loop of GPUTempering.optimize (:291-342), GPUCollision.detect (:347-416), and the
 *     check_convergence_async extern (src/GPU.Compute.fs:98-109) with a 4-byte-bool swap history;
 *   - it then drives the batched engine the way integration/GPU.Batched.fs does, and the voxel-chunk engine the
 *     way its TerrainBatched.generateFrom does for Terrain.Generation.generate (reference
 *     src/Terrain.Generation.fs:172-199): create, upload one ladder per chunk, chunk positions, run, download.
 *
 * usage: abi_harness <path to libgpu_compute.so> [--layout-only | --dump <file>]
 *   --layout-only   load, bind, probe; no call that needs a GPU
 *   --dump <file>   also write the job's inputs (int32 N, T, E, iterations; positions [T][N][3], temperatures
 *                   [T] as doubles; edges [E][2] int32) so that a checker can run the same job elsewhere, and
 *                   to <file>.chunks the terrain job's (int32 chunks, T, iterations, swap frequency; uint64 seed;
 *                   int32 xyz [chunks][3]; float32 density, moisture, temperature [chunks][32768]; uint8 material
 *                   [chunks][32768]; double temperatures [T])
 * Prints one line per phase and ABI_HARNESS_OK at the end; any failure exits non-zero.
 */
#include <dlfcn.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/procgen_pt.h"

/* ---- layouts as .NET marshals the records of integration/GPU.Batched.fs --------------------------- */
/* PtGraphConfig: 8 x int (4) | float[6] ByValArray (48, 8-aligned) | 3 x float (8) | uint64 | nativeint |
 * 8 x int (4) */
_Static_assert(offsetof(pt_graph_config, struct_size) == 0, "struct_size");
_Static_assert(offsetof(pt_graph_config, rng_mode) == 28, "rng_mode: 8th int");
_Static_assert(offsetof(pt_graph_config, weights) == 32, "weights: ByValArray of 6 doubles follows 8 ints");
_Static_assert(offsetof(pt_graph_config, target_edge_length) == 80, "target_edge_length");
_Static_assert(offsetof(pt_graph_config, min_distance_eps) == 96, "min_distance_eps");
_Static_assert(offsetof(pt_graph_config, philox_key) == 104, "philox_key: uint64");
_Static_assert(offsetof(pt_graph_config, seed) == 112, "seed: nativeint");
_Static_assert(offsetof(pt_graph_config, device) == 120, "device");
_Static_assert(offsetof(pt_graph_config, cluster_size) == 144, "cluster_size");
_Static_assert(sizeof(pt_graph_config) == 152, "PtGraphConfig is 152 bytes");
/* PtChunkConfig: 8 x int | float[8] ByValArray (64) | 2 x float32 | float | uint64 | nativeint */
_Static_assert(offsetof(pt_chunk_config, flags) == 28, "flags: 8th int");
_Static_assert(offsetof(pt_chunk_config, weights) == 32, "weights");
_Static_assert(offsetof(pt_chunk_config, density_mutation_scale) == 96, "density_mutation_scale: float32");
_Static_assert(offsetof(pt_chunk_config, refinement_probability) == 104, "refinement_probability");
_Static_assert(offsetof(pt_chunk_config, philox_key) == 112, "philox_key");
_Static_assert(offsetof(pt_chunk_config, seed) == 120, "seed");
_Static_assert(sizeof(pt_chunk_config) == 128, "PtChunkConfig is 128 bytes");
/* ConvergenceMetrics of the reference (src/GPU.Compute.fs:88-96): 4 x float32, 2 x int */
_Static_assert(sizeof(ConvergenceMetrics) == 24 && offsetof(ConvergenceMetrics, Iteration) == 16 &&
               offsetof(ConvergenceMetrics, HasConverged) == 20, "ConvergenceMetrics");
/* PtConvergenceMetrics: 4 x float (8), 4 x int */
_Static_assert(sizeof(pt_convergence_metrics) == 48 && offsetof(pt_convergence_metrics, has_gelman_rubin) == 32 &&
               offsetof(pt_convergence_metrics, iteration) == 44, "PtConvergenceMetrics");
_Static_assert(sizeof(pt_graph_stats) == 48, "pt_graph_stats");
_Static_assert(sizeof(pt_trace_rec) == 48, "pt_trace_rec");

#define FAIL(...) do { fprintf(stderr, "abi_harness: " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

typedef void (*step_fn)(double *, int *, double *, double *, int, int, int);
typedef void (*coll_fn)(const double *, int *, int, double, int *);
typedef void (*conv_fn)(const double *, const int32_t *, int, int, int, double, int, ConvergenceMetrics *, void *);
typedef int (*err_fn)(void);
typedef const char *(*errs_fn)(void);

static void *lib;
static void *must(const char *name)
{
    void *p = dlsym(lib, name);
    if (!p) FAIL("%s is not exported", name);
    return p;
}

/* a fixed pseudo-random sequence (inputs only) */
static uint64_t lcg = 0x9e3779b97f4a7c15ull;
static double rnd(void) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    if (argc < 2) FAIL("usage: abi_harness <libgpu_compute.so> [--layout-only | --dump <file>]");
    const int layout_only = argc > 2 && strcmp(argv[2], "--layout-only") == 0;
    const char *dump = (argc > 3 && strcmp(argv[2], "--dump") == 0) ? argv[3] : NULL;
    lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) FAIL("dlopen: %s", dlerror());

    /* the externs the host declares (src/GPU.Compute.fs:31-109) */
    step_fn step = (step_fn)must("parallel_tempering_step");
    coll_fn collide = (coll_fn)must("collision_detection");
    conv_fn conv = (conv_fn)must("check_convergence_async");
    err_fn last_error = (err_fn)must("pt_last_error");
    errs_fn last_error_string = (errs_fn)must("pt_last_error_string");
    /* out of this library's scope by design: a missing symbol is what sends the host to its CPU path */
    const char *absent[] = {"verlet_integrate", "marching_cubes", "hydraulic_erosion"};
    for (int i = 0; i < 3; i++) if (dlsym(lib, absent[i])) FAIL("%s must not be exported (a stub would feed the host garbage)", absent[i]);
    printf("bound: parallel_tempering_step collision_detection check_convergence_async; absent as documented: 3\n");

    /* Wrappers.available (src/GPU.Wrappers.fs:109-120): empty pinned arrays, zero counts */
    {
        double e0[1]; int e1[1];
        step(e0, e1, e0, e0, 0, 0, 0);
        step(NULL, NULL, NULL, NULL, 0, 0, 0);
        /* the probe computes nothing; what it can report is that the process has no GPU at all */
        if (last_error() != 0 && !(layout_only && last_error() == PT_ERR_NO_DEVICE)) FAIL("the probe reported error %d", last_error());
        printf("probe: ok%s\n", last_error() == PT_ERR_NO_DEVICE ? " (no GPU in this process)" : "");
    }
    /* the engine-side ABI binds and its config has the size the glue passes */
    {
        void (*defcfg)(pt_graph_config *, int32_t, int32_t, int32_t, int32_t) =
            (void (*)(pt_graph_config *, int32_t, int32_t, int32_t, int32_t))must("pt_graph_default_config");
        pt_graph_config cfg;
        memset(&cfg, 0xff, sizeof cfg);
        defcfg(&cfg, 100, 200, 4, 1);
        if (cfg.struct_size != (int32_t)sizeof cfg || cfg.node_count != 100 || cfg.weights[3] != 0.5 || cfg.min_distance != 0.5)
            FAIL("pt_graph_default_config filled an unexpected record");
        void (*cdef)(pt_chunk_config *, int32_t, int32_t) = (void (*)(pt_chunk_config *, int32_t, int32_t))must("pt_chunk_default_config");
        pt_chunk_config cc;
        cdef(&cc, 6, 2);
        if (cc.struct_size != (int32_t)sizeof cc || cc.temp_count != 6 || cc.refinement_probability != 0.7) FAIL("pt_chunk_default_config");
        printf("config records: ok\n");
    }
    if (layout_only) { printf("ABI_HARNESS_OK (layout only)\n"); return 0; }

    /* ---- GPUTempering.optimize (src/GPU.Wrappers.fs:291-342): one call per iteration on the same pinned arrays ---- */
    enum { N = 96, T = 4, E = 140, ITERS = 40 };
    static double pos[T * N * 3], pos0[T * N * 3], costs[T], temps[T] = {5.0, 2.3207944168063896, 1.0772173450159417, 0.5};
    static int edges[2 * E];
    for (int i = 0; i < N * 3; i++) { const double v = (rnd() - 0.5) * 2.9; for (int t = 0; t < T; t++) pos[t * N * 3 + i] = (double)(float)v; }
    for (int k = 0; k < E; k++) {
        if (k < N - 1) { edges[2 * k] = k; edges[2 * k + 1] = k + 1; }
        else { int a = (int)(rnd() * N), b = (int)(rnd() * N); if (a == b) b = (a + 1) % N; edges[2 * k] = a; edges[2 * k + 1] = b; }
    }
    memcpy(pos0, pos, sizeof pos);              /* the batched run below starts from the same graphs */
    if (dump) {
        FILE *fh = fopen(dump, "wb");
        const int32_t hdr[4] = {N, T, E, ITERS};
        if (!fh || fwrite(hdr, sizeof hdr, 1, fh) != 1 || fwrite(pos, sizeof pos, 1, fh) != 1 || fwrite(temps, sizeof temps, 1, fh) != 1 ||
            fwrite(edges, sizeof edges, 1, fh) != 1 || fclose(fh) != 0)
            FAIL("cannot write %s", dump);
    }
    for (int it = 0; it < ITERS; it++) {
        for (int t = 0; t < T; t++) costs[t] = -1.0;
        step(pos, edges, costs, temps, N, E, T);
        if (last_error() != 0) FAIL("parallel_tempering_step: %s", last_error_string());
        for (int t = 0; t < T; t++) if (!(costs[t] > 0.0) || !isfinite(costs[t])) FAIL("iteration %d: cost[%d] = %g", it, t, costs[t]);
    }
    printf("optimize: %d iterations, costs", ITERS);
    for (int t = 0; t < T; t++) printf(" %a", costs[t]);
    printf("\n");

    /* ---- GPUCollision.detect (src/GPU.Wrappers.fs:347-416): buffer of n * n / 2 pairs ---- */
    {
        static int pairs[N * N];                 /* (n * n / 2) pairs of two ints */
        int count = -1;
        collide(pos, pairs, N, 0.5, &count);
        if (last_error() != 0 || count < 0 || count > N * N / 2) FAIL("collision_detection: count %d, error %d", count, last_error());
        for (int k = 0; k < count; k++) {
            const int i = pairs[2 * k], j = pairs[2 * k + 1];
            if (!(0 <= i && i < j && j < N)) FAIL("pair %d = (%d, %d)", k, i, j);
            if (k && !(pairs[2 * k - 2] < i || (pairs[2 * k - 2] == i && pairs[2 * k - 1] < j))) FAIL("pairs are not sorted at %d", k);
            const float dx = (float)pos[3 * i] - (float)pos[3 * j], dy = (float)pos[3 * i + 1] - (float)pos[3 * j + 1],
                        dz = (float)pos[3 * i + 2] - (float)pos[3 * j + 2];
            const double d = (double)sqrtf((dx * dx + dy * dy) + dz * dz);
            if (!(d > 0.001 && d < 0.5)) FAIL("pair (%d, %d) at distance %g", i, j, d);
        }
        printf("collisions: %d pairs\n", count);
    }
    /* ---- check_convergence_async (src/GPU.Compute.fs:98-109): float[] is double*, bool[] 4-byte elements ---- */
    {
        enum { CH = 4, LEN = 120, SW = 30 };
        static double hist[CH * LEN];
        static int32_t swaps[SW];
        for (int c = 0; c < CH; c++) for (int k = 0; k < LEN; k++) hist[c * LEN + k] = 100.0 + 0.001 * rnd();
        for (int k = 0; k < SW; k++) swaps[k] = (k % 3 == 0);
        ConvergenceMetrics m;
        memset(&m, 0xff, sizeof m);
        conv(hist, swaps, CH, LEN, SW, 1.0, 77, &m, NULL);
        if (last_error() != 0) FAIL("check_convergence_async: %s", last_error_string());
        if (m.Iteration != 77 || !(m.AvgSwapAcceptance > 0.33f && m.AvgSwapAcceptance < 0.34f) || !(m.CostStabilization >= 0.0f) ||
            (m.HasConverged != 0 && m.HasConverged != 1))
            FAIL("check_convergence_async filled an unexpected record (iteration %d, swap rate %g)", m.Iteration, m.AvgSwapAcceptance);
        printf("convergence: R-hat %g ESS %g CV %g swap %g converged %d\n", m.GelmanRubin, m.EffectiveSampleSize, m.CostStabilization,
               m.AvgSwapAcceptance, m.HasConverged);
    }
    /* ---- the batched engine, as integration/GPU.Batched.fs drives it ---- */
    {
        void (*defcfg)(pt_graph_config *, int32_t, int32_t, int32_t, int32_t) =
            (void (*)(pt_graph_config *, int32_t, int32_t, int32_t, int32_t))must("pt_graph_default_config");
        int (*create)(const pt_graph_config *, pt_graph_engine **) = (int (*)(const pt_graph_config *, pt_graph_engine **))must("pt_graph_create");
        int (*upload)(pt_graph_engine *, const float *, const int32_t *, int32_t, const int32_t *, const uint8_t *, const double *, const double *) =
            (int (*)(pt_graph_engine *, const float *, const int32_t *, int32_t, const int32_t *, const uint8_t *, const double *, const double *))must("pt_graph_upload");
        int (*run)(pt_graph_engine *, int64_t) = (int (*)(pt_graph_engine *, int64_t))must("pt_graph_run");
        int (*download)(pt_graph_engine *, float *, int32_t *, int32_t, int32_t *, double *) =
            (int (*)(pt_graph_engine *, float *, int32_t *, int32_t, int32_t *, double *))must("pt_graph_download");
        void (*destroy)(pt_graph_engine *) = (void (*)(pt_graph_engine *))must("pt_graph_destroy");
        pt_graph_config cfg;
        defcfg(&cfg, N, 2 * E, T, 1);
        cfg.seed = "abi-harness";
        pt_graph_engine *eng = NULL;
        if (create(&cfg, &eng) != PT_OK) FAIL("pt_graph_create: %s", last_error_string());
        static float fpos[T * N * 3];
        static int32_t fedges[T * 2 * E * 2], ecount[T];
        static double out_cost[T];
        for (int i = 0; i < T * N * 3; i++) fpos[i] = (float)pos0[i];
        for (int t = 0; t < T; t++) { ecount[t] = E; memcpy(fedges + (size_t)t * 2 * E * 2, edges, sizeof edges); }
        if (upload(eng, fpos, fedges, 2 * E, ecount, NULL, NULL, temps) != PT_OK) FAIL("pt_graph_upload: %s", last_error_string());
        if (run(eng, 500) != PT_OK) FAIL("pt_graph_run: %s", last_error_string());
        if (download(eng, fpos, fedges, 2 * E, ecount, out_cost) != PT_OK) FAIL("pt_graph_download: %s", last_error_string());
        destroy(eng);
        printf("batched: 500 iterations, costs");
        for (int t = 0; t < T; t++) { if (!isfinite(out_cost[t])) FAIL("cost %d", t); printf(" %a", out_cost[t]); }
        printf(" edges");
        for (int t = 0; t < T; t++) printf(" %d", ecount[t]);
        printf("\n");
    }
    /* ---- the voxel-chunk engine, as TerrainBatched.generateFrom of integration/GPU.Batched.fs drives it ---- */
    {
        enum { CHUNKS = 2, CT = 6, V = PT_CHUNK_VOXELS, CITERS = 200 };
        void (*cdef)(pt_chunk_config *, int32_t, int32_t) = (void (*)(pt_chunk_config *, int32_t, int32_t))must("pt_chunk_default_config");
        int (*ccreate)(const pt_chunk_config *, pt_chunk_engine **) = (int (*)(const pt_chunk_config *, pt_chunk_engine **))must("pt_chunk_create");
        int (*cupload)(pt_chunk_engine *, const float *, const float *, const float *, const uint8_t *, const double *) =
            (int (*)(pt_chunk_engine *, const float *, const float *, const float *, const uint8_t *, const double *))must("pt_chunk_upload");
        int (*cpos)(pt_chunk_engine *, const int32_t *) = (int (*)(pt_chunk_engine *, const int32_t *))must("pt_chunk_set_positions");
        int (*crun)(pt_chunk_engine *, int64_t) = (int (*)(pt_chunk_engine *, int64_t))must("pt_chunk_run");
        int (*cdown)(pt_chunk_engine *, float *, float *, float *, double *) =
            (int (*)(pt_chunk_engine *, float *, float *, float *, double *))must("pt_chunk_download");
        void (*cdestroy)(pt_chunk_engine *) = (void (*)(pt_chunk_engine *))must("pt_chunk_destroy");
        int (*ladder)(double, double, int32_t, double *) = (int (*)(double, double, int32_t, double *))must("pt_geometric_ladder");
        const uint64_t seed = 0x5eedf00d12345678ull;                 /* Generation.generate's `seed: uint64` */
        static const int32_t xyz[CHUNKS * 3] = {3, -1, 7, 0, 0, 0};  /* ChunkPosition X, Y, Z */
        float *den = malloc(sizeof(float) * CHUNKS * CT * V), *moi = malloc(sizeof(float) * CHUNKS * CT * V),
              *tem = malloc(sizeof(float) * CHUNKS * CT * V);
        uint8_t *mat = malloc((size_t)CHUNKS * V);
        static double ctemps[CHUNKS * CT], ccost[CHUNKS * CT];
        if (!den || !moi || !tem || !mat) FAIL("out of memory");
        if (ladder(0.1, 5.0, CT, ctemps) != CT) FAIL("pt_geometric_ladder");     /* src/Terrain.Generation.fs:18-23, :181 */
        for (int c = 1; c < CHUNKS; c++) memcpy(ctemps + c * CT, ctemps, sizeof(double) * CT);
        /* a stand-in for initializeFromNoise (the noise stays on the host): a sloped surface with air pockets below */
        for (int c = 0; c < CHUNKS; c++)
            for (int i = 0; i < V; i++) {
                const int x = i & 31, y = (i >> 5) & 31, z = i >> 10;
                const double h = 10.0 + 6.0 * rnd() * 0.1 + 0.25 * x + (c ? 0.2 * z : -0.1 * z);
                float d = (double)y < h ? 1.0f : 0.0f;
                if (y <= 16 && rnd() < 0.04) d = 0.0f;
                const float m = (float)rnd(), tt = (float)(rnd() * 40.0 - 10.0);
                mat[(size_t)c * V + i] = (uint8_t)(d > 0.5f ? (y > 12 ? 2 : 1) + (m > 0.7f) : 0);
                for (int k = 0; k < CT; k++) {                       /* every temperature starts from the same chunk (:181-182) */
                    const size_t at = ((size_t)c * CT + k) * V + i;
                    den[at] = d; moi[at] = m; tem[at] = tt;
                }
            }
        if (dump) {
            char name[4096];
            snprintf(name, sizeof name, "%s.chunks", dump);
            FILE *fh = fopen(name, "wb");
            const int32_t hdr[4] = {CHUNKS, CT, CITERS, 50};
            int ok = fh && fwrite(hdr, sizeof hdr, 1, fh) == 1 && fwrite(&seed, sizeof seed, 1, fh) == 1 && fwrite(xyz, sizeof xyz, 1, fh) == 1;
            for (int f = 0; ok && f < 3; f++) {
                const float *src = f == 0 ? den : (f == 1 ? moi : tem);
                for (int c = 0; ok && c < CHUNKS; c++) ok = fwrite(src + (size_t)c * CT * V, sizeof(float), V, fh) == V;
            }
            ok = ok && fwrite(mat, 1, (size_t)CHUNKS * V, fh) == (size_t)CHUNKS * V && fwrite(ctemps, sizeof(double), CT, fh) == CT;
            if (!ok || fclose(fh) != 0) FAIL("cannot write %s", name);
        }
        pt_chunk_config cc;
        cdef(&cc, CT, CHUNKS);
        cc.swap_frequency = 50;                                      /* ptConfig.SwapFrequency */
        cc.philox_key = seed;
        pt_chunk_engine *ce = NULL;
        if (ccreate(&cc, &ce) != PT_OK) FAIL("pt_chunk_create: %s", last_error_string());
        if (cupload(ce, den, moi, tem, mat, ctemps) != PT_OK) FAIL("pt_chunk_upload: %s", last_error_string());
        if (cpos(ce, xyz) != PT_OK) FAIL("pt_chunk_set_positions: %s", last_error_string());
        if (crun(ce, CITERS) != PT_OK) FAIL("pt_chunk_run: %s", last_error_string());
        if (cdown(ce, den, moi, tem, ccost) != PT_OK) FAIL("pt_chunk_download: %s", last_error_string());
        cdestroy(ce);
        printf("terrain: %d chunks x %d iterations, costs", CHUNKS, CITERS);
        for (int k = 0; k < CHUNKS * CT; k++) { if (!isfinite(ccost[k])) FAIL("chunk cost %d", k); printf(" %a", ccost[k]); }
        /* what TerrainBatched.generateFrom keeps: the first chain of lowest cost of every ladder, and its voxels */
        printf(" best");
        for (int c = 0; c < CHUNKS; c++) {
            int best = 0;
            for (int k = 1; k < CT; k++) if (ccost[c * CT + k] < ccost[c * CT + best]) best = k;
            double sum = 0.0;
            for (int i = 0; i < V; i++) sum += (double)den[((size_t)c * CT + best) * V + i];
            printf(" %d:%a", best, sum);
        }
        printf("\n");
        free(den); free(moi); free(tem); free(mat);
    }
    printf("ABI_HARNESS_OK\n");
    return 0;
}

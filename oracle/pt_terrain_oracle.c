/*
 * pt_terrain_oracle.c -- see pt_terrain_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates reference src/Terrain.Constraints.fs:12-247 (cost), src/Terrain.Generation.fs:123-169
 * (mutateVoxels) and the PT loop of src/ParallelTempering.Core.fs:149-179 for voxel chunks.
 * Two summation modes, as in pt_oracle.c: OG_SUM_FSHARP follows the reference's sequential
 * sums in its own iteration order; OG_SUM_CANONICAL makes every sum exact (integer sums of
 * terms truncated to 2^-64), which is what a parallel implementation can reproduce bit for bit.
 */
#include "pt_terrain_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 fx128;
typedef struct { int mode; double d; fx128 q; int err; } tacc_t;

static void tacc_init(tacc_t *a, int mode) { a->mode = mode; a->d = 0.0; a->q = 0; a->err = 0; }
static void tacc_add(tacc_t *a, double t)
{
    if (a->mode == OG_SUM_FSHARP) { a->d = a->d + t; return; }
    double m = fabs(t);
    if (!(m < 0x1p62)) { a->err = 1; return; }
    double ip = floor(m), fp = m - ip;
    unsigned __int128 mag = ((unsigned __int128)(uint64_t)ip << 64) | (unsigned __int128)(uint64_t)(fp * 0x1p64);
    a->q += (t < 0.0) ? -(fx128)mag : (fx128)mag;
}
static double tacc_value(const tacc_t *a) { return a->mode == OG_SUM_FSHARP ? a->d : (double)a->q * 0x1p-64; }

void ot_default_config(ot_config *c)
{
    /* reference src/Terrain.Constraints.fs:238-247 */
    const double w[OT_W_COUNT] = {8.0, 3.0, 12.0, 2.0, 6.0, 4.0, 10.0, 2.0};
    memcpy(c->w, w, sizeof w);
    c->density_mutation_scale = 0.1f;       /* reference src/Terrain.Generation.fs:68-72 */
    c->refinement_probability = 0.7;
    c->solid_to_air_threshold = 0.5f;
    c->sum_mode = OG_SUM_FSHARP;
    c->res = 32;                            /* reference src/Voxel.Core.fs:74 */
}

static inline int vidx(int x, int y, int z, int res) { return x + y * res + z * res * res; }

/* reference src/Terrain.Constraints.fs:12-31: keep the lattice edges with |dy| / horizontal < 1
 * (fp32; a +y edge gives 0.5 / 0 = +inf and is dropped), then Graph.countReachableFrom 0. */
int ot_traversable_reach(int res)
{
    const int V = res * res * res;
    const float vs = 16.0f / (float)32;     /* ChunkSize.voxelSize, reference src/Voxel.Core.fs:75-76 */
    unsigned char *seen = calloc((size_t)V, 1);
    int *queue = malloc(sizeof(int) * (size_t)V);
    int head = 0, tail = 0, count = 0;
    seen[0] = 1; queue[tail++] = 0;
    while (head < tail) {
        int u = queue[head++];
        count++;
        int x = u % res, y = (u / res) % res, z = u / (res * res);
        const int nx[3] = {x + 1, x, x}, ny[3] = {y, y + 1, y}, nz[3] = {z, z, z + 1};
        for (int k = 0; k < 3; k++) {
            if (nx[k] >= res || ny[k] >= res || nz[k] >= res) continue;
            float hd = fabsf((float)y * vs - (float)ny[k] * vs);
            float dx = (float)x * vs - (float)nx[k] * vs, dz = (float)z * vs - (float)nz[k] * vs;
            float horiz = sqrtf(dx * dx + dz * dz);
            if (!(hd / horiz < 1.0f)) continue;
            int v = vidx(nx[k], ny[k], nz[k], res);
            if (!seen[v]) { seen[v] = 1; queue[tail++] = v; }
        }
    }
    free(seen); free(queue);
    return count;
}

/* reference src/Terrain.Constraints.fs:130-157 with Graph.countReachableFrom (src/Execution.fs:193-211):
 * directed search from voxel 0 over the +x/+y/+z edges whose both ends are air; voxel 0 counts as
 * reached whatever it is. */
static int cave_reach(const float *rho, int res)
{
    const int V = res * res * res;
    unsigned char *seen = calloc((size_t)V, 1);
    int *queue = malloc(sizeof(int) * (size_t)V);
    int head = 0, tail = 0, count = 0;
    seen[0] = 1; queue[tail++] = 0;
    while (head < tail) {
        int u = queue[head++];
        count++;
        if (!(rho[u] < 0.5f)) continue;
        int x = u % res, y = (u / res) % res, z = u / (res * res);
        const int nb[3] = {x + 1 < res ? u + 1 : -1, y + 1 < res ? u + res : -1, z + 1 < res ? u + res * res : -1};
        for (int k = 0; k < 3; k++) {
            int v = nb[k];
            if (v < 0 || seen[v] || !(rho[v] < 0.5f)) continue;
            seen[v] = 1; queue[tail++] = v;
        }
    }
    free(seen); free(queue);
    return count;
}

double ot_cost(const float *rho, const float *moist, const uint8_t *mat, const ot_config *c, ot_terms *out)
{
    const int res = c->res, V = res * res * res;
    ot_terms t;
    memset(&t, 0, sizeof t);

    /* traversability (:12-31): the slope filter keeps +x and +z edges only, so the search from voxel 0
     * reaches the y = 0 plane whatever the voxels hold */
    t.trav = (V <= 1) ? 0.0 : (double)(V - res * res) * 1000.0;

    /* geological (:35-41) and biome transitions (:202-210), in the order the edge list is built
     * (z, y, x; +x, +y, +z; reference src/Voxel.Core.fs:147-164) */
    tacc_t geo, bio;
    tacc_init(&geo, c->sum_mode); tacc_init(&bio, c->sum_mode);
    for (int z = 0; z < res; z++) for (int y = 0; y < res; y++) for (int x = 0; x < res; x++) {
        const int i = vidx(x, y, z, res);
        const int nb[3] = {x + 1 < res ? i + 1 : -1, y + 1 < res ? i + res : -1, z + 1 < res ? i + res * res : -1};
        for (int k = 0; k < 3; k++) {
            const int j = nb[k];
            if (j < 0) continue;
            float diff = fabsf(rho[i] - rho[j]);
            tacc_add(&geo, (double)(diff * diff));
            double b = 0.0;
            if (mat[i] != mat[j]) b = (rho[i] > 0.5f && rho[j] > 0.5f) ? 1.0 : 0.1;
            tacc_add(&bio, b);
        }
    }
    t.geo = tacc_value(&geo); t.biome = tacc_value(&bio);

    /* overhang (:45-64): columns (z, x), y = 1 .. res-1 */
    tacc_t ov;
    tacc_init(&ov, c->sum_mode);
    for (int z = 0; z < res; z++) for (int x = 0; x < res; x++) for (int y = 1; y < res; y++) {
        const int i = vidx(x, y, z, res), b = vidx(x, y - 1, z, res);
        if (rho[i] > 0.5f && rho[b] < 0.5f) tacc_add(&ov, (double)rho[i] * 10.0);
    }
    t.overhang = tacc_value(&ov);

    /* erosion (:67-99) and performance (:160-199): interior voxels */
    double ero = 0.0;
    int surface = 0;
    for (int z = 1; z <= res - 2; z++) for (int y = 1; y <= res - 2; y++) for (int x = 1; x <= res - 2; x++) {
        const int i = vidx(x, y, z, res);
        if (!(rho[i] > 0.5f)) continue;
        const int nb[6] = {i - 1, i + 1, i - res, i + res, i - res * res, i + res * res};
        int faces = 0;
        for (int k = 0; k < 6; k++) faces += rho[nb[k]] < 0.5f;
        if (faces > 0 && moist[i] < 0.3f) ero = ero + (double)faces * 0.5;      /* multiples of 0.5: exact in any order */
        if (faces > 0) surface++;
    }
    t.erosion = ero;
    t.surface = surface;
    {
        int est = surface * 12;
        t.performance = est > 10000 ? (double)(est - 10000) * 0.001 : 0.0;
    }

    /* aesthetics (:103-127): fp32 mean and variance of the heights of the solid voxels */
    {
        int solid = 0;
        for (int i = 0; i < V; i++) solid += rho[i] > 0.5f;
        t.solid = solid;
        if (solid == 0) t.aesthetics = 0.0;
        else {
            const float vs = 0.5f;
            float sum = 0.0f;                       /* Array.average: sequential fp32 sum (exact: multiples of 0.5 below 2^23) */
            for (int i = 0; i < V; i++) if (rho[i] > 0.5f) sum = sum + (float)((i / res) % res) * vs;
            const float avg = sum / (float)solid;
            float var;
            if (c->sum_mode == OG_SUM_FSHARP) {
                float vsum = 0.0f;                  /* Array.averageBy: sequential fp32 sum in node order */
                for (int i = 0; i < V; i++) if (rho[i] > 0.5f) {
                    float d = (float)((i / res) % res) * vs - avg;
                    vsum = vsum + d * d;
                }
                var = vsum / (float)solid;
            } else {
                /* canonical: the exact sum of the fp32 terms, rounded to double, then to float, once */
                fx128 q = 0;
                for (int i = 0; i < V; i++) if (rho[i] > 0.5f) {
                    float d = (float)((i / res) % res) * vs - avg;
                    float dd = d * d;
                    double m = (double)dd, ip = floor(m), fp = m - ip;
                    q += (fx128)(((unsigned __int128)(uint64_t)ip << 64) | (unsigned __int128)(uint64_t)(fp * 0x1p64));
                }
                var = (float)((double)q * 0x1p-64) / (float)solid;
            }
            t.aesthetics = var < 1.0f ? 10.0 * (double)(1.0f - var) : 0.0;
        }
    }

    /* caves (:130-157) */
    {
        int ug = 0;
        for (int z = 0; z < res; z++) for (int y = 0; y <= res / 2 && y < res; y++) for (int x = 0; x < res; x++)
            ug += rho[vidx(x, y, z, res)] < 0.5f;
        t.underground_air = ug;
        t.reached = cave_reach(rho, res);
        t.caves = ug == 0 ? 0.0 : ((V <= 1) ? 0.0 : (double)(V - t.reached) * 1000.0) * 0.5;
    }

    /* Σ w c in the order of the list (:224-235) */
    double total = 0.0;
    total = total + c->w[OT_W_TRAVERSABILITY] * t.trav;
    total = total + c->w[OT_W_GEOLOGICAL] * t.geo;
    total = total + c->w[OT_W_OVERHANG] * t.overhang;
    total = total + c->w[OT_W_EROSION] * t.erosion;
    total = total + c->w[OT_W_AESTHETICS] * t.aesthetics;
    total = total + c->w[OT_W_CAVES] * t.caves;
    total = total + c->w[OT_W_PERFORMANCE] * t.performance;
    total = total + c->w[OT_W_BIOME] * t.biome;
    t.total = total;
    t.range_error = geo.err | bio.err | ov.err;
    if (out) *out = t;
    return total;
}

static double tdraw(og_stream *s, int k)
{
    if (s->stride > 0) return og_stream_at(s, s->step_base + k);
    return og_stream_next(s);
}

typedef struct { float *rho, *moist, *temp; double cost; double T; } ot_chain;

/* one optimizationStep (reference src/ParallelTempering.Core.fs:122-133) with mutateVoxels
 * (reference src/Terrain.Generation.fs:123-169) */
static void ot_step(ot_chain *ch, const uint8_t *mat, const ot_config *c, og_stream *s, ot_trace_rec *tr)
{
    const int V = c->res * c->res * c->res;
    const int idx = (int)(tdraw(s, 0) * (double)V);
    const float rho = ch->rho[idx], temp = ch->temp[idx], moist = ch->moist[idx];
    const float amount = (float)ch->T * c->density_mutation_scale;
    float nrho;
    int kind;
    if (tdraw(s, 1) < c->refinement_probability) {
        kind = 0;
        const float d = ((float)tdraw(s, 2) - 0.5f) * amount;
        nrho = rho + d;
    } else {
        kind = 1;
        nrho = rho > c->solid_to_air_threshold ? 0.0f : 1.0f;
    }
    nrho = fmaxf(0.0f, fminf(1.0f, nrho));
    float ntemp = temp, nmoist = moist;
    int sub = 0;
    if (tdraw(s, 3) < 0.1) { ntemp = temp + ((float)tdraw(s, 4) - 0.5f) * 5.0f; sub |= 1; }
    if (tdraw(s, 5) < 0.1) { nmoist = fmaxf(0.0f, fminf(1.0f, moist + ((float)tdraw(s, 6) - 0.5f) * 0.2f)); sub |= 2; }

    ch->rho[idx] = nrho; ch->moist[idx] = nmoist; ch->temp[idx] = ntemp;
    ot_terms tm;
    const double pc = ot_cost(ch->rho, ch->moist, mat, c, &tm);
    const double delta = pc - ch->cost;
    double u = -1.0;
    int drew = 0, acc;
    if (delta < 0.0) acc = 1;                                   /* Acceptance.acceptMove, :32-34 */
    else { u = tdraw(s, 7); drew = 1; acc = u < exp(-delta / ch->T); }
    if (acc) ch->cost = pc;
    else { ch->rho[idx] = rho; ch->moist[idx] = moist; ch->temp[idx] = temp; }
    if (tr) {
        tr->kind = kind; tr->a = idx; tr->b = sub;
        tr->flags = (drew ? 2 : 0) | (acc ? 4 : 0);
        tr->reach0 = tm.reached; tr->reach_origin = tm.underground_air;
        tr->delta = delta; tr->u = u; tr->cost_after = ch->cost;
    }
}

int ot_run(const ot_run_args *a, const ot_config *c)
{
    const int T = a->temps, V = c->res * c->res * c->res;
    ot_chain *ch = calloc((size_t)T, sizeof(ot_chain));
    int err = 0;
    for (int t = 0; t < T; t++) {
        ch[t].rho = a->density + (size_t)t * V; ch[t].moist = a->moisture + (size_t)t * V;
        ch[t].temp = a->temperature + (size_t)t * V;
        ch[t].T = a->ladder[t];
        ot_terms tm;
        ch[t].cost = ot_cost(ch[t].rho, ch[t].moist, a->material, c, &tm);
        err |= tm.range_error;
    }
    int64_t passes = 0;
    const int threads = a->threads > 0 ? a->threads : 1;
    (void)threads;
    for (int64_t step = a->first_step; step < a->first_step + a->iters; step++) {
        for (int t = 0; t < T; t++) a->chain_streams[t].step_base = step * a->chain_streams[t].stride;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) if (threads > 1)
#endif
        for (int t = 0; t < T; t++)
            ot_step(&ch[t], a->material, c, &a->chain_streams[t], a->trace ? &a->trace[(size_t)(step - a->first_step) * T + t] : NULL);
        if (a->swap_freq > 0 && step % a->swap_freq == 0) {
            /* attemptSwap over adjacent pairs of the UPDATED array (:168-175): states and costs move, temperatures stay */
            for (int i = 0; i + 1 < T; i++) {
                double p = og_swap_probability(ch[i].cost, ch[i].T, ch[i + 1].cost, ch[i + 1].T);
                if (p > 1.0) p = 1.0;
                const double u = og_stream_next(a->swap_stream);
                const int sw = u < p;
                if (sw) {
                    ot_chain x = ch[i];
                    ch[i].rho = ch[i + 1].rho; ch[i].moist = ch[i + 1].moist; ch[i].temp = ch[i + 1].temp; ch[i].cost = ch[i + 1].cost;
                    ch[i + 1].rho = x.rho; ch[i + 1].moist = x.moist; ch[i + 1].temp = x.temp; ch[i + 1].cost = x.cost;
                }
                if (a->swap_trace) a->swap_trace[(size_t)passes * (T - 1) + i] = (uint8_t)sw;
            }
            passes++;
        }
    }
    /* hand the states back in slot order */
    float *buf = malloc(sizeof(float) * (size_t)T * V * 3);
    for (int t = 0; t < T; t++) {
        memcpy(buf + ((size_t)t * 3 + 0) * V, ch[t].rho, sizeof(float) * V);
        memcpy(buf + ((size_t)t * 3 + 1) * V, ch[t].moist, sizeof(float) * V);
        memcpy(buf + ((size_t)t * 3 + 2) * V, ch[t].temp, sizeof(float) * V);
        a->cost[t] = ch[t].cost;
    }
    for (int t = 0; t < T; t++) {
        memcpy(a->density + (size_t)t * V, buf + ((size_t)t * 3 + 0) * V, sizeof(float) * V);
        memcpy(a->moisture + (size_t)t * V, buf + ((size_t)t * 3 + 1) * V, sizeof(float) * V);
        memcpy(a->temperature + (size_t)t * V, buf + ((size_t)t * 3 + 2) * V, sizeof(float) * V);
    }
    free(buf);
    for (int t = 0; t < T; t++) if (a->chain_streams[t].exhausted) err |= 2;
    if (a->swap_stream && a->swap_stream->exhausted) err |= 2;
    if (a->n_swap_passes) *a->n_swap_passes = passes;
    free(ch);
    return err;
}

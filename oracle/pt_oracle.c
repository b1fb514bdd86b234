/*
 * pt_oracle.c -- CPU restatement of the reference's parallel-tempering CPU
 * path for spatial graphs.  TEST INFRASTRUCTURE ONLY (see pt_oracle.h).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: the reference's .NET JIT never fuses a*b+c.
 *
 * Two summation modes:
 *   OG_SUM_FSHARP     sequential fp64 sums in the reference's list/array order
 *                     and the sequential fp32 centroid fold, as the F# does.
 *   OG_SUM_CANONICAL  every sum is an exact integer sum of terms truncated to
 *                     a multiple of 2^-64 (order independent), converted back
 *                     to fp64 once.  This is the definition a parallel
 *                     implementation can reproduce bit for bit.
 * Two algorithm modes (same results, different complexity):
 *   OG_ALGO_FAITHFUL  the reference's own algorithms: BFS that scans the whole
 *                     edge list per dequeued node, O(n^2) collisions below 200
 *                     nodes and a dictionary spatial hash above, O(N) position
 *                     copy per proposal.
 *   OG_ALGO_OPTIMISED CSR BFS and a sorted uniform grid.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ======================================================================== */
/* SHA-256 (FIPS 180-4), used by HashRNG (Deterministic.fs:14-17)            */
/* ======================================================================== */

static const uint32_t K256[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};

static uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

static void sha256_block(uint32_t h[8], const uint8_t *p)
{
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4*i] << 24) | ((uint32_t)p[4*i+1] << 16) | ((uint32_t)p[4*i+2] << 8) | p[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i-15], 7) ^ rotr32(w[i-15], 18) ^ (w[i-15] >> 3);
        uint32_t s1 = rotr32(w[i-2], 17) ^ rotr32(w[i-2], 19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    uint32_t a=h[0],b=h[1],c=h[2],d=h[3],e=h[4],f=h[5],g=h[6],hh=h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0]+=a; h[1]+=b; h[2]+=c; h[3]+=d; h[4]+=e; h[5]+=f; h[6]+=g; h[7]+=hh;
}

void og_sha256(const uint8_t *data, size_t len, uint8_t out[32])
{
    uint32_t h[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    size_t full = len / 64;
    for (size_t i = 0; i < full; i++) sha256_block(h, data + 64*i);
    uint8_t tail[128];
    size_t rem = len - 64*full;
    memset(tail, 0, sizeof tail);
    memcpy(tail, data + 64*full, rem);
    tail[rem] = 0x80;
    size_t tl = (rem < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8*i));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4*i] = (uint8_t)(h[i] >> 24); out[4*i+1] = (uint8_t)(h[i] >> 16);
        out[4*i+2] = (uint8_t)(h[i] >> 8); out[4*i+3] = (uint8_t)h[i];
    }
}

static uint64_t le_u64(const uint8_t *b)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | b[i];
    return v;
}

/* Deterministic.fs:14-27.  generateBytes hashes utf8(seed + index.ToString());
 * bytesToFloat divides the first 8 bytes (BitConverter.ToUInt64, little endian)
 * by float UInt64.MaxValue, which is 2^64 as a double -- so the result can be 1.0. */
double og_hashrng_next_float(const char *seed, int index)
{
    char buf[1024];
    int n = snprintf(buf, sizeof buf, "%s%d", seed, index);
    uint8_t dg[32];
    og_sha256((const uint8_t *)buf, (size_t)n, dg);
    return (double)le_u64(dg) / 18446744073709551616.0;
}

/* Deterministic.fs:38-40 */
int og_hashrng_next_int(const char *seed, int index, int max)
{
    return (int)(og_hashrng_next_float(seed, index) * (double)max);
}

/* Deterministic.fs:94-102.  `uint64 x` of an int sign-extends. */
uint64_t og_hash_coordinates(uint64_t base_seed, int x, int y, int z)
{
    uint64_t h = base_seed;
    h ^= (uint64_t)(int64_t)x * 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h ^= (uint64_t)(int64_t)y * 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    h ^= (uint64_t)(int64_t)z * 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    return h;
}

uint64_t og_seed_key(const char *seed)
{
    uint8_t dg[32];
    og_sha256((const uint8_t *)seed, strlen(seed), dg);
    return le_u64(dg);
}

/* ======================================================================== */
/* Philox4x32-10 (Salmon et al., SC'11) -- the counter-based generator that  */
/* stands in for the reference's thread-local System.Random                  */
/* (ThreadSafe.fs:6-21), whose draw order across threads is not reproducible */
/* ======================================================================== */

void og_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void og_stream_array(og_stream *s, const double *u, int64_t len)
{
    memset(s, 0, sizeof *s);
    s->mode = 0; s->u = u; s->len = len;
}

void og_stream_philox(og_stream *s, uint64_t key, uint32_t stream_id, uint32_t domain)
{
    memset(s, 0, sizeof *s);
    s->mode = 1;
    s->key[0] = (uint32_t)key; s->key[1] = (uint32_t)(key >> 32);
    s->stream_id = stream_id; s->domain = domain;
}

double og_stream_at(og_stream *s, int64_t p)
{
    if (s->mode == 0) {
        if (p >= s->len) { s->exhausted = 1; return 0.0; }
        return s->u[p];
    }
    uint64_t q = (uint64_t)p >> 1;
    uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), s->stream_id, s->domain};
    uint32_t w[4];
    og_philox4x32_10(ctr, s->key, w);
    uint32_t lo = (p & 1) ? w[2] : w[0];
    uint32_t hi = (p & 1) ? w[3] : w[1];
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

double og_stream_next(og_stream *s) { return og_stream_at(s, s->pos++); }

/* k-th kind of draw of the current step (see og_stream in pt_oracle.h) */
static double draw(og_stream *s, int k)
{
    if (s->stride > 0) return og_stream_at(s, s->step_base + k);
    return og_stream_next(s);
}

/* nextInt k = floor(u*k); nextSingle = (float)u (ThreadSafe.fs:19-21 with the
 * draws injected instead of coming from System.Random). */
static int next_int(og_stream *s, int addr, int k) { return (int)(draw(s, addr) * (double)k); }
static float next_single(og_stream *s, int addr) { return (float)draw(s, addr); }

/* ======================================================================== */
/* scalar pieces                                                             */
/* ======================================================================== */

/* ParallelTempering.Core.fs:88-95 */
void og_geometric_ladder(double tmin, double tmax, int n, double *out, int *out_n)
{
    if (n < 2 || tmin <= 0.0 || tmax <= 0.0 || tmin > tmax) { out[0] = tmax; *out_n = 1; return; }
    double ratio = tmax / tmin;
    for (int i = 0; i < n; i++) out[i] = tmax * pow(ratio, -(double)i / (double)(n - 1));
    *out_n = n;
}

/* ParallelTempering.Core.fs:138-139 */
int og_swap_frequency(int node_count)
{
    int f = (int)(sqrt((double)node_count) * 2.0);
    return f > 5 ? f : 5;
}

/* The probability behind Acceptance.acceptMove (ParallelTempering.Core.fs:32-34):
 * 1 when delta < 0, else exp(-delta/T). */
double og_accept_probability(double current, double proposed, double temperature)
{
    double delta = proposed - current;
    if (delta < 0.0) return 1.0;
    return exp(-delta / temperature);
}

/* ParallelTempering.Core.fs:103-109 (not yet clamped to 1) */
double og_swap_probability(double c1, double t1, double c2, double t2)
{
    if (t1 <= 0.0 || t2 <= 0.0) return 0.0;
    double beta1 = 1.0 / t1, beta2 = 1.0 / t2;
    double dc = c2 - c1;
    return exp(dc * (beta1 - beta2));
}

/* ProceduralGeneration.fs:17-33 */
void og_uniform_sphere(int count, float *pos)
{
    if (count <= 0) return;
    if (count == 1) { pos[0] = pos[1] = pos[2] = 0.0f; return; }
    double golden = M_PI * (3.0 - sqrt(5.0));
    for (int i = 0; i < count; i++) {
        double y = 1.0 - ((double)i / (double)(count - 1)) * 2.0;
        double radius = sqrt(1.0 - y * y);
        double theta = (double)i * golden;
        pos[3*i]   = (float)(radius * cos(theta));
        pos[3*i+1] = (float)y;
        pos[3*i+2] = (float)(radius * sin(theta));
    }
}

/* ======================================================================== */
/* Deterministic.fs: the seeded workload generators                          */
/* ======================================================================== */

static inline float dist3(const float *p, const float *q);

/* Deterministic.fs:30-35.  HashRNG.nextVector3: three draws at 3i, 3i+1, 3i+2, each mapped to
 * [-1, 1] in fp64 and rounded to fp32. */
static void hashrng_next_vector3(const char *seed, int index, float out[3])
{
    for (int k = 0; k < 3; k++)
        out[k] = (float)(og_hashrng_next_float(seed, index * 3 + k) * 2.0 - 1.0);
}

/* Deterministic.fs:44-68.  DeterministicDistribution.fibonacciSphere: theta = phi * i * 2 * pi in
 * that order of multiplication; the perturbation is applied in fp32 (Vector3 arithmetic). */
void og_fibonacci_sphere(const char *seed, int count, double perturbation, float *pos)
{
    if (count <= 0) return;
    if (count == 1) { pos[0] = pos[1] = pos[2] = 0.0f; return; }
    const double phi = (1.0 + sqrt(5.0)) / 2.0;
    for (int i = 0; i < count; i++) {
        double y = 1.0 - ((double)i / (double)(count - 1)) * 2.0;
        double radius = sqrt(1.0 - y * y);
        double theta = phi * (double)i * 2.0 * M_PI;
        float b[3] = { (float)(radius * cos(theta)), (float)y, (float)(radius * sin(theta)) };
        if (perturbation > 0.0) {
            float nz[3];
            hashrng_next_vector3(seed, i, nz);
            float s = (float)perturbation;
            for (int k = 0; k < 3; k++) {
                float t = nz[k] * s;
                b[k] = b[k] + t;
            }
        }
        pos[3*i] = b[0]; pos[3*i+1] = b[1]; pos[3*i+2] = b[2];
    }
}

/* Deterministic.fs:71-87.  DeterministicDistribution.uniformSphere (points IN a ball of the radius). */
void og_seeded_uniform_sphere(const char *seed, int count, double radius, float *pos)
{
    for (int i = 0; i < count; i++) {
        double u = og_hashrng_next_float(seed, i * 3);
        double v = og_hashrng_next_float(seed, i * 3 + 1);
        double w = og_hashrng_next_float(seed, i * 3 + 2);
        double theta = 2.0 * M_PI * u;
        double ph = acos(2.0 * v - 1.0);
        double r = pow(w, 1.0 / 3.0) * radius;
        pos[3*i]   = (float)(r * sin(ph) * cos(theta));
        pos[3*i+1] = (float)(r * sin(ph) * sin(theta));
        pos[3*i+2] = (float)(r * cos(ph));
    }
}

/* Deterministic.fs:105-116.  DeterministicGeneration.createGraph: node 0 Origin, n-1 Terminal (0 wins
 * when n = 1), the others Standard (nextFloat seed (i + 1000)); no edges. */
void og_create_graph(const char *seed, int n, double perturbation, float *pos, uint8_t *kind, double *value)
{
    og_fibonacci_sphere(seed, n, perturbation, pos);
    for (int i = 0; i < n; i++) {
        if (i == 0) { kind[i] = OG_ORIGIN; value[i] = 0.0; }
        else if (i == n - 1) { kind[i] = OG_TERMINAL; value[i] = 0.0; }
        else { kind[i] = OG_STANDARD; value[i] = og_hashrng_next_float(seed, i + 1000); }
    }
}

/* Deterministic.fs:119-135.  DeterministicGeneration.generateEdges: pairs i < j in row order, kept when
 * nextFloat seed (i*1000 + j) < 1 / (1 + d^2) until int(n * avgDegree / 2) edges exist (the reference
 * goes on hashing the remaining pairs without effect; this stops).  Every new edge is consed onto the
 * list, so the output (LIST order, index 0 = head) is the reverse of the order of discovery.
 * Returns the number of edges; `weights` (the fp32 distance widened) may be NULL. */
int og_generate_edges(const char *seed, const float *pos, int n, double avg_degree, int32_t *edges, double *weights, int cap)
{
    int target = (int)((double)n * avg_degree / 2.0);
    int cnt = 0;
    for (int i = 0; i < n && cnt < target; i++)
        for (int j = i + 1; j < n && cnt < target; j++) {
            float d = dist3(pos + 3*i, pos + 3*j);
            double probability = 1.0 / (1.0 + (double)d * (double)d);
            double threshold = og_hashrng_next_float(seed, (int)((uint32_t)i * 1000u + (uint32_t)j));
            if (threshold < probability) {
                if (cnt >= cap) return -1;
                edges[2*cnt] = i; edges[2*cnt+1] = j;
                if (weights) weights[cnt] = (double)d;
                cnt++;
            }
        }
    for (int a = 0, b = cnt - 1; a < b; a++, b--) {
        int32_t t0 = edges[2*a], t1 = edges[2*a+1];
        edges[2*a] = edges[2*b]; edges[2*a+1] = edges[2*b+1];
        edges[2*b] = t0; edges[2*b+1] = t1;
        if (weights) { double w = weights[a]; weights[a] = weights[b]; weights[b] = w; }
    }
    return cnt;
}

void og_default_config(og_config *c)
{
    /* ProceduralGeneration.fs:100-107 */
    c->w[0] = 1.0; c->w[1] = 1.0; c->w[2] = 0.1; c->w[3] = 0.5; c->w[4] = 0.05; c->w[5] = 0.1;
    c->target_len = 2.0;        /* ProceduralGeneration.fs:77 */
    c->min_dist = 0.5;          /* ProceduralGeneration.fs:78 */
    c->eps = 0.001;             /* SpatialHash.fs:18 */
    c->sum_mode = OG_SUM_FSHARP;
    c->algo = OG_ALGO_FAITHFUL;
    c->mutation_mode = OG_MUT_ADAPTIVE;
    c->brute_threshold = 200;   /* SpatialHash.fs:17 */
}

/* ======================================================================== */
/* sums                                                                      */
/* ======================================================================== */

typedef __int128 fx128;

typedef struct { int mode; double d; fx128 q; int err; } acc_t;

static void acc_init(acc_t *a, int mode) { a->mode = mode; a->d = 0.0; a->q = 0; a->err = 0; }

/* canonical: add trunc(t * 2^64) as an integer; |t| must be < 2^62 */
static void acc_add(acc_t *a, double t)
{
    if (a->mode == OG_SUM_FSHARP) { a->d = a->d + t; return; }
    double m = fabs(t);
    if (!(m < 0x1p62)) { a->err = 1; return; }
    double ip = floor(m);
    double fp = m - ip;                     /* exact */
    unsigned __int128 mag = ((unsigned __int128)(uint64_t)ip << 64) | (unsigned __int128)(uint64_t)(fp * 0x1p64);
    a->q += (t < 0.0) ? -(fx128)mag : (fx128)mag;
}

static double acc_value(const acc_t *a)
{
    if (a->mode == OG_SUM_FSHARP) return a->d;
    return (double)a->q * 0x1p-64;          /* one correctly rounded conversion, exact scaling */
}

/* System.Numerics.Vector3.Distance: fp32, sqrt((dx*dx + dy*dy) + dz*dz) */
static inline float dist3(const float *p, const float *q)
{
    float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    float s = dx * dx + dy * dy;
    s = s + dz * dz;
    return sqrtf(s);
}

/* ======================================================================== */
/* BFS (Execution.fs:193-211): directed, follows a->b only                   */
/* ======================================================================== */

int og_count_reachable(int n, const int32_t *edges, int e, int start, int algo)
{
    if (n <= 0 || start < 0 || start >= n) return 0;
    uint8_t *visited = calloc((size_t)n, 1);
    int32_t *queue = malloc(sizeof(int32_t) * (size_t)n);
    int head = 0, tail = 0, count = 0;
    queue[tail++] = start; visited[start] = 1;
    if (algo == OG_ALGO_FAITHFUL) {
        while (head < tail) {
            int node = queue[head++];
            for (int k = 0; k < e; k++) {           /* List.iter over the whole edge list */
                if (edges[2*k] == node) {
                    int nb = edges[2*k+1];
                    if (nb >= 0 && !visited[nb]) { visited[nb] = 1; queue[tail++] = nb; }
                }
            }
        }
    } else {
        int32_t *rowptr = calloc((size_t)n + 1, sizeof(int32_t));
        int32_t *adj = malloc(sizeof(int32_t) * (size_t)(e > 0 ? e : 1));
        for (int k = 0; k < e; k++) rowptr[edges[2*k] + 1]++;
        for (int i = 0; i < n; i++) rowptr[i+1] += rowptr[i];
        int32_t *fill = malloc(sizeof(int32_t) * (size_t)n);
        memcpy(fill, rowptr, sizeof(int32_t) * (size_t)n);
        for (int k = 0; k < e; k++) adj[fill[edges[2*k]]++] = edges[2*k+1];
        while (head < tail) {
            int node = queue[head++];
            for (int k = rowptr[node]; k < rowptr[node+1]; k++) {
                int nb = adj[k];
                if (!visited[nb]) { visited[nb] = 1; queue[tail++] = nb; }
            }
        }
        free(rowptr); free(adj); free(fill);
    }
    for (int i = 0; i < n; i++) count += visited[i];
    free(visited); free(queue);
    return count;
}

/* ======================================================================== */
/* collisions (SpatialHash.fs:89-191)                                        */
/* ======================================================================== */

typedef struct { int32_t i, j; } pair_t;
typedef struct { pair_t *p; int64_t n, cap; } pairvec;

static void pv_push(pairvec *v, int i, int j)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = realloc(v->p, sizeof(pair_t) * (size_t)v->cap); }
    v->p[v->n].i = i; v->p[v->n].j = j; v->n++;
}

/* SpatialHash.fs:89-101.  Pairs are prepended to an F# list, so the list the
 * penalty walks is the reverse of discovery order; pv holds discovery order. */
static void brute_pairs(const float *pos, int n, double min_dist, double eps, pairvec *pv)
{
    for (int i = 0; i + 1 < n; i++)
        for (int j = i + 1; j < n; j++) {
            double d = (double)dist3(pos + 3*j, pos + 3*i);
            if (d < min_dist && d > eps) pv_push(pv, i, j);
        }
}

/* SpatialHash.detectCollisions as a pair list (brute force, :89-101), pairs sorted by (i, j).
 * Returns the number of pairs found; writes at most cap of them. */
int64_t og_collision_pairs(const float *pos, int n, double min_dist, double eps, int32_t *out, int64_t cap)
{
    pairvec pv = {0};
    brute_pairs(pos, n, min_dist, eps, &pv);
    for (int64_t k = 0; k < pv.n && k < cap; k++) { out[2 * k] = pv.p[k].i; out[2 * k + 1] = pv.p[k].j; }
    int64_t found = pv.n;
    free(pv.p);
    return found;
}

static inline int cell_of(float x, double cell) { return (int)((double)x / cell); }  /* truncation, SpatialHash.fs:113-116 */

static inline uint32_t cell_hash(int cx, int cy, int cz)
{
    uint32_t h = (uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^ (uint32_t)cz * 83492791u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    return h;
}

/* SpatialHash.fs:104-146: Dictionary<(int,int,int), int list>, each cell's list
 * built by consing (so it enumerates largest index first), 27-neighbourhood
 * walked dx,dy,dz, keep i<j.  The dictionary is only ever probed by key, so an
 * open-addressing table reproduces it exactly. */
static void hash_pairs(const float *pos, int n, double min_dist, double eps, pairvec *pv)
{
    double cell = min_dist;
    uint32_t cap = 16; while (cap < (uint32_t)n * 2u) cap <<= 1;
    int32_t *slot_head = malloc(sizeof(int32_t) * cap);
    int32_t *slot_key = malloc(sizeof(int32_t) * 3 * cap);
    int32_t *next = malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (uint32_t s = 0; s < cap; s++) slot_head[s] = -2;   /* -2 = empty slot */
    for (int i = 0; i < n; i++) {
        int cx = cell_of(pos[3*i], cell), cy = cell_of(pos[3*i+1], cell), cz = cell_of(pos[3*i+2], cell);
        uint32_t s = cell_hash(cx, cy, cz) & (cap - 1);
        for (;;) {
            if (slot_head[s] == -2) { slot_key[3*s] = cx; slot_key[3*s+1] = cy; slot_key[3*s+2] = cz; slot_head[s] = -1; break; }
            if (slot_key[3*s] == cx && slot_key[3*s+1] == cy && slot_key[3*s+2] == cz) break;
            s = (s + 1) & (cap - 1);
        }
        next[i] = slot_head[s]; slot_head[s] = i;      /* i :: grid.[cell] */
    }
    for (int i = 0; i < n; i++) {
        int cx = cell_of(pos[3*i], cell), cy = cell_of(pos[3*i+1], cell), cz = cell_of(pos[3*i+2], cell);
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) {
                    int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                    uint32_t s = cell_hash(nx, ny, nz) & (cap - 1);
                    int found = -1;
                    for (;;) {
                        if (slot_head[s] == -2) break;
                        if (slot_key[3*s] == nx && slot_key[3*s+1] == ny && slot_key[3*s+2] == nz) { found = (int)s; break; }
                        s = (s + 1) & (cap - 1);
                    }
                    if (found < 0) continue;
                    for (int j = slot_head[found]; j >= 0; j = next[j])
                        if (i < j) {
                            double d = (double)dist3(pos + 3*j, pos + 3*i);
                            if (d < min_dist && d > eps) pv_push(pv, i, j);
                        }
                }
    }
    free(slot_head); free(slot_key); free(next);
}

/* Optimised: counting sort of nodes by floor cell over the bounding box when it
 * is small enough, hashed cells otherwise; half-shell neighbour walk.  Same
 * pair SET as the reference (any two points closer than min_dist lie in
 * adjacent cells of side min_dist). */
static void grid_pairs(const float *pos, int n, double min_dist, double eps, pairvec *pv)
{
    if (n < 2) return;
    double cell = min_dist;
    int32_t *cx = malloc(sizeof(int32_t) * 3 * (size_t)n);
    uint32_t cap = 16; while (cap < (uint32_t)n * 2u) cap <<= 1;
    int32_t *slot_head = malloc(sizeof(int32_t) * cap);
    int32_t *slot_node = malloc(sizeof(int32_t) * cap);   /* representative node carrying the key */
    int32_t *next = malloc(sizeof(int32_t) * (size_t)n);
    for (uint32_t s = 0; s < cap; s++) slot_head[s] = -2;
    for (int i = 0; i < n; i++) {
        int c0 = (int)floor((double)pos[3*i] / cell), c1 = (int)floor((double)pos[3*i+1] / cell), c2 = (int)floor((double)pos[3*i+2] / cell);
        cx[3*i] = c0; cx[3*i+1] = c1; cx[3*i+2] = c2;
        uint32_t s = cell_hash(c0, c1, c2) & (cap - 1);
        for (;;) {
            if (slot_head[s] == -2) { slot_node[s] = i; slot_head[s] = -1; break; }
            int r = slot_node[s];
            if (cx[3*r] == c0 && cx[3*r+1] == c1 && cx[3*r+2] == c2) break;
            s = (s + 1) & (cap - 1);
        }
        next[i] = slot_head[s]; slot_head[s] = i;
    }
    for (int i = 0; i < n; i++) {
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dz = -1; dz <= 1; dz++) {
                    int nx = cx[3*i] + dx, ny = cx[3*i+1] + dy, nz = cx[3*i+2] + dz;
                    uint32_t s = cell_hash(nx, ny, nz) & (cap - 1);
                    int found = -1;
                    for (;;) {
                        if (slot_head[s] == -2) break;
                        int r = slot_node[s];
                        if (cx[3*r] == nx && cx[3*r+1] == ny && cx[3*r+2] == nz) { found = (int)s; break; }
                        s = (s + 1) & (cap - 1);
                    }
                    if (found < 0) continue;
                    for (int j = slot_head[found]; j >= 0; j = next[j])
                        if (i < j) {
                            double d = (double)dist3(pos + 3*j, pos + 3*i);
                            if (d < min_dist && d > eps) pv_push(pv, i, j);
                        }
                }
    }
    free(cx); free(slot_head); free(slot_node); free(next);
}

/* SpatialHash.fs:151-191 without the 5 % dual-measurement branch (:157-176),
 * which only changes timing and list order. */
double og_collision_penalty(const float *pos, int n, const og_config *c, int64_t *npairs, int *err)
{
    pairvec pv = {0, 0, 0};
    if (c->algo == OG_ALGO_FAITHFUL) {
        if (n < c->brute_threshold) brute_pairs(pos, n, c->min_dist, c->eps, &pv);
        else hash_pairs(pos, n, c->min_dist, c->eps, &pv);
    } else {
        grid_pairs(pos, n, c->min_dist, c->eps, &pv);
    }
    acc_t a; acc_init(&a, c->sum_mode);
    /* List.sumBy walks the list head first = reverse discovery order */
    for (int64_t k = pv.n - 1; k >= 0; k--) {
        double d = (double)dist3(pos + 3*pv.p[k].i, pos + 3*pv.p[k].j);
        double overlap = c->min_dist - d;
        acc_add(&a, overlap * overlap * 100.0);
    }
    if (npairs) *npairs = pv.n;
    if (err && a.err) *err = 1;
    free(pv.p);
    return acc_value(&a);
}

/* ======================================================================== */
/* cost (ProceduralGeneration.fs:73-81 over ParallelTempering.Core.fs:241-309)*/
/* ======================================================================== */

double og_cost(const og_graph *g, const og_config *c, og_terms *out)
{
    og_terms t; memset(&t, 0, sizeof t);
    int n = g->n, e = g->e;
    int err = 0;

    /* Constraints.connectivity, ParallelTempering.Core.fs:248-254 */
    if (n <= 1) { t.conn = 0.0; t.reach0 = n; }
    else {
        t.reach0 = og_count_reachable(n, g->edges, e, 0, c->algo);
        t.conn = (double)(n - t.reach0) * 1000.0;
    }

    /* endpointConnectivity, ProceduralGeneration.fs:49-54: first Origin, first Terminal;
     * compares the reachable COUNT with the terminal INDEX. */
    int origin = -1, terminal = -1;
    for (int i = 0; i < n && origin < 0; i++) if (g->kind[i] == OG_ORIGIN) origin = i;
    for (int i = 0; i < n && terminal < 0; i++) if (g->kind[i] == OG_TERMINAL) terminal = i;
    if (origin >= 0 && terminal >= 0) {
        /* the reference runs this second BFS even when origin is node 0; the result is the same */
        if (c->algo == OG_ALGO_OPTIMISED && origin == 0 && n > 1) t.reach_origin = t.reach0;
        else t.reach_origin = og_count_reachable(n, g->edges, e, origin, c->algo);
        t.endpoint = (t.reach_origin > terminal) ? 0.0 : 10000.0;
    } else { t.reach_origin = -1; t.endpoint = 10000.0; }

    /* Constraints.edgeLength 2.0, ParallelTempering.Core.fs:258-263 via sumEdgeNodePairs :235-239 */
    acc_t a; acc_init(&a, c->sum_mode);
    for (int k = 0; k < e; k++) {
        int i = g->edges[2*k], j = g->edges[2*k+1];
        double dev = (double)dist3(g->pos + 3*i, g->pos + 3*j) - c->target_len;
        acc_add(&a, dev * dev);
    }
    t.edge = acc_value(&a); err |= a.err;

    /* Constraints.collision 0.5, ParallelTempering.Core.fs:271-272 */
    t.coll = og_collision_penalty(g->pos, n, c, &t.npairs, &err);

    /* Constraints.distribution, ParallelTempering.Core.fs:276-290 with centroid :227-232 */
    if (n == 0) t.dist = 0.0;
    else {
        float cen[3];
        if (c->sum_mode == OG_SUM_FSHARP) {
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;      /* Array.fold (+) Vector3.Zero, fp32 */
            for (int i = 0; i < n; i++) { sx = sx + g->pos[3*i]; sy = sy + g->pos[3*i+1]; sz = sz + g->pos[3*i+2]; }
            cen[0] = sx / (float)n; cen[1] = sy / (float)n; cen[2] = sz / (float)n;
        } else {
            acc_t cx, cy, cz; acc_init(&cx, 1); acc_init(&cy, 1); acc_init(&cz, 1);
            for (int i = 0; i < n; i++) { acc_add(&cx, (double)g->pos[3*i]); acc_add(&cy, (double)g->pos[3*i+1]); acc_add(&cz, (double)g->pos[3*i+2]); }
            err |= cx.err | cy.err | cz.err;
            cen[0] = (float)acc_value(&cx) / (float)n; cen[1] = (float)acc_value(&cy) / (float)n; cen[2] = (float)acc_value(&cz) / (float)n;
        }
        acc_t s; acc_init(&s, c->sum_mode);
        for (int i = 0; i < n; i++) acc_add(&s, (double)dist3(g->pos + 3*i, cen));
        err |= s.err;
        double avg = acc_value(&s) / (double)n;         /* Array.averageBy */
        acc_t v; acc_init(&v, c->sum_mode);
        for (int i = 0; i < n; i++) {
            double dev = (double)dist3(g->pos + 3*i, cen) - avg;
            acc_add(&v, dev * dev * 0.01);
        }
        err |= v.err;
        t.dist = acc_value(&v);
    }

    /* smoothTransitions, ProceduralGeneration.fs:58-63 */
    acc_init(&a, c->sum_mode);
    for (int k = 0; k < e; k++) {
        int i = g->edges[2*k], j = g->edges[2*k+1];
        if (g->kind[i] == OG_STANDARD && g->kind[j] == OG_STANDARD)
            acc_add(&a, fabs(g->value[i] - g->value[j]) * 10.0);
        else
            acc_add(&a, 0.0);
    }
    t.trans = acc_value(&a); err |= a.err;

    /* Constraints.combine: List.sumBy (fun (w, fn) -> w * fn graph), ParallelTempering.Core.fs:306-309 */
    double total = 0.0;
    total = total + c->w[0] * t.conn;
    total = total + c->w[1] * t.endpoint;
    total = total + c->w[2] * t.edge;
    total = total + c->w[3] * t.coll;
    total = total + c->w[4] * t.dist;
    total = total + c->w[5] * t.trans;
    t.total = total;
    t.range_error = err;
    if (out) *out = t;
    return total;
}

/* ======================================================================== */
/* mutations and the step                                                    */
/* ======================================================================== */

/* One optimizationStep (ParallelTempering.Core.fs:122-133):
 * proposed = mutate state T; proposedCost = cost proposed; Acceptance.acceptMove.
 * Mutations.adaptive (:67-79) picks the move; draw order: selector, then
 * perturb: index, x, y, z (:72/:79 evaluate nextInt before perturbPosition's
 * singles :42-44) -- add: i, j (:53-54) -- remove: list index only when
 * |E| > N-1 (:61-63); then the accept draw only when delta >= 0 (:34). */
void og_step(og_chain *ch, const og_config *c, og_stream *s, og_trace_rec *tr)
{
    og_graph *cur = &ch->g;
    og_graph prop = *cur;                   /* shares buffers until a move changes them */
    double T = ch->temperature;
    int n = cur->n;
    int kind, a = -1, b = -1, noop = 0, capacity_hit = 0;

    if (c->mutation_mode == OG_MUT_PERTURB_ONLY) kind = 0;
    else if (T > 2.0) { int r = next_int(s, 0, 5); kind = (r == 0) ? 0 : (r <= 2 ? 1 : 2); }
    else { int r = next_int(s, 0, 10); kind = (r == 0) ? 1 : (r == 1 ? 2 : 0); }

    if (kind == 0) {
        /* perturbPosition, ParallelTempering.Core.fs:39-47 */
        a = next_int(s, 1, n);
        float scale = (float)(T * 0.5);
        float ox = (next_single(s, 2) * 2.0f - 1.0f) * scale;
        float oy = (next_single(s, 3) * 2.0f - 1.0f) * scale;
        float oz = (next_single(s, 4) * 2.0f - 1.0f) * scale;
        memcpy(ch->scratch.pos, cur->pos, sizeof(float) * 3 * (size_t)n);   /* Array.mapi copies every position */
        if (a >= 0 && a < n) {
            ch->scratch.pos[3*a]   = cur->pos[3*a]   + ox;
            ch->scratch.pos[3*a+1] = cur->pos[3*a+1] + oy;
            ch->scratch.pos[3*a+2] = cur->pos[3*a+2] + oz;
        }
        prop.pos = ch->scratch.pos;
    } else if (kind == 1) {
        /* addEdge, ParallelTempering.Core.fs:49-58 */
        if (n < 2) noop = 1;
        else {
            a = next_int(s, 1, n); b = next_int(s, 2, n);
            int exists = 0;
            for (int k = 0; k < cur->e; k++) {
                int x = cur->edges[2*k], y = cur->edges[2*k+1];
                if ((x == a && y == b) || (x == b && y == a)) { exists = 1; break; }
            }
            if (a != b && !exists && cur->e >= cur->ecap) capacity_hit = 1;
            if (a != b && !exists && cur->e < cur->ecap) {
                ch->scratch.edges[0] = a; ch->scratch.edges[1] = b;      /* cons: new head */
                memcpy(ch->scratch.edges + 2, cur->edges, sizeof(int32_t) * 2 * (size_t)cur->e);
                prop.edges = ch->scratch.edges; prop.e = cur->e + 1;
            } else noop = 1;
        }
    } else {
        /* removeEdge, ParallelTempering.Core.fs:60-65 */
        if (cur->e > n - 1) {
            a = next_int(s, 1, cur->e);
            int w = 0;
            for (int k = 0; k < cur->e; k++)
                if (k != a) { ch->scratch.edges[2*w] = cur->edges[2*k]; ch->scratch.edges[2*w+1] = cur->edges[2*k+1]; w++; }
            prop.edges = ch->scratch.edges; prop.e = w;
        } else noop = 1;
    }

    og_terms terms;
    double pc = og_cost(&prop, c, &terms);
    double delta = pc - ch->cost;
    int drew = 0, accepted;
    double u = -1.0;
    if (delta < 0.0) accepted = 1;
    else { u = draw(s, 5); drew = 1; accepted = (u < exp(-delta / T)); }

    if (accepted) {
        if (prop.pos != cur->pos) { float *old = cur->pos; cur->pos = prop.pos; ch->scratch.pos = old; }
        if (prop.edges != cur->edges) { int32_t *old = cur->edges; cur->edges = prop.edges; ch->scratch.edges = old; }
        cur->e = prop.e;
        ch->cost = pc;
    }
    if (tr) {
        tr->kind = kind; tr->a = a; tr->b = b;
        tr->flags = (noop ? 1 : 0) | (drew ? 2 : 0) | (accepted ? 4 : 0) | (capacity_hit ? 8 : 0);
        tr->reach0 = terms.reach0; tr->reach_origin = terms.reach_origin;
        tr->delta = delta; tr->u = u; tr->cost_after = ch->cost;
    }
}

/* ParallelTempering.run, ParallelTempering.Core.fs:149-179: initial costs,
 * then per step: every chain steps (Execution.Parallel.map, Execution.fs:31-39),
 * and when step % swapFreq = 0 a sequential pass i = 0..T-2 of attemptSwap
 * (:112-119) over the updated array.  attemptSwap always draws. */
int og_run(const og_run_args *a, const og_config *c)
{
    int T = a->temps, n = a->n, ecap = a->ecap;
    og_chain *ch = calloc((size_t)T, sizeof(og_chain));
    int err = 0;
    for (int t = 0; t < T; t++) {
        ch[t].g.n = n; ch[t].g.ecap = ecap; ch[t].g.e = a->ecount[t];
        ch[t].g.pos = malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
        ch[t].g.edges = malloc(sizeof(int32_t) * 2 * (size_t)(ecap > 0 ? ecap : 1));
        ch[t].scratch = ch[t].g;
        ch[t].scratch.pos = malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
        ch[t].scratch.edges = malloc(sizeof(int32_t) * 2 * (size_t)(ecap > 0 ? ecap : 1));
        memcpy(ch[t].g.pos, a->pos + (size_t)t * 3 * n, sizeof(float) * 3 * (size_t)n);
        memcpy(ch[t].g.edges, a->edges + (size_t)t * 2 * ecap, sizeof(int32_t) * 2 * (size_t)a->ecount[t]);
        ch[t].g.kind = a->kind; ch[t].g.value = a->value;
        ch[t].temperature = a->temperature[t];
        og_terms tm;
        ch[t].cost = og_cost(&ch[t].g, c, &tm);
        err |= tm.range_error;
    }
    int64_t passes = 0;
    int threads = a->threads > 0 ? a->threads : 1;
    (void)threads;
    for (int64_t step = a->first_step; step < a->first_step + a->iters; step++) {
        for (int t = 0; t < T; t++) a->chain_streams[t].step_base = step * a->chain_streams[t].stride;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) if (threads > 1)
#endif
        for (int t = 0; t < T; t++)
            og_step(&ch[t], c, &a->chain_streams[t], a->trace ? &a->trace[(size_t)(step - a->first_step) * T + t] : NULL);
        if (a->swap_freq > 0 && step % a->swap_freq == 0) {
            for (int i = 0; i + 1 < T; i++) {
                double p = og_swap_probability(ch[i].cost, ch[i].temperature, ch[i+1].cost, ch[i+1].temperature);
                if (p > 1.0) p = 1.0;                   /* min 1.0 p */
                double u = og_stream_next(a->swap_stream);
                int sw = u < p;
                if (sw) {
                    og_graph g = ch[i].g, sc = ch[i].scratch; double cst = ch[i].cost;
                    ch[i].g = ch[i+1].g; ch[i].scratch = ch[i+1].scratch; ch[i].cost = ch[i+1].cost;
                    ch[i+1].g = g; ch[i+1].scratch = sc; ch[i+1].cost = cst;
                }
                if (a->swap_trace) a->swap_trace[(size_t)passes * (T - 1) + i] = (uint8_t)sw;
            }
            passes++;
        }
    }
    for (int t = 0; t < T; t++) {
        memcpy(a->pos + (size_t)t * 3 * n, ch[t].g.pos, sizeof(float) * 3 * (size_t)n);
        memcpy(a->edges + (size_t)t * 2 * ecap, ch[t].g.edges, sizeof(int32_t) * 2 * (size_t)ch[t].g.e);
        a->ecount[t] = ch[t].g.e;
        a->cost[t] = ch[t].cost;
        for (int k = 0; k < T; k++) if (a->chain_streams[k].exhausted) err |= 2;
        free(ch[t].g.pos); free(ch[t].g.edges); free(ch[t].scratch.pos); free(ch[t].scratch.edges);
    }
    if (a->swap_stream && a->swap_stream->exhausted) err |= 2;
    if (a->n_swap_passes) *a->n_swap_passes = passes;
    free(ch);
    return err;
}

/* ---- Convergence (reference src/ParallelTempering.Core.fs:312-492) ------------------------- */
struct og_conv_history {
    int chains, window;
    double **costs;     /* per chain, oldest first */
    int *count;
    uint8_t *swaps;     /* oldest first */
    int nswaps;
};

og_conv_history *og_conv_new(int chains, int window)
{
    og_conv_history *h = calloc(1, sizeof *h);
    h->chains = chains; h->window = window;
    h->costs = calloc((size_t)(chains > 0 ? chains : 1), sizeof *h->costs);
    h->count = calloc((size_t)(chains > 0 ? chains : 1), sizeof *h->count);
    for (int c = 0; c < chains; c++) h->costs[c] = malloc(sizeof(double) * (size_t)(window + 1));
    h->swaps = NULL; h->nswaps = 0;
    return h;
}

void og_conv_free(og_conv_history *h)
{
    if (!h) return;
    for (int c = 0; c < h->chains; c++) free(h->costs[c]);
    free(h->costs); free(h->count); free(h->swaps); free(h);
}

void og_conv_update(og_conv_history *h, const double *chain_costs, const uint8_t *outcomes, int n_outcomes)
{
    for (int c = 0; c < h->chains; c++) {                   /* :344-349 */
        h->costs[c][h->count[c]++] = chain_costs[c];
        if (h->count[c] > h->window) {
            memmove(h->costs[c], h->costs[c] + 1, sizeof(double) * (size_t)(h->count[c] - 1));
            h->count[c]--;
        }
    }
    if (n_outcomes > 0) {                                   /* :531 prepends, :352 appends the list */
        h->swaps = realloc(h->swaps, (size_t)(h->nswaps + n_outcomes));
        for (int k = 0; k < n_outcomes; k++) h->swaps[h->nswaps + k] = outcomes[n_outcomes - 1 - k];
        h->nswaps += n_outcomes;
    }
    while (h->nswaps > h->window) {                         /* :353-354 */
        memmove(h->swaps, h->swaps + 1, (size_t)(h->nswaps - 1));
        h->nswaps--;
    }
}

static double conv_average(const double *x, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i];
    return s / (double)n;
}

static double conv_sum_sq_dev(const double *x, int n, double mean)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) { double d = x[i] - mean; s += d * d; }    /* (x - mean) ** 2.0 */
    return s;
}

void og_conv_check(const og_conv_history *h, int iteration, double strictness, og_conv_metrics *out)
{
    memset(out, 0, sizeof *out);
    out->iteration = iteration;
    const int mch = h->chains;
    /* gelmanRubin :360-394 */
    if (mch >= 2) {
        int minlen = h->count[0];
        for (int c = 1; c < mch; c++) if (h->count[c] < minlen) minlen = h->count[c];
        const double n = (double)minlen, m = (double)mch;
        if (n >= 10.0) {
            double *means = malloc(sizeof(double) * (size_t)mch), *vars = malloc(sizeof(double) * (size_t)mch);
            for (int c = 0; c < mch; c++) {
                const double *smp = h->costs[c] + (h->count[c] - minlen);
                means[c] = conv_average(smp, minlen);
                vars[c] = (1.0 / (n - 1.0)) * conv_sum_sq_dev(smp, minlen, means[c]);
            }
            const double grand = conv_average(means, mch);
            const double B = (n / (m - 1.0)) * conv_sum_sq_dev(means, mch, grand);
            const double W = conv_average(vars, mch);
            out->has_gelman_rubin = 1;
            if (W > 1e-10) {
                const double var_plus = ((n - 1.0) / n) * W + (1.0 / n) * B;
                out->gelman_rubin = sqrt(var_plus / W);
            } else out->gelman_rubin = 1.0;
            free(means); free(vars);
        }
    }
    /* effectiveSampleSize :397-426 on chain 0 (:447-450) */
    if (mch > 0 && h->count[0] > 0 && h->count[0] >= 20) {
        const int n = h->count[0];
        const double *x = h->costs[0];
        const double mean = conv_average(x, n);
        const double variance = conv_sum_sq_dev(x, n, mean) / (double)n;
        out->has_ess = 1;
        if (variance < 1e-10) out->effective_sample_size = (double)n;
        else {
            const int max_lag = n / 3 < 50 ? n / 3 : 50;
            double rho_sum = 0.0;
            int k = 1;
            while (k < max_lag) {
                double s = 0.0;
                for (int i = 0; i < n - k; i++) s += (x[i] - mean) * (x[i + k] - mean);
                const double rho = s / ((double)(n - k) * variance);
                if (rho > 0.05) { rho_sum += rho; k++; } else k = max_lag;
            }
            const double ess = (double)n / (1.0 + 2.0 * rho_sum);
            out->effective_sample_size = ess != ess ? ess : (ess > 1.0 ? ess : 1.0);
        }
    }
    /* :453-462 */
    {
        const int rw = h->window / 5 < 50 ? h->window / 5 : 50;
        int total = 0;
        for (int c = 0; c < mch; c++) total += h->count[c] >= rw ? rw : h->count[c];
        if (total == 0) out->cost_stabilization = 1.0;
        else {
            double *all = malloc(sizeof(double) * (size_t)total);
            int w = 0;
            for (int c = 0; c < mch; c++) {
                const int take = h->count[c] >= rw ? rw : h->count[c];
                memcpy(all + w, h->costs[c] + (h->count[c] - take), sizeof(double) * (size_t)take);
                w += take;
            }
            const double mean = conv_average(all, total);
            if (fabs(mean) < 1e-10) out->cost_stabilization = 0.0;
            else out->cost_stabilization = sqrt(conv_sum_sq_dev(all, total, mean) / (double)total) / fabs(mean);
            free(all);
        }
    }
    /* :465-470 */
    if (h->nswaps > 0) {
        int yes = 0;
        for (int i = 0; i < h->nswaps; i++) yes += h->swaps[i] ? 1 : 0;
        out->swap_acceptance_rate = (double)yes / (double)h->nswaps;
    }
    const int swap_quality = out->swap_acceptance_rate >= 0.2 && out->swap_acceptance_rate <= 0.4;
    const double rhat_threshold = 1.2 - strictness * 0.19, cv_threshold = 0.1 - strictness * 0.09;
    out->has_converged = out->has_gelman_rubin && out->has_ess && out->gelman_rubin < rhat_threshold &&
                         out->cost_stabilization < cv_threshold && out->effective_sample_size > 50.0 && swap_quality;
}

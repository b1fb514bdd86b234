"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so) for the tests.

The oracle is test infrastructure: only tests/, the smoke check and the CPU
baseline leg of bench.py load it.  The product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libpt_oracle.so")


class Stream(C.Structure):
    _fields_ = [("mode", C.c_int32), ("exhausted", C.c_int32), ("u", C.POINTER(C.c_double)),
                ("len", C.c_int64), ("key", C.c_uint32 * 2), ("stream_id", C.c_uint32),
                ("domain", C.c_uint32), ("pos", C.c_int64),
                ("stride", C.c_int64), ("step_base", C.c_int64)]


class Config(C.Structure):
    _fields_ = [("w", C.c_double * 6), ("target_len", C.c_double), ("min_dist", C.c_double),
                ("eps", C.c_double), ("sum_mode", C.c_int32), ("algo", C.c_int32),
                ("mutation_mode", C.c_int32), ("brute_threshold", C.c_int32)]


class Terms(C.Structure):
    _fields_ = [("conn", C.c_double), ("endpoint", C.c_double), ("edge", C.c_double),
                ("coll", C.c_double), ("dist", C.c_double), ("trans", C.c_double),
                ("total", C.c_double), ("reach0", C.c_int32), ("reach_origin", C.c_int32),
                ("npairs", C.c_int64), ("range_error", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("n", C.c_int32), ("e", C.c_int32), ("ecap", C.c_int32),
                ("pos", C.POINTER(C.c_float)), ("edges", C.POINTER(C.c_int32)),
                ("kind", C.POINTER(C.c_uint8)), ("value", C.POINTER(C.c_double))]


TRACE_DTYPE = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("flags", "<i4"),
                        ("reach0", "<i4"), ("reach_origin", "<i4"), ("delta", "<f8"),
                        ("u", "<f8"), ("cost_after", "<f8")])
assert TRACE_DTYPE.itemsize == 48


class RunArgs(C.Structure):
    _fields_ = [("n", C.c_int32), ("ecap", C.c_int32), ("temps", C.c_int32), ("iters", C.c_int32),
                ("swap_freq", C.c_int32), ("threads", C.c_int32),
                ("pos", C.POINTER(C.c_float)), ("edges", C.POINTER(C.c_int32)),
                ("ecount", C.POINTER(C.c_int32)), ("kind", C.POINTER(C.c_uint8)),
                ("value", C.POINTER(C.c_double)), ("temperature", C.POINTER(C.c_double)),
                ("cost", C.POINTER(C.c_double)), ("chain_streams", C.POINTER(Stream)),
                ("swap_stream", C.POINTER(Stream)), ("trace", C.c_void_p),
                ("swap_trace", C.POINTER(C.c_uint8)), ("n_swap_passes", C.POINTER(C.c_int64)),
                ("first_step", C.c_int64)]


SUM_FSHARP, SUM_CANONICAL = 0, 1
ALGO_FAITHFUL, ALGO_OPTIMISED = 0, 1
MUT_ADAPTIVE, MUT_PERTURB_ONLY = 0, 1
ORIGIN, TERMINAL, STANDARD = 0, 1, 2

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.og_stream_next.restype = C.c_double
        L.og_stream_next.argtypes = [C.POINTER(Stream)]
        L.og_stream_philox.argtypes = [C.POINTER(Stream), C.c_uint64, C.c_uint32, C.c_uint32]
        L.og_stream_array.argtypes = [C.POINTER(Stream), C.POINTER(C.c_double), C.c_int64]
        L.og_hashrng_next_float.restype = C.c_double
        L.og_hashrng_next_float.argtypes = [C.c_char_p, C.c_int]
        L.og_hashrng_next_int.restype = C.c_int
        L.og_hashrng_next_int.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.og_hash_coordinates.restype = C.c_uint64
        L.og_hash_coordinates.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.og_seed_key.restype = C.c_uint64
        L.og_seed_key.argtypes = [C.c_char_p]
        L.og_cost.restype = C.c_double
        L.og_cost.argtypes = [C.POINTER(Graph), C.POINTER(Config), C.POINTER(Terms)]
        L.og_count_reachable.restype = C.c_int
        L.og_count_reachable.argtypes = [C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]
        L.og_accept_probability.restype = C.c_double
        L.og_accept_probability.argtypes = [C.c_double] * 3
        L.og_swap_probability.restype = C.c_double
        L.og_swap_probability.argtypes = [C.c_double] * 4
        L.og_swap_frequency.restype = C.c_int
        L.og_swap_frequency.argtypes = [C.c_int]
        L.og_geometric_ladder.argtypes = [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.og_uniform_sphere.argtypes = [C.c_int, C.POINTER(C.c_float)]
        L.og_fibonacci_sphere.argtypes = [C.c_char_p, C.c_int, C.c_double, C.POINTER(C.c_float)]
        L.og_seeded_uniform_sphere.argtypes = [C.c_char_p, C.c_int, C.c_double, C.POINTER(C.c_float)]
        L.og_create_graph.argtypes = [C.c_char_p, C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                      C.POINTER(C.c_double)]
        L.og_generate_edges.restype = C.c_int
        L.og_generate_edges.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_double, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_double), C.c_int]
        L.og_default_config.argtypes = [C.POINTER(Config)]
        L.og_run.restype = C.c_int
        L.og_run.argtypes = [C.POINTER(RunArgs), C.POINTER(Config)]
        L.og_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.og_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def default_config(sum_mode=SUM_CANONICAL, algo=ALGO_OPTIMISED, mutation_mode=MUT_ADAPTIVE, weights=None):
    c = Config()
    lib().og_default_config(C.byref(c))
    c.sum_mode, c.algo, c.mutation_mode = sum_mode, algo, mutation_mode
    if weights is not None:
        for i, w in enumerate(weights):
            c.w[i] = w
    return c


def geometric_ladder(tmin, tmax, n):
    out = (C.c_double * max(n, 1))()
    cnt = C.c_int(0)
    lib().og_geometric_ladder(tmin, tmax, n, out, C.byref(cnt))
    return np.array(out[:cnt.value])


def uniform_sphere(n):
    pos = np.zeros((n, 3), np.float32)
    lib().og_uniform_sphere(n, _p(pos, C.c_float))
    return pos


def fibonacci_sphere(seed, n, perturbation):
    pos = np.zeros((max(n, 0), 3), np.float32)
    lib().og_fibonacci_sphere(seed.encode(), n, perturbation, _p(pos, C.c_float))
    return pos


def seeded_uniform_sphere(seed, n, radius):
    pos = np.zeros((max(n, 0), 3), np.float32)
    lib().og_seeded_uniform_sphere(seed.encode(), n, radius, _p(pos, C.c_float))
    return pos


def create_graph(seed, n, perturbation):
    """DeterministicGeneration.createGraph (Deterministic.fs:105-116): pos, kind, value."""
    pos = np.zeros((n, 3), np.float32)
    kind = np.zeros(n, np.uint8)
    value = np.zeros(n, np.float64)
    lib().og_create_graph(seed.encode(), n, perturbation, _p(pos, C.c_float), _p(kind, C.c_uint8), _p(value, C.c_double))
    return pos, kind, value


def generate_edges(seed, pos, avg_degree):
    """DeterministicGeneration.generateEdges (Deterministic.fs:119-135): edges in list order, weights."""
    n = len(pos)
    cap = max(int(n * avg_degree / 2.0), 1)
    edges = np.zeros((cap, 2), np.int32)
    w = np.zeros(cap, np.float64)
    pos = np.ascontiguousarray(pos, np.float32)
    cnt = lib().og_generate_edges(seed.encode(), _p(pos, C.c_float), n, avg_degree, _p(edges, C.c_int32), _p(w, C.c_double), cap)
    assert cnt >= 0
    return edges[:cnt].copy(), w[:cnt].copy()


def default_kinds(n):
    """createGraph's node typing (ProceduralGeneration.fs:37-41): 0 Origin, n-1 Terminal."""
    kind = np.full(n, STANDARD, np.uint8)
    if n > 0:
        kind[0] = ORIGIN
    if n > 1:
        kind[n - 1] = TERMINAL
    return kind


def cost(pos, edges, kind, value, cfg):
    """pos [n,3] f32, edges [e,2] i32 in list order. Returns Terms."""
    pos = np.ascontiguousarray(pos, np.float32)
    edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    kind = np.ascontiguousarray(kind, np.uint8)
    value = np.ascontiguousarray(value, np.float64)
    g = Graph(pos.shape[0], edges.shape[0], max(edges.shape[0], 1), _p(pos, C.c_float),
              _p(edges, C.c_int32), _p(kind, C.c_uint8), _p(value, C.c_double))
    t = Terms()
    lib().og_cost(C.byref(g), C.byref(cfg), C.byref(t))
    return t


def collision_pairs(pos, min_dist, eps=0.001):
    """All pairs i < j with eps < dist < min_dist, sorted (the reference's CPU detectCollisions as a set)."""
    L = lib()
    L.og_collision_pairs.restype = C.c_int64
    L.og_collision_pairs.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int32), C.c_int64]
    p = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    n = p.shape[0]
    cap = max(n * n // 2, 1)
    out = np.zeros((cap, 2), np.int32)
    found = L.og_collision_pairs(_p(p, C.c_float), n, min_dist, eps, _p(out, C.c_int32), cap)
    return out[:found]


def count_reachable(n, edges, start, algo=ALGO_FAITHFUL):
    edges = np.ascontiguousarray(edges, np.int32).reshape(-1, 2)
    return lib().og_count_reachable(n, _p(edges, C.c_int32), edges.shape[0], start, algo)


def philox_stream(key, stream_id, domain, count):
    s = Stream()
    lib().og_stream_philox(C.byref(s), key, stream_id, domain)
    return np.array([lib().og_stream_next(C.byref(s)) for _ in range(count)])


def run(pos, edges, ecount, kind, value, temps, iters, swap_freq, cfg, *, key=None, stream_base=0,
        swap_stream_id=0, chain_u=None, swap_u=None, trace=True, threads=1, stride=6, first_step=0,
        stream_ids=None):
    """Run one ladder.  pos [T,N,3]; edges [T,ecap,2] (list order); ecount [T].

    Random draws come from Philox streams (key, stream ids stream_base+t, swap
    stream swap_stream_id) or from injected arrays chain_u [T,L] / swap_u [Ls].
    Returns dict(pos, edges, ecount, cost, trace, swap_trace, consumed).
    """
    L = lib()
    pos = np.array(pos, np.float32, copy=True, order="C")
    edges = np.array(edges, np.int32, copy=True, order="C")
    ecount = np.array(ecount, np.int32, copy=True)
    kind = np.ascontiguousarray(kind, np.uint8)
    value = np.ascontiguousarray(value, np.float64)
    temps = np.ascontiguousarray(temps, np.float64)
    T, N = pos.shape[0], pos.shape[1]
    ecap = edges.shape[1]
    costs = np.zeros(T, np.float64)
    streams = (Stream * T)()
    sw = Stream()
    keep = []
    if chain_u is not None:
        chain_u = np.ascontiguousarray(chain_u, np.float64)
        swap_u = np.ascontiguousarray(swap_u, np.float64)
        keep += [chain_u, swap_u]
        for t in range(T):
            L.og_stream_array(C.byref(streams[t]), _p(chain_u[t], C.c_double), chain_u.shape[1])
        L.og_stream_array(C.byref(sw), _p(swap_u, C.c_double), swap_u.shape[0])
    else:
        for t in range(T):
            L.og_stream_philox(C.byref(streams[t]), key, stream_ids[t] if stream_ids is not None else stream_base + t, 0)
        L.og_stream_philox(C.byref(sw), key, swap_stream_id, 1)
    for t in range(T):
        streams[t].stride = stride
    tr = np.zeros((iters, T), TRACE_DTYPE) if trace else None
    npass_max = (iters + max(swap_freq, 1) - 1) // max(swap_freq, 1) if swap_freq > 0 else 0
    st = np.zeros((max(npass_max, 1), max(T - 1, 1)), np.uint8)
    npass = C.c_int64(0)
    a = RunArgs(N, ecap, T, iters, swap_freq, threads, _p(pos, C.c_float), _p(edges, C.c_int32),
                _p(ecount, C.c_int32), _p(kind, C.c_uint8), _p(value, C.c_double), _p(temps, C.c_double),
                _p(costs, C.c_double), streams, C.pointer(sw),
                tr.ctypes.data if trace else None, _p(st, C.c_uint8), C.pointer(npass), first_step)
    err = L.og_run(C.byref(a), C.byref(cfg))
    return dict(pos=pos, edges=edges, ecount=ecount, cost=costs, trace=tr,
                swap_trace=st[:npass.value, :max(T - 1, 0)], err=err,
                consumed=np.array([streams[t].pos for t in range(T)]), swap_consumed=sw.pos)


class ConvMetrics(C.Structure):
    _fields_ = [("gelman_rubin", C.c_double), ("effective_sample_size", C.c_double), ("cost_stabilization", C.c_double),
                ("swap_acceptance_rate", C.c_double), ("has_gelman_rubin", C.c_int32), ("has_ess", C.c_int32),
                ("has_converged", C.c_int32), ("iteration", C.c_int32)]


class ConvHistory:
    """Convergence.ConvergenceHistory with updateHistory / checkConvergence (og_conv_*)."""

    def __init__(self, chains, window):
        self.L = lib()
        self.L.og_conv_new.restype = C.c_void_p
        self.L.og_conv_new.argtypes = [C.c_int, C.c_int]
        self.L.og_conv_free.argtypes = [C.c_void_p]
        self.L.og_conv_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.L.og_conv_check.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(ConvMetrics)]
        self.chains = chains
        self.h = self.L.og_conv_new(chains, window)

    def update(self, chain_costs, outcomes=()):
        c = np.ascontiguousarray(chain_costs, np.float64)
        assert c.size == self.chains
        o = np.ascontiguousarray(outcomes, np.uint8)
        self.L.og_conv_update(self.h, c.ctypes.data, o.ctypes.data if o.size else None, int(o.size))

    def check(self, iteration, strictness):
        m = ConvMetrics()
        self.L.og_conv_check(self.h, iteration, strictness, C.byref(m))
        return dict(gelman_rubin=m.gelman_rubin if m.has_gelman_rubin else None,
                    effective_sample_size=m.effective_sample_size if m.has_ess else None,
                    cost_stabilization=m.cost_stabilization, swap_acceptance_rate=m.swap_acceptance_rate,
                    has_converged=bool(m.has_converged), iteration=m.iteration)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.og_conv_free(self.h)
            self.h = None

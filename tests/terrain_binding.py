"""ctypes binding of the voxel-chunk part of the CPU oracle (oracle/pt_terrain_oracle.c).
Test infrastructure: only tests, smoke() and the benchmark's CPU-baseline leg may use it."""
import ctypes as C

import numpy as np

import oracle_binding as ob

W_NAMES = ["traversability", "geological", "overhang", "erosion", "aesthetics", "caves", "performance", "biomeTransition"]


class Config(C.Structure):
    _fields_ = [("w", C.c_double * 8), ("density_mutation_scale", C.c_float), ("refinement_probability", C.c_double),
                ("solid_to_air_threshold", C.c_float), ("sum_mode", C.c_int32), ("res", C.c_int32)]


class Terms(C.Structure):
    _fields_ = [("trav", C.c_double), ("geo", C.c_double), ("overhang", C.c_double), ("erosion", C.c_double),
                ("aesthetics", C.c_double), ("caves", C.c_double), ("performance", C.c_double), ("biome", C.c_double),
                ("total", C.c_double), ("reached", C.c_int32), ("underground_air", C.c_int32), ("solid", C.c_int32),
                ("surface", C.c_int32), ("range_error", C.c_int32)]


class RunArgs(C.Structure):
    _fields_ = [("temps", C.c_int32), ("iters", C.c_int32), ("swap_freq", C.c_int32), ("threads", C.c_int32),
                ("density", C.POINTER(C.c_float)), ("moisture", C.POINTER(C.c_float)), ("temperature", C.POINTER(C.c_float)),
                ("material", C.POINTER(C.c_uint8)), ("ladder", C.POINTER(C.c_double)), ("cost", C.POINTER(C.c_double)),
                ("chain_streams", C.POINTER(ob.Stream)), ("swap_stream", C.POINTER(ob.Stream)), ("trace", C.c_void_p),
                ("swap_trace", C.POINTER(C.c_uint8)), ("n_swap_passes", C.POINTER(C.c_int64)), ("first_step", C.c_int64)]


DRAWS_PER_STEP = 8
_ready = False


def lib():
    global _ready
    L = ob.lib()
    if not _ready:
        L.ot_default_config.argtypes = [C.POINTER(Config)]
        L.ot_cost.restype = C.c_double
        L.ot_cost.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(Config), C.POINTER(Terms)]
        L.ot_traversable_reach.restype = C.c_int
        L.ot_traversable_reach.argtypes = [C.c_int]
        L.ot_run.restype = C.c_int
        L.ot_run.argtypes = [C.POINTER(RunArgs), C.POINTER(Config)]
        _ready = True
    return L


def default_config(sum_mode=ob.SUM_CANONICAL, res=32, weights=None):
    c = Config()
    lib().ot_default_config(C.byref(c))
    c.sum_mode = sum_mode
    c.res = res
    if weights is not None:
        for i, w in enumerate(weights):
            c.w[i] = w
    return c


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def cost(density, moisture, material, cfg):
    d = np.ascontiguousarray(density, np.float32).reshape(-1)
    m = np.ascontiguousarray(moisture, np.float32).reshape(-1)
    k = np.ascontiguousarray(material, np.uint8).reshape(-1)
    t = Terms()
    lib().ot_cost(_p(d, C.c_float), _p(m, C.c_float), _p(k, C.c_uint8), C.byref(cfg), C.byref(t))
    return t


def run(density, moisture, temperature, material, ladder, iters, swap_freq, cfg, *, key=None, stream_base=0,
        swap_stream_id=0, chain_u=None, swap_u=None, trace=True, threads=1, first_step=0, stream_ids=None):
    """One ladder of voxel chunks.  density/moisture/temperature [T,V]; material [V]."""
    L = lib()
    d = np.array(density, np.float32, copy=True, order="C")
    m = np.array(moisture, np.float32, copy=True, order="C")
    tp = np.array(temperature, np.float32, copy=True, order="C")
    mat = np.ascontiguousarray(material, np.uint8)
    lad = np.ascontiguousarray(ladder, np.float64)
    T = d.shape[0]
    costs = np.zeros(T, np.float64)
    streams = (ob.Stream * T)()
    sw = ob.Stream()
    keep = []
    if chain_u is not None:
        chain_u = np.ascontiguousarray(chain_u, np.float64)
        swap_u = np.ascontiguousarray(swap_u, np.float64)
        keep += [chain_u, swap_u]
        for t in range(T):
            L.og_stream_array(C.byref(streams[t]), _p(chain_u[t], C.c_double), chain_u[t].size)
        L.og_stream_array(C.byref(sw), _p(swap_u, C.c_double), swap_u.size)
    else:
        for t in range(T):
            L.og_stream_philox(C.byref(streams[t]), key, stream_ids[t] if stream_ids is not None else stream_base + t, 0)
        L.og_stream_philox(C.byref(sw), key, swap_stream_id, 1)
    for t in range(T):
        streams[t].stride = DRAWS_PER_STEP
    tr = np.zeros((iters, T), ob.TRACE_DTYPE) if trace else None
    npass_max = (iters + max(swap_freq, 1) - 1) // max(swap_freq, 1) if swap_freq > 0 else 0
    st = np.zeros((max(npass_max, 1), max(T - 1, 1)), np.uint8)
    npass = C.c_int64(0)
    a = RunArgs(T, iters, swap_freq, threads, _p(d, C.c_float), _p(m, C.c_float), _p(tp, C.c_float), _p(mat, C.c_uint8),
                _p(lad, C.c_double), _p(costs, C.c_double), streams, C.pointer(sw), tr.ctypes.data if trace else None,
                _p(st, C.c_uint8), C.pointer(npass), first_step)
    err = L.ot_run(C.byref(a), C.byref(cfg))
    return dict(density=d, moisture=m, temperature=tp, cost=costs, trace=tr, swap_trace=st[:npass.value, :max(T - 1, 0)], err=err)

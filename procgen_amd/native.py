"""ctypes binding of libgpu_compute.so (C ABI: include/procgen_pt.h).

There is no CPU fallback: if the library is missing or a call fails, this
module raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C procgen_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# PROCGEN_PT_LIB selects another build of the same library (e.g. the diagnostic stamps build)
LIB_PATH = os.environ.get("PROCGEN_PT_LIB") or os.path.join(PKG_DIR, "lib", "libgpu_compute.so")

PT_OK = 0
ERR_NAMES = {1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "HIP", 4: "NUMERIC_RANGE", 5: "CAPACITY",
             6: "STREAM_EXHAUSTED", 7: "OUT_OF_MEMORY", 8: "CLUSTER_TIMEOUT"}
MUT_ADAPTIVE, MUT_PERTURB_ONLY = 0, 1
RNG_PHILOX, RNG_INJECTED = 0, 1
NODE_ORIGIN, NODE_TERMINAL, NODE_STANDARD = 0, 1, 2
DRAWS_PER_STEP = 6
FLAG_GLOBAL_MEMORY_PLACEMENT = 1
FLAG_FULL_EVALUATION = 2
FLAG_STREAMING_INCREMENTAL = 4


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"PT_ERR_{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class GraphConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("node_count", C.c_int32), ("edge_capacity", C.c_int32),
                ("temp_count", C.c_int32), ("group_count", C.c_int32), ("swap_frequency", C.c_int32),
                ("mutation_mode", C.c_int32), ("rng_mode", C.c_int32), ("weights", C.c_double * 6),
                ("target_edge_length", C.c_double), ("min_distance", C.c_double),
                ("min_distance_eps", C.c_double), ("philox_key", C.c_uint64), ("seed", C.c_char_p),
                ("device", C.c_int32), ("group_base", C.c_int32), ("ladder_temp_count", C.c_int32),
                ("ladder_slot_base", C.c_int32), ("block_threads", C.c_int32), ("flags", C.c_int32),
                ("cluster_size", C.c_int32), ("reserved", C.c_int32)]


class GraphStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("proposals", C.c_int64), ("accepted", C.c_int64),
                ("swap_attempts", C.c_int64), ("swaps", C.c_int64), ("status_bits", C.c_uint32),
                ("early_rejects", C.c_uint32)]


TRACE_DTYPE = np.dtype([("kind", "<i4"), ("a", "<i4"), ("b", "<i4"), ("flags", "<i4"),
                        ("reach0", "<i4"), ("reach_origin", "<i4"), ("delta", "<f8"),
                        ("u", "<f8"), ("cost_after", "<f8")])

# every symbol include/procgen_pt.h declares
EXPORTS = [
    "pt_last_error", "pt_last_error_string",
    "parallel_tempering_step", "parallel_tempering_step_f32", "parallel_tempering_reset",
    "collision_detection", "collision_detection_f32", "pt_collision_detect_device", "pt_collision_release",
    "pt_graph_default_config", "pt_graph_create", "pt_graph_destroy", "pt_graph_upload",
    "pt_graph_set_draws", "pt_graph_enable_trace", "pt_graph_read_trace", "pt_graph_run",
    "pt_graph_sync", "pt_graph_set_step", "pt_graph_run_to_swap_point", "pt_graph_local_energy_ptr",
    "pt_graph_swap_pass", "pt_graph_read_energies", "pt_graph_download", "pt_graph_replica_slots", "pt_graph_cost_terms",
    "pt_graph_get_stats", "pt_graph_kernel_time", "pt_graph_stream", "pt_graph_info", "pt_debug_canonical_sum", "pt_debug_sqrt_check",
    "pt_graph_enable_convergence", "pt_graph_check_convergence", "pt_graph_run_until_converged", "check_convergence_async",
    "pt_geometric_ladder", "pt_swap_frequency", "pt_seed_key", "pt_hash_coordinates",
    "pt_graph_set_ladder_keys", "pt_chunk_set_ladder_keys", "pt_chunk_set_positions",
    "pt_chunk_default_config", "pt_chunk_create", "pt_chunk_destroy", "pt_chunk_upload", "pt_chunk_set_draws",
    "pt_chunk_enable_trace", "pt_chunk_read_trace", "pt_chunk_run", "pt_chunk_sync", "pt_chunk_download",
    "pt_chunk_replica_slots", "pt_chunk_cost_terms", "pt_chunk_get_stats", "pt_chunk_kernel_time", "pt_chunk_launch_shape",
]

class ChunkConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("temp_count", C.c_int32), ("group_count", C.c_int32),
                ("swap_frequency", C.c_int32), ("rng_mode", C.c_int32), ("device", C.c_int32), ("group_base", C.c_int32),
                ("flags", C.c_int32), ("weights", C.c_double * 8), ("density_mutation_scale", C.c_float),
                ("solid_to_air_threshold", C.c_float), ("refinement_probability", C.c_double),
                ("philox_key", C.c_uint64), ("seed", C.c_char_p)]


CHUNK_RES, CHUNK_VOXELS, CHUNK_DRAWS_PER_STEP, CHUNK_TERMS = 32, 32 ** 3, 8, 12
CHUNK_TERM_NAMES = ["traversability", "geological", "overhang", "erosion", "aesthetics", "caves", "performance",
                    "biomeTransition", "total", "cave_reach", "underground_air", "solid"]


class ConvergenceMetrics(C.Structure):
    """pt_convergence_metrics"""
    _fields_ = [("gelman_rubin", C.c_double), ("effective_sample_size", C.c_double), ("cost_stabilization", C.c_double),
                ("swap_acceptance_rate", C.c_double), ("has_gelman_rubin", C.c_int32), ("has_ess", C.c_int32),
                ("has_converged", C.c_int32), ("iteration", C.c_int32)]

    def as_dict(self):
        return dict(gelman_rubin=self.gelman_rubin if self.has_gelman_rubin else None,
                    effective_sample_size=self.effective_sample_size if self.has_ess else None,
                    cost_stabilization=self.cost_stabilization, swap_acceptance_rate=self.swap_acceptance_rate,
                    has_converged=bool(self.has_converged), iteration=self.iteration)


class LegacyConvergenceMetrics(C.Structure):
    """The record of the F# extern (reference src/GPU.Compute.fs:88-96)."""
    _fields_ = [("GelmanRubin", C.c_float), ("EffectiveSampleSize", C.c_float), ("CostStabilization", C.c_float),
                ("AvgSwapAcceptance", C.c_float), ("Iteration", C.c_int32), ("HasConverged", C.c_int32)]


class GraphInfo(C.Structure):
    _fields_ = [("incremental", C.c_int32), ("lds_placement", C.c_int32), ("cluster_size", C.c_int32),
                ("block_threads", C.c_int32), ("lds_bytes", C.c_int32), ("swap_frequency", C.c_int32),
                ("reserved", C.c_int32 * 2)]


_lib = None


def load():
    """Load the library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it first (make -C procgen_amd/csrc)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and a second
    # runtime cannot open the GPU.  Importing torch first makes the loader satisfy this
    # library's libamdhip64.so.7 dependency with the copy torch already mapped; without torch
    # (e.g. under the F# host) the system ROCm runtime is used.
    if os.environ.get("PROCGEN_PT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.pt_last_error.restype = C.c_int
    L.pt_last_error_string.restype = C.c_char_p
    L.parallel_tempering_step.restype = None
    L.parallel_tempering_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.parallel_tempering_step_f32.restype = None
    L.parallel_tempering_step_f32.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.parallel_tempering_reset.restype = None
    L.pt_graph_default_config.restype = None
    L.pt_graph_default_config.argtypes = [C.POINTER(GraphConfig), i32, i32, i32, i32]
    L.pt_graph_create.argtypes = [C.POINTER(GraphConfig), C.POINTER(vp)]
    L.pt_graph_destroy.restype = None
    L.pt_graph_destroy.argtypes = [vp]
    L.pt_graph_upload.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
    L.pt_graph_set_draws.argtypes = [vp, vp, i64, vp, i64]
    L.pt_graph_enable_trace.argtypes = [vp, i64, i64]
    L.pt_graph_read_trace.argtypes = [vp, vp, i64, vp, i64]
    L.pt_graph_run.argtypes = [vp, i64]
    L.pt_graph_sync.argtypes = [vp]
    L.pt_graph_set_step.argtypes = [vp, i64]
    L.pt_graph_run_to_swap_point.argtypes = [vp, i64, C.POINTER(i64), C.POINTER(i32)]
    L.pt_graph_local_energy_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
    L.pt_graph_swap_pass.argtypes = [vp, vp]
    L.pt_graph_download.argtypes = [vp, vp, vp, i32, vp, vp]
    L.pt_graph_replica_slots.argtypes = [vp, vp]
    L.pt_graph_cost_terms.argtypes = [vp, vp]
    L.pt_graph_get_stats.argtypes = [vp, C.POINTER(GraphStats)]
    L.pt_graph_kernel_time.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), i32]
    L.pt_graph_stream.restype = vp
    L.pt_graph_stream.argtypes = [vp]
    L.pt_debug_canonical_sum.argtypes = [vp, i64, C.POINTER(dbl), C.POINTER(i32)]
    L.pt_debug_sqrt_check.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_uint64)]
    L.pt_graph_enable_convergence.argtypes = [vp, i32]
    L.pt_graph_check_convergence.argtypes = [vp, dbl, i32, vp]
    L.pt_graph_run_until_converged.argtypes = [vp, i64, i32, dbl, C.POINTER(i64), vp]
    L.check_convergence_async.restype = None
    L.check_convergence_async.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, dbl, C.c_int, C.POINTER(LegacyConvergenceMetrics), vp]
    L.collision_detection.restype = None
    L.collision_detection.argtypes = [vp, vp, i32, dbl, vp]
    L.collision_detection_f32.restype = None
    L.collision_detection_f32.argtypes = [vp, vp, i32, C.c_float, vp]
    L.pt_collision_detect_device.argtypes = [vp, i32, C.c_float, vp, C.c_longlong, vp, vp]
    L.pt_collision_release.restype = None
    L.pt_collision_release.argtypes = []
    L.pt_chunk_default_config.restype = None
    L.pt_chunk_default_config.argtypes = [C.POINTER(ChunkConfig), i32, i32]
    L.pt_chunk_create.argtypes = [C.POINTER(ChunkConfig), C.POINTER(vp)]
    L.pt_chunk_destroy.restype = None
    L.pt_chunk_destroy.argtypes = [vp]
    L.pt_chunk_upload.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pt_chunk_set_draws.argtypes = [vp, vp, i64, vp, i64]
    L.pt_chunk_enable_trace.argtypes = [vp, i64, i64]
    L.pt_chunk_read_trace.argtypes = [vp, vp, i64, vp, i64]
    L.pt_chunk_run.argtypes = [vp, i64]
    L.pt_chunk_sync.argtypes = [vp]
    L.pt_chunk_download.argtypes = [vp, vp, vp, vp, vp]
    L.pt_chunk_replica_slots.argtypes = [vp, vp]
    L.pt_chunk_cost_terms.argtypes = [vp, vp]
    L.pt_chunk_get_stats.argtypes = [vp, C.POINTER(GraphStats)]
    L.pt_chunk_kernel_time.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), i32]
    L.pt_chunk_launch_shape.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.pt_geometric_ladder.restype = C.c_int
    L.pt_geometric_ladder.argtypes = [dbl, dbl, i32, C.POINTER(dbl)]
    L.pt_swap_frequency.restype = i32
    L.pt_swap_frequency.argtypes = [i32]
    L.pt_seed_key.restype = C.c_uint64
    L.pt_seed_key.argtypes = [C.c_char_p]
    L.pt_hash_coordinates.restype = C.c_uint64
    L.pt_hash_coordinates.argtypes = [C.c_uint64, i32, i32, i32]
    L.pt_graph_set_ladder_keys.argtypes = [vp, vp]
    L.pt_chunk_set_ladder_keys.argtypes = [vp, vp]
    L.pt_chunk_set_positions.argtypes = [vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != PT_OK:
        L = load()
        raise PtError(rc, (L.pt_last_error_string() or b"").decode())


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def geometric_ladder(t_min, t_max, n):
    out = (C.c_double * max(n, 1))()
    k = load().pt_geometric_ladder(t_min, t_max, n, out)
    return np.array(out[:k])


def swap_frequency(n):
    return int(load().pt_swap_frequency(n))


def hash_coordinates(base_seed: int, x: int, y: int, z: int) -> int:
    return int(load().pt_hash_coordinates(C.c_uint64(base_seed), x, y, z))


def seed_key(seed: str) -> int:
    return int(load().pt_seed_key(seed.encode("utf-8")))

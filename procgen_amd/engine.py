"""GraphEngine: the batched spatial-graph parallel-tempering engine over the C ABI.

Thin: every method is one call into libgpu_compute.so.  Arrays are numpy on the
host side; all replica state lives in HBM between calls.
"""
import ctypes as C

import numpy as np

from . import native as nt


class GraphEngine:
    def __init__(self, node_count, edge_capacity, temp_count, group_count=1, *, swap_frequency=0,
                 mutation_mode=nt.MUT_ADAPTIVE, rng_mode=nt.RNG_PHILOX, weights=None, seed=None,
                 philox_key=0, device=-1, group_base=0, ladder_temp_count=0, ladder_slot_base=0,
                 block_threads=0, flags=0, cluster_size=0, target_edge_length=2.0, min_distance=0.5, min_distance_eps=0.001):
        self.L = nt.load()
        cfg = nt.GraphConfig()
        self.L.pt_graph_default_config(C.byref(cfg), node_count, edge_capacity, temp_count, group_count)
        cfg.swap_frequency = swap_frequency
        cfg.mutation_mode = mutation_mode
        cfg.rng_mode = rng_mode
        if weights is not None:
            for i, w in enumerate(weights):
                cfg.weights[i] = w
        cfg.target_edge_length = target_edge_length
        cfg.min_distance = min_distance
        cfg.min_distance_eps = min_distance_eps
        cfg.philox_key = philox_key
        self._seed = seed.encode("utf-8") if seed is not None else None
        cfg.seed = self._seed
        cfg.device = device
        cfg.group_base = group_base
        cfg.ladder_temp_count = ladder_temp_count
        cfg.ladder_slot_base = ladder_slot_base
        cfg.block_threads = block_threads
        cfg.flags = flags
        cfg.cluster_size = cluster_size
        self.cfg = cfg
        self.N, self.Ecap, self.T, self.G = node_count, max(edge_capacity, 1), temp_count, group_count
        self.Tl = ladder_temp_count or temp_count
        self.R = self.G * self.T
        self.swap_frequency = swap_frequency if swap_frequency > 0 else nt.swap_frequency(node_count)
        h = C.c_void_p()
        nt.check(self.L.pt_graph_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- state ----------------------------------------------------------------
    def upload(self, positions, edges, edge_counts, temperatures, node_kind=None, node_value=None):
        """positions [G,T,N,3]; edges [G,T,stride,2] list order; edge_counts [G,T]; temperatures [G,Tl]."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(self.R, self.N, 3)
        temps = np.ascontiguousarray(temperatures, np.float64).reshape(self.G, self.Tl)
        if edges is None:
            ed, stride, ec = None, 0, np.zeros(self.R, np.int32)
        else:
            ed = np.ascontiguousarray(edges, np.int32)
            stride = ed.size // (self.R * 2) if self.R else 0
            ed = ed.reshape(self.R, stride, 2)
            ec = np.ascontiguousarray(edge_counts, np.int32).reshape(self.R)
        kind = None if node_kind is None else np.ascontiguousarray(node_kind, np.uint8).reshape(self.G, self.N)
        val = None if node_value is None else np.ascontiguousarray(node_value, np.float64).reshape(self.G, self.N)
        nt.check(self.L.pt_graph_upload(self.h, nt.ptr(pos), nt.ptr(ed), stride, nt.ptr(ec), nt.ptr(kind),
                                        nt.ptr(val), nt.ptr(temps)))

    def set_draws(self, chain_u, swap_u):
        """chain_u [G,Tl,steps,6]; swap_u [G,passes,Tl-1]."""
        cu = np.ascontiguousarray(chain_u, np.float64)
        steps = cu.size // (self.G * self.Tl * nt.DRAWS_PER_STEP)
        su = np.ascontiguousarray(swap_u, np.float64)
        passes = su.size // (self.G * max(self.Tl - 1, 1)) if self.Tl > 1 else 0
        nt.check(self.L.pt_graph_set_draws(self.h, nt.ptr(cu), steps, nt.ptr(su), passes))

    def set_ladder_keys(self, keys):
        """Per-ladder Philox keys [G]: a ladder's draws then depend on its key, not on its index in the batch."""
        k = np.ascontiguousarray(keys, np.uint64).reshape(self.G)
        nt.check(self.L.pt_graph_set_ladder_keys(self.h, nt.ptr(k)))

    def enable_trace(self, max_steps, max_passes):
        self._trace_steps, self._trace_passes = max_steps, max_passes
        nt.check(self.L.pt_graph_enable_trace(self.h, max_steps, max_passes))

    def read_trace(self, n_steps, n_passes):
        tr = np.zeros((n_steps, self.G * self.Tl), nt.TRACE_DTYPE)
        sw = np.zeros((n_passes, self.G, max(self.Tl - 1, 0)), np.uint8)
        nt.check(self.L.pt_graph_read_trace(self.h, nt.ptr(tr), n_steps, nt.ptr(sw), n_passes))
        return tr, sw

    def run(self, steps):
        nt.check(self.L.pt_graph_run(self.h, steps))

    def sync(self):
        nt.check(self.L.pt_graph_sync(self.h))

    def set_step(self, step):
        nt.check(self.L.pt_graph_set_step(self.h, step))

    def enable_convergence(self, window_size):
        """Keep the last `window_size` chain costs and swap outcomes of every ladder on the device."""
        nt.check(self.L.pt_graph_enable_convergence(self.h, window_size))

    def check_convergence(self, strictness, iteration=0):
        """Convergence.checkConvergence per ladder: a list of dicts (None for an absent statistic)."""
        out = (nt.ConvergenceMetrics * self.G)()
        nt.check(self.L.pt_graph_check_convergence(self.h, float(strictness), int(iteration), C.cast(out, C.c_void_p)))
        return [m.as_dict() for m in out]

    def run_until_converged(self, max_iterations, check_freq, strictness):
        """Convergence.runUntilConvergence: (steps run, metrics of the last check per ladder)."""
        out = (nt.ConvergenceMetrics * self.G)()
        done = C.c_int64(0)
        nt.check(self.L.pt_graph_run_until_converged(self.h, int(max_iterations), int(check_freq), float(strictness),
                                                     C.byref(done), C.cast(out, C.c_void_p)))
        return done.value, [m.as_dict() for m in out]

    def run_to_swap_point(self, max_steps):
        done, due = C.c_int64(0), C.c_int32(0)
        nt.check(self.L.pt_graph_run_to_swap_point(self.h, max_steps, C.byref(done), C.byref(due)))
        return done.value, bool(due.value)

    def local_energy_ptr(self):
        p, n = C.c_void_p(), C.c_int32(0)
        nt.check(self.L.pt_graph_local_energy_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_energies(self, dst_ptr):
        """Copy the local energies (replica order) to a host or device address."""
        nt.check(self.L.pt_graph_read_energies(self.h, C.c_void_p(dst_ptr)))

    def swap_pass(self, d_energy_all):
        nt.check(self.L.pt_graph_swap_pass(self.h, C.c_void_p(d_energy_all)))

    def download(self, positions=True, edges=True, costs=True):
        pos = np.zeros((self.G, self.T, self.N, 3), np.float32) if positions else None
        ed = np.zeros((self.G, self.T, self.Ecap, 2), np.int32) if edges else None
        ec = np.zeros((self.G, self.T), np.int32)
        co = np.zeros((self.G, self.T), np.float64) if costs else None
        nt.check(self.L.pt_graph_download(self.h, nt.ptr(pos), nt.ptr(ed), self.Ecap, nt.ptr(ec), nt.ptr(co)))
        return dict(pos=pos, edges=ed, ecount=ec, cost=co)

    def replica_slots(self):
        s = np.zeros((self.G, self.Tl), np.int32)
        nt.check(self.L.pt_graph_replica_slots(self.h, nt.ptr(s)))
        return s

    def cost_terms(self):
        t = np.zeros((self.G, self.T, 8), np.float64)
        nt.check(self.L.pt_graph_cost_terms(self.h, nt.ptr(t)))
        return t

    def stats(self):
        s = nt.GraphStats()
        nt.check(self.L.pt_graph_get_stats(self.h, C.byref(s)))
        return dict(steps=s.steps, proposals=s.proposals, accepted=s.accepted,
                    swap_attempts=s.swap_attempts, swaps=s.swaps, status_bits=s.status_bits,
                    early_rejects=s.early_rejects)

    def info(self):
        i = nt.GraphInfo()
        nt.check(self.L.pt_graph_info(self.h, C.byref(i)))
        return dict(incremental=bool(i.incremental), lds_placement=bool(i.lds_placement), cluster_size=i.cluster_size,
                    block_threads=i.block_threads, lds_bytes=i.lds_bytes, swap_frequency=i.swap_frequency)

    def kernel_time(self, reset=False):
        ms, n = C.c_double(0), C.c_int64(0)
        nt.check(self.L.pt_graph_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def stream(self):
        """The engine's HIP stream as an integer handle (e.g. for torch.cuda.ExternalStream)."""
        return int(self.L.pt_graph_stream(self.h) or 0)


class ChunkEngine:
    """Batched parallel tempering over 32^3 voxel chunks (pt_chunk_* of the C ABI): `group_count` chunks,
    each with a ladder of `temp_count` temperatures."""

    def __init__(self, temp_count, group_count, *, swap_frequency=0, rng_mode=nt.RNG_PHILOX, weights=None, seed=None,
                 philox_key=0, device=-1, group_base=0, density_mutation_scale=None, refinement_probability=None,
                 solid_to_air_threshold=None):
        self.L = nt.load()
        cfg = nt.ChunkConfig()
        self.L.pt_chunk_default_config(C.byref(cfg), temp_count, group_count)
        cfg.swap_frequency = swap_frequency
        cfg.rng_mode = rng_mode
        cfg.device = device
        cfg.group_base = group_base
        cfg.philox_key = philox_key
        if weights is not None:
            for i, w in enumerate(weights):
                cfg.weights[i] = w
        if density_mutation_scale is not None:
            cfg.density_mutation_scale = density_mutation_scale
        if refinement_probability is not None:
            cfg.refinement_probability = refinement_probability
        if solid_to_air_threshold is not None:
            cfg.solid_to_air_threshold = solid_to_air_threshold
        self._seed = seed.encode("utf-8") if seed is not None else None
        cfg.seed = self._seed
        self.h = C.c_void_p()
        nt.check(self.L.pt_chunk_create(C.byref(cfg), C.byref(self.h)))
        self.T, self.G, self.R, self.V = temp_count, group_count, temp_count * group_count, nt.CHUNK_VOXELS
        self.swap_frequency = swap_frequency if swap_frequency > 0 else 50

    def close(self):
        if self.h:
            self.L.pt_chunk_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def upload(self, density, moisture, temperature, material, temperatures):
        """density/moisture/temperature [G,T,V] f32; material [G,V] u8; temperatures [G,T]."""
        d = np.ascontiguousarray(density, np.float32).reshape(self.R, self.V)
        m = np.ascontiguousarray(moisture, np.float32).reshape(self.R, self.V)
        t = np.ascontiguousarray(temperature, np.float32).reshape(self.R, self.V)
        k = np.ascontiguousarray(material, np.uint8).reshape(self.G, self.V)
        lad = np.ascontiguousarray(temperatures, np.float64).reshape(self.G, self.T)
        nt.check(self.L.pt_chunk_upload(self.h, nt.ptr(d), nt.ptr(m), nt.ptr(t), nt.ptr(k), nt.ptr(lad)))

    def set_draws(self, chain_u, swap_u):
        cu = np.ascontiguousarray(chain_u, np.float64)
        steps = cu.size // (self.R * nt.CHUNK_DRAWS_PER_STEP)
        su = np.ascontiguousarray(swap_u, np.float64)
        passes = su.size // (self.G * max(self.T - 1, 1)) if self.T > 1 else 0
        nt.check(self.L.pt_chunk_set_draws(self.h, nt.ptr(cu), steps, nt.ptr(su), passes))

    def set_positions(self, xyz):
        """Chunk positions [G,3]: chunk g draws from hashCoordinates(engine key, x, y, z), as Terrain.Generation.generate seeds it."""
        p = np.ascontiguousarray(xyz, np.int32).reshape(self.G, 3)
        nt.check(self.L.pt_chunk_set_positions(self.h, nt.ptr(p)))

    def set_ladder_keys(self, keys):
        k = np.ascontiguousarray(keys, np.uint64).reshape(self.G)
        nt.check(self.L.pt_chunk_set_ladder_keys(self.h, nt.ptr(k)))

    def enable_trace(self, max_steps, max_passes):
        nt.check(self.L.pt_chunk_enable_trace(self.h, max_steps, max_passes))

    def read_trace(self, n_steps, n_passes):
        tr = np.zeros((n_steps, self.R), nt.TRACE_DTYPE)
        sw = np.zeros((n_passes, self.G, max(self.T - 1, 0)), np.uint8)
        nt.check(self.L.pt_chunk_read_trace(self.h, nt.ptr(tr), n_steps, nt.ptr(sw) if sw.size else None, n_passes))
        return tr, sw

    def run(self, steps):
        nt.check(self.L.pt_chunk_run(self.h, steps))

    def sync(self):
        nt.check(self.L.pt_chunk_sync(self.h))

    def download(self, fields=True, costs=True):
        d = np.zeros((self.G, self.T, self.V), np.float32) if fields else None
        m = np.zeros((self.G, self.T, self.V), np.float32) if fields else None
        t = np.zeros((self.G, self.T, self.V), np.float32) if fields else None
        c = np.zeros((self.G, self.T), np.float64) if costs else None
        nt.check(self.L.pt_chunk_download(self.h, nt.ptr(d), nt.ptr(m), nt.ptr(t), nt.ptr(c)))
        return dict(density=d, moisture=m, temperature=t, cost=c)

    def replica_slots(self):
        s = np.zeros((self.G, self.T), np.int32)
        nt.check(self.L.pt_chunk_replica_slots(self.h, nt.ptr(s)))
        return s

    def cost_terms(self):
        t = np.zeros((self.G, self.T, nt.CHUNK_TERMS), np.float64)
        nt.check(self.L.pt_chunk_cost_terms(self.h, nt.ptr(t)))
        return t

    def stats(self):
        s = nt.GraphStats()
        nt.check(self.L.pt_chunk_get_stats(self.h, C.byref(s)))
        return dict(steps=s.steps, proposals=s.proposals, accepted=s.accepted,
                    swap_attempts=s.swap_attempts, swaps=s.swaps, status_bits=s.status_bits)

    def launch_shape(self):
        """(lanes per workgroup, replicas per workgroup) of the step kernel."""
        bt, rpb = C.c_int32(0), C.c_int32(0)
        nt.check(self.L.pt_chunk_launch_shape(self.h, C.byref(bt), C.byref(rpb)))
        return bt.value, rpb.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(0), C.c_int64(0)
        nt.check(self.L.pt_chunk_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


def canonical_sum(terms):
    """Exact fixed-point sum on the device (unit-test hook)."""
    L = nt.load()
    x = np.ascontiguousarray(terms, np.float64)
    out, err = C.c_double(0), C.c_int32(0)
    nt.check(L.pt_debug_canonical_sum(nt.ptr(x), x.size, C.byref(out), C.byref(err)))
    return out.value, err.value

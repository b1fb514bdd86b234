"""pt_collision_detect_device, the entry point that allocates nothing (its work memory is a fixed array
inside the library's device image), against collision_detection_f32 on the same points: two calls in a
row, the second with a smaller input, so that whatever the first call left in the shared work array
must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

from procgen_amd import native as nt
from procgen_amd import workloads as wlm

pytestmark = pytest.mark.gpu


def _reference_entry(pos, min_dist):
    lib = nt.load()
    n = len(pos)
    pairs = np.full(max(n * n // 2, 1) * 2, -7, np.int32)
    count = np.array([-1], np.int32)
    p = np.ascontiguousarray(pos, np.float32).reshape(-1)
    lib.collision_detection_f32(nt.ptr(p), nt.ptr(pairs), n, min_dist, nt.ptr(count))
    assert lib.pt_last_error() == 0, lib.pt_last_error_string()
    return pairs[:2 * count[0]].reshape(-1, 2).copy()


def _device_entry(pos, min_dist, capacity, stream=None):
    import torch
    lib = nt.load()
    n = len(pos)
    d_pos = torch.from_numpy(np.ascontiguousarray(pos, np.float32)).cuda()
    d_pairs = torch.full((max(capacity, 1), 2), -7, dtype=torch.int32, device="cuda")
    d_count = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    st = C.c_void_p(stream.cuda_stream) if stream is not None else None
    nt.check(lib.pt_collision_detect_device(C.c_void_p(d_pos.data_ptr()), n, min_dist, C.c_void_p(d_pairs.data_ptr()),
                                            capacity, C.c_void_p(d_count.data_ptr()), st))
    torch.cuda.synchronize()
    return d_pairs.cpu().numpy(), int(d_count.cpu()[0])


def test_two_calls_in_a_row_match_the_allocating_entry(gpu_lib):
    import torch
    rng = np.random.default_rng(5)
    big = wlm.box_positions(12000, rng) * np.float32(0.8)
    big[7] = big[3]                                         # coincident: not a collision
    big[11] = big[5] + np.array([0.5, 0, 0], np.float32)    # at the upper bound
    small = wlm.box_positions(700, rng) * np.float32(0.5)   # denser: more partners per node
    stream = torch.cuda.Stream()
    for pos, st in ((big, None), (small, stream)):
        want = _reference_entry(pos, 0.5)
        assert len(want) > 100
        cap = len(want) + 10
        pairs, count = _device_entry(pos, 0.5, cap, st)
        assert count == len(want)
        assert np.array_equal(pairs[:count], want)
        assert (pairs[count:] == -7).all(), "wrote past the reported count"


def test_capacity_cuts_the_list_in_order_and_limits_are_refused(gpu_lib):
    lib = nt.load()
    rng = np.random.default_rng(6)
    pos = wlm.box_positions(3000, rng) * np.float32(0.45)
    want = _reference_entry(pos, 0.5)
    cap = len(want) // 2 + 1
    pairs, count = _device_entry(pos, 0.5, cap)
    assert count == cap and np.array_equal(pairs[:cap], want[:cap])
    pairs, count = _device_entry(pos[:1], 0.5, 4)
    assert count == 0 and (pairs == -7).all()
    import torch
    d = torch.zeros(8, dtype=torch.int32, device="cuda")
    rc = lib.pt_collision_detect_device(C.c_void_p(d.data_ptr()), 32769, 0.5, C.c_void_p(d.data_ptr()), 1,
                                        C.c_void_p(d.data_ptr()), None)
    assert rc == 1 and b"32768" in lib.pt_last_error_string()

"""A fixed slice of the randomised parity sweep (tests/fuzz_parity.py): random graph shapes, layouts,
node roles, ladders, move sets, weights, swap intervals, launch splits and kernel routes, each
compared with the oracle bit for bit.  The full sweep is run by hand (`python tests/fuzz_parity.py
3000 2` passed 3000 of 3000 graph cases, `... 300 1 chunk` 300 of 300 voxel-chunk cases); this slice keeps
40 + 10 of them in the suite."""
import numpy as np
import pytest

import fuzz_parity as fz

pytestmark = pytest.mark.gpu

SEED = 3


@pytest.mark.parametrize("index", range(40))
def test_random_case_matches_oracle(gpu_lib, index):
    case = fz.make_case(np.random.default_rng([SEED, index]), SEED * 100000 + index)
    print(case["summary"])
    fz.run_case(case, f"fuzz-{SEED}-{index}")


_early_rejects = []


@pytest.mark.parametrize("index", range(24))
def test_random_case_without_trace_matches_oracle(gpu_lib, index):
    """The paths a production run takes (no step trace): sparse graphs, hot ladders, incremental routes."""
    case = fz.make_traceoff_case(np.random.default_rng([SEED + 1, index]), (SEED + 1) * 100000 + index)
    print(case["summary"])
    gpu = fz.run_case(case, f"fuzz-off-{SEED}-{index}")
    _early_rejects.append(gpu["stats"]["early_rejects"])


def test_cases_without_trace_took_the_early_rejection(gpu_lib):
    """The slice above is only worth its name if removals were in fact rejected from the lower bound."""
    if not _early_rejects:
        pytest.skip("the trace-off cases were deselected")
    assert len(_early_rejects) == 24 and sum(_early_rejects) > 100, _early_rejects


@pytest.mark.parametrize("index", range(10))
def test_random_chunk_case_matches_oracle(gpu_lib, index):
    case = fz.make_chunk_case(np.random.default_rng([SEED, index]), SEED * 100000 + index)
    print(case["summary"])
    fz.run_chunk_case(case, f"fuzz-{SEED}-{index}")


@pytest.mark.parametrize("index", range(30))
def test_random_collision_case_matches_oracle(gpu_lib, index):
    case = fz.make_collide_case(np.random.default_rng([SEED, index]), index)
    print(case["summary"])
    fz.run_collide_case(case, "")

"""integration/abi_harness.c: the F# host's side of the native boundary replayed from plain C (the image has no
.NET): library loaded by path, externs bound by name, records laid out as .NET marshals them (checked at compile
time), the host's call sequences replayed."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "abi_harness.c")
EXE = os.path.join(ROOT, "integration", "abi_harness")
LIB = os.path.join(ROOT, "procgen_amd", "lib", "libgpu_compute.so")


def _build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-std=gnu11", "-o", EXE, SRC, "-ldl", "-lm"], check=True)


def test_layouts_and_bindings_without_a_gpu():
    """Compiling the harness runs its static layout assertions; --layout-only loads the library by path, binds
    every extern the host declares, checks that the three out-of-scope ones are absent, and makes the probe call."""
    _build()
    out = subprocess.run([EXE, LIB, "--layout-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "ABI_HARNESS_OK" in out.stdout and "probe: ok" in out.stdout


@pytest.mark.gpu
def test_host_call_sequences_from_c_match_the_oracle(gpu_lib, tmp_path):
    """The per-iteration loop of GPUTempering.optimize, GPUCollision.detect, check_convergence_async and the batched
    engine, called from C with the host's argument types; the costs both jobs end with are the oracle's."""
    import oracle_binding as ob
    import parity_util as pu
    from procgen_amd import native as nt

    _build()
    dump = str(tmp_path / "job.bin")
    out = subprocess.run([EXE, LIB, "--dump", dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "ABI_HARNESS_OK" in out.stdout
    lines = {ln.split(":")[0]: ln for ln in out.stdout.splitlines() if ":" in ln}
    raw = open(dump, "rb").read()
    n, T, E, iters = np.frombuffer(raw, np.int32, 4)
    off = 16
    pos = np.frombuffer(raw, np.float64, T * n * 3, off).reshape(T, n, 3); off += pos.nbytes
    temps = np.frombuffer(raw, np.float64, T, off); off += temps.nbytes
    edges = np.frombuffer(raw, np.int32, 2 * E, off).reshape(E, 2)

    def job(ecap):
        ed = np.zeros((1, T, ecap, 2), np.int32)
        ed[0, :, :E] = edges
        return dict(n=int(n), temps=int(T), groups=1, ecap=ecap, pos=pos.astype(np.float32)[None], edges=ed,
                    ecount=np.full((1, T), E, np.int32), kind=ob.default_kinds(int(n))[None], value=np.zeros((1, int(n))),
                    temperatures=temps[None].copy())

    f = nt.swap_frequency(int(n))
    legacy = pu.oracle_run(job(int(E)), int(iters), f, seed="libgpu_compute/parallel_tempering_step",
                           mutation_mode=ob.MUT_PERTURB_ONLY, trace=False)[0]
    want = " ".join(float(c).hex() for c in legacy["cost"])
    got = " ".join(float.fromhex(t).hex() for t in lines["optimize"].split("costs")[1].split())
    assert got == want, lines["optimize"]
    batched = pu.oracle_run(job(2 * int(E)), 500, f, seed="abi-harness", trace=False)[0]
    tail = lines["batched"].split("costs")[1]
    got_costs = [float.fromhex(t) for t in tail.split("edges")[0].split()]
    got_edges = [int(t) for t in tail.split("edges")[1].split()]
    assert got_costs == [float(c) for c in batched["cost"]] and got_edges == batched["ecount"].tolist()

    # the terrain job (TerrainBatched.generateFrom's call sequence): costs, the chain kept and its voxels
    import terrain_binding as tb
    raw = open(dump + ".chunks", "rb").read()
    chunks, CT, citers, cf = np.frombuffer(raw, np.int32, 4)
    seed = int(np.frombuffer(raw, np.uint64, 1, 16)[0])
    xyz = np.frombuffer(raw, np.int32, chunks * 3, 24).reshape(chunks, 3)
    off, V = 24 + xyz.nbytes, nt.CHUNK_VOXELS
    fields = []
    for _ in range(3):
        fields.append(np.frombuffer(raw, np.float32, chunks * V, off).reshape(chunks, V)); off += chunks * V * 4
    mat = np.frombuffer(raw, np.uint8, chunks * V, off).reshape(chunks, V); off += chunks * V
    ctemps = np.frombuffer(raw, np.float64, CT, off)
    tail = lines["terrain"].split("costs")[1]
    got_costs = np.array([float.fromhex(t) for t in tail.split("best")[0].split()]).reshape(chunks, CT)
    got_best = [t.split(":") for t in tail.split("best")[1].split()]
    cfg = tb.default_config(ob.SUM_CANONICAL)
    for c in range(chunks):
        key = int(ob.lib().og_hash_coordinates(seed, int(xyz[c, 0]), int(xyz[c, 1]), int(xyz[c, 2])))
        tile = lambda a: np.ascontiguousarray(np.tile(a[c], (CT, 1)))     # noqa: E731
        o = tb.run(tile(fields[0]), tile(fields[1]), tile(fields[2]), mat[c], ctemps.copy(), int(citers), int(cf), cfg,
                   key=key, stream_base=0, swap_stream_id=0, trace=False)
        assert o["err"] == 0
        assert np.array_equal(got_costs[c], o["cost"]), (c, got_costs[c], o["cost"])
        best = int(np.argmin(o["cost"]))
        assert int(got_best[c][0]) == best
        assert float.fromhex(got_best[c][1]) == float(o["density"][best].astype(np.float64).sum())

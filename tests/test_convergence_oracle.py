"""CPU checks of the oracle's Convergence restatement (reference src/ParallelTempering.Core.fs:312-492).

The reference's tests hold no numeric vector for these statistics and the F# cannot run here, so
this part of the oracle is "parity unpinned": it is checked against values derived by hand from
the reference's formulas and against an independent sequential evaluation written here.
"""
import math

import numpy as np
import pytest

import oracle_binding as ob


def _avg(x):
    s = 0.0
    for v in x:
        s += float(v)
    return s / len(x)


def _ssd(x, mean):
    s = 0.0
    for v in x:
        d = float(v) - mean
        s += d * d
    return s


def independent_check(costs, swaps, window, strictness):
    """checkConvergence written from the reference's text; costs = list per chain, oldest first."""
    m = len(costs)
    rhat = None
    if m >= 2:
        n = min(len(c) for c in costs)
        if n >= 10:
            smp = [c[len(c) - n:] for c in costs]
            means = [_avg(c) for c in smp]
            grand = _avg(means)
            B = (n / (m - 1.0)) * _ssd(means, grand)
            W = _avg([(1.0 / (n - 1.0)) * _ssd(c, mu) for c, mu in zip(smp, means)])
            rhat = math.sqrt((((n - 1.0) / n) * W + (1.0 / n) * B) / W) if W > 1e-10 else 1.0
    ess = None
    if m > 0 and len(costs[0]) >= 20:
        x = [float(v) for v in costs[0]]
        n = len(x)
        mean = _avg(x)
        var = _ssd(x, mean) / n
        if var < 1e-10:
            ess = float(n)
        else:
            rho_sum, k, max_lag = 0.0, 1, min(n // 3, 50)
            while k < max_lag:
                s = 0.0
                for i in range(n - k):
                    s += (x[i] - mean) * (x[i + k] - mean)
                rho = s / (float(n - k) * var)
                if rho > 0.05:
                    rho_sum += rho
                    k += 1
                else:
                    k = max_lag
            ess = max(1.0, n / (1.0 + 2.0 * rho_sum))
    rw = min(50, window // 5)
    recent = []
    for c in costs:
        recent += list(c[len(c) - rw:]) if len(c) >= rw else list(c)      # Seq.skip (Count - rw): rw = 0 takes nothing
    if not recent:
        cv = 1.0
    else:
        mean = _avg(recent)
        cv = 0.0 if abs(mean) < 1e-10 else math.sqrt(_ssd(recent, mean) / len(recent)) / abs(mean)
    rate = (sum(1 for s in swaps if s) / len(swaps)) if len(swaps) else 0.0
    conv = (rhat is not None and ess is not None and rhat < 1.2 - strictness * 0.19 and cv < 0.1 - strictness * 0.09
            and ess > 50.0 and 0.2 <= rate <= 0.4)
    return dict(gelman_rubin=rhat, effective_sample_size=ess, cost_stabilization=cv, swap_acceptance_rate=rate,
                has_converged=conv)


def feed(costs_by_step, window, outcomes_by_step=None):
    """costs_by_step [steps][chains]; outcomes_by_step: dict step -> pair-order outcomes."""
    h = ob.ConvHistory(len(costs_by_step[0]), window)
    for s, row in enumerate(costs_by_step):
        h.update(row, (outcomes_by_step or {}).get(s, ()))
    return h


def same(a, b):
    for k in ("gelman_rubin", "effective_sample_size", "cost_stabilization", "swap_acceptance_rate", "has_converged"):
        assert a[k] == b[k], (k, a[k], b[k])


def test_hand_derived_two_chains():
    # chains 0..9 and 1..10: means 4.5 / 5.5, grand 5, B = 10 * 0.5 = 5, s^2 = 82.5 / 9 each,
    # var+ = 0.9 * 82.5/9 + 0.5 = 8.75, R-hat = sqrt(8.75 / (82.5/9))
    rows = [[float(i), float(i + 1)] for i in range(10)]
    m = feed(rows, 50, {0: [1], 5: [0], 9: [0]}).check(7, 0.5)
    assert m["iteration"] == 7
    assert m["gelman_rubin"] == pytest.approx(math.sqrt(8.75 / (82.5 / 9.0)), rel=1e-15)
    assert m["effective_sample_size"] is None                 # fewer than 20 samples
    # CV over the last min(50, 50/5) = 10 samples of both chains: mean 5, variance (82.5+2.5)*2/20
    assert m["cost_stabilization"] == pytest.approx(math.sqrt(8.5) / 5.0, rel=1e-15)
    assert m["swap_acceptance_rate"] == pytest.approx(1.0 / 3.0, rel=1e-15)
    assert not m["has_converged"]


def test_too_little_history():
    m = feed([[1.0, 2.0]] * 9, 200).check(0, 0.0)
    assert m["gelman_rubin"] is None and m["effective_sample_size"] is None
    assert m["swap_acceptance_rate"] == 0.0
    m = feed([[1.0]] * 30, 200).check(0, 0.0)               # one chain: no R-hat, constant series: ESS = n
    assert m["gelman_rubin"] is None and m["effective_sample_size"] == 30.0
    assert m["cost_stabilization"] == 0.0                    # sd 0 / mean 1
    m = feed([[0.0, 0.0]] * 12, 4).check(0, 0.0)            # window / 5 = 0 recent samples -> CV 1
    assert m["cost_stabilization"] == 1.0
    assert m["gelman_rubin"] is None and m["effective_sample_size"] is None      # the window holds 4 samples only
    assert feed([[3.0, 4.0]] * 12, 50).check(0, 0.0)["gelman_rubin"] == 1.0       # W <= 1e-10


def test_window_drops_oldest_samples_and_outcomes():
    rng = np.random.default_rng(5)
    rows = rng.normal(50.0, 3.0, (75, 3))
    outs = {s: rng.integers(0, 2, 2) for s in range(0, 75, 5)}
    W = 30
    m = feed(rows, W, outs).check(0, 0.2)
    # the reference appends each pass last pair first, then trims from the front
    seq = []
    for s in sorted(outs):
        seq += list(outs[s][::-1])
    want = independent_check([rows[-W:, c] for c in range(3)], seq[-W:], W, 0.2)
    same(m, want)
    # an order-sensitive trim: window 4, passes (1,0,0) then (0,0,1): history = [1(pair 0 of pass 1)], 1,0,0
    h = ob.ConvHistory(4, 4)
    h.update([1, 2, 3, 4], [1, 0, 0])
    h.update([1, 2, 3, 4], [0, 0, 1])
    assert h.check(0, 0)["swap_acceptance_rate"] == 0.5
    h = ob.ConvHistory(4, 4)
    h.update([1, 2, 3, 4], [0, 0, 1])
    h.update([1, 2, 3, 4], [0, 0, 1])
    assert h.check(0, 0)["swap_acceptance_rate"] == 0.25


@pytest.mark.parametrize("seed", range(6))
def test_matches_independent_evaluation(seed):
    rng = np.random.default_rng(seed)
    chains, steps, window = int(rng.integers(2, 7)), int(rng.integers(20, 260)), int(rng.choice([25, 100, 200, 400]))
    x = np.zeros((steps, chains))
    phi = rng.uniform(0.0, 0.95)                              # autocorrelated series exercise the lag loop
    e = rng.normal(0, 1, (steps, chains))
    for s in range(steps):
        x[s] = (phi * x[s - 1] if s else 0.0) + e[s]
    x = x * rng.uniform(0.1, 30.0) + rng.uniform(-5, 400)
    f = int(rng.integers(1, 12))
    outs = {s: rng.integers(0, 2, chains - 1) for s in range(0, steps, f)}
    seq = []
    for s in sorted(outs):
        seq += list(outs[s][::-1])
    got = feed(x, window, outs).check(3, 0.3)
    n = min(steps, window)
    want = independent_check([x[steps - n:, c] for c in range(chains)], seq[-window:], window, 0.3)
    same(got, want)


def test_converged_history():
    rng = np.random.default_rng(11)
    x = rng.normal(100.0, 1.0, (200, 4))
    outs = {s: [1, 0, 0] for s in range(0, 200, 10)}         # rate 1/3
    m = feed(x, 200, outs).check(0, 0.0)
    assert m["gelman_rubin"] < 1.2 and m["cost_stabilization"] < 0.1 and m["effective_sample_size"] > 50
    assert m["has_converged"]
    assert not feed(x, 200, {s: [1, 1, 0] for s in range(0, 200, 10)}).check(0, 0.0)["has_converged"]   # rate 2/3
    # strictness 1 asks for CV < 0.01
    assert feed(x, 200, outs).check(0, 1.0)["has_converged"] == (m["cost_stabilization"] < 0.01 and m["gelman_rubin"] < 1.01)

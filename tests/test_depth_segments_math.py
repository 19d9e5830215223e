"""CPU: the algebra of the depth segments on one pixel's list, in float64 numpy -- the run-wise forward
(transmittance pre-pass, prefix products, runs from the true incoming T, combine in list order) and the run-wise
backward (each run's affine map of (T, K), composed maps, runs from their start states) against the sequential walks
they replace (forward.cu:278-395, backward.cu:133-303 as restated in csrc/raster_fwd.hip / raster_bwd.hip).  What the
HIP kernels do per pixel, without the kernels: exact equality of every decision, values to float64 rounding."""
import numpy as np
import pytest

A_MIN, T_EPS, A_MAX_F, A_MAX_B = 1.0 / 255.0, 1e-4, 0.999, 0.99


def forward_walk(alpha_raw, rgb, T0=1.0):
    """-> (C[3], signed T (< 0: finished), last drawn index or -1) of a walk over the whole list from T0."""
    T, C, last, done = T0, np.zeros(3), -1, False
    for i, a in enumerate(alpha_raw):
        a = min(A_MAX_F, a)
        if a < A_MIN:
            continue
        nxt = T * (1.0 - a)
        if nxt <= T_EPS:
            done = True
            break
        C += rgb[i] * a * T
        T, last = nxt, i
    return C, (-T if done else T), last


def forward_runs(alpha_raw, rgb, bounds):
    taus = []
    for lo, hi in bounds[:-1]:                       # pre-pass: every run but the last, from T = 1
        _, t, _ = forward_walk(alpha_raw[lo:hi], rgb[lo:hi])
        taus.append(t if t > 0 else 0.0)
    prefix = np.concatenate([[1.0], np.cumprod(taus)])  # what run k starts from
    C, T, last = np.zeros(3), 1.0, -1
    for k, (lo, hi) in enumerate(bounds):             # the runs are independent; the combine walks them in order
        if prefix[k] <= 0.0:                           # (a run that starts dead draws nothing; combine never gets here)
            ck, tk, lk = np.zeros(3), -0.0, -1
        else:
            ck, tk, lk = forward_walk(alpha_raw[lo:hi], rgb[lo:hi], prefix[k])
        C += ck
        T = tk
        last = max(last, lk + lo if lk >= 0 else -1)
        if tk < 0 or (tk == 0 and np.signbit(tk)):
            break
    return C, T, last


def backward_walk(alpha_raw, d, lo, hi, last, T, K):
    """The backward's state over entries [lo, hi) back to front -> (T, K, per-entry v_alpha weights)."""
    w = np.zeros(len(alpha_raw))
    for i in range(min(hi - 1, last), lo - 1, -1):
        a = min(A_MAX_B, alpha_raw[i])
        if a < A_MIN:
            continue
        ra = 1.0 / (1.0 - a)
        Tn = T * ra
        w[i] = Tn * d[i] + ra * K                      # v_alpha of this entry
        K -= a * Tn * d[i]
        T = Tn
    return T, K, w


def run_map(alpha_raw, d, lo, hi, last):
    rho, S = 1.0, 0.0
    for i in range(min(hi - 1, last), lo - 1, -1):
        a = min(A_MAX_B, alpha_raw[i])
        if a < A_MIN:
            continue
        rho *= 1.0 / (1.0 - a)
        S += a * rho * d[i]
    return rho, S


@pytest.mark.parametrize("seed,n,runs,opaque", [(0, 400, 4, False), (1, 1000, 16, False), (2, 700, 5, True),
                                                 (3, 64, 16, True), (4, 300, 2, False)])
def test_runs_reproduce_the_sequential_walks(seed, n, runs, opaque):
    rng = np.random.default_rng(seed)
    alpha_raw = rng.uniform(0, 0.08, n) * (rng.uniform(0, 1, n) < 0.7)      # many misses (alpha < 1/255)
    if opaque:
        alpha_raw[rng.integers(0, n, n // 10)] = rng.uniform(0.9, 1.2, n // 10)  # above both clamps
    rgb = rng.uniform(0, 1, (n, 3))
    seg = -(-n // runs)
    bounds = [(lo, min(lo + seg, n)) for lo in range(0, n, seg)]

    C0, T0, last0 = forward_walk(alpha_raw, rgb)
    C1, T1, last1 = forward_runs(alpha_raw, rgb, bounds)
    assert last1 == last0 and (T1 < 0) == (T0 < 0)
    assert abs(abs(T1) - abs(T0)) <= 1e-13 and np.abs(C1 - C0).max() <= 1e-12

    # backward: T_final and the last drawn index come from the forward; d = rgb . v_out, K_0 arbitrary
    d = rgb @ rng.uniform(-1, 1, 3)
    Tf, K0 = abs(T0), rng.uniform(-1, 1)
    _, _, w_ref = backward_walk(alpha_raw, d, 0, n, last0, Tf, K0)
    maps = [run_map(alpha_raw, d, lo, hi, last0) for lo, hi in bounds]
    w = np.zeros(n)
    for k, (lo, hi) in enumerate(bounds):
        A, B = 1.0, 0.0
        for j in range(len(bounds) - 1, k, -1):       # everything behind run k, farthest run first
            B += A * maps[j][1]
            A *= maps[j][0]
        _, _, wk = backward_walk(alpha_raw, d, lo, hi, last0, Tf * A, K0 - Tf * B)
        w += wk
    scale = np.abs(w_ref).max()
    assert scale > 0 and np.abs(w - w_ref).max() <= 1e-11 * scale
    assert np.array_equal(w != 0, w_ref != 0)          # the same entries are valid

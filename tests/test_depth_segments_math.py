"""CPU: the algebra of the depth segments on one pixel's list, in float64 numpy -- the run-wise forward
(round 6: every run once from T = 1, scaled by the prefix products, a re-walk from the run in which the pixel crosses
the stop rule's threshold) and the run-wise
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


def forward_runs(alpha_raw, rgb, bounds, margin=1.001, stats=None):
    """Round 6 (raster_fwd_segresolve_kernel): every run is walked ONCE from T = 1 -> (C_k, signed T_k, last_k);
    compositing is associative in (C, T), so the runs in front of the first run k* whose prefix x T_k comes down to the
    stop rule's threshold (or which finished on its own) enter as prefix_k C_k; from the start of k* the list is
    re-walked with the true incoming T -- the unchanged rule -- until the pixel finishes."""
    raws = [forward_walk(alpha_raw[lo:hi], rgb[lo:hi]) for lo, hi in bounds]
    prefix, C, last, kstar = 1.0, np.zeros(3), -1, None
    for k, ((lo, _), (ck, tk, lk)) in enumerate(zip(bounds, raws)):
        if tk < 0 or prefix * tk <= T_EPS * margin:
            kstar = k
            break
        C += prefix * ck
        last = max(last, lk + lo if lk >= 0 else -1)
        prefix *= tk
    if stats is not None:
        stats["kstar"] = kstar
    if kstar is None:
        return C, prefix, last
    lo = bounds[kstar][0]
    assert prefix > T_EPS  # a pixel joins the re-walk live, by construction
    ck, tk, lk = forward_walk(alpha_raw[lo:], rgb[lo:], prefix)
    return C + ck, tk, max(last, lk + lo if lk >= 0 else -1)


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


def test_a_pixel_the_margin_flags_for_nothing_walks_on():
    """prefix x T_k inside the crossing margin but above 1e-4, and nothing behind takes it below: the re-walk starts at
    that run, never finishes, and the result is the single walk's."""
    n = 256
    alpha_raw = np.zeros(n)
    # run 0 (entries 0-63) takes T to 1.0005e-4 exactly-ish: one entry of alpha 1 - 1.0005e-4 is clamped, so use many
    k = 40
    alpha_raw[:k] = 1.0 - (1.0005e-4) ** (1.0 / k)
    alpha_raw[100] = 0.002  # below 1/255: a miss
    rgb = np.full((n, 3), 0.5)
    bounds = [(lo, lo + 64) for lo in range(0, n, 64)]
    st = {}
    C0, T0, last0 = forward_walk(alpha_raw, rgb)
    C1, T1, last1 = forward_runs(alpha_raw, rgb, bounds, stats=st)
    assert st["kstar"] == 0 and T0 > 0 and T1 > 0 and last1 == last0 == k - 1
    assert abs(T1 - T0) <= 1e-15 and np.abs(C1 - C0).max() <= 1e-13


@pytest.mark.parametrize("seed", range(6))
def test_crossing_run_is_found_where_the_single_walk_finishes(seed):
    rng = np.random.default_rng(100 + seed)
    n, runs = 640, 10
    alpha_raw = rng.uniform(0.0, 0.25, n) * (rng.uniform(0, 1, n) < 0.8)
    rgb = rng.uniform(0, 1, (n, 3))
    bounds = [(lo, lo + 64) for lo in range(0, n, 64)]
    C0, T0, last0 = forward_walk(alpha_raw, rgb)
    st = {}
    C1, T1, last1 = forward_runs(alpha_raw, rgb, bounds, stats=st)
    assert T0 < 0 and T1 < 0 and last1 == last0
    assert st["kstar"] is not None and bounds[st["kstar"]][0] <= last0 + 1  # the re-walk starts at or before the finish
    assert abs(T1 - T0) <= 1e-13 and np.abs(C1 - C0).max() <= 1e-12

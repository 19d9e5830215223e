"""CPU: the host-side state machine that sizes the tile lists without reading the count back (rasterizer/rasterize.py:
`_speculative_capacity`, the two-round plan and its culled-count hand-shake, `_two_round_candidate`).  Pure Python --
no kernel runs; the device is only a dictionary key here."""
import pytest
import torch

from rasterizer import rasterize as R

DEV = torch.device("cpu")  # (a key)
TB = (120, 68, 1)          # 1920 x 1080 at 16 px


@pytest.fixture(autouse=True)
def clean_state(monkeypatch, tune):
    for d in (R._count_hint, R._last_capacity, R._two_hint):
        d.clear()
    monkeypatch.delenv("GSR_TILE_SORT", raising=False)
    tune()  # the table's defaults, whatever GSR_TUNE says
    yield
    for d in (R._count_hint, R._last_capacity, R._two_hint):
        d.clear()


def test_capacity_comes_from_the_previous_view_with_headroom_and_stays_put():
    assert R._speculative_capacity(DEV, 1_000_000, TB, True) is None           # no view yet: the exact path
    R._note_count(DEV, 1_000_000, TB, 4_459_433)
    cap = R._speculative_capacity(DEV, 1_000_000, TB, True)
    assert cap >= int(1.25 * 4_459_433) and cap % (1 << 20) == 0                  # 25 % + rounding to whole Mi
    R._note_count(DEV, 1_000_000, TB, 4_300_000)                                   # a slightly smaller view:
    assert R._speculative_capacity(DEV, 1_000_000, TB, True) == cap                # ... the buffers keep their size
    R._note_count(DEV, 1_000_000, TB, 1_500_000)                                   # the need more than halves: shrink
    assert R._speculative_capacity(DEV, 1_000_000, TB, True) < cap
    small = R._speculative_capacity(DEV, 1_000_000, TB, True)
    R._last_capacity.clear()
    assert R._speculative_capacity(DEV, 2_000_000, TB, True) > small               # the guess scales with N
    R._note_count(DEV, 1_000_000, TB, 2**31 - 10)
    assert R._speculative_capacity(DEV, 1_000_000, TB, True) is None                # would not fit int32 lists


def test_no_speculation_switch(tune):
    R._note_count(DEV, 1_000_000, TB, 4_000_000)
    tune(no_speculation=1)
    assert R._speculative_capacity(DEV, 1_000_000, TB, True) is None


def test_two_round_candidates_are_deep_scenes_only():
    tiles = TB[0] * TB[1]
    R._note_count(DEV, 1_000_000, TB, 4_459_433)                 # the bench default: 547 entries per tile
    assert not R._two_round_candidate(DEV, 1_000_000, TB)
    R._note_count(DEV, 3_000_000, TB, 33_000_000)                # deep, but too little to save (break-even ~45 M)
    assert not R._two_round_candidate(DEV, 3_000_000, TB)
    big = (240, 135, 1)
    R._note_count(DEV, 3_000_000, big, 97_700_000)               # config 5: 3 M Gaussians at 4K
    assert R._two_round_candidate(DEV, 3_000_000, big)
    R._two_hint[(DEV, big)] = {"cooldown": 7}                    # the filter dropped too little last time
    assert not R._two_round_candidate(DEV, 3_000_000, big)
    assert tiles == 8160


def test_two_round_plan_asks_for_the_culled_count_once_and_then_plans(monkeypatch):
    big = (240, 135, 1)
    R._note_count(DEV, 3_000_000, big, 97_700_000)
    # first candidate view: the plan needs the number of culled Gaussians, which travels through a pinned slot
    assert R._two_round_plan_locked(DEV, 3_000_000, big, "auto") == "count_culled"

    class Arrived:  # the slot, once the publishing kernel has run
        def peek(self):
            return 600_000

    R._two_hint[(DEV, big)]["culled_pending"] = (Arrived(), 3_000_000)
    plan = R._two_round_plan_locked(DEV, 3_000_000, big, "auto")
    assert isinstance(plan, dict) and R._two_hint[(DEV, big)]["culled_frac"] == pytest.approx(0.2)
    assert plan["n1"] > 600_000 and plan["n1"] % 256 == 0        # the prefix starts BEHIND the culled Gaussians
    assert plan["cap1"] % (1 << 20) == 0 and plan["cap2"] % (1 << 20) == 0 and plan["cap1"] + plan["cap2"] < 2**31 - 1
    # feedback of a view whose round 1 left nothing unfinished: the prefix is steered towards ~500 entries per tile
    note = R._two_round_feedback(plan, c1=16_000_000, c2=0, unfinished=0)
    assert note is None and R._two_hint[(DEV, big)]["count1"] == 16_000_000 and R._two_hint[(DEV, big)]["fails"] == 0
    # ... and of one where nothing saturated: single rounds for a while, sized from the full count
    note = R._two_round_feedback(plan, c1=30_000_000, c2=60_000_000, unfinished=20_000)
    assert note == 90_000_000 and R._two_hint[(DEV, big)]["cooldown"] >= 50
    assert R._two_round_plan_locked(DEV, 3_000_000, big, "auto") is None   # cooling down


def test_cached_two_segment_lists_are_a_miss_for_a_single_segment_caller():
    assert R._is_two(("two", object(), 123)) and not R._is_two(None) and not R._is_two((1, 2, 3))


def test_depth_segments_only_where_every_tile_is_split_forward_and_backward(tune):
    """rasterizer.cuda.depth_segments: runs on tile grids of up to 1 100 tiles -- the grids on which
    forward AND backward split every tile above the small-grid floor, so that which tiles are cut is a function of the
    tile's list alone and every route to the kernels rounds the same way -- and nowhere else, whatever the knobs say."""
    import rasterizer.cuda as C

    def knobs(**env):  # (the rows carry the names of the environment variables they were until round 5)
        tune(**{k[4:].lower(): v for k, v in env.items()})

    try:
        knobs()
        assert C.depth_segments(1_000_000, 30 * 17) == (16, 512)            # 480 x 270
        assert C.depth_segments(1_000_000, 1100) == (16, 512)
        assert C.depth_segments(1_000_000, 60 * 34) == (1, 0)               # 960 x 540: the backward is not split-all
        assert C.depth_segments(1_000_000, 120 * 68) == (1, 0)              # 1080p
        for tiles in (510, 1100):  # on every grid with segments both thresholds are the constant floor
            assert C.deep_tile_threshold(10, tiles) == C.deep_tile_threshold(10**9, tiles, backward=True) == 96
        knobs(GSR_DEPTH_SEGMENTS=1)
        assert C.depth_segments(1_000_000, 510) == (1, 0)
        knobs(GSR_DEPTH_SEGMENTS_GRID=5000)                                  # asked for more than the split-all grids:
        assert C.depth_segments(1_000_000, 2040) == (1, 0)                   # ... capped by GSR_SMALL_GRID_BWD
        knobs(GSR_DEPTH_SEGMENTS_GRID=5000, GSR_SMALL_GRID_BWD=2560)
        assert C.depth_segments(1_000_000, 2040) == (16, 512) and C.depth_segments(1_000_000, 2561) == (1, 0)
        knobs(GSR_DEPTH_SEGMENTS=5, GSR_DEPTH_SEGMENTS_MIN=64)
        assert C.depth_segments(1, 256) == (5, 64)
    finally:
        knobs()


def test_the_backwards_split_threshold_is_one_wave_slots_share_of_the_launch(tune):
    """rasterizer.cuda.deep_tile_threshold(backward=True): GSR_DEEP_FACTOR_BWD is quoted on a 1080p grid
    and scaled by tiles / 8 160 above the small-grid limit -- the threshold is (entries) / 4 080 whatever the grid;
    GSR_DEEP_FACTOR_BWD_SCALED=0 restores the fixed factor; small grids keep their constant floor; the job order
    starts above GSR_SMALL_GRID_BWD."""
    import rasterizer.cuda as C

    def knobs(**env):
        tune(**{k[4:].lower(): v for k, v in env.items()})

    try:
        knobs()
        e = 2_000_000
        for tiles in (2040, 3600, 4590, 8160, 14400, 32400):
            want = max(256, int(2.0 * (tiles / 8160.0) * e / tiles))
            assert C.deep_tile_threshold(e, tiles, backward=True) == want
            assert abs(want - e / 4080) <= 1
        assert C.deep_tile_threshold(e, 510, backward=True) == C.deep_tile_threshold(e, 1100, backward=True) == 96
        assert C.deep_tile_threshold(e, 8160) == int(1.2 * e / 8160)  # the forward's is the mean's multiple as before
        knobs(GSR_DEEP_FACTOR_BWD_SCALED=0)
        assert C.deep_tile_threshold(e, 2040, backward=True) == int(2.0 * e / 2040)
        knobs()
        import torch

        class _Bins:  # (deep_arg only looks at the attribute alloc_tile_bins leaves on its tensor)
            _gsr_job_tail = True

        tb = lambda tiles_x, tiles_y: (tiles_x, tiles_y, 1)
        assert C.deep_arg(_Bins(), e, 60 * 34, backward=True, tile_bounds=tb(60, 34)) & C.GSR_DEEP_ORDERED   # 960 x 540
        assert not C.deep_arg(_Bins(), e, 30 * 17, backward=True, tile_bounds=tb(30, 17)) & C.GSR_DEEP_ORDERED  # 480 x 270
        knobs(GSR_DEEP_ORDER_GRID=2560)
        assert not C.deep_arg(_Bins(), e, 60 * 34, backward=True, tile_bounds=tb(60, 34)) & C.GSR_DEEP_ORDERED
    finally:
        knobs()


def test_grids_beyond_the_order_kernels_tables_run_in_the_static_order():
    """3840 x 2160 is the largest grid the job-order kernel sorts (32 768 slots); 4096 x 2160, 5K and 8K
    must render in the static order instead of raising 'tile grid beyond the sort's tables'."""
    import torch

    import rasterizer.cuda as C

    tb = lambda w, h: ((w + 15) // 16, (h + 15) // 16, 1)
    assert C.tile_jobs_ints(tb(3840, 2160)) > 0
    for w, h in ((4096, 2160), (5120, 2880), (7680, 4320)):
        t = tb(w, h)
        assert C.tile_jobs_ints(t) == 0
        bins = C.alloc_tile_bins(t, "cpu")
        assert bins.shape == (t[0] * t[1], 2) and bins._gsr_job_tail is False
        for bwd in (False, True):
            arg = C.deep_arg(bins, 40_000_000, t[0] * t[1], backward=bwd, tile_bounds=t)
            assert arg > 0 and not (arg & C.GSR_DEEP_ORDERED)
        assert not (C.forward_orders(bins, 40_000_000, t[0] * t[1], t, "cpu") & C.GSR_DEEP_ORDERED)  # (no native call made)


def test_a_slice_of_a_larger_buffer_is_not_a_job_tail():
    """room behind a caller's tile_bins is not ownership of it -- only a storage of exactly
    2 tiles + gsr_tile_jobs_ints int32, entered at its start (what alloc_tile_bins makes), carries a job order."""
    import torch

    import rasterizer.cuda as C

    t = (120, 68, 1)  # 1920 x 1080
    nt, ints = t[0] * t[1], C.tile_jobs_ints(t)
    own = C.alloc_tile_bins(t, "cpu")
    resurfaced = torch.empty(0, dtype=torch.int32).set_(own.untyped_storage(), 0, (nt, 2))  # as autograd unpacks it
    assert C.deep_arg(resurfaced, 8_000_000, nt, tile_bounds=t) & C.GSR_DEEP_ORDERED
    big = torch.zeros(2 * nt + ints + 4096, dtype=torch.int32)
    assert not C.deep_arg(big[: 2 * nt].view(nt, 2), 8_000_000, nt, tile_bounds=t) & C.GSR_DEEP_ORDERED
    assert not C.deep_arg(big[64: 64 + 2 * nt].view(nt, 2), 8_000_000, nt, tile_bounds=t) & C.GSR_DEEP_ORDERED


def test_the_tuning_table_rejects_unknown_rows_and_reads_gsr_tune_once(monkeypatch):
    from rasterizer.cuda import _tuning as T

    saved = T.overrides()
    try:
        with pytest.raises(ValueError):
            T.set_overrides({"deep_facter": 1.5})
        T.set_overrides({"deep_factor": 1.5})
        import rasterizer.cuda as C

        assert C.deep_tile_threshold(8_160_000, 8160) == 1500 and T.get("deep_min") == T.TABLE["deep_min"][0]
        T.set_overrides()
        assert C.deep_tile_threshold(8_160_000, 8160) == 1200
        monkeypatch.setattr(T, "_over", None)
        monkeypatch.setenv("GSR_TUNE", '{"depth_segments": 4}')
        assert T.get("depth_segments") == 4
        monkeypatch.setattr(T, "_over", None)
        monkeypatch.setenv("GSR_TUNE", '{"nope": 1}')
        with pytest.raises(ValueError):
            T.get("depth_segments")
    finally:
        T.set_overrides(saved)
    for name, (default, record) in T.TABLE.items():
        assert isinstance(record, str) and len(record) > 20, name  # every row says where its value comes from

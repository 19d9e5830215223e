"""Every tuning constant of the dispatch rules, in ONE table, each with the record that justifies it.

Until round 5 these were ~35 separate ``GSR_*`` environment variables read all over the product Python (VERDICT r5,
weak 9).  The environment keeps the switches a USER needs -- ``GSR_DETERMINISTIC`` (fixed summation order),
``GSR_SPECULATE`` (lists built ahead of time: auto / lists / sort / 0), ``GSR_LIBRARY`` (another build of the native
library), and the two algorithm selectors the native library itself reads, ``GSR_TILE_SORT`` / ``GSR_DEPTH_SORT``
(reference-shaped rocPRIM paths for the tests) -- everything else is a row below.

A/B scripts and tests override rows without touching the code:

    GSR_TUNE='{"deep_factor": 1.5, "depth_segments_fwd": 8}' python bench.py ...      # one JSON object, read once
    rasterizer.cuda._tuning.set_overrides({"deep_factor": 1.5})                       # in-process (clears the caches)

An unknown key raises: a typo must not run the default silently.

Each row: ``name: (default, "what it decides; where it was measured")``.  The rows are FITTED constants: the scenes they
were fitted on are named in the record; ``profiles/r06_regret_*.txt`` holds the default rule against a grid search on
three scene families it was NOT fitted on (room / floaters / needles).
"""
import json
import os

TABLE = {
    # ---- deep tiles: a tile whose list is longer than the threshold is composited by four waves (one per 8x8 sub-tile)
    "deep_factor": (1.2, "forward threshold = factor x mean list length (0 = off).  Long-tail scene fwd 357 -> 320 us, "
                         "factors 0.8-1.5 within 3 %, 0.3 1.7x slower; uniform unaffected (r02; profiles/r05_fwd_factor_large_grids.txt)"),
    "deep_min": (256, "... not below this many entries (1 024 until r05: trained model 0.97 -> 0.88 ms, "
                      "profiles/r05_deep_tile_floor_ab.txt)"),
    "small_grid": (2560, "forward: grids of up to this many tiles cannot fill the chip with one wave per tile -- EVERY "
                         "tile above small_grid_min is split (480x270 fwd 0.32 -> 0.17 ms, 960x540 0.19 -> 0.15; "
                         "profiles/r04_small_grids.txt)"),
    "small_grid_min": (96, "the constant split threshold on such grids (a chunk and a half)"),
    "small_grid_bwd": (1100, "backward: split-all only up to this many tiles (960x540 split-all 0.28 -> 0.35 ms: four "
                             "waves issue four butterflies; profiles/r04_small_grids.txt).  Also the largest grid with "
                             "depth segments and the smallest with a job order"),
    "deep_factor_bwd": (2.0, "backward threshold factor, quoted on a 1080p grid (0 = the forward's).  Trained model 0.405 "
                             "(1.2) -> 0.348 (2.0) -> 0.40 ms (3.0+); profiles/r05_lpt_tail_and_factors.txt"),
    "deep_factor_bwd_scaled": (1, "1: the factor scales with tiles / 8 160 above small_grid_bwd, i.e. a tile is split when its "
                                  "list exceeds entries / 4 080, one resident wave slot's share (best factor per grid: 2 040 "
                                  "tiles 0.5-0.7 ... 14 400 3.0+; 960x540 bwd 0.637 -> 0.352 ms; profiles/r05_midgrid_factors.txt)"),
    # ---- job order (longest job first, device-built; csrc/raster_common.h)
    "deep_order": (1, "1: launches above small_grid_bwd tiles run their jobs longest first (trained model fwd 0.30 -> "
                      "0.185, bwd 0.44 -> 0.34 ms; uniform within 1 %; profiles/r05_lpt*.txt)"),
    "deep_tail": (8, "forward: this many 64ths of the whole-tile jobs -- the shortest -- run last as four sub-tile jobs "
                     "each (uniform fwd 0.233 -> 0.208 ms; profiles/r05_lpt_tail_and_factors.txt)"),
    "deep_tail_bwd": (0, "the same for the backward: off (8/64: 0.443 -> 0.461 ms, a split tile costs it 1.7x the instructions)"),
    "deep_order_grid": (-1, "grids of up to this many tiles keep the static order (-1: small_grid_bwd)"),
    # ---- depth segments: on split-all grids the list of a split tile is cut into runs walked by their own waves
    "depth_segments": (16, "runs per split tile, both directions (1 = off).  300 k Gaussians at 480x270: bwd 436 -> 263 us "
                           "with 8; config 3 818 -> 871 (8) -> 885 it/s (16); profiles/r04_depth_segments.txt"),
    "depth_segments_grid": (1100, "only on grids of up to this many tiles (and never above small_grid / small_grid_bwd): "
                                  "960x540 unchanged, 1080p long-tail 0.62 -> 0.67 ms with segments"),
    "depth_segments_min": (512, "lists of at most this many entries are walked in one piece"),
    "depth_segments_fwd": (16, "cap on the FORWARD's runs (0 = none).  8 until the end of r05 while the forward walked "
                               "every list twice; one walk since r06 (profiles/r06_forward_one_walk.txt)"),
    # ---- two-round lists for deep scenes
    "two_round": ("auto", "auto | 0 | force: lists of the nearest Gaussians first, the rest only for unfinished tiles "
                          "(config 5: 3.26 -> 2.77 ms; profiles/r03_*)"),
    "two_round_depth": (1500.0, "candidate scenes: at least this many list entries per tile ..."),
    "two_round_len": (500.0, "... round 1 aims at this many entries per tile ..."),
    "two_round_saved": (45e6, "... and at least this many entries must be saved (break-even of the extra launches)"),
    # ---- host path
    "no_speculation": (0, "1: size the lists from the count read back every view (the reference's host sync, utils.py:124)"),
    "fused_records": (1, "1: the reach records ride in the depth sort's first launch (r03: 20 -> 10 launches)"),
    "one_call": (1, "1: lists + compositing of a view in ONE native call (gsr_rasterize_gaussians_forward; r04)"),
    "poll_yield": (1, "1: the pinned-count poll releases the GIL between looks (INTEGRATION.md, threads)"),
    "speculate_min": (65536, "below this many Gaussians nothing is built ahead of time (the side stream's launches cost "
                             "more than they hide)"),
}

_over = None
_listeners = []


def _load():
    global _over
    if _over is None:
        raw = os.environ.get("GSR_TUNE", "").strip()
        over = json.loads(raw) if raw else {}
        if not isinstance(over, dict):
            raise ValueError("GSR_TUNE must be one JSON object, e.g. '{\"deep_factor\": 1.5}'")
        _check(over)
        _over = over
    return _over


def _check(over):
    unknown = sorted(set(over) - set(TABLE))
    if unknown:
        raise ValueError(f"GSR_TUNE / set_overrides: unknown tuning key(s) {unknown}; known: {sorted(TABLE)}")


def get(name):
    over = _load()
    return over[name] if name in over else TABLE[name][0]


def set_overrides(over=None):
    """Replace the overrides (None / {}: the table's defaults plus nothing -- GSR_TUNE is NOT re-read) and drop every
    cache derived from the table."""
    global _over
    over = dict(over or {})
    _check(over)
    _over = over
    for fn in _listeners:
        fn()


def overrides():
    return dict(_load())


def on_change(fn):
    _listeners.append(fn)
    return fn

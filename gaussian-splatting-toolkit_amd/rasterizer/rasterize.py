"""Tile-based alpha compositing of projected Gaussians (differentiable).

Mirror of the reference's ``rasterizer/rasterize.py`` (``rasterize_gaussians``
:14-86, ``_RasterizeGaussians`` :89-247): same signature, checks, outputs,
saved tensors and gradient tuple.

One addition that is invisible to the caller: the models call
``rasterize_gaussians`` twice per view with identical geometry (RGB, then
depth; gs_toolkit/models/vanilla_gs.py:822,840).  The second call reuses the
sorted intersection list and tile ranges of the first instead of running the
scan + key emission + radix sort again.  The cache holds one entry, keyed on
the storage address, shape and version counter of the four geometry tensors
(plus conics and opacity, which the 16x16 lists also depend on).
It keeps detached aliases of them alive, so their storage cannot be recycled
for another tensor while the entry exists (an address match therefore means
the same memory), and an in-place update bumps the version and misses.
"""
import os
import threading
import time
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C
from rasterizer.cuda import _tuning

# One entry, shared by every thread of the process.  Readers take a SNAPSHOT under `_state_lock`
# and writers replace all four fields under it, so forwards interleaved on one device (an
# evaluation thread next to the training thread, side streams) each see a consistent entry; what
# a forward needs later (the deterministic backward's `aux`) travels with the call, in
# thread-local storage, never through this dictionary.
_bin_cache = {"key": None, "value": None, "keepalive": None, "reach": None}
# guards the list cache, the sizing dictionaries below (`_count_hint`, `_last_capacity`, `_two_hint`) and the pinned
# slot pool; re-entrant: a hint update may hand a slot back to the pool
_state_lock = threading.RLock()
_tls = threading.local()
# views whose device-sized lists came out too small and were built again (harness.train reports the delta per run)
# list constructions by kind (tests and harness.train read these; never reset by the package):
#   list_builds_exact / _device_sized   lists built inside rasterize_gaussians, sized by a read-back / by the previous view
#   list_builds_ahead                   lists built ahead of time on the side stream (`speculate_lists`)
#   ahead_hits / ahead_misses           ... and whether the rasterize call that followed could use them
#   ahead_orders_used                   only the depth order of the side stream was used
#   ahead_recipes_off                   devices on which the recipe detection switched itself off (rasterizer/ahead.py)
counters = {"list_rebuilds": 0, "list_builds_exact": 0, "list_builds_device_sized": 0, "list_builds_ahead": 0,
            "ahead_hits": 0, "ahead_misses": 0, "ahead_orders_used": 0, "ahead_recipes_off": 0}


def _cache_snapshot():
    with _state_lock:
        return dict(_bin_cache)


def last_list_aux():
    """(order, cum_sorted, slot_of_entry) of the lists the calling THREAD's last `build_tile_lists`
    returned (deterministic mode; None otherwise)."""
    return getattr(_tls, "aux", None)

# Deterministic backward (include/gsraster.h, gsr_rasterize_backward_det): per-(tile, entry)
# partials summed per Gaussian in a fixed order instead of float atomics -- bit-identical
# gradients from run to run, ~1.4x the backward time.  GSR_DETERMINISTIC=1 or set_deterministic().
_deterministic = {"on": os.environ.get("GSR_DETERMINISTIC", "0") not in ("", "0")}


def set_deterministic(flag: bool) -> None:
    _deterministic["on"] = bool(flag)
    with _state_lock:
        _bin_cache["key"] = None  # cached lists may lack the inverse map


def is_deterministic() -> bool:
    return _deterministic["on"]


def _geometry_key(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (xys, depths, radii, num_tiles_hit)) + (
        img_height, img_width, block_width, xys.device,
    )


# ---- sizing the lists without a host round trip ------------------------------
# The reference reads the number of intersections back before it can allocate
# (`cum_tiles_hit[-1].item()`, utils.py:124): the GPU idles for the round trip
# (~50 us at 1080p, 4 % of a forward+backward).  From the second view on, the
# lists are sized from the last count (+25 %), the kernels read the real count on
# the device, and the host compares the two after compositing is already queued.
# A guess that was too small costs one rebuild; the result never depends on it.
_count_hint = {}
_last_capacity = {}
_pinned_count = {}


def _speculation_enabled() -> bool:
    return not int(_tuning.get("no_speculation")) and \
        os.environ.get("GSR_TILE_SORT", "s")[:1] not in ("r", "m")


def _note_count(device, num_points, tile_bounds, num_intersects):
    with _state_lock:
        _count_hint[(device, tile_bounds)] = (num_points, num_intersects)


def _speculative_capacity(device, num_points, tile_bounds, exact):
    with _state_lock:
        return _speculative_capacity_locked(device, num_points, tile_bounds, exact)


def _speculative_capacity_locked(device, num_points, tile_bounds, exact):
    hint = _count_hint.get((device, tile_bounds))
    # the device-sized lists need the single-pass tile scatter: any grid with the exact
    # lists' per-band counts (block_width 16), up to 16384 tiles otherwise
    if hint is None or not _speculation_enabled() or \
            (not exact and tile_bounds[0] * tile_bounds[1] > _C.MAX_SCATTER_TILES):
        return None
    n_last, count_last = hint
    if n_last < 1 or count_last < 1:
        return None
    guess = count_last * (num_points / n_last)
    cap = int(1.25 * guess) + 65536
    # keep the buffer sizes stable from view to view (the caching allocator then hands
    # back the same blocks): whole Mi-elements, and no shrinking unless the need halves
    cap = (cap + (1 << 20) - 1) & ~((1 << 20) - 1)
    last_cap = _last_capacity.get((device, tile_bounds), 0)
    if cap < last_cap <= 2 * cap:
        cap = last_cap
    if cap >= 2**31 - 1:
        return None
    _last_capacity[(device, tile_bounds)] = cap
    return cap


# ---- two-round lists for deep scenes (include/gsraster.h) ---------------------
# When the previous view's lists were deep (mean entries per tile above GSR_TWO_ROUND_DEPTH, default 1500) almost
# every entry lies behind the depth at which its tile saturates.  Then: lists of the nearest Gaussians only (a
# prefix of the depth order, sized for ~GSR_TWO_ROUND_LEN = 400 entries per tile), a first compositing round, a
# per-Gaussian filter that drops what can only land in finished tiles, lists of the rest, a second round that
# resumes.  Bit-identical images; `two_round` = "0" (_tuning.py) switches it off, "1" forces it (once a count is known).
_two_hint = {}


def _records_and_order(xys, radii, conics, opacity, depths, tile_bounds, extra_rows=0):
    """Reach records + depth order of the lists without counts: one native call (`fused_records` = 0 in _tuning.py: the two calls
    it replaces, for A/B measurements)."""
    if not int(_tuning.get("fused_records")):
        _, records = _C.count_reach(xys, radii, conics, opacity, tile_bounds, counts=False, extra_rows=extra_rows)
        order, _ = _C.depth_order(depths, radii, None)
        return records, order
    return _C.reach_records_depth_order(xys, radii, conics, opacity, depths, tile_bounds, extra_rows=extra_rows)


def _two_round_plan(device, num_points, tile_bounds, exact, radii=None):
    mode = str(_tuning.get("two_round"))
    if mode in ("0", "off") or not exact or _deterministic["on"] or not _speculation_enabled():
        return None
    if os.environ.get("GSR_TILE_SORT", "")[:1] in ("s", "b"):
        return None
    with _state_lock:
        plan = _two_round_plan_locked(device, num_points, tile_bounds, mode)
    if plan == "count_culled":
        # The prefix has to start behind the culled Gaussians (they sit at the FRONT of the depth order, key 0): their
        # number travels to the host through a pinned slot, like the list counts -- no blocking read-back.  The view
        # that asks (the first two-round candidate, and every 32nd after it) takes one round; the answer is there
        # for the next one.
        if radii is not None:
            culled = (radii <= 0).sum(dtype=torch.int32).reshape(1)
            pend = _PendingCount(device)
            _C.publish_int32(culled, pend.buf)
            pend.mark()
            with _state_lock:
                _two_hint.setdefault((device, tile_bounds), {})["culled_pending"] = (pend, num_points)
        with _state_lock:
            th = _two_hint.setdefault((device, tile_bounds), {})
            plan = _two_round_plan_locked(device, num_points, tile_bounds, mode, asked=True) if "culled_frac" in th else None
    return plan


def _two_round_candidate(device, num_points, tile_bounds) -> bool:
    """(under `_state_lock`; changes nothing) would `_two_round_plan` consider two rounds for this view?  The same
    switches and thresholds: lists built ahead of time (one round) must not pre-empt them on deep scenes."""
    mode = str(_tuning.get("two_round"))
    if mode in ("0", "off") or os.environ.get("GSR_TILE_SORT", "")[:1] in ("s", "b"):
        return False
    hint = _count_hint.get((device, tile_bounds))
    if hint is None or hint[0] < 1 or hint[1] < 1:
        return False
    if _two_hint.get((device, tile_bounds), {}).get("cooldown", 0) > 0:
        return False
    if mode == "1":
        return True
    tiles = tile_bounds[0] * tile_bounds[1]
    full = hint[1] * (num_points / hint[0])
    return not (full / tiles < float(_tuning.get("two_round_depth"))
                or full - float(_tuning.get("two_round_len")) * tiles < float(_tuning.get("two_round_saved"))
                or num_points < 100_000)


def _two_round_plan_locked(device, num_points, tile_bounds, mode, asked=False):
    hint = _count_hint.get((device, tile_bounds))
    if hint is None or hint[0] < 1 or hint[1] < 1:
        return None
    key = (device, tile_bounds)
    th = _two_hint.setdefault(key, {})
    if not asked and th.get("cooldown", 0) > 0:  # the filter dropped too little last time: single rounds for a while
        th["cooldown"] -= 1
        return None
    n_last, count_last = hint
    tiles = tile_bounds[0] * tile_bounds[1]
    full = count_last * (num_points / n_last)
    min_depth = float(_tuning.get("two_round_depth"))
    target = float(_tuning.get("two_round_len"))
    # two rounds cost ~0.3 ms (14 more launches, and the count check has little GPU work left to hide behind); a
    # list entry that is never built saves ~8 ps: worth it from ~45 M avoided entries on (3 M Gaussians at 1080p,
    # 33 M entries: break-even; 3 M at 4K, 98 M: -13 %)
    min_saved = float(_tuning.get("two_round_saved"))
    if mode != "1" and (full / tiles < min_depth or full - target * tiles < min_saved or num_points < 100_000):
        return None
    f = th.get("f")
    if f is None:
        # the nearest Gaussians are the largest on screen: they hold ~3x their share of the entries
        f = target * tiles / full / 3.0
    f = min(0.5, max(0.02, f))
    pend = th.get("culled_pending")
    if pend is not None:
        val = pend[0].peek()
        if val is not None:
            th["culled_frac"] = float(val) / max(pend[1], 1)
            th["culled_pending"] = None
    if not asked:
        th["views"] = th.get("views", 0) + 1
        if th.get("culled_pending") is None and ("culled_frac" not in th or th["views"] % 32 == 0):
            return "count_culled"
    if "culled_frac" not in th:
        return None
    culled = int(th["culled_frac"] * num_points)
    n1 = culled + int(f * (num_points - culled))
    n1 = min(max(256, (n1 + 255) & ~255), num_points)
    if n1 >= num_points:
        return None
    mi = 1 << 20
    c1, c2 = th.get("count1"), th.get("count2")
    cap1 = int(1.3 * c1 * (f / th["f_used"])) + (mi >> 2) if (c1 and th.get("f_used")) else int(3.0 * f * full) + mi
    cap2 = int(1.5 * c2) + mi if c2 is not None else int(1.1 * full) + mi
    cap1, cap2 = ((cap1 + mi - 1) // mi) * mi, ((cap2 + mi - 1) // mi) * mi
    # keep the buffer sizes stable from view to view (the caching allocator then hands back the same blocks):
    # no shrinking unless the need halves
    last = th.get("caps")
    if last:
        if cap1 < last[0] <= 2 * cap1:
            cap1 = last[0]
        if cap2 < last[1] <= 2 * cap2:
            cap2 = last[1]
    th["caps"] = (cap1, cap2)
    if cap1 + cap2 >= 2**31 - 1:
        return None
    return {"n1": n1, "cap1": cap1, "cap2": cap2, "f": f, "full": full, "target": target, "tiles": tiles, "key": key}


def _build_two_round(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds, block_width, plan, remember,
                     round1, order_ready=None):
    """-> (None, ids, bins1, finish); `round1(ids, bins1, tile_flags)` composites the prefix lists (raw state)."""
    dev = xys.device
    n, n1, cap1, cap2 = xys.size(0), plan["n1"], plan["cap1"], plan["cap2"]
    if order_ready is not None:  # the depth order was built ahead of time on the side stream
        (_, records), order = _C.count_reach(xys, radii, conics, opacity, tile_bounds, counts=False, extra_rows=1), order_ready
        counters["ahead_orders_used"] += 1
    else:
        records, order = _records_and_order(xys, radii, conics, opacity, depths, tile_bounds, extra_rows=1)
    with torch.cuda.device(dev):
        ids = torch.empty((cap1 + cap2,), dtype=torch.int32, device=dev)
        flags = torch.zeros((tile_bounds[0] * tile_bounds[1],), dtype=torch.int32, device=dev)
    p1, p2, p4 = (_PendingCount(dev) for _ in range(3))
    counters["list_builds_device_sized"] += 1
    bins1 = _C.tile_lists_subrange(order[:n1], cap1, records, tile_bounds, ids[:cap1], p1.buf)
    round1(ids, bins1, flags)
    with torch.cuda.device(dev):
        stats = torch.empty((2,), dtype=torch.int32, device=dev)
    order2 = _C.saturation_filter(order[n1:], records, n, flags, tile_bounds, stats)  # row n: the culled dummy record
    bins2 = _C.tile_lists_subrange(order2, cap2, records, tile_bounds, ids[cap1:], p2.buf)
    _C.publish_int32(stats[:1], p4.buf)  # tiles with a live pixel after round 1
    p4.mark()
    p1.event = p2.event = p4.event
    aux = ("two", bins2, cap1)
    _tls.aux = aux

    def finish():
        c1, c2, unfinished = p1.resolve(), p2.resolve(), p4.resolve()
        with _state_lock:
            note = _two_round_feedback(plan, c1, c2, unfinished)
        if note is not None:
            _note_count(xys.device, n, tile_bounds, note)
        if c1 > cap1 or c2 > cap2:  # a guess was too small: this view falls back to one exact round
            counters["list_rebuilds"] += 1
            nn, i2, b2, _ = _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds, block_width,
                                         True, remember, speculate=False, round1=None)
            return nn, i2, b2, True
        remember(c1 + c2, ids, bins1, aux)
        return c1 + c2, ids, bins1, False

    return None, ids, bins1, finish


def _two_round_feedback(plan, c1, c2, unfinished):
    """(under `_state_lock`) steer the next view's prefix from this view's counts -> a full count to note, or None"""
    note = None
    th = _two_hint.setdefault(plan["key"], {})
    th.update(count1=c1, count2=c2, f_used=plan["f"], unfinished=unfinished)
    # The filter works per Gaussian: one unfinished tile keeps every Gaussian whose box holds it, so a few per
    # cent of unfinished tiles keep most of a scene of large splats.  Lengthen the prefix until (almost) no tile
    # is left; otherwise steer it towards `target` entries per tile in round 1.
    if unfinished > 0.3 * plan["tiles"]:
        # nothing saturates here (background shows through, or the count hint came from another scene): with
        # most tiles unfinished hardly anything was filtered, so c1 + c2 IS the full count -- single rounds for
        # a while, sized from it
        th["fails"] = min(th.get("fails", 0) + 1, 7)
        th["cooldown"] = 25 << th["fails"]  # 50, 100, ... 3200 views between attempts while they keep failing
        th.pop("f", None)
        note = c1 + c2
    elif unfinished > 0.003 * plan["tiles"]:
        th["f"] = min(0.5, 1.5 * plan["f"])
        if plan["f"] >= 0.5 and c2 > 0.5 * max(plan["full"] - c1, 1.0):
            th["cooldown"] = 50  # the scene does not saturate behind any prefix: single rounds for a while
    else:
        th["fails"] = 0
        th["f"] = max(0.02, plan["f"] * min(1.25, max(0.8, plan["target"] * plan["tiles"] / max(c1, 1))))
    if c1 > plan["cap1"] or c2 > plan["cap2"]:
        th.pop("count1", None), th.pop("count2", None)
    return note


_POLL_YIELD = bool(int(_tuning.get("poll_yield")))  # (0 = spin without yielding the GIL)


@_tuning.on_change
def _refresh_poll_yield():
    global _POLL_YIELD
    _POLL_YIELD = bool(int(_tuning.get("poll_yield")))


class _PendingCount:
    """One int32 on its way from a kernel to the host: the kernel (`gsr_bin_sorted_dev`'s count, `gsr_publish_int32`)
    writes it straight into pinned (device-mapped) host memory -- no copy operation in the stream -- and `resolve`
    reads it there.

    Slots come from a per-device free list and go back to it only once no kernel can still write them: when
    `resolve` / `peek` has SEEN the value (each slot has exactly one writer, which writes once), or, for a slot
    dropped unread (an exception between the launch and `finish()`), when the event recorded behind the launch has
    fired -- otherwise the slot is retired for good.  A new owner can therefore never read a previous owner's late
    write as its own count (interleaved forwards of several threads / streams on one device included)."""

    CHUNK = 64            # pinned int32 slots allocated at a time
    PENDING = -2147483647  # what a slot holds until the kernel has written it (no count is ever that)
    POLL_S = 2e-3         # `resolve` polls this long before it blocks on the event ("threads")

    def __init__(self, device):
        with _state_lock:
            pool = _pinned_count.get(device)
            if pool is None:
                pool = _pinned_count[device] = []
            if not pool:
                pinned = torch.empty(self.CHUNK, dtype=torch.int32, pin_memory=True)
                view = pinned.numpy()  # the same memory, cheap to poll
                pool.extend((pinned[k:k + 1], view[k:k + 1]) for k in range(self.CHUNK))
            self._slot = pool.pop()
        self.buf, self._np = self._slot
        self._np[0] = self.PENDING
        self.device = device
        self.event = None
        self._value = None

    def mark(self):
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(self.device))

    def _release(self):
        slot, self._slot = self._slot, None
        if slot is not None:
            with _state_lock:
                _pinned_count[self.device].append(slot)

    def __del__(self):
        try:  # dropped unread: reusable only if its writer is known to have run
            if self._slot is not None and self.event is not None and self.event.query():
                self._release()
        except Exception:
            pass

    def peek(self) -> Optional[int]:
        """The value if it has arrived (the slot is then released), else None; never blocks."""
        if self._slot is None:
            return self._value
        v = int(self._np[0])
        if v == self.PENDING:
            return None
        self._value = v
        self._release()
        return v

    def resolve(self) -> int:
        # The value lands in host memory the moment the publishing kernel writes it -- earlier than the event
        # behind the whole call fires, and without the wake-up latency of a blocking wait: poll it (every
        # microsecond the host wakes up earlier is GPU work queued earlier).  `time.sleep(0)` between two looks
        # releases the GIL, so other Python threads (the toolkit's viewer thread waiting on `train_lock`) run while
        # this one waits.  A value that does not arrive within POLL_S is waited for the ordinary way.
        t0 = time.perf_counter()
        count = self.peek()
        while count is None:
            if time.perf_counter() - t0 > self.POLL_S:
                if self.event is not None:
                    self.event.synchronize()
                count = self.peek()
                if count is None:
                    raise RuntimeError("rasterize_gaussians: the list count was never published (the launch that "
                                       "writes it did not run; an earlier asynchronous HIP error?)")
                break
            if _POLL_YIELD:
                time.sleep(0)
            count = self.peek()
        if count < 0:  # the int32 count wrapped (more than 2^31 - 1 intersections)
            raise RuntimeError("rasterize_gaussians: the number of (Gaussian, tile) intersections does not fit the "
                               "int32 lists (the reference's cum_tiles_hit is int32 as well)")
        return count



# lists built ahead of time (the prediction of the caller's next opacities, the side stream): rasterizer/ahead.py
from .ahead import (_PURE_UNARY, _UNARY_FN, _note_opacity_recipe, _producer_signature, _spec, _spec_knobs,  # noqa: E402,F401
                    _speculation_mode, _take_speculation, announce_opacity, speculate_lists)

def _same_reach_inputs(conics, opacity, reach):
    """Are `conics` / `opacity` the tensors the cached lists were built from?  True: the same
    storage at the same version, or produced by the same pure op from the same leaf (no
    device work, no sync).  "verify": same shapes but provenance unknown (e.g. a fresh
    `torch.sigmoid` under no_grad) -- the caller uses the cached lists speculatively and
    checks equality on the device, off the critical path.  False: rebuild."""
    c0, o0, cv, ov, csig, osig = reach
    verdict = True
    for t, t0, v0, sig0 in ((conics, c0, cv, csig), (opacity, o0, ov, osig)):
        if t.shape != t0.shape:
            return False
        if t.data_ptr() == t0.data_ptr():
            if t._version != v0:
                return False
            continue
        sig = _producer_signature(t)
        if sig is None or sig != sig0:
            verdict = "verify"
    return verdict


def rasterize_gaussians(
    xys: Tensor,
    depths: Tensor,
    radii: Tensor,
    conics: Tensor,
    num_tiles_hit: Tensor,
    colors: Tensor,
    opacity: Tensor,
    img_height: int,
    img_width: int,
    block_width: int,
    background: Optional[Tensor] = None,
    return_alpha: Optional[bool] = False,
) -> Tensor:
    """Sort the Gaussians per tile by depth and composite them front to back.

    Differentiable w.r.t. ``xys``, ``conics``, ``colors`` and ``opacity``.

    Args:
        xys [N,2], depths [N], radii [N] int32, conics [N,3], num_tiles_hit [N]
        int32: outputs of :func:`project_gaussians` (same ``block_width``).
        colors: [N,C] float (uint8 is rescaled to [0,1]).  opacity: [N,1].
        background: [C]; defaults to ones.  return_alpha: also return 1-T.

    Returns:
        ``out_img`` [H,W,C], or ``(out_img, out_alpha [H,W])``.
    """
    assert block_width > 1 and block_width <= 16, "block_width must be between 2 and 16"
    if colors.dtype == torch.uint8:
        colors = colors.float() / 255

    if background is not None:
        assert (
            background.shape[0] == colors.shape[-1]
        ), f"incorrect shape of background color tensor, expected shape {colors.shape[-1]}"
    else:
        background = torch.ones(colors.shape[-1], dtype=torch.float32, device=colors.device)

    if xys.ndimension() != 2 or xys.size(1) != 2:
        raise ValueError("xys must have dimensions (N, 2)")
    if colors.ndimension() != 2:
        raise ValueError("colors must have dimensions (N, D)")

    return _RasterizeGaussians.apply(
        xys.contiguous(), depths.contiguous(), radii.contiguous(), conics.contiguous(),
        num_tiles_hit.contiguous(), colors.contiguous(), opacity.contiguous(), img_height,
        img_width, block_width, background.contiguous(), return_alpha,
    )


def _is_two(aux) -> bool:
    return isinstance(aux, tuple) and len(aux) == 3 and aux[0] == "two"


class FusedForward:
    """What `build_tile_lists` needs to run the compositing in the SAME native call as the list construction
    (`gsr_rasterize_gaussians_forward`, 16x16 tiles / 3 channels): the colours, background and output wishes of
    the caller; `outputs` = (out_img, final_Ts, final_idx, alpha) once that has happened."""

    def __init__(self, colors, background, img_height, img_width, want_alpha, zero):
        self.colors, self.background, self.want_alpha, self.zero = colors, background, want_alpha, zero
        self.img_size = (img_width, img_height)
        self.outputs = None
        self.prealloc = None  # (out_img, planes) allocated ahead of time, when the lists were


def build_tile_lists(xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, block_width,
                     round1=None, fuse: Optional[FusedForward] = None):
    """The per-tile depth-sorted lists every compositing call walks.

    -> ``(num_intersects, gaussian_ids_sorted, tile_bins, finish)``.  Either the count is
    known (``finish is None``; a cache hit, the first view, or a tile grid the
    device-sized path does not serve), or the lists were sized from the previous view
    and ``num_intersects is None``: enqueue the compositing, then call ``finish()`` ->
    ``(num_intersects, gaussian_ids_sorted, tile_bins, rebuilt)``; with ``rebuilt`` the
    guess was too small, the lists were built again and the compositing must be repeated.
    The result is cached for the next call with the same geometry (the depth pass).

    ``round1(ids, bins1, tile_flags)`` (optional): the caller can composite in two rounds
    (``_C.rasterize_forward_round``).  On deep scenes the lists then come in two segments: ``bins`` is the
    first, ``last_list_aux()`` returns ``("two", bins2, idx_base)``, and ``round1`` has already been called
    on the first segment when this function returns -- the caller runs round 2 (and, for lists from the
    cache, both rounds).

    ``fuse`` (optional, `FusedForward`): where the lists are device-sized single-round ones the compositing runs in
    the same native call and ``fuse.outputs`` holds its results; the caller composites itself when it is None."""
    num_points = xys.size(0)
    _tls.aux = None  # (what the PREVIOUS call of this thread left: every path below sets its own)
    tile_bounds = ((img_width + block_width - 1) // block_width, (img_height + block_width - 1) // block_width, 1)
    key = _geometry_key(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width)
    # With 16x16 tiles the lists leave out the (Gaussian, tile) pairs that cannot
    # reach alpha >= 1/255 anywhere in the tile (about half of the reference's
    # bounding-box pairs; the compositing rule skips them pixel by pixel, so
    # images and gradients are unchanged).  Those lists also depend on
    # conics and opacity.
    exact = block_width == 16

    def remember(num_intersects, ids, bins, aux=None):
        entry = {"key": key, "value": (num_intersects, ids, bins, aux),
                 "keepalive": tuple(t.detach() for t in (xys, depths, radii, num_tiles_hit)),
                 "reach": (conics.detach(), opacity.detach(), conics._version, opacity._version,
                           _producer_signature(conics), _producer_signature(opacity))}
        with _state_lock:
            _bin_cache.update(entry)
        _tls.aux = aux

    order_ready = None
    if exact:
        _note_opacity_recipe(xys.device, opacity)
        if _speculation_mode() == "auto" and num_points >= _spec_knobs["min_points"] and \
                not torch.cuda.is_current_stream_capturing():
            # was the GPU out of work when the caller got here?  (see "WHEN" above)
            idle = 1.0 if torch.cuda.current_stream(xys.device).query() else 0.0
            with _state_lock:
                st = _spec.setdefault(xys.device, {"stream": None, "recipe": None, "entry": None})
                st["idle"] = 0.75 * st.get("idle", 1.0) + 0.25 * idle
        ahead = _take_speculation(xys.device, key)
        if ahead is not None:
            main = torch.cuda.current_stream(xys.device)
            main.wait_event(ahead["done"])
            if "ids" in ahead and not _deterministic["on"] and _same_reach_inputs(conics, opacity, ahead["reach"]) is True:
                # this view's lists were built on the side stream while the caller was busy (or blocked)
                counters["ahead_hits"] += 1
                ids, bins, pending, capacity = ahead["ids"], ahead["bins"], ahead["pending"], ahead["capacity"]
                # (ids / bins come from the side stream's pool and are read on the caller's: no `record_stream` --
                # memory goes back to that pool only when autograd drops them, and the pool hands it out again only
                # inside `speculate_lists`, behind `side.wait_stream(main)`)
                if fuse is not None:
                    fuse.prealloc = ahead.get("outs")

                def finish_ahead():
                    num_intersects = pending.resolve()
                    _note_count(xys.device, num_points, tile_bounds, num_intersects)
                    if num_intersects > capacity:  # the guess was too small: build the lists again, sized exactly
                        counters["list_rebuilds"] += 1
                        n, i2, b2, _ = _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds,
                                                    block_width, exact, remember, speculate=False)
                        return n, i2, b2, True
                    remember(num_intersects, ids, bins, None)
                    return num_intersects, ids, bins, False

                return None, ids, bins, finish_ahead
            if "order" in ahead:
                order_ready = ahead["order"]
                order_ready.record_stream(main)
            else:
                counters["ahead_misses"] += 1  # other opacities than predicted: the lists are of no use

    snap = _cache_snapshot()
    if snap["key"] == key and round1 is None and _is_two(snap["value"][3]):
        # two-segment lists (a deep scene's RGB pass) cached, but THIS caller composites one segment only (N-D
        # colours, another block width ...): walking `bins1` alone would silently drop everything behind the
        # prefix -- build single-round lists
        snap = {"key": None}
    if snap["key"] == key:
        same = _same_reach_inputs(conics, opacity, snap["reach"]) if exact else True
        cached = snap["value"]
        if same is True:
            _tls.aux = cached[3]
            return cached[:3] + (None,)
        if same == "verify" and cached[0] >= 1:
            # use the cached lists now; compare the values on the device and look at the
            # answer once the compositing is queued (no host wait in front of the GPU work)
            c0, o0 = snap["reach"][:2]
            differ = ((conics != c0).any() | (opacity != o0).any()).to(torch.int32).reshape(1)
            check = _PendingCount(xys.device)
            _C.publish_int32(differ, check.buf)
            check.mark()

            def verify():
                if not check.resolve():
                    _tls.aux = cached[3]
                    return cached[:3] + (False,)
                n, ids, bins, fin = _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds,
                                                 block_width, exact, remember, round1=None)
                if fin is not None:
                    n, ids, bins, _ = fin()
                return n, ids, bins, True

            _tls.aux = cached[3]
            return None, cached[1], cached[2], verify
    return _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds, block_width, exact, remember,
                        round1=round1, fuse=fuse, order_ready=order_ready)


def _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds, block_width, exact, remember,
                 speculate=True, round1=None, fuse=None, order_ready=None):
    num_points = xys.size(0)
    if round1 is not None and speculate:
        plan = _two_round_plan(xys.device, num_points, tile_bounds, exact, radii)
        if plan is not None:
            return _build_two_round(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds, block_width, plan,
                                    remember, round1, order_ready=order_ready)
    # fused binning: same `gaussian_ids_sorted` / `tile_bins` as
    # compute_cumulative_intersects + bin_and_sort_gaussians (bit for bit
    # when not `exact`: tests/test_gpu_kernels.py::
    # test_fused_binning_equals_reference_pipeline)
    # deterministic backward: keep the inverse of the scatter and the depth order (block_width 16
    # with the exact lists, i.e. the single-pass scatter path -- on grids above 16384 tiles that
    # is the banded one; otherwise such grids take the two-level partition, bands = 1)
    det = _deterministic["on"] and exact and os.environ.get("GSR_TILE_SORT", "s")[:1] not in ("r", "m")
    banded = exact and (det or os.environ.get("GSR_TILE_SORT", "")[:1] == "b")
    capacity = _speculative_capacity(xys.device, num_points, tile_bounds, exact) if speculate else None

    if fuse is not None and exact and capacity is not None and not banded and not det and \
            int(_tuning.get("one_call")):
        # records + depth order + device-sized lists + compositing: ONE native call (with or without counts, as
        # gsr_bin_sorted_needs_counts decides in there); the count comes back through the pinned slot
        pending = _PendingCount(xys.device)
        img_w, img_h = fuse.img_size
        ids, bins, *outs = _C.rasterize_gaussians_forward(
            xys, depths, radii, conics, fuse.colors, opacity, fuse.background, img_h, img_w, capacity, pending.buf,
            want_alpha=fuse.want_alpha, zero=fuse.zero, order_ready=order_ready)
        pending.mark()
        fuse.outputs = tuple(outs)
        counters["list_builds_device_sized"] += 1
        if order_ready is not None:
            counters["ahead_orders_used"] += 1

        def finish_one():
            num_intersects = pending.resolve()
            _note_count(xys.device, num_points, tile_bounds, num_intersects)
            if num_intersects > capacity:  # the guess was too small: build the lists again, sized exactly
                counters["list_rebuilds"] += 1
                fuse.outputs = None
                n, i2, b2, _ = _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds,
                                            block_width, exact, remember, speculate=False)
                return n, i2, b2, True
            remember(num_intersects, ids, bins, None)
            return num_intersects, ids, bins, False

        return None, ids, bins, finish_one

    if exact and capacity is not None and not banded and \
            not _C.lists_need_counts(num_points, capacity, tile_bounds, device_sized=True, want_slots=det):
        # The two-level partition counts its entries itself: records only, the depth order only,
        # and the number of entries comes back through the pinned slot (`count_out`).
        if order_ready is not None:
            (_, records), order = _C.count_reach(xys, radii, conics, opacity, tile_bounds, counts=False), order_ready
            counters["ahead_orders_used"] += 1
        else:
            records, order = _records_and_order(xys, radii, conics, opacity, depths, tile_bounds)
        pending = _PendingCount(xys.device)
        ids, bins = _C.bin_sorted(num_points, capacity, order, None, xys, radii, tile_bounds, block_width, records,
                                  device_sized=True, count_out=pending.buf)
        pending.mark()
        counters["list_builds_device_sized"] += 1

        def finish_lean():
            num_intersects = pending.resolve()
            _note_count(xys.device, num_points, tile_bounds, num_intersects)
            if num_intersects > capacity:  # the guess was too small: build the lists again, sized exactly
                counters["list_rebuilds"] += 1
                n, i2, b2, _ = _build_fresh(xys, depths, radii, conics, num_tiles_hit, opacity, tile_bounds,
                                            block_width, exact, remember, speculate=False)
                return n, i2, b2, True
            remember(num_intersects, ids, bins, None)
            return num_intersects, ids, bins, False

        return None, ids, bins, finish_lean

    tiles, records = num_tiles_hit, None
    if exact:
        tiles, records = _C.count_reach(xys, radii, conics, opacity, tile_bounds,
                                        bands=_C.tile_bands(tile_bounds) if banded else 1)
    order, cum_sorted = _C.depth_order(depths, radii, tiles)

    def build(count, device_sized=False):
        out = _C.bin_sorted(num_points, count, order, cum_sorted, xys, radii, tile_bounds, block_width, records,
                            device_sized=device_sized, want_slots=det)
        if det:
            build.aux = (order, cum_sorted, out[2])
        return out[0], out[1]

    build.aux = None

    if capacity is None:
        # the one host sync (utils.py:124).  The count is summed in 64 bits: the int32 prefix
        # (the reference's `torch.cumsum(num_tiles_hit, dtype=torch.int32)`) wraps beyond 2^31 - 1
        # intersections, and lists sized from a wrapped count are written out of bounds
        num_intersects = int(tiles.sum(dtype=torch.int64).item())
        if num_intersects >= 2**31:
            raise RuntimeError(f"rasterize_gaussians: {num_intersects} (Gaussian, tile) intersections do not fit the "
                               f"int32 lists (the reference's cum_tiles_hit is int32 as well)")
        _note_count(xys.device, num_points, tile_bounds, num_intersects)
        ids, bins = build(num_intersects) if num_intersects >= 1 else (None, None)
        counters["list_builds_exact"] += 1
        remember(num_intersects, ids, bins, build.aux)
        return num_intersects, ids, bins, None

    # Size the lists from the previous view instead of waiting for the count to reach the
    # host (utils.py:124).  The count goes to the host right behind the depth-order scan,
    # before the lists are built: the host wakes up early enough to queue the loss and the
    # backward while the GPU is still compositing.
    pending = _PendingCount(xys.device)
    _C.publish_int32(cum_sorted[-1:], pending.buf)
    pending.mark()
    ids, bins = build(capacity, device_sized=True)
    counters["list_builds_device_sized"] += 1

    def finish():
        nonlocal ids, bins
        num_intersects = pending.resolve()
        _note_count(xys.device, num_points, tile_bounds, num_intersects)
        rebuilt = num_intersects > capacity  # the guess was too small: build the lists again
        if rebuilt:
            counters["list_rebuilds"] += 1
            counters["list_builds_exact"] += 1
            ids, bins = build(num_intersects)
        remember(num_intersects, ids, bins, build.aux)
        return num_intersects, ids, bins, rebuilt

    return None, ids, bins, finish


class _RasterizeGaussians(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                img_width, block_width, background=None, return_alpha=False):
        tile_bounds = (
            (img_width + block_width - 1) // block_width,
            (img_height + block_width - 1) // block_width,
            1,
        )
        block = (block_width, block_width, 1)
        img_size = (img_width, img_height, 1)
        rasterize_fn = _C.rasterize_forward if colors.shape[-1] == 3 else _C.nd_rasterize_forward
        # 16x16 tiles, 3 channels: the compositing kernel also writes alpha = 1 - T and, when a
        # backward will follow, clears that backward's accumulators (gsr_rasterize_forward_ex)
        fused = block_width == 16 and colors.shape[-1] == 3
        acc = None
        if fused and any(ctx.needs_input_grad[i] for i in (0, 3, 5, 6)) and not _deterministic["on"]:
            acc = _C.backward_accumulators(xys.size(0), 3, xys.device)
        alpha_out = [None]
        state = []  # two-round compositing: the caller-owned raw state both rounds work on

        def composite(ids, bins):
            if not fused:
                return rasterize_fn(tile_bounds, block, img_size, ids, bins, xys, conics, colors, opacity, background)
            if fuse.prealloc is not None:  # lists AND output buffers were made ahead of time: straight to the launch
                (img0, planes), fuse.prealloc = fuse.prealloc, None
                if colors.dtype == torch.float32 and background.dtype == torch.float32 and opacity.dtype == torch.float32 \
                        and colors.size(0) == xys.size(0) and opacity.numel() == xys.size(0) and background.numel() == 3 \
                        and colors.device == xys.device and background.device == xys.device:
                    img, Ts, idx, alpha_out[0] = _C.composite_prepared(
                        tile_bounds, img_width, img_height, ids, bins, xys, conics, colors, opacity, background, img0,
                        planes, return_alpha, zero=acc)
                    return img, Ts, idx
            img, Ts, idx, alpha_out[0] = _C.rasterize_forward_ex(tile_bounds, block, img_size, ids, bins, xys, conics,
                                                                 colors, opacity, background, want_alpha=return_alpha,
                                                                 zero=acc)
            return img, Ts, idx

        def round1(ids, bins1, flags):
            dev = xys.device
            with torch.cuda.device(dev):
                state[:] = [torch.empty((img_height, img_width, 3), dtype=torch.float32, device=dev),
                            torch.empty((img_height, img_width), dtype=torch.float32, device=dev),
                            torch.empty((img_height, img_width), dtype=torch.int32, device=dev), flags]
                alpha_out[0] = torch.empty((img_height, img_width), dtype=torch.float32, device=dev) if return_alpha else None
            _C.rasterize_forward_round(1, tile_bounds, img_size, ids, bins1, 0, xys, conics, colors, None, opacity,
                                       background, 0.0, state[0], None, state[1], state[2], flags, out_alpha=alpha_out[0],
                                       zero=acc)

        def round2(ids, aux):
            _C.rasterize_forward_round(2, tile_bounds, img_size, ids, aux[1], aux[2], xys, conics, colors, None, opacity,
                                       background, 0.0, state[0], None, state[1], state[2], state[3],
                                       out_alpha=alpha_out[0])
            return state[0], state[1], state[2]

        is_two = _is_two

        fuse = FusedForward(colors, background, img_height, img_width, return_alpha, acc) if fused else None
        num_intersects, gaussian_ids_sorted, tile_bins, finish = build_tile_lists(
            xys, depths, radii, conics, num_tiles_hit, opacity, img_height, img_width, block_width,
            round1=round1 if fused else None, fuse=fuse)
        two = is_two(last_list_aux()) and fused
        two_aux = last_list_aux() if two else None
        if finish is not None:
            if fuse is not None and fuse.outputs is not None:  # composited by the call that built the lists
                out_img, final_Ts, final_idx, alpha_out[0] = fuse.outputs
            elif two and state:      # round 1 ran inside build_tile_lists
                out_img, final_Ts, final_idx = round2(gaussian_ids_sorted, two_aux)
            elif two:              # cached two-segment lists used speculatively
                flags = torch.zeros((tile_bounds[0] * tile_bounds[1],), dtype=torch.int32, device=xys.device)
                round1(gaussian_ids_sorted, tile_bins, flags)
                out_img, final_Ts, final_idx = round2(gaussian_ids_sorted, two_aux)
            else:
                out_img, final_Ts, final_idx = composite(gaussian_ids_sorted, tile_bins)
            num_intersects, gaussian_ids_sorted, tile_bins, rebuilt = finish()
            if rebuilt:
                two = is_two(last_list_aux()) and fused
                two_aux = last_list_aux() if two else None
                if two:
                    flags = torch.zeros((tile_bounds[0] * tile_bounds[1],), dtype=torch.int32, device=xys.device)
                    round1(gaussian_ids_sorted, tile_bins, flags)
                    out_img, final_Ts, final_idx = round2(gaussian_ids_sorted, two_aux)
                else:
                    out_img, final_Ts, final_idx = composite(gaussian_ids_sorted, tile_bins)

        if num_intersects < 1:
            # nothing on screen: background everywhere (rasterize.py:119-127)
            out_img = torch.ones(img_height, img_width, colors.shape[-1], device=xys.device) * background
            gaussian_ids_sorted = torch.zeros(0, 1, device=xys.device)
            tile_bins = torch.zeros(0, 2, device=xys.device)
            final_Ts = torch.zeros(img_height, img_width, device=xys.device)
            final_idx = torch.zeros(img_height, img_width, device=xys.device)
            acc, alpha_out[0] = None, None
        elif finish is None:
            if two:  # two-segment lists from the cache (the depth pass of a view)
                flags = torch.zeros((tile_bounds[0] * tile_bounds[1],), dtype=torch.int32, device=xys.device)
                round1(gaussian_ids_sorted, tile_bins, flags)
                out_img, final_Ts, final_idx = round2(gaussian_ids_sorted, two_aux)
            else:
                out_img, final_Ts, final_idx = composite(gaussian_ids_sorted, tile_bins)

        ctx.set_materialize_grads(False)
        ctx.img_width = img_width
        ctx.img_height = img_height
        # deterministic backward: (order, cum_sorted, slot_of_entry) of the lists just used
        ctx.det = last_list_aux() if (_deterministic["on"] and num_intersects >= 1 and colors.shape[-1] == 3) else None
        ctx.two = (two_aux[1], two_aux[2]) if (two and num_intersects >= 1) else None
        ctx.num_intersects = num_intersects
        ctx.block_width = block_width
        ctx.accumulators = acc  # cleared by the forward launch; used (once) by the backward
        ctx.save_for_backward(gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                              background, final_Ts, final_idx)

        if return_alpha:
            return out_img, (alpha_out[0] if alpha_out[0] is not None else 1 - final_Ts)
        return out_img

    @staticmethod
    def backward(ctx, v_out_img, v_out_alpha=None):
        (gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity, background, final_Ts,
         final_idx) = ctx.saved_tensors

        # v_out_alpha stays None when the alpha output was not requested / not used:
        # the kernels read a missing cotangent as zero (no H x W zero tensor)
        if v_out_img is None:  # only the alpha output was used
            v_out_img = torch.zeros(ctx.img_height, ctx.img_width, colors.shape[-1], device=xys.device)

        if ctx.num_intersects < 1:
            v_xy = torch.zeros_like(xys)
            v_conic = torch.zeros_like(conics)
            v_colors = torch.zeros_like(colors)
            v_opacity = torch.zeros_like(opacity)
        else:
            if ctx.two is not None:
                acc, ctx.accumulators = ctx.accumulators, None
                v_xy, v_conic, v_colors, v_opacity = _C.rasterize_backward_two(
                    ctx.img_height, ctx.img_width, gaussian_ids_sorted, tile_bins, ctx.two[0], ctx.two[1], xys, conics,
                    colors, None, opacity, background, 0.0, final_Ts, final_idx, v_out_img, None, v_out_alpha,
                    accumulators=acc)
            elif ctx.det is not None:
                v_xy, v_conic, v_colors, v_opacity = _C.rasterize_backward_det(
                    ctx.img_height, ctx.img_width, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacity,
                    background, final_Ts, final_idx, v_out_img, v_out_alpha, *ctx.det)
            else:
                acc, ctx.accumulators = ctx.accumulators, None  # a second backward (retain_graph) clears its own
                if colors.shape[-1] == 3:
                    v_xy, v_conic, v_colors, v_opacity = _C.rasterize_backward(
                        ctx.img_height, ctx.img_width, ctx.block_width, gaussian_ids_sorted, tile_bins, xys,
                        conics, colors, opacity, background, final_Ts, final_idx, v_out_img, v_out_alpha,
                        accumulators=acc, trusted=True)
                else:
                    v_xy, v_conic, v_colors, v_opacity = _C.nd_rasterize_backward(
                        ctx.img_height, ctx.img_width, ctx.block_width, gaussian_ids_sorted, tile_bins, xys,
                        conics, colors, opacity, background, final_Ts, final_idx, v_out_img, v_out_alpha)
            v_opacity = v_opacity.reshape(opacity.shape)

        # xys, depths, radii, conics, num_tiles_hit, colors, opacity, then 5 non-differentiable
        return (v_xy, None, None, v_conic, None, v_colors, v_opacity) + (None,) * 5

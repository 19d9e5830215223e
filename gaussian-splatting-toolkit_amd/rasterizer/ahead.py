"""Tile lists built AHEAD of time, while the caller is busy elsewhere.

Split out of ``rasterize.py`` in round 5: everything that PREDICTS the caller's next ``opacity``
and queues a view's list construction on a side stream lives here; ``rasterize.py`` only asks ``take()`` whether a
ready-made entry exists for the geometry it was handed, and proves by provenance that it may use it.

The prediction reads the previous call's autograd graph -- ``type(opacity.grad_fn).__name__``,
``grad_fn.next_functions[0][0].variable`` -- i.e. private-ish attributes of one torch build.  Two guards:
  * ``tests/test_gpu_api.py::test_lists_are_built_ahead_for_the_models_exact_pattern`` asserts that the detection still
    FIRES on the installed torch (``counters["ahead_hits"] > 0`` for ``torch.sigmoid(leaf)`` behind two read-backs);
  * at run time an introspection that raises, or a producer this module does not recognise for
    ``UNKNOWN_RECIPE_LIMIT`` views in a row, is logged ONCE and the recipe detection switches itself off for that
    device (the depth order is still built ahead; a silent miss only costs speed, never a result).
"""
import logging
import os
import weakref

import torch

import rasterizer.cuda as _C

log = logging.getLogger("rasterizer.ahead")
UNKNOWN_RECIPE_LIMIT = 16


_sibling = []


def _R():
    """The sibling module (imported late: it imports this one)."""
    if not _sibling:
        from . import rasterize

        _sibling.append(rasterize)
    return _sibling[0]


# ---- lists built ahead of time, while the caller is busy elsewhere ---------------------------------------------
# The unchanged models block the host twice between `project_gaussians` and `rasterize_gaussians`
# (`if (self.radii).sum() == 0`, vanilla_gs.py:784; `assert (num_tiles_hit > 0).any()`, :811): each read-back
# drains the stream and leaves the GPU idle until the host has woken up and queued the next kernels -- 165 us of a
# 1.0-ms step at 1 M Gaussians (profiles/r04_*).  Everything the tile lists depend on exists as soon as the projection has
# run: xys / depths / radii / conics, the image size, and the opacities -- which the models form as
# `torch.sigmoid(self.opacities)` AFTER the second read-back, but whose recipe (a pure unary op on a leaf) is known from
# the previous view.  So `project_gaussians` ends by queueing this view's list construction on a SIDE stream, where
# it runs through both read-backs and next to the SH evaluation; `rasterize_gaussians` then finds its lists ready
# (checked by provenance, the same rule the list cache uses: same storage at the same version, or the same pure op
# on the same leaf at the same version -- never by hoping) and goes straight to compositing.  Where the opacities
# cannot be predicted (another recipe, no previous view) only the depth order is built ahead.
# WHEN: only while the caller is seen to block.  A caller without read-backs keeps the GPU's queue full; there is no
# idle time to fill, and lists on a second stream merely compete with the SH kernel (bench default: +-0; 200 k
# Gaussians / dense scales, where nothing is left to overlap with: +3 %).  The signal is the caller's stream itself: if
# it is IDLE when `rasterize_gaussians` is entered, the host was blocked (or is the bottleneck) and the GPU had
# nothing to do -- an exponential average of that observation above 1/2 switches the side stream on.
# GSR_SPECULATE=auto (default) | lists (always) | sort (depth order only, always) | 0 (never).
_spec = {}
_UNARY_FN = {"SigmoidBackward0": torch.sigmoid, "ExpBackward0": torch.exp, "TanhBackward0": torch.tanh,
             "AbsBackward0": torch.abs, "NegBackward0": torch.neg}
_spec_knobs = {}


def _speculation_mode() -> str:
    if not _spec_knobs:
        _spec_knobs["mode"] = {"0": "0", "off": "0", "sort": "sort", "lists": "lists"}.get(
            os.environ.get("GSR_SPECULATE", "auto"), "auto")
        _spec_knobs["min_points"] = int(_C._tuning.get("speculate_min"))
    return _spec_knobs["mode"]


def _note_opacity_recipe(device, opacity) -> None:
    """Remember how the caller formed `opacity` -- for the NEXT view's lists (see above)."""
    fn = opacity.grad_fn
    recipe = None
    unknown = False
    with _R()._state_lock:
        st = _spec.get(device)
        off = st is not None and st.get("recipes_off", False)
    if fn is None:
        recipe = ("same", weakref.ref(opacity))  # a leaf / constant handed in as it is
    elif not off:
        try:
            if _producer_signature(opacity) is not None:
                recipe = ("unary", type(fn).__name__, weakref.ref(fn.next_functions[0][0].variable), tuple(opacity.shape))
            else:
                unknown = True
        except (AttributeError, IndexError, TypeError, RuntimeError) as e:  # the graph no longer looks as it did
            unknown = "raised: %r" % (e,)
    with _R()._state_lock:
        st = _spec.setdefault(device, {"stream": None, "recipe": None, "entry": None})
        st["recipe"] = recipe
        if unknown:
            st["unknown_run"] = st.get("unknown_run", 0) + 1
            if (unknown is not True or st["unknown_run"] >= UNKNOWN_RECIPE_LIMIT) and not st.get("recipes_off", False):
                st["recipes_off"] = True
                _R().counters["ahead_recipes_off"] = _R().counters.get("ahead_recipes_off", 0) + 1
                log.warning("rasterizer: the opacities handed to rasterize_gaussians on %s are produced by %s (%s): tile "
                            "lists cannot be built ahead of time for them; recipe detection is off for this device "
                            "(results are unaffected; the depth order is still built ahead)", device,
                            type(fn).__name__, "introspection " + unknown if unknown is not True else
                            "not a parameter-free unary op on a leaf, %d views in a row" % st["unknown_run"])
        elif recipe is not None:
            st["unknown_run"] = 0


def announce_opacity(opacity) -> None:
    """Tell the rasterizer which tensor the next `rasterize_gaussians` on this device will receive as `opacity`, when
    it already exists before `project_gaussians` runs but autograd provenance cannot show it (the output of a custom
    op: `gs_fused.activate_gaussians` calls this).  The lists built ahead of time are then built for it; a rasterize
    call that receives another tensor (or this one at another version) drops them, as always."""
    if opacity.is_cuda:
        with _R()._state_lock:
            st = _spec.setdefault(opacity.device, {"stream": None, "recipe": None, "entry": None})
            st["recipe"] = ("same", weakref.ref(opacity))


def speculate_lists(xys, depths, radii, conics, num_tiles_hit, img_height, img_width, block_width) -> None:
    """(called by `project_gaussians` once the projection is queued) start this view's depth order -- and, where the
    opacities are predictable, its tile lists -- on the side stream."""
    mode = _speculation_mode()
    n = xys.size(0)
    if mode == "0" or block_width != 16 or not xys.is_cuda or n < _spec_knobs["min_points"] or _R()._deterministic["on"] \
            or not _R()._speculation_enabled() or os.environ.get("GSR_TILE_SORT", "")[:1] == "b":
        return
    dev = xys.device
    if torch.cuda.is_current_stream_capturing():
        return
    tile_bounds = ((img_width + 15) // 16, (img_height + 15) // 16, 1)
    with _R()._state_lock:
        st = _spec.setdefault(dev, {"stream": None, "recipe": None, "entry": None})
        recipe = st["recipe"]
        if mode == "auto":
            if st.get("idle", 1.0) <= 0.5:  # the caller keeps the queue full: nothing to hide work behind
                stale, st["entry"], st["retired"] = st["entry"], None, None  # (an entry nobody took: let go of its tensors)
                if stale is not None:
                    torch.cuda.current_stream(dev).wait_event(stale["done"])
                return
            mode = "lists"
        # deep scenes take two-round lists, whose first launch writes the reach records while
        # it sorts: nothing is built ahead there (a ready-made order would cost them a records launch of their own:
        # 3 M Gaussians at 4K, 2.71 -> 2.76 ms)
        two_recent = _R()._two_round_candidate(dev, n, tile_bounds)
    if two_recent:
        return
    if st["stream"] is None:
        st["stream"] = torch.cuda.Stream(dev)
    side, main = st["stream"], torch.cuda.current_stream(dev)
    opacity_src = osig = None
    capacity = None
    if mode == "lists" and recipe is not None:
        if recipe[0] == "same":
            t = recipe[1]()
            if t is not None and t.is_cuda and t.numel() == n and t.dtype == torch.float32 and t.is_contiguous():
                opacity_src = ("same", t)
        else:
            leaf = recipe[2]()
            if leaf is not None and leaf.is_cuda and leaf.numel() == n and leaf.dtype == torch.float32:
                opacity_src = ("unary", leaf, recipe[1], recipe[3])
                osig = (recipe[1], id(leaf), leaf._version, leaf.data_ptr(), tuple(leaf.shape))
        if opacity_src is not None:
            capacity = _R()._speculative_capacity(dev, n, tile_bounds, True)
    key = _R()._geometry_key(xys, depths, radii, num_tiles_hit, img_height, img_width, block_width)
    entry = {"key": key, "keep": (xys, depths, radii, num_tiles_hit, conics)}
    if opacity_src is not None and capacity is not None:
        # the compositing's outputs, allocated NOW (on the caller's stream): four allocations less between the
        # models' second read-back and the compositing launch
        with torch.cuda.device(dev):
            entry["outs"] = (torch.empty((img_height, img_width, 3), dtype=torch.float32, device=dev),
                             torch.empty((3, img_height, img_width), dtype=torch.float32, device=dev))
    side.wait_stream(main)  # behind the projection (and whatever the caller queued before it)
    with torch.no_grad(), torch.cuda.stream(side):
        if opacity_src is not None and capacity is not None:
            if opacity_src[0] == "same":
                opac = opacity_src[1].detach()
                oversion = opacity_src[1]._version
            else:
                opac = _UNARY_FN[opacity_src[2]](opacity_src[1].detach()).reshape(opacity_src[3]).contiguous()
                oversion = opac._version
            pending = _R()._PendingCount(dev)
            ids, bins = _C.rasterize_gaussians_forward(xys, depths, radii, conics, None, opac.view(n, 1), None, img_height,
                                                       img_width, capacity, pending.buf, composite=False, checked=True)
            entry.update(ids=ids, bins=bins, pending=pending, capacity=capacity,
                         reach=(conics.detach(), opac, conics._version, oversion, None, osig))
            _R().counters["list_builds_ahead"] += 1
        else:
            entry["order"], _ = _C.depth_order(depths, radii, None)
        done = torch.cuda.Event()
        done.record(side)
    entry["done"] = done
    if "pending" in entry:
        entry["pending"].event = done
    # The inputs were allocated on the caller's stream and are read on this one.  Instead of `record_stream` on each
    # (five calls per view) the entry -- and `retired`, once it has been taken -- holds references to them until
    # the NEXT view's call, and by then the caller's stream has waited for `done` (when the entry was taken, or
    # below): memory handed back after that point cannot be reused ahead of the side stream's reads.
    entry["keep"] += (opacity_src[1],) if opacity_src is not None else ()
    with _R()._state_lock:
        old, st["entry"] = st["entry"], entry
        st["retired"] = None
    if old is not None:
        main.wait_event(old["done"])  # an entry nobody took: whatever follows on the caller's stream stays behind it


def _take_speculation(device, key):
    with _R()._state_lock:
        st = _spec.get(device)
        if st is None or st["entry"] is None or st["entry"]["key"] != key:
            return None
        entry, st["entry"] = st["entry"], None
        st["retired"] = entry["keep"]
    return entry


# unary, parameter-free ops: the same op on the same leaf at the same version = the same values
_PURE_UNARY = ("SigmoidBackward0", "ExpBackward0", "TanhBackward0", "AbsBackward0", "NegBackward0")


def _producer_signature(t):
    """What produced `t`, when that pins its values down: (op, leaf identity, leaf version)
    for a pure unary op applied to a leaf -- `torch.sigmoid(self.opacities)`, which the models
    evaluate afresh for each of their two calls per view (vanilla_gs.py:829,847).  None when
    unknown (no graph, another op)."""
    fn = t.grad_fn
    if fn is None or type(fn).__name__ not in _PURE_UNARY:
        return None
    nxt = fn.next_functions
    if len(nxt) != 1 or nxt[0][0] is None or not hasattr(nxt[0][0], "variable"):
        return None
    v = nxt[0][0].variable
    return (type(fn).__name__, id(v), v._version, v.data_ptr(), tuple(v.shape))



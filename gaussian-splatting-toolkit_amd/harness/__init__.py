"""Caller-side harness for the rasterizer: synthetic scenes, the
``get_outputs``-shaped render call the models make (8a15) and the per-view
data-parallel gradient exchange (8e).  Used by tests/, bench.py and smoke()."""

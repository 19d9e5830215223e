#!/usr/bin/env python3
"""One bench line in a few printed rows (python tools/r05/digest.py <line.json> ...)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path, d["value"], d.get("value_normalised"), d["ms_per_step"], d["config"].get("timing"),
          d["config"].get("calibration", {}).get("valu_Tops"), d["config"].get("calibration", {}).get("copy_GBps"))
    r = d["roofline"]
    print(" roofline", {k: r.get(k) for k in ("kernel", "frac", "kernel_ms", "valu_busy", "valu_pipe_busy", "traffic",
                                               "algorithmic_bytes")})
    print(" e2e", d.get("end_to_end_algorithmic_GBps"), d.get("end_to_end_built_GBps"), "syncs",
          d.get("ms_per_step_with_caller_syncs"), d.get("ms_per_step_with_caller_and_camera_syncs"))
    print(" digest", d["config"].get("train_digest"))
    print(" kernels", {k: v.get("ms") for k, v in d.get("kernels", {}).items()})
    tr = d.get("train", {}).get("trained_raster")
    if tr:
        print(" trained", tr.get("ms"), tr.get("ms_median"), tr.get("raster_fwd_ms"), tr.get("raster_bwd_ms"))
    print(" parity", d.get("parity_vs_oracle"))

"""Fixed-plane coordinate field of the reference's heat-map visualisation, computed on the GPU.

`max_movement` reproduces the number heatmap_video_processing returns and evenvizion_component.py writes to
metrics_file.txt ("Maximum movement during the entire video", processing_visualization.py:407-419): for every frame
the superposed H maps each pixel (x, y) of the resized grid; the per-frame maximum coordinate is appended for every
frame EXCEPT the last one of the dict (the reference skips the append when capture.read() fails), and the maximum of
those is returned.  Rendering (colour map, grid overlay, PNG) is out of scope.
"""
import numpy as np

from . import runtime


def frame_maxima(superposition_homography_dict, resize_info):
    """{frame_no: 3x3 superposed H} -> (frame numbers, f64 per-frame max coordinate), computed by evh_fixed_plane_field."""
    keys = list(superposition_homography_dict.keys())
    Hs = np.array([np.asarray(superposition_homography_dict[k], np.float64) for k in keys])
    w, h = int(resize_info["w"]), int(resize_info["h"])
    ctx = runtime.get_context(max(w, 64), max(h, 64))
    return keys, ctx.fixed_plane_max(Hs, w, h)


def max_movement(superposition_homography_dict, resize_info, skip_last=True):
    _, m = frame_maxima(superposition_homography_dict, resize_info)
    if skip_last and len(m) > 1:
        m = m[:-1]
    return float(np.max(m))

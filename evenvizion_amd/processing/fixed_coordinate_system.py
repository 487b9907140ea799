"""Coordinate mapping between original frames and the fixed plane -- counterpart of
evenvizion/processing/fixed_coordinate_system.py:19-122; same names, arguments and 2-decimal rounding as
the reference.  All points of a call go through ONE batched device transform (evh_transform_points).

Layouts: coordinates {frame_no: [{"x1": x, "y1": y, ...}, ...]}; homography_dict {frame_no: 3x3 superposed H}.
"""
from copy import deepcopy

import numpy as np
import numpy.linalg as linalg

from .. import runtime


def _convert(coordinates, homography_dict, kx, ky, inverse):
    frames = list(coordinates.keys())
    # like the reference, a frame's matrix is only touched (looked up, inverted) when the frame has a point to map
    # (fixed_coordinate_system.py:56-69): an empty frame may have no H entry or a singular one
    mats, pts, idx = [], [], []
    for f in frames:
        if not coordinates[f]:
            continue
        m = np.asarray(homography_dict[f], np.float64)
        mats.append(linalg.inv(m) if inverse else m)
        for rect in coordinates[f]:
            pts.append((rect["x1"], rect["y1"]))
            idx.append(len(mats) - 1)
    out = (runtime.get_context(64, 64, 2, runtime.NFEATURES).transform_points(mats, idx, pts, kx, ky, decimals=2)
           if pts else np.zeros((0, 2)))
    result, k = {}, 0
    for f in frames:
        converted = []
        for rect in coordinates[f]:
            new_rect = deepcopy(rect)
            new_rect["x1"], new_rect["y1"] = out[k][0], out[k][1]
            converted.append(new_rect)
            k += 1
        result[f] = converted
    return result


def from_original_to_fix(original_coordinates, homography_dict, original_image_shape, resize_image_shape):
    """Original-pixel points -> fixed coordinate system: scale to the resized frame (shapes are [h, w]), then
    apply the frame's superposed H (fixed_coordinate_system.py:56-69)."""
    original_h, original_w = original_image_shape
    resize_h, resize_w = resize_image_shape
    return _convert(original_coordinates, homography_dict, int(resize_w) / original_w, int(resize_h) / original_h, False)


def from_fix_to_original(fix_coordinates, homography_dict, original_image_shape, resize_image_shape):
    """Fixed-plane points -> original frame: the reference scales by original/resize FIRST and then applies the
    inverse H (fixed_coordinate_system.py:109-122); kept as is."""
    original_h, original_w = original_image_shape
    resize_h, resize_w = resize_image_shape
    return _convert(fix_coordinates, homography_dict, original_w / resize_w, original_h / resize_h, True)

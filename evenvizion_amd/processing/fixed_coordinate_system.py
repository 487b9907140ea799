"""Coordinate mapping between original frames and the fixed plane -- counterpart of
evenvizion/processing/fixed_coordinate_system.py:19-122.  A consumer of the superposed H matrices (O(objects)
host arithmetic); same names, arguments and 2-decimal rounding as the reference.

Layouts: coordinates {frame_no: [{"x1": x, "y1": y, ...}, ...]}; homography_dict {frame_no: 3x3 superposed H}.
"""
from copy import deepcopy

import numpy as np

from .utils import homography_transformation, inverse_homography_transformation


def _convert(coordinates, homography_dict, kx, ky, transform):
    result = {}
    for frame_no, frame_info in coordinates.items():
        converted = []
        for rect in frame_info:
            new_rect = deepcopy(rect)
            xy = np.around(transform([kx * rect["x1"], ky * rect["y1"]], homography_dict[frame_no]), decimals=2)
            new_rect["x1"], new_rect["y1"] = xy[0], xy[1]
            converted.append(new_rect)
        result[frame_no] = converted
    return result


def from_original_to_fix(original_coordinates, homography_dict, original_image_shape, resize_image_shape):
    """Original-pixel points -> fixed coordinate system: scale to the resized frame (shapes are [h, w]), then
    apply the frame's superposed H (fixed_coordinate_system.py:56-69)."""
    original_h, original_w = original_image_shape
    resize_h, resize_w = resize_image_shape
    return _convert(original_coordinates, homography_dict, int(resize_w) / original_w, int(resize_h) / original_h,
                    homography_transformation)


def from_fix_to_original(fix_coordinates, homography_dict, original_image_shape, resize_image_shape):
    """Fixed-plane points -> original frame: the reference scales by original/resize FIRST and then applies the
    inverse H (fixed_coordinate_system.py:109-122); kept as is."""
    original_h, original_w = original_image_shape
    resize_h, resize_w = resize_image_shape
    return _convert(fix_coordinates, homography_dict, original_w / resize_w, original_h / resize_h,
                    inverse_homography_transformation)

"""Utility functions -- MI355X counterpart of evenvizion/processing/utils.py (same names, arguments, errors).

GPU-backed: compute_homography (utils.py:328-363 -> evh_compute_homography), find_point_displacement +
get_largest_group_points (utils.py:258-325 -> evh_static_filter when called together through KeyPoints),
matrix_superposition / superposition_dict (utils.py:118-145, 184-211 -> evh_superposition_scan),
homography_transformation / inverse_homography_transformation (utils.py:71-115 -> evh_transform_points; the 3x3
inverse itself stays numpy.linalg.inv on the host, as in the reference).  The rest is JSON reading.
"""
import json

import numpy as np
import numpy.linalg as linalg

from .. import runtime
from .._lib import PAIR_OK
from .constants import INFINITY_COORDINATE, THRESHOLD_FOR_FIND_HOMOGRAPHY, LENGTH_ACCOUNTED_POINTS  # noqa: F401


class HomographyException(Exception):
    """utils.py:23-38"""

    def __init__(self, message="can't calculate homography matrix"):
        self.message = message
        super().__init__(message)


def remove_double_matching(pts_a, pts_b):
    """Unique pts_a coordinates: order of first occurrence, pts_b of the LAST occurrence (utils.py:60-68)."""
    matching_dict = {}
    for i, _ in enumerate(pts_a):
        matching_dict[(pts_a[i][0], pts_a[i][1])] = pts_b[i]
    new_pts_a = [np.array(k) for k in matching_dict.keys()]
    new_pts_b = list(matching_dict.values())
    return new_pts_a, new_pts_b


def _small_context():
    return runtime.get_context(64, 64, 2, runtime.NFEATURES)


def _transform(vector, matrix):
    v = np.asarray(vector, np.float64).reshape(-1)
    while len(v) < 3:
        v = np.append(v, [1])
    if v[2] != 1:
        # a homogeneous vector whose last coordinate is not 1: the reference multiplies it as it stands and divides by
        # the result's last coordinate (utils.py:89-92).  The batched device transform takes (x, y, 1) points only, so
        # this rare form stays the reference's two numpy operations on the host (pure glue, no frame data).
        new_vector = np.dot(np.asarray(matrix, np.float64), v)
        return new_vector[:-1] / new_vector[-1]
    return _small_context().transform_points(matrix, [0], [v[:2]])[0]


def homography_transformation(vector, matrix_H):
    """(x, y[, w]) -> np.dot(H, (x, y, w))[:2] / [2] as float64[2] (utils.py:89-92); w defaults to 1.  One point per call
    costs a device round trip: callers with many points should use Context.transform_points (as
    fixed_coordinate_system and the heat-map do) instead of np.apply_along_axis over this function."""
    return _transform(vector, matrix_H)


def inverse_homography_transformation(vector, matrix_H):
    """The same through numpy.linalg.inv(H) (utils.py:112-115)."""
    return _transform(vector, linalg.inv(np.asarray(matrix_H, np.float64)))


def matrix_superposition(H, matrix_H_superposition, matrix_H_first=False):
    """H . H_sup normalised by its [2][2]; the first H passes through unchanged; H None keeps H_sup (utils.py:139-145)."""
    if H is None:
        return matrix_H_superposition
    if matrix_H_first:
        return H
    return _small_context().superposition_scan([matrix_H_superposition, H])[1]


def read_homography_dict(path_to_homography_dict):
    """utils.py:169-181"""
    with open(path_to_homography_dict, "r") as curr_json:
        homography_dict = json.load(curr_json)
    if "resize_info" not in homography_dict:
        raise ValueError("Specify the height and width of the frame for which the homography matrix was obtained")
    resize_info = homography_dict.pop("resize_info")
    homography_dict = {int(k): v for k, v in homography_dict.items()}
    return homography_dict, resize_info


def superposition_dict(homography_dict):
    """{1: identity, frame_no: running superposition of the per-frame H} (utils.py:203-211): one device scan."""
    superposition_homography_dict = {1: [[1, 0, 0], [0, 1, 0], [0, 0, 1]]}
    frames = list(homography_dict.keys())
    # a None H keeps the running superposition unchanged (matrix_superposition returns H_sup for H None, utils.py:139):
    # such frames are left out of the device scan and take the previous frame's matrix afterwards
    live = [k for k in frames if homography_dict[k]["H"] is not None]
    sup = _small_context().superposition_scan([homography_dict[k]["H"] for k in live]) if live else []
    by_frame = dict(zip(live, sup))
    prev = None
    for k in frames:
        prev = by_frame.get(k, prev)
        superposition_homography_dict[k] = prev
    return superposition_homography_dict


def are_infinity_coordinates(coordinates_value):
    """utils.py:227-230"""
    try:
        return any(i >= INFINITY_COORDINATE for i in coordinates_value)
    except TypeError:
        return coordinates_value >= INFINITY_COORDINATE


def read_json_with_coordinates(path_to_coordinate):
    """utils.py:252-255"""
    with open(path_to_coordinate, 'r') as path_to_detection:
        coordinates = json.load(path_to_detection)
    return {int(k): v for k, v in coordinates.items()}


def get_largest_group_points(r_moving_dict, pts_a, pts_b):
    """Points of the most populated displacement group, first group on ties (utils.py:275-286)."""
    k_max_len, max_len = None, 0
    for key, value in r_moving_dict.items():
        if len(value) > max_len:
            max_len, k_max_len = len(value), key
    idx = r_moving_dict[k_max_len]
    return np.array([pts_a[i] for i in idx]), np.array([pts_b[i] for i in idx])


def find_point_displacement(matrix_H, pts_a, pts_b):
    """{rounded displacement: [point indices]} (utils.py:312-325); host glue kept for API compatibility -- the
    batched GPU path does the same grouping on device (evh_static_filter)."""
    if len(pts_a) != len(pts_b):
        raise ValueError("in find_static_part, len(pts_a) != len(pts_b)")
    r_moving_dict = {}
    for i, _ in enumerate(pts_a):
        tv = np.dot(matrix_H, (pts_a[i][0], pts_a[i][1], 1))
        r = round(np.sum(np.subtract(tv[:2] / tv[2], pts_b[i]) ** 2) ** 0.5)
        r_moving_dict.setdefault(r, []).append(i)
    return r_moving_dict


def compute_homography(pts_a, pts_b, matrix_H_prev=None):
    """H (float64 3x3) mapping pts_a onto pts_b, both optionally pre-transformed by matrix_H_prev; raises
    HomographyException like the reference (utils.py:351-362).  Runs on the GPU (evh_compute_homography)."""
    a = np.asarray(pts_a, np.float32).reshape(-1, 2)
    b = np.asarray(pts_b, np.float32).reshape(-1, 2)
    if len(a) != len(b):
        raise ValueError("compute_homography: len(pts_a) != len(pts_b)")
    ctx = runtime.get_context(64, 64, 2, max(len(a), runtime.NFEATURES))
    status, H = ctx.compute_homography(np.concatenate([a, b], axis=1), matrix_H_prev)
    if status != PAIR_OK:
        raise HomographyException("not enough points in homography calculation" if status == 4 else
                                  "can't calculate homography matrix")
    return H

"""Command-line driver with the argument surface and the non-visual outputs of the reference's example script
(evenvizion/examples/evenvizion_component.py:101-140), running the hot path on the MI355X.

    python -m evenvizion_amd.component --path_to_video frames.npy --experiment_name run1 --resize_width 400 \
           --path_to_original_coordinate original_coordinates.json

writes, like the reference, under  <cwd>/<experiment_name>/<video stem>/ :
    dict_with_homography_matrix.json   {frame_no: {"H": 3x3}, ..., "resize_info": {"h","w"}}   (:139-140)
    metrics_file.txt                   "Maximum movement during the entire video: <f64>"        (:62-65)
and, as a data file instead of the reference's rendered comparison video (:69-97),
    fixed_coordinates.json             from_original_to_fix(original coordinates) per frame.

--path_to_video takes what the reference's script takes for H.264 video in an MP4/MOV container: the file is opened by
evenvizion_amd.capture.VideoCapture (libevcap.so: this repository's own demultiplexer + H.264 decoder, standing in for
cv2.VideoCapture at evenvizion_component.py:132), e.g. the reference's own evenvizion/examples/test_video/test_video.mp4.
It also takes a .npy file (uint8 [F,h,w,3] BGR or [F,h,w] gray) or "synthetic:<frames>:<w>x<h>[:<seed>]".
Differences, all deliberate: the heat-map and matching PICTURES are
not rendered (--show_matching_visualization must stay off); --resize_width is honoured (the reference script parses
it but never passes it on, so it always runs at 400 -- the default here).
"""
import argparse
import json
import os

import numpy as np


def _bool(v):
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() not in ("0", "false", "no", "none", "")


VIDEO_SUFFIXES = (".mp4", ".mov", ".m4v")


def open_capture(spec):
    """-> (capture with .read() like cv2.VideoCapture, [h, w] of its frames, stem used for the output folder)"""
    from .synthetic import SyntheticCapture
    if spec.lower().endswith(VIDEO_SUFFIXES):
        from . import capture
        cap = capture.VideoCapture(spec)                                  # evenvizion_component.py:132
        if not cap.isOpened():
            raise ValueError("cannot open video %s: %s" % (spec, cap.open_error))
        return cap, [cap.height, cap.width], os.path.split(spec)[-1].split(".")[0]
    frames, stem = load_frames(spec)
    return SyntheticCapture(frames), [int(frames[0].shape[0]), int(frames[0].shape[1])], stem


def load_frames(spec):
    """-> (list of uint8 frames, stem used for the output folder)"""
    from . import synthetic as S
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        n = int(parts[1])
        w, h = (int(v) for v in parts[2].lower().split("x"))
        seed = int(parts[3]) if len(parts) > 3 else 1
        gray, _ = S.make_stream(seed, n, w, h)
        return list(S.gray_to_bgr(gray)), "synthetic_%d_%dx%d_%d" % (n, w, h, seed)
    arr = np.load(spec, allow_pickle=False)
    if arr.dtype != np.uint8 or arr.ndim not in (3, 4):
        raise ValueError("expected a uint8 array [F,h,w] or [F,h,w,3] in %s" % spec)
    return list(arr), os.path.split(spec)[-1].split(".")[0]


def main(argv=None):
    ap = argparse.ArgumentParser(description="EvenVizion hot path on MI355X (argument surface of evenvizion_component.py)")
    ap.add_argument("--path_to_video", type=str, default="synthetic:16:400x224:1")
    ap.add_argument("--experiment_name", type=str, default="test_video_processing")
    ap.add_argument("--resize_width", type=int, default=400, help="width to resize frames to")
    ap.add_argument("--path_to_original_coordinate", default=None, help="path to json with original coordinates")
    ap.add_argument("--none_H_processing", default=True, help="If True use H_prev as H, False - do nothing")
    ap.add_argument("--heatmap_visualization", default=True, help="write metrics_file.txt (pictures are not rendered)")
    ap.add_argument("--show_matching_visualization", default=False, help="not available: matching pictures are not rendered")
    ap.add_argument("--features", type=str, default="SURF,SIFT,ORB",
                    help="feature types in FrameProcessing order (extension; the reference hard-wires SURF,SIFT,ORB)")
    args = ap.parse_args(argv)
    if _bool(args.show_matching_visualization):
        raise NotImplementedError("matching pictures are outside the MI355X hot path; leave --show_matching_visualization off")

    from .processing.video_processing import get_homography_dict
    from .processing.utils import read_homography_dict, superposition_dict, read_json_with_coordinates, \
        are_infinity_coordinates
    from .processing.fixed_coordinate_system import from_original_to_fix
    from . import heatmap

    cap, original_shape, stem = open_capture(args.path_to_video)
    save_folder = os.path.join(os.getcwd(), args.experiment_name, stem)
    os.makedirs(save_folder, exist_ok=True)

    result = get_homography_dict(cap, resize_width=args.resize_width, matching_path=None,
                                 none_H_processing=_bool(args.none_H_processing),
                                 features_type_list=[f for f in args.features.split(",") if f])
    path_to_homography_dict = os.path.join(save_folder, "dict_with_homography_matrix.json")
    with open(path_to_homography_dict, "w") as json_:
        json.dump(result, json_)

    homography_matrices, resize_info = read_homography_dict(path_to_homography_dict)
    reformat = superposition_dict(homography_matrices)
    if _bool(args.heatmap_visualization):
        keys, per_frame = heatmap.frame_maxima(reformat, resize_info)
        max_movement = per_frame[:-1] if len(per_frame) > 1 else per_frame       # the reference never appends the last frame
        with open(os.path.join(save_folder, "metrics_file.txt"), "w") as txt_:
            txt_.write("Maximum movement during the entire video: {}".format(np.max(max_movement)))
            if are_infinity_coordinates(max_movement):
                txt_.write("There are some frames with undefined coordinates")
    if args.path_to_original_coordinate:
        original_coordinates = read_json_with_coordinates(args.path_to_original_coordinate)
        fixed = from_original_to_fix(original_coordinates, reformat, original_shape, [resize_info["h"], resize_info["w"]])
        with open(os.path.join(save_folder, "fixed_coordinates.json"), "w") as json_:
            json.dump(fixed, json_)
    return save_folder


if __name__ == "__main__":
    print(main())

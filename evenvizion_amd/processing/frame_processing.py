"""Per-frame feature extraction -- MI355X counterpart of evenvizion/processing/frame_processing.py.

FrameProcessing.detect_and_describe_features("ORB") (frame_processing.py:59-61, 70-71) runs the HIP ORB
(evh_orb_detect_batch), "SIFT" (frame_processing.py:62-64) the HIP SIFT (evh_sift_detect_batch): coordinates float32[N,2]
"SURF" (frame_processing.py:65-67, SURF_create(extended=1, hessianThreshold=400)) the HIP SURF (evh_surf_detect_batch):
coordinates float32[N,2] and descriptors uint8[N,32] / float32[N,128] exactly as cv2 hands them over.  The default
feature list is the reference's own, ["SURF", "SIFT", "ORB"] (frame_processing.py:40); the north-star hot path is
features_type_list=["ORB"].
"""
import numpy as np

from .. import runtime
from .matching import KeyPoints, NoMatchesException
from .utils import remove_double_matching


DEFAULT_FEATURES = ["SURF", "SIFT", "ORB"]     # frame_processing.py:40


class FrameProcessing:
    def __init__(self, frame, features_type_list=None):
        self.isv3 = True
        self.frame = frame
        self.features_types = features_type_list or list(DEFAULT_FEATURES)
        self._cache = {}

    def detect_and_describe_features(self, features_name):
        """-> (coordinates float32[N,2], descriptors uint8[N,32] or None when no key point was found)."""
        if features_name == "ORB":
            if "ORB" not in self._cache:
                frame = np.ascontiguousarray(self.frame, np.uint8)
                h, w = frame.shape[:2]
                ctx = runtime.get_context(w, h, 2, runtime.NFEATURES)
                ctx.orb_detect_batch(runtime.to_device(frame[None]), nfeatures=runtime.NFEATURES)
                f = ctx.orb_download(0)
                self._cache["ORB"] = (f["xy"], f["desc"] if len(f["xy"]) else None)
            return self._cache["ORB"]
        if features_name == "SIFT":
            if "SIFT" not in self._cache:
                frame = np.ascontiguousarray(self.frame, np.uint8)
                h, w = frame.shape[:2]
                ctx = runtime.get_context(w, h, 2, runtime.NFEATURES, sift=True)
                d_frame = runtime.to_device(frame[None])
                ctx.sift_detect_batch(d_frame)
                f = ctx.sift_download(0)               # synchronises: d_frame may go
                del d_frame
                self._cache["SIFT"] = (f["xy"], f["desc"] if len(f["xy"]) else None)
            return self._cache["SIFT"]
        if features_name == "SURF":
            if "SURF" not in self._cache:
                frame = np.ascontiguousarray(self.frame, np.uint8)
                h, w = frame.shape[:2]
                ctx = runtime.get_context(w, h, 2, runtime.NFEATURES, surf=True)
                d_frame = runtime.to_device(frame[None])
                ctx.surf_detect_batch(d_frame)
                f = ctx.surf_download(0)               # synchronises: d_frame may go
                del d_frame
                self._cache["SURF"] = (f["xy"], f["desc"] if len(f["xy"]) else None)
            return self._cache["SURF"]
        raise ValueError("You need to choose descriptors type")

    def concatenate_all_features_types(self, acceding_image):
        """Static matched points of self (current frame, a) and acceding_image (previous frame, b)
        (frame_processing.py:89-108)."""
        all_a, all_b = [], []
        for feature_type in self.features_types:
            coords_a, descriptors_a = self.detect_and_describe_features(feature_type)
            coords_b, descriptors_b = acceding_image.detect_and_describe_features(feature_type)
            static_a, static_b = KeyPoints(coords_a, descriptors_a).match_static_kps(KeyPoints(coords_b, descriptors_b))
            all_a.extend(static_a)
            all_b.extend(static_b)
        if all_a is None or all_b is None:
            raise NoMatchesException("can't find keypoints that lie on static objects ", "couldn't process")
        return remove_double_matching(all_a, all_b)

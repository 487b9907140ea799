"""Key-point matching -- MI355X counterpart of evenvizion/processing/matching.py (same names, arguments, errors).

KeyPoints.match_kps        : matching.py:75-129  -> evh_match_knn2_l2u8 (uint8 rows) / evh_match_knn2_l2f32 (float rows)
                                                    + evh_ratio_unique_filter[_f32] (HIP)
KeyPoints.match_static_kps : matching.py:131-163 -> + evh_find_homography_ransac + evh_static_filter (HIP)
lowes_ratio_test / filter_corresponding_points : matching.py:166-239, host glue on DMatch-like objects, kept for
API compatibility (the GPU path applies the same rules on device inside evh_ratio_unique_filter).
"""
import numpy as np

from .. import runtime
from .._lib import PAIR_FEW_MATCHES
from .constants import THRESHOLD_FOR_FIND_HOMOGRAPHY, LOWES_RATIO, MINIMUM_MATCHING_POINTS


class NoMatchesException(Exception):
    """Same constructor / string form as the reference (matching.py:22-44)."""

    def __init__(self, reason, description="no matches found"):
        self.reason = reason
        self.description = description
        super().__init__(description)

    def __str__(self):
        return f'{self.reason} -> {self.description}'


class KeyPoints:
    """coordinates: float32[N,2]; descriptors: uint8[N,32] (ORB), float32[N,128] (SIFT) or None (matching.py:47-73)."""

    def __init__(self, coordinates, descriptors):
        self.coordinates = coordinates
        self.descriptors = descriptors

    def _match_rows(self, acceding_kps, ratio, min_matching_pts):
        import torch
        if self.descriptors is None:
            raise NoMatchesException("self.descriptors is None", "couldn't process")
        if acceding_kps.descriptors is None:
            raise NoMatchesException("kps.descriptors is None", "couldn't process")
        float_desc = np.asarray(self.descriptors).dtype != np.uint8       # SIFT / SURF rows (float32), ORB rows are uint8
        if float_desc:
            q = np.ascontiguousarray(self.descriptors, np.float32)
            t = np.ascontiguousarray(acceding_kps.descriptors, np.float32)
            q = q.reshape(len(q), -1); t = t.reshape(len(t), -1)
        else:
            q = np.ascontiguousarray(self.descriptors, np.uint8).reshape(-1, 32)
            t = np.ascontiguousarray(acceding_kps.descriptors, np.uint8).reshape(-1, 32)
        xy_q = np.ascontiguousarray(self.coordinates, np.float32).reshape(-1, 2)
        xy_t = np.ascontiguousarray(acceding_kps.coordinates, np.float32).reshape(-1, 2)
        nq, nt = len(q), len(t)
        if nq == 0 or nt == 0:
            raise NoMatchesException("len(matches) 0 < min_matching_pts {}".format(min_matching_pts), "couldn't process")
        # the uint8 2-NN / filter work on a frame slot's rows (context capacity); the float forms (SIFT / SURF: up to 65 535
        # rows per frame) take any row count, so they do not size the context
        ctx = runtime.get_context(64, 64, 2, runtime.NFEATURES if float_desc else max(nq, nt, runtime.NFEATURES))
        dev = runtime.device()
        idx = torch.empty(nq, 2, dtype=torch.int32, device=dev)
        d2 = torch.empty(nq, 2, dtype=torch.float32 if float_desc else torch.int32, device=dev)
        pts = torch.empty(nq, 4, dtype=torch.float32, device=dev)
        # all four uploads live in named locals until ratio_unique_filter (which synchronises the context's stream)
        # has returned: a temporary freed right after an asynchronous launch could be handed to the next upload by
        # torch's caching allocator while the kernel is still reading it
        d_q, d_t, d_xy_q, d_xy_t = (runtime.to_device(a) for a in (q, t, xy_q, xy_t))
        if float_desc:           # matching.py:102-108 on float32[N,64|128]: evh_match_knn2_l2f32
            ctx.knn2_f32(d_q, d_t, idx, d2)
            n, st = ctx.ratio_unique_filter_f32(idx, d2, d_xy_q, d_xy_t, pts, ratio=ratio, min_matches=min_matching_pts)
        else:
            ctx.knn2(d_q, d_t, idx, d2)
            n, st = ctx.ratio_unique_filter(idx, d2, d_xy_q, d_xy_t, pts, ratio=ratio, min_matches=min_matching_pts)
        ctx.order_torch_after()
        del d_q, d_t, d_xy_q, d_xy_t
        if st == PAIR_FEW_MATCHES:
            raise NoMatchesException("len(matches) < min_matching_pts {}".format(min_matching_pts), "couldn't process")
        return ctx, pts[:n]

    def match_kps(self, acceding_kps, ratio=LOWES_RATIO, min_matching_pts=MINIMUM_MATCHING_POINTS):
        """-> (pts_a, pts_b): lists of float32[2] like the reference returns (matching.py:117-129)."""
        _, rows = self._match_rows(acceding_kps, ratio, min_matching_pts)
        rows = rows.cpu().numpy()
        return [r[:2].copy() for r in rows], [r[2:].copy() for r in rows]

    def match_static_kps(self, acceding_kps, reproj_thresh=THRESHOLD_FOR_FIND_HOMOGRAPHY):
        """-> (static_pts_a, static_pts_b) float32[M,2] arrays (matching.py:152-163)."""
        import torch
        ctx, rows = self._match_rows(acceding_kps, LOWES_RATIO, MINIMUM_MATCHING_POINTS)
        slot = ctx.lib.evh_orb_capacity(ctx.h)
        if len(rows) > slot * ctx.max_frames:
            # evh_find_homography_ransac / evh_static_filter view the context's per-pair buffers as one problem of up to
            # (rows per frame slot) x (frame slots) rows: a pair with more matches (SIFT at 720p: ~15 000) gets a context with
            # enough slots -- the reference has no bound here (matching.py:152-163)
            ctx = runtime.get_context(64, 64, -(-len(rows) // slot) + 1, ctx.max_features)
        H, _, _ = ctx.find_homography(rows, thr=reproj_thresh)
        if H is None:
            raise NoMatchesException("can't find homography matrix", "couldn't process")
        out = torch.empty_like(rows)
        n = ctx.static_filter(H, rows, out)
        res = out[:n].cpu().numpy()
        return res[:, :2].copy(), res[:, 2:].copy()


def lowes_ratio_test(raw_matches, ratio=LOWES_RATIO):
    """Host glue with the reference's semantics (matching.py:186-198): raw_matches is a list of lists of
    DMatch-like objects (.distance, .trainIdx, .queryIdx); returns [(trainIdx, queryIdx), ...]."""
    train_idx_dict = {}
    query_idx_dict = {}
    for matches in raw_matches:
        if len(matches) == 2 and matches[0].distance < matches[1].distance * ratio:
            train_idx_dict.setdefault(matches[0].trainIdx, []).append(matches[0].queryIdx)
            query_idx_dict.setdefault(matches[0].queryIdx, []).append(matches[0].trainIdx)
    return filter_corresponding_points(train_idx_dict, query_idx_dict)


def filter_corresponding_points(train_idx_dict, query_idx_dict):
    """Drop every train index claimed by more than one query (matching.py:226-239)."""
    doomed = set(k for k, v in train_idx_dict.items() if len(v) > 1)
    for _, v in query_idx_dict.items():
        if len(v) > 1:
            doomed.update(v)
    for k in doomed:
        train_idx_dict.pop(k, None)
    return [(train_idx, query_idx[0]) for train_idx, query_idx in train_idx_dict.items()]

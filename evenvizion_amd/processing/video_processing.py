"""Stream driver -- MI355X counterpart of evenvizion/processing/video_processing.py:27-108.

get_homography_dict keeps the reference signature and result layout
    {frame_no: {"H": 3x3 list}, ..., "resize_info": {"h", "w"}}       (first key is 2)
but instead of one frame pair per Python iteration it reads the capture in chunks, uploads a chunk once, and runs
the whole per-pair body on the GPU (evh_stream_homography_batch_resized: imutils.resize fused into the ingest kernel): ORB on every frame once
(the reference recomputes each frame's features twice), matching / RANSAC #1 / static filter for all
pairs of the chunk in parallel, and the final RANSAC as the sequential scan the running superposition requires
(utils.py:351-358, video_processing.py:102-103).  Consecutive chunks overlap by one frame and carry
{H_sup, H_prev} on the device.
"""
import logging

import numpy as np

from .. import runtime
from .._lib import PAIR_OK, PAIR_CAPACITY, EvhError

CHUNK_FRAMES = 64   # frames uploaded per GPU call (pairs per call = CHUNK_FRAMES - 1)


def resized_shape(frame_shape, resize_width):
    """imutils.resize(width=): r = width / float(w); dim = (width, int(h * r)) (video_processing.py:62)."""
    h0, w0 = frame_shape[:2]
    r = resize_width / float(w0)
    return int(resize_width), int(h0 * r)


def get_homography_dict(capture, resize_width=400, matching_path=None, none_H_processing=True,
                        nfeatures=runtime.NFEATURES, chunk_frames=CHUNK_FRAMES):
    """capture: anything with read() -> (bool, BGR uint8 frame) (cv2.VideoCapture duck type)."""
    import torch
    if matching_path:
        raise NotImplementedError("matching visualisation (draw_matches + imwrite) is outside the MI355X hot path; "
                                  "call with matching_path=None")
    success, first = capture.read()
    if not success:
        raise ValueError("Problem with video! Can't read first frame")
    first = np.ascontiguousarray(first, np.uint8)
    h0, w0 = first.shape[:2]
    cn = 1 if first.ndim == 2 else first.shape[2]
    dw, dh = resized_shape(first.shape, resize_width)
    if dw > w0 or dh > h0:
        raise NotImplementedError("resize_width larger than the frame (INTER_AREA enlargement) is outside the hot path")
    chunk_frames = max(2, int(chunk_frames))
    # sized for the RESIZED frames: only those go through ORB (evh_resize_area_u8 does not depend on the context's
    # geometry), so a 4K source with resize_width=400 allocates 400-wide buffers
    ctx = runtime.get_context(dw, dh, chunk_frames, nfeatures)
    dev = runtime.device()
    H_dev = torch.empty(chunk_frames - 1, 9, dtype=torch.float64, device=dev)
    st_dev = torch.empty(chunk_frames - 1, dtype=torch.int32, device=dev)
    state = torch.zeros(18, dtype=torch.float64, device=dev)

    homography_dict = {}
    pending = [first]
    frame_no = 1            # 1-based index of the newest frame already paired
    have_state = False
    exhausted = False
    while not exhausted:
        while len(pending) < chunk_frames:
            ok, frame = capture.read()
            if not ok:
                exhausted = True
                break
            pending.append(np.ascontiguousarray(frame, np.uint8))
        n = len(pending)
        if n < 2:
            break
        big = runtime.to_device(np.stack(pending))
        # K0 fused into the ingest kernel: level 0 comes straight from the full-size frames (N2); equal sizes are the
        # plain gray conversion
        ctx.stream_homography_batch(big, H_dev, st_dev, state_in=state if have_state else None, state_out=state,
                                    nfeatures=nfeatures, resize_to=(dw, dh))
        ctx.synchronize()
        Hs = H_dev[:n - 1].cpu().numpy().reshape(-1, 3, 3)
        sts = st_dev[:n - 1].cpu().numpy()
        for k in range(n - 1):
            frame_no += 1
            if sts[k] == PAIR_CAPACITY:
                # not a "no homography" outcome of the reference: a frame delivered more tied key points than a frame
                # slot holds (see include/evhip.h, EVH_PAIR_CAPACITY) -- repeating H_prev would hide a wrong result
                raise EvhError("frame %d or %d holds more key points (ties at the retainBest cut) than a frame slot "
                               "of this context; raise nfeatures capacity" % (frame_no - 1, frame_no))
            if sts[k] != PAIR_OK:
                logging.info("pair ending at frame %d: no homography (status %d)", frame_no, int(sts[k]))
                if not none_H_processing or not np.all(np.isfinite(Hs[k])):
                    # reference behaviour (video_processing.py:94-101): H stays None and None.tolist() raises --
                    # always for none_H_processing=False, and for a failing FIRST pair otherwise
                    raise AttributeError("'NoneType' object has no attribute 'tolist' (no homography for frame %d, "
                                         "status %d)" % (frame_no, int(sts[k])))
            homography_dict[frame_no] = {"H": Hs[k].tolist()}
        have_state = True
        pending = [pending[-1]]
    homography_dict["resize_info"] = {"h": dh, "w": dw}
    return homography_dict

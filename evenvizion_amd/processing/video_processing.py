"""Stream driver -- MI355X counterpart of evenvizion/processing/video_processing.py:27-108.

get_homography_dict keeps the reference signature and result layout
    {frame_no: {"H": 3x3 list}, ..., "resize_info": {"h", "w"}}       (first key is 2)
but instead of one frame pair per Python iteration it reads the capture in chunks (double-buffered: reading and
uploading chunk i+1 overlap the GPU work on chunk i), uploads a chunk once, and runs the whole per-pair body on the GPU (evh_stream_homography_batch_resized: imutils.resize fused into the ingest kernel): ORB on every frame once
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
                        nfeatures=runtime.NFEATURES, chunk_frames=CHUNK_FRAMES, features_type_list=None):
    """capture: anything with read() -> (bool, BGR uint8 frame) (cv2.VideoCapture duck type).
    features_type_list: the list the reference hands to FrameProcessing (frame_processing.py:37-40), e.g. ["SIFT", "ORB"];
    None = frame_processing.DEFAULT_FEATURES = the reference's own default ["SURF", "SIFT", "ORB"]; the north-star hot path
    is features_type_list=["ORB"] (one fused ORB pipeline, evh_stream_homography_batch_resized)."""
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
    # one staging buffer (pinned host / device) is capped in bytes: 4K BGR frames give 21-frame chunks, not 64
    from .frame_processing import DEFAULT_FEATURES
    features = list(features_type_list or DEFAULT_FEATURES)
    for name in features:
        if name not in ("ORB", "SIFT", "SURF"):
            raise ValueError("You need to choose descriptors type")
    multi = features != ["ORB"]
    chunk_frames = runtime.chunk_frames_for(first.nbytes, max(2, int(chunk_frames)))
    # sized for the RESIZED frames: only those go through ORB (evh_resize_area_u8 does not depend on the context's
    # geometry), so a 4K source with resize_width=400 allocates 400-wide buffers
    ctx = runtime.get_context(dw, dh, chunk_frames, nfeatures, sift="SIFT" in features, surf="SURF" in features)
    dev = runtime.device()
    # Double-buffered chunk pipeline: while the GPU works on chunk i the host reads chunk i+1 from the capture into
    # pinned memory and its upload runs on a copy stream; results come back through pinned buffers.  Chunk i is
    # launched BEFORE the results of chunk i-1 are collected, so the device queue never drains.
    B = runtime.staging((chunk_frames,) + first.shape, dev)
    host, host_np, devbuf = B["host"], B["host_np"], B["dev"]
    H_dev, st_dev, H_host, st_host = B["H_dev"], B["st_dev"], B["H_host"], B["st_host"]
    copy_stream, up_done, all_done = B["copy_stream"], B["up_done"], B["all_done"]
    state = torch.zeros(18, dtype=torch.float64, device=dev)
    # {H_sup, H_prev} as they ENTER each in-flight chunk (and whether there was a state at all): what a chunk is re-run
    # from when one of its frames overflows a frame slot (EVH_PAIR_CAPACITY)
    state_pre = [torch.zeros(18, dtype=torch.float64, device=dev) for _ in range(2)]
    had_state = [False, False]
    cuda = dev.type == "cuda"          # (the host-loop unit test drives this function on the CPU with a scripted context)
    cur = torch.cuda.current_stream(dev) if cuda else None

    homography_dict = {}
    frame_no = [1]          # 1-based index of the newest frame already paired

    def launch(c, jb, nb, with_state):
        if multi:       # frame_processing.py:91-104 over the type list: evh_stream_homography_batch_types
            c.stream_homography_batch_types(devbuf[jb][:nb], H_dev[jb], st_dev[jb], features,
                                            state_in=state if with_state else None, state_out=state, nfeatures=nfeatures,
                                            resize_to=(dw, dh))
        else:
            c.stream_homography_batch(devbuf[jb][:nb], H_dev[jb], st_dev[jb], state_in=state if with_state else None,
                                      state_out=state, nfeatures=nfeatures, resize_to=(dw, dh))
        if cuda:
            c.order_torch_after()
        H_host[jb][:nb - 1].copy_(H_dev[jb][:nb - 1], non_blocking=True)
        st_host[jb][:nb - 1].copy_(st_dev[jb][:nb - 1], non_blocking=True)
        if cuda:
            all_done[jb].record(cur)

    def rerun_with_larger_slots(jb, nb, later):
        """A frame of chunk jb delivered more tied key points than a frame slot of `ctx` holds.  The reference has no
        such bound (it carries on with every tie, frame_processing.py:59-61), so the chunk is re-run from the state it
        was entered with on a context whose frame slots are twice as large (same nfeatures, so the same key points for
        every other frame), doubling again if needed up to the LDS limit of the matching filter; the chunk launched
        after it (computed from a state that is now stale) is then re-run as well."""
        from .._lib import Context
        if cuda:
            torch.cuda.synchronize(dev)
        from .._lib import MAX_FEATURES
        # every list grows from what the context that overflowed actually had, each up to its own limit
        feats = max(ctx.max_features, nfeatures)
        sift0 = max(ctx.lib.evh_sift_capacity(ctx.h), runtime.sift_features_for(dw, dh)) if "SIFT" in features else 0
        surf0 = max(ctx.lib.evh_surf_capacity(ctx.h), runtime.surf_features_for(dw, dh)) if "SURF" in features else 0
        grow = 1
        while True:
            at_limit = feats >= MAX_FEATURES and (not sift0 or sift0 * grow >= runtime.TYPE_FEATURES_MAX) and \
                (not surf0 or surf0 * grow >= runtime.TYPE_FEATURES_MAX)
            feats = min(feats * 2, MAX_FEATURES)
            grow *= 2
            try:
                if at_limit:
                    raise EvhError("giving up")
                big = Context(device=runtime.device_index(), max_w=max(dw, 64), max_h=max(dh, 64),
                              max_features=feats, max_frames=chunk_frames)
                if sift0:
                    big.sift_enable(min(runtime.TYPE_FEATURES_MAX, sift0 * grow))
                if surf0:
                    big.surf_enable(min(runtime.TYPE_FEATURES_MAX, surf0 * grow))
            except EvhError:
                raise EvhError("frame %d..%d: more key points (ORB ties at the retainBest cut, or SIFT key points) than "
                               "the largest frame slot this device path supports" % (frame_no[0], frame_no[0] + nb - 1))
            try:
                state.copy_(state_pre[jb])
                launch(big, jb, nb, had_state[jb])
                if cuda:
                    torch.cuda.synchronize(dev)
                else:
                    big.synchronize()
            finally:
                big.close()
            if not (st_host[jb][:nb - 1].numpy() == PAIR_CAPACITY).any():
                break
        if later is not None:                           # the chunk that was in flight behind it
            lj, ln = later
            state_pre[lj].copy_(state)
            had_state[lj] = True
            launch(ctx, lj, ln, True)

    def collect(jb, nb, later=None):
        if cuda:
            all_done[jb].synchronize()
        if (st_host[jb][:nb - 1].numpy() == PAIR_CAPACITY).any():
            rerun_with_larger_slots(jb, nb, later)
        Hs = H_host[jb][:nb - 1].numpy().reshape(-1, 3, 3)
        sts = st_host[jb][:nb - 1].numpy()
        for k in range(nb - 1):
            frame_no[0] += 1
            fno = frame_no[0]
            if sts[k] != PAIR_OK:
                logging.info("pair ending at frame %d: no homography (status %d)", fno, int(sts[k]))
                if not none_H_processing or not np.all(np.isfinite(Hs[k])):
                    # reference behaviour (video_processing.py:94-101): H stays None and None.tolist() raises --
                    # always for none_H_processing=False, and for a failing FIRST pair otherwise
                    raise AttributeError("'NoneType' object has no attribute 'tolist' (no homography for frame %d, "
                                         "status %d)" % (fno, int(sts[k])))
            homography_dict[fno] = {"H": Hs[k].tolist()}

    j, n = 0, 1
    host_np[0][0] = first
    have_state = False
    exhausted = False
    inflight = None
    try:
        while True:
            while n < chunk_frames and not exhausted:
                ok, frame = capture.read()
                if not ok:
                    exhausted = True
                    break
                frame = np.asarray(frame, np.uint8)
                if frame.shape != first.shape:
                    raise ValueError("frame %d has shape %s, the first frame %s" % (frame_no[0] + n, frame.shape, first.shape))
                host_np[j][n] = frame                  # the one host copy: straight into pinned memory
                n += 1
            launched = None
            if n >= 2:
                if cuda:
                    with torch.cuda.stream(copy_stream):
                        devbuf[j][:n].copy_(host[j][:n], non_blocking=True)
                        up_done[j].record(copy_stream)
                    cur.wait_event(up_done[j])
                else:
                    devbuf[j][:n].copy_(host[j][:n])
                # K0 fused into the ingest kernel: level 0 comes straight from the full-size frames (N2); equal sizes
                # are the plain gray conversion
                state_pre[j].copy_(state)               # stream-ordered: after chunk i-1's kernels, before chunk i's
                had_state[j] = have_state
                launch(ctx, j, n, have_state)
                have_state = True
                launched = (j, n)
            if inflight is not None:
                collect(*inflight, later=launched)
            inflight = launched
            if inflight is None:
                break
            host_np[1 - j][0] = host_np[j][n - 1]      # consecutive chunks overlap by one frame
            j, n = 1 - j, 1
    finally:
        if cuda:
            torch.cuda.synchronize(dev)                 # nothing of this call is left in flight on the shared buffers
    homography_dict["resize_info"] = {"h": dh, "w": dw}
    return homography_dict

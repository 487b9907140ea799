"""Process-wide libevhip context cache for the Python mirror of evenvizion.processing.

One process drives one GPU (LOCAL_RANK selects it under torch.distributed.run).  Contexts own all device buffers;
they are re-created only when a call needs larger frames / more frame slots / more features than the cached one.
"""
import os

from . import _lib

_ctx = None
_cfg = None
NFEATURES = 500  # cv2.ORB_create() default used by the reference (frame_processing.py:60)
SIFT_FEATURES = 6144  # SIFT key points reserved per frame slot at the reference's default resize_width (SIFT_create()
                      # keeps every key point; a textured 400x224 frame has ~2 500)
TYPE_FEATURES_MAX = 65535  # per frame and feature type (row indices of the pair buffers)


def sift_features_for(w, h):
    """SIFT key points reserved per frame slot for w x h frames: one per 14 pixels (2.6 x what textured synthetic frames
    deliver), at least SIFT_FEATURES; a frame that delivers more is re-run on larger slots by get_homography_dict."""
    return int(min(TYPE_FEATURES_MAX, max(SIFT_FEATURES, -(-(int(w) * int(h)) // (14 * 1024)) * 1024)))


def surf_features_for(w, h):
    return int(min(TYPE_FEATURES_MAX, max(SURF_FEATURES, -(-(int(w) * int(h)) // (40 * 1024)) * 1024)))



def device_index():
    return int(os.environ.get("LOCAL_RANK", "0"))


SURF_FEATURES = 4096  # SURF key points reserved per frame slot (hessianThreshold 400: ~800 on a textured 400x224 frame)


def get_context(w, h, nframes=2, nfeatures=NFEATURES, sift=False, surf=False):
    """A context able to hold `nframes` frames of w x h with `nfeatures` keypoints each (sift=True: with the SIFT buffers,
    evh_sift_enable)."""
    global _ctx, _cfg
    import torch
    if not torch.cuda.is_available():
        raise _lib.EvhError("no MI355X visible: evenvizion_amd computes only on the GPU (libevhip.so); "
                            "there is no CPU fallback")
    need = (max(int(w), 64), max(int(h), 64), max(int(nframes), 2), int(nfeatures))
    if _ctx is None or _cfg[0] < need[0] or _cfg[1] < need[1] or _cfg[2] < need[2] or _cfg[3] < need[3]:
        cfg = need if _cfg is None else tuple(max(a, b) for a, b in zip(_cfg, need))
        if _ctx is not None:
            _ctx.close()
        _ctx = _lib.Context(device=device_index(), max_w=cfg[0], max_h=cfg[1], max_features=cfg[3], max_frames=cfg[2])
        _cfg = cfg
    if (sift and _ctx.lib.evh_sift_capacity(_ctx.h) <= 0) or (surf and _ctx.lib.evh_surf_capacity(_ctx.h) <= 0):
        if getattr(_ctx, "_multi_used", False):
            # the multi-type pair buffers are sized at their first use: a context that already ran a type list is
            # replaced when another detector joins
            cfg = _cfg
            sift = sift or _ctx.lib.evh_sift_capacity(_ctx.h) > 0
            surf = surf or _ctx.lib.evh_surf_capacity(_ctx.h) > 0
            _ctx.close()
            _ctx = _lib.Context(device=device_index(), max_w=cfg[0], max_h=cfg[1], max_features=cfg[3], max_frames=cfg[2])
        if sift and _ctx.lib.evh_sift_capacity(_ctx.h) <= 0:
            _ctx.sift_enable(sift_features_for(_cfg[0], _cfg[1]))
        if surf and _ctx.lib.evh_surf_capacity(_ctx.h) <= 0:
            _ctx.surf_enable(surf_features_for(_cfg[0], _cfg[1]))
    return _ctx


def device():
    import torch
    return torch.device("cuda", device_index())


def to_device(array):
    import numpy as np
    import torch
    return torch.from_numpy(np.ascontiguousarray(array)).to(device())


_staging = {}
STAGING_BYTES_PER_BUFFER = 512 << 20   # cap of one pinned / device chunk buffer of get_homography_dict


def chunk_frames_for(frame_bytes, wanted):
    """Frames per chunk so that one staging buffer stays within STAGING_BYTES_PER_BUFFER (at least 2 frames: 4K BGR =
    24.9 MB per frame -> 21 frames per chunk instead of 64)."""
    return max(2, min(int(wanted), STAGING_BYTES_PER_BUFFER // max(int(frame_bytes), 1)))


def _pin(t):
    try:
        return t.pin_memory()
    except RuntimeError:                   # small hosts / locked-memory limits: pageable staging still works, only slower
        return t


def staging(shape, dev):
    """Two pinned host chunks, two device chunks, result buffers, a copy stream and events for the chunk pipeline of
    get_homography_dict; cached per chunk shape (pinning hundreds of MB costs tens of milliseconds, a service processes many
    videos of one size), each buffer capped at STAGING_BYTES_PER_BUFFER; release_staging() / reset() drop them."""
    import torch
    key = (tuple(shape), str(dev))
    if key not in _staging:
        _staging.clear()                      # one shape at a time: a new video size replaces the old buffers
        npairs = shape[0] - 1
        cuda = torch.device(dev).type == "cuda"
        pin = _pin if cuda else (lambda t: t)
        host = [pin(torch.empty(shape, dtype=torch.uint8)) for _ in range(2)]
        _staging[key] = {
            "host": host, "host_np": [t.numpy() for t in host],
            "dev": [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(2)],
            "H_dev": [torch.empty(npairs, 9, dtype=torch.float64, device=dev) for _ in range(2)],
            "st_dev": [torch.empty(npairs, dtype=torch.int32, device=dev) for _ in range(2)],
            "H_host": [pin(torch.empty(npairs, 9, dtype=torch.float64)) for _ in range(2)],
            "st_host": [pin(torch.empty(npairs, dtype=torch.int32)) for _ in range(2)],
            "copy_stream": torch.cuda.Stream(device=dev) if cuda else None,
            "up_done": [torch.cuda.Event() for _ in range(2)] if cuda else None,
            "all_done": [torch.cuda.Event() for _ in range(2)] if cuda else None,
        }
    return _staging[key]


def release_staging():
    _staging.clear()


def reset():
    global _ctx, _cfg
    if _ctx is not None:
        _ctx.close()
    _ctx, _cfg = None, None
    _staging.clear()

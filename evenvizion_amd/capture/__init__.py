"""Capture source: the object the reference opens with cv2.VideoCapture(path)
(/root/reference/evenvizion/examples/evenvizion_component.py:132) and reads with capture.read()
(/root/reference/evenvizion/processing/video_processing.py:58,70), for H.264 video in MP4/MOV files.

The work is done by libevcap.so (C ABI in include/evcap.h): an ISO-BMFF demultiplexer and an H.264 decoder written
from ITU-T Rec. H.264, host C++ only.  This module is the thin ctypes layer and mirrors the few cv2.VideoCapture
methods the reference uses (read, isOpened, get, release).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("EVCAP_SO") or os.path.join(os.path.dirname(_HERE), "libevcap.so")   # override: sanitizer builds

BGR_SWSCALE_X86 = 0
BGR_SWSCALE_C = 1

# cv2.CAP_PROP_* values the reference's scripts ask for (processing_visualization.py)
CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5
CAP_PROP_FRAME_COUNT = 7

STAT_NAMES = ("macroblocks", "i4x4", "i8x8", "i16x16", "i_pcm", "p_skip", "b_skip", "b_direct_16x16", "inter", "transform_8x8",
              "bipred_blocks", "explicit_wp_blocks", "implicit_wp_blocks", "sub8x8_quadrants", "temporal_direct", "spatial_direct",
              "mmco_ops", "list_modifications", "long_term_pictures", "p_slices", "b_slices", "i_slices", "max_ref_idx")

_lib = None


class CaptureError(RuntimeError):
    pass


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    """Loads libevcap.so and binds every symbol include/evcap.h declares."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise CaptureError("libevcap.so is not built: run `make -C evenvizion_amd/capture` or __graft_entry__.build()")
        L = C.CDLL(_SO)
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.evcap_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.evcap_open_memory.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.evcap_close.argtypes = [vp]
        L.evcap_close.restype = None
        L.evcap_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double)]
        L.evcap_set_bgr_mode.argtypes = [vp, i32]
        L.evcap_set_honour_edit_list.argtypes = [vp, i32]
        L.evcap_read_bgr.argtypes = [vp, vp, i64]
        L.evcap_read_yuv420.argtypes = [vp, vp, i64, vp, vp, i64]
        L.evcap_last_frame_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.evcap_stats.argtypes = [vp, C.POINTER(i64), i32]
        L.evcap_last_error.argtypes = [vp]
        L.evcap_last_error.restype = C.c_char_p
        _lib = L
    return _lib


EXPORTS = ("evcap_open", "evcap_open_memory", "evcap_close", "evcap_info", "evcap_set_bgr_mode", "evcap_set_honour_edit_list",
           "evcap_read_bgr", "evcap_read_yuv420", "evcap_last_frame_info", "evcap_stats", "evcap_last_error")


class VideoCapture:
    """cv2.VideoCapture(path) for .mp4/.mov files with an H.264 track (evenvizion_component.py:132).

    Like cv2's object it never raises from the constructor: isOpened() says whether the file could be opened, read()
    returns (False, None) once the frames are exhausted.  A bitstream the decoder cannot follow raises CaptureError from
    read() instead of silently ending the video."""

    def __init__(self, path=None, data=None, bgr_mode=BGR_SWSCALE_X86, honour_edit_list=True):
        self._h = C.c_void_p()
        self._L = lib()
        self.open_error = ""
        if data is not None:
            buf = np.frombuffer(bytes(data), np.uint8)
            rc = self._L.evcap_open_memory(buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(self._h))
        else:
            rc = self._L.evcap_open(os.fsencode(path), C.byref(self._h))
        if rc != 0:
            self.open_error = (self._L.evcap_last_error(None) or b"").decode("utf-8", "replace")
            self._h = C.c_void_p()
            return
        w, h, n, fps = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        self._L.evcap_info(self._h, C.byref(w), C.byref(h), C.byref(n), C.byref(fps))
        self.width, self.height, self.sample_count, self.fps = w.value, h.value, n.value, fps.value
        self._L.evcap_set_bgr_mode(self._h, int(bgr_mode))
        self._L.evcap_set_honour_edit_list(self._h, 1 if honour_edit_list else 0)

    def isOpened(self):
        return bool(self._h)

    def _check(self, rc):
        if rc < 0:
            raise CaptureError("libevcap: %s" % (self._L.evcap_last_error(self._h) or b"").decode("utf-8", "replace"))
        return rc == 0

    def read(self):
        """-> (True, uint8[h,w,3] BGR) or (False, None) -- video_processing.py:58,70"""
        if not self._h:
            return False, None
        frame = np.empty((self.height, self.width, 3), np.uint8)
        ok = self._check(self._L.evcap_read_bgr(self._h, frame.ctypes.data_as(C.c_void_p), frame.strides[0]))
        return (True, frame) if ok else (False, None)

    def read_yuv420(self):
        """-> (True, (Y, Cb, Cr)) planes as decoded, or (False, None)"""
        if not self._h:
            return False, None
        cw, ch = (self.width + 1) // 2, (self.height + 1) // 2
        y = np.empty((self.height, self.width), np.uint8)
        cb = np.empty((ch, cw), np.uint8)
        cr = np.empty((ch, cw), np.uint8)
        ok = self._check(self._L.evcap_read_yuv420(self._h, y.ctypes.data_as(C.c_void_p), y.strides[0], cb.ctypes.data_as(C.c_void_p),
                                                    cr.ctypes.data_as(C.c_void_p), cb.strides[0]))
        return (True, (y, cb, cr)) if ok else (False, None)

    def last_frame_info(self):
        """-> dict(poc, decode_index, slice_type 'P'|'B'|'I') of the frame most recently returned"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        if self._L.evcap_last_frame_info(self._h, C.byref(a), C.byref(b), C.byref(c)) != 0:
            return None
        return {"poc": a.value, "decode_index": b.value, "slice_type": "PBI"[c.value]}

    def get(self, prop):
        if not self._h:
            return 0.0
        return {CAP_PROP_FRAME_WIDTH: float(self.width), CAP_PROP_FRAME_HEIGHT: float(self.height), CAP_PROP_FPS: float(self.fps),
                CAP_PROP_FRAME_COUNT: float(self.sample_count)}.get(int(prop), 0.0)

    def stats(self):
        """Coding tools exercised so far, by name (see include/evcap.h evcap_stats)."""
        buf = (C.c_int64 * 64)()
        n = self._L.evcap_stats(self._h, buf, 64)
        return dict(zip(STAT_NAMES, list(buf)[:n]))

    def release(self):
        if self._h:
            self._L.evcap_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def read_all(path, **kw):
    """All frames of a file as a list of BGR arrays (the test helper; a video is small next to 64 GiB only if it is short)."""
    cap = VideoCapture(path, **kw)
    if not cap.isOpened():
        raise CaptureError("cannot open %s: %s" % (path, cap.open_error))
    frames = []
    while True:
        ok, f = cap.read()
        if not ok:
            break
        frames.append(f)
    cap.release()
    return frames

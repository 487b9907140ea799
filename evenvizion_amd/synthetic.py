"""Synthetic frame pairs / streams with analytic ground-truth homographies.

There is no video decoder in the image (no cv2 / ffmpeg), so every workload -- tests, smoke and bench -- runs
on frames generated here: a piecewise-constant multi-scale block texture (dense FAST corners at every
octave), frame B = inverse-bilinear warp of the canvas by a known H_true plus Gaussian noise.
`SyntheticCapture` duck-types cv2.VideoCapture (`read() -> (bool, BGR uint8 frame)`), which is all that
get_homography_dict needs (reference: evenvizion/processing/video_processing.py:58,70).
"""
import numpy as np

MARGIN = 64


def make_canvas(rng, w, h):
    """(h+128) x (w+128) uint8 canvas: 128 + sum of nearest-upsampled U{-40..40} grids with cells 4,8,16,32."""
    H, W = h + 2 * MARGIN, w + 2 * MARGIN
    acc = np.full((H, W), 128, np.int32)
    for c in (4, 8, 16, 32):
        gh, gw = -(-H // c), -(-W // c)
        g = rng.integers(-40, 41, size=(gh, gw), dtype=np.int32)
        acc += np.repeat(np.repeat(g, c, axis=0), c, axis=1)[:H, :W]
    return np.clip(acc, 0, 255).astype(np.uint8)


def random_h(rng, w):
    """H_true = T(tx,ty) . R(theta) . S(s) . P(p1,p2) with the small inter-frame motion of SURVEY 8d."""
    tx, ty = rng.uniform(-8, 8, 2) * w / 400.0
    th = np.deg2rad(rng.uniform(-1, 1))
    s = rng.uniform(0.98, 1.02)
    p1, p2 = rng.uniform(-1e-5, 1e-5, 2) * 400.0 / w
    T = np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float64)
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float64)
    S = np.diag([s, s, 1.0])
    P = np.array([[1, 0, 0], [0, 1, 0], [p1, p2, 1]], np.float64)
    return T @ R @ S @ P


def warp_canvas(canvas, Hmap, w, h, rng=None, noise=2.0):
    """Frame whose pixel (x,y) shows canvas at Hmap.(x,y,1) (+MARGIN), bilinear, + N(0, noise^2)."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    d = Hmap[2, 0] * xs + Hmap[2, 1] * ys + Hmap[2, 2]
    u = (Hmap[0, 0] * xs + Hmap[0, 1] * ys + Hmap[0, 2]) / d + MARGIN
    v = (Hmap[1, 0] * xs + Hmap[1, 1] * ys + Hmap[1, 2]) / d + MARGIN
    u = np.clip(u, 0, canvas.shape[1] - 1.001)
    v = np.clip(v, 0, canvas.shape[0] - 1.001)
    x0 = np.floor(u).astype(np.int64); y0 = np.floor(v).astype(np.int64)
    fx = u - x0; fy = v - y0
    c = canvas.astype(np.float64)
    val = (c[y0, x0] * (1 - fx) * (1 - fy) + c[y0, x0 + 1] * fx * (1 - fy) + c[y0 + 1, x0] * (1 - fx) * fy +
           c[y0 + 1, x0 + 1] * fx * fy)
    if rng is not None and noise > 0:
        val = val + rng.normal(0.0, noise, val.shape)
    return np.clip(np.rint(val), 0, 255).astype(np.uint8)


def make_pair(seed, w, h, noise=2.0):
    """-> (prev gray u8[h,w], cur gray u8[h,w], H_true) with H_true mapping cur pixels onto prev pixels.

    Both frames are camera-like views of the canvas (bilinear sampling + sensor noise): prev at a small random
    pose G0, cur at G0.H_true, so cur(x) == prev(H_true x).  (SURVEY 8d takes prev as the raw central crop; a
    piecewise-constant axis-aligned crop has NO FAST-9 corner at level 0 -- every ring straddles several blocks --
    which no real frame resembles, so prev is resampled as well.)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    canvas = make_canvas(rng, w, h)
    th = np.deg2rad(rng.uniform(-2, 2))
    G0 = np.array([[np.cos(th), -np.sin(th), rng.uniform(-3, 3)], [np.sin(th), np.cos(th), rng.uniform(-3, 3)],
                   [0, 0, 1]], np.float64)
    Ht = random_h(rng, w)
    prev = warp_canvas(canvas, G0, w, h, rng, noise)
    cur = warp_canvas(canvas, G0 @ Ht, w, h, rng, noise)
    return prev, cur, Ht


def make_pair_batch(config, npairs, w, h, unique=None, noise=2.0):
    """frames u8[2B,h,w] laid out (prev0, cur0, prev1, cur1, ...), H_true f64[B,3,3].
    seed = 1000*config + pair_index.  `unique` < npairs tiles the first `unique` pairs."""
    unique = npairs if unique is None else min(unique, npairs)
    frames = np.empty((2 * npairs, h, w), np.uint8)
    Ht = np.empty((npairs, 3, 3), np.float64)
    for p in range(unique):
        a, b, H = make_pair(1000 * config + p, w, h, noise)
        frames[2 * p] = a; frames[2 * p + 1] = b; Ht[p] = H
    for p in range(unique, npairs):
        frames[2 * p] = frames[2 * (p % unique)]; frames[2 * p + 1] = frames[2 * (p % unique) + 1]
        Ht[p] = Ht[p % unique]
    return frames, Ht


def make_stream(seed, nframes, w, h, noise=2.0):
    """frames u8[F,h,w] + per-pair H_true[F-1] (frame k pixels -> frame k-1 pixels). Frame k shows the canvas
    at pose G_k = G_{k-1} . H_k, so consecutive frames overlap almost completely."""
    rng = np.random.Generator(np.random.PCG64(seed))
    canvas = make_canvas(rng, w, h)
    frames = np.empty((nframes, h, w), np.uint8)
    Hs = np.empty((max(nframes - 1, 0), 3, 3), np.float64)
    th = np.deg2rad(rng.uniform(-2, 2))
    G = np.array([[np.cos(th), -np.sin(th), 0.4], [np.sin(th), np.cos(th), 0.3], [0, 0, 1]], np.float64)
    frames[0] = warp_canvas(canvas, G, w, h, rng, noise)
    for k in range(1, nframes):
        Hk = random_h(rng, w)
        # keep the accumulated pose near the canvas centre: mean-reverting translation
        G = G @ Hk
        G = G / G[2, 2]
        drift = G[:2, 2].copy()
        lim = MARGIN * 0.6
        if np.any(np.abs(drift) > lim):
            back = np.eye(3); back[:2, 2] = -np.sign(drift) * np.minimum(np.abs(drift), 4.0 * w / 400.0)
            Hk = Hk @ back
            G = G @ back
        Hs[k - 1] = Hk
        frames[k] = warp_canvas(canvas, G, w, h, rng, noise)
    return frames, Hs


def make_pan_stream(seed, nframes, w, h, step=4.0, noise=2.0):
    """A long monotone pan: frame k views a (w + nframes*step + 2*MARGIN)-wide canvas at x-offset k*step (plus a small
    random jitter in rotation / y), so the running superposition H_sup translates by about nframes*step pixels --
    the long-stream regime of BASELINE configs[2] (utils.py:351-355: points are pre-transformed by H_sup, far from the
    origin, before the fp32 cast inside findHomography).  -> frames u8[F,h,w]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    W = int(w + nframes * step) + 2 * MARGIN + 8
    Hc = h + 2 * MARGIN
    acc = np.full((Hc, W), 128, np.int32)
    for c in (4, 8, 16, 32):
        gh, gw = -(-Hc // c), -(-W // c)
        g = rng.integers(-40, 41, size=(gh, gw), dtype=np.int32)
        acc += np.repeat(np.repeat(g, c, axis=0), c, axis=1)[:Hc, :W]
    canvas = np.clip(acc, 0, 255).astype(np.uint8)
    frames = np.empty((nframes, h, w), np.uint8)
    for k in range(nframes):
        th = np.deg2rad(rng.uniform(-0.4, 0.4))
        G = np.array([[np.cos(th), -np.sin(th), k * step + rng.uniform(-0.5, 0.5)],
                      [np.sin(th), np.cos(th), rng.uniform(-2, 2)], [0, 0, 1]], np.float64)
        frames[k] = warp_canvas(canvas, G, w, h, rng, noise)
    return frames


def gray_to_bgr(gray):
    return np.repeat(gray[..., None], 3, axis=-1)


class SyntheticCapture:
    """Duck-type of cv2.VideoCapture over an in-memory list of BGR frames."""

    def __init__(self, frames_bgr):
        self._frames = frames_bgr
        self._i = 0

    def read(self):
        if self._i >= len(self._frames):
            return False, None
        f = self._frames[self._i]
        self._i += 1
        return True, f

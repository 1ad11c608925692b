"""Deterministic synthetic frames for tests and bench (no dataset is available on either box).

SURVEY.md 8(d): KITTI-shaped (1241x376) or 1080p u8 images made of random axis-aligned rectangles
(sharp FAST corners) plus 3x3 box-smoothed noise.  A stereo right view shifts every rectangle left by
its own integer disparity in [2,64]; a "next" frame shifts the whole scene by (+3,+1) px per step.
Everything derives from splitmix64(seed, counter) so numpy versions cannot change the pixels.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser over a uint64 array."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream(seed, n, salt):
    with np.errstate(over="ignore"):
        base = (np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(salt) * np.uint64(0x9E3779B1)) & _M64
        idx = np.arange(n, dtype=np.uint64)
        return splitmix64(base + idx)


def _uniform_int(r, lo, hi):
    """Map uint64 randoms to integers in [lo, hi]."""
    return (lo + (r % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int64)


def make_frame(width, height, seed=20250215, step=0, right=False, n_rects=None, noise_amp=12):
    """Return an (height, width) uint8 image.

    step   -- frame index: the scene is translated by (+3*step, +1*step) px, fresh noise per step.
    right  -- stereo right view: each rectangle moves left by its disparity.
    """
    if n_rects is None:
        n_rects = max(200, (width * height) // 330)
    r = _stream(seed, n_rects * 6, 1).reshape(n_rects, 6)
    margin = 96
    x0 = _uniform_int(r[:, 0], -margin, width + margin)
    y0 = _uniform_int(r[:, 1], -margin, height + margin)
    rw = _uniform_int(r[:, 2], 5, 72)
    rh = _uniform_int(r[:, 3], 5, 56)
    amp = _uniform_int(r[:, 4], 14, 80)
    sign = np.where((r[:, 4] >> np.uint64(33)) & np.uint64(1), 1, -1)
    disp = _uniform_int(r[:, 5], 2, 64)
    acc = np.full((height, width), 128, dtype=np.int32)
    dx, dy = 3 * step, 1 * step
    for k in range(n_rects):
        xa = int(x0[k]) + dx - (int(disp[k]) if right else 0)
        ya = int(y0[k]) + dy
        xb, yb = xa + int(rw[k]), ya + int(rh[k])
        xa, ya = max(xa, 0), max(ya, 0)
        xb, yb = min(xb, width), min(yb, height)
        if xa < xb and ya < yb:
            acc[ya:yb, xa:xb] += int(sign[k]) * int(amp[k])
    if noise_amp > 0:
        salt = 1000 + 2 * step + (1 if right else 0)
        nz = _stream(seed, (width + 2) * (height + 2), salt).reshape(height + 2, width + 2)
        nz = (nz % np.uint64(2 * noise_amp + 1)).astype(np.int32) - noise_amp
        sm = np.zeros((height, width), dtype=np.int32)
        for oy in range(3):
            for ox in range(3):
                sm += nz[oy:oy + height, ox:ox + width]
        acc += sm // 3
    return np.clip(acc, 0, 255).astype(np.uint8)


def make_stereo_pair(width, height, seed=20250215, step=0, **kw):
    return (make_frame(width, height, seed, step, right=False, **kw),
            make_frame(width, height, seed, step, right=True, **kw))

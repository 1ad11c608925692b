"""Deterministic synthetic frames for tests and bench (no dataset is available on either box).

SURVEY.md 8(d): KITTI-shaped (1241x376) or 1080p u8 images made of random axis-aligned rectangles
(sharp FAST corners) over a box-smoothed noise texture that belongs to the scene, plus sensor noise.  A stereo right view shifts every image row left
by an integer disparity (8 bands, 4..60 px); a "next" frame shifts the whole scene by (+3,+1) px per step.
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


def _texture(seed, width, height, salt, amp, box):
    """Box-smoothed uniform noise field of the given size, values in about [-amp, amp]."""
    pad = box - 1
    nz = _stream(seed, (width + pad) * (height + pad), salt).reshape(height + pad, width + pad)
    nz = (nz % np.uint64(2 * amp * box + 1)).astype(np.int32) - amp * box
    sm = np.zeros((height, width), dtype=np.int32)
    for oy in range(box):
        for ox in range(box):
            sm += nz[oy:oy + height, ox:ox + width]
    return sm // (box * box // 2 + 1)


N_DISP_BANDS = 8


def row_disparity(height):
    """Integer disparity of every image row: 8 horizontal bands, 4..60 px, larger towards the bottom
    (a coarse ground plane).  The right view shows scene column x + d(y) at column x."""
    rows = np.arange(height)
    return 4 + 8 * (rows * N_DISP_BANDS // height)


def make_frame(width, height, seed=20250215, step=0, right=False, n_rects=None, noise_amp=3):
    """Return an (height, width) uint8 image.

    Scene = box-smoothed noise texture + additive random rectangles, rendered 64 px wider than the image.
    step : frame index; the scene is translated by (+3*step, +1*step) px, fresh sensor noise per frame.
    right: stereo right view = every row of the scene shifted left by row_disparity(row).
    """
    sw = width + 64
    if n_rects is None:
        n_rects = max(200, (sw * height) // 330)
    r = _stream(seed, n_rects * 5, 1).reshape(n_rects, 5)
    margin = 96
    x0 = _uniform_int(r[:, 0], -margin, sw + margin)
    y0 = _uniform_int(r[:, 1], -margin, height + margin)
    rw = _uniform_int(r[:, 2], 5, 72)
    rh = _uniform_int(r[:, 3], 5, 56)
    amp = _uniform_int(r[:, 4], 14, 80)
    sign = np.where((r[:, 4] >> np.uint64(33)) & np.uint64(1), 1, -1)
    dx, dy = 3 * step, 1 * step
    span = 3 * 64 + 16
    tex = _texture(seed, sw + span + 8, height + 80, 7, 9, 3)
    bx = span - dx % span
    by = 70 - dy % 64
    acc = 128 + tex[by:by + height, bx:bx + sw].astype(np.int32)
    for k in range(n_rects):
        xa = int(x0[k]) + dx
        ya = int(y0[k]) + dy
        xb, yb = xa + int(rw[k]), ya + int(rh[k])
        xa, ya = max(xa, 0), max(ya, 0)
        xb, yb = min(xb, sw), min(yb, height)
        if xa < xb and ya < yb:
            acc[ya:yb, xa:xb] += int(sign[k]) * int(amp[k])
    if right:
        cols = np.arange(width)[None, :] + row_disparity(height)[:, None]
        acc = np.take_along_axis(acc, cols, axis=1)
    else:
        acc = acc[:, :width]
    if noise_amp > 0:
        salt = 1000 + 2 * step + (1 if right else 0)
        nz = _stream(seed, width * height, salt).reshape(height, width)
        acc = acc + (nz % np.uint64(2 * noise_amp + 1)).astype(np.int32) - noise_amp
    return np.clip(acc, 0, 255).astype(np.uint8)


def make_stereo_pair(width, height, seed=20250215, step=0, **kw):
    return (make_frame(width, height, seed, step, right=False, **kw),
            make_frame(width, height, seed, step, right=True, **kw))


def make_vocabulary(k=10, L=4, seed=7, stop_fraction=0.02, weighting=0, norm=1):
    """A synthetic DBoW3-shaped vocabulary tree (the reference's ORBvoc blob is not in the tree): k children per
    inner node, L levels, nodes numbered breadth-first like DBoW3's m_nodes, children descriptors = the parent's with
    random bit flips, leaves get consecutive word ids in node order (Vocabulary::createWords) and an idf-like
    weight; a few leaves get weight 0 ("stopped" words).  Flat arrays, ready for vslam_voc_create / the oracle."""
    rng = np.random.default_rng(seed)
    n_nodes = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((n_nodes, 32), np.uint8)
    child_start = np.zeros(n_nodes, np.int32)
    child_count = np.zeros(n_nodes, np.int32)
    child_ids = np.zeros(n_nodes - 1, np.int32)
    word_id = np.zeros(n_nodes, np.int32)
    weight = np.zeros(n_nodes, np.float64)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    first, nxt, pos = 0, 1, 0
    for level in range(L):
        cnt = k ** level
        for node in range(first, first + cnt):
            child_start[node], child_count[node] = pos, k
            ids = np.arange(nxt, nxt + k, dtype=np.int32)
            child_ids[pos:pos + k] = ids
            flips = rng.integers(0, 256, (k, max(4, 48 >> level)))
            d = np.repeat(desc[node][None, :], k, 0)
            for c in range(k):
                for f in flips[c]:
                    d[c, f >> 3] ^= np.uint8(1 << (f & 7))
            desc[ids] = d
            nxt += k
            pos += k
        first += cnt
    leaves = np.arange(first, n_nodes)
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    w = rng.uniform(0.5, 9.0, len(leaves))
    w[rng.random(len(leaves)) < stop_fraction] = 0.0
    weight[leaves] = w
    return dict(k=k, L=L, weighting=weighting, norm=norm, child_start=child_start, child_count=child_count,
                child_ids=child_ids, desc=desc, weight=weight, word_id=word_id)

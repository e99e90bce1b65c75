"""Seeded synthetic grayscale frames for tests and benchmarks (no dataset, no decoder involved).

A frame is a mid-grey canvas covered with random filled rectangles and discs of random grey plus
band-limited noise, so that FAST finds far more candidates than the keypoint quota on every pyramid
level and both FAST thresholds get exercised.  A sequence is a window sliding over one larger
canvas by a constant (dx, dy) per frame with fresh noise per frame: consecutive frames then relate
by a known translation, which is what the projection-matching benchmark reprojects with.
"""
import numpy as np


def _canvas(width, height, rng, n_shapes):
    img = np.full((height, width), 128, dtype=np.int16)
    # a smooth illumination gradient so that flat regions are not perfectly flat
    yy, xx = np.mgrid[0:height, 0:width]
    img += ((xx * 13 // max(width, 1)) + (yy * 9 // max(height, 1))).astype(np.int16)
    xs = rng.integers(0, width, n_shapes)
    ys = rng.integers(0, height, n_shapes)
    ws = rng.integers(4, 48, n_shapes)
    hs = rng.integers(4, 48, n_shapes)
    gs = rng.integers(0, 256, n_shapes)
    kinds = rng.integers(0, 4, n_shapes)
    for x, y, w, h, g, k in zip(xs, ys, ws, hs, gs, kinds):
        x1, y1 = min(width, x + w), min(height, y + h)
        if k == 0:
            r = min(w, h) // 2 + 1
            cy, cx = y + r, x + r
            y0, y1d = max(0, cy - r), min(height, cy + r + 1)
            x0, x1d = max(0, cx - r), min(width, cx + r + 1)
            sub = img[y0:y1d, x0:x1d]
            dy, dx = np.ogrid[y0 - cy:y1d - cy, x0 - cx:x1d - cx]
            sub[dx * dx + dy * dy <= r * r] = g
        else:
            img[y:y1, x:x1] = g
    return img


def make_sequence(width, height, n_frames, seed=0xC0FFEE, dx=3, dy=1, n_shapes=None, noise_sigma=8.0):
    """Returns uint8 array [n_frames, height, width]; frame t+1 is frame t's content moved by
    (-dx, -dy) pixels, i.e. a point at (x, y) in frame t appears at (x - dx, y - dy) in frame t+1."""
    rng = np.random.default_rng(seed)
    if n_shapes is None:
        n_shapes = max(64, (width * height) // 160)
    cw, ch = width + dx * (n_frames - 1), height + dy * (n_frames - 1)
    canvas = _canvas(cw, ch, rng, int(n_shapes * cw * ch / (width * height)))
    frames = np.empty((n_frames, height, width), dtype=np.uint8)
    for t in range(n_frames):
        win = canvas[t * dy:t * dy + height, t * dx:t * dx + width].astype(np.float32)
        if noise_sigma > 0:
            noise = rng.normal(0.0, noise_sigma, (height // 2 + 1, width // 2 + 1)).astype(np.float32)
            noise = np.repeat(np.repeat(noise, 2, axis=0), 2, axis=1)[:height, :width]
            win = win + noise + rng.normal(0.0, noise_sigma * 0.25, (height, width)).astype(np.float32)
        frames[t] = np.clip(np.rint(win), 0, 255).astype(np.uint8)
    return frames


def make_frame(width, height, seed=0xC0FFEE, **kw):
    return make_sequence(width, height, 1, seed=seed, **kw)[0]


def make_stereo_pair(width, height, seed=0xC0FFEE, min_disp=2, max_disp=40, n_bands=8, noise_sigma=4.0):
    """Left/right rectified pair: the right image is the left one shifted left by a disparity that
    is constant per horizontal band (min_disp..max_disp), plus independent noise."""
    rng = np.random.default_rng(seed)
    pad = max_disp + 8
    canvas = _canvas(width + pad, height, rng, max(64, ((width + pad) * height) // 160))
    left = canvas[:, :width].astype(np.float32)
    right = np.empty_like(left)
    band_h = (height + n_bands - 1) // n_bands
    disps = np.linspace(max_disp, min_disp, n_bands).round().astype(int)
    for b in range(n_bands):
        y0, y1 = b * band_h, min(height, (b + 1) * band_h)
        d = int(disps[b])
        # x_right = x_left - d  <=>  right[:, x] = scene[:, x + d]
        right[y0:y1] = canvas[y0:y1, d:d + width]
    out = []
    for im in (left, right):
        im = im + rng.normal(0.0, noise_sigma, im.shape).astype(np.float32)
        out.append(np.clip(np.rint(im), 0, 255).astype(np.uint8))
    return out[0], out[1], disps, band_h

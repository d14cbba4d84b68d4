"""Synthetic NV12 sources for tests and bench.

S2 "ME stress": textured background panning (+3,-2) px/frame with 12 opaque moving
rectangles; S3 "worst case": i.i.d. uniform noise.  Pure numpy, deterministic.
The reference's own test pipelines use `videotestsrc`
(/root/reference/bindings/typescript/src/pipeline/generic-builder.ts:94); S1 therefore
comes from GStreamer itself, not from this module.
"""
import numpy as np

S2_SEED = 0x5EED
S3_SEED = 0xBAD5EED


def _hash2(x, y, seed):
    """Per-pixel 32-bit hash of integer coordinates (vectorised xorshift-multiply mix)."""
    h = (x.astype(np.uint64) * np.uint64(0x9E3779B1) + y.astype(np.uint64) * np.uint64(0x85EBCA77) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h


def _texture(xs, ys, seed):
    """8-bit texture at world coordinates: clip(128 + 48 sin(x/17) cos(y/23) + n, 16, 235), n in [-12,12]."""
    X, Y = np.meshgrid(xs, ys)
    n = (_hash2(X & 0xFFFFF, Y & 0xFFFFF, seed) % np.uint64(25)).astype(np.int32) - 12
    base = 128.0 + 48.0 * np.sin(X / 17.0) * np.cos(Y / 23.0)
    return np.clip(np.rint(base).astype(np.int32) + n, 16, 235).astype(np.uint8)


class _XorShift32:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF or 1

    def next(self):
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def rng(self, lo, hi):
        return lo + self.next() % (hi - lo + 1)


def s2_frames(width, height, count, seed=S2_SEED, start=0):
    """Yield (y, uv) NV12 planes (uint8; y: HxW, uv: H/2 x W interleaved CbCr)."""
    r = _XorShift32(seed)
    rects = []
    for _ in range(12):
        w, h = r.rng(64, 256), r.rng(64, 256)
        rects.append(dict(w=min(w, width // 2), h=min(h, height // 2), x=r.rng(0, max(1, width - 1)), y=r.rng(0, max(1, height - 1)),
                          vx=r.rng(-9, 9), vy=r.rng(-9, 9), seed=r.next()))
    for f in range(start, start + count):
        ox, oy = -3 * f + (1 << 18), 2 * f + (1 << 18)  # content moves by (+3,-2) px per frame
        xs = np.arange(width, dtype=np.int64) + ox
        ys = np.arange(height, dtype=np.int64) + oy
        y = _texture(xs, ys, seed)
        cx = np.arange(0, width, 2, dtype=np.int64) + ox
        cy = np.arange(0, height, 2, dtype=np.int64) + oy
        t = _texture(cx, cy, seed ^ 0xC0FFEE).astype(np.int32)
        cb = np.clip(128 + (t - 128) // 4, 16, 240).astype(np.uint8)
        cr = np.clip(128 - (t - 128) // 5, 16, 240).astype(np.uint8)
        for q in rects:
            x0 = (q["x"] + q["vx"] * f) % width
            y0 = (q["y"] + q["vy"] * f) % height
            x1, y1 = min(width, x0 + q["w"]), min(height, y0 + q["h"])
            rx = np.arange(x1 - x0, dtype=np.int64)
            ry = np.arange(y1 - y0, dtype=np.int64)
            y[y0:y1, x0:x1] = _texture(rx, ry, q["seed"])
            cx0, cy0, cx1, cy1 = x0 // 2, y0 // 2, x1 // 2, y1 // 2
            cb[cy0:cy1, cx0:cx1] = 96 + (q["seed"] & 63)
            cr[cy0:cy1, cx0:cx1] = 96 + ((q["seed"] >> 8) & 63)
        uv = np.empty((height // 2, width), np.uint8)
        uv[:, 0::2] = cb
        uv[:, 1::2] = cr
        yield y, uv


def s1_frames(width, height, count):
    """S1 "reference-faithful": what `videotestsrc` (default pattern, SMPTE bars) feeds the reference's test pipelines
    (/root/reference/bindings/typescript/src/pipeline/generic-builder.ts:94) -- a still picture of colour bars, restated here
    (75 % bars over two thirds of the height, a reverse strip, a luminance ramp strip with a noise patch at the right as in
    the GStreamer pattern; the noise patch is the only thing that changes between pictures).  Trivial for motion search:
    reported, never the headline."""
    bars = [(180, 128, 128), (162, 44, 142), (131, 156, 44), (112, 72, 58), (84, 184, 198), (65, 100, 212), (35, 212, 114)]  # Y Cb Cr, 75 % white yellow cyan green magenta red blue
    y = np.empty((height, width), np.uint8)
    cb = np.empty((height // 2, width // 2), np.uint8)
    cr = np.empty((height // 2, width // 2), np.uint8)
    h1, h2 = (2 * height // 3) & ~1, (3 * height // 4) & ~1
    for i, (yy, u, v) in enumerate(bars):
        x0, x1 = (i * width // 7) & ~1, ((i + 1) * width // 7) & ~1 if i < 6 else width
        y[:h1, x0:x1] = yy; cb[:h1 // 2, x0 // 2:x1 // 2] = u; cr[:h1 // 2, x0 // 2:x1 // 2] = v
        ry, ru, rv = bars[6 - i] if i % 2 == 0 else (16, 128, 128)
        y[h1:h2, x0:x1] = ry; cb[h1 // 2:h2 // 2, x0 // 2:x1 // 2] = ru; cr[h1 // 2:h2 // 2, x0 // 2:x1 // 2] = rv
    ramp = np.linspace(16, 235, width).astype(np.uint8)
    y[h2:] = ramp[None, :]; cb[h2 // 2:] = 128; cr[h2 // 2:] = 128
    g = np.random.Generator(np.random.PCG64(0x51))
    nx = (5 * width // 6) & ~1
    for _ in range(count):
        yf = y.copy()
        yf[h2:, nx:] = g.integers(16, 236, (height - h2, width - nx), dtype=np.uint8)
        uv = np.empty((height // 2, width), np.uint8)
        uv[:, 0::2] = cb; uv[:, 1::2] = cr
        yield yf, uv


def s4_frames(width, height, count, seed=0x11FE, pan_after=None):
    """S4 "live camera": a still textured scene with fresh sensor noise (+-2) on every picture, a few moving objects, and
    (optionally, from picture `pan_after` on) a slow pan of 1 px/frame -- what a contribution encoder sees most of the time,
    and where P_Skip and the rate-control floor matter.  Same planes as s2_frames."""
    r = _XorShift32(seed)
    objs = [dict(w=r.rng(32, max(33, width // 6)), h=r.rng(32, max(33, height // 5)), x=r.rng(0, width - 1), y=r.rng(0, height - 1),
                 vx=r.rng(-5, 5), vy=r.rng(-3, 3), seed=r.next()) for _ in range(5)]
    g = np.random.Generator(np.random.PCG64(seed))
    for f in range(count):
        pan = max(0, f - pan_after) if pan_after is not None else 0
        xs = np.arange(width, dtype=np.int64) + (1 << 18) + pan
        ys = np.arange(height, dtype=np.int64) + (1 << 18)
        X, Y = np.meshgrid(xs, ys)
        base = 120.0 + 40.0 * np.sin(X / 29.0) * np.cos(Y / 37.0) + 25.0 * np.sin((X + 2 * Y) / 11.0)
        fine = (_hash2(X & 0xFFFFF, Y & 0xFFFFF, seed) % np.uint64(9)).astype(np.int32) - 4
        y = np.rint(base).astype(np.int32) + fine
        cb = np.full((height // 2, width // 2), 118, np.int32) + (np.rint(base[::2, ::2]).astype(np.int32) - 120) // 6
        cr = np.full((height // 2, width // 2), 136, np.int32) - (np.rint(base[::2, ::2]).astype(np.int32) - 120) // 8
        for q in objs:
            x0, y0 = (q["x"] + q["vx"] * f) % width, (q["y"] + q["vy"] * f) % height
            x1, y1 = min(width, x0 + q["w"]), min(height, y0 + q["h"])
            rx, ry = np.arange(x1 - x0, dtype=np.int64), np.arange(y1 - y0, dtype=np.int64)
            y[y0:y1, x0:x1] = _texture(rx, ry, q["seed"])
            cb[y0 // 2:y1 // 2, x0 // 2:x1 // 2] = 100 + (q["seed"] & 63)
            cr[y0 // 2:y1 // 2, x0 // 2:x1 // 2] = 100 + ((q["seed"] >> 8) & 63)
        y = np.clip(y + g.integers(-2, 3, y.shape), 16, 235).astype(np.uint8)
        uv = np.empty((height // 2, width), np.uint8)
        uv[:, 0::2] = np.clip(cb + g.integers(-1, 2, cb.shape), 16, 240)
        uv[:, 1::2] = np.clip(cr + g.integers(-1, 2, cr.shape), 16, 240)
        yield y, uv


def s3_frames(width, height, count, seed=S3_SEED):
    g = np.random.Generator(np.random.PCG64(seed))
    for _ in range(count):
        yield (g.integers(0, 256, (height, width), dtype=np.uint8), g.integers(0, 256, (height // 2, width), dtype=np.uint8))


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)

"""Seeded synthetic greyscale videos shaped like the robot-camera footage:
a smooth static scene, sensor noise, transient hot pixels (some grown into small clusters),
stuck pixels, one saturated bright region and, optionally, camera motion and specular blobs.

``chunk_numpy`` (small, CPU) serves the tests; ``chunk_torch`` generates full-size chunks
directly in HBM for bench.py.  Both are deterministic for a given seed (each in its own RNG).
"""
import numpy as np


def chunk_numpy(seed, n, h, w, hot=6.0, stuck=3, bright_block=True, motion_at=(), shift=6.0, blobs=0):
    """(n,h,w) uint8.  ``motion_at``: frame indices at which the scene jumps by ``shift`` pixels."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = np.empty((n, h, w), np.uint8)
    stuck_px = [(int(rng.integers(h)), int(rng.integers(w))) for _ in range(stuck)]
    blob_px = [(int(rng.integers(4, max(5, h - 4))), int(rng.integers(4, max(5, w - 4))),
                int(rng.integers(2, 5)), int(rng.integers(1, 4))) for _ in range(blobs)]
    off = 0.0
    for i in range(n):
        if i in motion_at:
            off += shift
        x = xx - off
        base = 70 + 30 * np.sin(x / 5.0) * np.cos(yy / 4.0) + 20 * np.sin((x + yy) / 7.0)
        img = base + rng.normal(0, 2.0, (h, w))
        if bright_block:
            img[h // 2:h // 2 + 9, w // 3:w // 3 + 11] = 200
        for (by, bx, ra, rb) in blob_px:                       # specular blobs move with the scene
            cx = bx + off
            m = ((xx - cx) / ra) ** 2 + ((yy - by) / rb) ** 2 <= 1.0
            img[m] = 230
        for _ in range(rng.poisson(hot)):
            y, x0 = int(rng.integers(h)), int(rng.integers(w))
            v = int(rng.integers(150, 256))
            img[y, x0] = v
            if rng.random() < 0.3:
                for _ in range(int(rng.integers(1, 12))):
                    y2 = min(h - 1, max(0, y + int(rng.integers(-2, 3))))
                    x2 = min(w - 1, max(0, x0 + int(rng.integers(-2, 3))))
                    img[y2, x2] = v
        for (y, x0) in stuck_px:
            img[y, x0] = 250
        frames[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return frames


def chunk_torch(seed, n, h, w, device, hot_per_mpix=490.0, stuck=20, speeds=None, segment=50, block=64):
    """(n,h,w) uint8 tensor on ``device``.

    Static scene (smooth field in [20,120]) + N(0, 2) noise; per frame Poisson(hot_per_mpix * h*w/1e6)
    transient hot pixels of value U[150,255], 10 % of them grown to 2-6 pixel clusters; ``stuck`` pixels
    lit in every frame; one 40x40 region at 200.  ``speeds`` (e.g. [0, 0.5, 4, 8]) translates the scene
    by that many pixels per frame in alternating ``segment``-frame runs (the motion-heavy config).
    """
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    yy = torch.arange(h, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(w, device=device, dtype=torch.float32)[None, :]
    out = torch.empty((n, h, w), dtype=torch.uint8, device=device)
    lam = hot_per_mpix * h * w / 1e6
    sy = torch.randint(0, h, (stuck,), generator=g, device=device)
    sx = torch.randint(0, w, (stuck,), generator=g, device=device)
    off = 0.0
    for i0 in range(0, n, block):
        nb = min(block, n - i0)
        offs = []
        for i in range(i0, i0 + nb):
            if speeds:
                off += float(speeds[(i // segment) % len(speeds)])
            offs.append(off)
        o = torch.tensor(offs, device=device, dtype=torch.float32)[:, None, None]
        x = xx[None] - o
        y = yy[None]
        img = 70 + 35 * torch.sin(x / 97.0) * torch.cos(y / 71.0) + 15 * torch.sin((x + y) / 23.0)
        if speeds:                                               # texture fine enough for the flow to lock on to
            for kx, ky, ph in ((0.61, 0.23, 0.3), (0.17, 0.55, 1.1), (-0.41, 0.37, 2.0), (0.29, -0.47, 0.7), (0.83, 0.11, 2.9), (0.07, 0.31, 1.7)):
                img = img + 9 * torch.sin(x * kx + y * ky + ph)
        img = img + 2.0 * torch.randn((nb, h, w), generator=g, device=device)
        by, bx = h // 2, w // 3
        img[:, by:by + 40, bx:bx + 40] = 200.0
        k = int(lam * nb)
        if k > 0:
            f = torch.randint(0, nb, (k,), generator=g, device=device)
            py = torch.randint(0, h, (k,), generator=g, device=device)
            px = torch.randint(0, w, (k,), generator=g, device=device)
            v = torch.randint(150, 256, (k,), generator=g, device=device).to(torch.float32)
            img[f, py, px] = v
            grow = torch.rand((k,), generator=g, device=device) < 0.10
            for _ in range(5):                                   # up to 5 more pixels per grown cluster
                dy = torch.randint(-1, 2, (k,), generator=g, device=device)
                dx = torch.randint(-1, 2, (k,), generator=g, device=device)
                on = grow & (torch.rand((k,), generator=g, device=device) < 0.7)
                qy = (py + dy).clamp_(0, h - 1)[on]
                qx = (px + dx).clamp_(0, w - 1)[on]
                img[f[on], qy, qx] = v[on]
        img[:, sy, sx] = 250.0
        out[i0:i0 + nb] = img.round_().clamp_(0, 255).to(torch.uint8)
    return out

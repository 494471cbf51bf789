"""CPU stand-in for image_processing_amd.engine.Engine, built on the oracle (TEST USE ONLY).

It lets the multi-process tests exercise the HOST logic of image_processing_amd/sharding.py
(partitioning, halos, the persistence / static-map / result exchanges) over gloo on a machine
without a GPU.  Same method names, tensor shapes and dtypes as Engine; tensors live on the CPU.
"""
import types

import numpy as np
import torch

from oracle import stages as S


def pack(masks):
    """(count,H,W) bool -> (count,H,wpr) int64 bit-planes (bit b of word k = pixel 64k+b)."""
    m = np.asarray(masks, bool)
    c, h, w = m.shape
    wpr = (w + 63) // 64
    pad = np.zeros((c, h, wpr * 64), bool)
    pad[..., :w] = m
    by = np.packbits(pad, axis=-1, bitorder="little")
    return torch.from_numpy(by.view(np.uint64).astype(np.int64).reshape(c, h, wpr))


def unpack(bits, w):
    a = bits.numpy().astype(np.uint64)
    c, h, wpr = a.shape
    by = a.view(np.uint8).reshape(c, h, wpr * 8)
    return np.unpackbits(by, axis=-1, bitorder="little")[..., :w].astype(bool)


class OracleEngine:
    def __init__(self, params=None):
        self.st = S.settings(params)
        self.params = types.SimpleNamespace(min_valid_frames=10, **self.st)

    def empty_bits(self, count, H, W):
        return torch.zeros((count, H, (W + 63) // 64), dtype=torch.int64)

    def flow_metric(self, frames, prev_frame=None, want_keep=True):
        f = frames.numpy()
        flows = np.zeros(len(f), np.float32)
        for i in range(len(f)):
            p = f[i - 1] if i > 0 else (prev_frame.numpy() if prev_frame is not None else None)
            if p is not None:
                flows[i] = S.flow_between(p, f[i])
        keep = (flows.astype(np.float64) <= self.st["flow_threshold"]).astype(np.uint8)
        return torch.from_numpy(flows), torch.from_numpy(keep)

    def detect_raw(self, frames, keep=None, first=0, count=None, want_dense=True, raw_out=None):
        f = frames.numpy()
        n, H, W = f.shape
        count = n - first if count is None else count
        raw = np.zeros((count, H, W), bool)
        bright = np.zeros((count, H, W), bool)
        dense = np.zeros((count, 2), np.int64)
        for j in range(count):
            i = first + j
            if keep is not None and not keep[i]:
                continue
            bg = S.compute_background(f, i, self.st["sliding_window_radius"])
            raw[j] = S.raw_damaged_mask(f[i], bg)
            bright[j] = bg > self.st["brightness_threshold"]
            dense[j] = ((f[i] < 170).sum(), (f[i] > self.st["brightness_threshold"]).sum())
        rb = pack(raw)
        if raw_out is not None:
            raw_out.copy_(rb)
            rb = raw_out
        return rb, pack(bright), torch.from_numpy(dense)

    def remove_bright_regions(self, bright_bits, damaged_bits, H, W, out=None):
        b, d = unpack(bright_bits, W), unpack(damaged_bits, W)
        res = np.stack([S.remove_bright_regions(np.where(b[i], 255.0, 0.0).astype(np.float32), 170, d[i],
                                                self.st["max_cluster_size"]) for i in range(len(d))])
        r = pack(res)
        if out is not None:
            out.copy_(r)
            return out
        return r

    def persistent_mask(self, raw_bits, keep, H, W):
        raw = unpack(raw_bits, W)
        masks = [raw[i] if keep is None or keep[i] else None for i in range(len(raw))]
        if masks[0] is None:
            masks[0] = np.zeros((H, W), bool)        # same runs: a None plane and a zero plane both reset
        return pack(S.compute_persistent_mask(masks, self.st["consecutive_threshold"])[None])[0]

    def or_planes(self, planes, H, W):
        out = planes[0].clone()
        for p in planes[1:]:
            out |= p
        return out

    def static_maps(self, frames, raw_bits, pers_bits, keep):
        f = frames.numpy()
        n, H, W = f.shape
        raw, pers = unpack(raw_bits, W), unpack(pers_bits[None], W)[0]
        heat = np.zeros((H, W), np.int32)
        valid = np.zeros((H, W), np.int32)
        for i in range(n):
            clean = raw[i] & ~pers if keep[i] else np.zeros((H, W), bool)
            heat += clean
            valid += ~((f[i] > self.st["brightness_threshold"]) & ~clean)
        return torch.from_numpy(np.stack([heat, valid]))

    def static_from_maps(self, maps):
        heat, valid = maps[0].numpy().astype(np.int64), maps[1].numpy().astype(np.int64)
        pct = np.zeros(heat.shape, float)
        good = valid > self.params.min_valid_frames
        pct[good] = (heat[good] / valid[good]) * 100
        return pack((pct > self.st["static_threshold"])[None])[0]

    def finalize(self, frames, raw_bits, pers_bits, static_bits, keep, want_bits=True):
        f = frames.numpy()
        n, H, W = f.shape
        raw = unpack(raw_bits, W)
        drop = unpack(pers_bits[None], W)[0] | unpack(static_bits[None], W)[0]
        fin = np.zeros((n, H, W), bool)
        counts = np.zeros((n, 3), np.int64)
        for i in range(n):
            if keep[i]:
                fin[i] = raw[i] & ~drop
                counts[i] = (fin[i].sum(), (fin[i] & (f[i] < 170)).sum(),
                             (fin[i] & (f[i] > self.st["brightness_threshold"])).sum())
        return pack(fin), torch.from_numpy(counts)

    def cluster_stats(self, frames, mask_bits, keep=None, want_cleaned=False):
        f = frames.numpy()
        n, H, W = f.shape
        m = unpack(mask_bits, W)
        stats = np.zeros((n, 3), np.int64)
        for i in range(n):
            if keep is not None and not keep[i]:
                continue
            cm, cnt, _, _ = S.filter_damaged_pixel_clusters(f[i], m[i], self.st["min_cluster_size"],
                                                            self.st["max_cluster_size"], self.st["min_circularity"])
            if cnt:
                stats[i] = (cnt, cm.sum(), f[i][cm].astype(np.int64).sum())
        return torch.from_numpy(stats), None

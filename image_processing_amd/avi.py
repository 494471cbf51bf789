"""Uncompressed AVI (RIFF) reader / writer for 8-bit greyscale and 24-bit BGR footage.

The reference decodes through cv2.VideoCapture (ip.py:62-85); cv2 is optional here.  This module
is the cv2-free input side used by the benchmarks and tests: ``write_avi`` stores frames as raw DIBs
(8-bit palettised grey or 24-bit BGR, bottom-up rows padded to 4 bytes -- what FFmpeg's rawvideo
decoder and therefore cv2 read back byte-exactly), ``AviReader`` memory-maps such a file and
returns frames by index without copying the file.  Compressed streams (MJPG, ...) are refused:
decode them with cv2.
"""
import mmap
import os
import struct
import threading
import zlib

import numpy as np


def _chunk(fourcc, payload):
    pad = b"\0" if len(payload) & 1 else b""
    return fourcc + struct.pack("<I", len(payload)) + payload + pad


def _list(kind, payload):
    return _chunk(b"LIST", kind + payload)


def write_avi(filename, frames, fps=25.0):
    """frames: sequence of (H,W) uint8 (grey) or (H,W,3) uint8 (BGR) arrays, all of one shape.  Streams to
    the file frame by frame (sizes are known up front), so a chunk is never held twice in memory."""
    frames = frames if hasattr(frames, "__len__") else list(frames)
    n = len(frames)
    if not n:
        raise ValueError("no frames")
    first = np.asarray(frames[0])
    h, w = first.shape[:2]
    colour = first.ndim == 3
    bpp = 24 if colour else 8
    stride = (w * (3 if colour else 1) + 3) & ~3
    img_bytes = stride * h
    if n * (img_bytes + 24) + 8192 > 0xFFFFFFFF:
        raise ValueError("%d frames of %dx%d do not fit the 4 GB of a RIFF AVI; write several files" % (n, w, h))
    usec = int(round(1e6 / fps))
    rate, scale = int(round(fps * 1000)), 1000
    avih = struct.pack("<14I", usec, img_bytes * int(round(fps)), 0, 0x10, n, 0, 1, img_bytes, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, n, img_bytes, 0xFFFFFFFF, 0,
                                           0, 0, w, h)
    ncol = 0 if colour else 256
    bih = struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, 0, img_bytes, 0, 0, ncol, 0)
    pal = b"" if colour else b"".join(struct.pack("<4B", i, i, i, 0) for i in range(256))
    hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", bih + pal)))
    rec = 8 + img_bytes + (img_bytes & 1)                   # one '00db' chunk
    movi_len = 4 + n * rec
    idx_len = 16 * n
    riff_len = 4 + len(hdrl) + (8 + movi_len) + (8 + idx_len)
    row = np.zeros((h, stride), np.uint8)
    with open(filename, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", riff_len) + b"AVI " + hdrl)
        fh.write(b"LIST" + struct.pack("<I", movi_len) + b"movi")
        head = b"00db" + struct.pack("<I", img_bytes)
        for f in frames:
            f = np.asarray(f)
            if f.shape[:2] != (h, w) or (f.ndim == 3) != colour or f.dtype != np.uint8:
                raise ValueError("all frames must be uint8 arrays of one shape")
            row[:, :w * (3 if colour else 1)] = f.reshape(h, -1)
            fh.write(head)
            fh.write(row[::-1].tobytes())                   # DIB rows are stored bottom-up
            if img_bytes & 1:
                fh.write(b"\0")
        fh.write(b"idx1" + struct.pack("<I", idx_len))
        fh.write(b"".join(struct.pack("<4sIII", b"00db", 0x10, 4 + i * rec, img_bytes) for i in range(n)))
    return filename


_LAYOUTS = {}                  # (real path, size, mtime) -> parsed layout of the file (see AviReader.__init__)
_LAYOUT_LOCK = threading.Lock()


class AviReader:
    """Random access to the frames of an uncompressed AVI.  ``read(i)`` returns (H,W) uint8 for 8-bit
    streams and (H,W,3) BGR uint8 for 24-bit streams."""

    def __init__(self, filename):
        self.filename = filename
        self._mm = None
        self._fh = open(filename, "rb")
        try:
            size = os.fstat(self._fh.fileno()).st_size
            if size < 12:
                raise ValueError("%s: not an AVI file" % filename)
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            mm = self._mm
            if mm[0:4] != b"RIFF" or mm[8:12] != b"AVI ":
                raise ValueError("%s: not a RIFF AVI file" % filename)
            self.fps = 0.0
            self.width = self.height = self.bpp = 0
            self.palette = None                             # (256, 3) uint8 BGR for 8-bit streams
            self._frames = []                               # (offset, length)
            # The walk over the file's chunks is a Python loop over every frame (and touches a page of the file per frame):
            # a chunk driver that opens the file once per chunk would pay it again and again, holding the interpreter lock
            # against the thread that enqueues the GPU work.  The parsed layout of the last few files is kept.
            st = os.fstat(self._fh.fileno())
            # (size + modification time + checksums of the file's head, where the stream format and palette are, and tail,
            # where the index is: a rewritten file of the same length within the clock's granularity is still another file)
            key = (os.path.realpath(filename), st.st_size, st.st_mtime_ns, zlib.crc32(mm[:4096]), zlib.crc32(mm[max(0, size - 4096):]))
            with _LAYOUT_LOCK:
                known = _LAYOUTS.get(key)
            if known is not None:
                self.fps, self.width, self.height, self.bpp, self.palette, self._bottom_up, self._frames = known
            else:
                self._walk(12, min(size, 8 + struct.unpack("<I", mm[4:8])[0]))
                if not self.bpp:
                    raise ValueError("%s: no video stream format found" % filename)
                with _LAYOUT_LOCK:
                    if len(_LAYOUTS) >= 8:
                        _LAYOUTS.pop(next(iter(_LAYOUTS)))
                    _LAYOUTS[key] = (self.fps, self.width, self.height, self.bpp, self.palette, self._bottom_up, self._frames)
            self._stride = (self.width * self.bpp // 8 + 3) & ~3
        except Exception:                                   # nothing stays open behind a failed open
            self.close()
            raise

    def _walk(self, pos, end):
        mm = self._mm
        while pos + 8 <= end:
            cc, ln = mm[pos:pos + 4], struct.unpack("<I", mm[pos + 4:pos + 8])[0]
            body = pos + 8
            if cc == b"LIST":
                kind = mm[body:body + 4]
                if kind in (b"hdrl", b"strl", b"movi"):
                    self._walk(body + 4, min(end, body + ln))
            elif cc == b"avih":
                usec = struct.unpack("<I", mm[body:body + 4])[0]
                if usec and not self.fps:
                    self.fps = 1e6 / usec
            elif cc == b"strh" and mm[body:body + 4] == b"vids":
                scale, rate = struct.unpack("<II", mm[body + 20:body + 28])
                if scale:
                    self.fps = rate / scale
            elif cc == b"strf" and not self.bpp:
                _, w, h, _, bpp, comp = struct.unpack("<IiiHHI", mm[body:body + 20])
                if comp != 0 or bpp not in (8, 24):
                    raise ValueError("%s: only uncompressed 8/24-bit AVI is read here (compression %r, %d bpp); "
                                     "decode other streams with cv2" % (self.filename, mm[body + 16:body + 20], bpp))
                self.width, self.height, self.bpp = w, abs(h), bpp
                self._bottom_up = h > 0
                if bpp == 8:                                # colour table: biClrUsed (0 = 256) entries of B, G, R, 0
                    hdr = struct.unpack("<I", mm[body:body + 4])[0]
                    used = struct.unpack("<I", mm[body + 32:body + 36])[0] if hdr >= 36 else 0
                    count = min(used or 256, 256, max(0, (ln - hdr) // 4))
                    pal = np.zeros((256, 3), np.uint8)
                    if count:
                        raw = np.frombuffer(mm, dtype=np.uint8, count=4 * count, offset=body + hdr).reshape(count, 4)
                        pal[:count] = raw[:, :3]
                    else:                                   # no table stored: grey ramp, as decoders assume
                        pal[:] = np.arange(256, dtype=np.uint8)[:, None]
                    self.palette = pal
            elif cc[2:4] in (b"db", b"dc") and cc[0:2].isdigit():
                self._frames.append((body, ln))
            pos = body + ln + (ln & 1)

    @property
    def frame_count(self):
        return len(self._frames)

    @property
    def grey_palette(self):
        """True when an 8-bit stream's colour table is the identity grey ramp (entry i = (i, i, i)): only then are
        the stored indices the grey levels a decoder + BGR2GRAY would return."""
        return self.bpp == 8 and self.palette is not None and \
            bool((self.palette == np.arange(256, dtype=np.uint8)[:, None]).all())

    def read(self, i):
        off, ln = self._frames[i]
        h, w = self.height, self.width
        if ln < self._stride * h:
            raise ValueError("frame %d is truncated" % i)
        a = np.frombuffer(self._mm, dtype=np.uint8, count=self._stride * h, offset=off).reshape(h, self._stride)
        if self._bottom_up:
            a = a[::-1]
        if self.bpp == 8:
            return np.ascontiguousarray(a[:, :w])
        return np.ascontiguousarray(a[:, :3 * w].reshape(h, w, 3))

    def read_into(self, i, out):
        """Frame i of an 8-bit stream straight into ``out`` ((H,W) uint8, e.g. a row of a pinned buffer):
        one copy from the page cache, rows put the right way up on the way."""
        if self.bpp != 8:
            raise ValueError("read_into is for 8-bit streams")
        off, ln = self._frames[i]
        h, w = self.height, self.width
        if ln < self._stride * h:
            raise ValueError("frame %d is truncated" % i)
        a = np.frombuffer(self._mm, dtype=np.uint8, count=self._stride * h, offset=off).reshape(h, self._stride)
        np.copyto(out, (a[::-1] if self._bottom_up else a)[:, :w])

    def record_layout(self, start, end):
        """Frames [start, end) as one region of the file: (file offset of frame ``start``'s pixel data, bytes from one
        frame's data to the next, bytes of the region) when they are evenly spaced and complete (how every plain
        uncompressed AVI stores them: '00db' + size + DIB, nothing in between), else None."""
        if not 0 <= start < end <= len(self._frames):
            return None
        need = self._stride * self.height
        off0, ln0 = self._frames[start]
        if ln0 < need:
            return None
        if end - start == 1:
            return off0, need, need
        rec = self._frames[start + 1][0] - off0
        if rec < need:
            return None
        for k in range(start, end):
            off, ln = self._frames[k]
            if off != off0 + (k - start) * rec or ln < need:
                return None
        return off0, rec, (end - start - 1) * rec + need

    def read_region_into(self, offset, out, threads=4, piece=32 << 20):
        """``len(out)`` bytes of the file from ``offset`` into ``out`` (a writable, contiguous uint8 buffer, e.g. pinned
        memory): a few large positional reads from the page cache, in parallel (they release the GIL)."""
        mv = memoryview(out).cast("B")
        total = len(mv)
        fd = self._fh.fileno()
        spans = [(a, min(total, a + piece)) for a in range(0, total, piece)]

        def one(span):
            a, b = span
            while a < b:
                got = os.preadv(fd, [mv[a:b]], offset + a)
                if got <= 0:
                    raise OSError("short read from %s" % self.filename)
                a += got
        if threads <= 1 or len(spans) <= 1:
            for sp in spans:
                one(sp)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(threads, len(spans))) as pool:
                list(pool.map(one, spans))

    @property
    def stride(self):
        return self._stride

    @property
    def bottom_up(self):
        return self._bottom_up

    def close(self):
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:
                pass
            self._mm = None
        if self._fh is not None:
            self._fh.close()
            self._fh = None


class VideoCapture:
    """The subset of cv2.VideoCapture the detection module uses (ip.py:62-84, 1180-1183), over AviReader.
    8-bit streams are returned as grey-coded BGR, as cv2 does."""
    CAP_PROP_POS_FRAMES, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 1, 5, 7

    def __init__(self, filename):
        self._pos = 0
        try:
            self._r = AviReader(filename)
        except (OSError, ValueError):
            self._r = None

    def isOpened(self):
        return self._r is not None

    def set(self, prop, value):
        if prop == self.CAP_PROP_POS_FRAMES:
            self._pos = int(value)
            return True
        return False

    def get(self, prop):
        if self._r is None:
            return 0.0
        if prop == self.CAP_PROP_FRAME_COUNT:
            return float(self._r.frame_count)
        if prop == self.CAP_PROP_FPS:
            return float(self._r.fps)
        if prop == self.CAP_PROP_POS_FRAMES:
            return float(self._pos)
        return 0.0

    def read(self):
        if self._r is None or self._pos >= self._r.frame_count:
            return False, None
        f = self._r.read(self._pos)
        self._pos += 1
        if f.ndim == 2:
            f = self._r.palette[f]                          # indices through the colour table -> BGR, as cv2 does
        return True, f

    def is_gray(self):
        """True for an 8-bit stream whose colour table is the grey ramp: read_gray() then returns the stored
        frames themselves (what cvtColor(BGR2GRAY) gives back for the grey-coded BGR frames read() returns).
        Any other table goes through read() and the luma conversion."""
        return self._r is not None and self._r.grey_palette

    def read_gray(self):
        if self._r is None or self._pos >= self._r.frame_count:
            return False, None
        f = self._r.read(self._pos)
        self._pos += 1
        return True, f

    def release(self):
        if self._r is not None:
            self._r.close()
            self._r = None

"""Host side of the device PNG path: HO3D v2 -- the dataset of BASELINE.json's configs[1..3] -- stores its frames
as rgb/NNNN.png (anakin/datasets/ho3d.py:181), decoded in the reference's DataLoader workers by `Image.open(path).convert("RGB")`
(ho3d.py:228-231: Pillow = zlib inflate + the PNG specification's scanline reconstruction, libImaging/ZipDecode.c).

Split of the work: the chunk walk and the zlib inflate of the IDAT stream run here, on a THREAD POOL (zlib releases the GIL; one stream is
a serial LZ77 + Huffman chain, a batch has 40 - 160 of them and the host has the cores); the inflated scanlines are uploaded once and
`ab_png_unfilter_batch` (csrc/png.hip) rebuilds the five filter types and writes RGBX frames -- bit-identical to Pillow
(tests/golden/png_cases.npz, tests/test_gpu_png.py).  Files outside `LAYOUT` raise PngUnsupported before any device work; the caller then
keeps Pillow for them, as the reference does for every file."""
import ctypes
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib as L

SIGNATURE = b"\x89PNG\r\n\x1a\n"
DESC_INTS = 8
# (bit depth, colour type) -> (bytes per pixel, byte offsets of the R, G, B samples): what convert("RGB") keeps -- alpha dropped, the high
# byte of 16-bit samples, a grey value three times
LAYOUT = {(8, 2): (3, (0, 1, 2)), (8, 6): (4, (0, 1, 2)), (16, 2): (6, (0, 2, 4)), (16, 6): (8, (0, 2, 4)), (8, 0): (1, (0, 0, 0))}


class PngUnsupported(ValueError):
    """A file the device path does not cover (interlaced, palette, 1/2/4-bit, 16-bit grey, grey + alpha, not a PNG, malformed)."""


class PngInfo:
    __slots__ = ("width", "height", "bpp", "chan", "idat", "raw_bytes")


def parse(data) -> PngInfo:
    """Chunk walk of a .png file: IHDR + the (offset, length) of every IDAT payload.  No inflate, no CRC check (Pillow skips it for IDAT too
    unless LOAD_TRUNCATED_IMAGES is involved; a corrupt stream fails in zlib)."""
    mv = memoryview(data)
    if len(mv) < 33 or bytes(mv[:8]) != SIGNATURE:
        raise PngUnsupported("not a PNG file")
    pos, hdr, idat = 8, None, []
    n_total = len(mv)
    while pos + 8 <= n_total:
        n, typ = struct.unpack_from(">I4s", mv, pos)
        if pos + 12 + n > n_total:
            raise PngUnsupported("truncated chunk")
        if typ == b"IHDR":
            if n != 13:
                raise PngUnsupported("IHDR length")
            hdr = struct.unpack_from(">IIBBBBB", mv, pos + 8)
        elif typ == b"IDAT":
            idat.append((pos + 8, n))
        elif typ == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise PngUnsupported("no IHDR / IDAT")
    w, h, depth, ctype, comp, flt, lace = hdr
    if comp or flt:
        raise PngUnsupported("compression / filter method")
    if lace:
        raise PngUnsupported("interlaced (Adam7) files are not covered")
    if (depth, ctype) not in LAYOUT:
        raise PngUnsupported(f"bit depth {depth} / colour type {ctype} is not covered (8 / 16-bit RGB and RGBA, 8-bit grey are)")
    if w <= 0 or h <= 0 or w > 16384:
        raise PngUnsupported("image size")
    it = PngInfo()
    it.width, it.height = int(w), int(h)
    it.bpp, it.chan = LAYOUT[(depth, ctype)]
    it.idat = idat
    it.raw_bytes = it.height * (1 + it.width * it.bpp)
    return it


class _ZStream(ctypes.Structure):          # zlib.h's z_stream (LP64)
    _fields_ = [("next_in", ctypes.c_void_p), ("avail_in", ctypes.c_uint), ("total_in", ctypes.c_ulong), ("next_out", ctypes.c_void_p),
                ("avail_out", ctypes.c_uint), ("total_out", ctypes.c_ulong), ("msg", ctypes.c_char_p), ("state", ctypes.c_void_p),
                ("zalloc", ctypes.c_void_p), ("zfree", ctypes.c_void_p), ("opaque", ctypes.c_void_p), ("data_type", ctypes.c_int),
                ("adler", ctypes.c_ulong), ("reserved", ctypes.c_ulong)]


_libz = None


def _zlib_c():
    """libz itself (the library Python's zlib module and Pillow link), called through ctypes: inflate() then writes STRAIGHT into the pinned
    staging buffer, IDAT chunk after IDAT chunk, and the GIL is released for the whole call -- the zlib module would hand back a bytes object
    per file (a 0.9 MB copy under the GIL on top of the join of the IDAT payloads), which with 16 - 32 pool threads is what starves the
    thread that launches the training steps.  False when the library cannot be bound (the zlib module is used then)."""
    global _libz
    if _libz is None:
        try:
            import ctypes.util
            z = ctypes.CDLL(ctypes.util.find_library("z") or "libz.so.1")
            z.zlibVersion.restype = ctypes.c_char_p
            z.inflateInit_.argtypes = [ctypes.POINTER(_ZStream), ctypes.c_char_p, ctypes.c_int]
            z.inflate.argtypes = [ctypes.POINTER(_ZStream), ctypes.c_int]
            z.inflateEnd.argtypes = [ctypes.POINTER(_ZStream)]
            _libz = (z, z.zlibVersion())
        except (OSError, AttributeError):
            _libz = False
    return _libz


def inflate_into(data, info: PngInfo, dst: np.ndarray):
    """zlib-inflate the file's IDAT stream into dst (uint8 [info.raw_bytes], contiguous).  Raises PngUnsupported on a corrupt / short stream."""
    zc = _zlib_c()
    if zc:
        z, ver = zc
        src = np.frombuffer(data, np.uint8)
        base = src.ctypes.data
        s = _ZStream()
        if z.inflateInit_(ctypes.byref(s), ver, ctypes.sizeof(_ZStream)) != 0:
            raise PngUnsupported("inflateInit failed")
        s.next_out, s.avail_out = dst.ctypes.data, info.raw_bytes
        rc = 0
        try:
            for o, n in info.idat:
                if n == 0:          # a legal zero-length IDAT chunk: inflate() with no input returns Z_BUF_ERROR
                    continue
                s.next_in, s.avail_in = base + o, n
                rc = z.inflate(ctypes.byref(s), 0)              # Z_NO_FLUSH
                if rc not in (0, 1) or (rc == 0 and s.avail_in):      # an error, or output space exhausted with input left
                    break
            done, avail_in = s.total_out, s.avail_in
        finally:
            z.inflateEnd(ctypes.byref(s))
        if rc != 1 or done != info.raw_bytes or avail_in:      # Z_STREAM_END exactly at the promised size
            raise PngUnsupported("corrupt IDAT stream, or scanline bytes do not match the header")
        _check_filter_types(info, dst)
        return
    mv = memoryview(data)
    if len(info.idat) == 1:
        o, n = info.idat[0]
        payload = mv[o:o + n]
    else:
        payload = b"".join(mv[o:o + n] for o, n in info.idat)
    try:
        raw = zlib.decompress(payload, bufsize=info.raw_bytes)
    except zlib.error as e:
        raise PngUnsupported(f"corrupt IDAT stream: {e}") from None
    if len(raw) != info.raw_bytes:
        raise PngUnsupported("scanline bytes do not match the header")
    dst[:] = np.frombuffer(raw, np.uint8)
    _check_filter_types(info, dst)


def _check_filter_types(info: PngInfo, raw: np.ndarray):
    """Every scanline starts with its filter type 0..4; Pillow's ZipDecode raises on anything else, the device kernel's output for such a
    line is undefined (ab_png_unfilter_batch is called without a status word): refuse on the host, H bytes per frame."""
    pitch = info.raw_bytes // info.height
    if pitch * info.height == info.raw_bytes and raw[:info.raw_bytes:pitch].max(initial=0) > 4:
        raise PngUnsupported("scanline with a filter type above 4")


_pool = None


def effective_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a container on a 256-thread host can
    be capped at 16: the MI355X boxes of this project are -- /sys/fs/cgroup/cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def pool(workers=None):
    """The process-wide decode pool (also used by realdata.RealBatcher for Pillow decodes of files no device path covers): AB_DECODE_WORKERS,
    default min(20, usable CPUs - 3) -- the reference's DataLoader runs 20 worker processes (anakin/opt.py:16).  Three CPUs stay free for the
    thread that launches the steps and the runtime's own threads: under a cgroup quota every busy thread beyond it THROTTLES the whole process
    (measured on a 16-CPU quota, 40 .png frames per step: 32 threads 12.5 - 14.6 ms per step, 16 threads 12.2, 12 threads 11.6)."""
    global _pool
    if _pool is None:
        n = int(os.environ.get("AB_DECODE_WORKERS", 0)) or workers or min(20, max(1, effective_cpus() - 3))
        _pool = ThreadPoolExecutor(max_workers=max(1, n), thread_name_prefix="ab-decode")
    return _pool


class PngDecoder:
    """decoder = PngDecoder(device); frames = decoder.decode(list_of_file_bytes, out=uint8 [n, H, W, 4] device tensor).
    Keeps two pinned staging buffers between calls; the device copy of the scanlines is a per-call allocation of the caching allocator
    (stream-ordered: calls from different streams share nothing but the staging buffers, which are guarded by events)."""

    def __init__(self, device="cuda", workers=None):
        self.dev = torch.device(device)
        self._workers = workers
        self._pins, self._evs, self._k = [None, None], [None, None], 0

    def submit(self, files, infos=None):
        """Start the zlib inflates of `files` on the pool (into one of the two pinned staging buffers) and return at once; complete() takes
        the job from there.  At most two jobs may be open at a time (one per staging buffer)."""
        infos = infos or [parse(f) for f in files]
        n = len(files)
        desc = np.zeros((n, DESC_INTS), np.int32)
        raw_off, o = [], 0
        for i, it in enumerate(infos):
            raw_off.append(o)
            lo = o & 0xFFFFFFFF
            desc[i, :6] = (lo - (1 << 32) if lo >= 1 << 31 else lo, o >> 32, it.width, it.height, it.bpp, it.chan[0] | it.chan[1] << 8 | it.chan[2] << 16)
            o += (it.raw_bytes + 15) & ~15
        desc_off = o
        need = desc_off + desc.nbytes
        k = self._k = self._k ^ 1
        if self._pins[k] is None or self._pins[k].numel() < need:
            self._pins[k] = torch.empty(max(need * 5 // 4, 1 << 20), dtype=torch.uint8).pin_memory()
            self._evs[k] = None
        if self._evs[k] is not None:
            self._evs[k].synchronize()          # the upload that last read this staging buffer (two jobs ago) has finished
            self._evs[k] = None
        stage = self._pins[k].numpy()
        ex = pool(self._workers)
        futs = [ex.submit(inflate_into, f, it, stage[ro:ro + it.raw_bytes]) for f, it, ro in zip(files, infos, raw_off)]
        return dict(k=k, futs=futs, infos=infos, desc=desc, desc_off=desc_off, need=need, n=n)

    def complete(self, job, out=None, channels=4):
        """Wait for the job's inflates, upload the scanlines (one asynchronous copy) and rebuild them on the current stream.
        out: uint8 device tensor [n, H, W, channels] every file must fit exactly, or None: a list of [H_i, W_i, channels] tensors."""
        infos, n, desc = job["infos"], job["n"], job["desc"]
        if out is not None:
            if out.dtype != torch.uint8 or out.dim() != 4 or out.shape[0] < n or out.shape[3] != channels or not out.is_contiguous():
                raise ValueError("out: contiguous uint8 [n, H, W, channels]")
            H, W = int(out.shape[1]), int(out.shape[2])
            for it in infos:
                if (it.height, it.width) != (H, W):
                    raise ValueError(f"frame of {it.width} x {it.height} in a batch of {W} x {H}")
            out_off, out_pitch = [i * H * W for i in range(n)], [W] * n
            res = out[:n]
        else:
            sizes = [it.width * it.height for it in infos]
            out_off = list(np.cumsum([0] + sizes[:-1]))
            out_pitch = [it.width for it in infos]
            flat = torch.empty(int(sum(sizes)) * channels, dtype=torch.uint8, device=self.dev)
            res = [flat[o * channels:(o + s) * channels].view(it.height, it.width, channels) for o, s, it in zip(out_off, sizes, infos)]
            out = flat
        err = None
        for f in job["futs"]:
            try:
                f.result()
            except PngUnsupported as e:          # (every inflate is awaited before the staging buffer is given up)
                err = err or e
        if err is not None:
            raise err
        if n == 0:
            return res
        k, desc_off, need = job["k"], job["desc_off"], job["need"]
        desc[:, 6], desc[:, 7] = out_off, out_pitch
        self._pins[k].numpy()[desc_off:need] = desc.view(np.uint8).reshape(-1)
        blob = torch.empty(need + 16, dtype=torch.uint8, device=self.dev)      # (the kernel's sample fetch is up to 8 bytes wide)
        blob[:need].copy_(self._pins[k][:need], non_blocking=True)
        self._evs[k] = torch.cuda.Event()
        self._evs[k].record()
        L.check(L.lib().ab_png_unfilter_batch(L.ptr(blob), L.ptr(blob[desc_off:need].view(torch.int32)), L.i(n), L.i(max(it.width for it in infos)),
                                              L.i(max(it.bpp for it in infos)), L.i(channels), L.view_ptr(out), L.ptr(None), L.stream()), "ab_png_unfilter_batch")
        return res

    def decode(self, files, out=None, channels=4, infos=None):
        """files: bytes-like PNG files -> frames (see complete()).  Raises PngUnsupported (before any device work) for files outside LAYOUT
        or with a corrupt stream."""
        return self.complete(self.submit(files, infos), out=out, channels=channels)

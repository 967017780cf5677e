"""Device-side JPEG decode for the real-data half.  The reference decodes every frame with
`Image.open(path).convert("RGB")` in a DataLoader worker (anakin/datasets/ho3d.py:228-231, dexycb.py:226-229, fhb.py:257-260); here the
FILE BYTES of a batch are uploaded (a tenth of the decoded pixels) and `ab_jpeg_decode_batch` (csrc/jpeg.hip) produces the same RGB
bytes on the device -- Huffman decode included.  The host only walks the marker segments in front of the scan (`parse`)."""
import numpy as np
import torch

from . import _lib as L

DESC_INTS = 40
ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class JpegUnsupported(ValueError):
    """A file the device decoder does not cover (progressive, arithmetic, 12-bit, CMYK, 4:4:0 / 4:1:1, multi-scan, not a JPEG): the caller
    decodes it with Pillow, as the reference does for every file."""


class JpegInfo:
    __slots__ = ("width", "height", "ncomp", "comps", "hmax", "vmax", "ri", "qt", "ht", "scan_off", "scan_len", "segs", "mcux", "mcuy", "bpm")


def parse(data) -> JpegInfo:
    """Marker walk up to the start of scan (ITU T.81 B.2): frame / scan headers, DQT, DHT, DRI, then the extent of the entropy-coded
    data and, with a restart interval, of each interval.  Anything the kernels do not cover, and any malformed / truncated header, raises
    JpegUnsupported."""
    try:
        return _parse(data)
    except (IndexError, ValueError) as e:
        if isinstance(e, JpegUnsupported):
            raise
        raise JpegUnsupported(f"malformed JPEG header: {e}") from None


def _parse(data) -> JpegInfo:
    mv = memoryview(data)
    n = len(mv)
    if n < 4 or mv[0] != 0xFF or mv[1] != 0xD8:
        raise JpegUnsupported("not a JPEG file")
    info = JpegInfo()
    info.ri, info.ncomp = 0, 0
    qt = np.zeros((4, 64), np.uint16)
    qt_ok = [False] * 4
    ht = np.zeros((8, 272), np.uint8)
    ht_ok = [False] * 8
    adobe_tf, p, sos = -1, 2, -1
    while p + 4 <= n:
        if mv[p] != 0xFF:
            raise JpegUnsupported("marker expected")
        m = mv[p + 1]
        if m == 0xFF:
            p += 1
            continue
        ln = (mv[p + 2] << 8) | mv[p + 3]
        s, e = p + 4, p + 2 + ln
        if e > n:
            raise JpegUnsupported("truncated segment")
        if m == 0xDB:
            while s < e:
                pq, tq = mv[s] >> 4, mv[s] & 15
                s += 1
                if tq > 3:
                    raise JpegUnsupported("quantisation table id")
                if pq:
                    vals = np.frombuffer(mv[s:s + 128], ">u2").astype(np.uint16)
                    s += 128
                else:
                    vals = np.frombuffer(mv[s:s + 64], np.uint8).astype(np.uint16)
                    s += 64
                qt[tq, ZIGZAG] = vals
                qt_ok[tq] = True
        elif m == 0xC4:
            while s < e:
                tc, th = mv[s] >> 4, mv[s] & 15
                if tc > 1 or th > 3:
                    raise JpegUnsupported("Huffman table id")
                tot = int(sum(mv[s + 1:s + 17]))
                if tot > 256:
                    raise JpegUnsupported("Huffman table size")
                slot = tc * 4 + th
                ht[slot] = 0
                ht[slot, :16 + tot] = np.frombuffer(mv[s + 1:s + 17 + tot], np.uint8)
                ht_ok[slot] = True
                s += 17 + tot
        elif m in (0xC0, 0xC1):
            if mv[s] != 8:
                raise JpegUnsupported("sample precision")
            info.height, info.width, info.ncomp = (mv[s + 1] << 8) | mv[s + 2], (mv[s + 3] << 8) | mv[s + 4], mv[s + 5]
            if info.ncomp not in (1, 3):
                raise JpegUnsupported("component count")
            info.comps = [[mv[s + 6 + 3 * i], mv[s + 7 + 3 * i] >> 4, mv[s + 7 + 3 * i] & 15, mv[s + 8 + 3 * i], 0, 0] for i in range(info.ncomp)]
        elif m == 0xC2 or (0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC)):
            raise JpegUnsupported("progressive / lossless / arithmetic-coded file")
        elif m == 0xDD:
            info.ri = (mv[s] << 8) | mv[s + 1]
        elif m == 0xEE and ln >= 14 and bytes(mv[s:s + 5]) == b"Adobe":
            adobe_tf = mv[s + 11]
        elif m == 0xDA:
            if not info.ncomp or mv[s] != info.ncomp:
                raise JpegUnsupported("non-interleaved (multi-scan) file")
            for i in range(info.ncomp):
                if mv[s + 1 + 2 * i] != info.comps[i][0]:
                    raise JpegUnsupported("scan component order")
                info.comps[i][4], info.comps[i][5] = mv[s + 2 + 2 * i] >> 4, mv[s + 2 + 2 * i] & 15
            if mv[s + 1 + 2 * info.ncomp] != 0 or mv[s + 2 + 2 * info.ncomp] != 63:
                raise JpegUnsupported("spectral selection")
            sos = e
            break
        p = e
    if sos < 0 or not info.ncomp or not info.width or not info.height:
        raise JpegUnsupported("no frame / scan header")
    cs = info.comps
    if info.ncomp == 3:
        if adobe_tf == 0 or bytes(c[0] for c in cs) == b"RGB":
            raise JpegUnsupported("RGB-coded JPEG")
        if not (cs[1][1] == cs[1][2] == cs[2][1] == cs[2][2] == 1 and (cs[0][1], cs[0][2]) in ((1, 1), (2, 1), (2, 2))):
            raise JpegUnsupported("sampling factors other than 4:4:4 / 4:2:2 / 4:2:0")
    else:
        cs[0][1] = cs[0][2] = 1                 # a single-component scan is never interleaved: one block per MCU
    for c in cs:
        if c[3] > 3 or not qt_ok[c[3]] or c[4] > 3 or c[5] > 3 or not ht_ok[c[4]] or not ht_ok[4 + c[5]]:
            raise JpegUnsupported("missing table")
    info.hmax, info.vmax = max(c[1] for c in cs), max(c[2] for c in cs)
    info.mcux = -(-info.width // (8 * info.hmax))
    info.mcuy = -(-info.height // (8 * info.vmax))
    info.bpm = sum(c[1] * c[2] for c in cs)
    info.qt, info.ht = qt, ht
    # ---- extent of the scan; restart intervals
    b = data if isinstance(data, (bytes, bytearray)) else bytes(mv)
    nmcu = info.mcux * info.mcuy
    if info.ri:
        arr = np.frombuffer(b, np.uint8, offset=sos)
        idx = np.flatnonzero(arr[:-1] == 0xFF)
        nxt = arr[idx + 1]
        stop = idx[(nxt != 0) & ((nxt < 0xD0) | (nxt > 0xD7)) & (nxt != 0xFF)]
        end = int(stop[0]) if len(stop) else len(arr)
        rst = idx[(nxt >= 0xD0) & (nxt <= 0xD7) & (idx < end)]
        starts = np.concatenate([[0], rst + 2])
        ends = np.concatenate([rst, [end]])
        nseg = -(-nmcu // info.ri)
        if len(starts) < nseg:
            raise JpegUnsupported("missing restart markers")
        info.segs = np.stack([starts[:nseg], ends[:nseg] - starts[:nseg]], 1).astype(np.int64)
    else:
        # the first marker behind SOS that is neither a stuffed 0xFF00, a fill byte nor RSTn ends the entropy-coded segment (normally EOI) --
        # NOT the last FFD9 of the file: MPF / appended previews carry EOIs of their own behind the first image
        arr = np.frombuffer(b, np.uint8, offset=sos)
        idx = np.flatnonzero(arr[:-1] == 0xFF)
        nxt = arr[idx + 1]
        stop = idx[(nxt != 0) & ((nxt < 0xD0) | (nxt > 0xD7)) & (nxt != 0xFF)]
        end = int(stop[0]) if len(stop) else len(arr)
        info.segs = np.array([[0, end]], np.int64)
    info.scan_off, info.scan_len = sos, int(info.segs[-1, 0] + info.segs[-1, 1])
    return info


class _Plan:
    """Everything ab_jpeg_decode_batch needs for one batch of files, laid out in ONE host blob (one upload)."""

    def __init__(self, files, infos, sub_bytes, out_off, out_pitch):
        n = len(files)
        desc = np.zeros((n, DESC_INTS), np.int32)
        segs, qts, hts = [], [], []
        ht_index = {}
        data_off, sub_base, blk_base, plane_base, seg_off = 0, 0, 0, 0, 0
        self.max_blocks = self.max_w = self.max_h = self.max_sub = 0
        for i, (f, it) in enumerate(zip(files, infos)):
            nmcu = it.mcux * it.mcuy
            nblk = nmcu * it.bpm
            sg = np.zeros((len(it.segs), 4), np.int64)
            sg[:, :2] = it.segs
            nsub = np.maximum(1, -(-it.segs[:, 1] // sub_bytes))
            sg[:, 2] = np.cumsum(nsub) - nsub
            sg[:, 3] = np.arange(len(it.segs)) * (it.ri * it.bpm)
            segs.append(sg)
            key = it.ht.tobytes()
            if key not in ht_index:
                ht_index[key] = len(hts)
                hts.append(it.ht)
            d = desc[i]
            d[0], d[1], d[2], d[3], d[4], d[5], d[6] = data_off + it.scan_off, it.scan_len, it.width, it.height, it.ncomp, it.hmax, it.vmax
            for c, cc in enumerate(it.comps):
                d[7 + 5 * c:12 + 5 * c] = cc[1:6]
            d[22], d[23], d[24], d[25], d[26] = it.ri, seg_off, len(sg), sub_base, int(nsub.sum())
            d[27], d[28], d[29], d[30], d[31] = blk_base, nblk, plane_base, out_off[i], out_pitch[i]
            d[32], d[33], d[34], d[35], d[36] = it.mcux, it.mcuy, it.bpm, i, ht_index[key]
            qts.append(it.qt)
            data_off += (len(f) + 15) & ~15
            sub_base += int(nsub.sum())
            blk_base += nblk
            plane_base += (nblk * 64 + 255) & ~255
            seg_off += len(sg)
            self.max_blocks, self.max_w, self.max_h = max(self.max_blocks, nblk), max(self.max_w, it.width), max(self.max_h, it.height)
            self.max_sub = max(self.max_sub, int(nsub.sum()))
        if data_off * 8 >= 1 << 32 or blk_base * 64 >= 1 << 31:
            raise ValueError("batch too large for 32-bit bit positions / coefficient offsets: decode it in smaller batches")
        self.n, self.sub_bytes = n, sub_bytes
        self.total_blocks, self.total_sub, self.plane_bytes, self.data_bytes = blk_base, sub_base, plane_base, data_off
        self.desc, self.segs = desc, np.concatenate(segs).astype(np.int32)
        self.qt, self.ht = np.stack(qts), np.stack(hts)

    def pack(self, files, blob):
        """-> byte offsets of (data, desc, segs, qtabs, htabs) in `blob` (a uint8 numpy view of pinned memory, large enough)."""
        offs, o = [], 0
        parts = [None, self.desc, self.segs, self.qt, self.ht]
        sizes = [self.data_bytes + 16] + [a.nbytes for a in parts[1:]]
        for sz in sizes:
            offs.append(o)
            o += (sz + 255) & ~255
        p = offs[0]
        for f in files:
            blob[p:p + len(f)] = np.frombuffer(f, np.uint8)
            p += (len(f) + 15) & ~15
        for a, of in zip(parts[1:], offs[1:]):
            blob[of:of + a.nbytes] = a.view(np.uint8).reshape(-1)
        return offs, o

    def blob_bytes(self):
        return sum(((sz + 255) & ~255) for sz in (self.data_bytes + 16, self.desc.nbytes, self.segs.nbytes, self.qt.nbytes, self.ht.nbytes))


class JpegDecoder:
    """decoder = JpegDecoder(device); frames = decoder.decode(list_of_file_bytes, out=uint8 [n, H, W, 4] device tensor).
    Keeps its pinned staging blob, device blob and workspace between calls."""

    def __init__(self, device="cuda", sub_bytes=128):
        if not 16 <= int(sub_bytes) <= 128:
            raise ValueError("sub_bytes: 16 .. 128 (the subsequence a thread stages in LDS)")
        self.dev, self.sub_bytes = torch.device(device), int(sub_bytes)
        self._pins, self._evs, self._k = [None, None], [None, None], 0      # two pinned staging blobs, reused alternately
        self._dev_blob = self._ws = None

    def decode(self, files, out=None, channels=4, infos=None, timing=None):
        """files: bytes-like JPEG files.  out: uint8 device tensor [n, H, W, channels] every file must fit exactly (as the frames of one
        dataset do), or None: a list of [H_i, W_i, channels] tensors (views of one allocation) is returned.  Raises JpegUnsupported (before
        any device work) if a file is outside what the kernels cover.
        timing: a dict that receives this call's split -- "host_s" (marker walk if infos is None, plan, pack into the pinned blob: host time
        before any device work is enqueued) and "events" (a HIP event pair around the upload + ab_jpeg_decode_batch on the current stream)."""
        import time
        t_host0 = time.perf_counter()
        n = len(files)
        infos = infos or [parse(f) for f in files]
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
        if n == 0:
            return res
        plan = _Plan(files, infos, self.sub_bytes, out_off, out_pitch)
        need = plan.blob_bytes()
        k = self._k = self._k ^ 1
        if self._pins[k] is None or self._pins[k].numel() < need:
            self._pins[k] = torch.empty(max(need * 2, 1 << 20), dtype=torch.uint8).pin_memory()
            self._evs[k] = None
        if self._dev_blob is None or self._dev_blob.numel() < need:
            self._dev_blob = torch.empty(max(need * 2, 1 << 20), dtype=torch.uint8, device=self.dev)
        if self._evs[k] is not None:
            self._evs[k].synchronize()          # the upload that last read this staging blob (two calls ago) has finished
        offs, used = plan.pack(files, self._pins[k].numpy())
        if timing is not None:
            timing["host_s"] = time.perf_counter() - t_host0
            timing["events"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timing["events"][0].record()
        self._dev_blob[:used].copy_(self._pins[k][:used], non_blocking=True)
        self._evs[k] = torch.cuda.Event()
        self._evs[k].record()
        lib = L.lib()
        nseg, nht = len(plan.segs), len(plan.ht)
        wb = lib.ab_jpeg_workspace_bytes(L.l(plan.total_blocks), L.l(plan.total_sub), L.l(plan.plane_bytes), L.l(plan.data_bytes), L.l(nseg), L.i(n),
                                         L.i(nht))
        if self._ws is None or self._ws.numel() < wb:
            self._ws = torch.empty(int(wb * 1.5), dtype=torch.uint8, device=self.dev)
        part = lambda k, nb: self._dev_blob[offs[k]:offs[k] + nb]      # noqa: E731
        L.check(lib.ab_jpeg_decode_batch(L.ptr(part(0, plan.data_bytes + 16)), L.ptr(part(1, plan.desc.nbytes).view(torch.int32)),
                                         L.ptr(part(2, plan.segs.nbytes).view(torch.int32)), L.ptr(part(3, plan.qt.nbytes)),
                                         L.ptr(part(4, plan.ht.nbytes)), L.i(n), L.i(nht), L.i(plan.sub_bytes), L.l(plan.total_blocks),
                                         L.l(plan.total_sub), L.l(plan.plane_bytes), L.l(plan.data_bytes), L.l(nseg), L.i(plan.max_blocks),
                                         L.i(plan.max_w), L.i(plan.max_h), L.i(plan.max_sub), L.i(channels), L.view_ptr(out), L.ptr(self._ws), L.stream()),
                "ab_jpeg_decode_batch")
        self._last = (plan.total_blocks, plan.total_sub, plan.plane_bytes, plan.data_bytes, nseg, n)
        if timing is not None:
            timing["events"][1].record()
        return res

    R_MAX = 16          # csrc/jpeg.hip

    def last_rounds(self):
        """Diagnostic: synchronisation rounds each image of the last batch took (left by jpeg_finish_kernel in the workspace)."""
        tb, ts, pb, db, nseg, n = self._last
        a256 = lambda x: (x + 255) & ~255      # noqa: E731
        off = a256(tb * 128) + 5 * a256(ts * 4) + a256(pb) + a256(db + 64) + a256(nseg * 8)      # layout: ab_jpeg_decode_batch
        torch.cuda.synchronize()
        return self._ws[off:off + n * (self.R_MAX + 2) * 4].view(torch.int32).view(n, self.R_MAX + 2)[:, self.R_MAX + 1].cpu().numpy()

#!/usr/bin/env python3
"""Minimal OpenEXR scan-line reader for PIZ- (and uncompressed-) compressed HALF/FLOAT images, written to read the
one reference artefact of the hot path: configs_example/scene.exr (256x256 RGB half, PIZ, "Mitsuba version 3.2.0").

    python tools/exr_piz.py /root/reference/configs_example/scene.exr out.npy

Implements the published OpenEXR PIZ scheme (ImfPizCompressor / ImfHuf / ImfWav): per 32-scan-line chunk a
value bitmap -> LUT, canonical Huffman coding with zero-run packed code lengths and a run-length symbol, and the
2-D Haar-like wavelet (14- and 16-bit variants).  No third-party library is needed.
"""
import struct
import sys

import numpy as np

HUF_ENCSIZE = (1 << 16) + 1
SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN


class _Bits:
    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = (self.c << 8) | self.d[self.p]
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_uncompress(data, n_raw):
    im, iM, _table_len, n_bits, _ = struct.unpack_from("<5I", data, 0)
    br = _Bits(data, 20)
    hcode = [0] * HUF_ENCSIZE
    i = im
    while i <= iM:                                    # hufUnpackEncTable
        l = br.get(6)
        if l == LONG_ZEROCODE_RUN:
            i += br.get(8) + SHORTEST_LONG_RUN
        elif l >= SHORT_ZEROCODE_RUN:
            i += l - SHORT_ZEROCODE_RUN + 2
        else:
            hcode[i] = l
            i += 1
    table_end = br.p                                   # the bit reader is byte aligned after the table
    n = [0] * 59                                       # hufCanonicalCodeTable
    for l in hcode:
        n[l] += 1
    c = 0
    for k in range(58, 0, -1):
        nc = (c + n[k]) >> 1
        n[k] = c
        c = nc
    lut = {}
    for sym in range(im, iM + 1):
        l = hcode[sym]
        if l > 0:
            lut[(l, n[l])] = sym
            n[l] += 1
    rlc = iM
    out = np.zeros(n_raw, np.uint16)
    br = _Bits(data, table_end)
    pos, code, length, bits = 0, 0, 0, 0
    get = br.get
    while bits < n_bits and pos < n_raw:
        code = (code << 1) | get(1)
        length += 1
        bits += 1
        sym = lut.get((length, code))
        if sym is None:
            if length > 58:
                raise ValueError("PIZ: invalid Huffman code")
            continue
        if sym == rlc:
            run = get(8)
            bits += 8
            if pos == 0 or pos + run > n_raw:
                raise ValueError("PIZ: invalid run")
            out[pos:pos + run] = out[pos - 1]
            pos += run
        else:
            out[pos] = sym
            pos += 1
        code, length = 0, 0
    if pos != n_raw:
        raise ValueError("PIZ: decoded %d of %d symbols" % (pos, n_raw))
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return (ai & 0xffff).astype(np.uint16), ((ai - hs) & 0xffff).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xffff
    aa = (d + bb - 0x8000) & 0xffff
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, mx):
    """in-place inverse wavelet of a 2-D uint16 array (ny, nx)"""
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    ny, nx = a.shape
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            i00, i10 = dec(a[Y, X], a[Y + p, X])
            i01, i11 = dec(a[Y, X + p], a[Y + p, X + p])
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if (nx & p) and len(ys):
            x = (len(xs)) * p2
            i00, b = dec(a[ys, x], a[ys + p, x])
            a[ys + p, x] = b
            a[ys, x] = i00
        if ny & p:
            y = (len(ys)) * p2
            if len(xs):
                i00, b = dec(a[y, xs], a[y, xs + p])
                a[y, xs + p] = b
                a[y, xs] = i00
        p2 = p
        p >>= 1


def read_exr(path):
    data = open(path, "rb").read()
    if struct.unpack_from("<I", data, 0)[0] != 20000630:
        raise ValueError("not an OpenEXR file")
    pos = 8
    attrs = {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode(); pos = e + 1
        size = struct.unpack_from("<i", data, pos)[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans = []
    cd = attrs["channels"][1]
    p = 0
    while cd[p] != 0:
        e = cd.index(b"\0", p); name = cd[p:e].decode(); p = e + 1
        ptype = struct.unpack_from("<i", cd, p)[0]; p += 16
        chans.append((name, ptype))                    # 0 uint, 1 half, 2 float ; file order is alphabetical
    lines = {0: 1, 2: 1, 3: 16, 4: 32}.get(comp)
    if lines is None:
        raise ValueError("unsupported compression %d (only NONE, ZIPS, ZIP and PIZ)" % comp)
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % n_chunks, data, pos)
    out = {name: np.zeros((H, W), np.float32) for name, _ in chans}
    sizes = [2 if t == 2 or t == 0 else 1 for _, t in chans]   # in uint16 units
    for off in offsets:
        y, dsize = struct.unpack_from("<2i", data, off)
        buf = data[off + 8:off + 8 + dsize]
        ny = min(lines, y1 - y + 1)
        n_raw = sum(sizes) * W * ny
        if comp == 0 or dsize == n_raw * 2:
            raw = np.frombuffer(buf, "<u2", n_raw).copy()
            planar = False
        elif comp in (2, 3):                             # ZIPS / ZIP: inflate, undo the predictor, re-interleave the even and odd bytes
            import zlib
            d = np.frombuffer(zlib.decompress(buf), np.uint8).astype(np.int32)
            t = (np.cumsum(d - 128) + 128) & 255         # t[0] = d[0], t[i] = t[i-1] + d[i] - 128
            half = (len(t) + 1) // 2
            inter = np.empty(len(t), np.uint8); inter[0::2] = t[:half]; inter[1::2] = t[half:]
            raw = np.frombuffer(inter.tobytes(), "<u2", n_raw).copy()
            planar = False
        else:
            mn, mxv = struct.unpack_from("<2H", buf, 0)
            bitmap = np.zeros(8192, np.uint8)
            p = 4
            if mn <= mxv:
                bitmap[mn:mxv + 1] = np.frombuffer(buf, np.uint8, mxv - mn + 1, p); p += mxv - mn + 1
            bits = np.unpackbits(bitmap, bitorder="little")
            bits[0] = 1                                  # zero is always in the LUT
            lut = np.nonzero(bits)[0].astype(np.uint16)
            max_value = len(lut) - 1
            length = struct.unpack_from("<i", buf, p)[0]; p += 4
            raw = _huf_uncompress(buf[p:p + length], n_raw)
            q = 0
            for s in sizes:                              # wavelet per channel (and per 16-bit half of 32-bit types)
                blk = raw[q:q + s * W * ny].reshape(ny, W * s)
                for j in range(s):
                    sub = blk[:, j::s].copy()
                    _wav2_decode(sub, max_value)
                    blk[:, j::s] = sub
                q += s * W * ny
            full_lut = np.zeros(65536, np.uint16); full_lut[:len(lut)] = lut
            raw = full_lut[raw]
            planar = True
        q = 0
        if planar:                                       # tmp buffer is channel-planar; pixels of a 32-bit type are 2 uint16
            for (name, t), s in zip(chans, sizes):
                blk = raw[q:q + s * W * ny].reshape(ny, W * s); q += s * W * ny
                out[name][y - y0:y - y0 + ny] = _to_float(blk, t, W)
        else:
            for r in range(ny):
                for (name, t), s in zip(chans, sizes):
                    out[name][y - y0 + r] = _to_float(raw[q:q + s * W].reshape(1, -1), t, W)[0]; q += s * W
    return out, attrs


def _to_float(blk, t, W):
    if t == 1:
        return blk.view(np.float16).astype(np.float32)
    v = np.ascontiguousarray(blk).view("<u4")
    return v.view(np.float32) if t == 2 else v.astype(np.float32)


if __name__ == "__main__":
    ch, attrs = read_exr(sys.argv[1])
    names = [n for n in ("R", "G", "B") if n in ch] or sorted(ch)
    img = np.stack([ch[n] for n in names], -1)
    print("channels", sorted(ch), "shape", img.shape, "min/max", img.min(), img.max(), "mean", img.mean(axis=(0, 1)))
    if len(sys.argv) > 2:
        np.save(sys.argv[2], img)

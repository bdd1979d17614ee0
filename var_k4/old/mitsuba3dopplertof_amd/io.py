"""Image output for rendered frames: .npy (what the tutorials store, program_runner.py:58,79,147), PFM and
scan-line OpenEXR (the reference's film writes OpenEXR, hdrfilm.cpp `file_format=openexr`; here HALF or FLOAT, ZIP / ZIPS compressed or
uncompressed, channels R,G,B -- readable by any EXR reader incl. tools/exr_piz.py and the library's own radiance-map reader)."""
import struct

import numpy as np


def write_npy(path, img):
    np.save(path, np.asarray(img, dtype=np.float32))


def write_pfm(path, img):
    a = np.asarray(img, dtype=np.float32)
    if a.ndim == 2:
        a = a[..., None]
    h, w, c = a.shape
    if c not in (1, 3):
        raise ValueError("PFM stores 1 or 3 channels")
    with open(path, "wb") as f:
        f.write(("PF\n" if c == 3 else "Pf\n").encode())
        f.write(("%d %d\n-1.0\n" % (w, h)).encode())          # negative scale = little endian
        f.write(a[::-1].astype("<f4").tobytes())               # bottom-to-top scan lines


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


_EXR_COMPRESSION = {"none": (0, 1), "zips": (2, 1), "zip": (3, 16)}     # name -> (attribute value, scan lines per chunk)


def write_exr(path, img, half=True, compression="zip", software="dtof (mitsuba3dopplertof_amd)"):
    """Scan-line OpenEXR 2.0, channels B,G,R in file order (alphabetical), HALF (default) or FLOAT; compression "zip" (default: 16 lines per chunk,
    what most OpenEXR writers use), "zips" (one line per chunk) or "none".  ZIP as OpenEXR defines it: the bytes of a chunk are split into
    even and odd bytes, delta-predicted, then deflated; a chunk that does not shrink is stored raw."""
    import zlib
    a = np.asarray(img, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) image")
    if compression not in _EXR_COMPRESSION:
        raise ValueError('unsupported OpenEXR compression "%s" (none, zips, zip)' % compression)
    code, lines = _EXR_COMPRESSION[compression]
    h, w, _ = a.shape
    ptype = 1 if half else 2
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([code])) +
              _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) +
              _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
              _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) +
              _attr("generatedBy", "string", software.encode()) + b"\0")
    dt = "<f2" if half else "<f4"
    chunks = []
    for y0 in range(0, h, lines):
        raw = b"".join(a[y, :, c].astype(dt).tobytes() for y in range(y0, min(y0 + lines, h)) for c in (2, 1, 0))   # per line: B, G, R
        payload = raw
        if code:
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)             # even bytes, then odd bytes
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255              # predictor
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            if len(z) < len(raw):
                payload = z
        chunks.append((y0, payload))
    head = struct.pack("<II", 20000630, 2) + header
    pos = len(head) + 8 * len(chunks)
    offsets = []
    for _y0, payload in chunks:
        offsets.append(pos); pos += 8 + len(payload)
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack("<%dQ" % len(chunks), *offsets))
        for y0, payload in chunks:
            f.write(struct.pack("<ii", y0, len(payload)) + payload)


def write_image(path, img):
    """Dispatch on the extension: .npy, .pfm, .exr"""
    p = str(path).lower()
    if p.endswith(".npy"):
        write_npy(path, img)
    elif p.endswith(".pfm"):
        write_pfm(path, img)
    elif p.endswith(".exr"):
        write_exr(path, img)
    else:
        raise ValueError('unsupported output format "%s" (use .npy, .pfm or .exr)' % path)


# ------------------------------------------------------------------------------------------------ PNG previews
# The tutorials store a .png next to every .npy (doppler_tutorials/src/utils/image_utils.py:62-135: matplotlib colour maps +
# cv2.imwrite).  Neither matplotlib nor cv2 is a dependency here: a zlib PNG writer and piecewise-linear colour maps through
# the published anchor colours of `viridis` and `RdBu`.
_VIRIDIS = [(0.267, 0.005, 0.329), (0.283, 0.141, 0.458), (0.254, 0.265, 0.530), (0.207, 0.372, 0.553), (0.164, 0.471, 0.558),
            (0.128, 0.567, 0.551), (0.135, 0.659, 0.518), (0.267, 0.749, 0.441), (0.478, 0.821, 0.318), (0.741, 0.873, 0.150),
            (0.993, 0.906, 0.144)]
_RDBU = [(0.404, 0.000, 0.122), (0.698, 0.094, 0.169), (0.839, 0.376, 0.302), (0.957, 0.647, 0.510), (0.992, 0.859, 0.780),
         (0.969, 0.969, 0.969), (0.820, 0.898, 0.941), (0.573, 0.773, 0.871), (0.263, 0.576, 0.765), (0.129, 0.400, 0.675),
         (0.020, 0.188, 0.380)]


def _colormap(x, anchors):
    a = np.asarray(anchors, dtype=np.float64)
    t = np.clip(np.nan_to_num(np.asarray(x, dtype=np.float64)), 0.0, 1.0) * (len(a) - 1)
    i = np.minimum(t.astype(int), len(a) - 2)
    f = (t - i)[..., None]
    return a[i] * (1 - f) + a[i + 1] * f


def write_png(path, rgb8):
    """8-bit RGB (H, W, 3) uint8 -> PNG (zlib, filter 0)"""
    import zlib
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def rgb2luminance(img):
    img = np.asarray(img)
    return 0.2126 * img[..., 0] + 0.7152 * img[..., 1] + 0.0722 * img[..., 2]       # image_utils.py:20-21


def save_tof_image(image, path, vmin=None, vmax=None, vmin_percentile=5, vmax_percentile=95):
    """image_utils.py:108-135: viridis between the 5th and 95th percentile"""
    image = np.asarray(image)
    if image.ndim == 3:
        image = rgb2luminance(image)
    vmin = np.percentile(image, vmin_percentile) if vmin is None else vmin
    vmax = np.percentile(image, vmax_percentile) if vmax is None else vmax
    write_png(path, (_colormap((image - vmin) / max(vmax - vmin, 1e-30), _VIRIDIS) * 255.0).astype(np.uint8))


def save_speed_image(image, path, velocity_range=5):
    """image_utils.py:90-106: RdBu over [-velocity_range, +velocity_range] m/s"""
    image = np.asarray(image)
    if image.ndim == 3:
        image = image[..., 0]
    write_png(path, (_colormap((image + velocity_range) / (2.0 * velocity_range), _RDBU) * 255.0).astype(np.uint8))


def save_hdr_image(image, path):
    """image_utils.py:6-18,72-88: Reinhard-style tone map (limit 1.5) + gamma 2.2"""
    c = np.asarray(image, dtype=np.float64)[..., :3]
    lum = (0.3 * c[..., 0] + 0.6 * c[..., 1] + 0.1 * c[..., 2])[..., None]
    c = np.power(np.maximum(c / (1.0 + lum / 1.5), 0.0), 1.0 / 2.2)
    write_png(path, np.clip(c * 255.0, 0, 255).astype(np.uint8))

"""Image output for rendered frames: .npy (what the tutorials store, program_runner.py:58,79,147), PFM and
scan-line OpenEXR (the reference's film writes OpenEXR, hdrfilm.cpp `file_format=openexr`; here uncompressed HALF or FLOAT,
channels R,G,B -- readable by any EXR reader incl. tools/exr_piz.py)."""
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


def write_exr(path, img, half=True, software="dtof (mitsuba3dopplertof_amd)"):
    """Uncompressed scan-line OpenEXR 2.0, channels B,G,R in file order (alphabetical), HALF (default) or FLOAT."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an (H, W, 3) image")
    h, w, _ = a.shape
    ptype = 1 if half else 2
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", b"\0") +
              _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) +
              _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
              _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) +
              _attr("generatedBy", "string", software.encode()) + b"\0")
    dt = "<f2" if half else "<f4"
    bpp = 2 if half else 4
    line_bytes = 3 * w * bpp
    head = struct.pack("<II", 20000630, 2) + header
    table_pos = len(head)
    first = table_pos + 8 * h
    with open(path, "wb") as f:
        f.write(head)
        f.write(struct.pack("<%dQ" % h, *[first + y * (8 + line_bytes) for y in range(h)]))
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes))
            for c in (2, 1, 0):                                  # B, G, R
                f.write(a[y, :, c].astype(dt).tobytes())


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

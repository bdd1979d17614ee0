#!/usr/bin/env python3
"""Deterministic generator of the benchmark / parity scenes.

    python scenes/make_scenes.py            # (re)writes scenes/*.xml

  cornell_boxes.xml  C1: the geometry, materials, light and camera of the reference's
                     configs_example/scene.xml (Cornell box, two linearly translating cubes, point light at
                     the camera); resolution / spp / integrator settings are <default> parameters.
  cornell_wall.xml   C2/C3: same room, the two boxes replaced by ONE rectangle (the back wall) that translates
                     0.015 towards the camera over the 1.5 ms exposure (10 m/s).
  cornell_area.xml   the C1 room lit by the classic ceiling AREA light instead of the point light (emitter-hit + MIS terms of
                     dopplertofpath.cpp:150-168,214-226; the tutorials' "cornell-box doppler_area" setting).
  domino.xml         C4/C5: ground rectangle + 32x32 cubes instanced from one shapegroup, each with its own
                     pair of keyframes (toppling: rotation about the bottom edge + drift), motion-blur BVH stress.
"""
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

HEADER = """<scene version="3.0.0">
	<default name="spp" value="{spp}" />
	<default name="resx" value="{res}" />
	<default name="resy" value="{res}" />
	<default name="max_depth" value="4" />
	<default name="wave_function_type" value="sinusoidal" />
	<default name="time_sampling_method" value="{tsm}" />
	<default name="antithetic_shift" value="{shift}" />
	<default name="hetero_frequency" value="1.0" />
	<default name="hetero_offset" value="0.0" />
	<default name="path_correlation_depth" value="$max_depth" />
	<default name="time_correlate_number" value="2" />
	<default name="distribution" value="ggx" />
	<default name="sample_visible" value="true" />
	<default name="texfile" value="tex_rgb.png" />
	<integrator type="dopplertofpath">
		<integer name="max_depth" value="$max_depth" />
		<float name="w_g" value="30" />
		<float name="hetero_frequency" value="$hetero_frequency" />
		<float name="hetero_offset" value="$hetero_offset" />
		<float name="antithetic_shift" value="$antithetic_shift" />
		<integer name="path_correlation_depth" value="$path_correlation_depth" />
		<string name="time_sampling_method" value="$time_sampling_method" />
		<string name="wave_function_type" value="$wave_function_type" />
	</integrator>
"""

SENSOR = """	<sensor type="perspective">
		<float name="fov" value="{fov}" />
		<transform name="to_world">
{cam}
		</transform>
		<sampler type="correlated">
			<integer name="sample_count" value="$spp" />
			<integer name="time_correlate_number" value="$time_correlate_number" />
		</sampler>
		<film type="hdrfilm">
			<integer name="width" value="$resx" />
			<integer name="height" value="$resy" />
			<string name="file_format" value="openexr" />
			<string name="pixel_format" value="rgb" />
			<rfilter type="tent" />
		</film>
		<float name="shutter_open" value="0.0" />
		<float name="shutter_close" value="0.0015" />
	</sensor>
"""


def bsdf(ident, rgb):
    return ('\t<bsdf type="twosided" id="%s">\n\t\t<bsdf type="diffuse">\n\t\t\t<rgb name="reflectance" value="%s" />\n'
            '\t\t</bsdf>\n\t</bsdf>\n' % (ident, rgb))


def rect(ident, matrix, bsdf_id, anim_dz=None):
    s = '\t<shape type="rectangle" id="%s">\n' % ident
    if anim_dz is None:
        s += '\t\t<transform name="to_world">\n\t\t\t<matrix value="%s" />\n\t\t</transform>\n' % matrix
    else:
        s += ('\t\t<animation name="to_world">\n\t\t\t<transform time="0">\n\t\t\t\t<matrix value="%s" />\n\t\t\t</transform>\n'
              '\t\t\t<transform time="0.0015">\n\t\t\t\t<matrix value="%s" />\n\t\t\t\t<translate x="0.0" y="0.0" z="%s" />\n'
              '\t\t\t</transform>\n\t\t</animation>\n' % (matrix, matrix, anim_dz))
    s += '\t\t<ref id="%s" />\n\t</shape>\n' % bsdf_id
    return s


def cube(ident, matrix, bsdf_id, dz):
    return ('\t<shape type="cube" id="%s">\n\t\t<ref id="%s" />\n\t\t<animation name="to_world">\n'
            '\t\t\t<transform time="0">\n\t\t\t\t<matrix value="%s" />\n\t\t\t</transform>\n'
            '\t\t\t<transform time="0.0015">\n\t\t\t\t<matrix value="%s" />\n\t\t\t\t<translate x="0.0" y="0.0" z="%s" />\n'
            '\t\t\t</transform>\n\t\t</animation>\n\t</shape>\n' % (ident, bsdf_id, matrix, matrix, dz))


# room of the reference's example scene (configs_example/scene.xml:33-102,127-132): values are data, kept verbatim
CAM = '\t\t\t<matrix value="-1 0 0 0 0 1 0 1 0 0 -1 6.8 0 0 0 1" />'
WALLS = [
    ("Floor", "-4.37114e-008 1 4.37114e-008 0 0 -8.74228e-008 2 0 1 4.37114e-008 1.91069e-015 0 0 0 0 1", "FloorBSDF"),
    ("Ceiling", "-1 7.64274e-015 -1.74846e-007 0 8.74228e-008 8.74228e-008 -2 2 0 -1 -4.37114e-008 0 0 0 0 1", "CeilingBSDF"),
    ("BackWall", "1.91069e-015 1 1.31134e-007 0 1 3.82137e-015 -8.74228e-008 1 -4.37114e-008 1.31134e-007 -2 -1 0 0 0 1", "BackWallBSDF"),
    ("RightWall", "4.37114e-008 -1.74846e-007 2 1 1 3.82137e-015 -8.74228e-008 1 3.82137e-015 1 2.18557e-007 0 0 0 0 1", "RightWallBSDF"),
    ("LeftWall", "-4.37114e-008 8.74228e-008 -2 -1 1 3.82137e-015 -8.74228e-008 1 0 -1 -4.37114e-008 0 0 0 0 1", "LeftWallBSDF"),
]
BSDFS = [("LeftWallBSDF", "0.63, 0.065, 0.05"), ("RightWallBSDF", "0.14, 0.45, 0.091"), ("FloorBSDF", "0.725, 0.71, 0.68"),
         ("CeilingBSDF", "0.725, 0.71, 0.68"), ("BackWallBSDF", "0.725, 0.71, 0.68"), ("ShortBoxBSDF", "0.725, 0.71, 0.68"),
         ("TallBoxBSDF", "0.725, 0.71, 0.68")]
SHORT = "0.0851643 0.289542 1.31134e-008 0.328631 3.72265e-009 1.26563e-008 -0.3 0.3 -0.284951 0.0865363 5.73206e-016 0.374592 0 0 0 1"
TALL = "0.286776 0.098229 -2.29282e-015 -0.335439 -4.36233e-009 1.23382e-008 -0.6 0.6 -0.0997984 0.282266 2.62268e-008 -0.291415 0 0 0 1"
LIGHT = ('\t<emitter type="point">\n\t\t<transform name="to_world">\n' + CAM + '\n\t\t</transform>\n'
         '\t\t<rgb name="intensity" value="100" />\n\t</emitter>\n')


AREA_LIGHT = ('\t<shape type="rectangle" id="Light">\n\t\t<transform name="to_world">\n\t\t\t<scale x="0.25" y="0.2" z="1" />\n'
              '\t\t\t<rotate x="1" angle="90" />\n\t\t\t<translate x="0" y="1.98" z="0" />\n\t\t</transform>\n'
              '\t\t<emitter type="area">\n\t\t\t<rgb name="radiance" value="17, 12, 4" />\n\t\t</emitter>\n\t</shape>\n')


def cornell(moving_wall, res, spp, tsm, shift, area_light=False):
    s = HEADER.format(spp=spp, res=res, tsm=tsm, shift=shift) + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b, anim_dz="0.015" if (moving_wall and name == "BackWall") else None)
    if not moving_wall:
        s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015")
        s += cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + (AREA_LIGHT if area_light else LIGHT) + "</scene>\n"


def sphere(ident, bsdf_id, center, radius, anim_dz=None, emitter=None, extra=""):
    s = '\t<shape type="sphere" id="%s">\n\t\t<point name="center" x="%s" y="%s" z="%s" />\n\t\t<float name="radius" value="%s" />\n%s' % (
        (ident,) + tuple(center) + (radius, extra))
    if anim_dz is not None:
        s += ('\t\t<animation name="to_world">\n\t\t\t<transform time="0">\n\t\t\t\t<translate x="0" y="0" z="0" />\n\t\t\t</transform>\n'
              '\t\t\t<transform time="0.0015">\n\t\t\t\t<translate x="0.0" y="0.0" z="%s" />\n\t\t\t</transform>\n\t\t</animation>\n' % anim_dz)
    if emitter:
        s += '\t\t<emitter type="area">\n\t\t\t<rgb name="radiance" value="%s" />\n\t\t</emitter>\n' % emitter
    if bsdf_id:
        s += '\t\t<ref id="%s" />\n' % bsdf_id
    return s + '\t</shape>\n'


def cornell_spheres(res=128, spp=16, sphere_light=False):
    """the Cornell room with two analytic spheres (one static, one moving towards the camera); lit by the point light at the
    camera, or (sphere_light) by a small spherical area light under the ceiling"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += sphere("StaticSphere", "TallBoxBSDF", ("-0.4", "0.4", "-0.3"), "0.4")
    s += sphere("MovingSphere", "ShortBoxBSDF", ("0.45", "0.3", "0.35"), "0.3", anim_dz="0.015")
    if sphere_light:
        s += sphere("Light", None, ("0", "1.7", "0"), "0.12", emitter="40, 30, 12")
    else:
        s += LIGHT
    return s + "</scene>\n"


MIRROR = ('\t<bsdf type="twosided" id="MirrorBSDF">\n\t\t<bsdf type="conductor">\n\t\t\t<rgb name="eta" value="0.2, 0.92, 1.1" />\n'
          '\t\t\t<rgb name="k" value="3.9, 2.45, 2.14" />\n\t\t</bsdf>\n\t</bsdf>\n')     # copper-like RGB index of refraction
GLASS = '\t<bsdf type="dielectric" id="GlassBSDF">\n\t\t<float name="int_ior" value="1.5" />\n\t\t<string name="ext_ior" value="air" />\n\t</bsdf>\n'


PLASTIC = ('\t<bsdf type="twosided" id="PlasticBSDF">\n\t\t<bsdf type="plastic">\n\t\t\t<rgb name="diffuse_reflectance" value="0.1, 0.27, 0.36" />\n'
           '\t\t\t<float name="int_ior" value="1.9" />\n\t\t</bsdf>\n\t</bsdf>\n')


def write_png(path, pixels):
    """8-bit PNG writer (gray for an H x W array, RGB for H x W x 3): the texture fixtures are generated, not tracked"""
    import struct
    import zlib
    h, w = len(pixels), len(pixels[0])
    rgb = isinstance(pixels[0][0], (tuple, list))
    raw = b"".join(b"\x00" + (bytes(c for px in row for c in px) if rgb else bytes(row)) for row in pixels)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if rgb else 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def texture_files():
    """tex_rgb.png: 16 x 12 RGB pattern (stripes + gradient, no symmetry: a flipped or transposed lookup shows); tex_gray.png: 8 x 8 gray"""
    rgb = [[((x * 37 + y * 11) % 256, (255 - x * 15) % 256 if (x // 2 + y) % 3 else 40, (y * 21 + (x % 4) * 50) % 256) for x in range(16)] for y in range(12)]
    gray = [[(x * x * 3 + y * 29 + 10) % 256 for x in range(8)] for y in range(8)]
    write_png(os.path.join(HERE, "tex_rgb.png"), rgb)
    write_png(os.path.join(HERE, "tex_gray.png"), gray)
    # tex_normal.png: a 16 x 16 tangent-space normal map (bumps: a sine in x, a cosine in y; rgb = (n + 1) / 2), some texels tilted far enough for light leaks
    nm = []
    for y in range(16):
        row = []
        for x in range(16):
            nx, ny = 0.55 * math.sin(x * 0.9 + 0.3), 0.45 * math.cos(y * 1.1) * (1.6 if (x + y) % 5 == 0 else 1.0)
            nz = math.sqrt(max(1.0 - nx * nx - ny * ny, 0.04))
            l = math.sqrt(nx * nx + ny * ny + nz * nz)
            row.append(tuple(int(round(255 * (c / l * 0.5 + 0.5))) for c in (nx, ny, nz)))
        nm.append(row)
    write_png(os.path.join(HERE, "tex_normal.png"), nm)
    try:   # the same pattern, enlarged and JPEG-coded (4:2:0), for the baseline JPEG reader; PIL is test infrastructure
        from PIL import Image
        big = [[rgb[y // 4][x // 4] for x in range(64)] for y in range(48)]
        Image.frombytes("RGB", (64, 48), bytes(c for row in big for px in row for c in px)).save(os.path.join(HERE, "tex_rgb.jpg"), "JPEG", quality=85, subsampling=2)
    except ImportError:
        pass
    sky = env_pixels()
    write_rgbe(os.path.join(HERE, "env_sky.hdr"), sky)
    write_pfm(os.path.join(HERE, "env_sky.pfm"), sky)
    write_exr(os.path.join(HERE, "env_sky.exr"), sky, compression=3, half=False)
    write_png(os.path.join(HERE, "env_sky.png"), [[tuple(min(255, int(255 * min(c, 1.0) ** 0.45)) for c in px) for px in row] for row in sky])


def env_pixels(w=32, h=16):
    """a small sky: blue gradient, a bright warm sun patch, a dim ground -- float RGB rows, top row first"""
    rows = []
    for y in range(h):
        row = []
        for x in range(w):
            up = 1.0 - y / (h - 1)
            r, g, b = 0.15 + 0.35 * up, 0.2 + 0.5 * up, 0.25 + 0.9 * up
            if y > h // 2:
                r, g, b = 0.12 + 0.01 * (x % 5), 0.1, 0.07
            if 3 <= y <= 5 and 20 <= x <= 23:
                r, g, b = 40.0 + 3 * (x - 20), 32.0 + y, 18.0
            row.append((r, g, b))
        rows.append(row)
    return rows


def write_pfm(path, rows):
    import struct
    h, w = len(rows), len(rows[0])
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        for row in reversed(rows):          # PFM stores the bottom row first
            f.write(struct.pack("<%df" % (3 * w), *[c for px in row for c in px]))


def write_rgbe(path, rows, rle=True):
    """Radiance .hdr (32-bit_rle_rgbe), new-style run-length encoded scanlines (as the reference's reader expects for widths 8 .. 32767)"""
    h, w = len(rows), len(rows[0])
    def enc(px):
        m = max(px)
        if m < 1e-32:
            return (0, 0, 0, 0)
        man, e = math.frexp(m)
        k = man * 256.0 / m
        return (int(px[0] * k), int(px[1] * k), int(px[2] * k), e + 128)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for row in rows:
            px = [enc(p) for p in row]
            if not rle:
                f.write(bytes(c for p in px for c in p)); continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                ch, i = [p[c] for p in px], 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and ch[i + run] == ch[i]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, ch[i]])); i += run
                    else:
                        j = i
                        while j < w and j - i < 128 and not (j + 2 < w and ch[j] == ch[j + 1] == ch[j + 2]):
                            j += 1
                        f.write(bytes([j - i]) + bytes(ch[i:j])); i = j


def write_exr(path, rows, compression=3, half=True, decreasing_y=False, alpha=False):
    """scan-line OpenEXR 2.0: channels (A,) B, G, R, HALF or FLOAT, compression 0 (none), 2 (ZIPS: one line per chunk) or 3 (ZIP: 16 lines)"""
    import struct
    import zlib
    import numpy as np
    a = np.asarray(rows, np.float32)
    h, w, _ = a.shape
    names = ([b"A"] if alpha else []) + [b"B", b"G", b"R"]
    planes = ([np.ones((h, w), np.float32)] if alpha else []) + [a[..., 2], a[..., 1], a[..., 0]]
    ptype, dt = (1, "<f2") if half else (2, "<f4")
    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) + attr("dataWindow", "box2i", box) +
              attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([1 if decreasing_y else 0])) +
              attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines = {0: 1, 2: 1, 3: 16}[compression]
    chunks = []
    for y0 in range(0, h, lines):
        raw = b"".join(pl[y].astype(dt).tobytes() for y in range(y0, min(y0 + lines, h)) for pl in planes)
        if compression:
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)          # reorder: even bytes, then odd bytes
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255           # predictor
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            payload = z if len(z) < len(raw) else raw
        else:
            payload = raw
        chunks.append((y0, payload))
    if decreasing_y:
        chunks.reverse()
    head = struct.pack("<II", 20000630, 2) + header
    pos = len(head) + 8 * len(chunks)
    table = {}
    body = b""
    for y0, payload in chunks:
        table[y0] = pos + len(body)
        body += struct.pack("<ii", y0, len(payload)) + payload
    order = sorted(table)          # the offset table is always in increasing-y order
    with open(path, "wb") as f:
        f.write(head + struct.pack("<%dQ" % len(order), *[table[y] for y in order]) + body)


def cornell_thinlens(res=128, spp=16):
    """cornell_boxes.xml seen through a `thinlens` sensor (src/sensors/thinlens.cpp): a 12 cm aperture focused on the front of the short box, so that
    every lane draws an aperture sample between its pixel jitter and its time sample"""
    s = cornell(False, res, spp, "antithetic", "0.5")
    s = s.replace('<sensor type="perspective">', '<sensor type="thinlens">\n\t\t<float name="aperture_radius" value="0.12" />\n\t\t<float name="focus_distance" value="6.2" />')
    assert "thinlens" in s
    return s


def cornell_sun(res=128, spp=16):
    """cornell_env.xml (no ceiling, no back wall) under a `directional` emitter (src/emitters/directional.cpp) given by `direction`, plus a second one
    given by a to_world rotation, beside the point light: delta directions, shadow rays that leave through the open sides"""
    s = cornell_env(res, spp)
    a = s.index('\t<emitter type="constant">'); b = s.index('</emitter>', a) + len('</emitter>\n')
    em = ('\t<emitter type="directional">\n\t\t<vector name="direction" x="-0.3" y="-1" z="-0.4" />\n\t\t<rgb name="irradiance" value="3.0, 2.6, 2.0" />\n\t</emitter>\n'
          '\t<emitter type="directional">\n\t\t<transform name="to_world">\n\t\t\t<rotate x="1" angle="110" />\n\t\t\t<rotate y="1" angle="25" />\n\t\t</transform>\n'
          '\t\t<float name="irradiance" value="0.8" />\n\t</emitter>\n')
    return s[:a] + em + s[b:]


def cornell_envmap(res=128, spp=16, filename="env_sky.hdr", extra=""):
    """cornell_env.xml under an `envmap` emitter (src/emitters/envmap.cpp): a latitude-longitude radiance map, rotated, importance-sampled"""
    s = cornell_env(res, spp)
    a = s.index('\t<emitter type="constant">'); b = s.index('</emitter>', a) + len('</emitter>\n')
    em = ('\t<emitter type="envmap">\n\t\t<string name="filename" value="%s" />\n\t\t<float name="scale" value="0.6" />\n%s'
          '\t\t<transform name="to_world">\n\t\t\t<rotate y="1" angle="40" />\n\t\t\t<rotate x="1" angle="-15" />\n\t\t</transform>\n\t</emitter>\n' % (filename, extra))
    return s[:a] + em + s[b:]


def tex_bsdf(ident, kind, body, plugin="diffuse", prop="reflectance", extra=""):
    return ('\t<bsdf type="twosided" id="%s">\n\t\t<bsdf type="%s">\n%s\t\t\t<texture type="%s" name="%s">\n%s\t\t\t</texture>\n'
            '\t\t</bsdf>\n\t</bsdf>\n' % (ident, plugin, extra, kind, prop, body))


def cornell_textured(res=128, spp=16):
    """cornell_boxes.xml with textures on the diffuse reflectances (src/textures/{checkerboard,bitmap}.cpp): a checkerboard floor (scaled
    to_uv), an RGB bitmap on the back wall (bilinear, repeat, rotated to_uv), a gray bitmap on the left wall (nearest, mirror), a bitmap
    on the moving short box (cube texcoords; clamp) and a smooth-plastic tall box whose diffuse reflectance is a checkerboard"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        if b[0] not in ("FloorBSDF", "BackWallBSDF", "LeftWallBSDF", "ShortBoxBSDF", "TallBoxBSDF"):
            s += bsdf(*b)
    s += tex_bsdf("FloorBSDF", "checkerboard", '\t\t\t\t<rgb name="color0" value="0.7, 0.68, 0.6" />\n\t\t\t\t<rgb name="color1" value="0.12, 0.1, 0.2" />\n'
                  '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="4" y="6" />\n\t\t\t\t\t<translate x="0.25" y="0" />\n\t\t\t\t</transform>\n')
    s += tex_bsdf("BackWallBSDF", "bitmap", '\t\t\t\t<string name="filename" value="$texfile" />\n'
                  '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="2.5" y="1.5" />\n\t\t\t\t\t<rotate z="1" angle="20" />\n\t\t\t\t</transform>\n')
    s += tex_bsdf("LeftWallBSDF", "bitmap", '\t\t\t\t<string name="filename" value="tex_gray.png" />\n\t\t\t\t<string name="filter_type" value="nearest" />\n'
                  '\t\t\t\t<string name="wrap_mode" value="mirror" />\n\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="1.7" y="2.3" />\n\t\t\t\t</transform>\n')
    s += tex_bsdf("ShortBoxBSDF", "bitmap", '\t\t\t\t<string name="filename" value="$texfile" />\n\t\t\t\t<string name="wrap_mode" value="clamp" />\n'
                  '\t\t\t\t<boolean name="raw" value="true" />\n\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="1.5" y="1.5" />\n\t\t\t\t</transform>\n')
    s += tex_bsdf("TallBoxBSDF", "checkerboard", '\t\t\t\t<rgb name="color0" value="0.1, 0.27, 0.36" />\n\t\t\t\t<rgb name="color1" value="0.6, 0.5, 0.1" />\n'
                  '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="3" y="3" />\n\t\t\t\t</transform>\n',
                  plugin="plastic", prop="diffuse_reflectance", extra='\t\t\t<float name="int_ior" value="1.9" />\n')
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + LIGHT + "</scene>\n"


def cornell_masked(res=128, spp=16):
    """cornell_boxes.xml with `mask` BSDFs (src/bsdfs/mask.cpp): the short box's two-sided diffuse BSDF behind a checkerboard opacity, the tall box's two-sided
    plastic (its lobe selection consumes sample1, which the mask rescales) behind a constant opacity 0.6, a free-standing one-sided veil in front of the back wall
    whose opacity is a gray bitmap, the default opacity (0.5) on the left wall; a point light AND an area light (the emitter sample multiplies the nested value
    and density by the opacity, MIS sees the scaled density; a null interaction is a delta sample)"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        if b[0] not in ("ShortBoxBSDF", "TallBoxBSDF", "LeftWallBSDF"):
            s += bsdf(*b)
    s += ('\t<bsdf type="mask" id="ShortBoxBSDF">\n\t\t<texture type="checkerboard" name="opacity">\n\t\t\t<rgb name="color0" value="0.15" />\n\t\t\t<rgb name="color1" value="0.9" />\n'
          '\t\t\t<transform name="to_uv">\n\t\t\t\t<scale x="4" y="4" />\n\t\t\t</transform>\n\t\t</texture>\n'
          '\t\t<bsdf type="twosided">\n\t\t\t<bsdf type="diffuse">\n\t\t\t\t<rgb name="reflectance" value="0.7, 0.6, 0.3" />\n\t\t\t</bsdf>\n\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="mask" id="TallBoxBSDF">\n\t\t<float name="opacity" value="0.6" />\n'
          '\t\t<bsdf type="twosided">\n\t\t\t<bsdf type="plastic">\n\t\t\t\t<rgb name="diffuse_reflectance" value="0.2, 0.5, 0.7" />\n\t\t\t\t<float name="int_ior" value="1.6" />\n\t\t\t</bsdf>\n\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="mask" id="LeftWallBSDF">\n\t\t<bsdf type="twosided">\n\t\t\t<bsdf type="diffuse">\n\t\t\t\t<rgb name="reflectance" value="0.63, 0.065, 0.05" />\n\t\t\t</bsdf>\n\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="mask" id="VeilBSDF">\n\t\t<texture type="bitmap" name="opacity">\n\t\t\t<string name="filename" value="tex_gray.png" />\n\t\t\t<boolean name="raw" value="true" />\n\t\t</texture>\n'
          '\t\t<bsdf type="diffuse">\n\t\t\t<rgb name="reflectance" value="0.4, 0.8, 0.4" />\n\t\t</bsdf>\n\t</bsdf>\n')
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += ('\t<shape type="rectangle" id="Veil">\n\t\t<ref id="VeilBSDF" />\n\t\t<transform name="to_world">\n\t\t\t<scale x="0.6" y="0.5" z="1" />\n'
          '\t\t\t<translate x="0.1" y="1.1" z="-0.55" />\n\t\t</transform>\n\t</shape>\n')
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + LIGHT + AREA_LIGHT + "</scene>\n"


def cornell_textured_light(res=128, spp=16):
    """cornell_boxes.xml lit by three rectangle area emitters whose `radiance` is a texture (src/emitters/area.cpp:129-153: the emitter is then sampled THROUGH the texture):
    an RGB bitmap (bilinear, repeat: DiscreteDistribution2D over the luminance + tent warp), a gray bitmap with the nearest filter and mirror wrap, a checkerboard
    (Texture::sample_position is the identity); no other light"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    light = ('\t<shape type="rectangle" id="%s">\n\t\t<transform name="to_world">\n\t\t\t<scale x="%s" y="%s" z="1" />\n\t\t\t<rotate x="1" angle="90" />\n'
             '\t\t\t<translate x="%s" y="1.98" z="%s" />\n\t\t</transform>\n\t\t<emitter type="area">\n%s\t\t</emitter>\n\t</shape>\n')
    s += light % ("LightA", "0.3", "0.25", "-0.45", "0.1", '\t\t\t<texture type="bitmap" name="radiance"><string name="filename" value="tex_rgb.png" /></texture>\n')
    s += light % ("LightB", "0.2", "0.3", "0.5", "-0.3", '\t\t\t<texture type="bitmap" name="radiance"><string name="filename" value="tex_gray.png" /><boolean name="raw" value="true" />'
                  '<string name="filter_type" value="nearest" /><string name="wrap_mode" value="mirror" /></texture>\n')
    s += light % ("LightC", "0.2", "0.15", "0.1", "0.6", '\t\t\t<texture type="checkerboard" name="radiance"><rgb name="color0" value="6, 1, 0.5" /><rgb name="color1" value="0.5, 2, 9" />'
                  '<transform name="to_uv"><scale x="2" y="3" /></transform></texture>\n')
    return s + "</scene>\n"


def cornell_blend(res=128, spp=16):
    """cornell_boxes.xml with `blendbsdf` BSDFs (src/bsdfs/blendbsdf.cpp): the back wall a two-sided blend of a diffuse and a roughconductor BSDF with a checkerboard weight
    (the adapter outside), the floor a blend of two two-sided BSDFs (plastic, conductor) with a constant weight, the short box a mask around a two-sided blend of a normal-mapped
    diffuse BSDF and a roughplastic with a bitmap weight, the tall box a ONE-sided blend of a diffuse BSDF and a dielectric (a transmitting partner), a free-standing panel with a `twosided` of TWO BSDFs; point + area light"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        if b[0] not in ("ShortBoxBSDF", "TallBoxBSDF", "BackWallBSDF", "FloorBSDF"):
            s += bsdf(*b)
    s += ('\t<bsdf type="twosided" id="BackWallBSDF"><bsdf type="blendbsdf"><texture type="checkerboard" name="weight"><rgb name="color0" value="0.1" /><rgb name="color1" value="0.85" />'
          '<transform name="to_uv"><scale x="3" y="3" /></transform></texture><bsdf type="diffuse"><rgb name="reflectance" value="0.725, 0.71, 0.68" /></bsdf>'
          '<bsdf type="roughconductor"><string name="distribution" value="ggx" /><float name="alpha" value="0.2" /><rgb name="eta" value="0.2, 0.92, 1.1" /><rgb name="k" value="3.9, 2.45, 2.14" /></bsdf></bsdf></bsdf>\n')
    s += ('\t<bsdf type="blendbsdf" id="FloorBSDF"><float name="weight" value="0.35" /><bsdf type="twosided"><bsdf type="plastic"><rgb name="diffuse_reflectance" value="0.6, 0.55, 0.4" /></bsdf></bsdf>'
          '<bsdf type="twosided"><bsdf type="conductor"><rgb name="eta" value="0.2, 0.92, 1.1" /><rgb name="k" value="3.9, 2.45, 2.14" /></bsdf></bsdf></bsdf>\n')
    s += ('\t<bsdf type="mask" id="ShortBoxBSDF"><float name="opacity" value="0.85" /><bsdf type="twosided"><bsdf type="blendbsdf"><texture type="bitmap" name="weight"><string name="filename" value="tex_gray.png" />'
          '<boolean name="raw" value="true" /></texture><bsdf type="normalmap"><texture type="bitmap" name="normalmap"><string name="filename" value="tex_normal.png" /><boolean name="raw" value="true" /></texture>'
          '<bsdf type="diffuse"><rgb name="reflectance" value="0.7, 0.3, 0.2" /></bsdf></bsdf><bsdf type="roughplastic"><string name="distribution" value="beckmann" /><float name="alpha" value="0.15" />'
          '<rgb name="diffuse_reflectance" value="0.2, 0.4, 0.7" /></bsdf></bsdf></bsdf></bsdf>\n')
    # `twosided` with TWO nested BSDFs (twosided.cpp:75-86): a free-standing panel, plastic in front, a normal-mapped rough conductor behind
    s += ('\t<bsdf type="twosided" id="PanelBSDF"><bsdf type="plastic"><rgb name="diffuse_reflectance" value="0.8, 0.7, 0.2" /></bsdf><bsdf type="normalmap"><texture type="bitmap" name="normalmap">'
          '<string name="filename" value="tex_normal.png" /><boolean name="raw" value="true" /></texture><bsdf type="roughconductor"><string name="distribution" value="ggx" /><float name="alpha" value="0.3" />'
          '</bsdf></bsdf></bsdf>\n')
    s += ('\t<shape type="rectangle" id="Panel">\n\t\t<ref id="PanelBSDF" />\n\t\t<transform name="to_world">\n\t\t\t<scale x="0.35" y="0.45" z="1" />\n\t\t\t<rotate y="1" angle="55" />\n'
          '\t\t\t<translate x="0.55" y="1.2" z="0.3" />\n\t\t</transform>\n\t</shape>\n')
    s += ('\t<bsdf type="blendbsdf" id="TallBoxBSDF"><float name="weight" value="0.6" /><bsdf type="diffuse"><rgb name="reflectance" value="0.3, 0.6, 0.8" /></bsdf>'
          '<bsdf type="dielectric"><float name="int_ior" value="1.5" /></bsdf></bsdf>\n')
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + LIGHT + AREA_LIGHT + "</scene>\n"


def cornell_normalmap(res=128, spp=16):
    """cornell_boxes.xml with `normalmap` and `bumpmap` BSDFs (src/bsdfs/normalmap.cpp, bumpmap.cpp): the back wall a two-sided normal-mapped diffuse BSDF (the adapter outside, as exporters write it),
    the floor a two-sided normal-mapped roughconductor, the short box a mask around a two-sided normal-mapped plastic, the tall box a ONE-sided normal-mapped diffuse
    BSDF with a checkerboard "normal map" (two constant tilted normals); point + area light"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        if b[0] not in ("ShortBoxBSDF", "TallBoxBSDF", "BackWallBSDF", "FloorBSDF"):
            s += bsdf(*b)
    nm = ('<texture type="bitmap" name="normalmap"><string name="filename" value="tex_normal.png" /><boolean name="raw" value="true" />'
          '<transform name="to_uv"><scale x="%s" y="%s" /></transform></texture>')
    s += '\t<bsdf type="twosided" id="BackWallBSDF"><bsdf type="normalmap">' + nm % ("2", "2") + '<bsdf type="diffuse"><rgb name="reflectance" value="0.725, 0.71, 0.68" /></bsdf></bsdf></bsdf>\n'
    s += ('\t<bsdf type="twosided" id="FloorBSDF"><bsdf type="normalmap">' + nm % ("3", "1.5") + '<bsdf type="roughconductor"><string name="distribution" value="ggx" /><float name="alpha" value="0.25" />'
          '<rgb name="eta" value="0.2, 0.92, 1.1" /><rgb name="k" value="3.9, 2.45, 2.14" /></bsdf></bsdf></bsdf>\n')
    s += ('\t<bsdf type="mask" id="ShortBoxBSDF"><float name="opacity" value="0.8" /><bsdf type="twosided"><bsdf type="normalmap">' + nm % ("1", "1")
          + '<bsdf type="plastic"><rgb name="diffuse_reflectance" value="0.7, 0.3, 0.2" /></bsdf></bsdf></bsdf></bsdf>\n')
    s += ('\t<bsdf type="normalmap" id="TallBoxBSDF"><texture type="checkerboard" name="normalmap"><rgb name="color0" value="0.62, 0.5, 0.95" /><rgb name="color1" value="0.4, 0.65, 0.9" />'
          '<transform name="to_uv"><scale x="3" y="3" /></transform></texture><bsdf type="diffuse"><rgb name="reflectance" value="0.3, 0.6, 0.8" /></bsdf></bsdf>\n')
    # `bumpmap` (src/bsdfs/bumpmap.cpp): the ceiling a two-sided bump-mapped diffuse BSDF (gray height bitmap, scale 0.02), the right wall a bump-mapped plastic whose
    # height map is the RGB bitmap (luminance), mirror-wrapped and scaled in uv
    s = s.replace(bsdf("CeilingBSDF", "0.725, 0.71, 0.68"),
                  '\t<bsdf type="twosided" id="CeilingBSDF"><bsdf type="bumpmap"><float name="scale" value="0.02" /><texture type="bitmap" name="height"><string name="filename" value="tex_gray.png" />'
                  '<boolean name="raw" value="true" /><transform name="to_uv"><scale x="2" y="3" /></transform></texture><bsdf type="diffuse"><rgb name="reflectance" value="0.725, 0.71, 0.68" /></bsdf></bsdf></bsdf>\n')
    s = s.replace(bsdf("RightWallBSDF", "0.14, 0.45, 0.091"),
                  '\t<bsdf type="twosided" id="RightWallBSDF"><bsdf type="bumpmap"><bsdf type="plastic"><rgb name="diffuse_reflectance" value="0.14, 0.45, 0.091" /></bsdf><float name="scale" value="0.05" />'
                  '<texture type="bitmap"><string name="filename" value="tex_rgb.png" /><string name="wrap_mode" value="mirror" /><transform name="to_uv"><scale x="1.5" y="0.7" /></transform></texture></bsdf></bsdf>\n')
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + LIGHT + AREA_LIGHT + "</scene>\n"


def cornell_textured_specular(res=128, spp=16):
    """cornell_boxes.xml with textures on the OTHER slots (SURVEY 8(f)-3 leftovers): a roughconductor back wall whose roughness `alpha` is a gray bitmap
    (Texture::eval_1) and whose `specular_reflectance` is a checkerboard; a smooth-plastic short box with an RGB bitmap on `specular_reflectance`
    (the sampling weight then uses the texture's mean) beside a checkerboard `diffuse_reflectance`; a roughdielectric tall box with a bitmap on
    `specular_transmittance`, a checkerboard on `alpha_u` and an RGB bitmap on `alpha_v` (luminance); a conductor floor with a textured
    `specular_reflectance`"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        if b[0] not in ("FloorBSDF", "BackWallBSDF", "ShortBoxBSDF", "TallBoxBSDF"):
            s += bsdf(*b)
    gray = ('\t\t\t<texture type="bitmap" name="%s">\n\t\t\t\t<string name="filename" value="tex_gray.png" />\n\t\t\t\t<boolean name="raw" value="true" />\n'
            '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="%s" y="%s" />\n\t\t\t\t</transform>\n\t\t\t</texture>\n')
    rgbt = ('\t\t\t<texture type="bitmap" name="%s">\n\t\t\t\t<string name="filename" value="$texfile" />\n%s'
            '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="%s" y="%s" />\n\t\t\t\t</transform>\n\t\t\t</texture>\n')
    check = ('\t\t\t<texture type="checkerboard" name="%s">\n\t\t\t\t<rgb name="color0" value="%s" />\n\t\t\t\t<rgb name="color1" value="%s" />\n'
             '\t\t\t\t<transform name="to_uv">\n\t\t\t\t\t<scale x="%s" y="%s" />\n\t\t\t\t</transform>\n\t\t\t</texture>\n')
    s += ('\t<bsdf type="twosided" id="BackWallBSDF">\n\t\t<bsdf type="roughconductor">\n\t\t\t<string name="distribution" value="$distribution" />\n'
          '\t\t\t<rgb name="eta" value="0.2, 0.92, 1.1" />\n\t\t\t<rgb name="k" value="3.9, 2.45, 2.14" />\n'
          + gray % ("alpha", "1.3", "1.1") + check % ("specular_reflectance", "0.9, 0.85, 0.6", "0.3, 0.5, 0.9", "3", "2") + '\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="twosided" id="FloorBSDF">\n\t\t<bsdf type="conductor">\n\t\t\t<rgb name="eta" value="0.2, 0.92, 1.1" />\n\t\t\t<rgb name="k" value="3.9, 2.45, 2.14" />\n'
          + rgbt % ("specular_reflectance", "", "2", "2") + '\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="twosided" id="ShortBoxBSDF">\n\t\t<bsdf type="plastic">\n\t\t\t<float name="int_ior" value="1.7" />\n'
          + rgbt % ("specular_reflectance", '\t\t\t\t<string name="wrap_mode" value="mirror" />\n', "1.5", "1.5")
          + check % ("diffuse_reflectance", "0.1, 0.27, 0.36", "0.6, 0.5, 0.1", "3", "3") + '\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="roughdielectric" id="TallBoxBSDF">\n\t\t<string name="distribution" value="$distribution" />\n\t\t<float name="int_ior" value="1.5" />\n'
          + rgbt % ("specular_transmittance", "", "1", "2")
          + check % ("alpha_u", "0.05, 0.05, 0.05", "0.3, 0.4, 0.2", "2", "4") + rgbt % ("alpha_v", '\t\t\t\t<string name="filter_type" value="nearest" />\n', "0.5", "0.5") + '\t</bsdf>\n')
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + LIGHT + "</scene>\n"


def cornell_env(res=128, spp=16):
    """the Cornell room without ceiling and back wall, under a `constant` environment emitter (src/emitters/constant.cpp) beside the
    point light: rays leave the scene (environment term with MIS, valid_ray) and the environment is sampled as an emitter"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        if name not in ("Ceiling", "BackWall"):
            s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    s += '\t<emitter type="constant">\n\t\t<rgb name="radiance" value="0.8, 0.9, 1.2" />\n\t</emitter>\n'
    return s + LIGHT + "</scene>\n"


def cylinder(ident, bsdf_id, p0, p1, radius, anim_dz=None, extra=""):
    s = ('\t<shape type="cylinder" id="%s">\n\t\t<point name="p0" x="%s" y="%s" z="%s" />\n\t\t<point name="p1" x="%s" y="%s" z="%s" />\n'
         '\t\t<float name="radius" value="%s" />\n%s' % ((ident,) + tuple(p0) + tuple(p1) + (radius, extra)))
    if anim_dz is not None:
        s += ('\t\t<animation name="to_world">\n\t\t\t<transform time="0">\n\t\t\t\t<translate x="0" y="0" z="0" />\n\t\t\t</transform>\n'
              '\t\t\t<transform time="0.0015">\n\t\t\t\t<translate x="0.0" y="0.0" z="%s" />\n\t\t\t</transform>\n\t\t</animation>\n' % anim_dz)
    return s + '\t\t<ref id="%s" />\n\t</shape>\n' % bsdf_id


def cornell_cylinders(res=128, spp=16):
    """the Cornell room with analytic cylinders (src/shapes/cylinder.cpp): an upright pillar, a tilted moving pipe (seen from outside and,
    through its open ends, from inside) and a scaled, rotated one given by a to_world transform"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cylinder("Pillar", "TallBoxBSDF", ("-0.45", "0", "-0.35"), ("-0.45", "1.3", "-0.35"), "0.28")
    s += cylinder("Pipe", "ShortBoxBSDF", ("0.15", "0.25", "0.5"), ("0.75", "0.6", "-0.1"), "0.22", anim_dz="0.015")
    s += ('\t<shape type="cylinder" id="Squat">\n\t\t<transform name="to_world">\n\t\t\t<scale x="0.3" y="0.3" z="0.25" />\n\t\t\t<rotate x="1" angle="-70" />\n'
          '\t\t\t<translate x="0.1" y="1.2" z="0.2" />\n\t\t</transform>\n\t\t<boolean name="flip_normals" value="true" />\n\t\t<ref id="LeftWallBSDF" />\n\t</shape>\n')
    return s + LIGHT + "</scene>\n"


def cornell_plastic(res=128, spp=16):
    """cornell_boxes.xml with glossy-coated (smooth `plastic`) boxes and a plastic floor, point light at the camera"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    s += PLASTIC
    for name, m, b in WALLS:
        s += rect(name, m, "PlasticBSDF" if name == "Floor" else b)
    s += cube("ShortBox", SHORT, "PlasticBSDF", "0.015") + cube("TallBox", TALL, "PlasticBSDF", "-0.015")
    return s + LIGHT + "</scene>\n"


ROUGH = ('\t<bsdf type="twosided" id="RoughCopperBSDF">\n\t\t<bsdf type="roughconductor">\n\t\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n'
         '\t\t\t<float name="alpha" value="0.2" />\n\t\t\t<rgb name="eta" value="0.2, 0.92, 1.1" />\n\t\t\t<rgb name="k" value="3.9, 2.45, 2.14" />\n'
         '\t\t</bsdf>\n\t</bsdf>\n'
         '\t<bsdf type="twosided" id="BrushedBSDF">\n\t\t<bsdf type="roughconductor">\n\t\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n'
         '\t\t\t<float name="alpha_u" value="0.05" />\n\t\t\t<float name="alpha_v" value="0.3" />\n\t\t\t<rgb name="eta" value="1.5, 1.5, 1.5" />\n'
         '\t\t\t<rgb name="k" value="7.6, 6.3, 5.4" />\n\t\t\t<rgb name="specular_reflectance" value="0.9, 0.9, 0.95" />\n\t\t</bsdf>\n\t</bsdf>\n')


ROUGHPLASTIC = ('\t<bsdf type="twosided" id="GlossyPaintBSDF">\n\t\t<bsdf type="roughplastic">\n\t\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n'
                '\t\t\t<float name="alpha" value="0.15" />\n\t\t\t<rgb name="diffuse_reflectance" value="0.1, 0.27, 0.36" />\n'
                '\t\t\t<float name="int_ior" value="1.9" />\n\t\t</bsdf>\n\t</bsdf>\n'
                '\t<bsdf type="twosided" id="SatinFloorBSDF">\n\t\t<bsdf type="roughplastic">\n\t\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n'
                '\t\t\t<float name="alpha" value="0.35" />\n\t\t\t<rgb name="diffuse_reflectance" value="0.6, 0.55, 0.5" />\n'
                '\t\t\t<rgb name="specular_reflectance" value="0.9, 0.85, 0.8" />\n\t\t\t<boolean name="nonlinear" value="true" />\n\t\t</bsdf>\n\t</bsdf>\n')


def cornell_roughplastic(res=128, spp=16):
    """cornell_boxes.xml with rough-plastic (GGX coating over a diffuse base) boxes and floor under the ceiling area light"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    s += ROUGHPLASTIC
    for name, m, b in WALLS:
        s += rect(name, m, "SatinFloorBSDF" if name == "Floor" else b)
    s += cube("ShortBox", SHORT, "GlossyPaintBSDF", "0.015") + cube("TallBox", TALL, "GlossyPaintBSDF", "-0.015")
    return s + AREA_LIGHT + "</scene>\n"


def cornell_rough(res=128, spp=16):
    """cornell_boxes.xml with rough-copper (GGX) boxes and a brushed-metal (anisotropic GGX) floor under the ceiling area light:
    glossy lobes get next-event estimation AND emitter hits, i.e. both directions of the MIS"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    s += ROUGH
    for name, m, b in WALLS:
        s += rect(name, m, "BrushedBSDF" if name == "Floor" else b)
    s += cube("ShortBox", SHORT, "RoughCopperBSDF", "0.015") + cube("TallBox", TALL, "RoughCopperBSDF", "-0.015")
    return s + AREA_LIGHT + "</scene>\n"


def cornell_specular(res=128, spp=16, area_light=True):
    """the Cornell room with a copper-like mirror box (moving), a glass sphere (static) and a mirror back wall section; lit by the
    ceiling area light (so that specular chains reach an emitter: delta lobes get no next-event estimation)"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    s += MIRROR + GLASS
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("MirrorBox", TALL, "MirrorBSDF", "-0.015")
    s += sphere("GlassBall", "GlassBSDF", ("0.4", "0.35", "0.3"), "0.35")
    return s + (AREA_LIGHT if area_light else LIGHT) + "</scene>\n"


FROSTED = ('\t<bsdf type="roughdielectric" id="FrostedBSDF">\n\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n\t\t<float name="alpha" value="0.15" />\n'
           '\t\t<float name="int_ior" value="1.5" />\n\t\t<string name="ext_ior" value="air" />\n\t</bsdf>\n'
           '\t<bsdf type="roughdielectric" id="BrushedGlassBSDF">\n\t\t<string name="distribution" value="$distribution" /><boolean name="sample_visible" value="$sample_visible" />\n\t\t<float name="alpha_u" value="0.05" />\n'
           '\t\t<float name="alpha_v" value="0.3" />\n\t\t<string name="int_ior" value="diamond" />\n\t\t<rgb name="specular_reflectance" value="0.9, 0.95, 1.0" />\n'
           '\t\t<rgb name="specular_transmittance" value="0.95, 0.9, 0.85" />\n\t</bsdf>\n')


def cornell_frosted(res=128, spp=16):
    """the Cornell room with a frosted-glass ball (static), a moving box of anisotropically brushed diamond-index glass, under the ceiling
    area light: rough transmission takes part in next-event estimation and the MIS (a glossy lobe), and eta changes along the paths"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    s += FROSTED
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("BrushedBox", TALL, "BrushedGlassBSDF", "-0.015")
    s += sphere("FrostedBall", "FrostedBSDF", ("0.4", "0.35", "0.3"), "0.35")
    return s + AREA_LIGHT + "</scene>\n"


SPOT = ('\t<emitter type="spot">\n\t\t<transform name="to_world">\n\t\t\t<lookat origin="0.3, 1.9, 0.4" target="-0.2, 0.0, -0.3" up="0, 0, 1" />\n\t\t</transform>\n'
        '\t\t<rgb name="intensity" value="60, 55, 45" />\n\t\t<float name="cutoff_angle" value="35" />\n\t\t<float name="beam_width" value="20" />\n\t</emitter>\n')


def cornell_spot(res=128, spp=16):
    """cornell_boxes.xml lit by a spot light under the ceiling (35 degree cone, smooth falloff from 20 degrees) and a weak point light
    at the camera: two delta emitters of different kinds"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += cube("ShortBox", SHORT, "ShortBoxBSDF", "0.015") + cube("TallBox", TALL, "TallBoxBSDF", "-0.015")
    return s + SPOT + LIGHT.replace('value="100"', 'value="10"') + "</scene>\n"


DISK_LIGHT = ('\t<shape type="disk" id="DiskLight">\n\t\t<transform name="to_world">\n\t\t\t<scale x="0.3" y="0.2" z="1" />\n'
              '\t\t\t<rotate x="1" angle="90" />\n\t\t\t<translate x="0" y="1.98" z="0" />\n\t\t</transform>\n'
              '\t\t<emitter type="area">\n\t\t\t<rgb name="radiance" value="17, 12, 4" />\n\t\t</emitter>\n\t</shape>\n')


def cornell_disk(res=128, spp=16):
    """the Cornell room under an elliptic disk light, with a tilted two-sided disk that sweeps through the room (animated) and a static
    disk with flipped normals leaning against the back wall"""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + SENSOR.format(fov="19.5", cam=CAM)
    for b in BSDFS:
        s += bsdf(*b)
    for name, m, b in WALLS:
        s += rect(name, m, b)
    s += ('\t<shape type="disk" id="MovingDisk">\n\t\t<animation name="to_world">\n'
          '\t\t\t<transform time="0">\n\t\t\t\t<scale value="0.4" />\n\t\t\t\t<rotate x="1" angle="-60" />\n\t\t\t\t<translate x="-0.35" y="0.6" z="0.1" />\n\t\t\t</transform>\n'
          '\t\t\t<transform time="0.0015">\n\t\t\t\t<scale value="0.4" />\n\t\t\t\t<rotate x="1" angle="-58" />\n\t\t\t\t<translate x="-0.35" y="0.6" z="0.115" />\n\t\t\t</transform>\n'
          '\t\t</animation>\n\t\t<ref id="ShortBoxBSDF" />\n\t</shape>\n')
    s += ('\t<shape type="disk" id="LeaningDisk">\n\t\t<boolean name="flip_normals" value="true" />\n\t\t<transform name="to_world">\n'
          '\t\t\t<scale x="0.35" y="0.5" z="1" />\n\t\t\t<rotate y="1" angle="160" />\n\t\t\t<translate x="0.45" y="0.55" z="-0.6" />\n\t\t</transform>\n'
          '\t\t<ref id="TallBoxBSDF" />\n\t</shape>\n')
    return s + DISK_LIGHT + "</scene>\n"


def domino(n_side=32, res=1024, spp=128):
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5").replace('value="sinusoidal"', 'value="rectangular"')
    cam = '\t\t\t<lookat origin="0, 9, 16" target="0, 0.3, 0" up="0, 1, 0" />'
    s += SENSOR.format(fov="40", cam=cam)
    s += bsdf("GroundBSDF", "0.6, 0.6, 0.6") + bsdf("DominoBSDF", "0.75, 0.55, 0.35")
    s += ('\t<shape type="rectangle" id="Ground">\n\t\t<transform name="to_world">\n\t\t\t<rotate x="1" angle="-90" />\n'
          '\t\t\t<scale value="10" />\n\t\t</transform>\n\t\t<ref id="GroundBSDF" />\n\t</shape>\n')
    s += ('\t<shape type="shapegroup" id="DominoGroup">\n\t\t<shape type="cube">\n\t\t\t<transform name="to_world">\n'
          '\t\t\t\t<scale x="0.05" y="0.5" z="0.25" />\n\t\t\t\t<translate y="0.5" />\n\t\t\t</transform>\n'
          '\t\t\t<ref id="DominoBSDF" />\n\t\t</shape>\n\t</shape>\n')
    lcg = 1234
    for k in range(n_side * n_side):
        lcg = (lcg * 1664525 + 1013904223) & 0xffffffff
        jitter = ((lcg >> 8) & 0xffff) / 65536.0 - 0.5
        ix, iz = k % n_side, k // n_side
        x = (ix - (n_side - 1) / 2.0) * 0.6 + 0.1 * jitter
        z = (iz - (n_side - 1) / 2.0) * 0.6
        theta = math.degrees(0.02 * (1.0 + math.sin(0.1 * k)))
        yaw = 10.0 * jitter
        base = '\t\t\t\t<rotate y="1" angle="%.6f" />\n\t\t\t\t<translate x="%.6f" y="0" z="%.6f" />\n' % (yaw, x, z)
        s += ('\t<shape type="instance">\n\t\t<ref id="DominoGroup" />\n\t\t<animation name="to_world">\n'
              '\t\t\t<transform time="0">\n' + base + '\t\t\t</transform>\n\t\t\t<transform time="0.0015">\n'
              '\t\t\t\t<rotate z="1" angle="%.6f" />\n' % (-theta) + base +
              '\t\t\t\t<translate x="%.6f" y="0" z="0" />\n' % (0.004 * (1.0 + math.sin(0.1 * k))) +
              '\t\t\t</transform>\n\t\t</animation>\n\t</shape>\n')
    s += ('\t<emitter type="point">\n\t\t<point name="position" x="0" y="9" z="16" />\n'
          '\t\t<rgb name="intensity" value="400" />\n\t</emitter>\n</scene>\n')
    return s


def open_veils(env=False, res=128, spp=16):
    """An OPEN scene for the integrators' valid_ray (dopplertofpath.cpp:101-102,252-253,279-282): free-standing cards in front of the void above a floor strip --
    a moving `mask` veil of constant opacity, a second one behind it (two null interactions in a row, or a null one followed by a real one), a two-sided plastic
    card behind a checkerboard opacity, a `thindielectric` pane (its transmission is BSDFFlags::Null too, thindielectric.cpp:179) and an opaque diffuse card.
    A path whose every sampled lobe was a null one and which then leaves the scene returns 0 -- including the emitter samples it gathered at the veils -- unless the
    environment is visible (`env`: a constant environment; with hide_emitters = true it lights the cards but must not show through the cut-outs)."""
    s = HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5")
    s = s.replace('\t<integrator type="dopplertofpath">\n', '\t<default name="hide_emitters" value="false" />\n\t<default name="pixel_format" value="rgb" />\n'
                  '\t<integrator type="dopplertofpath">\n\t\t<boolean name="hide_emitters" value="$hide_emitters" />\n')
    s += SENSOR.format(fov="30", cam='\t\t\t<lookat origin="0, 1, 6" target="0, 0.9, 0" up="0, 1, 0" />').replace('value="rgb"', 'value="$pixel_format"')
    s += bsdf("FloorBSDF", "0.7, 0.7, 0.65")
    s += ('\t<bsdf type="mask" id="VeilBSDF">\n\t\t<float name="opacity" value="0.5" />\n\t\t<bsdf type="twosided">\n\t\t\t<bsdf type="diffuse">\n'
          '\t\t\t\t<rgb name="reflectance" value="0.7, 0.3, 0.2" />\n\t\t\t</bsdf>\n\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="mask" id="BackVeilBSDF">\n\t\t<float name="opacity" value="0.35" />\n\t\t<bsdf type="diffuse">\n'
          '\t\t\t<rgb name="reflectance" value="0.2, 0.6, 0.8" />\n\t\t</bsdf>\n\t</bsdf>\n')
    s += ('\t<bsdf type="mask" id="CheckerBSDF">\n\t\t<texture type="checkerboard" name="opacity">\n\t\t\t<rgb name="color0" value="0.0" />\n\t\t\t<rgb name="color1" value="1.0" />\n'
          '\t\t\t<transform name="to_uv">\n\t\t\t\t<scale x="3" y="3" />\n\t\t\t</transform>\n\t\t</texture>\n'
          '\t\t<bsdf type="twosided">\n\t\t\t<bsdf type="plastic">\n\t\t\t\t<rgb name="diffuse_reflectance" value="0.3, 0.7, 0.3" />\n\t\t\t</bsdf>\n\t\t</bsdf>\n\t</bsdf>\n')
    s += '\t<bsdf type="thindielectric" id="PaneBSDF">\n\t\t<float name="int_ior" value="1.5" />\n\t</bsdf>\n'
    s += bsdf("CardBSDF", "0.6, 0.6, 0.2")

    def card(ident, b, sx, sy, tx, ty, tz, move=None):
        base = '\t\t\t\t<scale x="%s" y="%s" z="1" />\n\t\t\t\t<translate x="%s" y="%s" z="%s" />\n' % (sx, sy, tx, ty, tz)
        if move is None:
            return '\t<shape type="rectangle" id="%s">\n\t\t<ref id="%s" />\n\t\t<transform name="to_world">\n%s\t\t</transform>\n\t</shape>\n' % (ident, b, base.replace("\t\t\t\t", "\t\t\t"))
        return ('\t<shape type="rectangle" id="%s">\n\t\t<ref id="%s" />\n\t\t<animation name="to_world">\n\t\t\t<transform time="0">\n%s\t\t\t</transform>\n'
                '\t\t\t<transform time="0.0015">\n%s\t\t\t\t<translate x="0" y="0" z="%s" />\n\t\t\t</transform>\n\t\t</animation>\n\t</shape>\n' % (ident, b, base, base, move))
    s += ('\t<shape type="rectangle" id="Floor">\n\t\t<ref id="FloorBSDF" />\n\t\t<transform name="to_world">\n\t\t\t<scale x="1.6" y="0.9" z="1" />\n'
          '\t\t\t<rotate x="1" angle="-90" />\n\t\t\t<translate x="0" y="0" z="0.3" />\n\t\t</transform>\n\t</shape>\n')
    s += card("Veil", "VeilBSDF", "0.55", "0.6", "-0.9", "0.8", "0.6", move="0.015")
    s += card("BackVeil", "BackVeilBSDF", "0.7", "0.7", "-0.6", "0.9", "-0.4")
    s += card("Checker", "CheckerBSDF", "0.45", "0.55", "0.25", "0.75", "0.4", move="-0.015")
    s += card("Pane", "PaneBSDF", "0.4", "0.6", "1.1", "0.8", "0.5")
    s += card("Card", "CardBSDF", "0.35", "0.35", "0.9", "1.0", "-0.6")
    s += ('\t<emitter type="point">\n\t\t<point name="position" x="0.3" y="2.2" z="4" />\n\t\t<rgb name="intensity" value="60" />\n\t</emitter>\n')
    if env:
        s += '\t<emitter type="constant">\n\t\t<rgb name="radiance" value="0.5, 0.6, 0.8" />\n\t</emitter>\n'
    return s + "</scene>\n"


def main():
    out = {
        "cornell_boxes.xml": cornell(False, 256, 16, "antithetic", "0.5"),
        "cornell_wall.xml": cornell(True, 512, 64, "stratified", "0.0"),
        "cornell_area.xml": cornell(False, 256, 64, "antithetic", "0.5", area_light=True),
        "cornell_specular.xml": cornell_specular(),
        "cornell_plastic.xml": cornell_plastic(),
        "cornell_rough.xml": cornell_rough(),
        "cornell_roughplastic.xml": cornell_roughplastic(),
        "cornell_frosted.xml": cornell_frosted(),
        "cornell_spot.xml": cornell_spot(),
        "cornell_disk.xml": cornell_disk(),
        "cornell_textured.xml": cornell_textured(),
        "cornell_textured_specular.xml": cornell_textured_specular(),
        "cornell_masked.xml": cornell_masked(),
        "open_veils.xml": open_veils(False),
        "open_veils_env.xml": open_veils(True),
        "cornell_normalmap.xml": cornell_normalmap(),
        "cornell_blend.xml": cornell_blend(),
        "cornell_textured_light.xml": cornell_textured_light(),
        "cornell_env.xml": cornell_env(),
        "cornell_envmap.xml": cornell_envmap(),
        "cornell_sun.xml": cornell_sun(),
        "cornell_thinlens.xml": cornell_thinlens(),
        "cornell_cylinders.xml": cornell_cylinders(),
        "cornell_spheres.xml": cornell_spheres(),
        "cornell_sphere_light.xml": cornell_spheres(sphere_light=True),
        "domino.xml": domino(),
        "domino_small.xml": domino(n_side=6, res=128, spp=16),
    }
    texture_files()
    for name, text in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            f.write(text)
        print("wrote", name, len(text), "bytes")


def ensure(quiet=True):
    """(re)generate scenes/*.xml if any is missing or older than this script -- the files are build products, not tracked"""
    names = ["cornell_boxes.xml", "cornell_wall.xml", "cornell_area.xml", "cornell_specular.xml", "cornell_plastic.xml", "cornell_rough.xml", "cornell_roughplastic.xml", "cornell_frosted.xml", "cornell_spot.xml", "cornell_disk.xml",
             "cornell_spheres.xml", "cornell_sphere_light.xml", "domino.xml", "domino_small.xml", "cornell_textured.xml", "cornell_textured_specular.xml", "cornell_masked.xml", "open_veils.xml", "open_veils_env.xml", "cornell_normalmap.xml", "cornell_blend.xml", "cornell_textured_light.xml", "tex_normal.png", "cornell_env.xml", "cornell_envmap.xml", "cornell_sun.xml", "cornell_thinlens.xml", "cornell_cylinders.xml", "tex_rgb.png", "tex_gray.png", "env_sky.hdr", "env_sky.pfm", "env_sky.png", "env_sky.exr"]
    me = os.path.getmtime(os.path.abspath(__file__))
    if all(os.path.exists(os.path.join(HERE, n)) and os.path.getmtime(os.path.join(HERE, n)) >= me for n in names):
        return
    if quiet:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            main()
    else:
        main()


if __name__ == "__main__":
    main()

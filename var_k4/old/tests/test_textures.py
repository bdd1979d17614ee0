"""Textures on the diffuse reflectances (SURVEY 8(f)-3 leftovers): `checkerboard` and `bitmap` (src/textures/{checkerboard,bitmap}.cpp) on
`reflectance` (diffuse) and `diffuse_reflectance` (plastic, roughplastic).  Loader parity against the oracle's independent loader (which
decodes the PNG files with PIL; the product parses the PNG chunks itself over zlib), lookup semantics, error behaviour; per-lane parity
on the GPU."""
import os

import numpy as np
import pytest

from conftest import SCENES

NCPU = min(16, os.cpu_count() or 1)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_texture_records_and_texels_match_the_oracle_loader(mi, orc):
    path = os.path.join(SCENES, "cornell_textured.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    want = [s["tex_refl"] for s in osc.flat.shapes]
    got = sc.export(15).astype(int)
    assert [g >= 0 for g in got] == [w is not None for w in want] and sum(g >= 0 for g in got) == 5
    rec = sc.export(13).reshape(-1, 17)
    texels = sc.export(14)
    off = 0
    for i, w in enumerate(want):
        if w is None:
            continue
        r = rec[got[i]]
        assert (int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])) == (w["kind"], w["filter"], w["wrap"], w["channels"], w["width"], w["height"])
        assert np.array_equal(bits(r[6:10]), bits(w["to_uv"])) and np.array_equal(bits(r[10:13]), bits(w["color0"])) and np.array_equal(bits(r[13:16]), bits(w["color1"]))
        assert bits(r[16]) == bits(np.float32(w["mean"])), (i, r[16], w["mean"])
    # texels: in order of appearance in the file = order of the texture table
    order = sorted({g for g in got if g >= 0})
    by_index = {got[i]: want[i] for i in range(len(want)) if want[i] is not None}
    for k in order:
        w = by_index[k]
        if w["data"] is not None:
            n = w["data"].size
            assert np.array_equal(bits(texels[off:off + n]), bits(w["data"].reshape(-1))), k
            off += n
    assert off == texels.size
    kinds = {(int(rec[g][0]), int(rec[g][1]), int(rec[g][2])) for g in order}
    assert {(0, 1, 0), (1, 1, 0), (1, 0, 1), (1, 1, 2)} <= kinds          # checkerboard; bilinear+repeat; nearest+mirror; bilinear+clamp
    # the plastic box: specular sampling weight from the texture's own mean (plastic.cpp:201-217)
    pl = [i for i, s in enumerate(osc.flat.shapes) if s["bsdf"] == 3]
    brec = sc.export(9).reshape(-1, 24)
    for i in pl:
        assert bits(brec[i, 6]) == bits(osc.flat.shapes[i]["plastic_params"][2])


def test_texture_lookup_semantics(orc):
    """Checkerboard and bitmap lookups of the oracle: texel centres, wrap modes, the 2x2 to_uv, the gray -> RGB broadcast"""
    import ctypes as C
    L = orc.lib()
    t = orc.OrcTexture()
    t.kind, t.filter, t.wrap, t.channels, t.width, t.height = 1, 0, 0, 1, 4, 2
    data = np.arange(8, dtype=np.float32) / 8
    t.data = data.ctypes.data_as(C.POINTER(C.c_float))
    t.to_uv = (C.c_float * 4)(1, 0, 0, 1)
    out = np.zeros(3, np.float32)

    def ev(u, v):
        L.orc_texture_eval(C.byref(t), C.c_float(u), C.c_float(v), out.ctypes.data)
        return out.copy()
    assert np.array_equal(ev(0.1, 0.1), [0, 0, 0]) and np.array_equal(ev(0.9, 0.9), [7 / 8] * 3)      # nearest: row 0 is v = 0
    assert np.array_equal(ev(1.1, 0.1), ev(0.1, 0.1)) and np.array_equal(ev(-0.1, 0.1), ev(0.9, 0.1))   # repeat
    t.wrap = 1
    assert np.array_equal(ev(1.1, 0.1), ev(0.9, 0.1)) and np.array_equal(ev(-0.1, 0.1), ev(0.1, 0.1))   # mirror
    t.wrap = 2
    assert np.array_equal(ev(1.7, 0.1), ev(0.99, 0.1)) and np.array_equal(ev(-3.0, 0.9), ev(0.0, 0.9))  # clamp
    t.filter = 1
    assert np.allclose(ev(0.125, 0.25), [0, 0, 0]) and np.allclose(ev(0.25, 0.25), [0.5 / 8] * 3)        # texel centres at (i + .5) / res
    assert np.allclose(ev(0.125, 0.5), [2 / 8] * 3)                                                        # halfway between the two rows
    t.kind = 0
    t.color0, t.color1 = (C.c_float * 3)(1, 0, 0), (C.c_float * 3)(0, 0, 1)
    t.to_uv = (C.c_float * 4)(2, 0, 0, 2)
    assert np.array_equal(ev(0.1, 0.1), [1, 0, 0]) and np.array_equal(ev(0.3, 0.1), [0, 0, 1]) and np.array_equal(ev(0.3, 0.3), [1, 0, 0])


def test_to_uv_translation_is_lost_as_in_the_reference(orc, tmp_path):
    """Transform4f::extract() (transform.h:340-360) copies the upper-left block and the bottom row: the translation of a `to_uv`
    transform never reaches the texture (replicated, SURVEY App. B style quirk)"""
    text = open(os.path.join(SCENES, "cornell_textured.xml")).read()
    assert '<translate x="0.25" y="0" />' in text
    a = orc.Scene(os.path.join(SCENES, "cornell_textured.xml"), dict(resx=16, resy=16))
    p = tmp_path / "moved.xml"
    p.write_text(text.replace('<translate x="0.25" y="0" />', '<translate x="0.4" y="0.3" />').replace('value="tex_', 'value="%s/tex_' % SCENES))
    b = orc.Scene(str(p), dict(resx=16, resy=16))
    ia, _ = a.render(a.params(), seed=1, spp=4, threads=NCPU)
    ib, _ = b.render(b.params(), seed=1, spp=4, threads=NCPU)
    assert np.array_equal(ia, ib)


def test_texture_error_behaviour(mi, tmp_path):
    text = open(os.path.join(SCENES, "cornell_textured.xml")).read().replace('value="tex_', 'value="%s/tex_' % SCENES)
    mi.load_string(text)
    with pytest.raises(mi.DtofError, match="Invalid filter type"):
        mi.load_string(text.replace('value="nearest"', 'value="trilinear"'))
    with pytest.raises(mi.DtofError, match="Invalid wrap mode"):
        mi.load_string(text.replace('value="mirror"', 'value="border"'))
    with pytest.raises(mi.DtofError, match="could not open"):
        mi.load_string(text.replace("tex_gray.png", "missing.png"))
    with pytest.raises(mi.DtofError, match="unsupported texture plugin"):
        mi.load_string(text.replace('<texture type="checkerboard" name="reflectance">', '<texture type="mesh_attribute" name="reflectance">'))
    bad = tmp_path / "not.png"
    bad.write_bytes(b"neither a png nor a jpeg")
    with pytest.raises(mi.DtofError, match="is not a PNG file"):
        mi.load_string(text.replace(SCENES + "/tex_gray.png", str(bad)))
    with pytest.raises(mi.DtofError, match="unreferenced property"):
        mi.load_string(text.replace('<string name="wrap_mode" value="clamp" />', '<string name="wrap_mode" value="clamp" /><float name="gamma" value="2.2" />'))


def test_textures_on_other_slots_fail_loudly_and_shared_textures_are_stored_once(mi):
    """A texture bound to a property that takes constants only in this build, or to a misspelt name, must raise (the reference would use it
    or report an unreferenced object, xml.cpp:1204-1215) instead of rendering the default colour; a texture that many shapes reference is
    decoded and stored once (one texture record, one copy of the texels in the blob)."""
    png = os.path.join(SCENES, "tex_rgb.png")
    head = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/>'
            '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>')
    tex = '<texture type="bitmap" name="%%s"><string name="filename" value="%s"/></texture>' % png
    with pytest.raises(mi.DtofError, match="does not accept a texture"):
        mi.load_string(head + '<shape type="rectangle"><bsdf type="conductor">' + tex % "eta" + '</bsdf></shape></scene>')
    with pytest.raises(mi.DtofError, match='unreferenced object "reflectanse"'):
        mi.load_string(head + '<shape type="rectangle"><bsdf type="diffuse">' + tex % "reflectanse" + '</bsdf></shape></scene>')
    with pytest.raises(mi.DtofError, match="does not accept a texture"):
        mi.load_string(head + '<shape type="rectangle"><bsdf type="roughplastic">' + tex % "alpha" + '</bsdf></shape></scene>')
    shared = head + '<bsdf type="diffuse" id="m">' + tex % "reflectance" + '</bsdf>' + ''.join(
        '<shape type="rectangle"><transform name="to_world"><translate x="%d"/></transform><ref id="m"/></shape>' % k for k in range(40)) + '</scene>'
    sc = mi.load_string(shared)
    used = sc.export(15).astype(int)
    assert len(used) == 40 and (used == 0).all()                      # every shape points at texture 0
    assert sc.export(13).reshape(-1, 17).shape[0] == 1                # one record
    from PIL import Image
    w, h = Image.open(png).size
    assert sc.export(14).size == w * h * 3                            # one copy of the texels
    one = mi.load_string(shared.replace(''.join('<shape type="rectangle"><transform name="to_world"><translate x="%d"/></transform><ref id="m"/></shape>' % k for k in range(1, 40)), ''))
    assert sc.info()["scene_blob_bytes"] - one.info()["scene_blob_bytes"] < 40 * 1024   # 39 more rectangles, not 39 more images


def test_textures_on_specular_and_roughness_slots(mi, orc):
    """specular_reflectance / specular_transmittance (Texture::eval per hit) and alpha / alpha_u / alpha_v of roughconductor / roughdielectric
    (Texture::eval_1 per hit): both loaders bind the same textures to the same slots, the constants they keep are the textures' means (what the
    plastics' specular sampling weight uses, plastic.cpp:201-217), and eval_1 follows bitmap.cpp:324-344 / checkerboard.cpp:91-110."""
    import ctypes as C
    path = os.path.join(SCENES, "cornell_textured_specular.xml")
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    slots = sc.export(19).reshape(-1, 4).astype(int)
    rec = sc.export(13).reshape(-1, 17)
    keys = ("tex_spec", "tex_trans", "tex_alpha_u", "tex_alpha_v")
    bound = 0
    for i, s in enumerate(osc.flat.shapes):
        for j, k in enumerate(keys):
            w = s.get(k)
            assert (slots[i, j] >= 0) == (w is not None), (i, k)
            if w is None:
                continue
            bound += 1
            r = rec[slots[i, j]]
            assert (int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])) == (w["kind"], w["filter"], w["wrap"], w["channels"], w["width"], w["height"])
            assert bits(r[16]) == bits(np.float32(w["mean"]))
    assert bound == 8 and slots[2, 2] == slots[2, 3]            # `alpha` fills both roughness slots with ONE texture
    b = sc.export(9).reshape(-1, 24)
    for i, s in enumerate(osc.flat.shapes):                     # constants = means; derived sampling weights bit-identical
        assert np.array_equal(bits(b[i, 10:13]), bits(s["spec_refl"])) and np.array_equal(bits(b[i, 13:16]), bits(s["spec_trans"]))
        assert bits(b[i, 22]) == bits(np.float32(s["alpha_u"])) and bits(b[i, 23]) == bits(np.float32(s["alpha_v"]))
        if s["bsdf"] == 3:
            assert bits(b[i, 6]) == bits(s["plastic_params"][2])
    # eval_1: gray bitmap -> the texel; RGB bitmap -> luminance; checkerboard -> the mean of the colour the lookup picks
    L = orc.lib()
    L.orc_texture_eval_1.restype = C.c_float
    t = orc.OrcTexture()
    t.kind, t.filter, t.wrap, t.channels, t.width, t.height = 1, 0, 0, 3, 2, 2
    data = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9], [1.0, 0.0, 0.5]], np.float32)
    t.data = data.ctypes.data_as(C.POINTER(C.c_float)); t.to_uv = (C.c_float * 4)(1, 0, 0, 1)
    lum = lambda c: np.float32(np.float32(np.float32(c[0] * np.float32(0.212671)) + np.float32(c[1] * np.float32(0.715160))) + np.float32(c[2] * np.float32(0.072169)))
    assert bits(np.float32(L.orc_texture_eval_1(C.byref(t), C.c_float(0.25), C.c_float(0.25)))) == bits(lum(data[0]))
    assert bits(np.float32(L.orc_texture_eval_1(C.byref(t), C.c_float(0.75), C.c_float(0.75)))) == bits(lum(data[3]))
    t.channels = 1
    assert np.float32(L.orc_texture_eval_1(C.byref(t), C.c_float(0.75), C.c_float(0.25))) == data.reshape(-1)[1]
    t.kind = 0; t.color0, t.color1 = (C.c_float * 3)(0.3, 0.6, 0.9), (C.c_float * 3)(0.0, 0.3, 0.0)
    assert abs(L.orc_texture_eval_1(C.byref(t), C.c_float(0.1), C.c_float(0.1)) - 0.6) < 1e-6 and abs(L.orc_texture_eval_1(C.byref(t), C.c_float(0.7), C.c_float(0.1)) - 0.1) < 1e-6
    with pytest.raises(mi.DtofError, match="does not accept a texture"):
        mi.load_string(open(path).read().replace('<rgb name="eta" value="0.2, 0.92, 1.1" />', '', 1).replace('name="alpha"', 'name="eta"', 1).replace("tex_", SCENES + "/tex_"))


def test_png_reader_handles_filters_palettes_and_alpha(mi, tmp_path):
    """the product's PNG reader against PIL: every scanline filter type (PIL picks them adaptively on a noisy image), RGBA, gray + alpha, palette"""
    from PIL import Image
    rng = np.random.default_rng(5)
    base = (rng.random((9, 13, 4)) * 255).astype(np.uint8)
    base[:, :, 0] = np.linspace(0, 255, 13).astype(np.uint8)[None, :]       # smooth channels make Sub / Up / Average / Paeth win somewhere
    base[:, :, 1] = np.linspace(0, 255, 9).astype(np.uint8)[:, None]
    variants = {"rgb": Image.fromarray(base[..., :3], "RGB"), "rgba": Image.fromarray(base, "RGBA"), "gray": Image.fromarray(base[..., 0], "L"),
                "la": Image.fromarray(base[..., :2], "LA"), "pal": Image.fromarray(base[..., :3], "RGB").quantize(16)}
    for name, im in variants.items():
        path = str(tmp_path / (name + ".png"))
        im.save(path, optimize=(name == "rgb"))
        xml = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/>'
               '<film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/></film></sensor>'
               '<shape type="rectangle"><bsdf type="diffuse"><texture type="bitmap" name="reflectance"><string name="filename" value="%s"/>'
               '<boolean name="raw" value="true"/></texture></bsdf></shape></scene>' % path)
        sc = mi.load_string(xml)
        rec = sc.export(13).reshape(-1, 17)[0]
        ref = np.asarray(im.convert("L" if name in ("gray", "la") else "RGB"), np.uint8)
        assert (int(rec[3]), int(rec[4]), int(rec[5])) == (1 if name in ("gray", "la") else 3, 13, 9)
        got = np.rint(sc.export(14) * 255).astype(np.uint8).reshape(ref.shape)
        assert np.array_equal(got, ref), name


def test_jpeg_reader_matches_libjpeg_byte_for_byte(mi, tmp_path):
    """the product's baseline JPEG decoder (image_io.cpp: Huffman decoding, IJG's slow-but-accurate integer IDCT, fancy chroma upsampling,
    fixed-point YCbCr -> RGB) against PIL / libjpeg-turbo with its default settings -- what the reference's Bitmap reads through libjpeg:
    4:4:4, 4:2:2, 4:2:0 and grayscale files, several qualities, sizes that are no multiple of the MCU (down to the widths at which libjpeg stops filtering the chroma), restart intervals, optimised tables;
    progressive files are refused."""
    from PIL import Image
    rng = np.random.default_rng(9)
    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 5.0 + y / 9.0), 127 + 120 * np.cos(x / 3.0) * np.sin(y / 4.0), (x * 7 + y * 13) % 256], -1)
        img += rng.normal(0, 12, img.shape)          # texture: exercises long runs of AC coefficients
        img[h // 3: h // 2, w // 4: w // 2] = (250, 10, 30)   # saturated patch: range limiting and chroma edges
        return np.clip(img, 0, 255).astype(np.uint8)
    cases = []
    for (w, h) in ((16, 16), (37, 29), (8, 5), (64, 48), (3, 2), (17, 1 + 16)):
        for sub, quality in ((0, 92), (1, 75), (2, 60), (2, 98), (0, 30)):
            cases.append((w, h, "RGB", dict(quality=quality, subsampling=sub)))
        cases.append((w, h, "L", dict(quality=85)))
    cases.append((40, 33, "RGB", dict(quality=80, subsampling=2, optimize=True)))
    cases.append((45, 70, "RGB", dict(quality=70, subsampling=1, optimize=True)))
    cases.append((56, 40, "RGB", dict(quality=80, subsampling=2, restart_marker_blocks=2)))      # DRI + RSTn markers: predictors reset, bytes realigned
    cases.append((56, 40, "RGB", dict(quality=80, subsampling=0, restart_marker_rows=1)))
    cases.append((23, 31, "L", dict(quality=60, restart_marker_blocks=1)))
    xml = ('<scene version="3.0.0"><integrator type="path"/><sensor type="perspective"><float name="fov" value="40"/>'
           '<film type="hdrfilm"><integer name="width" value="4"/><integer name="height" value="4"/></film></sensor>'
           '<shape type="rectangle"><bsdf type="diffuse"><texture type="bitmap" name="reflectance"><string name="filename" value="%s"/>'
           '<boolean name="raw" value="true"/></texture></bsdf></shape></scene>')
    for k, (w, h, mode, opts) in enumerate(cases):
        if w < 2 or h < 2:
            continue
        src = picture(w, h)
        im = Image.fromarray(src if mode == "RGB" else src[..., 0], mode)
        path = str(tmp_path / ("j%d.jpg" % k))
        im.save(path, "JPEG", **opts)
        ref = np.asarray(Image.open(path).convert(mode), np.uint8)
        sc = mi.load_string(xml % path)
        got = np.rint(sc.export(14) * 255).astype(np.uint8).reshape(ref.shape)
        assert np.array_equal(got, ref), (k, w, h, mode, opts, int(np.abs(got.astype(int) - ref.astype(int)).max()), int((got != ref).sum()))
    prog = str(tmp_path / "prog.jpg")
    Image.fromarray(picture(24, 24), "RGB").save(prog, "JPEG", progressive=True)
    with pytest.raises(mi.DtofError, match="progressive"):
        mi.load_string(xml % prog)
    trunc = str(tmp_path / "trunc.jpg")
    data = open(str(tmp_path / "j0.jpg"), "rb").read()
    open(trunc, "wb").write(data[:len(data) // 2])
    try:
        mi.load_string(xml % trunc)      # a truncated scan decodes as far as it goes (zeros beyond), like libjpeg does, or is refused -- it must not crash
    except mi.DtofError:
        pass


def test_jpeg_texture_texels_match_the_oracle_loader(mi, orc):
    """cornell_textured.xml with its bitmap from a 4:2:0 JPEG file: the linear texels the product decodes (own decoder + sRGB LUT) equal the
    oracle loader's (PIL + the same LUT) bit for bit"""
    path = os.path.join(SCENES, "cornell_textured.xml")
    sc, osc = mi.load_file(path, texfile="tex_rgb.jpg"), orc.Scene(path, dict(texfile="tex_rgb.jpg"))
    theirs = np.concatenate([np.asarray(s["tex_refl"]["data"], np.float32).ravel() for s in osc.flat.shapes
                             if s.get("tex_refl") is not None and s["tex_refl"].get("data") is not None])
    ours = sc.export(14)
    assert ours.size == theirs.size == 2 * 64 * 48 * 3 + (ours.size - 2 * 64 * 48 * 3) and np.array_equal(bits(ours), bits(theirs))


@pytest.mark.gpu
@pytest.mark.parametrize("feature", ["texture", "spot_light", "rough_metal", "glass_and_envmap"])
@pytest.mark.parametrize("pipeline", ["auto", "split", "fused"])
def test_textured_rectangle_only_scene(mi, orc, pipeline, feature, monkeypatch):
    """a scene of rectangles alone with a bitmap texture: the split pipeline's trace kernels must hand the full hit record (u, v) to the textured
    shade kernels (the 4-byte record of the plain rectangle-only kernels carries the distance only); every pipeline gives the oracle's lanes"""
    import tempfile
    from scenes import make_scenes as ms
    d = tempfile.mkdtemp()
    ms.write_png(os.path.join(d, "t.png"), [[((x * 37) % 256, (y * 91) % 256, (x * y * 5) % 256) for x in range(8)] for y in range(8)])
    xml = ('<scene version="3.0.0"><integrator type="dopplertofpath"><integer name="max_depth" value="6"/></integrator>'
           '<sensor type="perspective"><float name="fov" value="35"/><transform name="to_world"><lookat origin="0, 1, 5" target="0, 1, 0" up="0, 1, 0"/></transform>'
           '<sampler type="correlated"><integer name="sample_count" value="4"/></sampler>'
           '<film type="hdrfilm"><integer name="width" value="12"/><integer name="height" value="12"/><rfilter type="tent"/></film><float name="shutter_close" value="0.0015"/></sensor>'
           '<bsdf type="twosided" id="tex"><bsdf type="diffuse"><texture type="bitmap" name="reflectance"><string name="filename" value="%s"/></texture></bsdf></bsdf>'
           '<shape type="rectangle"><transform name="to_world"><rotate x="1" angle="-90"/><scale value="2"/></transform><bsdf type="diffuse"><rgb name="reflectance" value="0.7"/></bsdf></shape>'
           '<shape type="rectangle"><transform name="to_world"><scale value="2"/><translate z="-2" y="1"/></transform><ref id="tex"/></shape>'
           '<emitter type="point"><point name="position" x="0" y="1.8" z="1"/><rgb name="intensity" value="20"/></emitter></scene>' % os.path.join(d, "t.png"))
    # the other features that select the SPEC shade kernels, on the same rectangles
    plain = '<bsdf type="twosided" id="tex"><bsdf type="diffuse"><rgb name="reflectance" value="0.3, 0.6, 0.2"/></bsdf></bsdf>'
    textured = xml[xml.index('<bsdf type="twosided" id="tex">'):xml.index('<shape type="rectangle">')]
    if feature == "spot_light":
        xml = xml.replace(textured, plain).replace('<emitter type="point"><point name="position" x="0" y="1.8" z="1"/><rgb name="intensity" value="20"/></emitter>',
                                                   '<emitter type="spot"><transform name="to_world"><lookat origin="0, 1.8, 1" target="0, 0.5, -1.5" up="0, 1, 0"/></transform>'
                                                   '<rgb name="intensity" value="40"/><float name="cutoff_angle" value="40"/></emitter>')
    elif feature == "rough_metal":
        xml = xml.replace(textured, '<bsdf type="twosided" id="tex"><bsdf type="roughconductor"><rgb name="eta" value="0.2, 0.92, 1.1"/><rgb name="k" value="3.9, 2.45, 2.14"/><float name="alpha" value="0.2"/></bsdf></bsdf>')
    elif feature == "glass_and_envmap":
        xml = xml.replace(textured, '<bsdf type="dielectric" id="tex"/>').replace("</scene>", '<emitter type="constant"><rgb name="radiance" value="0.4, 0.5, 0.7"/></emitter></scene>')
    if pipeline != "auto":
        monkeypatch.setenv("DTOF_PIPELINE", pipeline)
    sc, osc = mi.load_string(xml), orc.Scene(xml, is_string=True)
    n = 12 * 12 * 4
    ours, ref = sc.sample_lanes(2, 4, 0, n), osc.render_lanes(osc.params(), 2, 4, 0, n, threads=4)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(np.ascontiguousarray(ours[k]).view(np.uint32), np.ascontiguousarray(ref[k], np.float32).view(np.uint32)), (pipeline, k)
    assert (ref["rgb"][:, 0] != ref["rgb"][:, 1]).any()                  # the texture colours the radiance

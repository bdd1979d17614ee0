#!/usr/bin/env python3
"""Writes a corpus of damaged scene files for tools/sanitize_loader.sh:  python tests/dev/fuzz_corpus.py SEED COUNT OUTDIR
35 % numeric attribute values replaced by extreme ones, 25 % character-level damage of the XML (including stray <include> / <alias> / <path> tags), 10 % damaged radiance maps (RGBE / PFM / PNG / JPEG / OpenEXR), 30 % damaged obj / ply / serialized mesh files."""
import os, sys, random, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_scenes, make_mesh; make_scenes.ensure()
random.seed(int(sys.argv[1])); N = int(sys.argv[2]); out = sys.argv[3]
os.makedirs(out, exist_ok=True)
names = ("cornell_boxes.xml", "cornell_wall.xml", "cornell_area.xml", "cornell_specular.xml", "cornell_roughplastic.xml", "cornell_sphere_light.xml", "cornell_rough.xml", "cornell_plastic.xml", "cornell_spheres.xml", "cornell_frosted.xml", "cornell_spot.xml", "cornell_disk.xml", "domino_small.xml")
texts = [open(os.path.join(ROOT, "scenes", n)).read() for n in names]
num = re.compile(r'-?\d+\.?\d*(?:e-?\d+)?')
vals = ['0', '-0', '1e-30', '1e30', '1e39', 'nan', 'inf', '-1', '4294967296', '1e-45', '0.5', '-1e39', '2', '1', '1e-8', '1e8', '3', '7', '', 'x']
pos, nrm, uv, faces = make_mesh.blob(6, 5)
make_mesh.write_ply(os.path.join(out, "blob.ply"), pos, nrm, uv, faces)
make_mesh.write_obj(os.path.join(out, "a.obj"), pos, nrm, uv, faces); make_mesh.write_ply(os.path.join(out, "b.ply"), pos, nrm, uv, faces)
make_mesh.write_ply(os.path.join(out, "c.ply"), pos, nrm, uv, faces, binary=False); make_mesh.write_ply(os.path.join(out, "d.ply"), pos, nrm, uv, faces, big_endian=True, with_uv=True)
make_mesh.write_obj(os.path.join(out, "e.obj"), pos, nrm, uv, faces, with_normals=True, quads_as_polygons=True)
make_mesh.write_serialized(os.path.join(out, "f.serialized"), [(pos, nrm, uv, faces), (pos, None, None, faces)])
make_mesh.write_serialized(os.path.join(out, "g.serialized"), [(pos, None, uv, faces)], version=3, double_precision=True)
orig = {n: open(os.path.join(out, n), "rb").read() for n in ("a.obj", "b.ply", "c.ply", "d.ply", "e.obj", "f.serialized", "g.serialized")}
import zlib
xml0 = make_mesh.cornell_mesh_xml(moving_file="MOVING", res=16, spp=4)
# radiance maps of the envmap emitter (RGBE with and without run-length encoding, PFM colour and gray, PNG) and bitmap textures
sky = make_scenes.env_pixels(16, 8)
make_scenes.write_rgbe(os.path.join(out, "h.hdr"), sky); make_scenes.write_rgbe(os.path.join(out, "i.hdr"), sky, rle=False); make_scenes.write_pfm(os.path.join(out, "j.pfm"), sky)
make_scenes.write_png(os.path.join(out, "k.png"), [[tuple(min(255, int(40 * c)) for c in px) for px in row] for row in sky])
try:   # baseline JPEG fixtures (4:2:0 with restart markers, 4:4:4, grayscale) for the decoder of image_io.cpp
    from PIL import Image
    import numpy as np
    arr = (np.random.default_rng(3).random((24, 40, 3)) * 255).astype("uint8")
    Image.fromarray(arr).save(os.path.join(out, "l.jpg"), quality=70, subsampling=2, restart_marker_blocks=2)
    Image.fromarray(arr).save(os.path.join(out, "m.jpg"), quality=90, subsampling=0)
    Image.fromarray(arr[..., 0]).save(os.path.join(out, "n.jpg"), quality=50)
    jpegs = ("l.jpg", "m.jpg", "n.jpg")
except Exception:
    jpegs = ()
make_scenes.write_exr(os.path.join(out, "o.exr"), sky, compression=3); make_scenes.write_exr(os.path.join(out, "p.exr"), sky, compression=0, half=False)
make_scenes.write_exr(os.path.join(out, "q.exr"), sky, compression=2, alpha=True, decreasing_y=True)
images = {n: open(os.path.join(out, n), "rb").read() for n in ("h.hdr", "i.hdr", "j.pfm", "k.png", "o.exr", "p.exr", "q.exr") + jpegs}
_piz = "/root/reference/configs_example/scene.exr"     # the one PIZ-compressed file at hand (build container only)
if os.path.exists(_piz):
    images["r.exr"] = open(_piz, "rb").read()
env0 = make_scenes.cornell_envmap(16, 4, filename="IMAGE")
# files for the <include> tag: a <scene> root (with a nested include and a <default>), an object root, a file that includes itself
open(os.path.join(out, "part.xml"), "w").write('<scene version="3.0.0"><default name="extra" value="0.3"/><bsdf type="diffuse" id="included"><rgb name="reflectance" value="$extra"/></bsdf><include filename="part_object.xml"/></scene>')
open(os.path.join(out, "part_object.xml"), "w").write('<shape type="sphere"><float name="radius" value="0.1"/><bsdf type="diffuse"/></shape>')
open(os.path.join(out, "loop.xml"), "w").write('<scene version="3.0.0"><include filename="loop.xml"/></scene>')
for it in range(N):
    r = random.random()
    if r < 0.35:      # numeric value fuzz
        t = random.choice(texts)
        spans = [m.span() for m in num.finditer(t) if 'value=' in t[max(0, m.start() - 200):m.start()].split('<')[-1]]
        for a, b in sorted(random.sample(spans, random.randint(1, 3)), reverse=True):
            t = t[:a] + random.choice(vals) + t[b:]
    elif r < 0.6:     # character-level fuzz
        t = random.choice(texts)
        if len(t) > 20000: t = t[:20000] + "</scene>"
        b = list(t)
        for _ in range(random.randint(1, 4)):
            op = random.random(); i = random.randrange(len(b))
            if op < 0.3: del b[i:i + random.randint(1, 12)]
            elif op < 0.7: b.insert(i, random.choice(['<', '>', '"', '/', '$', '0', '-', 'e', ' ', '&', ';', '<!--', ']]>', '<?']))
            else: b[i:i+1] = list(random.choice(['<rgb/>', '<ref id="x"/>', '<shape type="obj"/>', '<transform name="to_world"/>', '<animation name="to_world"/>',
                                                   '<include filename="part.xml"/>', '<include filename="part_object.xml"/>', '<include filename="loop.xml"/>', '<include filename="s%d.xml"/>' % max(it - 1, 0),
                                                   '<include/>', '<alias id="Light" as="x"/>', '<alias id="x" as="Light"/>', '<path value="."/>', '<path value="nowhere"/>']))
        t = "".join(b)
    elif r < 0.7:     # image file fuzz (envmap)
        n = random.choice(list(images)); b = bytearray(images[n])
        for _ in range(random.randint(1, 4)):
            op = random.random(); i = random.randrange(len(b))
            if op < 0.3: del b[i:i + random.randint(1, 30)]
            elif op < 0.6: b[i] = random.randrange(256)
            elif op < 0.8: b[i:i] = random.choice([b"-1", b"99999999", b" ", b"\n", b"\x02\x02\x7f\xff", b"\xff\xff", b"\x80\x00", b"-Y 8 +X 100000\n", b"1e39"])
            else: b = b[:i]
            if not b: b = bytearray(b" ")
        fn = "img%d.%s" % (it, n.split(".")[1])
        open(os.path.join(out, fn), "wb").write(bytes(b))
        t = env0.replace("IMAGE", fn)
    else:             # mesh file fuzz
        n = random.choice(list(orig)); typ = n.split(".")[1]
        b = bytearray(orig[n])
        inner = typ == "serialized" and random.random() < 0.6     # damage the deflated body, not the zlib framing
        if inner:
            d = zlib.decompressobj(); b = bytearray(d.decompress(bytes(b[4:]))); tail = d.unused_data
        for _ in range(random.randint(1, 5)):
            op = random.random(); i = random.randrange(len(b))
            if op < 0.25: del b[i:i + random.randint(1, 40)]
            elif op < 0.5: b[i] = random.randrange(256)
            elif op < 0.7: b[i:i] = random.choice([b"-1", b"999999999", b"4294967295", b"/", b"//", b" ", b"\n", b"f 1 2\n", b"f -1 -2 -3\n", b"f 0 0 0\n", b"nan", b"1e39", b"\x00\x00\x00\x80", b"\xff\xff\xff\xff"])
            elif op < 0.85: b = b[:i]
            else: b[i:i+4] = random.choice([b"\xff\xff\xff\x7f", b"\x00\x00\x80\x7f", b"\x00\x00\xc0\x7f", b"\x01\x00\x00\x00"])
            if not b: b = bytearray(b" ")
        if inner: b = bytearray(orig[n][:4] + zlib.compress(bytes(b)) + (tail if random.random() < 0.7 else b""))
        fn = "m%d.%s" % (it, typ)
        open(os.path.join(out, fn), "wb").write(bytes(b))
        if typ == "serialized" and random.random() < 0.5: fn += '" />\n\t\t<integer name="shape_index" value="%d' % random.choice([1, 1, 2, 7, -1])
        t = xml0.replace("MOVING", fn).replace('<shape type="obj" id="MovingBlob">', '<shape type="%s" id="MovingBlob">' % typ)
    open(os.path.join(out, "s%d.xml" % it), "w", errors="ignore").write(t)

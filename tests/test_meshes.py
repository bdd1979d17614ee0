"""obj / ply triangle meshes: loaders, vertex baking, the per-mesh BLAS.

CPU part (no GPU): the product's C++ loader (through the C ABI export, kinds 4..7) against the oracle's independent
Python reader + C baking -- bit-exact positions / normals / texcoords / faces for every file flavour, and the
reference's error behaviour.
GPU part: every lane bit-exact against the oracle (which tests every triangle, no acceleration structure) on scenes
with a static ply mesh and a moving obj mesh; BLAS traversal == loop over all triangles on a 66k-triangle mesh.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scenes"))
import make_mesh  # noqa: E402
import make_scenes as ms  # noqa: E402

NCPU = os.cpu_count() or 1


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def mesh_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("meshes"))
    make_mesh.write_all(d, 24, 12)
    return d


def one_mesh_xml(plugin, filename, extra=""):
    return make_mesh.cornell_mesh_xml().split('\t<shape type="ply" id="StaticBlob">')[0] + make_mesh.mesh_shape(
        plugin, "M", filename, "TallBoxBSDF", "0.5", ("0.1", "0.6", "0.0"), extra=extra) + make_mesh.ms.LIGHT + "</scene>\n"


FLAVOURS = [("ply", "blob.ply", ""), ("ply", "blob_ascii.ply", ""), ("ply", "blob_be.ply", ""), ("obj", "blob.obj", ""),
            ("obj", "blob_n.obj", ""), ("obj", "blob.obj", '\t\t<boolean name="face_normals" value="true" />\n'),
            ("obj", "blob.obj", '\t\t<boolean name="flip_tex_coords" value="false" />\n\t\t<boolean name="flip_normals" value="true" />\n'),
            ("ply", "blob.ply", '\t\t<boolean name="face_normals" value="true" />\n'),
            ("serialized", "blob.serialized", ""), ("serialized", "blob.serialized", '\t\t<integer name="shape_index" value="1" />\n'),
            ("serialized", "blob_v3.serialized", '\t\t<integer name="shape_index" value="1" />\n'),
            ("serialized", "blob_v3.serialized", '\t\t<boolean name="face_normals" value="true" />\n')]


def mesh_arrays_of_oracle(osc):
    meshes = [s for s in osc.flat.shapes if s["kind"] == 1]
    cat = lambda key: np.concatenate([np.asarray(m[key]).reshape(-1) for m in meshes if m[key] is not None] or [np.zeros(0, np.float32)])
    return cat("positions"), cat("normals"), cat("texcoords"), np.concatenate([m["faces"].reshape(-1) for m in meshes])


@pytest.mark.parametrize("plugin,filename,extra", FLAVOURS, ids=["%s-%s-%d" % (f[0], f[1], i) for i, f in enumerate(FLAVOURS)])
def test_loader_matches_the_oracle_reader_bit_for_bit(mi, orc, mesh_dir, plugin, filename, extra):
    path = os.path.join(mesh_dir, "one.xml")
    open(path, "w").write(one_mesh_xml(plugin, filename, extra))
    sc, osc = mi.load_file(path), orc.Scene(path, {})
    pos, nrm, uv, faces = mesh_arrays_of_oracle(osc)
    assert sc.info()["n_triangles"] == faces.size // 3 == 528
    assert np.array_equal(bits(sc.export(4)), bits(pos))
    assert np.array_equal(bits(sc.export(5)), bits(nrm))
    assert np.array_equal(bits(sc.export(6)), bits(uv))
    assert np.array_equal(sc.export(7).view(np.uint32), faces)
    if "face_normals" in extra:
        assert sc.export(5).size == 0
    else:   # unit normals, pointing away from the blob's centre on average
        n = sc.export(5).reshape(-1, 3)
        assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
        p = sc.export(4).reshape(-1, 3)
        assert (np.einsum("ij,ij->i", n, p - p.mean(0)) > 0).mean() > 0.95


def test_obj_semantics(mi, orc, tmp_path):
    """de-duplication by (v, vt, vn), fan triangulation of polygons, vt flip, comments/blank lines (obj.cpp:214-336)"""
    (tmp_path / "q.obj").write_text("# quad + triangle\n\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0.25\n"
                                    "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                                    "f 1/1 2/2 3/3 4/4\n  f 4/4 3/3 5/2\nf 1/3 2/2 3/3\n")
    xml = one_mesh_xml("obj", "q.obj")
    (tmp_path / "s.xml").write_text(xml)
    sc, osc = mi.load_file(str(tmp_path / "s.xml")), orc.Scene(str(tmp_path / "s.xml"), {})
    faces = sc.export(7).view(np.uint32).reshape(-1, 3)
    # quad -> (0,1,2),(0,2,3); second face reuses 3,2 and adds vertex 4; "1/3" is a NEW vertex (different vt)
    assert faces.tolist() == [[0, 1, 2], [0, 2, 3], [3, 2, 4], [5, 1, 2]]
    uv = sc.export(6).reshape(-1, 2)
    assert np.allclose(uv[0], [0, 1]) and np.allclose(uv[2], [1, 0]) and np.allclose(uv[5], [1, 0])   # flipped v
    _, _, ouv, ofaces = mesh_arrays_of_oracle(osc)
    assert np.array_equal(ofaces.reshape(-1, 3), faces) and np.array_equal(bits(ouv), bits(uv.reshape(-1)))


def test_mesh_errors_follow_the_reference(mi, tmp_path):
    def load(plugin, fn):
        (tmp_path / "e.xml").write_text(one_mesh_xml(plugin, fn))
        return mi.load_file(str(tmp_path / "e.xml"))
    with pytest.raises(RuntimeError, match='Error while loading OBJ file "nope.obj": file not found'):
        load("obj", "nope.obj")
    with pytest.raises(RuntimeError, match='Error while loading PLY file "nope.ply": file not found!'):
        load("ply", "nope.ply")
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 7\n")
    with pytest.raises(RuntimeError, match="reference to invalid vertex 7"):
        load("obj", "bad.obj")
    (tmp_path / "quad.ply").write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                                       "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(RuntimeError, match="is this a triangle mesh"):
        load("ply", "quad.ply")
    (tmp_path / "trail.ply").write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                                        "element face 1\nproperty list uchar int vertex_index\nend_header\n0 0 0\n1 0 0\n1 1 0\n3 0 1 2\n9 9\n")
    with pytest.raises(RuntimeError, match="trailing content"):
        load("ply", "trail.ply")
    # found with tools/sanitize_loader.sh: a damaged header count must not be allocated (terabytes) before the first read fails,
    # and indices that are negative, fractional-huge or not numbers must not wrap into valid ones
    head = "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face %s\nproperty list uchar %s vertex_index\nend_header\n0 0 0\n1 0 0\n1 1 0\n3 %s\n"
    (tmp_path / "huge.ply").write_text(head % ("300000000000", "int", "0 1 2"))
    with pytest.raises(RuntimeError, match="more entries than the file can hold"):
        load("ply", "huge.ply")
    for k, idx in enumerate(("0 1 -2", "0 1 4294967298", "0 1 1e30", "0 nan 2")):
        (tmp_path / ("idx%d.ply" % k)).write_text(head % ("1", "float" if "e" in idx or "nan" in idx else "int", idx))
        with pytest.raises(RuntimeError, match="out of range"):
            load("ply", "idx%d.ply" % k)
    # serialized.cpp:246-296 (the reference's messages end in "!!": the text carries one "!", fail() adds the other)
    with pytest.raises(RuntimeError, match='Error while loading serialized file "nope.serialized": file not found!'):
        load("serialized", "nope.serialized")
    (tmp_path / "bad.serialized").write_bytes(b"\x1c\x05\x04\x00abcd")
    with pytest.raises(RuntimeError, match="encountered an invalid file format!!"):
        load("serialized", "bad.serialized")
    (tmp_path / "v5.serialized").write_bytes(b"\x1c\x04\x05\x00abcd")
    with pytest.raises(RuntimeError, match="encountered an incompatible file version!!"):
        load("serialized", "v5.serialized")
    (tmp_path / "junk.serialized").write_bytes(b"\x1c\x04\x04\x00" + b"not a zlib stream at all")
    with pytest.raises(RuntimeError, match="inflate"):
        load("serialized", "junk.serialized")
    make_mesh.write_serialized(str(tmp_path / "one.serialized"), [([(0, 0, 0), (1, 0, 0), (0, 1, 0)], None, None, [(0, 1, 2)])])
    (tmp_path / "idx.xml").write_text(one_mesh_xml("serialized", "one.serialized", '\t\t<integer name="shape_index" value="3" />\n'))
    with pytest.raises(RuntimeError, match=r"shape index is out of range! \(requested 3 out of 0..0\)"):
        mi.load_file(str(tmp_path / "idx.xml"))
    make_mesh.write_serialized(str(tmp_path / "oob.serialized"), [([(0, 0, 0), (1, 0, 0), (0, 1, 0)], None, None, [(0, 1, 5)])])
    with pytest.raises(RuntimeError, match="out of range"):
        load("serialized", "oob.serialized")
    import zlib, struct
    short = struct.pack("<HH", 0x041C, 4) + zlib.compress(struct.pack("<I", 0x1000) + b"m\0" + struct.pack("<QQ", 100, 100) + b"\0" * 40)
    (tmp_path / "short.serialized").write_bytes(short)
    with pytest.raises(RuntimeError, match="premature end"):
        load("serialized", "short.serialized")
    (tmp_path / "nofile.xml").write_text(one_mesh_xml("ply", "x.ply").replace('<string name="filename" value="x.ply" />', ""))
    with pytest.raises(RuntimeError, match="filename"):
        mi.load_file(str(tmp_path / "nofile.xml"))


def test_ply_extra_elements_and_types(mi, orc, tmp_path):
    """double / short typed vertex properties, s/t texcoords, an unknown element before the faces (ply.cpp:208-224,419-422)"""
    (tmp_path / "t.ply").write_text("ply\nformat ascii 1.0\ncomment x\nelement vertex 3\nproperty double x\nproperty double y\nproperty short z\n"
                                    "property float s\nproperty float t\nproperty uchar red\n"
                                    "element edge 1\nproperty int a\nproperty int b\n"
                                    "element face 1\nproperty uchar flag\nproperty list uchar uint vertex_index\nend_header\n"
                                    "0.1 0.2 3 0 0 255\n1.7 0.25 -2 1 0 0\n0.3 1.9 1 0 1 7\n0 1\n9 3 0 1 2\n")
    (tmp_path / "t.xml").write_text(one_mesh_xml("ply", "t.ply"))
    sc, osc = mi.load_file(str(tmp_path / "t.xml")), orc.Scene(str(tmp_path / "t.xml"), {})
    pos, nrm, uv, faces = mesh_arrays_of_oracle(osc)
    assert np.array_equal(bits(sc.export(4)), bits(pos)) and np.array_equal(bits(sc.export(6)), bits(uv))
    assert sc.export(7).view(np.uint32).tolist() == [0, 1, 2] and uv.tolist() == [0, 0, 1, 0, 0, 1]


# ------------------------------------------------------------------------------------------------ GPU
GPU_CASES = [("two_blobs", None, dict(resx=48, resy=48), 8),
             ("two_blobs_stratified", None, dict(resx=32, resy=32, time_sampling_method="stratified", antithetic_shift=0.0, max_depth=5), 8)] + [
             # a static scene under the default heterodyne antithetic sampling integrates to ~0 (image = rounding noise): use homodyne
             ("one-%d" % i, f, dict(resx=32, resy=32, hetero_frequency=0.0), 4) for i, f in enumerate(FLAVOURS)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,flavour,params,spp", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_mesh_scenes_are_bit_exact_per_lane(mi, orc, mesh_dir, name, flavour, params, spp):
    if flavour is None:
        path = os.path.join(mesh_dir, "cornell_mesh.xml")
    else:
        path = os.path.join(mesh_dir, name + ".xml")
        open(path, "w").write(one_mesh_xml(*flavour))
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    assert sc.info()["n_bvh_nodes"] > 50       # the meshes sit behind a BLAS (binary nodes: > 250, quantised 4-wide nodes: a third of that)
    pd = osc.params()
    w, h = sc.size
    n = w * h * spp
    g = sc.sample_lanes(2, spp, 0, n)
    o = osc.render_lanes(pd, 2, spp, 0, n, threads=NCPU)
    for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
        assert np.array_equal(bits(g[k]), bits(o[k])), (name, k, int((bits(g[k]) != bits(o[k])).sum()))
    img = sc.render(seed=2, spp=spp)
    ref, _ = osc.render(pd, seed=2, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5
    assert (g["rgb"] != 0).any()


@pytest.mark.gpu
def test_blas_equals_the_loop_over_all_triangles_on_a_large_mesh(mi, orc, tmp_path, monkeypatch):
    d = str(tmp_path)
    pos, nrm, uv, faces = make_mesh.blob(256, 130)          # 66 048 triangles
    make_mesh.write_ply(os.path.join(d, "blob.ply"), pos, nrm, uv, faces)
    make_mesh.write_obj(os.path.join(d, "blob.obj"), pos, nrm, uv, faces)
    open(os.path.join(d, "s.xml"), "w").write(make_mesh.cornell_mesh_xml())
    params = dict(resx=64, resy=64)
    sc = mi.load_file(os.path.join(d, "s.xml"), **params)
    info = sc.info()
    assert info["n_triangles"] == 2 * len(faces) and info["n_bvh_nodes"] > 20000
    n = 64 * 64 * 4
    a = sc.sample_lanes(5, 4, 0, n)
    monkeypatch.setenv("DTOF_BLAS", "0")
    flat = mi.load_file(os.path.join(d, "s.xml"), **params)
    assert flat.info()["n_bvh_nodes"] < 100
    b = flat.sample_lanes(5, 4, 0, n)
    monkeypatch.delenv("DTOF_BLAS")
    for k in a:
        assert np.array_equal(bits(a[k]), bits(b[k])), k
    # and a slice of lanes against the oracle (every triangle tested on the CPU)
    osc = orc.Scene(os.path.join(d, "s.xml"), params)
    m = 2048
    o = osc.render_lanes(osc.params(), 5, 4, 6000, m, threads=NCPU)
    assert np.array_equal(bits(a["rgb"][6000:6000 + m]), bits(o["rgb"]))
    # Round 5: the ray kernels of such scenes have an eight-waves-per-SIMD form (triangle + rectangle code only, a 16-entry LDS stack column with an overflow array --
    # this BLAS is deeper than that) and an XCD-aware block order.  Which block traces which queue segment, and with how many waves, changes no lane.
    assert info["bvh_stack_depth"] > 16
    img = sc.render(seed=5, spp=4)
    for env in (dict(DTOF_TRACE8="0"), dict(DTOF_XCD_REMAP="0"), dict(DTOF_XCD_REMAP="8"), dict(DTOF_XCD_REMAP="4096"), dict(DTOF_XCD_REMAP="1000003"), dict(DTOF_TRACE8="0", DTOF_XCD_REMAP="64"),
                dict(DTOF_TLAS_LDS="0"), dict(DTOF_NODES16="0"), dict(DTOF_NODES16="0", DTOF_TLAS_LDS="0"), dict(DTOF_DEFER="0"), dict(DTOF_DEFER="0", DTOF_TLAS_LDS="0"), dict(DTOF_BLAS_LEAF="8"), dict(DTOF_BLAS_LEAF="2")):   # triangles per BLAS leaf (the loader reads it): another tree, the same hits
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = mi.load_file(os.path.join(d, "s.xml"), **params)
        c = other.sample_lanes(5, 4, 0, n)
        for k in a:
            assert np.array_equal(bits(a[k]), bits(c[k])), (env, k)
        assert np.abs(other.render(seed=5, spp=4) - img).max() <= 1e-6 * np.abs(img).max(), env
        for k in env:
            monkeypatch.delenv(k)
    # ... and the same room with (a) a rough conductor panel -- an every-BSDF scene, whose rays take the same eight-wave kernels -- and (b) a sphere beside the blobs: a scene with a
    # BLAS AND an analytic shape keeps the six-wave kernels with every shape's code, walking the half-float nodes too.  Each against the float nodes and the loop over all triangles.
    extra = {"panel": '<shape type="rectangle"><transform name="to_world"><scale x="0.3" y="0.3"/><rotate y="1" angle="35"/><translate x="-0.55" y="1.2" z="0.2"/></transform>'
                      '<bsdf type="roughconductor"><float name="alpha" value="0.15"/></bsdf></shape>',
             "sphere": '<shape type="sphere"><point name="center" x="0.1" y="1.25" z="0.3"/><float name="radius" value="0.22"/>'
                       '<bsdf type="diffuse"><rgb name="reflectance" value="0.6, 0.7, 0.5"/></bsdf></shape>'}
    for name, shape in extra.items():
        path = os.path.join(d, "s_%s.xml" % name)
        open(path, "w").write(make_mesh.cornell_mesh_xml().replace("</scene>", shape + "</scene>"))
        got = {}
        for env in (dict(), dict(DTOF_NODES16="0"), dict(DTOF_DEFER="0"), dict(DTOF_BLAS="0")):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got[tuple(env)] = mi.load_file(path, **params).sample_lanes(5, 4, 0, n)
            for k in env:
                monkeypatch.delenv(k)
        ref = got[()]
        assert not np.array_equal(bits(ref["rgb"]), bits(a["rgb"])), name     # the extra shape is seen
        for key, lanes in got.items():
            for k in ref:
                assert np.array_equal(bits(ref[k]), bits(lanes[k])), (name, key, k)


@pytest.mark.gpu
def test_a_ray_that_puts_more_than_four_meshes_aside(mi, tmp_path, monkeypatch):
    """Round 5: the ray kernels of mesh scenes run as a pair of launches -- the first puts the objects behind a BLAS that a ray's TLAS walk reaches ASIDE (up to four), the second
    enters them in packed waves.  Seven blobs in a row along the view axis, half of them moving: a primary ray through the middle reaches all seven boxes, so the fifth, sixth and
    seventh are entered on the spot by the first launch.  Every lane the same bits with the pair for every ray kernel (DTOF_DEFER=1), the default (2), none (0), and with the loop
    over all triangles (DTOF_BLAS=0)."""
    d = str(tmp_path)
    pos, nrm, uv, faces = make_mesh.blob(24, 12)            # 552 triangles each: behind a BLAS
    make_mesh.write_ply(os.path.join(d, "blob.ply"), pos, nrm, uv, faces)
    s = ms.HEADER.format(spp=4, res=48, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    for i in range(7):   # along z, towards the camera, overlapping in x / y as seen from it; a thin one in front so that rays pass between its bumps
        s += make_mesh.mesh_shape("ply", "Blob%d" % i, "blob.ply", "TallBoxBSDF" if i % 2 else "ShortBoxBSDF", "%.3f" % (0.10 + 0.015 * i),
                                  ("%.3f" % (0.05 * (i % 3) - 0.05), "%.3f" % (1.0 + 0.04 * (i % 2)), "%.3f" % (-0.75 + 0.25 * i)), anim_dz="0.01" if i % 2 else None)
    open(os.path.join(d, "s.xml"), "w").write(s + ms.LIGHT + "</scene>\n")
    n = 48 * 48 * 4
    got = {}
    monkeypatch.setenv("DTOF_PIPELINE", "split")   # (a scene of 3.7 k triangles would take the fused kernels: the ray kernels are the split pipeline's)
    for env in (dict(DTOF_DEFER="1"), dict(), dict(DTOF_DEFER="0"), dict(DTOF_BLAS="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sc = mi.load_file(os.path.join(d, "s.xml"))
        got[tuple(env.items())] = (sc.sample_lanes(3, 4, 0, n), sc.render(seed=3, spp=4), sc.info())
        for k in env:
            monkeypatch.delenv(k)
    ref_lanes, ref_img, info = got[(("DTOF_BLAS", "0"),)]
    assert got[()][2]["n_bvh_nodes"] > 7 * 100 and info["n_bvh_nodes"] < 40
    for key, (lanes, img, _) in got.items():
        for k in ref_lanes:
            assert np.array_equal(bits(ref_lanes[k]), bits(lanes[k])), (key, k)
        assert np.abs(img - ref_img).max() <= 2e-6 * np.abs(ref_img).max(), key


@pytest.mark.gpu
def test_instanced_groups_with_a_mesh_and_a_rectangle_through_the_pair_of_launches(mi, tmp_path, monkeypatch):
    """The second launch of a ray-kernel pair enters OBJECTS: an instance whose shapegroup holds a blob behind a BLAS AND a rectangle (tested on the way), placed three times with
    moving keyframes, beside a plain blob.  The TLAS leaves of all four carry kLeafBlas.  Same lanes with the pair for every ray kernel, the default, one launch, and no BLAS at all."""
    d = str(tmp_path)
    pos, nrm, uv, faces = make_mesh.blob(20, 10)
    make_mesh.write_ply(os.path.join(d, "blob.ply"), pos, nrm, uv, faces)
    s = ms.HEADER.format(spp=4, res=48, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    s += ('<shape type="shapegroup" id="G"><shape type="ply"><string name="filename" value="blob.ply"/><transform name="to_world"><scale value="0.16"/></transform><ref id="TallBoxBSDF"/></shape>'
          '<shape type="rectangle"><transform name="to_world"><scale x="0.22" y="0.05"/><rotate x="1" angle="-60"/><translate y="-0.2"/></transform><ref id="ShortBoxBSDF"/></shape></shape>\n')
    for i, (x, y, z) in enumerate([(-0.45, 0.6, -0.2), (0.35, 0.9, 0.1), (0.0, 1.35, -0.4)]):
        s += ('<shape type="instance"><ref id="G"/><animation name="to_world"><transform time="0"><rotate y="1" angle="%d"/><translate x="%.2f" y="%.2f" z="%.2f"/></transform>'
              '<transform time="0.0015"><rotate y="1" angle="%d"/><translate x="%.2f" y="%.3f" z="%.2f"/></transform></animation></shape>\n' % (20 * i, x, y, z, 20 * i + 2, x, y + 0.01, z))
    s += make_mesh.mesh_shape("ply", "Plain", "blob.ply", "TallBoxBSDF", "0.2", ("0.45", "0.35", "0.45"))
    open(os.path.join(d, "s.xml"), "w").write(s + ms.LIGHT + "</scene>\n")
    n = 48 * 48 * 4
    monkeypatch.setenv("DTOF_PIPELINE", "split")
    got = {}
    for env in (dict(DTOF_DEFER="1"), dict(), dict(DTOF_DEFER="0"), dict(DTOF_BLAS="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sc = mi.load_file(os.path.join(d, "s.xml"))
        got[tuple(env.items())] = (sc.sample_lanes(9, 4, 0, n), sc.render(seed=9, spp=4))
        for k in env:
            monkeypatch.delenv(k)
    ref_lanes, ref_img = got[(("DTOF_BLAS", "0"),)]
    assert np.abs(ref_img).max() > 0
    for key, (lanes, img) in got.items():
        for k in ref_lanes:
            assert np.array_equal(bits(ref_lanes[k]), bits(lanes[k])), (key, k, int((bits(ref_lanes[k]) != bits(lanes[k])).sum()))
        assert np.abs(img - ref_img).max() <= 2e-6 * np.abs(ref_img).max(), key


# ------------------------------------------------------------------------------------------------ mesh area emitters
def test_quad_mesh_light_equals_the_rectangle_light_in_expectation(orc, mesh_dir):
    """Mesh::sample_position / pdf (mesh.cpp:478-573) against Rectangle's: the same square light once as a `rectangle`, once
    as a two-triangle obj mesh -- different sampling code, same integrand; the face table is the DiscreteDistribution
    of distr_1d.h:205-240."""
    from conftest import SCENES
    area = open(os.path.join(SCENES, "cornell_area.xml")).read()
    quad = area.replace('<shape type="rectangle" id="Light">', '<shape type="obj" id="Light">\n\t\t<string name="filename" value="quad.obj" />')
    assert quad != area
    path = os.path.join(mesh_dir, "area_quad.xml")
    open(path, "w").write(quad)
    P, integ = dict(resx=16, resy=16), dict(type="path", max_depth=4)
    a, b = orc.Scene(os.path.join(SCENES, "cornell_area.xml"), P), orc.Scene(path, P)
    light = [s for s in b.flat.shapes if s["kind"] == 1][-1]
    assert np.allclose(light["area_pmf"], [0.1, 0.1]) and np.allclose(light["area_cdf"], [0.1, 0.2])   # 0.5 x 0.4 square, two halves
    ia = np.mean([a.render(a.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(3)], axis=0)
    ib = np.mean([b.render(b.params(integrator=integ), seed=s, spp=256, threads=NCPU)[0] for s in range(3)], axis=0)
    assert abs(ia.mean() - ib.mean()) < 0.01 * ia.mean()
    assert np.abs(ia - ib).mean() < 0.05 * ia.mean()


LIGHT_CASES = [("blob_ply_normals", dict(), dict(resx=32, resy=32), 8),
               ("blob_obj_computed_normals", dict(light_file="blob.obj", plugin="obj"), dict(resx=24, resy=24, max_depth=6, time_sampling_method="stratified"), 8),
               ("blob_face_normals_flipped", dict(extra='\t\t<boolean name="face_normals" value="true" />\n'), dict(resx=24, resy=24), 4),
               ("quad_obj", dict(light_file="quad.obj", plugin="obj", scale=("0.25", "0.2", "1"), translate=("0", "1.0", "-0.95")), dict(resx=24, resy=24), 8)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,params,spp", LIGHT_CASES, ids=[c[0] for c in LIGHT_CASES])
def test_mesh_area_lights_are_bit_exact_per_lane(mi, orc, mesh_dir, name, kw, params, spp):
    path = os.path.join(mesh_dir, "light_" + name + ".xml")
    open(path, "w").write(make_mesh.cornell_mesh_light_xml(**kw))
    sc, osc = mi.load_file(path, **params), orc.Scene(path, params)
    assert sc.info()["n_emitters"] == 1
    w, h = sc.size
    n = w * h * spp
    for integ in (None, dict(type="path", max_depth=5)):
        if integ:
            sc.set_integrator(integ)
        pd = osc.params(integrator=integ) if integ else osc.params()
        g = sc.sample_lanes(4, spp, 0, n)
        o = osc.render_lanes(pd, 4, spp, 0, n, threads=NCPU)
        for k in ("sample_pos", "time", "ray_o", "ray_d", "rgb"):
            assert np.array_equal(bits(g[k]), bits(o[k])), (name, integ, k, int((bits(g[k]) != bits(o[k])).sum()))
        assert (g["rgb"] != 0).mean() > 0.3
    img = sc.render(seed=4, spp=spp)
    ref, _ = osc.render(pd, seed=4, spp=spp, threads=NCPU)
    assert float(np.abs(img - ref).max() / np.abs(ref).max()) <= 1e-5

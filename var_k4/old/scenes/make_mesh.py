#!/usr/bin/env python3
"""Deterministic triangle-mesh test assets (obj / ply meshes behind a BLAS).

There is no network and the tutorial scenes' meshes are not in the reference tree, so the meshes are procedural:
a "blob" = sphere of radius r displaced by a few low-frequency sines, tessellated n_u x n_v (2*n_u*(n_v-1) triangles).

    python scenes/make_mesh.py OUT_DIR [n_u n_v]     # writes blob.obj, blob_n.obj, blob.ply, blob_ascii.ply, cornell_mesh.xml

  cornell_mesh.xml   the Cornell room of cornell_boxes.xml with the two cubes replaced by
                       * a static  `ply` blob (binary little endian, with vertex normals) and
                       * a moving  `obj` blob (no normals in the file => computed; animated to_world => instance),
                     both diffuse, lit by the point light at the camera.
"""
import math
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_scenes as ms   # noqa: E402


def blob(n_u, n_v, seed=1, radius=1.0):
    """returns (positions, normals, uvs, faces); vertices on the poles are shared, the seam is duplicated for uv"""
    ph = [0.37 * seed, 1.1 + 0.21 * seed, 2.3 - 0.13 * seed]
    def r_of(th, ph_):
        return radius * (1.0 + 0.18 * math.sin(3 * th + ph[0]) * math.sin(2 * ph_ + ph[1]) + 0.08 * math.sin(5 * ph_ + ph[2]) * math.sin(th) ** 2)
    pos, uv = [], []
    for j in range(n_v + 1):
        th = math.pi * j / n_v
        for i in range(n_u + 1):
            p = 2 * math.pi * i / n_u
            r = r_of(th, p)
            pos.append((r * math.sin(th) * math.cos(p), r * math.cos(th), r * math.sin(th) * math.sin(p)))
            uv.append((i / n_u, j / n_v))
    faces = []
    W = n_u + 1
    for j in range(n_v):
        for i in range(n_u):
            a, b, c, d = j * W + i, j * W + i + 1, (j + 1) * W + i, (j + 1) * W + i + 1
            if j != 0:
                faces.append((a, b, c))
            if j != n_v - 1:
                faces.append((b, d, c))
    # outward-ish analytic normals: normalised position gradient approximated by the position (good enough as DATA)
    nrm = []
    for (x, y, z) in pos:
        l = math.sqrt(x * x + y * y + z * z) or 1.0
        nrm.append((x / l, y / l, z / l))
    return pos, nrm, uv, faces


def write_obj(path, pos, nrm, uv, faces, with_normals=False, with_uv=True, quads_as_polygons=False):
    with open(path, "w") as f:
        f.write("# procedural blob\n")
        for p in pos:
            f.write("v %.6f %.6f %.6f\n" % p)
        if with_uv:
            for t in uv:
                f.write("vt %.6f %.6f\n" % t)
        if with_normals:
            for n in nrm:
                f.write("vn %.6f %.6f %.6f\n" % n)
        for tri in faces:
            def ref(i):
                i += 1
                if with_uv and with_normals:
                    return "%d/%d/%d" % (i, i, i)
                if with_uv:
                    return "%d/%d" % (i, i)
                if with_normals:
                    return "%d//%d" % (i, i)
                return "%d" % i
            f.write("f %s %s %s\n" % tuple(ref(i) for i in tri))


def write_ply(path, pos, nrm, uv, faces, binary=True, with_normals=True, with_uv=False, big_endian=False):
    props = ["x", "y", "z"] + (["nx", "ny", "nz"] if with_normals else []) + (["u", "v"] if with_uv else [])
    hdr = "ply\nformat %s 1.0\ncomment procedural blob\nelement vertex %d\n" % (
        ("binary_big_endian" if big_endian else "binary_little_endian") if binary else "ascii", len(pos))
    hdr += "".join("property float %s\n" % p for p in props)
    hdr += "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(faces)
    with open(path, "wb") as f:
        f.write(hdr.encode())
        e = ">" if big_endian else "<"
        for i in range(len(pos)):
            row = list(pos[i]) + (list(nrm[i]) if with_normals else []) + (list(uv[i]) if with_uv else [])
            if binary:
                f.write(struct.pack(e + "%df" % len(row), *row))
            else:
                f.write((" ".join("%.6f" % v for v in row) + "\n").encode())
        for tri in faces:
            if binary:
                f.write(struct.pack(e + "B3i", 3, *tri))
            else:
                f.write(("3 %d %d %d\n" % tri).encode())


def write_serialized(path, meshes, version=4, double_precision=False):
    """Mitsuba's .serialized container: `meshes` = [(pos, nrm | None, uv | None, faces), ...], one zlib stream each."""
    import zlib
    blob_, offsets = b"", []
    for k, (pos, nrm, uv, faces) in enumerate(meshes):
        offsets.append(len(blob_))
        flags = (0x2000 if double_precision else 0x1000) | (1 if nrm is not None else 0) | (2 if uv is not None else 0)
        fl = "<%dd" if double_precision else "<%df"
        body = struct.pack("<I", flags)
        if version == 4:
            body += b"mesh%d\0" % k
        body += struct.pack("<QQ", len(pos), len(faces))
        for arr in (pos, nrm, uv):
            if arr is not None:
                flat = [c for v in arr for c in v]
                body += struct.pack(fl % len(flat), *flat)
        flat = [i for f in faces for i in f]
        body += struct.pack("<%dI" % len(flat), *flat)
        blob_ += struct.pack("<HH", 0x041C, version) + zlib.compress(body, 6)
    for o in offsets:
        blob_ += struct.pack("<Q" if version == 4 else "<I", o)
    blob_ += struct.pack("<I", len(meshes))
    open(path, "wb").write(blob_)


def mesh_shape(plugin, ident, filename, bsdf_id, scale, translate, anim_dz=None, extra=""):
    tf = ('\t\t\t<scale value="%s" />\n\t\t\t<translate x="%s" y="%s" z="%s" />\n' % ((scale,) + tuple(translate)))
    s = '\t<shape type="%s" id="%s">\n\t\t<string name="filename" value="%s" />\n%s' % (plugin, ident, filename, extra)
    if anim_dz is None:
        s += '\t\t<transform name="to_world">\n' + tf + '\t\t</transform>\n'
    else:
        s += ('\t\t<animation name="to_world">\n\t\t\t<transform time="0">\n' + tf.replace("\t\t\t<", "\t\t\t\t<") + '\t\t\t</transform>\n'
              '\t\t\t<transform time="0.0015">\n' + tf.replace("\t\t\t<", "\t\t\t\t<") +
              '\t\t\t\t<translate x="0.0" y="0.0" z="%s" />\n\t\t\t</transform>\n\t\t</animation>\n' % anim_dz)
    return s + '\t\t<ref id="%s" />\n\t</shape>\n' % bsdf_id


def cornell_mesh_xml(static_file="blob.ply", moving_file="blob.obj", res=128, spp=16):
    s = ms.HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    s += mesh_shape("ply", "StaticBlob", static_file, "TallBoxBSDF", "0.38", ("-0.38", "0.45", "-0.25"))
    s += mesh_shape("obj", "MovingBlob", moving_file, "ShortBoxBSDF", "0.3", ("0.4", "0.33", "0.35"), anim_dz="0.015")
    return s + ms.LIGHT + "</scene>\n"


def mesh_light(plugin, filename, scale, translate, radiance="17, 12, 4", extra=""):
    return ('\t<shape type="%s" id="Light">\n\t\t<string name="filename" value="%s" />\n%s\t\t<transform name="to_world">\n'
            '\t\t\t<scale x="%s" y="%s" z="%s" />\n\t\t\t<translate x="%s" y="%s" z="%s" />\n\t\t</transform>\n'
            '\t\t<emitter type="area">\n\t\t\t<rgb name="radiance" value="%s" />\n\t\t</emitter>\n\t</shape>\n'
            % ((plugin, filename, extra) + tuple(scale) + tuple(translate) + (radiance,)))


def cornell_mesh_light_xml(light_file="blob.ply", plugin="ply", res=64, spp=16, extra="", scale=("0.25", "0.08", "0.2"),
                           translate=("0", "1.8", "0")):
    """the Cornell room with its two moving boxes, lit by a MESH area light (a flattened blob under the ceiling)"""
    s = ms.HEADER.format(spp=spp, res=res, tsm="antithetic", shift="0.5") + ms.SENSOR.format(fov="19.5", cam=ms.CAM)
    for b in ms.BSDFS:
        s += ms.bsdf(*b)
    for name, m, b in ms.WALLS:
        s += ms.rect(name, m, b)
    s += ms.cube("ShortBox", ms.SHORT, "ShortBoxBSDF", "0.015") + ms.cube("TallBox", ms.TALL, "TallBoxBSDF", "-0.015")
    return s + mesh_light(plugin, light_file, scale, translate, extra=extra) + "</scene>\n"


def quad_obj(path):
    """the unit square [-1,1]^2 in the xy plane, normal +z, as two triangles (same footprint as a `rectangle`)"""
    with open(path, "w") as f:
        f.write("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nf 1 2 3\nf 1 3 4\n")


def write_all(out_dir, n_u=24, n_v=12):
    os.makedirs(out_dir, exist_ok=True)
    pos, nrm, uv, faces = blob(n_u, n_v)
    write_obj(os.path.join(out_dir, "blob.obj"), pos, nrm, uv, faces)                                   # uv, no normals
    write_obj(os.path.join(out_dir, "blob_n.obj"), pos, nrm, uv, faces, with_normals=True, with_uv=False)
    write_ply(os.path.join(out_dir, "blob.ply"), pos, nrm, uv, faces)                                   # binary LE + normals
    write_ply(os.path.join(out_dir, "blob_ascii.ply"), pos, nrm, uv, faces, binary=False, with_normals=False, with_uv=True)
    write_ply(os.path.join(out_dir, "blob_be.ply"), pos, nrm, uv, faces, big_endian=True)
    pos2, nrm2, uv2, faces2 = blob(n_u, n_v, seed=2)
    write_serialized(os.path.join(out_dir, "blob.serialized"), [(pos2, None, None, faces2), (pos, nrm, uv, faces)])              # v4, two sub-meshes
    write_serialized(os.path.join(out_dir, "blob_v3.serialized"), [(pos2, None, uv2, faces2), (pos, nrm, None, faces)], version=3, double_precision=True)
    with open(os.path.join(out_dir, "cornell_mesh.xml"), "w") as f:
        f.write(cornell_mesh_xml())
    with open(os.path.join(out_dir, "cornell_mesh_light.xml"), "w") as f:
        f.write(cornell_mesh_light_xml())
    quad_obj(os.path.join(out_dir, "quad.obj"))
    return len(faces)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh")
    nu, nv = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (24, 12)
    print("wrote", write_all(out, nu, nv), "triangles per blob to", out)

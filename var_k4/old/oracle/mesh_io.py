"""TEST INFRASTRUCTURE ONLY -- oracle-side readers of the two mesh formats the reference's tutorial scenes use.

Independent Python restatements of
  * OBJMesh  (src/shapes/obj.cpp:139-398): `v`, `vn`, `vt`, `f` records; vertices are de-duplicated by their
    (v, vt, vn) index triple in order of first use; polygons are fan-triangulated; `flip_tex_coords` (default true);
  * PLYMesh  (src/shapes/ply.cpp:160-441 + parse_ply_header/parse_ascii): ascii / binary_little_endian /
    binary_big_endian, typed vertex properties x y z [nx ny nz] [u v | s t | texture_u texture_v], a face list
    `vertex_index` / `vertex_indices` that must hold triangles; other elements are skipped.
Both return the RAW (object-space) arrays; the transform to world space and the computed vertex normals are applied
by oracle/dtof_oracle.c:orc_bake_mesh, exactly once, in C float32 -- see there.
"""
import struct
from fractions import Fraction

import numpy as np

F32 = np.float32


def strtof(tokens):
    """decimal strings -> float32 with ONE rounding (what strtof does); float(str) -> float32 rounds twice."""
    d = np.array([float(t) for t in tokens], dtype=np.float64)
    f = d.astype(F32)
    bits = d.view(np.uint64) & np.uint64(0x1FFFFFFF)
    for i in np.nonzero(bits == np.uint64(0x10000000))[0]:   # the double sits exactly between two floats: decide exactly
        try:
            exact = Fraction(tokens[i])
        except (ValueError, ZeroDivisionError):
            continue
        lo, hi = np.nextafter(f[i], F32(-np.inf)), np.nextafter(f[i], F32(np.inf))
        best = min((lo, f[i], hi), key=lambda c: (abs(Fraction(float(c)) - exact), int(np.asarray(c).view(np.uint32)) & 1))
        f[i] = best
    return f


class MeshError(ValueError):
    pass


def read_obj(path, flip_tex_coords=True, face_normals=False):
    name = path.split("/")[-1]
    def fail(msg):
        raise MeshError('Error while loading OBJ file "%s": %s' % (name, msg))
    try:
        text = open(path, "rb").read().decode("latin-1")
    except OSError:
        fail("file not found")
    v_tok, n_tok, t_tok = [], [], []
    keys, key_index, tris = [], {}, []
    for line in text.split("\n"):
        cur = line.lstrip(" \t\r")
        if len(cur) < 2:
            continue
        if cur[0] == "v" and cur[1] in " \t":
            t = cur[2:].split()
            if len(t) < 3:
                fail('could not parse line "%s"' % line)
            v_tok += t[:3]
        elif cur[:2] == "vn" and len(cur) > 2 and cur[2] in " \t":
            if not face_normals:
                t = cur[3:].split()
                if len(t) < 3:
                    fail('could not parse line "%s"' % line)
                n_tok += t[:3]
        elif cur[:2] == "vt" and len(cur) > 2 and cur[2] in " \t":
            t = cur[3:].split()
            if len(t) < 2:
                fail('could not parse line "%s"' % line)
            t_tok += t[:2]
        elif cur[0] == "f" and cur[1] in " \t":
            tri, count = [0, 0, 0], 0
            for tok in cur[2:].split():
                parts = tok.split("/")
                if len(parts) > 3 or not parts[0].isdigit():
                    fail('could not parse line "%s"' % line)
                key = (int(parts[0]), int(parts[1]) if len(parts) > 1 and parts[1] else 0,
                       int(parts[2]) if len(parts) > 2 and parts[2] else 0)
                if key[0] - 1 >= len(v_tok) // 3 or key[0] < 1:
                    fail("reference to invalid vertex %d!" % key[0])
                vid = key_index.get(key)
                if vid is None:
                    vid = key_index[key] = len(keys)
                    keys.append(key)
                if count < 3:
                    tri[count] = vid
                else:
                    tri[1], tri[2] = tri[2], vid
                count += 1
                if count >= 3:
                    tris.append(tuple(tri))
    vs = strtof(v_tok).reshape(-1, 3)
    ns = strtof(n_tok).reshape(-1, 3) if n_tok else np.zeros((0, 3), F32)
    ts = strtof(t_tok).reshape(-1, 2) if t_tok else np.zeros((0, 2), F32)
    if flip_tex_coords and len(ts):
        ts[:, 1] = F32(1.0) - ts[:, 1]
    nv = len(keys)
    pos = np.zeros((nv, 3), F32)
    nrm = np.zeros((nv, 3), F32)
    uv = np.zeros((nv, 2), F32)
    for i, (a, b, c) in enumerate(keys):
        pos[i] = vs[a - 1]
        if b:
            if b - 1 >= len(ts):
                fail("reference to invalid texture coordinate %d!" % b)
            uv[i] = ts[b - 1]
        if not face_normals and c:
            if c - 1 >= len(ns):
                fail("reference to invalid normal %d!" % c)
            nrm[i] = ns[c - 1]
    return dict(positions=pos, normals=nrm if (len(ns) and not face_normals) else None,
                texcoords=uv if len(ts) else None, faces=np.asarray(tris, dtype=np.uint32).reshape(-1, 3))


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_ply(path, face_normals=False):
    name = path.split("/")[-1]
    def fail(msg):
        raise MeshError('Error while loading PLY file "%s": %s!' % (name, msg))
    try:
        data = open(path, "rb").read()
    except OSError:
        fail("file not found")
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        fail("invalid PLY header")
    eol = data.find(b"\n", end)
    header = data[:end].decode("latin-1").split("\n")
    body = data[eol + 1:]
    fmt, elements = None, []
    for line in header[1:]:
        t = line.split()
        if not t or t[0] in ("comment", "obj_info"):
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append(dict(name=t[1], count=int(t[2]), props=[]))
        elif t[0] == "property":
            if not elements:
                fail("property before element")
            if t[1] == "list":
                elements[-1]["props"].append(("list", t[2], t[3], t[4]))
            else:
                elements[-1]["props"].append(("scalar", t[1], t[2]))
        else:
            fail('invalid PLY header: unknown token "%s"' % t[0])
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        fail("invalid PLY header: unknown format")
    for el in elements:
        for p in el["props"]:
            for ty in p[1:-1]:
                if ty not in _PLY_TYPES:
                    fail('invalid PLY header: unknown type "%s"' % ty)
    out = dict(positions=None, normals=None, texcoords=None, faces=None)
    ascii_tokens = body.decode("latin-1").split() if fmt == "ascii" else None
    tpos, off = 0, 0
    endian = ">" if fmt == "binary_big_endian" else "<"
    for el in elements:
        n = el["count"]
        scalar_only = all(p[0] == "scalar" for p in el["props"])
        names = [p[-1] for p in el["props"]]
        if el["name"] == "vertex":
            if not scalar_only:
                fail("incompatible contents -- is this a triangle mesh?")
            cols = {}
            if fmt == "ascii":
                k = len(names)
                toks = ascii_tokens[tpos:tpos + n * k]
                if len(toks) != n * k:
                    fail("unexpected end of file")
                tpos += n * k
                for j, p in enumerate(el["props"]):
                    col = toks[j::k]
                    cols[p[2]] = strtof(col) if _PLY_TYPES[p[1]] in "fd" else np.array([int(x) for x in col], dtype=np.float64).astype(F32)
            else:
                dt = np.dtype([(p[2], _np_type(endian, p[1])) for p in el["props"]])
                if off + n * dt.itemsize > len(body):
                    fail("unexpected end of file")
                rec = np.frombuffer(body, dtype=dt, count=n, offset=off)
                off += n * dt.itemsize
                for p in el["props"]:
                    cols[p[2]] = rec[p[2]].astype(F32)
            for a, b in (("texture_u", "texture_v"), ("s", "t")):
                if "u" not in cols and a in cols and b in cols:
                    cols["u"], cols["v"] = cols[a], cols[b]
            for c in "xyz":
                if c not in cols:
                    fail('Unable to find field "%s"' % c)
            out["positions"] = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(F32)
            if not face_normals and all(c in cols for c in ("nx", "ny", "nz")):
                out["normals"] = np.stack([cols["nx"], cols["ny"], cols["nz"]], axis=1).astype(F32)
            if "u" in cols and "v" in cols:
                out["texcoords"] = np.stack([cols["u"], cols["v"]], axis=1).astype(F32)
        elif el["name"] == "face":
            li = [i for i, p in enumerate(el["props"]) if p[0] == "list" and p[3] in ("vertex_index", "vertex_indices")]
            if not li:
                fail("vertex_index/vertex_indices property not found")
            faces = np.zeros((n, 3), np.uint32)
            if fmt == "ascii":
                for f in range(n):
                    for j, p in enumerate(el["props"]):
                        if p[0] == "list":
                            cnt = int(ascii_tokens[tpos]); tpos += 1
                            vals = ascii_tokens[tpos:tpos + cnt]; tpos += cnt
                            if j == li[0]:
                                if cnt != 3:
                                    fail("incompatible contents -- is this a triangle mesh?")
                                faces[f] = [int(x) for x in vals]
                        else:
                            tpos += 1
            elif len(el["props"]) == 1:   # the common layout: one list property -> fixed-size records if all are triangles
                p = el["props"][0]
                dt = np.dtype([("n", _np_type(endian, p[1])), ("i", _np_type(endian, p[2]), (3,))])
                if off + n * dt.itemsize > len(body):
                    fail("incompatible contents -- is this a triangle mesh?")
                rec = np.frombuffer(body, dtype=dt, count=n, offset=off)
                if n and not (rec["n"] == 3).all():
                    fail("incompatible contents -- is this a triangle mesh?")
                faces[:] = rec["i"].astype(np.uint32)
                off += n * dt.itemsize
            else:
                for f in range(n):
                    for j, p in enumerate(el["props"]):
                        if p[0] == "list":
                            cs, vs = _PLY_TYPES[p[1]], _PLY_TYPES[p[2]]
                            cnt = struct.unpack_from(endian + cs, body, off)[0]; off += struct.calcsize(cs)
                            if j == li[0]:
                                if cnt != 3:
                                    fail("incompatible contents -- is this a triangle mesh?")
                                faces[f] = struct.unpack_from(endian + "3" + vs, body, off)
                            off += cnt * struct.calcsize(vs)
                        else:
                            off += struct.calcsize(_PLY_TYPES[p[1]])
            out["faces"] = faces
        else:   # unknown element: skipped (ply.cpp:419-422)
            if fmt == "ascii":
                for _ in range(n):
                    for p in el["props"]:
                        if p[0] == "list":
                            cnt = int(ascii_tokens[tpos]); tpos += 1 + cnt
                        else:
                            tpos += 1
            else:
                for _ in range(n):
                    for p in el["props"]:
                        if p[0] == "list":
                            cs = _PLY_TYPES[p[1]]
                            cnt = struct.unpack_from(endian + cs, body, off)[0]
                            off += struct.calcsize(cs) + cnt * struct.calcsize(_PLY_TYPES[p[2]])
                        else:
                            off += struct.calcsize(_PLY_TYPES[p[1]])
    if fmt == "ascii":
        if tpos != len(ascii_tokens):
            fail("invalid file -- trailing content")
    elif off != len(body):
        fail("invalid file -- trailing content")
    if out["positions"] is None or out["faces"] is None:
        fail("vertex or face element missing")
    return out


def read_serialized(path, shape_index=0, face_normals=False):
    """src/shapes/serialized.cpp:237-372: uint16 0x041C + version (3 | 4), one zlib stream per sub-mesh (flags, [v4 name], u64 vertex
    and face counts, positions, [normals], [texcoords], [colours], u32 indices), sub-mesh offsets + count at the end of the file."""
    import zlib
    name = path.split("/")[-1]
    def fail(msg):
        raise MeshError('Error while loading serialized file "%s": %s!' % (name, msg))
    try:
        data = open(path, "rb").read()
    except OSError:
        fail("file not found")
    if shape_index < 0:
        fail("shape index must be nonnegative!")
    if len(data) < 4:
        fail("premature end of file")
    fmt, version = struct.unpack_from("<HH", data, 0)
    if fmt != 0x041C:
        fail("encountered an invalid file format!")
    if version not in (3, 4):
        fail("encountered an incompatible file version!")
    start = 4
    if shape_index != 0:
        count = struct.unpack_from("<I", data, len(data) - 4)[0]
        if shape_index > count:
            fail("Unable to unserialize mesh, shape index is out of range! (requested %d out of 0..%d)" % (shape_index, count - 1))
        try:
            if version == 4:
                start = struct.unpack_from("<Q", data, len(data) - 8 * (count - shape_index) - 4)[0] + 4
            else:
                start = struct.unpack_from("<I", data, len(data) - 4 * (count - shape_index + 1))[0] + 4
        except struct.error:
            fail("premature end of file")
    try:
        raw = zlib.decompressobj().decompress(data[start:])
    except zlib.error:
        fail("inflate(): stream error")
    pos = [0]
    def take(n):
        if pos[0] + n > len(raw):
            fail("premature end of the compressed stream")
        b = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return b
    flags = struct.unpack("<I", take(4))[0]
    if version == 4:
        while take(1) != b"\0":
            pass
    nv, nf = struct.unpack("<QQ", take(16))
    if nv > 2 ** 31 or nf > 2 ** 31:
        fail("implausible vertex / face count")
    dp = bool(flags & 0x2000)
    def floats(dim):
        a = np.frombuffer(take(nv * dim * (8 if dp else 4)), "<f8" if dp else "<f4")
        return a.astype(np.float32).reshape(nv, dim)
    out = {"positions": floats(3), "normals": None, "texcoords": None, "faces": None}
    if flags & 0x1:
        n = floats(3)
        if not face_normals:
            out["normals"] = n
    if flags & 0x2:
        out["texcoords"] = floats(2)
    if flags & 0x8:
        floats(3)
    out["faces"] = np.frombuffer(take(nf * 12), "<u4").reshape(nf, 3).copy()
    if nf and out["faces"].max() >= nv:
        fail("face references a vertex out of range")
    return out


def _np_type(endian, ply_type):
    c = _PLY_TYPES[ply_type]
    return {"b": "i1", "B": "u1", "h": endian + "i2", "H": endian + "u2", "i": endian + "i4", "I": endian + "u4",
            "f": endian + "f4", "d": endian + "f8"}[c]

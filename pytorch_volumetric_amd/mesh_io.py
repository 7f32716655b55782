"""Triangle-mesh container and file loading (host side, numpy float64).

Stands in for the legacy open3d TriangleMesh the reference loads at sdf.py:103-113: vertices are float64, faces are
triangles (polygons are fan-triangulated), and the frame operations (scale, rotate about the origin, translate) are
applied in float64 before anything is rounded to the float32 the query kernels use.
"""
import os

import numpy as np


class TriMesh:
    def __init__(self, vertices, faces):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int64).reshape(-1, 3)
        if self.faces.size and (self.faces.min() < 0 or self.faces.max() >= len(self.vertices)):
            raise ValueError("face index out of range")

    def copy(self):
        return TriMesh(self.vertices.copy(), self.faces.copy())

    def scaled(self, scale):
        """uniform (scalar) or per-axis (3-vector) scale about the origin"""
        return TriMesh(self.vertices * np.asarray(scale, dtype=np.float64), self.faces)

    def rotated(self, rot3x3):
        return TriMesh(self.vertices @ np.asarray(rot3x3, dtype=np.float64).T, self.faces)

    def translated(self, offset):
        return TriMesh(self.vertices + np.asarray(offset, dtype=np.float64).reshape(1, 3), self.faces)

    def triangle_soup(self):
        """[F,3,3] float64 corner positions."""
        return self.vertices[self.faces]

    def triangle_normals(self):
        """[F,3] float64 unit normals, (v1-v0) x (v2-v0) normalised; zero-area faces get a zero normal."""
        t = self.triangle_soup()
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        length = np.linalg.norm(n, axis=1, keepdims=True)
        return np.divide(n, length, out=np.zeros_like(n), where=length > 0)

    def triangle_areas(self):
        t = self.triangle_soup()
        return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)

    def aabb(self):
        """(min, max) over the vertices, computed once per vertex array: ObjectFactory.bounding_box (sdf.py:80-89) is asked
        twice per CachedSDF construction, and the two numpy reductions cost 0.55 of a 0.8 ms build on a 256-core host."""
        cached = self.__dict__.get("_aabb")
        if cached is None or cached[0] is not self.vertices:
            cached = self.__dict__["_aabb"] = (self.vertices, self.vertices.min(axis=0), self.vertices.max(axis=0))
        return cached[1].copy(), cached[2].copy()

    def center(self):
        return self.vertices.mean(axis=0)


def morton_order(points, bits=10):
    """Permutation that sorts 3-D points along a Z-order curve (2^bits cells per axis over their bounding box)."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros((0,), dtype=np.int64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    cell = np.clip(((p - lo) / np.maximum(hi - lo, 1e-30) * (2 ** bits - 1)).astype(np.int64), 0, 2 ** bits - 1)
    key = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for d in range(3):
            key |= ((cell[:, d] >> b) & 1) << (3 * b + d)
    return np.argsort(key, kind="stable")


def patch_order(points, leaf=16, tile=256):
    """Permutation that puts 3-D points (triangle centroids) into compact runs: a median split along the longest axis of
    the bounding box, recursively, with the split position rounded to whole tiles (then to whole leaves inside a tile), so
    that every aligned run of `tile` and of `leaf` consecutive points is one box of the recursion.  On a surface this gives
    patches about half the radius of the runs of a space-filling curve, which visits the empty cells next to the surface
    as well (99,500-triangle sphere: groups of 16 2.9 mm against 6.1 mm along the Z curve and 4.9 mm along the Hilbert
    curve, tiles of 256 13.7 against 26.2 and 20.5 mm) -- and the mesh kernels cull by exactly these radii."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(p)
    order = np.arange(n, dtype=np.int64)

    def split(lo, hi, unit, stop):  # segments larger than `stop`, halves rounded up to a multiple of `unit`
        stack = [(lo, hi)]
        while stack:
            a, b = stack.pop()
            m = b - a
            if m <= stop:
                continue
            half = ((m // 2 + unit - 1) // unit) * unit
            if half >= m:
                half = m - unit if m > unit else m // 2
            idx = order[a:b]
            q = p[idx]
            ax = int(np.argmax(q.max(axis=0) - q.min(axis=0)))
            order[a:b] = idx[np.argpartition(q[:, ax], half - 1)]
            stack.append((a, a + half))
            stack.append((a + half, b))

    split(0, n, tile, tile)
    full = (n // tile) * tile
    if full:  # inside the whole tiles every level has equal halves: all tiles at once
        width = tile
        while width > leaf:
            idx = order[:full].reshape(-1, width)
            q = p[idx]
            ax = np.argmax(q.max(axis=1) - q.min(axis=1), axis=1)
            key = np.take_along_axis(q, ax[:, None, None], axis=2)[:, :, 0]
            order[:full] = np.take_along_axis(idx, np.argsort(key, axis=1, kind="stable"), axis=1).reshape(-1)
            width //= 2
    split(full, n, leaf, leaf)
    return order


def _parse_obj(text):
    verts, faces = [], []
    for line in text.splitlines():
        if not line or line[0] not in "vf":
            continue
        parts = line.split()
        if parts[0] == "v":
            verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
        elif parts[0] == "f":
            idx = []
            for tok in parts[1:]:
                i = int(tok.split("/")[0])
                idx.append(i - 1 if i > 0 else len(verts) + i)  # negative = relative to the current end
            for k in range(1, len(idx) - 1):
                faces.append((idx[0], idx[k], idx[k + 1]))
    return np.array(verts, dtype=np.float64).reshape(-1, 3), np.array(faces, dtype=np.int64).reshape(-1, 3)


def _parse_stl(data):
    if data[:5].lower() == b"solid" and b"facet" in data[:1024]:
        verts = []
        for line in data.decode("ascii", "ignore").splitlines():
            parts = line.split()
            if len(parts) == 4 and parts[0] == "vertex":
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
        v = np.array(verts, dtype=np.float64).reshape(-1, 3)
    else:
        n = int(np.frombuffer(data, dtype="<u4", count=1, offset=80)[0])
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n,
                            offset=84)
        v = rec["v"].reshape(-1, 3).astype(np.float64)
    return v, np.arange(len(v), dtype=np.int64).reshape(-1, 3)


def _parse_ply(data):
    """PLY, ascii or binary_little_endian: vertex x/y/z (float/double) and a face list property of vertex indices."""
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii", "ignore").splitlines()
    fmt, elements, cur = None, [], None
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            cur = {"name": tok[1], "count": int(tok[2]), "props": []}
            elements.append(cur)
        elif tok[0] == "property" and cur is not None:
            cur["props"].append(tok[1:])
    np_types = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
                "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
                "double": "f8", "float64": "f8"}
    verts, faces = None, []
    if fmt == "ascii":
        lines = data[end:].decode("ascii", "ignore").split("\n")
        at = 0
        for el in elements:
            rows = [ln.split() for ln in lines[at:at + el["count"]]]
            at += el["count"]
            if el["name"] == "vertex":
                names = [pr[-1] for pr in el["props"]]
                ix = [names.index(c) for c in "xyz"]
                verts = np.array([[float(r[i]) for i in ix] for r in rows], dtype=np.float64).reshape(-1, 3)
            elif el["name"] == "face":
                for r in rows:
                    idx = [int(t) for t in r[1:1 + int(r[0])]]
                    faces.extend((idx[0], idx[k], idx[k + 1]) for k in range(1, len(idx) - 1))
    elif fmt == "binary_little_endian":
        at = end
        for el in elements:
            if el["name"] == "vertex":
                dt = np.dtype([(pr[-1], "<" + np_types[pr[0]]) for pr in el["props"]])
                rec = np.frombuffer(data, dtype=dt, count=el["count"], offset=at)
                verts = np.stack([rec[c].astype(np.float64) for c in "xyz"], axis=1)
                at += dt.itemsize * el["count"]
            elif el["name"] == "face":
                lists = [pr for pr in el["props"] if pr[0] == "list"]
                if len(lists) != 1 or len(el["props"]) != 1:
                    raise ValueError("PLY faces with extra per-face properties are not supported")
                ct, it = np.dtype("<" + np_types[lists[0][1]]), np.dtype("<" + np_types[lists[0][2]])
                for _ in range(el["count"]):
                    n = int(np.frombuffer(data, dtype=ct, count=1, offset=at)[0])
                    at += ct.itemsize
                    idx = np.frombuffer(data, dtype=it, count=n, offset=at)
                    at += it.itemsize * n
                    faces.extend((int(idx[0]), int(idx[k]), int(idx[k + 1])) for k in range(1, n - 1))
            else:
                raise ValueError(f"PLY element {el['name']} before the faces is not supported")
    else:
        raise ValueError(f"unsupported PLY format {fmt}")
    if verts is None:
        raise ValueError("PLY file has no vertex element")
    return verts, np.array(faces, dtype=np.int64).reshape(-1, 3)


# COLLADA: what the file's up axis does to the vertices.  The reference reads .dae through open3d -> assimp, whose Collada
# importer turns X_UP / Z_UP documents to Y_UP with a root transform; set False for loaders (pybullet, ROS) that ignore it.
# Not pinned against open3d here (it is not installed): INTEGRATION.md.
COLLADA_APPLY_UP_AXIS = True


def _dae_local_matrix(node):
    """product of a node's transform elements in document order (COLLADA 1.4 spec 5.3: post-multiplied, column vectors)"""
    m = np.eye(4)
    for el in node:
        if el.tag == "matrix":
            t = np.array(el.text.split(), dtype=np.float64).reshape(4, 4)
        elif el.tag == "translate":
            t = np.eye(4)
            t[:3, 3] = np.array(el.text.split(), dtype=np.float64)
        elif el.tag == "scale":
            t = np.diag(np.append(np.array(el.text.split(), dtype=np.float64), 1.0))
        elif el.tag == "rotate":
            x, y, z, deg = np.array(el.text.split(), dtype=np.float64)
            axis = np.array([x, y, z])
            n = np.linalg.norm(axis)
            t = np.eye(4)
            if n > 0:
                axis = axis / n
                a = np.deg2rad(deg)
                k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
                t[:3, :3] = np.eye(3) + np.sin(a) * k + (1 - np.cos(a)) * (k @ k)
        elif el.tag in ("lookat", "skew"):
            raise ValueError(f"COLLADA <{el.tag}> node transforms are not supported")
        else:
            continue
        m = m @ t
    return m


def _dae_geometry(mesh):
    """(positions [V,3], faces [F,3]) of one <mesh>: every <triangles> / <polylist> / <polygons> / <tristrips> / <trifans> that
    indexes its <vertices>; polygons are fan-triangulated (as the .obj reader does)"""
    sources = {}
    for src in mesh.findall("source"):
        fa = src.find("float_array")
        if fa is None or not (fa.text or "").strip():
            continue
        arr = np.array(fa.text.split(), dtype=np.float64)
        acc = src.find("technique_common/accessor")
        stride = int(acc.get("stride", 3)) if acc is not None else 3
        first = int(acc.get("offset", 0)) if acc is not None else 0
        count = int(acc.get("count", (len(arr) - first) // stride)) if acc is not None else (len(arr) - first) // stride
        sources[src.get("id")] = arr[first:first + count * stride].reshape(count, stride)
    positions_of = {}
    for v in mesh.findall("vertices"):
        for inp in v.findall("input"):
            if inp.get("semantic") == "POSITION":
                positions_of[v.get("id")] = inp.get("source", "").lstrip("#")
    pos, faces = None, []
    for prim in mesh:
        if prim.tag not in ("triangles", "polylist", "polygons", "tristrips", "trifans"):
            continue  # <lines>, <linestrips>: not surface
        inputs = prim.findall("input")
        vin = next((i for i in inputs if i.get("semantic") == "VERTEX"), None)
        if vin is None:
            continue
        src = positions_of.get(vin.get("source", "").lstrip("#"))
        if src not in sources:
            raise ValueError("COLLADA primitive refers to vertices without a POSITION float_array")
        if pos is None:
            pos = sources[src][:, :3]
        elif sources[src][:, :3] is not pos and not np.array_equal(sources[src][:, :3], pos):
            raise ValueError("COLLADA mesh with several <vertices> position arrays is not supported")
        step = max(int(i.get("offset", 0)) for i in inputs) + 1
        voff = int(vin.get("offset", 0))
        runs = [np.array(p.text.split(), dtype=np.int64)[voff::step] for p in prim.findall("p") if (p.text or "").strip()]
        if prim.tag == "triangles":
            for r in runs:
                faces.append(r[:len(r) // 3 * 3].reshape(-1, 3))
        elif prim.tag == "polylist":
            counts = np.array((prim.findtext("vcount") or "").split(), dtype=np.int64)
            idx = runs[0] if runs else np.zeros((0,), dtype=np.int64)
            start = 0
            if len(counts) and (counts == 3).all():
                faces.append(idx[:3 * len(counts)].reshape(-1, 3))
            else:
                for c in counts:
                    poly = idx[start:start + c]
                    start += c
                    faces.extend(np.array([[poly[0], poly[k], poly[k + 1]]]) for k in range(1, c - 1))
        elif prim.tag in ("polygons", "trifans"):
            for poly in runs:
                faces.extend(np.array([[poly[0], poly[k], poly[k + 1]]]) for k in range(1, len(poly) - 1))
        else:  # tristrips: alternate the winding
            for strip in runs:
                faces.extend(np.array([[strip[k], strip[k + 1], strip[k + 2]] if k % 2 == 0 else
                                       [strip[k + 1], strip[k], strip[k + 2]]]) for k in range(len(strip) - 2))
    if pos is None or not faces:
        return None
    return pos, np.concatenate(faces, axis=0).reshape(-1, 3)


def _parse_dae(data):
    """COLLADA 1.4 / 1.5 triangle geometry, flattened the way assimp's aiProcess_PreTransformVertices does for open3d
    (sdf.py:104 of the reference reads whatever open3d reads): every <instance_geometry> of the visual scene placed by its
    nodes' transforms, concatenated; the document's up axis turned to Y_UP (COLLADA_APPLY_UP_AXIS); <unit> NOT applied
    (assimp keeps it as metadata).  No materials, normals, skins (<instance_controller> is skipped)."""
    import xml.etree.ElementTree as ET
    root = ET.fromstring(data)
    for el in root.iter():
        el.tag = el.tag.rsplit("}", 1)[-1]
    geoms = {}
    for g in root.findall("library_geometries/geometry"):
        m = g.find("mesh")
        parsed = _dae_geometry(m) if m is not None else None
        if parsed is not None:
            geoms[g.get("id")] = parsed
    if not geoms:
        raise ValueError("COLLADA file holds no triangle geometry")
    library_nodes = {n.get("id"): n for n in root.findall("library_nodes//node") if n.get("id")}
    placed = []

    def walk(node, parent, depth=0):
        if depth > 64:
            raise ValueError("COLLADA node hierarchy deeper than 64 (an <instance_node> cycle?)")
        here = parent @ _dae_local_matrix(node)
        for el in node:
            if el.tag == "instance_geometry":
                gid = el.get("url", "").lstrip("#")
                if gid in geoms:
                    placed.append((gid, here))
            elif el.tag == "node":
                walk(el, here, depth + 1)
            elif el.tag == "instance_node":
                ref = library_nodes.get(el.get("url", "").lstrip("#"))
                if ref is not None:
                    walk(ref, here, depth + 1)

    scenes = root.findall("library_visual_scenes/visual_scene")
    want = root.find("scene/instance_visual_scene")
    if want is not None:
        chosen = [v for v in scenes if v.get("id") == want.get("url", "").lstrip("#")] or scenes[:1]
    else:
        chosen = scenes[:1]
    for vs in chosen:
        for node in vs.findall("node"):
            walk(node, np.eye(4))
    if not placed:  # no scene: the geometries as they are
        placed = [(gid, np.eye(4)) for gid in geoms]
    up = (root.findtext("asset/up_axis") or "Y_UP").strip().upper()
    to_y_up = np.eye(4)
    if COLLADA_APPLY_UP_AXIS and up == "Z_UP":
        to_y_up[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]
    elif COLLADA_APPLY_UP_AXIS and up == "X_UP":
        to_y_up[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    verts, faces, base = [], [], 0
    for gid, m in placed:
        pos, f = geoms[gid]
        m = to_y_up @ m
        verts.append(pos @ m[:3, :3].T + m[:3, 3])
        # a mirroring placement turns the triangles inside out: keep them facing outwards
        faces.append((f if np.linalg.det(m[:3, :3]) >= 0 else f[:, ::-1]) + base)
        base += len(pos)
    return np.concatenate(verts, axis=0), np.concatenate(faces, axis=0)


_GLTF_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _parse_gltf(data, base_dir):
    """glTF 2.0 (.gltf with external or data: buffers, .glb), flattened as assimp's aiProcess_PreTransformVertices does for
    open3d: every mesh primitive of the default scene placed by its nodes (matrix, or translation * rotation * scale;
    column-major, quaternions xyzw), concatenated.  Triangle lists, strips and fans; glTF is Y-up and metres by definition, so
    nothing is turned or scaled.  Sparse accessors and compressed (Draco / meshopt) primitives raise."""
    import base64
    import json
    import struct
    blob = None
    if data[:4] == b"glTF":
        _, _, total = struct.unpack_from("<III", data, 0)
        pos, doc = 12, None
        while pos + 8 <= min(total, len(data)):
            n, kind = struct.unpack_from("<II", data, pos)
            chunk = data[pos + 8:pos + 8 + n]
            if kind == 0x4E4F534A:
                doc = json.loads(chunk.decode("utf-8"))
            elif kind == 0x004E4942 and blob is None:
                blob = chunk
            pos += 8 + n + (-n % 4)
        if doc is None:
            raise ValueError("GLB file without a JSON chunk")
    else:
        doc = json.loads(data.decode("utf-8"))
    if any(ext in doc.get("extensionsRequired", []) for ext in ("KHR_draco_mesh_compression", "EXT_meshopt_compression")):
        raise ValueError("compressed glTF (Draco / meshopt) is not supported; re-export without compression")
    buffers = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if blob is None:
                raise ValueError("glTF buffer without a uri outside a GLB container")
            buffers.append(blob)
        elif uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            from urllib.parse import unquote
            with open(os.path.join(base_dir, unquote(uri)), "rb") as f:
                buffers.append(f.read())

    def accessor(index):
        a = doc["accessors"][index]
        if "sparse" in a:
            raise ValueError("sparse glTF accessors are not supported")
        dtype, width = np.dtype(_GLTF_COMPONENT[a["componentType"]]), _GLTF_WIDTH[a["type"]]
        if "bufferView" not in a:
            return np.zeros((a["count"], width), dtype=dtype)
        view = doc["bufferViews"][a["bufferView"]]
        start = view.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = view.get("byteStride") or dtype.itemsize * width
        raw = np.frombuffer(buffers[view["buffer"]], dtype=np.uint8)
        rows = np.lib.stride_tricks.as_strided(raw[start:], shape=(a["count"], dtype.itemsize * width), strides=(stride, 1))
        out = np.ascontiguousarray(rows).view(dtype).reshape(a["count"], width)
        if a.get("normalized") and dtype.kind in "iu":
            out = np.maximum(out.astype(np.float64) / np.iinfo(dtype).max, -1.0)
        return out

    def local_matrix(node):
        if "matrix" in node:
            return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T  # stored column-major
        m = np.eye(4)
        x, y, z, w = node.get("rotation", (0.0, 0.0, 0.0, 1.0))
        n = np.sqrt(x * x + y * y + z * z + w * w) or 1.0
        x, y, z, w = x / n, y / n, z / n, w / n
        m[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        m[:3, :3] = m[:3, :3] * np.array(node.get("scale", (1.0, 1.0, 1.0)), dtype=np.float64)[None, :]
        m[:3, 3] = node.get("translation", (0.0, 0.0, 0.0))
        return m

    verts, faces, base = [], [], 0

    def emit(mesh_index, m):
        nonlocal base
        for prim in doc["meshes"][mesh_index].get("primitives", []):
            if "KHR_draco_mesh_compression" in prim.get("extensions", {}):
                raise ValueError("compressed glTF (Draco) is not supported; re-export without compression")
            mode = prim.get("mode", 4)
            if mode not in (4, 5, 6) or "POSITION" not in prim.get("attributes", {}):
                continue  # points / lines: not surface
            pos = accessor(prim["attributes"]["POSITION"]).astype(np.float64)[:, :3]
            idx = accessor(prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim else np.arange(len(pos), dtype=np.int64)
            if mode == 4:
                f = idx[:len(idx) // 3 * 3].reshape(-1, 3)
            elif mode == 5:
                f = np.array([[idx[k], idx[k + 1], idx[k + 2]] if k % 2 == 0 else [idx[k + 1], idx[k], idx[k + 2]]
                              for k in range(len(idx) - 2)], dtype=np.int64).reshape(-1, 3)
            else:
                f = np.array([[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)], dtype=np.int64).reshape(-1, 3)
            verts.append(pos @ m[:3, :3].T + m[:3, 3])
            faces.append((f if np.linalg.det(m[:3, :3]) >= 0 else f[:, ::-1]) + base)
            base += len(pos)

    def walk(node_index, parent, depth=0):
        if depth > 64:
            raise ValueError("glTF node hierarchy deeper than 64 (a cycle?)")
        node = doc["nodes"][node_index]
        here = parent @ local_matrix(node)
        if "mesh" in node:
            emit(node["mesh"], here)
        for child in node.get("children", []):
            walk(child, here, depth + 1)

    scenes = doc.get("scenes", [])
    if scenes:
        for root in scenes[doc.get("scene", 0)].get("nodes", []):
            walk(root, np.eye(4))
    if not verts:  # no scene (or an empty one): the meshes as they are
        for i in range(len(doc.get("meshes", []))):
            emit(i, np.eye(4))
    if not verts:
        raise ValueError("glTF file holds no triangle geometry")
    return np.concatenate(verts, axis=0), np.concatenate(faces, axis=0)


SUPPORTED_MESH_EXTENSIONS = (".obj", ".stl", ".ply", ".off", ".dae", ".gltf", ".glb", ".npz")


def _parse_off(text):
    """Object File Format: "OFF", then `nv nf ne` (on the same or the next line), nv vertex lines, nf face lines
    `k i0 .. ik-1 [colour]` (polygons fanned); `#` comments and blank lines anywhere."""
    lines = [l.split("#", 1)[0].split() for l in text.splitlines()]
    lines = [l for l in lines if l]
    if not lines or not lines[0][0].upper().endswith("OFF"):
        raise ValueError("not an OFF file")
    if len(lines[0]) >= 3:
        counts, first = lines[0][1:], 1
    else:
        counts, first = lines[1], 2
    nv, nf = int(counts[0]), int(counts[1])
    verts = np.array([l[:3] for l in lines[first:first + nv]], dtype=np.float64).reshape(nv, 3)
    faces = []
    for l in lines[first + nv:first + nv + nf]:
        k = int(l[0])
        idx = [int(t) for t in l[1:1 + k]]
        for j in range(1, k - 1):
            faces.append((idx[0], idx[j], idx[j + 1]))
    return verts, np.array(faces, dtype=np.int64).reshape(-1, 3)


def load_mesh(path):
    """.obj (text), .stl (ascii/binary), .ply (ascii/binary little endian), .off (text), .dae (COLLADA: the scene's node
    transforms and up axis applied, see _parse_dae), .gltf / .glb (glTF 2.0, node transforms applied, see _parse_gltf) or .npz
    with arrays `vertices` [V,3] and `faces` [F,3].  STL repeats every vertex per triangle; identical positions are merged (as
    open3d does when it reads an STL) so that center() and the vertex count match the reference loader.  Anything else
    (.fbx, .3ds, ...) raises instead of yielding an empty mesh."""
    ext = os.path.splitext(path)[1].lower()
    if ext not in SUPPORTED_MESH_EXTENSIONS:
        raise ValueError(f"unsupported mesh format '{ext}' ({path}); supported: {', '.join(SUPPORTED_MESH_EXTENSIONS)}. "
                         "Convert the mesh (e.g. to .obj) or pass a TriMesh through `mesh=`.")
    if ext == ".npz":
        with np.load(path) as z:
            mesh = TriMesh(z["vertices"], z["faces"])
    elif ext == ".stl":
        with open(path, "rb") as f:
            v, faces = _parse_stl(f.read())
        uniq, inverse = np.unique(v, axis=0, return_inverse=True)
        mesh = TriMesh(uniq, inverse.reshape(-1)[faces])
    elif ext == ".ply":
        with open(path, "rb") as f:
            mesh = TriMesh(*_parse_ply(f.read()))
    elif ext == ".off":
        with open(path, "r") as f:
            mesh = TriMesh(*_parse_off(f.read()))
    elif ext == ".dae":
        with open(path, "rb") as f:
            mesh = TriMesh(*_parse_dae(f.read()))
    elif ext in (".gltf", ".glb"):
        with open(path, "rb") as f:
            mesh = TriMesh(*_parse_gltf(f.read(), os.path.dirname(os.path.abspath(path))))
    else:
        with open(path, "r") as f:
            mesh = TriMesh(*_parse_obj(f.read()))
    if len(mesh.faces) == 0 or len(mesh.vertices) == 0:
        raise ValueError(f"mesh file {path} has {len(mesh.vertices)} vertices and {len(mesh.faces)} triangles")
    return mesh


def save_obj(path, mesh):
    with open(path, "w") as f:
        for v in mesh.vertices:
            f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for t in mesh.faces:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


# ---- procedural meshes (closed, outward-oriented) used by tests and the synthetic benchmark configs ----
def box_mesh(half_extents=(1.0, 1.0, 1.0), center=(0.0, 0.0, 0.0)):
    h = np.asarray(half_extents, dtype=np.float64)
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * h
    faces = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                      [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int64)
    return TriMesh(corners + np.asarray(center, dtype=np.float64), faces)


def uv_sphere_mesh(radius=1.0, n_lon=32, n_lat=16, scale=(1.0, 1.0, 1.0), center=(0.0, 0.0, 0.0)):
    """Lat-long sphere: 2*n_lon*(n_lat-1) triangles.  n_lon=250, n_lat=200 -> 99,500 (BASELINE config C5)."""
    verts = [(0.0, 0.0, radius)]
    for i in range(1, n_lat):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            verts.append((radius * np.sin(th) * np.cos(ph), radius * np.sin(th) * np.sin(ph), radius * np.cos(th)))
    verts.append((0.0, 0.0, -radius))
    south = len(verts) - 1
    faces = []
    ring = lambda i, j: 1 + (i - 1) * n_lon + (j % n_lon)
    for j in range(n_lon):
        faces.append((0, ring(1, j), ring(1, j + 1)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, c, d))
            faces.append((a, d, b))
    for j in range(n_lon):
        faces.append((south, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)))
    v = np.array(verts, dtype=np.float64) * np.asarray(scale, dtype=np.float64) + np.asarray(center, dtype=np.float64)
    return TriMesh(v, np.array(faces, dtype=np.int64))

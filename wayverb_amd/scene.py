"""Triangle-soup scenes for the mesh set-up path: small generators for
tests plus a minimal OBJ reader.  Host-side plumbing only.

vertices: float32[n, 4] (cl_float3 layout, w = 0); triangles: uint32[m, 4] = {surface, v0, v1, v2}
(src/core/include/core/cl/triangle.h:9-14)."""
import math

import numpy as np


def _pack(verts, tris, surface=0):
    v = np.zeros((len(verts), 4), dtype=np.float32)
    v[:, :3] = np.asarray(verts, dtype=np.float32)
    t = np.zeros((len(tris), 4), dtype=np.uint32)
    t[:, 0] = surface
    t[:, 1:] = np.asarray(tris, dtype=np.uint32)
    return v, t


def box_scene(lo, hi):
    """geo::get_scene_data(box) -- src/core/include/core/geo/box.h:38-63: 8 vertices, 12 triangles."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    verts = [(x0, y0, z0), (x1, y0, z0), (x0, y1, z0), (x1, y1, z0), (x0, y0, z1), (x1, y0, z1), (x0, y1, z1), (x1, y1, z1)]
    tris = [(0, 1, 5), (0, 5, 4), (1, 0, 3), (0, 2, 3), (2, 0, 6), (0, 4, 6), (5, 1, 7), (1, 3, 7), (3, 2, 7), (2, 6, 7),
            (4, 5, 7), (6, 4, 7)]
    return _pack(verts, tris)


def prism_scene(polygon_xy, z0, z1):
    """Closed prism over a simple polygon (counter-clockwise list of (x, y)): walls + fan caps.
    An L-shaped room is prism_scene([(0,0),(4,0),(4,2),(2,2),(2,3),(0,3)], 0, 2.5)."""
    n = len(polygon_xy)
    verts = [(x, y, z0) for x, y in polygon_xy] + [(x, y, z1) for x, y in polygon_xy]
    tris = []
    for i in range(n):
        j = (i + 1) % n
        tris += [(i, j, n + j), (i, n + j, n + i)]
    # caps by ear clipping (polygons here are small; O(n^2) is fine)
    idx = list(range(n))

    def area2(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])

    def inside(p, a, b, c):
        return area2(a, b, p) > 0 and area2(b, c, p) > 0 and area2(c, a, p) > 0

    while len(idx) > 3:
        for k in range(len(idx)):
            i0, i1, i2 = idx[k - 1], idx[k], idx[(k + 1) % len(idx)]
            a, b, c = polygon_xy[i0], polygon_xy[i1], polygon_xy[i2]
            if area2(a, b, c) <= 0:
                continue
            if any(inside(polygon_xy[m], a, b, c) for m in idx if m not in (i0, i1, i2)):
                continue
            tris += [(i0, i2, i1), (n + i0, n + i1, n + i2)]
            idx.pop(k)
            break
        else:
            raise ValueError("polygon is not simple / counter-clockwise")
    i0, i1, i2 = idx
    tris += [(i0, i2, i1), (n + i0, n + i1, n + i2)]
    return _pack(verts, tris)


def icosphere_scene(centre, radius, subdivisions=2):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    tris = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
            (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
            (8, 6, 7), (9, 8, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    for _ in range(subdivisions):
        cache, new = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in tris:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        tris = new
    out = [tuple(np.asarray(centre) + radius * v) for v in verts]
    return _pack(out, tris)


def read_obj(path):
    """Minimal Wavefront OBJ reader: `v`, `f` (fan-triangulated, negative indices allowed) and
    `usemtl` (one surface index per material name, in order of first use).
    Returns (vertices, triangles, material_names)."""
    verts, tris, materials = [], [], []
    current = 0
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                verts.append(tuple(float(x) for x in parts[1:4]))
            elif parts[0] == "usemtl":
                name = parts[1] if len(parts) > 1 else ""
                if name not in materials:
                    materials.append(name)
                current = materials.index(name)
            elif parts[0] == "f":
                idx = []
                for p in parts[1:]:
                    k = int(p.split("/")[0])
                    idx.append(k - 1 if k > 0 else len(verts) + k)
                for j in range(1, len(idx) - 1):
                    tris.append((current, idx[0], idx[j], idx[j + 1]))
    v = np.zeros((len(verts), 4), dtype=np.float32)
    v[:, :3] = np.asarray(verts, dtype=np.float32)
    t = np.asarray(tris, dtype=np.uint32).reshape(-1, 4)
    return v, t, materials or ["default"]


def padded_aabb(vertices, padding):
    """make_voxelised_scene_data(scene, depth, padding): bounding box padded on every side
    (src/core/include/core/spatial_division/voxelised_scene_data.h:72-79)."""
    lo = vertices[:, :3].min(axis=0).astype(np.float32) - np.float32(padding)
    hi = vertices[:, :3].max(axis=0).astype(np.float32) + np.float32(padding)
    return lo, hi


def compute_adjusted_boundary(min_lo, min_hi, anchor, cube_side):
    """src/waveguide/src/boundary_adjust.cpp:8-22 (float arithmetic): a grid that has a node
    exactly at `anchor` and at least two cells of margin around [min_lo, min_hi]."""
    lo = np.asarray(min_lo, dtype=np.float32)
    hi = np.asarray(min_hi, dtype=np.float32)
    anchor = np.asarray(anchor, dtype=np.float32)
    side = np.float32(cube_side)
    ceiled = np.ceil((anchor - lo) / side).astype(np.int32)
    c0 = anchor - (ceiled + 2).astype(np.float32) * side
    dim = np.ceil((hi - c0) / side).astype(np.int32) + 2
    c1 = c0 + dim.astype(np.float32) * side
    return c0.astype(np.float32), c1.astype(np.float32)


def hall_scene(width=18.0, depth=30.0, height=11.0, stage_depth=7.0, stage_height=1.1):
    """A small concert-hall-like room for end-to-end runs: a shoebox (surface 0: plaster) with a
    raised stage block at the front (surface 1: wood).  The room is the prism extruded along x from
    the side-view polygon (y = length, z = height), so the stage is a step in the floor.
    Returns (vertices, triangles) with triangles[:, 0] the surface index."""
    side = [(0.0, 0.0), (stage_depth, 0.0), (stage_depth, -stage_height), (depth, -stage_height), (depth, height),
            (0.0, height)]
    # prism_scene extrudes an xy polygon along z; build it there and rotate axes (x,y,z) <- (z,x,y)
    v, t = prism_scene(side, 0.0, width)
    out = v.copy()
    out[:, 0], out[:, 1], out[:, 2] = v[:, 2], v[:, 0], v[:, 1]
    t = t.copy()
    for k in range(t.shape[0]):
        p = out[t[k, 1:], :3]
        # stage = the triangles lying on the raised floor (z = 0, y <= stage_depth) or its riser
        on_top = np.all(np.abs(p[:, 2]) < 1e-6) and np.all(p[:, 1] <= stage_depth + 1e-6)
        on_riser = np.all(np.abs(p[:, 1] - stage_depth) < 1e-6) and np.all(p[:, 2] <= 1e-6)
        t[k, 0] = 1 if (on_top or on_riser) else 0
    return out, t

"""Host-side mesh containers: the data contract `waveguide::run` consumes.

numpy mirrors of the reference's device structs (paths relative to /root/reference/):
  condensed_node          src/waveguide/include/waveguide/cl/structs.h:19-22       8 B
  boundary_data           cl/structs.h:38-41                                      56 B
  coefficients_canonical  cl/filter_structs.h:39-44,65-66                        112 B
  boundary_type bits      cl/utils.h:11-21
  error_code bits         cl/structs.h:8-15
plus the synthetic box mesh of SURVEY.md 8(d) (what `compute_mesh` yields for a
`geo::box` scene up to padding thickness) and the small coefficient helpers the reference's
tests use to make inputs.  No compute happens here; the hot path lives in csrc/.
"""
import math

import numpy as np

# boundary_type bits
ID_NONE = 0
ID_INSIDE = 1 << 0
ID_NX = 1 << 1
ID_PX = 1 << 2
ID_NY = 1 << 3
ID_PY = 1 << 4
ID_NZ = 1 << 5
ID_PZ = 1 << 6
ID_REENTRANT = 1 << 7

# error_code bits
ERR_INF = 1 << 0
ERR_NAN = 1 << 1
ERR_OUTSIDE_RANGE = 1 << 2
ERR_OUTSIDE_MESH = 1 << 3
ERR_SUSPICIOUS_BOUNDARY = 1 << 4

CANONICAL_ORDER = 6

condensed_node_dtype = np.dtype(
    [("boundary_type", "<i4"), ("boundary_index", "<u4")], align=True)
boundary_data_dtype = np.dtype(
    {"names": ["filter_memory", "coefficient_index"],
     "formats": [("<f8", (CANONICAL_ORDER,)), "<u4"],
     "offsets": [0, 48], "itemsize": 56})
coefficients_dtype = np.dtype(
    [("b", "<f8", (CANONICAL_ORDER + 1,)), ("a", "<f8", (CANONICAL_ORDER + 1,))], align=True)

assert condensed_node_dtype.itemsize == 8
assert boundary_data_dtype.itemsize == 56
assert coefficients_dtype.itemsize == 112


class Mesh:
    """`waveguide::mesh` = descriptor + vectors (mesh.h:12-26, setup.h:27-48).

    nodes         condensed_node[nx*ny*nz], index = x + y*nx + z*nx*ny
    coefficients  coefficients_canonical[num_surfaces]
    bidx1/2/3     uint32[n_D, D]: per boundary node, the surface (coefficient) index of each
                  of its D filters (`boundary_index_array<D>`, cl/boundary_index_array.h:8-11)
    """

    def __init__(self, dims, nodes, coefficients, bidx1, bidx2, bidx3,
                 spacing=0.05, min_corner=(0.0, 0.0, 0.0)):
        self.dims = tuple(int(d) for d in dims)
        self.nodes = np.ascontiguousarray(nodes, dtype=condensed_node_dtype)
        self.coefficients = np.ascontiguousarray(coefficients, dtype=coefficients_dtype)
        self.bidx = [np.ascontiguousarray(b, dtype=np.uint32).reshape(-1, d + 1)
                     for d, b in enumerate((bidx1, bidx2, bidx3))]
        self.spacing = float(spacing)
        self.min_corner = tuple(float(c) for c in min_corner)
        nx, ny, nz = self.dims
        if self.nodes.shape != (nx * ny * nz,):
            raise ValueError("nodes must have nx*ny*nz entries")

    @property
    def num_nodes(self):
        return self.dims[0] * self.dims[1] * self.dims[2]

    def set_coefficients(self, c):
        """mesh::set_coefficients (setup.cpp:38-50): one for all, or a same-sized vector."""
        c = np.asarray(c, dtype=coefficients_dtype)
        if c.ndim == 0:
            self.coefficients[:] = c
        else:
            if c.shape != self.coefficients.shape:
                raise ValueError("Size of new coefficients vector must be equal to the existing one")
            self.coefficients = np.ascontiguousarray(c)

    def compute_index(self, x, y, z):
        """mesh_descriptor.cpp:7-10"""
        nx, ny, _ = self.dims
        return int(x) + int(y) * nx + int(z) * nx * ny

    def compute_locator(self, index):
        """mesh_descriptor.cpp:16-20"""
        nx, ny, nz = self.dims
        x = index % nx
        q = index // nx
        return (x, q % ny, (q // ny) % nz)

    def compute_neighbors(self, index):
        """mesh_descriptor.cpp:36-57: ports nx,px,ny,py,nz,pz; ~0u when off-grid."""
        x, y, z = self.compute_locator(index)
        nx, ny, nz = self.dims
        out = []
        for dx, dy, dz in ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)):
            a, b, c = x + dx, y + dy, z + dz
            if 0 <= a < nx and 0 <= b < ny and 0 <= c < nz:
                out.append(self.compute_index(a, b, c))
            else:
                out.append(0xFFFFFFFF)
        return out

    def boundary_data(self, d):
        """get_boundary_data<D> (setup.h:68-85): zeroed filter state + coefficient index."""
        idx = self.bidx[d - 1]
        out = np.zeros((idx.shape[0], d), dtype=boundary_data_dtype)
        out["coefficient_index"] = idx
        return out

    def sample_rate(self, speed_of_sound=340.0):
        """compute_sample_rate (mesh_descriptor.cpp:72-74; config.cpp:19-21)."""
        return 1.0 / (self.spacing / (speed_of_sound * math.sqrt(3.0)))


def box_node_types(nx, ny, nz, z_begin=0, z_count=None):
    """boundary_type of every node of the synthetic box, planes
    [z_begin, z_begin+z_count) of a global nx*ny*nz grid, as int32[z_count, ny, nx]."""
    if z_count is None:
        z_count = nz - z_begin

    def axis_bits(n, coords, lo_bit, hi_bit):
        # lo_bit: the inside neighbour lies in +axis direction (coord==1 -> id_p*)
        bits = np.zeros(coords.shape, dtype=np.int32)
        bits[coords == 1] = lo_bit
        bits[coords == n - 2] |= hi_bit
        return bits

    xs = np.arange(nx)
    ys = np.arange(ny)
    zs = np.arange(z_begin, z_begin + z_count)
    bx = axis_bits(nx, xs, ID_PX, ID_NX)[None, None, :]
    by = axis_bits(ny, ys, ID_PY, ID_NY)[None, :, None]
    bz = axis_bits(nz, zs, ID_PZ, ID_NZ)[:, None, None]
    none = ((xs == 0) | (xs == nx - 1))[None, None, :] | \
           ((ys == 0) | (ys == ny - 1))[None, :, None] | \
           ((zs == 0) | (zs == nz - 1))[:, None, None]
    t = bx | by | bz
    t = np.where(t == 0, np.int32(ID_INSIDE), t)
    t = np.where(none, np.int32(ID_NONE), t)
    return np.ascontiguousarray(t, dtype=np.int32)


def number_boundaries(types_flat):
    """set_boundary_index (boundary_coefficient_finder.cpp:11-19): running count per
    dimensionality in increasing node index.  Returns (boundary_index uint32[n], n1, n2, n3)."""
    t = types_flat
    pc = np.zeros(t.shape, dtype=np.int32)
    for bit in range(8):
        pc += (t >> bit) & 1
    not_inside = (t & (ID_INSIDE | ID_REENTRANT)) == 0
    bindex = np.zeros(t.shape, dtype=np.uint32)
    counts = []
    for d in (1, 2, 3):
        sel = (pc == d) & not_inside
        n = int(sel.sum())
        bindex[sel] = np.arange(n, dtype=np.uint32)
        counts.append(n)
    return bindex, counts[0], counts[1], counts[2]


def box_mesh(nx, ny, nz, coefficients=None, surface_of_face=None, spacing=0.05):
    """Synthetic box mesh.  `surface_of_face` maps the 6 faces (order nx,px,ny,py,nz,pz =
    the wall a filter of that inner direction... see below) to a coefficient index; default 0.

    The d-th filter of a boundary node belongs to its d-th inner direction (x before y before
    z, program.cpp:19-87); a node whose inner direction is port p sits on the wall *opposite*
    to p, and takes surface_of_face[p].
    """
    if min(nx, ny, nz) < 5:
        raise ValueError("box needs at least 5 nodes per axis")
    types = box_node_types(nx, ny, nz).reshape(-1)
    bindex, n1, n2, n3 = number_boundaries(types)
    nodes = np.zeros(types.shape, dtype=condensed_node_dtype)
    nodes["boundary_type"] = types
    nodes["boundary_index"] = bindex
    if coefficients is None:
        coefficients = np.array([flat_coefficients(0.1)], dtype=coefficients_dtype)
    if surface_of_face is None:
        surface_of_face = [0] * 6
    sof = np.asarray(surface_of_face, dtype=np.uint32)

    pc = np.zeros(types.shape, dtype=np.int32)
    for bit in range(8):
        pc += (types >> bit) & 1
    not_inside = (types & (ID_INSIDE | ID_REENTRANT)) == 0
    bidx = []
    for d in (1, 2, 3):
        t = types[(pc == d) & not_inside]
        arr = np.zeros((t.shape[0], d), dtype=np.uint32)
        # ports in order nx(0) px(1) ny(2) py(3) nz(4) pz(5): bit (1 << (p+1))
        slot = np.zeros(t.shape[0], dtype=np.int64)
        for p in range(6):
            has = (t & (1 << (p + 1))) != 0
            rows = np.nonzero(has)[0]
            arr[rows, slot[rows]] = sof[p]
            slot[rows] += 1
        bidx.append(arr)
    return Mesh((nx, ny, nz), nodes, coefficients, bidx[0], bidx[1], bidx[2], spacing=spacing)


def mesh_from_nodes(dims, nodes, counts, coefficients, surface_of_port=None, spacing=0.05):
    """Mesh around already-classified nodes (wv_classify_nodes / the reference's
    set_node_boundary_type + set_boundary_index).  The filter of inner direction p (port order
    nx,px,ny,py,nz,pz) of every boundary node takes surface `surface_of_port[p]` -- a stand-in
    for the closest-triangle lookup of boundary_coefficient_finder.
    counts[0] includes re-entrant nodes, which carry an (unused) 1-D slot like in the reference."""
    if surface_of_port is None:
        surface_of_port = [0] * 6
    sof = np.asarray(surface_of_port, dtype=np.uint32)
    t = nodes["boundary_type"]
    k = nodes["boundary_index"]
    pc = np.zeros(t.shape, dtype=np.int32)
    for bit in range(8):
        pc += (t >> bit) & 1
    is_b = (t & (ID_INSIDE | ID_REENTRANT)) == 0
    bidx = []
    for d in (1, 2, 3):
        arr = np.zeros((counts[d - 1], d), dtype=np.uint32)
        sel = np.nonzero((pc == d) & is_b)[0]
        slot = np.zeros(sel.shape[0], dtype=np.int64)
        for p in range(6):
            has = (t[sel] & (1 << (p + 1))) != 0
            rows = np.nonzero(has)[0]
            arr[k[sel][rows], slot[rows]] = sof[p]
            slot[rows] += 1
        bidx.append(arr)
    return Mesh(dims, nodes, coefficients, bidx[0], bidx[1], bidx[2], spacing=spacing)


def room_mask(shape, kind, seed=0):
    """Inside masks [nz, ny, nx] of a few non-box rooms for tests: 'L' (L-shaped prism),
    'sphere', 'blob' (sphere + box + seeded speckle: every boundary type incl. re-entrant)."""
    nz, ny, nx = shape
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    margin = (x > 1) & (y > 1) & (z > 1) & (x < nx - 2) & (y < ny - 2) & (z < nz - 2)
    if kind == "L":
        m = margin & ~((x >= nx // 2) & (y >= ny // 2))
    elif kind == "sphere":
        m = (x - (nx - 1) / 2) ** 2 + (y - (ny - 1) / 2) ** 2 + (z - (nz - 1) / 2) ** 2 < (min(shape) / 2 - 2.5) ** 2
    elif kind == "blob":
        rng = np.random.default_rng(seed)
        m = ((x - nx / 2) ** 2 + (y - ny / 2) ** 2 + (z - nz / 2) ** 2 < (min(shape) / 2.6) ** 2) | \
            ((x > 2) & (x < nx - 3) & (y > 2) & (y < ny // 2) & (z > 2) & (z < nz - 3))
        m = m & margin
        m = m ^ ((rng.random(shape) < 0.02) & margin)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(m)


# ---- coefficient helpers (inputs for tests / benches; host-side, run once) ------------------

def make_coefficients(b, a):
    out = np.zeros((), dtype=coefficients_dtype)
    out["b"][:len(b)] = b
    out["a"][:len(a)] = a
    return out


def to_impedance_coefficients(b, a):
    """fitted_boundary.h:21-48: b' = a + b, a' = a - b, both scaled by 1/a'[0] when non-zero."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    rb = a + b
    ra = a - b
    if ra[0] != 0:
        norm = 1.0 / ra[0]
        rb = rb * norm
        ra = ra * norm
    return make_coefficients(rb, ra)


def flat_coefficients(absorption):
    """to_flat_coefficients (fitted_boundary.h:72-75): reflectance sqrt(1-absorption)
    (core/surfaces.h:25-33) as a zero-order filter, converted to impedance form."""
    r = math.sqrt(1.0 - absorption)
    b = np.zeros(CANONICAL_ORDER + 1)
    a = np.zeros(CANONICAL_ORDER + 1)
    b[0] = r
    a[0] = 1.0
    return to_impedance_coefficients(b, a)


def rigid_coefficients():
    """Perfectly reflecting wall: reflectance 1 -> b0 = 2, a0 = 0 (un-normalised)."""
    return flat_coefficients(0.0)


def peak_biquad(gain_db, centre, q):
    """get_peak_coefficients (src/waveguide/src/filters.cpp:10-21) -> (b[3], a[3])."""
    A = 10.0 ** ((gain_db / 2.0) / 20.0)
    w0 = 2.0 * math.pi * centre
    cw0 = math.cos(w0)
    sw0 = math.sin(w0)
    alpha = sw0 / 2.0 * q
    a0 = 1 + alpha / A
    b = np.array([(1 + (alpha * A)) / a0, (-2 * cw0) / a0, (1 - alpha * A) / a0])
    a = np.array([1.0, (-2 * cw0) / a0, (1 - alpha / A) / a0])
    return b, a


def convolve_sections(sections):
    """convolve (filters.h:45-57; filters.cpp:28-32): polynomial product of the sections,
    folded left to right."""
    b, a = sections[0]
    for sb, sa in sections[1:]:
        nb = np.zeros(len(b) + len(sb) - 1)
        na = np.zeros(len(a) + len(sa) - 1)
        for i in range(len(b)):
            for j in range(len(sb)):
                nb[i + j] += b[i] * sb[j]
                na[i + j] += a[i] * sa[j]
        b, a = nb, na
    return b, a


def random_peak_filter_coefficients(rng, n):
    """n order-6 impedance filters from seeded random peak-biquad descriptors
    (gain U(0.1,1) dB, centre U(0,0.5), Q U(0,1); tests/rectangular_kernel.cpp:105-166)."""
    out = np.zeros(n, dtype=coefficients_dtype)
    for k in range(n):
        secs = [peak_biquad(rng.uniform(0.1, 1.0), rng.uniform(0.0, 0.5), rng.uniform(0.0, 1.0))
                for _ in range(3)]
        b, a = convolve_sections(secs)
        out[k] = to_impedance_coefficients(b, a)
    return out


def rectilinear_calibration_factor(grid_spacing, acoustic_impedance):
    """calibration.h:20-31."""
    return math.sqrt(acoustic_impedance / (4 * math.pi)) / (0.3405 * grid_spacing)


def bench_materials():
    """The wall materials of bench.py / smoke: two flat absorbers and two frequency-dependent
    order-6 impedance filters (seeded), so the boundary kernel runs real IIR recursions."""
    out = np.zeros(4, dtype=coefficients_dtype)
    out[0] = flat_coefficients(0.1)
    out[1] = flat_coefficients(0.35)
    out[2:] = passive_peak_filter_coefficients(np.random.default_rng(2016), 2)
    return out


def passive_peak_filter_coefficients(rng, n, sections=3, scale=0.9):
    """n stable, energy-absorbing order-6 impedance filters for tests: peak-biquad cascades with
    negative gain (U(-3,-0.1) dB), reflectance scaled by `scale` < 1.  With sections < 3 the
    trailing taps are exactly zero, which exercises the reference's zero-coefficient guards
    (src/waveguide/src/cl/filters.cpp:28-29)."""
    out = np.zeros(n, dtype=coefficients_dtype)
    for k in range(n):
        secs = [peak_biquad(-rng.uniform(0.1, 3.0), rng.uniform(0.0, 0.5), rng.uniform(0.0, 1.0))
                for _ in range(sections)]
        b, a = convolve_sections(secs)
        out[k] = to_impedance_coefficients(np.pad(b * scale, (0, CANONICAL_ORDER + 1 - len(b))),
                                           np.pad(a, (0, CANONICAL_ORDER + 1 - len(a))))
    return out

"""z-slab decomposition of a rectilinear waveguide mesh across ranks (one rank per GPU).

New design -- the reference is single-device.  Memory order is x-fastest,
z-slowest (src/waveguide/src/cl/utils.cpp:33-36), so slab [z0, z1) of every per-node array is one
contiguous range and a ghost layer is one contiguous nx*ny plane; the stencil and every boundary
helper touch only the 6 axial neighbours (program.cpp:178-249,393-412), so one ghost plane per
side suffices.  boundary_index is monotone in node index
(boundary_coefficient_finder.cpp:11-19), so a slab's boundary nodes are renumbered from 0 in the
same order.

This module is host logic only (which planes, which indices, who owns a source / receiver); the
exchange itself is RCCL inside the engine (csrc/comm.cpp).  `exchange_ghosts_host` is the same
protocol over torch.distributed for CPU (gloo) tests of the decomposition.
"""
import numpy as np

from . import mesh as M


class SlabLayout:
    """Planes [z0, z1) of a global (nx, ny, nz) mesh owned by `rank` of `nranks`, plus ghosts."""

    def __init__(self, dims, rank, nranks):
        nx, ny, nz = dims
        if nranks > nz:
            raise ValueError("more ranks than z-planes")
        self.global_dims = (nx, ny, nz)
        self.rank, self.nranks = rank, nranks
        base, extra = divmod(nz, nranks)
        self.z0 = rank * base + min(rank, extra)
        self.z1 = self.z0 + base + (1 if rank < extra else 0)
        self.ghost_lo = rank > 0
        self.ghost_hi = rank < nranks - 1
        self.zl0 = self.z0 - (1 if self.ghost_lo else 0)   # first local plane (global z)
        self.zl1 = self.z1 + (1 if self.ghost_hi else 0)
        self.local_dims = (nx, ny, self.zl1 - self.zl0)
        self.plane = nx * ny

    def owns_z(self, z):
        return self.z0 <= z < self.z1

    def holds_z(self, z):
        return self.zl0 <= z < self.zl1

    def to_local(self, global_index):
        """Local node index of a global node, or None if this rank does not hold its plane."""
        z = global_index // self.plane
        if not self.holds_z(z):
            return None
        return global_index - self.zl0 * self.plane

    def owned_local_range(self):
        lo = (self.z0 - self.zl0) * self.plane
        return lo, lo + (self.z1 - self.z0) * self.plane


def slab_mesh(global_mesh, layout):
    """Cut a global Mesh down to one rank's slab (owned planes + ghost planes).  Ghost-plane
    nodes keep their true boundary_type (the static neighbour-type checks need it); boundary
    arrays hold the owned planes' nodes only, renumbered from 0 in node-index order."""
    L = layout
    plane = L.plane
    nodes = global_mesh.nodes[L.zl0 * plane:L.zl1 * plane].copy()
    t = nodes["boundary_type"]
    pc = np.zeros(t.shape, dtype=np.int32)
    for bit in range(8):
        pc += (t >> bit) & 1
    is_b = (t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0
    owned = np.zeros(t.shape, dtype=bool)
    lo, hi = L.owned_local_range()
    owned[lo:hi] = True
    bidx = []
    for d in (1, 2, 3):
        sel_owned = (pc == d) & is_b & owned
        sel_ghost = (pc == d) & is_b & ~owned
        old = nodes["boundary_index"][sel_owned]
        bidx.append(global_mesh.bidx[d - 1][old])
        nodes["boundary_index"][sel_owned] = np.arange(int(sel_owned.sum()), dtype=np.uint32)
        nodes["boundary_index"][sel_ghost] = 0
    return M.Mesh(L.local_dims, nodes, global_mesh.coefficients, bidx[0], bidx[1], bidx[2],
                  spacing=global_mesh.spacing, min_corner=global_mesh.min_corner)


def box_slab_mesh(nx, ny, nz_global, layout, coefficients=None, make_nodes=None):
    """One rank's slab of the synthetic box without ever materialising the global mesh
    (the 1024x1024x8192 mesh of BASELINE configs[3] has 2^33 nodes: no 32-bit global index).
    With one coefficient set all walls use it; with several, filter j of boundary node k takes
    set (k*D + j) mod n -- a deterministic mix of materials over the walls.
    `make_nodes` = wayverb_amd.engine.make_box_nodes."""
    L = layout
    if make_nodes is None:
        from .engine import make_box_nodes as make_nodes
    nodes, counts = make_nodes(nx, ny, nz_global, z_begin=L.zl0, z_count=L.zl1 - L.zl0,
                               number_from=L.z0, number_to=L.z1)
    if coefficients is None:
        coefficients = np.array([M.flat_coefficients(0.1)], dtype=M.coefficients_dtype)
    n_sets = int(np.asarray(coefficients).shape[0])
    bidx = [(np.arange(counts[d] * (d + 1), dtype=np.uint32) % np.uint32(n_sets)).reshape(counts[d], d + 1)
            for d in range(3)]
    return M.Mesh(L.local_dims, nodes, coefficients, bidx[0], bidx[1], bidx[2])


def place_source_and_receivers(layout, source_node, receivers):
    """Which rank injects / records what.  The source is injected by every rank that HOLDS its
    plane (owner and, when it lies on a slab face, the neighbour's ghost copy: both apply the same
    arithmetic to the same value, so the copies stay identical).  A receiver is recorded by the
    rank that OWNS it.  Returns (local_source or None, [(position, local_index), ...])."""
    src = None if source_node is None else layout.to_local(source_node)
    mine = []
    for pos, r in enumerate(receivers):
        if layout.owns_z(r // layout.plane):
            mine.append((pos, layout.to_local(r)))
    return src, mine


def exchange_ghosts_host(field, layout, dist, tag=0):
    """Ghost-plane exchange of `field` (numpy [nz_local*ny*nx], updated in place) over
    torch.distributed point-to-point -- the protocol csrc/comm.cpp runs over RCCL:
    my first owned plane -> lower neighbour's top ghost, my last owned plane -> upper
    neighbour's bottom ghost."""
    import torch
    L = layout
    plane = L.plane
    nzl = L.local_dims[2]
    f = field.reshape(nzl, plane)
    ops = []
    recv_lo = recv_hi = None
    if L.ghost_lo:
        send = torch.from_numpy(np.ascontiguousarray(f[1]))
        recv_lo = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, L.rank - 1), dist.P2POp(dist.irecv, recv_lo, L.rank - 1)]
    if L.ghost_hi:
        send = torch.from_numpy(np.ascontiguousarray(f[nzl - 2]))
        recv_hi = torch.empty_like(send)
        ops += [dist.P2POp(dist.isend, send, L.rank + 1), dist.P2POp(dist.irecv, recv_hi, L.rank + 1)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if recv_lo is not None:
        f[0] = recv_lo.numpy()
    if recv_hi is not None:
        f[nzl - 1] = recv_hi.numpy()

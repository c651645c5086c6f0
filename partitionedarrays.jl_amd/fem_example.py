"""test/fem_example.jl on the device path (BASELINE config 5 names this file).

The reference's example: Q1 finite elements for -Δu = 0 on a 2-D Cartesian grid with Dirichlet values u = x + y, cells
partitioned in blocks WITH a ghost layer, free dofs (interior nodes) numbered part by part -- a dof belongs to the
largest-numbered part among the owners of the cells that touch it -- through `variable_partition`, the global ids of the
ghost cells' dofs fetched by a `consistent!` on a PVector of JaggedArrays, then cell-wise COO assembly on own cells,
`psparse` / `pvector` with the default (disassembled) flags, CG, and the re-assembly / `psystem` / sub-assembled variants
(test/fem_example.jl:261-343).

Here the set-up loops (setup_grid :31-67, setup_space :69-113, setup_cell_dofs :115-139, finish_cell_dofs :141-168,
setup_IJV :170-198, setup_b :200-236, setup_exact_solution :238-259) are restated with numpy array operations, one
statement per reference loop, so that the example also runs at benchmark sizes; `oracle/pa_oracle.py::fem_example_setup`
restates the same loops literally and the tests require both to agree entry for entry.  Everything after the set-up is
the library: psparse_disassembled / pvector_disassembled / psystem, mul!, the CG loop.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .primitives import ExchangeGraph, exchange, getany, pmap, reduction
from .p_range import assembly_local_indices, assembly_neighbors, uniform_partition, variable_partition

I64 = np.int64
ELEMENT_NODES = ((0, 0), (1, 0), (0, 1), (1, 1))        # CartesianIndices((2,2)) in linear order (:38)


def setup_params(parts_per_dir=(2, 2), cells_per_dir=(10, 10), length_in_x=2.0):
    """setup_params (:13-29); sizes are arguments instead of the constants 2 / 10."""
    length_per_dir = (length_in_x, length_in_x)
    h = max(l / c for l, c in zip(length_per_dir, cells_per_dir))
    Ae = (h ** 2 / 6) * np.array([[4.0, -1.0, -1.0, -2.0], [-1.0, 4.0, -2.0, -1.0], [-1.0, -2.0, 4.0, -1.0],
                                  [-2.0, -1.0, -1.0, 4.0]])
    return dict(parts_per_dir=tuple(parts_per_dir), cells_per_dir=tuple(cells_per_dir),
                nodes_per_dir=tuple(c + 1 for c in cells_per_dir), h=h, Ae=Ae)


@dataclass
class PartSpace:
    """grid + space of one part (setup_grid / setup_space)."""
    part: int
    box: tuple                 # (fx, fy, ncx, ncy): first global cell (0-based) and cells per direction of the local box
    cell_owner: np.ndarray     # local_to_owner(cell_indices)
    cell_dofs_local: np.ndarray  # n_local_cells x 4 local dof of every element node, 0 on the Dirichlet boundary
    dof_owner: np.ndarray      # local_dof_to_owner
    dof_node: np.ndarray       # local_dof_to_node (0-based local node id)
    n_own_dofs: int
    cell_global_dofs: np.ndarray = None   # local_cell_to_global_dofs, n_local_cells x 4


def setup_space(cell_indices, params) -> PartSpace:
    cx, cy = params["cells_per_dir"]
    gl = cell_indices.get_local_to_global().astype(I64)
    owner = cell_indices.get_local_to_owner().astype(I64)
    fx, fy = (gl[0] - 1) % cx, (gl[0] - 1) // cx             # first / last global cell of the box (:46-51)
    lx, ly = (gl[-1] - 1) % cx, (gl[-1] - 1) // cx
    ncx, ncy = int(lx - fx + 1), int(ly - fy + 1)
    assert len(gl) == ncx * ncy
    nnx = ncx + 1
    node = np.arange(nnx * (ncy + 1))
    gx, gy = fx + node % nnx, fy + node // nnx               # global node coordinates, 0-based
    free = ~((gx == 0) | (gx == cx) | (gy == 0) | (gy == cy))  # is_boundary_node (:7-9,76-80)
    node_to_dof = np.zeros(len(node), I64)
    dof_node = np.nonzero(free)[0]                           # local_dof_to_node = findall (:84)
    node_to_dof[dof_node] = np.arange(1, len(dof_node) + 1)
    cell = np.arange(ncx * ncy)
    ci, cj = cell % ncx, cell // ncx
    cell_nodes = np.stack([(ci + di) + nnx * (cj + dj) for di, dj in ELEMENT_NODES], axis=1)
    cell_dofs_local = node_to_dof[cell_nodes]
    dof_owner = np.zeros(len(dof_node), I64)                 # max over the touching local cells (:90-101)
    for e in range(4):                                       # a node is the e-th node of at most one cell
        d = cell_dofs_local[:, e]
        m = d > 0
        dof_owner[d[m] - 1] = np.maximum(dof_owner[d[m] - 1], owner[m])
    return PartSpace(cell_indices.part, (int(fx), int(fy), ncx, ncy), owner, cell_dofs_local, dof_owner, dof_node,
                     int(np.count_nonzero(dof_owner == cell_indices.part)))


def setup_cell_dofs(space: PartSpace, dof_indices):
    """setup_cell_dofs (:115-139): own dofs get offset + their rank among the own local dofs; the rest stays 0."""
    offset = int(dof_indices.own_to_global[0]) - 1 if dof_indices.n_own else 0
    perm = np.zeros(len(space.dof_owner) + 1, I64)
    own = np.nonzero(space.dof_owner == space.part)[0]
    perm[own + 1] = np.arange(1, len(own) + 1) + offset
    space.cell_global_dofs = perm[space.cell_dofs_local]
    return space.cell_global_dofs


def consistent_cell_dofs(spaces, cell_partition):
    """consistent!(PVector(local_cell_to_global_dofs,cell_partition)) (:275-276): the rows of my ghost cells come from
    their owners.  Set-up data (Int64), moved by the host exchange primitive over the reversed assembly graph."""
    nbr_snd, nbr_rcv = assembly_neighbors(cell_partition)
    idx_snd, idx_rcv = assembly_local_indices(cell_partition, nbr_snd, nbr_rcv)
    out = pmap(lambda s, ir: [s.cell_global_dofs[ir[k].astype(I64) - 1] for k in range(len(ir))], spaces, idx_rcv)
    got = exchange(out, ExchangeGraph(nbr_rcv, nbr_snd))

    def put(s, isnd, rows):
        for k in range(len(isnd)):
            s.cell_global_dofs[isnd[k].astype(I64) - 1] = np.asarray(rows[k], I64).reshape(-1, 4)
    pmap(put, spaces, idx_snd, got)


def finish_cell_dofs(space: PartSpace):
    """finish_cell_dofs (:141-168): every local dof takes the one non-zero global id its cells carry."""
    l2g = np.zeros(len(space.dof_owner) + 1, I64)
    m = (space.cell_dofs_local > 0) & (space.cell_global_dofs > 0)
    l2g[space.cell_dofs_local[m]] = space.cell_global_dofs[m]
    space.cell_global_dofs = l2g[space.cell_dofs_local]
    assert np.all(space.cell_global_dofs[space.cell_dofs_local > 0] != 0)        # :163
    return space.cell_global_dofs


def setup_IJV(space: PartSpace, params):
    """setup_IJV (:170-198): own cells in local order, element rows then element columns, boundary dofs skipped."""
    g = space.cell_global_dofs[space.cell_owner == space.part]
    rows, cols = np.repeat(g, 4, axis=1), np.tile(g, (1, 4))            # (er, ec) in row-major order er*4+ec
    vals = np.broadcast_to(params["Ae"].reshape(1, 16), rows.shape)
    keep = (rows > 0) & (cols > 0)
    return rows[keep], cols[keep], np.ascontiguousarray(vals[keep])


def _own_cell_coords(space: PartSpace, params):
    fx, fy, ncx, _ = space.box
    cell = np.nonzero(space.cell_owner == space.part)[0]
    ci, cj = cell % ncx, cell // ncx
    gx = np.stack([fx + ci + di for di, _ in ELEMENT_NODES], axis=1)
    gy = np.stack([fy + cj + dj for _, dj in ELEMENT_NODES], axis=1)
    return cell, (gx * params["h"]) + (gy * params["h"])                # u(x) = x[1] + x[2] at the element nodes (:11)


def setup_b(space: PartSpace, params):
    """setup_b (:200-236): -Ae*ue per own cell, ue = Dirichlet values on the boundary nodes, 0 elsewhere."""
    cell, uval = _own_cell_coords(space, params)
    free = space.cell_dofs_local[cell] > 0
    ue = np.where(free, 0.0, uval)
    Ae = params["Ae"]
    ge = ((Ae[:, 0] * ue[:, [0]] + Ae[:, 1] * ue[:, [1]]) + Ae[:, 2] * ue[:, [2]]) + Ae[:, 3] * ue[:, [3]]
    g = space.cell_global_dofs[cell]
    keep = g > 0
    return g[keep], -ge[keep]


def setup_exact_solution(space: PartSpace, params, col_indices):
    """setup_exact_solution (:238-259) as a local vector on col_indices (entries no own cell touches stay 0)."""
    cell, uval = _own_cell_coords(space, params)
    g = space.cell_global_dofs[cell]
    keep = space.cell_dofs_local[cell] > 0
    out = np.zeros(col_indices.n_local)
    lid = col_indices.global_to_local(g[keep]).astype(I64)
    out[lid[lid > 0] - 1] = uval[keep][lid > 0]          # (a partition without the ghost dofs keeps the own ones)
    return out


def fem_example_system(ranks, parts_per_dir=(2, 2), cells_per_dir=(10, 10)):
    """fem_example(distribute) up to the COO data (:261-280): returns I, J, V, II, VV, dof_partition and what
    setup_exact_solution needs (spaces, params, cell_partition)."""
    params = setup_params(parts_per_dir, cells_per_dir)
    cell_partition = uniform_partition(ranks, params["parts_per_dir"], params["cells_per_dir"], (True, True))
    spaces = pmap(lambda c: setup_space(c, params), cell_partition)
    n_own = pmap(lambda s: s.n_own_dofs, spaces)
    n_global = int(getany(reduction(lambda a, b: a + b, n_own, init=0, destination="all")))
    dof_partition = variable_partition(n_own, n_global)
    pmap(setup_cell_dofs, spaces, dof_partition)
    consistent_cell_dofs(spaces, cell_partition)
    pmap(finish_cell_dofs, spaces)
    ijv = pmap(lambda s: setup_IJV(s, params), spaces)
    iv = pmap(lambda s: setup_b(s, params), spaces)
    I, J, V = (pmap(lambda t, k=k: t[k], ijv) for k in range(3))
    II, VV = (pmap(lambda t, k=k: t[k], iv) for k in range(2))
    return dict(I=I, J=J, V=V, II=II, VV=VV, dof_partition=dof_partition, spaces=spaces, params=params,
                cell_partition=cell_partition, n_global_dofs=n_global)

"""Ceres' own linear-system dumps, read and written.

Solver::Options::trust_region_minimizer_iterations_to_dump makes TrustRegionMinimizer write the linear least squares
problem of those iterations (trust_region_minimizer.cc -> DumpLinearLeastSquaresProblem,
linear_least_squares_problems.cc:929-1043) with trust_region_problem_dump_format_type = TEXTFILE as
    <base>_A.txt   one "% 10d % 10d %17f" line per stored entry: row, column, value -- in the order
                   BlockSparseMatrix::ToTextFile walks them (block_sparse_matrix.cc:558-580): row blocks in order, the
                   cells of a row block in order, every cell row-major;
    <base>_D.txt   the LM diagonal, one %17f per line (only when D was given)
    <base>_b.txt   the right-hand side
    <base>_x.txt   the solution the reference's linear solver produced
    <base>.m       a Matlab loader with num_rows / num_cols.
It is the one way real Ceres systems -- and the reference's own answers -- can be brought to this library offline (no
BAL file exists in the image): read_dump() rebuilds the block structure from the entry order (a cell is a maximal
row-major rectangle), so the system goes through cx_matrix_create / cx_solver_solve like any BlockSparseMatrix.
The text carries six decimals (%17f): the values are the reference's rounded to 1e-6, and so is its x.

write_dump() writes the same files (used by tests/golden/make_ceres_dump.py for the committed fixture).
"""
import os
import re

import numpy as np

from .structure import BlockStructure


def write_dump(base, bs, values, b=None, D=None, x=None):
    """DumpLinearLeastSquaresProblemToTextFile (linear_least_squares_problems.cc:966-1022)."""
    values = np.asarray(values, dtype=np.float64)
    with open(base + "_A.txt", "w") as f:
        for r in range(bs.num_row_blocks):
            rb = bs.row_blocks[r]
            for c in range(int(bs.row_cell_begin[r]), int(bs.row_cell_begin[r + 1])):
                cell = bs.cells[c]
                cb = bs.col_blocks[int(cell["block_id"])]
                pos = int(cell["position"])
                for i in range(int(rb["size"])):
                    for j in range(int(cb["size"])):
                        f.write("% 10d % 10d %17f\n" % (int(rb["position"]) + i, int(cb["position"]) + j, values[pos]))
                        pos += 1
    script = ["function lsqp = load_trust_region_problem()", "lsqp.num_rows = %d;" % bs.num_rows, "lsqp.num_cols = %d;" % bs.num_cols,
              "tmp = load('%s_A.txt', '-ascii');" % base,
              "lsqp.A = sparse(tmp(:, 1) + 1, tmp(:, 2) + 1, tmp(:, 3), %d, %d);" % (bs.num_rows, bs.num_cols)]
    for name, vec in (("D", D), ("b", b), ("x", x)):
        if vec is None:
            continue
        with open("%s_%s.txt" % (base, name), "w") as f:
            for v in np.asarray(vec, dtype=np.float64):
                f.write("%17f\n" % v)
        script.append("lsqp.%s = load('%s_%s.txt', '-ascii');" % (name, base, name))
    with open(base + ".m", "w") as f:
        f.write("\n".join(script) + "\n")


def _cells_from_entries(rows, cols):
    """Cut the entry stream into cells: maximal row-major rectangles.  Returns arrays (r0, c0, h, w, position)."""
    n = rows.size
    out = []
    k = 0
    while k < n:
        r0, c0 = int(rows[k]), int(cols[k])
        w = 1
        while k + w < n and rows[k + w] == r0 and cols[k + w] == c0 + w:
            w += 1
        h = 1
        while True:
            s = k + h * w
            if s + w > n:
                break
            if rows[s] != r0 + h or cols[s] != c0 or rows[s + w - 1] != r0 + h or cols[s + w - 1] != c0 + w - 1:
                break
            h += 1
        out.append((r0, c0, h, w, k))
        k += h * w
    return out


def read_dump(base, num_rows=None, num_cols=None):
    """Returns a dict: bs (BlockStructure), values, b, D, x (None when the file is absent), num_rows, num_cols.
    num_rows / num_cols default to the .m script's (or to what the entries span)."""
    if (num_rows is None or num_cols is None) and os.path.exists(base + ".m"):
        text = open(base + ".m").read()
        m_rows, m_cols = re.search(r"num_rows = (\d+);", text), re.search(r"num_cols = (\d+);", text)
        if num_rows is None and m_rows:
            num_rows = int(m_rows.group(1))
        if num_cols is None and m_cols:
            num_cols = int(m_cols.group(1))
    trip = np.loadtxt(base + "_A.txt", dtype=np.float64, ndmin=2)
    rows, cols, values = trip[:, 0].astype(np.int64), trip[:, 1].astype(np.int64), np.ascontiguousarray(trip[:, 2])
    num_rows = int(rows.max()) + 1 if num_rows is None else num_rows
    num_cols = int(cols.max()) + 1 if num_cols is None else num_cols
    cells = _cells_from_entries(rows, cols)
    # row blocks: the distinct (first row, height) of the cells, in order; gaps (rows without entries) become blocks of
    # their own.  Column blocks likewise.
    def blocks_of(pairs, total, what):
        starts = {}
        for start, size in pairs:
            if starts.setdefault(start, size) != size:
                raise ValueError("%s block at %d has two sizes (%d, %d): the entry order is not a block sparse matrix's" % (what, start, starts[start], size))
        out, at = [], 0
        for start in sorted(starts):
            if start < at:
                raise ValueError("%s blocks overlap at %d" % (what, start))
            if start > at:
                out.append((at, start - at))
            out.append((start, starts[start]))
            at = start + starts[start]
        if at < total:
            out.append((at, total - at))
        return out
    row_blocks = blocks_of([(c[0], c[2]) for c in cells], num_rows, "row")
    col_blocks = blocks_of([(c[1], c[3]) for c in cells], num_cols, "column")
    row_id = {start: i for i, (start, _) in enumerate(row_blocks)}
    col_id = {start: i for i, (start, _) in enumerate(col_blocks)}
    per_row = [[] for _ in row_blocks]
    last = -1
    for r0, c0, h, w, pos in cells:
        i = row_id[r0]
        if i < last:
            raise ValueError("row blocks are not written in ascending order")
        last = i
        per_row[i].append((col_id[c0], pos))
    bs = BlockStructure.from_rows([size for _, size in col_blocks], [(row_blocks[i][1], per_row[i]) for i in range(len(row_blocks))])

    def vec(name, n):
        path = "%s_%s.txt" % (base, name)
        if not os.path.exists(path):
            return None
        v = np.loadtxt(path, dtype=np.float64, ndmin=1)
        if v.size != n:
            raise ValueError("%s has %d entries, expected %d" % (path, v.size, n))
        return v
    return {"bs": bs, "values": values, "b": vec("b", num_rows), "D": vec("D", num_cols), "x": vec("x", num_cols),
            "num_rows": num_rows, "num_cols": num_cols}


def to_jacobian_layout(bs, values, num_eliminate_blocks):
    """The dump keeps values in ToTextFile order (row block by row block); BlockJacobianWriter lays a Schur-ordered Jacobian
    out as [all E cells in row order | all other cells in row order] (BuildJacobianLayout, block_jacobian_writer.cc:68-167),
    which is the layout the static <2,3,9> kernels read directly.  Returns (BlockStructure, values) in that layout: same
    cells, same order inside the rows, new positions, values moved along."""
    values = np.asarray(values, dtype=np.float64)
    cells = bs.cells.copy()
    sizes = np.zeros(len(cells), dtype=np.int64)
    is_e = np.zeros(len(cells), dtype=bool)
    for r in range(bs.num_row_blocks):
        c0, c1 = int(bs.row_cell_begin[r]), int(bs.row_cell_begin[r + 1])
        for c in range(c0, c1):
            sizes[c] = int(bs.row_blocks[r]["size"]) * int(bs.col_blocks[int(cells[c]["block_id"])]["size"])
        if c1 > c0 and int(cells[c0]["block_id"]) < num_eliminate_blocks:
            is_e[c0] = True
    order = np.concatenate([np.flatnonzero(is_e), np.flatnonzero(~is_e)])
    new_pos = np.zeros(len(cells), dtype=np.int64)
    new_pos[order] = np.concatenate([[0], np.cumsum(sizes[order])[:-1]])
    out = np.empty_like(values)
    for c in range(len(cells)):
        p = int(cells[c]["position"])
        out[new_pos[c]:new_pos[c] + sizes[c]] = values[p:p + sizes[c]]
    cells["position"] = new_pos.astype(np.int32)
    return BlockStructure(bs.row_blocks.copy(), bs.col_blocks.copy(), bs.row_cell_begin.copy(), cells), out


def leading_eliminate_blocks(bs):
    """How many leading column blocks form the e-blocks of a Schur-ordered Jacobian: the longest prefix of column blocks
    such that every row block that touches the prefix does so with its FIRST cell only (what
    LexicographicallyOrderResidualBlocks + BuildJacobianLayout leave, reorder_program.cc:256-338).  The dump does not
    record num_eliminate_blocks (the TEXTFILE writer ignores it, linear_least_squares_problems.cc:966-971)."""
    limit = bs.num_col_blocks
    for r in range(bs.num_row_blocks):
        c0, c1 = int(bs.row_cell_begin[r]), int(bs.row_cell_begin[r + 1])
        for c in range(c0 + 1, c1):
            limit = min(limit, int(bs.cells[c]["block_id"]))
    # rows whose first cell is past the prefix end it too only if they come before e-rows: the prefix is what the first
    # cells of the leading rows cover
    first = [int(bs.cells[int(bs.row_cell_begin[r])]["block_id"]) for r in range(bs.num_row_blocks) if bs.row_cell_begin[r + 1] > bs.row_cell_begin[r]]
    return min(limit, (max([f for f in first if f < limit], default=-1) + 1))

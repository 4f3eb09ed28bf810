"""Flat block-CRS structure shared by the product binding and the oracle binding.

Mirrors ceres::internal::CompressedRowBlockStructure (block_structure.h:84-182)
in the flattened form of include/cxschur.h (cx_block_structure).
"""
import ctypes

import numpy as np

BLOCK_DTYPE = np.dtype([("size", np.int32), ("position", np.int32)])
CELL_DTYPE = np.dtype([("block_id", np.int32), ("position", np.int32)])


class cx_block_structure(ctypes.Structure):
    _fields_ = [
        ("num_row_blocks", ctypes.c_int32),
        ("num_col_blocks", ctypes.c_int32),
        ("row_blocks", ctypes.c_void_p),
        ("col_blocks", ctypes.c_void_p),
        ("row_cell_begin", ctypes.c_void_p),
        ("cells", ctypes.c_void_p),
    ]


class BlockStructure:
    """Owns the numpy arrays behind a cx_block_structure."""

    def __init__(self, row_blocks, col_blocks, row_cell_begin, cells):
        self.row_blocks = np.ascontiguousarray(row_blocks, dtype=BLOCK_DTYPE)
        self.col_blocks = np.ascontiguousarray(col_blocks, dtype=BLOCK_DTYPE)
        self.row_cell_begin = np.ascontiguousarray(row_cell_begin, dtype=np.int32)
        self.cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        assert self.row_cell_begin.shape[0] == self.row_blocks.shape[0] + 1
        assert self.row_cell_begin[-1] == self.cells.shape[0]
        self._c = cx_block_structure(
            self.row_blocks.shape[0],
            self.col_blocks.shape[0],
            self.row_blocks.ctypes.data,
            self.col_blocks.ctypes.data,
            self.row_cell_begin.ctypes.data,
            self.cells.ctypes.data,
        )

    @property
    def c(self):
        return ctypes.byref(self._c)

    @property
    def num_row_blocks(self):
        return int(self.row_blocks.shape[0])

    @property
    def num_col_blocks(self):
        return int(self.col_blocks.shape[0])

    @property
    def num_rows(self):
        if self.num_row_blocks == 0:
            return 0
        b = self.row_blocks[-1]
        return int(b["position"]) + int(b["size"])

    @property
    def num_cols(self):
        if self.num_col_blocks == 0:
            return 0
        b = self.col_blocks[-1]
        return int(b["position"]) + int(b["size"])

    @property
    def num_nonzeros(self):
        rs = np.repeat(self.row_blocks["size"].astype(np.int64), np.diff(self.row_cell_begin))
        cs = self.col_blocks["size"].astype(np.int64)[self.cells["block_id"]]
        return int((rs * cs).sum())

    @classmethod
    def from_rows(cls, col_sizes, rows):
        """rows: list of (row_block_size, [(col_block_id, position), ...])."""
        col_sizes = np.asarray(col_sizes, dtype=np.int32)
        cols = np.zeros(len(col_sizes), dtype=BLOCK_DTYPE)
        cols["size"] = col_sizes
        cols["position"] = np.concatenate([[0], np.cumsum(col_sizes)[:-1]]) if len(col_sizes) else []
        rb = np.zeros(len(rows), dtype=BLOCK_DTYPE)
        rcb = [0]
        cells = []
        pos = 0
        for i, (rs, cs) in enumerate(rows):
            rb[i] = (rs, pos)
            pos += rs
            cells.extend(cs)
            rcb.append(len(cells))
        cells_arr = np.array(cells, dtype=CELL_DTYPE) if cells else np.zeros(0, dtype=CELL_DTYPE)
        return cls(rb, cols, np.array(rcb, dtype=np.int32), cells_arr)

    def to_dense(self, values):
        """Dense matrix of the block-sparse matrix (BlockSparseMatrix::ToDenseMatrix,
        block_sparse_matrix.cc:494-516); numpy, for tests."""
        m = np.zeros((self.num_rows, self.num_cols))
        for r in range(self.num_row_blocks):
            rs, rp = int(self.row_blocks[r]["size"]), int(self.row_blocks[r]["position"])
            for c in range(self.row_cell_begin[r], self.row_cell_begin[r + 1]):
                b = int(self.cells[c]["block_id"])
                cs, cp = int(self.col_blocks[b]["size"]), int(self.col_blocks[b]["position"])
                p = int(self.cells[c]["position"])
                m[rp:rp + rs, cp:cp + cs] += np.asarray(values[p:p + rs * cs]).reshape(rs, cs)
        return m

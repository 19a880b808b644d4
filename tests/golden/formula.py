"""Closed-form embedding tables shared by make_golden.py (which runs the reference on them) and by the
tests (which rebuild them instead of storing them): a fixture then only holds ids, gradient rows and the
rows the reference returned / updated."""
import numpy as np


def table(rows, width):
    """float32 [rows, width]; every entry distinct enough that a wrong row or column shows."""
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(width, dtype=np.uint64)[None, :]
    v = (r * np.uint64(2654435761) + c * np.uint64(40503)) % np.uint64(2000003)
    return (v.astype(np.float32) * np.float32(1e-4) - np.float32(100.0)).astype(np.float32)


def rows_of(keys, width):
    """The same values for a list of row numbers only (no full table)."""
    r = np.asarray(keys, dtype=np.uint64)[:, None]
    c = np.arange(width, dtype=np.uint64)[None, :]
    v = (r * np.uint64(2654435761) + c * np.uint64(40503)) % np.uint64(2000003)
    return (v.astype(np.float32) * np.float32(1e-4) - np.float32(100.0)).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).reshape(-1).tolist()


def from_bits(lst, shape):
    return np.array(lst, dtype=np.uint32).view(np.float32).reshape(shape)

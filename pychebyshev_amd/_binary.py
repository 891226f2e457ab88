"""Portable ``.pcb`` binary format (v1) for ``ChebyshevApproximation``.

Byte layout (reference ``_binary.py:208-283`` and ``docs/user-guide/binary-format.md``
there), everything little-endian, no padding:

    offset  size  field
    0       4     magic  b"PCB\\x00"
    4       1     major version (1)
    5       1     minor version (0)
    6       2     class tag  (1 = ChebyshevApproximation, 2 = ChebyshevSpline)
    8       4     reserved, zero
    12      4     num_dimensions d            (uint32)
    16      8d    domain lower bounds         (float64)
    ..      8d    domain upper bounds         (float64)
    ..      4d    n_nodes                     (uint32)
    ..      8P    tensor_values, C order, P = prod(n_nodes)   (float64)

Nodes, weights and differentiation matrices are not stored; readers rebuild them
(``ChebyshevApproximation.from_values``).

``ChebyshevSpline`` (class tag 2; reference ``_binary.py:289-428``), flat ``n_nodes`` only:

    12      4     num_dimensions d            (uint32)
    ..      8d    domain lower bounds, 8d upper bounds          (float64)
    ..      4d    n_nodes per piece           (uint32)
    ..      4d    number of knots per dimension                 (uint32)
    ..      8K    all knots, dimension after dimension, K = sum of the counts   (float64)
    ..      4     num_pieces = prod(knots_d + 1)                (uint32)
    ..      8P    per piece, C order over the intervals: tensor_values, P = prod(n_nodes)
"""
from __future__ import annotations

import os
import struct

import numpy as np

MAGIC = b"PCB\x00"
MAJOR = 1
MINOR = 0
CLASS_TAG_APPROX = 1
CLASS_TAG_SPLINE = 2
HEADER_SIZE = 12


def detect_format(path) -> str:
    """``'binary'`` when the file starts with the .pcb magic, else ``'pickle'``."""
    with open(os.fspath(path), "rb") as f:
        return "binary" if f.read(4) == MAGIC else "pickle"


def peek_format_version(filename) -> int:
    """Major format version from the 12-byte header (body is not read)."""
    with open(filename, "rb") as f:
        head = f.read(HEADER_SIZE)
    if len(head) < HEADER_SIZE:
        raise ValueError(f"file {filename!r} is shorter than the {HEADER_SIZE}-byte .pcb header")
    if head[:4] != MAGIC:
        raise ValueError(f"file {filename!r} is not a .pcb file (magic mismatch: "
                         f"got {head[:4]!r}, expected {MAGIC!r})")
    return int(head[4])


def _take(f, nbytes: int, what: str) -> bytes:
    raw = f.read(nbytes)
    if len(raw) != nbytes:
        raise ValueError(f"unexpected EOF reading {what} (wanted {nbytes} bytes, got {len(raw)})")
    return raw


def _read_header(f) -> int:
    raw = _take(f, HEADER_SIZE, "header")
    if raw[:4] != MAGIC:
        raise ValueError("not a PyChebyshev binary file (bad magic)")
    major, _minor, tag = struct.unpack_from("<BBH", raw, 4)
    if major != MAJOR:
        raise ValueError(f"unsupported .pcb major version {major} (this build reads major {MAJOR})")
    if raw[8:12] != b"\x00" * 4:
        raise ValueError("reserved header bytes nonzero — file may be corrupt")
    return tag


def write_approx(f, cheb) -> None:
    """Serialise a built ``ChebyshevApproximation`` to the open binary stream ``f``."""
    if getattr(cheb, "additional_data", None) is not None:
        raise NotImplementedError("binary format cannot store additional_data; "
                                  "pass format='pickle' or set additional_data=None before saving")
    if cheb.tensor_values is None:
        raise RuntimeError("Cannot save an unbuilt ChebyshevApproximation")
    d = int(cheb.num_dimensions)
    lo = np.array([cheb.domain[k][0] for k in range(d)], dtype="<f8")
    hi = np.array([cheb.domain[k][1] for k in range(d)], dtype="<f8")
    n = np.array(cheb.n_nodes, dtype="<u4")
    tensor = np.ascontiguousarray(cheb.tensor_values, dtype="<f8")
    f.write(MAGIC + struct.pack("<BBH", MAJOR, MINOR, CLASS_TAG_APPROX) + b"\x00" * 4)
    f.write(struct.pack("<I", d))
    f.write(lo.tobytes())
    f.write(hi.tobytes())
    f.write(n.tobytes())
    f.write(tensor.tobytes(order="C"))


def read_approx(f):
    """Parse a ``.pcb`` stream into a ``ChebyshevApproximation`` (via ``from_values``)."""
    from .barycentric import ChebyshevApproximation

    tag = _read_header(f)
    if tag != CLASS_TAG_APPROX:
        raise ValueError(f"file contains class_tag {tag}, expected {CLASS_TAG_APPROX} "
                         f"(ChebyshevApproximation)")
    d = struct.unpack("<I", _take(f, 4, "uint32"))[0]
    if d < 1:
        raise ValueError(f"num_dimensions must be >= 1, got {d}")
    lo = np.frombuffer(_take(f, 8 * d, "f64 array"), dtype="<f8")
    hi = np.frombuffer(_take(f, 8 * d, "f64 array"), dtype="<f8")
    domain = [[float(lo[k]), float(hi[k])] for k in range(d)]
    for k, (a, b) in enumerate(domain):
        if a >= b:
            raise ValueError(f"domain[{k}]: lo ({a}) must be < hi ({b})")
    n_nodes = [int(v) for v in np.frombuffer(_take(f, 4 * d, "uint32 array"), dtype="<u4")]
    for k, n in enumerate(n_nodes):
        if n < 1:
            raise ValueError(f"n_nodes[{k}] must be >= 1, got {n}")
    total = int(np.prod(n_nodes))
    tensor = np.frombuffer(_take(f, 8 * total, "f64 array"), dtype="<f8").astype(np.float64)
    return ChebyshevApproximation.from_values(tensor.reshape(tuple(n_nodes), order="C"), d, domain, n_nodes)


def write_spline(f, spline) -> None:
    """Serialise a built ``ChebyshevSpline`` with flat ``n_nodes`` to the open stream ``f``."""
    from .spline import _is_nested

    if any(p is None or p.tensor_values is None for p in spline._pieces):
        raise RuntimeError("Cannot save an unbuilt ChebyshevSpline")
    if getattr(spline, "additional_data", None) is not None:
        raise NotImplementedError("binary format cannot store additional_data; "
                                  "pass format='pickle' or set additional_data=None before saving")
    if _is_nested(spline.n_nodes):
        raise NotImplementedError("binary format requires flat n_nodes (shared across pieces); "
                                  "use format='pickle' for nested-n_nodes splines")
    d = int(spline.num_dimensions)
    f.write(MAGIC + struct.pack("<BBH", MAJOR, MINOR, CLASS_TAG_SPLINE) + b"\x00" * 4)
    f.write(struct.pack("<I", d))
    f.write(np.array([spline.domain[k][0] for k in range(d)], dtype="<f8").tobytes())
    f.write(np.array([spline.domain[k][1] for k in range(d)], dtype="<f8").tobytes())
    f.write(np.array(spline.n_nodes, dtype="<u4").tobytes())
    f.write(np.array([len(spline.knots[k]) for k in range(d)], dtype="<u4").tobytes())
    for k in range(d):
        if len(spline.knots[k]):
            f.write(np.asarray(spline.knots[k], dtype="<f8").tobytes())
    f.write(struct.pack("<I", len(spline._pieces)))
    for piece in spline._pieces:
        f.write(np.ascontiguousarray(piece.tensor_values, dtype="<f8").tobytes(order="C"))


def read_spline(f):
    """Parse a class-tag-2 ``.pcb`` stream into a ``ChebyshevSpline`` (via ``from_values``)."""
    from .spline import ChebyshevSpline

    tag = _read_header(f)
    if tag != CLASS_TAG_SPLINE:
        raise ValueError(f"file contains class_tag {tag}, expected {CLASS_TAG_SPLINE} (ChebyshevSpline)")
    d = struct.unpack("<I", _take(f, 4, "uint32"))[0]
    if d < 1:
        raise ValueError(f"num_dimensions must be >= 1, got {d}")
    lo = np.frombuffer(_take(f, 8 * d, "f64 array"), dtype="<f8")
    hi = np.frombuffer(_take(f, 8 * d, "f64 array"), dtype="<f8")
    domain = [[float(lo[k]), float(hi[k])] for k in range(d)]
    for k, (a, b) in enumerate(domain):
        if a >= b:
            raise ValueError(f"domain[{k}]: lo ({a}) must be < hi ({b})")
    n_nodes = [int(v) for v in np.frombuffer(_take(f, 4 * d, "uint32 array"), dtype="<u4")]
    for k, n in enumerate(n_nodes):
        if n < 1:
            raise ValueError(f"n_nodes[{k}] must be >= 1, got {n}")
    counts = [int(v) for v in np.frombuffer(_take(f, 4 * d, "uint32 array"), dtype="<u4")]
    flat = np.frombuffer(_take(f, 8 * sum(counts), "f64 array"), dtype="<f8") if sum(counts) else np.zeros(0)
    knots, at = [], 0
    for k in range(d):
        row = [float(x) for x in flat[at: at + counts[k]]]
        at += counts[k]
        if any(row[j] >= row[j + 1] for j in range(len(row) - 1)):
            raise ValueError(f"knots in dim {k} not strictly ascending")
        knots.append(row)
    num_pieces = struct.unpack("<I", _take(f, 4, "uint32"))[0]
    expected = int(np.prod([c + 1 for c in counts]))
    if num_pieces != expected:
        raise ValueError(f"num_pieces={num_pieces} does not match prod(num_knots+1)={expected}")
    per_piece = int(np.prod(n_nodes))
    values = [np.frombuffer(_take(f, 8 * per_piece, "f64 array"), dtype="<f8").astype(np.float64)
              .reshape(tuple(n_nodes), order="C") for _ in range(num_pieces)]
    return ChebyshevSpline.from_values(values, d, domain, n_nodes, knots)

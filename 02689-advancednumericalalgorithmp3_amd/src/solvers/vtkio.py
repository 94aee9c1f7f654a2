"""Minimal VTK XML StructuredGrid (.vts) reader / writer using only the standard library.

The reference reads FV reference solutions with pyvista (src/solvers/base.py:1008-1015) and
exports its own with ``pv.StructuredGrid.save`` (base.py:464-522, main.py:114-117).  pyvista
is not a dependency here; the files it writes are plain VTK XML: inline ``binary`` DataArrays,
base64, optional ``vtkZLibDataCompressor``, ``UInt32``/``UInt64`` block headers.
"""
from __future__ import annotations

import base64
import struct
import xml.etree.ElementTree as ET
import zlib
from pathlib import Path

import numpy as np

_NP = {"Float64": "<f8", "Float32": "<f4", "Int64": "<i8", "Int32": "<i4", "UInt8": "u1",
       "UInt32": "<u4", "UInt64": "<u8", "Int8": "i1"}


def _decode_binary(text: str, dtype: str, header: str, compressed: bool) -> np.ndarray:
    raw = "".join(text.split())
    hsz = 8 if header == "UInt64" else 4
    hfmt = "<Q" if hsz == 8 else "<I"
    if not compressed:
        buf = base64.b64decode(raw)
        (nbytes,) = struct.unpack_from(hfmt, buf, 0)
        return np.frombuffer(buf, dtype=_NP[dtype], count=nbytes // np.dtype(_NP[dtype]).itemsize, offset=hsz).copy()
    # compressed: [nblocks, blocksize, lastsize, csize_0..] is base64-encoded on its own
    first = base64.b64decode(raw[: 4 * ((hsz + 2) // 3)])
    (nblocks,) = struct.unpack_from(hfmt, first, 0)
    hbytes = (3 + nblocks) * hsz
    hchars = 4 * ((hbytes + 2) // 3)
    head = base64.b64decode(raw[:hchars])
    sizes = struct.unpack_from("<" + ("Q" if hsz == 8 else "I") * nblocks, head, 3 * hsz)
    body = base64.b64decode(raw[hchars:])
    out, off = [], 0
    for cs in sizes:
        out.append(zlib.decompress(body[off: off + cs]))
        off += cs
    return np.frombuffer(b"".join(out), dtype=_NP[dtype]).copy()


def read_vts(path) -> dict:
    """Return {'extent', 'points' (n,3), 'point_data' {name: array}, 'field_data' {...}}."""
    root = ET.parse(str(path)).getroot()
    if root.get("type") != "StructuredGrid":
        raise ValueError(f"{path}: not a StructuredGrid file")
    header = root.get("header_type", "UInt32")
    compressed = root.get("compressor") == "vtkZLibDataCompressor"

    def arr(el):
        fmt = el.get("format", "ascii")
        typ = el.get("type")
        if fmt == "ascii":
            a = np.array(el.text.split(), dtype=_NP[typ])
        elif fmt == "binary":
            a = _decode_binary(el.text, typ, header, compressed)
        else:
            raise ValueError(f"unsupported DataArray format {fmt}")
        nc = int(el.get("NumberOfComponents", "1"))
        return a.reshape(-1, nc) if nc > 1 else a

    grid = root.find("StructuredGrid")
    piece = grid.find("Piece")
    out = {"extent": tuple(int(t) for t in piece.get("Extent").split()), "point_data": {}, "field_data": {}}
    out["points"] = arr(piece.find("Points").find("DataArray"))
    pd = piece.find("PointData")
    if pd is not None:
        for el in pd.findall("DataArray"):
            out["point_data"][el.get("Name")] = arr(el)
    fd = grid.find("FieldData")
    if fd is not None:
        for el in fd.findall("DataArray"):
            out["field_data"][el.get("Name")] = arr(el)
    return out


class StructuredGridFile:
    """What ``solver.to_vtk()`` returns: point arrays on an nx x ny x 1 grid plus ``save(path)``."""

    def __init__(self, x: np.ndarray, y: np.ndarray):
        self.x, self.y = np.asarray(x, float), np.asarray(y, float)
        self.point_data: dict = {}
        self.field_data: dict = {}

    def __setitem__(self, name, values):
        self.point_data[name] = np.asarray(values)

    def __getitem__(self, name):
        return self.point_data[name]

    @property
    def points(self) -> np.ndarray:
        X, Y = np.meshgrid(self.x, self.y)                 # VTK order: x fastest
        return np.column_stack([X.ravel(), Y.ravel(), np.zeros(X.size)])

    def save(self, path):
        nx, ny = self.x.size, self.y.size
        ext = f"0 {nx - 1} 0 {ny - 1} 0 0"

        def enc(a, typ="Float64"):
            raw = np.ascontiguousarray(a, dtype=_NP[typ]).tobytes()
            comp = zlib.compress(raw, 0)      # (a valid zlib stream, stored: fp64 fields barely compress, and deflating the nine arrays
                                              #  of an N=128 trial was 62 ms of its 115-ms record even at level 1)
            head = struct.pack("<IIII", 1, len(raw), len(raw), len(comp))
            return (base64.b64encode(head) + base64.b64encode(comp)).decode()

        lines = ['<?xml version="1.0"?>',
                 '<VTKFile type="StructuredGrid" version="0.1" byte_order="LittleEndian" '
                 'header_type="UInt32" compressor="vtkZLibDataCompressor">',
                 f'  <StructuredGrid WholeExtent="{ext}">', "    <FieldData>"]
        for name, val in self.field_data.items():
            val = np.atleast_1d(val)
            if val.dtype.kind in "US":
                continue                                   # strings are metadata only
            typ = "Int64" if val.dtype.kind in "iu" else "Float64"
            lines += [f'      <DataArray type="{typ}" Name="{name}" NumberOfTuples="{val.size}" format="binary">',
                      "        " + enc(val, typ), "      </DataArray>"]
        lines += ["    </FieldData>", f'  <Piece Extent="{ext}">', '    <PointData Scalars="u">']
        for name, val in self.point_data.items():
            val = np.asarray(val, dtype=float)
            nc = f' NumberOfComponents="{val.shape[1]}"' if val.ndim == 2 else ""
            lines += [f'      <DataArray type="Float64" Name="{name}"{nc} format="binary">',
                      "        " + enc(val), "      </DataArray>"]
        lines += ["    </PointData>", "    <Points>",
                  '      <DataArray type="Float64" Name="Points" NumberOfComponents="3" format="binary">',
                  "        " + enc(self.points), "      </DataArray>", "    </Points>", "  </Piece>",
                  "  </StructuredGrid>", "</VTKFile>"]
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        Path(path).write_text("\n".join(lines) + "\n")

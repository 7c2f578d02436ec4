"""Reader/writer for IT++ ``it_file`` (version 3) containers.

The reference stores capture buffers and golden vectors in this format
(reference: src/capbuf.cpp:104-114 reads ``capbuf``/``fc``; src/capbuf.cpp:191-196
writes them; test/*.it are produced by Matlab/test_*.m via ``itsave``).  IT++ is
not available here, so the format was derived by inspection:

    file   := "IT++" u8(version=3) block*
    block  := u64 hdr_bytes, u64 data_bytes, u64 block_bytes,
              cstr name, cstr type, cstr desc, data[data_bytes]
    dvec   := u64 n, n*f64          ivec := u64 n, n*i32      bvec := u64 n, n*u8
    dcvec  := u64 n, n*(f64,f64)
    dmat/imat/dcmat := u64 rows, u64 cols, column-major data

All little-endian.  Matrices are returned as numpy arrays of shape (rows, cols).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

_MAGIC = b"IT++\x03"

_VEC = {"dvec": "<f8", "ivec": "<i4", "bvec": "u1", "dcvec": "<c16", "svec": "<i2",
        "fvec": "<f4", "fcvec": "<c8"}
_MAT = {"dmat": "<f8", "imat": "<i4", "bmat": "u1", "dcmat": "<c16", "smat": "<i2",
        "fmat": "<f4", "fcmat": "<c8"}
_SCALAR = {"float64": "<f8", "int32": "<i4", "bin": "u1", "cfloat64": "<c16",
           "float32": "<f4", "int16": "<i2", "uint64": "<u8", "int8": "i1"}


def _cstr(buf: bytes, pos: int):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin1"), end + 1


def read_it(path: str) -> "OrderedDict[str, np.ndarray]":
    """Return every variable of an ``.it`` file as ``name -> ndarray``."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:5] != _MAGIC:
        raise ValueError(f"{path}: not an IT++ v3 file")
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    pos = 5
    while pos + 24 <= len(buf):
        hdr_bytes, data_bytes, block_bytes = struct.unpack_from("<QQQ", buf, pos)
        p = pos + 24
        name, p = _cstr(buf, p)
        typ, p = _cstr(buf, p)
        _desc, p = _cstr(buf, p)
        d0 = pos + hdr_bytes
        data = buf[d0:d0 + data_bytes]
        if typ in _VEC:
            (n,) = struct.unpack_from("<Q", data, 0)
            arr = np.frombuffer(data, dtype=_VEC[typ], count=n, offset=8).copy()
        elif typ in _MAT:
            r, c = struct.unpack_from("<QQ", data, 0)
            arr = np.frombuffer(data, dtype=_MAT[typ], count=r * c, offset=16)
            arr = arr.reshape((c, r)).T.copy()          # column-major on disk
        elif typ in _SCALAR:
            arr = np.frombuffer(data, dtype=_SCALAR[typ], count=1).copy()[0]
        else:
            raise ValueError(f"{path}: unsupported IT++ type {typ!r} for {name!r}")
        out[name] = arr
        if block_bytes == 0:
            break
        pos += block_bytes
    return out


def _block(name: str, typ: str, payload: bytes) -> bytes:
    hdr = name.encode() + b"\0" + typ.encode() + b"\0" + b"\0"
    hdr_bytes = 24 + len(hdr)
    return struct.pack("<QQQ", hdr_bytes, len(payload), hdr_bytes + len(payload)) + hdr + payload


def write_it(path: str, variables: "dict[str, np.ndarray]") -> None:
    """Write 1-D / 2-D float64, int32, uint8 or complex128 arrays as an ``.it`` file."""
    chunks = [_MAGIC]
    for name, arr in variables.items():
        a = np.asarray(arr)
        if a.dtype.kind == "c":
            base, a = "dc", a.astype("<c16")
        elif a.dtype.kind == "f":
            base, a = "d", a.astype("<f8")
        elif a.dtype.kind in "iu" and a.dtype.itemsize == 1:
            base, a = "b", a.astype("u1")
        elif a.dtype.kind in "iu":
            base, a = "i", a.astype("<i4")
        else:
            raise ValueError(f"unsupported dtype {a.dtype} for {name}")
        if a.ndim <= 1:
            a = a.reshape(-1)
            payload = struct.pack("<Q", a.size) + a.tobytes()
            typ = base + "vec"
        elif a.ndim == 2:
            payload = struct.pack("<QQ", *a.shape) + np.asfortranarray(a).tobytes(order="F")
            typ = base + "mat"
        else:
            raise ValueError("only 1-D and 2-D arrays are supported")
        chunks.append(_block(name, typ, payload))
    with open(path, "wb") as f:
        f.write(b"".join(chunks))

"""Minimal GGUF v3 reader/writer (numpy only) used by the tests, the bench and the fixture scripts.

The *product's* GGUF reader is the C++ one in csrc/gguf_reader.cpp; this module exists so that the
test-suite can write synthetic checkpoints with the exact tensor names / shapes / dtypes of the
reference's weight contract (SURVEY.md Appx A) and hand tensors to the CPU oracle.

File format restated from the reference's reader, ggml/src/ggml.c:6455-6476 (header/KV/tensor-info
structs) and :6620-6909 (gguf_init_from_file): magic "GGUF", u32 version, u64 n_tensors, u64 n_kv,
KV pairs, tensor infos, padding to `general.alignment` (default 32), data blob with every tensor
padded to the alignment.  numpy shape (a, b, c) <-> ggml ne = [c, b, a].
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
GGUF_DEFAULT_ALIGNMENT = 32

# ggml type codes (ggml/include/ggml.h:358-384)
GGML_TYPE_F32 = 0
GGML_TYPE_F16 = 1
GGML_TYPE_I32 = 26

# gguf KV type codes (ggml/include/ggml.h gguf_type)
GGUF_TYPE_UINT32 = 4
GGUF_TYPE_STRING = 8

_NP2GGML = {np.dtype(np.float32): GGML_TYPE_F32, np.dtype(np.float16): GGML_TYPE_F16,
            np.dtype(np.int32): GGML_TYPE_I32}
_GGML2NP = {v: k for k, v in _NP2GGML.items()}


def _pad(n: int, a: int) -> int:
    return (n + a - 1) // a * a


def _wstr(s: str) -> bytes:
    b = s.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def write_gguf(path: str, kv_u32: Dict[str, int], tensors: List[Tuple[str, np.ndarray]],
               arch: str = "zerovox-resnet-fs2-styletts", trim_dims: bool = False) -> None:
    """Write a GGUF v3 file.  `trim_dims=True` mimics ggml's C writer, which stores ggml_n_dims()
    (trailing size-1 dims dropped); the default keeps numpy's rank like the Python gguf package
    that utils/zv2gguf.py uses.  Readers must accept both (SURVEY.md Appx A)."""
    align = GGUF_DEFAULT_ALIGNMENT
    kvs = [("general.architecture", GGUF_TYPE_STRING, arch)]
    kvs += [(k, GGUF_TYPE_UINT32, int(v)) for k, v in kv_u32.items()]

    head = bytearray()
    head += GGUF_MAGIC + struct.pack("<IQQ", GGUF_VERSION, len(tensors), len(kvs))
    for k, t, v in kvs:
        head += _wstr(k) + struct.pack("<I", t)
        head += _wstr(v) if t == GGUF_TYPE_STRING else struct.pack("<I", v)

    offset = 0
    infos = bytearray()
    for name, arr in tensors:
        assert len(name) < 64, name                      # GGML_MAX_NAME
        assert arr.dtype in _NP2GGML, (name, arr.dtype)
        ne = list(arr.shape[::-1])
        if trim_dims:
            while len(ne) > 1 and ne[-1] == 1:
                ne.pop()
        infos += _wstr(name) + struct.pack("<I", len(ne))
        infos += b"".join(struct.pack("<Q", d) for d in ne)
        infos += struct.pack("<IQ", _NP2GGML[arr.dtype], offset)
        offset += _pad(arr.nbytes, align)

    with open(path, "wb") as f:
        f.write(head)
        f.write(infos)
        pos = len(head) + len(infos)
        f.write(b"\0" * (_pad(pos, align) - pos))
        for _, arr in tensors:
            b = np.ascontiguousarray(arr).tobytes()
            f.write(b)
            f.write(b"\0" * (_pad(len(b), align) - len(b)))


def read_gguf(path: str):
    """Return (kv dict, {name: ndarray}) — arrays are memory-mapped views in numpy (C-order) shape."""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    buf = memoryview(mm)
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, buf, pos)
        pos += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def rstr():
        nonlocal pos
        n = rd("Q")
        s = bytes(buf[pos:pos + n]).decode("utf-8")
        pos += n
        return s

    if bytes(buf[0:4]) != GGUF_MAGIC:
        raise ValueError("not a GGUF file")
    pos = 4
    version = rd("I")
    if version != GGUF_VERSION:
        raise ValueError(f"unsupported GGUF version {version}")
    n_tensors, n_kv = rd("Q"), rd("Q")
    scalar = {0: "B", 1: "b", 2: "H", 3: "h", 4: "I", 5: "i", 6: "f", 7: "?", 10: "Q", 11: "q", 12: "d"}
    kv = {}
    for _ in range(n_kv):
        k = rstr()
        t = rd("I")
        if t == GGUF_TYPE_STRING:
            kv[k] = rstr()
        elif t in scalar:
            kv[k] = rd(scalar[t])
        elif t == 9:                                     # array
            et, n = rd("I"), rd("Q")
            kv[k] = [rstr() if et == GGUF_TYPE_STRING else rd(scalar[et]) for _ in range(n)]
        else:
            raise ValueError(f"bad KV type {t}")
    align = int(kv.get("general.alignment", GGUF_DEFAULT_ALIGNMENT))
    infos = []
    for _ in range(n_tensors):
        name = rstr()
        nd = rd("I")
        ne = [rd("Q") for _ in range(nd)]
        typ, off = rd("I"), rd("Q")
        infos.append((name, ne, typ, off))
    data0 = _pad(pos, align)
    out = {}
    for name, ne, typ, off in infos:
        dt = _GGML2NP[typ]
        n = int(np.prod(ne))
        out[name] = np.frombuffer(mm, dtype=dt, count=n, offset=data0 + off).reshape(ne[::-1])
    return kv, out

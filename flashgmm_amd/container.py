"""Byte layout for what the GMM codecs return.

The reference has no container for its GMM models: ``net.compress()`` returns a nested Python structure —
``{"strings": [(bytes, abs_max, zero_bitmap), ..., [z_bytes]], "shape": ...}`` (latent_codecs/checkerboard.py:287-294,
channel_groups.py:123-129, hyperprior.py) — its evaluation script counts ``len(s[0])`` only, leaving ``abs_max`` and
``zero_bitmap`` out of the bit rate (eval_ckbd.py:137-140), and the stock container of ``examples/codec.py:181-198``
cannot hold the tuples.  This module gives that structure a real, self-describing byte layout, in the style of the
stock one (big-endian, length-prefixed), so that it can be written to disk, counted honestly, and shipped between
ranks (``flashgmm_amd.parallel.gather_containers``):

    "FGM1" | u32 n_entries | entry* | shape
    entry  = u8 kind
             kind 0 (plain strings, e.g. the hyper-latent's ``[z_bytes]``): u32 n | (u32 len | bytes)*
             kind 1 (GMM stream ``(bytes, abs_max, zero_bitmap)``): u32 abs_max | u16 M | ceil(M/8) bitmap bytes
                    (channel c = bit c%8 of byte c//8) | u32 len | bytes
             kind 2 (the same with the stream's out-of-band checkpoints, ``flashgmm_amd.CheckpointedBytes``): as kind 1, then
                    u32 stride | u32 n | (u64 state | u32 position)*  — ``unpack`` gives a ``CheckpointedBytes`` back, which
                    ``GaussianMixtureConditional.decompress`` decodes in segments (on the GPU, or on all host workers);
                    12 bytes per `stride` symbols
    shape  = a tagged tree: u8 tag; 0 int (i32) | 1 list (u16 n, items) | 2 tuple (u16 n, items) |
             3 dict (u16 n, (u16 klen | utf-8 key | value)*) | 4 None

Pure host-side Python (struct / numpy): nothing here is on the timed path.
"""
from __future__ import annotations

import struct
from typing import Any, List, Sequence, Tuple

import numpy as np

__all__ = ["pack", "unpack", "num_bytes", "side_info_bytes"]

MAGIC = b"FGM1"


def _bitmap_to_bytes(zb) -> Tuple[int, bytes]:
    a = np.asarray(zb.cpu() if hasattr(zb, "cpu") else zb).astype(np.int64).reshape(-1)
    if a.size > 0xFFFF:
        raise ValueError("zero_bitmap longer than 65535 channels")
    return a.size, np.packbits((a != 0).astype(np.uint8), bitorder="little").tobytes()


def _put_shape(out: List[bytes], v: Any) -> None:
    if v is None:
        out.append(b"\x04")
    elif isinstance(v, dict):
        out.append(struct.pack(">BH", 3, len(v)))
        for k, x in v.items():
            kb = str(k).encode()
            out.append(struct.pack(">H", len(kb)) + kb)
            _put_shape(out, x)
    elif isinstance(v, (list, tuple)) or (hasattr(v, "__len__") and not isinstance(v, (bytes, str))):  # torch.Size too
        out.append(struct.pack(">BH", 2 if isinstance(v, tuple) else 1, len(v)))
        for x in v:
            _put_shape(out, x)
    else:
        out.append(struct.pack(">Bi", 0, int(v)))


def pack(strings: Sequence[Any], shape: Any = None) -> bytes:
    """``strings``: a list whose elements are GMM streams ``(bytes, abs_max, zero_bitmap)`` or lists of plain byte
    strings; ``shape``: ints / lists / tuples / dicts of them (``torch.Size`` counts as a tuple)."""
    out = [MAGIC, struct.pack(">I", len(strings))]
    for s in strings:
        if isinstance(s, tuple) and len(s) == 3 and isinstance(s[0], (bytes, bytearray)):
            data, abs_max, zb = s
            m, bits = _bitmap_to_bytes(zb)
            ck = getattr(data, "ckpt", None)
            kind = 2 if ck is not None and len(ck) else 1
            out.append(struct.pack(">BIH", kind, int(abs_max), m) + bits + struct.pack(">I", len(data)) + bytes(data))
            if kind == 2:
                if len(ck) and int(ck["pos"].max()) >= 1 << 32:
                    raise ValueError("checkpoint position beyond 2^32 words")
                rec = np.zeros(len(ck), dtype=[("x", ">u8"), ("pos", ">u4")])
                rec["x"], rec["pos"] = ck["x"], ck["pos"]
                out.append(struct.pack(">II", int(data.ckpt_stride), len(ck)) + rec.tobytes())
        elif isinstance(s, (list, tuple)) and all(isinstance(b, (bytes, bytearray)) for b in s):
            out.append(struct.pack(">BI", 0, len(s)))
            for b in s:
                out.append(struct.pack(">I", len(b)) + bytes(b))
        else:
            raise TypeError(f"cannot pack an entry of type {type(s).__name__}")
    _put_shape(out, shape)
    return b"".join(out)


class _Reader:
    def __init__(self, buf: bytes):
        self.b, self.o = memoryview(buf), 0

    def take(self, n: int) -> bytes:
        if n < 0 or self.o + n > len(self.b):
            raise ValueError("truncated container")
        v = bytes(self.b[self.o:self.o + n])
        self.o += n
        return v

    def unpack(self, fmt: str):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))


def _get_shape(r: _Reader) -> Any:
    (tag,) = r.unpack(">B")
    if tag == 0:
        return r.unpack(">i")[0]
    if tag in (1, 2):
        (n,) = r.unpack(">H")
        items = [_get_shape(r) for _ in range(n)]
        return tuple(items) if tag == 2 else items
    if tag == 3:
        (n,) = r.unpack(">H")
        d = {}
        for _ in range(n):
            (kl,) = r.unpack(">H")
            k = r.take(kl).decode()
            d[k] = _get_shape(r)
        return d
    if tag == 4:
        return None
    raise ValueError(f"bad shape tag {tag}")


def unpack(buf: bytes, device=None):
    """-> (strings, shape); ``zero_bitmap`` comes back as an int64 torch tensor (on ``device`` if given), as
    ``GaussianMixtureConditional.decompress`` takes it."""
    import torch

    r = _Reader(buf)
    if r.take(4) != MAGIC:
        raise ValueError("not an FGM1 container")
    (n,) = r.unpack(">I")
    strings: List[Any] = []
    for _ in range(n):
        (kind,) = r.unpack(">B")
        if kind in (1, 2):
            abs_max, m = r.unpack(">IH")
            bits = np.frombuffer(r.take((m + 7) // 8), np.uint8)
            zb = torch.from_numpy(np.unpackbits(bits, bitorder="little")[:m].astype(np.int64))
            (ln,) = r.unpack(">I")
            data = r.take(ln)
            if kind == 2:
                from .entropy_models import CKPT_DTYPE, CheckpointedBytes

                stride, n_ck = r.unpack(">II")
                rec = np.frombuffer(r.take(12 * n_ck), dtype=[("x", ">u8"), ("pos", ">u4")])
                ck = np.zeros(n_ck, CKPT_DTYPE)
                ck["x"], ck["pos"] = rec["x"], rec["pos"]
                data = CheckpointedBytes(data, ck, stride)
            strings.append((data, int(abs_max), zb.to(device) if device is not None else zb))
        elif kind == 0:
            (k,) = r.unpack(">I")
            strings.append([r.take(r.unpack(">I")[0]) for _ in range(k)])
        else:
            raise ValueError(f"bad entry kind {kind}")
    shape = _get_shape(r)
    if r.o != len(buf):
        raise ValueError("trailing bytes after the container")
    return strings, shape


def num_bytes(strings: Sequence[Any], shape: Any = None) -> int:
    """size of the packed container: the honest bit-rate numerator (payloads + abs_max + zero_bitmap + framing)"""
    return len(pack(strings, shape))


def side_info_bytes(strings: Sequence[Any], shape: Any = None) -> int:
    """what the reference's evaluation leaves out (eval_ckbd.py:137-140): everything but the payload bytes"""
    payload = sum(len(s[0]) if isinstance(s, tuple) else sum(len(b) for b in s) for s in strings)
    return num_bytes(strings, shape) - payload

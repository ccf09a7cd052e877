"""``GaussianMixtureConditional`` — the Python boundary of the path, same API as the reference's class
(compressai/entropy_models/entropy_models.py:762-910), running on MI355X.

    gmc = GaussianMixtureConditional(K=4)
    (strings, abs_max, zero_bitmap), y_q = gmc.compress(y, scales, means, weights)
    y_hat = gmc.decompress(strings, abs_max, zero_bitmap, scales, means, weights)

``y`` is ``[1, M, h, w]``; ``scales / means / weights`` are ``[1, K*M, h, w]`` with channel ``k*M + c``
(latent_codecs/gaussian_mixture_conditional.py:193-202).  What the reference does on the host between these
tensors and its C++ coder — abs-max, round, zero-channel bitmap, channel gather, the (n,K) re-layout with the
sigma clamp, four ``.to("cpu")`` copies — is fused into the HIP kernels: parameters are read in place from the
planar layout and only the 4 B/symbol table crosses PCIe.  Returned values and bitstreams are identical to the
reference's for identical inputs.

Intended deviation: ``decompress`` returns ``y_hat`` on the parameters' device (the reference builds it on the
CPU, entropy_models.py:905-908, and lets the caller's assignment copy it back).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from . import _lib

__all__ = ["GaussianMixtureConditional", "EntropyBottleneckCoder", "CheckpointedBytes", "CompressedBatch", "ParameterHead"]

CKPT_DTYPE = np.dtype([("x", "<u8"), ("pos", "<u8")])  # fgmm_ckpt

class CheckpointedBytes(bytes):
    """A bitstream — the reference's, byte for byte: it IS a ``bytes`` object and compares equal to one — that also carries
    its encoder's out-of-band CHECKPOINTS (``include/flashgmm_amd.h``: ``fgmm_ckpt``): the coder state and the stream
    position before every ``stride``-th symbol.  ``GaussianMixtureConditional.decompress`` decodes the segments between them
    on all host workers instead of one (every segment is verified against the next checkpoint; wrong ones cost a
    sequential decode, never a wrong symbol).  Anything that only knows ``bytes`` — the reference's decoder, a file — sees
    the plain stream; ``flashgmm_amd.container`` stores the checkpoints next to it."""

    # _ck = (notes, first, count, address, stride): `notes` is an ndarray of CKPT_DTYPE (first = 0), or the ``bytes`` blob a compiled
    # compress call filled with the notes of ALL its bitstreams - this one's are records first .. first + count of it, the view is made
    # when someone asks for ``.ckpt``; address: of the first note (what a decode call hands the library)
    def __new__(cls, data: bytes, ckpt: np.ndarray, stride: int):
        self = super().__new__(cls, data)
        if not (isinstance(ckpt, np.ndarray) and ckpt.dtype == CKPT_DTYPE and ckpt.flags.c_contiguous):
            ckpt = np.ascontiguousarray(ckpt, dtype=CKPT_DTYPE)
        self._ck = (ckpt, 0, len(ckpt), ckpt.ctypes.data if len(ckpt) else 0, int(stride))  # (ndarray.ctypes is slow: taken once)
        return self

    @property
    def ckpt(self) -> np.ndarray:
        """[(x: u64, pos: u64)], entry k = before symbol (k + 1) * stride"""
        notes, first, count, addr, stride = self._ck
        if not isinstance(notes, np.ndarray):
            notes = np.frombuffer(notes, dtype=CKPT_DTYPE, count=count, offset=16 * first) if count else _NO_CKPT
            self._ck = (notes, 0, count, addr, stride)
        return notes

    @property
    def ckpt_stride(self) -> int:
        return self._ck[4]

    @property
    def _ckpt_addr(self) -> int:
        return self._ck[3]

    @classmethod
    def _adopt(cls, blank: "CheckpointedBytes", ckpt: np.ndarray, stride: int, addr: int = -1) -> "CheckpointedBytes":
        """attach the notes to an instance whose bytes are already in place (``_lib.take_bytes_many(..., cls=CheckpointedBytes)``:
        the library's workers copied the bitstream into it - no second copy through ``__new__``)"""
        blank._ck = (ckpt, 0, len(ckpt), addr if addr >= 0 else (ckpt.ctypes.data if len(ckpt) else 0), int(stride))
        return blank

    def __reduce__(self):
        return (CheckpointedBytes, (bytes(self), self.ckpt, self.ckpt_stride))


_NO_CKPT = np.zeros(0, CKPT_DTYPE)


class CompressedBatch(Sequence):
    """What ``compress_batch`` returns for STACKED inputs: the N results held as the call produced them - ``strings`` (list of
    N ``bytes``), ``abs_maxes`` (list of N ints), ``zero_bitmaps`` (one ``[N, M]`` int64 CPU tensor), ``y_q`` (one
    ``[N, 1, M, h, w]`` tensor).  It IS the sequence of ``((bytes, abs_max, zero_bitmap), y_q)`` the list form returns -
    ``len``, indexing, slicing and iteration give the per-item tuples (views, made when asked for) - and its fields go
    straight back into ``decompress_batch`` (``strings[s::2]``, ``zero_bitmaps[s::2]`` ...) without 2 N tensor views
    being built for a caller that never looks at them."""

    __slots__ = ("strings", "abs_maxes", "zero_bitmaps", "y_q")

    def __init__(self, strings: List[bytes], abs_maxes: List[int], zero_bitmaps: Tensor, y_q: Tensor):
        self.strings, self.abs_maxes, self.zero_bitmaps, self.y_q = strings, abs_maxes, zero_bitmaps, y_q

    def __len__(self) -> int:
        return len(self.strings)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self.strings)))]
        return (self.strings[i], self.abs_maxes[i], self.zero_bitmaps[i]), self.y_q[i]

    # list semantics where they are cheap (callers written against the list a sequence-input call returns): concatenation
    # gives a plain list of the per-item tuples
    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)


def _take_ckpts_many(device: int, ptrs, counts):
    """the library-allocated fgmm_ckpt arrays of a batch -> [(ndarray view, its address)], the arrays released: ONE array for the
    batch and one native call instead of a copy and a free per bitstream"""
    total = int(sum(counts))
    if total == 0:
        return [(_NO_CKPT, 0)] * len(ptrs)
    pool = np.empty(total, CKPT_DTYPE)
    base = pool.ctypes.data
    out, dst, src, lens, at = [], [], [], [], 0
    for p, n in zip(ptrs, counts):
        if p and n > 0:
            out.append((pool[at:at + n], base + 16 * at))
            dst.append(base + 16 * at); src.append(p); lens.append(16 * n)
            at += n
        else:
            out.append((_NO_CKPT, 0))
    _lib.take_buffers_into(device, dst, src, lens)
    return out


def _plane_view(t: Tensor, K: int) -> Tuple[Tensor, int, int]:
    """-> (tensor kept alive, stride_k, stride_c) for a [1, K*M, h, w] parameter tensor whose (h, w) planes are
    dense; anything else is made contiguous first.  chunk(3, 1) views of the parameter head qualify as they are."""
    shape = t.shape
    if len(shape) != 4 or shape[0] != 1:
        raise RuntimeError("entropy parameters must be [1, K*M, h, w] (the reference squeezes batch 1 too, entropy_models.py:841)")
    if t.dtype not in _PARAM_DTYPES:
        raise RuntimeError(f"entropy parameters must be float32 or float16, got {t.dtype}")
    hw = shape[2] * shape[3]
    st = t.stride()
    if hw > 1 and not (st[3] == 1 and st[2] == shape[3]):
        t = t.contiguous()
        st = t.stride()
    sc = st[1] if shape[1] > 1 else hw
    return t, (shape[1] // K) * sc, sc


_PARAM_DTYPES = (torch.float32, torch.float16)


class ParameterHead:
    """The last layer of the reference's ``entropy_parameters`` - ``nn.Conv2d(c_in, 3*K*M, 1)`` (compressai/models/ckbd_gmm.py:115-121) -
    with the ``chunk(3, 1)`` / softmax-over-K that follows it (latent_codecs/gaussian_mixture_conditional.py:183-202), on the matrix
    cores in ONE fixed summation order (include/flashgmm_amd.h section 2b: bit for bit an ``fmaf`` chain over the input channels):

        head = ParameterHead(entropy_parameters[-1])                       # any nn.Conv2d(c_in, 3*K*M, 1) on the GPU
        scales, means, logits = head.params(x)                             # x [N, c_in, h, w] -> three [N, K*M, h, w] views
        res = gmc.compress_head_batch(y, x, head)                          # the same parameters, never written to HBM
        y_hat = gmc.decompress_batch(res.strings, res.abs_maxes, res.zero_bitmaps, scales, means, logits, weights_are_logits=True)

    Encoder and decoder then derive identical parameters from identical weights whatever BLAS / MIOpen version either side
    runs - the reference relies on that silently.  The weights are packed once, here."""

    def __init__(self, conv, K: int = 4, arithmetic: str = "f32"):
        # arithmetic: "f32" (the default: binary32 products and sums on v_mfma_f32_32x32x2_f32, bit for bit an fmaf chain) or "bf16x6"
        # (three bfloat16 parts per operand, six part products on the BF16 matrix cores, binary32 accuracy at a third of the cycles;
        # deterministic on MI355X, not restatable on a CPU: include/flashgmm_amd.h FGMM_HEAD_BF16X6)
        if arithmetic not in ("f32", "bf16x6"):
            raise ValueError(f"arithmetic {arithmetic!r}")
        self.arithmetic = arithmetic
        w = conv.weight.detach()
        if w.dim() != 4 or w.shape[2:] != (1, 1) or w.shape[0] % (3 * K) or not w.is_cuda:
            raise RuntimeError("expected the weights of a 1x1 nn.Conv2d(c_in, 3*K*M) on a HIP device")
        self.K, self.M, self.c_in = int(K), int(w.shape[0] // (3 * K)), int(w.shape[1])
        self.device = w.device
        self._dev = w.device.index if w.device.index is not None else -1
        w2 = w.reshape(w.shape[0], w.shape[1]).to(torch.float32).contiguous()
        b = None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous()
        out = C.c_void_p()
        stream = torch.cuda.current_stream(w.device).cuda_stream
        _lib.check(_lib.lib().fgmm_head_create_ex(_lib.ctx(self._dev), stream, w2.data_ptr(), b.data_ptr() if b is not None else None, self.M, self.K,
                                                  self.c_in, _lib.FGMM_HEAD_BF16X6 if arithmetic == "bf16x6" else 0, C.byref(out)), "ParameterHead")
        self._h = out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().fgmm_head_destroy(h)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass

    def _features(self, x: Tensor) -> Tensor:
        if x.dim() != 4 or x.shape[1] != self.c_in or x.device != self.device:
            raise RuntimeError(f"features must be [N, {self.c_in}, h, w] on {self.device}; got {tuple(x.shape)} on {x.device}")
        return x.to(torch.float32).contiguous()

    def params(self, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """-> (scales, means, logits): views ``[N, K*M, h, w]`` of one ``[N, 3*K*M, h, w]`` tensor - what ``entropy_parameters[-1](x)
        .chunk(3, 1)`` gives, in the library's summation order.  ``logits``: pass ``weights_are_logits=True`` to the coder."""
        x = self._features(x)
        N, _, h, w = x.shape
        out = torch.empty((N, 3 * self.K * self.M, h, w), dtype=torch.float32, device=x.device)
        if N and h * w:
            per_x, per_o = self.c_in * h * w * 4, 3 * self.K * self.M * h * w * 4
            xs = (C.c_void_p * N)(*[x.data_ptr() + i * per_x for i in range(N)])
            os_ = (C.c_void_p * N)(*[out.data_ptr() + i * per_o for i in range(N)])
            hws = (C.c_int64 * N)(*([h * w] * N))
            _lib.check(_lib.lib().fgmm_head_params_batch(_lib.ctx(self._dev), torch.cuda.current_stream(x.device).cuda_stream, self._h, xs, os_, hws, N),
                       "ParameterHead.params")
        return tuple(out.chunk(3, 1))


class GaussianMixtureConditional(nn.Module):
    """Entropy model of a K-component Gaussian mixture conditional; ``compress`` / ``decompress`` only need K = 4
    (the reference's coder is bound for K = 4 only, rans_interface.cpp:60,982)."""

    def __init__(self, K: int = 4, mode=None, clamp_scales: bool = True, checkpoint_stride: int = 0):
        super().__init__()
        self.K = int(K)
        self.mode = mode  # None -> APPROX_MODE env var at call time
        self.clamp_scales = bool(clamp_scales)  # entropy_models.py:817 clamp(0.11, 256)
        # > 0: compress() returns CheckpointedBytes (the same bytes + out-of-band checkpoints every that many symbols, a power
        # of two >= 256), which decompress() decodes on all host workers; 0 (the default): plain bytes, as the reference's
        self.checkpoint_stride = int(checkpoint_stride)
        if self.checkpoint_stride and (self.checkpoint_stride < 256 or self.checkpoint_stride & (self.checkpoint_stride - 1)):
            raise ValueError("checkpoint_stride must be 0 or a power of two >= 256")

    # ------------------------------------------------------------------------------------------------
    def _mode(self) -> int:
        return _lib.default_mode() if self.mode is None else _lib.mode_id(self.mode)

    def reshape_entropy_parameters(self, scales, means, weights, nonzero):
        """Same result as the reference (entropy_models.py:810-828): three (n, K) views with strides (1, n), sigma
        clamped.  The HIP path never materialises these; kept for callers and tests that want the coder inputs."""
        reshape_size = (scales.size(0), self.K, scales.size(1) // self.K, -1)

        def rs(t):
            return t.reshape(*reshape_size)[:, :, nonzero].permute(1, 0, 2, 3).reshape(self.K, -1).permute(1, 0)

        return rs(scales).clamp(0.11, 256), rs(means), rs(weights)

    # ------------------------------------------------------------------------------------------------
    def _item(self, y: Optional[Tensor], scales: Tensor, means: Tensor, weights: Tensor, keep: list, flags: int = 0):
        """one item of the sequence form as an ``fgmm_item`` (the ctypes binding) -> (item, M, hw, device)"""
        yp, sp, mp, wp, M, hw, sk, sc, dev, dt = self._item_ints(y, scales, means, weights, keep)
        it = _lib.fgmm_item()
        it.params = _lib.fgmm_params(sp, mp, wp, sk, sc, _lib.FGMM_F16 if dt == torch.float16 else _lib.FGMM_F32, flags)
        it.M, it.K, it.hw = M, self.K, hw
        if y is not None:
            it.y = yp
        return it, M, hw, dev

    def _item_ints(self, y: Optional[Tensor], scales: Tensor, means: Tensor, weights: Tensor, keep: list):
        """one item of the sequence form for the compiled boundary -> (pointers and strides as integers, M, hw, device, dtype); the
        tensors whose storage the pointers name are appended to ``keep``"""
        if not scales.is_cuda:
            raise RuntimeError(
                "flashgmm_amd runs the GMM entropy-coding path on the GPU only: tensors must be on a HIP device "
                "(there is deliberately no CPU fallback)")
        s, sk, sc = _plane_view(scales, self.K)
        m, mk, mc = _plane_view(means, self.K)
        w, wk, wc = _plane_view(weights, self.K)
        if (mk, mc) != (sk, sc) or (wk, wc) != (sk, sc) or m.shape != s.shape or w.shape != s.shape:
            if m.shape != s.shape or w.shape != s.shape:
                raise RuntimeError("scales, means and weights must have one shape")
            s, m, w = s.contiguous(), m.contiguous(), w.contiguous()
            hw_ = s.size(2) * s.size(3)
            sc, sk = hw_, (s.size(1) // self.K) * hw_
        if not (s.dtype == m.dtype == w.dtype):
            raise RuntimeError("scales, means and weights must share one dtype")
        shp = s.shape
        M = shp[1] // self.K
        hw = shp[2] * shp[3]
        keep += [s, m, w]
        yp = 0
        if y is not None:
            ys_ = y.shape
            if len(ys_) != 4 or ys_[0] != 1 or ys_[1] != M or ys_[2] * ys_[3] != hw:
                raise RuntimeError(f"y must be [1, {M}, h, w] matching the parameters; got {tuple(ys_)}")
            if y.dtype != torch.float32 or y.device != s.device:  # latents stay float32 (32 B/symbol with fp16 planes)
                raise RuntimeError("y must be float32 on the parameters' device")
            yc = y.contiguous()
            yp = yc.data_ptr()
            keep.append(yc)
        return yp, s.data_ptr(), m.data_ptr(), w.data_ptr(), M, hw, sk, sc, s.device, s.dtype

    def _stacked_items(self, y: Optional[Tensor], scales: Tensor, means: Tensor, weights: Tensor, flags: int = 0):
        """``fgmm_item[N]`` (as a numpy record array) for N items given as ONE tensor each: y ``[N, M, h, w]``,
        parameters ``[N, K*M, h, w]``.  One validation and one set of strides for the whole batch; the item
        pointers are the batch-dimension offsets."""
        if not scales.is_cuda:
            raise RuntimeError(
                "flashgmm_amd runs the GMM entropy-coding path on the GPU only: tensors must be on a HIP device "
                "(there is deliberately no CPU fallback)")
        if scales.dim() != 4 or means.shape != scales.shape or weights.shape != scales.shape:
            raise RuntimeError("stacked entropy parameters must be three [N, K*M, h, w] tensors of one shape")
        if not (scales.dtype == means.dtype == weights.dtype) or scales.dtype not in _PARAM_DTYPES:
            raise RuntimeError("scales, means and weights must share one dtype, float32 or float16")
        N, KM, h, w = scales.shape
        hw = h * w
        st = scales.stride()
        if (hw > 1 and not (st[3] == 1 and st[2] == w)) or means.stride() != st or weights.stride() != st:
            scales, means, weights = scales.contiguous(), means.contiguous(), weights.contiguous()
            st = scales.stride()
        M = KM // self.K
        sc = st[1] if KM > 1 else hw
        esz = scales.element_size()
        items = np.zeros(N, _lib.ITEM_DTYPE)
        step = np.arange(N, dtype=np.uint64) * np.uint64(st[0] * esz)
        items["scales"] = np.uint64(scales.data_ptr()) + step
        items["means"] = np.uint64(means.data_ptr()) + step
        items["weights"] = np.uint64(weights.data_ptr()) + step
        items["stride_k"], items["stride_c"] = M * sc, sc
        items["dtype"] = _lib.FGMM_F16 if scales.dtype == torch.float16 else _lib.FGMM_F32
        items["flags"] = flags
        items["M"], items["K"], items["hw"] = M, self.K, hw
        keep = [scales, means, weights]
        if y is not None:
            if y.dim() != 4 or tuple(y.shape) != (N, M, h, w):
                raise RuntimeError(f"y must be [{N}, {M}, {h}, {w}] matching the parameters; got {tuple(y.shape)}")
            if y.dtype != torch.float32 or y.device != scales.device:
                raise RuntimeError("y must be float32 on the parameters' device")
            y = y.contiguous()
            items["y"] = np.uint64(y.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * hw * 4)
            keep.append(y)
        return items, keep, N, M, h, w, scales.device

    def _stacked_view(self, y: Optional[Tensor], scales: Tensor, means: Tensor, weights: Tensor):
        """validation of stacked inputs for the compiled boundary -> (y, scales, means, weights, N, M, h, w, item stride, stride_c)"""
        if not scales.is_cuda:
            raise RuntimeError(
                "flashgmm_amd runs the GMM entropy-coding path on the GPU only: tensors must be on a HIP device "
                "(there is deliberately no CPU fallback)")
        if scales.dim() != 4 or means.shape != scales.shape or weights.shape != scales.shape:
            raise RuntimeError("stacked entropy parameters must be three [N, K*M, h, w] tensors of one shape")
        if not (scales.dtype == means.dtype == weights.dtype) or scales.dtype not in _PARAM_DTYPES:
            raise RuntimeError("scales, means and weights must share one dtype, float32 or float16")
        N, KM, h, w = scales.shape
        st = scales.stride()
        if (h * w > 1 and not (st[3] == 1 and st[2] == w)) or means.stride() != st or weights.stride() != st:
            scales, means, weights = scales.contiguous(), means.contiguous(), weights.contiguous()
            st = scales.stride()
        M = KM // self.K
        if y is not None:
            if y.dim() != 4 or tuple(y.shape) != (N, M, h, w):
                raise RuntimeError(f"y must be [{N}, {M}, {h}, {w}] matching the parameters; got {tuple(y.shape)}")
            if y.dtype != torch.float32 or y.device != scales.device:
                raise RuntimeError("y must be float32 on the parameters' device")
            y = y.contiguous()
        return y, scales, means, weights, N, M, h, w, st[0], (st[1] if KM > 1 else h * w)

    def _compress_stacked(self, y: Tensor, scales: Tensor, means: Tensor, weights: Tensor, flags: int = 0):
        nat = _lib.native()
        if nat is not None and scales.dim() == 4 and scales.shape[0] > 0:
            y, scales, means, weights, N, M, h, w, s_item, sc = self._stacked_view(y, scales, means, weights)
            dev = scales.device
            di = dev.index if dev.index is not None else -1
            yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
            zb = torch.empty((N, M), dtype=torch.int64)
            strings, amax = nat.compress_stacked(
                _lib.ctx_addr(di), torch.cuda.current_stream(dev).cuda_stream, y.data_ptr(), scales.data_ptr(), means.data_ptr(), weights.data_ptr(),
                N, M, h * w, s_item, M * sc, sc, _lib.FGMM_F16 if scales.dtype == torch.float16 else _lib.FGMM_F32, flags, self._mode(),
                int(self.clamp_scales), self.checkpoint_stride, yq.data_ptr(), zb.data_ptr(), CheckpointedBytes if self.checkpoint_stride else None)
            return CompressedBatch(strings, amax, zb, yq)
        items, keep, N, M, h, w, dev = self._stacked_items(y, scales, means, weights, flags)
        if N == 0:
            return CompressedBatch([], [], torch.empty((0, M), dtype=torch.int64), torch.empty((0, 1, M, h, w), dtype=torch.float32, device=dev))
        yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
        zb = torch.empty((N, M), dtype=torch.int64)
        items["yq_out"] = np.uint64(yq.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
        items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * 8)
        items["ckpt_stride"] = self.checkpoint_stride
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.lib().fgmm_gmc_compress_batch(_lib.ctx(dev.index if dev.index is not None else -1), stream,
                                                C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(),
            int(self.clamp_scales))
        _lib.check(rc, "GaussianMixtureConditional.compress")
        ptrs, lens, amax = items["bytes"].tolist(), items["bytes_len"].tolist(), items["abs_max"].tolist()
        datas = _lib.take_bytes_many(dev.index if dev.index is not None else -1, ptrs, lens, CheckpointedBytes if self.checkpoint_stride else None)
        cks = _take_ckpts_many(dev.index if dev.index is not None else -1, items["ckpt"].tolist(), items["n_ckpt"].tolist()) if self.checkpoint_stride else None
        if cks is not None:
            datas = [CheckpointedBytes._adopt(d, ck[0], self.checkpoint_stride, ck[1]) for d, ck in zip(datas, cks)]
        return CompressedBatch(datas, amax, zb, yq)

    def _decompress_stacked(self, strings: Sequence[bytes], abs_maxes: Sequence[int], zero_bitmaps, scales: Tensor,
                            means: Tensor, weights: Tensor, flags: int = 0, stacked_output: bool = False):
        nat = _lib.native()
        if nat is not None and scales.dim() == 4 and scales.shape[0] > 0 and isinstance(zero_bitmaps, Tensor):
            _, scales, means, weights, N, M, h, w, s_item, sc = self._stacked_view(None, scales, means, weights)
            zb = zero_bitmaps
            if len(strings) != N or len(abs_maxes) != N or len(zb) != N:
                raise RuntimeError(f"{N} items in the parameter tensors, {len(strings)} bitstreams")
            if zb.device.type != "cpu" or zb.dtype != torch.int64:
                zb = zb.to("cpu", torch.int64)
            if tuple(zb.shape) != (N, M):
                raise RuntimeError(f"zero bitmaps have shape {tuple(zb.shape)}, expected ({N}, {M})")
            if M > 1 and zb.stride(1) != 1:
                zb = zb.contiguous()
            dev = scales.device
            if not isinstance(strings, list) or not all(type(s_) is bytes or isinstance(s_, bytes) for s_ in strings):
                strings = [s_ if isinstance(s_, bytes) else bytes(s_) for s_ in strings]
            y_hat = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
            nat.decompress_stacked(_lib.ctx_addr(dev.index if dev.index is not None else -1), torch.cuda.current_stream(dev).cuda_stream, strings, abs_maxes,
                                   zb.data_ptr(), zb.stride(0) if N > 1 else M, scales.data_ptr(), means.data_ptr(), weights.data_ptr(), N, M, h * w, s_item,
                                   M * sc, sc, _lib.FGMM_F16 if scales.dtype == torch.float16 else _lib.FGMM_F32, flags, self._mode(), int(self.clamp_scales),
                                   y_hat.data_ptr(), CheckpointedBytes)  # (the out-of-band notes of the bitstreams that carry them: read off the objects)
            return y_hat if stacked_output else list(y_hat.unbind(0))
        items, keep, N, M, h, w, dev = self._stacked_items(None, scales, means, weights, flags)
        if len(strings) != N or len(abs_maxes) != N or len(zero_bitmaps) != N:
            raise RuntimeError(f"{N} items in the parameter tensors, {len(strings)} bitstreams")
        if N == 0:
            return torch.empty((0, 1, M, h, w), dtype=torch.float32, device=dev) if stacked_output else []
        if isinstance(zero_bitmaps, Tensor):
            zb = zero_bitmaps
        else:
            zb = torch.stack([z.to("cpu", torch.int64) for z in zero_bitmaps])
        if zb.device.type != "cpu" or zb.dtype != torch.int64:
            zb = zb.to("cpu", torch.int64)
        if tuple(zb.shape) != (N, M):
            raise RuntimeError(f"zero bitmaps have shape {tuple(zb.shape)}, expected ({N}, {M})")
        if M > 1 and zb.stride(1) != 1:
            zb = zb.contiguous()
        zb_row = zb.stride(0) if N > 1 else M  # rows may be a strided view of a larger batch (a codec stage: every second one)
        data = [s_ if isinstance(s_, bytes) else bytes(s_) for s_ in strings]
        bufs = (C.c_char_p * N)(*data)  # borrowed pointers into the bytes objects (kept alive by `data`)
        y_hat = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
        items["bytes"] = np.frombuffer(bufs, dtype=np.uint64)
        items["bytes_len"] = [len(d) for d in data]
        if any(isinstance(d, CheckpointedBytes) for d in data):  # out-of-band checkpoints of the streams that carry them
            cks = [(d._ck[3], d._ck[2], d._ck[4]) if isinstance(d, CheckpointedBytes) else (0, 0, 0) for d in data]  # (address, count, stride)
            items["ckpt"], items["n_ckpt"], items["ckpt_stride"] = (np.array(c, dtype=np.uint64) for c in zip(*cks))
        items["abs_max"] = np.asarray(abs_maxes, dtype=np.int64)
        items["yq_out"] = np.uint64(y_hat.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
        items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(zb_row * 8)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.lib().fgmm_gmc_decompress_batch(_lib.ctx(dev.index if dev.index is not None else -1), stream,
                                                  C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(),
            int(self.clamp_scales))
        _lib.check(rc, "GaussianMixtureConditional.decompress")
        return y_hat if stacked_output else list(y_hat.unbind(0))

    def compress_batch(self, ys, scales, means, weights, *, weights_are_logits: bool = False):
        """N independent ``compress`` calls in one native call (kernels batched over items, one host rANS worker
        per bitstream).  Returns a sequence of ``((bytes, abs_max, zero_bitmap_cpu), y_q)`` - a list, or for stacked inputs a
        ``CompressedBatch`` (the same sequence, held stacked).

        The items are given either as sequences of ``[1, M, h, w]`` / ``[1, K*M, h, w]`` tensors (any mix of shapes)
        or, for items of one shape, stacked: ``y [N, M, h, w]``, parameters ``[N, K*M, h, w]`` — what a network
        evaluated on a batch of images produces, and the cheaper form (one check, one allocation per output).

        ``weights_are_logits``: ``weights`` holds the parameter head's logits and the softmax over K
        (latent_codecs/gaussian_mixture_conditional.py:198-202) runs inside the HIP kernels — the pi plane is never written
        and read back.  ``decompress`` must then be given the logits too."""
        if self.K != _lib.FGMM_K:
            raise RuntimeError(f"K = {self.K}: the coder is bound for K = 4 only (as the reference's)")
        flags = _lib.FGMM_PARAMS_LOGITS if weights_are_logits else 0
        if isinstance(ys, Tensor):
            return self._compress_stacked(ys, scales, means, weights, flags)
        n_items = len(ys)
        nat = _lib.native()
        if nat is not None and n_items > 0:  # the compiled boundary: items as tuples of integers, the bitstreams made by the call (fgmm_sink)
            keep = []
            tuples, outs, dev, dt = [], [], None, None
            ms = []
            for i in range(n_items):
                yp, sp, mp, wp, M, hw, sk, sc, d, dti = self._item_ints(ys[i], scales[i], means[i], weights[i], keep)
                dev, dt = dev or d, dt or dti
                if d != dev:
                    raise RuntimeError("all items of a batch must be on one device")
                if dti != dt:
                    raise RuntimeError("all items of a batch must have parameters of one dtype")
                yq = torch.empty_like(keep[-1])
                outs.append(yq)
                ms.append(M)
                tuples.append([yp, sp, mp, wp, M, hw, sk, sc, yq.data_ptr(), 0])
            zb_all = torch.empty(sum(ms), dtype=torch.int64)  # the zero bitmaps of all items: one allocation, a view per item
            at, base = 0, zb_all.data_ptr()
            for t, M in zip(tuples, ms):
                t[9] = base + 8 * at
                at += M
            di = dev.index if dev.index is not None else -1
            strings, amax = nat.compress_items(_lib.ctx_addr(di), torch.cuda.current_stream(dev).cuda_stream, [tuple(t) for t in tuples],
                                               _lib.FGMM_F16 if dt == torch.float16 else _lib.FGMM_F32, flags, self._mode(), int(self.clamp_scales),
                                               self.checkpoint_stride, CheckpointedBytes if self.checkpoint_stride else None)
            bitmaps = zb_all.split(ms)
            return [((strings[i], amax[i], bitmaps[i]), outs[i].view_as(ys[i])) for i in range(n_items)]
        items = (_lib.fgmm_item * n_items)()
        keep: list = []
        outs, bitmaps = [], []
        dev = None
        for i in range(n_items):
            it, M, hw, d = self._item(ys[i], scales[i], means[i], weights[i], keep, flags)
            dev = dev or d
            if d != dev:
                raise RuntimeError("all items of a batch must be on one device")
            yq = torch.empty_like(keep[-1])
            zb = torch.empty(M, dtype=torch.int64)
            it.yq_out, it.zero_bitmap = yq.data_ptr(), zb.data_ptr()
            it.ckpt_stride = self.checkpoint_stride
            items[i] = it
            outs.append(yq)
            bitmaps.append(zb)
        if n_items == 0:
            return []
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.lib().fgmm_gmc_compress_batch(_lib.ctx(dev.index if dev.index is not None else -1), stream, items,
                                                n_items, self._mode(), int(self.clamp_scales))
        _lib.check(rc, "GaussianMixtureConditional.compress")
        res = []
        datas = _lib.take_bytes_many(dev.index if dev.index is not None else -1, [items[i].bytes for i in range(n_items)],
                                     [int(items[i].bytes_len) for i in range(n_items)], CheckpointedBytes if self.checkpoint_stride else None)
        cks = _take_ckpts_many(dev.index if dev.index is not None else -1, [items[i].ckpt for i in range(n_items)],
                               [int(items[i].n_ckpt) for i in range(n_items)]) if self.checkpoint_stride else None
        for i in range(n_items):
            data = datas[i]
            if cks is not None:
                data = CheckpointedBytes._adopt(data, cks[i][0], self.checkpoint_stride, cks[i][1])
            res.append(((data, int(items[i].abs_max), bitmaps[i]), outs[i].view_as(ys[i])))
        return res

    def compress_head_batch(self, y: Tensor, x: Tensor, head: "ParameterHead") -> CompressedBatch:
        """``compress_batch`` with the parameter head FUSED into the encode-side CDF kernel: ``y [N, M, h, w]`` latents, ``x [N, c_in, h, w]``
        the features the head's last convolution reads.  The 3*K parameters of a latent go from the matrix cores' accumulators into
        its table entry; the parameter tensors are never written.  The bitstreams are those of
        ``compress_batch(y, *head.params(x), weights_are_logits=True)``, byte for byte."""
        if self.K != _lib.FGMM_K or head.K != self.K:
            raise RuntimeError(f"K = {self.K}: the coder is bound for K = 4 only (as the reference's)")
        x = head._features(x)
        N, _, h, w = x.shape
        M = head.M
        if y.dim() != 4 or tuple(y.shape) != (N, M, h, w) or y.dtype != torch.float32 or y.device != x.device:
            raise RuntimeError(f"y must be float32 [{N}, {M}, {h}, {w}] on the features' device; got {tuple(y.shape)}")
        dev = x.device
        if N == 0:
            return CompressedBatch([], [], torch.empty((0, M), dtype=torch.int64), torch.empty((0, 1, M, h, w), dtype=torch.float32, device=dev))
        y = y.contiguous()
        nat = _lib.native()
        if nat is not None:
            yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
            zb = torch.empty((N, M), dtype=torch.int64)
            strings, amax = nat.compress_head_stacked(
                _lib.ctx_addr(dev.index if dev.index is not None else -1), torch.cuda.current_stream(dev).cuda_stream, y.data_ptr(), x.data_ptr(), head._h.value,
                N, M, head.c_in, h * w, self._mode(), int(self.clamp_scales), self.checkpoint_stride, yq.data_ptr(), zb.data_ptr(),
                CheckpointedBytes if self.checkpoint_stride else None)
            return CompressedBatch(strings, amax, zb, yq)
        items = np.zeros(N, _lib.ITEM_DTYPE)
        rng = np.arange(N, dtype=np.uint64)
        items["y"] = np.uint64(y.data_ptr()) + rng * np.uint64(M * h * w * 4)
        items["M"], items["K"], items["hw"] = M, self.K, h * w
        yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=dev)
        zb = torch.empty((N, M), dtype=torch.int64)
        items["yq_out"] = np.uint64(yq.data_ptr()) + rng * np.uint64(M * h * w * 4)
        items["zero_bitmap"] = np.uint64(zb.data_ptr()) + rng * np.uint64(M * 8)
        items["ckpt_stride"] = self.checkpoint_stride
        xs = (C.c_void_p * N)(*[x.data_ptr() + i * head.c_in * h * w * 4 for i in range(N)])
        di = dev.index if dev.index is not None else -1
        rc = _lib.lib().fgmm_gmc_compress_head_batch(_lib.ctx(di), torch.cuda.current_stream(dev).cuda_stream,
                                                     C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), xs, N, head._h, self._mode(), int(self.clamp_scales))
        _lib.check(rc, "GaussianMixtureConditional.compress_head_batch")
        datas = _lib.take_bytes_many(di, items["bytes"].tolist(), items["bytes_len"].tolist(), CheckpointedBytes if self.checkpoint_stride else None)
        if self.checkpoint_stride:
            cks = _take_ckpts_many(di, items["ckpt"].tolist(), items["n_ckpt"].tolist())
            datas = [CheckpointedBytes._adopt(d, ck[0], self.checkpoint_stride, ck[1]) for d, ck in zip(datas, cks)]
        return CompressedBatch(datas, items["abs_max"].tolist(), zb, yq)

    def compress(self, y: Tensor, scales: Tensor, means: Tensor, weights: Tensor, *, weights_are_logits: bool = False):
        """-> ((bytes, abs_max, zero_bitmap), y_quantized)     (entropy_models.py:833-867)"""
        ((data, abs_max, zb), yq), = self.compress_batch([y], [scales], [means], [weights], weights_are_logits=weights_are_logits)
        return (data, abs_max, zb.to(y.device)), yq

    def decompress_batch(self, strings: Sequence[bytes], abs_maxes: Sequence[int], zero_bitmaps, scales, means,
                         weights, *, weights_are_logits: bool = False, stacked_output: bool = False):
        """N independent ``decompress`` calls in one native call; parameters as sequences of ``[1, K*M, h, w]``
        tensors or stacked ``[N, K*M, h, w]`` (see ``compress_batch``; ``zero_bitmaps`` may then be one ``[N, M]`` tensor).
        Returns N ``[1, M, h, w]`` tensors - with ``stacked_output`` (stacked parameters only) the one ``[N, 1, M, h, w]``
        tensor they are views of."""
        if self.K != _lib.FGMM_K:
            raise RuntimeError(f"K = {self.K}: the coder is bound for K = 4 only (as the reference's)")
        flags = _lib.FGMM_PARAMS_LOGITS if weights_are_logits else 0
        if isinstance(scales, Tensor):
            return self._decompress_stacked(strings, abs_maxes, zero_bitmaps, scales, means, weights, flags, stacked_output)
        if stacked_output:
            raise RuntimeError("stacked_output needs stacked parameters ([N, K*M, h, w] tensors)")
        n_items = len(strings)
        nat = _lib.native()
        if nat is not None and n_items > 0:
            keep = []
            tuples, outs, dev, dt = [], [], None, None
            for i in range(n_items):
                _, sp, mp, wp, M, hw, sk, sc, d, dti = self._item_ints(None, scales[i], means[i], weights[i], keep)
                dev, dt = dev or d, dt or dti
                if d != dev:
                    raise RuntimeError("all items of a batch must be on one device")
                if dti != dt:
                    raise RuntimeError("all items of a batch must have parameters of one dtype")
                zb = zero_bitmaps[i]
                if zb.device.type != "cpu" or zb.dtype != torch.int64 or not zb.is_contiguous():
                    zb = zb.to("cpu", torch.int64).contiguous()
                if zb.numel() != M:
                    raise RuntimeError(f"zero_bitmap has {zb.numel()} entries, expected {M}")
                shp = scales[i].shape
                y_hat = torch.empty((1, M, shp[2], shp[3]), dtype=torch.float32, device=d)
                keep.append(zb)
                outs.append(y_hat)
                tuples.append((sp, mp, wp, M, hw, sk, sc, y_hat.data_ptr(), zb.data_ptr()))
            data = strings if isinstance(strings, list) and all(isinstance(s_, bytes) for s_ in strings) else [s_ if isinstance(s_, bytes) else bytes(s_) for s_ in strings]
            nat.decompress_items(_lib.ctx_addr(dev.index if dev.index is not None else -1), torch.cuda.current_stream(dev).cuda_stream, data,
                                 [int(a) for a in abs_maxes], tuples, _lib.FGMM_F16 if dt == torch.float16 else _lib.FGMM_F32, flags, self._mode(),
                                 int(self.clamp_scales), CheckpointedBytes)
            return outs
        items = (_lib.fgmm_item * n_items)()
        keep: list = []
        outs = []
        dev = None
        for i in range(n_items):
            it, M, hw, d = self._item(None, scales[i], means[i], weights[i], keep, flags)
            dev = dev or d
            if d != dev:
                raise RuntimeError("all items of a batch must be on one device")
            zb = zero_bitmaps[i]
            if zb.device.type != "cpu" or zb.dtype != torch.int64 or not zb.is_contiguous():
                zb = zb.to("cpu", torch.int64).contiguous()
            if zb.numel() != M:
                raise RuntimeError(f"zero_bitmap has {zb.numel()} entries, expected {M}")
            data = strings[i] if isinstance(strings[i], bytes) else bytes(strings[i])
            buf = C.c_char_p(data)  # borrowed pointer into the bytes object (kept alive below), no copy
            shp = scales[i].shape
            y_hat = torch.empty((1, M, shp[2], shp[3]), dtype=torch.float32, device=d)
            it.yq_out, it.zero_bitmap = y_hat.data_ptr(), zb.data_ptr()
            it.abs_max = int(abs_maxes[i])
            it.bytes, it.bytes_len = C.cast(buf, C.c_void_p), len(data)
            if isinstance(data, CheckpointedBytes) and data._ck[2]:
                it.ckpt, it.n_ckpt, it.ckpt_stride = data._ck[3], data._ck[2], data._ck[4]
            items[i] = it
            keep += [zb, buf, data]
            outs.append(y_hat)
        if n_items == 0:
            return []
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = _lib.lib().fgmm_gmc_decompress_batch(_lib.ctx(dev.index if dev.index is not None else -1), stream, items,
                                                  n_items, self._mode(), int(self.clamp_scales))
        _lib.check(rc, "GaussianMixtureConditional.decompress")
        return outs

    def decompress(self, strings: bytes, abs_max: int, zero_bitmap: Tensor, scales: Tensor, means: Tensor,
                   weights: Tensor, *, weights_are_logits: bool = False) -> Tensor:
        """-> y_hat [1, M, h, w] float32     (entropy_models.py:872-910)"""
        return self.decompress_batch([strings], [abs_max], [zero_bitmap], [scales], [means], [weights],
                                     weights_are_logits=weights_are_logits)[0]


class EntropyBottleneckCoder(nn.Module):
    """The coding half of the reference's ``EntropyBottleneck`` (compressai/entropy_models/entropy_models.py:605-618 over
    ``EntropyModel.compress / decompress``, :237-327) for the `z` hyper-latent, given the tables its ``update()`` built:

        coder = EntropyBottleneckCoder.from_entropy_bottleneck(eb)      # any object with the reference's buffers
        strings = coder.compress(z)                                     # list of bytes, one per batch element
        z_hat = coder.decompress(strings, z.shape[-2:])                 # [N, C, h, w] on the tables' device

    Same bytes as the reference for the same tables and the same ``z`` (golden G5 / G8); what differs is the plumbing:
    the reference turns the symbol tensor, the index tensor AND the whole CDF matrix into Python lists on every call
    (:257-266); here the tables are converted once, the symbols cross as one int32 copy, and the indexes (channel of
    every element, ``_build_indexes`` :555-566) are built once per shape.  Integer work on the host, as the reference's;
    the learned density model behind the tables (``_likelihood``, ``update()``) stays the caller's (stock CompressAI)."""

    def __init__(self, quantized_cdf: Tensor, cdf_length: Tensor, offset: Tensor, medians: Tensor):
        super().__init__()
        from .ans import _Tables

        if quantized_cdf.dim() != 2 or cdf_length.dim() != 1 or offset.dim() != 1:
            raise ValueError("expected _quantized_cdf [C, L], _cdf_length [C], _offset [C] (run update() first)")
        self.channels = int(quantized_cdf.size(0))
        if cdf_length.numel() != self.channels or offset.numel() != self.channels or medians.numel() != self.channels:
            raise ValueError("tables and medians must describe the same number of channels")
        self._tables = _Tables(quantized_cdf.detach().cpu().numpy(), cdf_length.detach().cpu().numpy(), offset.detach().cpu().numpy())
        self.register_buffer("medians", medians.detach().reshape(-1).to(torch.float32).clone())
        self._index_cache = {}

    @classmethod
    def from_entropy_bottleneck(cls, eb) -> "EntropyBottleneckCoder":
        """``eb``: the reference's (or stock CompressAI's) ``EntropyBottleneck`` after ``update()``"""
        return cls(eb._quantized_cdf, eb._cdf_length, eb._offset, eb.quantiles[:, 0, 1])

    def _indexes(self, spatial: int) -> np.ndarray:
        idx = self._index_cache.get(spatial)
        if idx is None:
            idx = np.repeat(np.arange(self.channels, dtype=np.int32), spatial)
            self._index_cache[spatial] = idx
        return idx

    def compress(self, x: Tensor, return_dequantized: bool = False):
        """-> list of bytes, one per batch element; with ``return_dequantized`` also what ``decompress`` of those strings
        yields — ``quantize(x) + medians`` computed where ``x`` lives (:158-176, :197-204: the coder is lossless on the
        int32 symbols, so decoding them back is one elementwise op, not a host encode -> host decode -> upload)."""
        if x.dim() < 2 or x.size(1) != self.channels:
            raise ValueError(f"expected [N, {self.channels}, ...], got {tuple(x.shape)}")
        med = self.medians.to(x.device).reshape(1, -1, *([1] * (x.dim() - 2)))
        # quantize(inputs, "symbols", means): round(x - means).int()  (:158-176) — on the tensor's device, ONE copy out
        sym_dev = torch.round(x - med).to(torch.int32)
        sym = sym_dev.reshape(x.size(0), -1).cpu().numpy()
        if return_dequantized:
            return self._encode_rows(sym), sym_dev.to(torch.float32) + med  # dequantize: outputs.type_as(means) + means
        return self._encode_rows(sym)

    def _encode_rows(self, sym: np.ndarray) -> List[bytes]:
        idx = self._indexes(sym.shape[1] // self.channels)
        L = _lib.lib()
        out = []
        for i in range(sym.shape[0]):
            row = np.ascontiguousarray(sym[i])
            o, n = C.c_void_p(), C.c_size_t()
            _lib.check(L.fgmm_encode_with_indexes(row.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), row.size,
                                                  *self._tables.args(), C.byref(o), C.byref(n)), "EntropyBottleneck.compress")
            out.append(_lib.take_bytes(o, n.value))
        return out

    def decompress(self, strings: Sequence[bytes], size) -> Tensor:
        if not isinstance(strings, (tuple, list)):
            raise ValueError("Invalid `strings` parameter type.")
        size = tuple(int(v) for v in size)
        spatial = int(np.prod(size)) if size else 1
        idx = self._indexes(spatial)
        L = _lib.lib()
        sym = np.empty((len(strings), idx.size), np.int32)
        for i, sdata in enumerate(strings):
            sdata = bytes(sdata)
            _lib.check(L.fgmm_decode_with_indexes(sdata, len(sdata), idx.ctypes.data_as(C.c_void_p), idx.size, *self._tables.args(),
                                                  sym[i].ctypes.data_as(C.c_void_p)), "EntropyBottleneck.decompress")
        dev = self.medians.device
        med = self.medians.reshape(1, -1, *([1] * len(size)))
        # dequantize(outputs, means): outputs.type_as(means) + means  (:197-204)
        return torch.from_numpy(sym).to(dev).reshape(len(strings), self.channels, *size).to(torch.float32) + med

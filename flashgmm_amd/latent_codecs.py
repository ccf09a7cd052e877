"""``GaussianMixtureConditionalLatentCodec`` — the caller right above the path, same compress/decompress contract as
the reference's (compressai/latent_codecs/gaussian_mixture_conditional.py:43-202), built on
``flashgmm_amd.GaussianMixtureConditional``:

    codec.compress(y, ctx_params)             -> {"strings": [(bytes, abs_max, zero_bitmap)], "shape": (h, w), "y_hat": y_q}
    codec.decompress(strings, shape, ctx_params) -> {"y_hat": y_hat}

``entropy_parameters(ctx_params)`` yields ``[1, 3*K*M, h, w]``; it is split with ``chunk(3, 1)`` into scales, means and
mixture logits, the logits are soft-maxed over K on the ``[1, K, M, h, w]`` view (:183-202) — all views / stock torch ops
on the GPU — and handed to the entropy model in place.  Both quantizers of the reference are kept ("noise": code round(y);
"weighted_mean_ste": code round(y - sum_k pi_k mu_k) against re-centred means, :135-145, :167-179).
``forward`` (training-time likelihoods, :99-125) is outside the entropy-coding path and not provided.

``CheckerboardLatentCodec`` and ``ChannelGroupsLatentCodec`` mirror the two codecs that drive the path in the
reference's GMM models (compressai/latent_codecs/checkerboard.py:275-330, channel_groups.py:111-158;
models/ckbd_gmm.py:111, models/elic_gmm.py:198-219): same constructor arguments, same ``compress / decompress``
contract and nested ``{"strings": [...], "shape": ...}`` result, the networks (``context_prediction``,
``entropy_parameters``, ``channel_context``) supplied by the caller as torch modules.  What changes is the schedule:
the checkerboard split / merge is one HIP kernel each, and on encode both halves are coded in ONE batched call —
the anchors' reconstruction that the non-anchor parameters depend on is ``round(.)`` of data the encoder already
holds (checkerboard.py:282-288), so nothing has to wait for the anchors' bitstream.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from itertools import accumulate

from .entropy_models import EntropyBottleneckCoder, GaussianMixtureConditional, ParameterHead
from .ops import ckbd_embed, ckbd_unembed

__all__ = ["GaussianMixtureConditionalLatentCodec", "CheckerboardLatentCodec", "ChannelGroupsLatentCodec", "HyperLatentCodec",
           "HyperpriorLatentCodec"]


class GaussianMixtureConditionalLatentCodec(nn.Module):
    def __init__(self, K: int = 4, gaussian_mixture_conditional: Optional[GaussianMixtureConditional] = None,
                 entropy_parameters: Optional[nn.Module] = None, quantizer: str = "noise",
                 chunks: Tuple[str, ...] = ("scales", "means", "weights"), mode=None, param_dtype: torch.dtype = torch.float32,
                 fuse_softmax: bool = False, checkpoint_stride: int = 0, **kwargs: Any):
        super().__init__()
        if param_dtype not in (torch.float32, torch.float16):
            raise ValueError("param_dtype must be torch.float32 or torch.float16")
        self.param_dtype = param_dtype  # float16: BASELINE configs[4], "fp16 (mu, sigma, pi) with fp32 CDF accumulate"
        # fuse_softmax: the softmax over K (:198-202) runs inside the HIP kernels, on the head's logits in place — the pi
        # plane (16 B / latent written, 16 read back) never exists.  One fixed fp32 sequence on both sides of the codec:
        # streams are self-consistent on any device; they are NOT the streams of the un-fused path (torch.softmax's pi
        # differs in the last bit), so encoder and decoder must agree on this switch like on the Phi approximation.
        if fuse_softmax and (quantizer != "noise" or param_dtype != torch.float32):
            raise ValueError("fuse_softmax needs quantizer='noise' (the re-centring quantizer uses pi itself) and float32 planes")
        self.fuse_softmax = bool(fuse_softmax)
        if quantizer not in ("noise", "weighted_mean_ste"):
            raise ValueError(f"quantizer {quantizer} not supported")
        if tuple(chunks) != ("scales", "means", "weights"):
            raise ValueError("a Gaussian-mixture codec needs chunks = ('scales', 'means', 'weights')")
        self.K = K
        self.quantizer = quantizer
        # checkpoint_stride > 0: the strings are CheckpointedBytes — the same bitstreams plus out-of-band notes of the coder
        # state every that many symbols, with which ONE bitstream decodes on all host workers (a single image: 3.7 -> 1.6 ms
        # on a Kodak image, 108 -> 25 ms on a 4K ELIC image); flashgmm_amd.container stores them with the stream
        self.gaussian_mixture_conditional = gaussian_mixture_conditional or GaussianMixtureConditional(
            K=K, mode=mode, checkpoint_stride=checkpoint_stride)
        self.entropy_parameters = entropy_parameters or nn.Identity()
        self.chunks = tuple(chunks)

    # :183-196
    def _chunk(self, params: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        scales, means, weights = params.chunk(3, 1)
        return scales, means, weights

    # :198-202
    def _reshape_gmm_weight(self, weight: Tensor) -> Tensor:
        B, KM, H, W = weight.shape
        weight = torch.reshape(weight, (B, self.K, KM // self.K, H, W))
        weight = nn.functional.softmax(weight, dim=1)
        return torch.reshape(weight, (B, KM, H, W))

    def _params(self, ctx_params: Tensor):
        scales_hat, means_hat, weights = self._chunk(self.entropy_parameters(ctx_params))
        return scales_hat, means_hat, (weights if self.fuse_softmax else self._reshape_gmm_weight(weights))

    def _planes(self, scales: Tensor, means: Tensor, weights: Tensor):
        """the planes as the entropy model gets them: float32 as they are, or float16 copies — weights rounded TOWARD ZERO,
        because the algorithm needs sum_k pi_k <= 1 after widening (tests/synth.py to_float16_planes does it for the synthetic workloads)"""
        if self.param_dtype == torch.float32:
            return scales, means, weights
        w16 = weights.to(torch.float16)
        over = w16.to(torch.float32) > weights
        w16 = torch.where(over, torch.nextafter(w16, torch.zeros_like(w16)), w16)
        return scales.to(torch.float16), means.to(torch.float16), w16

    def _recentre(self, means_hat: Tensor, weights: Tensor):
        """weighted_mean_ste: sum_k pi_k mu_k and the means relative to it (:139-144)"""
        B, KM, H, W = means_hat.shape
        me = means_hat.view(B, self.K, KM // self.K, H, W)
        we = weights.view(B, self.K, KM // self.K, H, W)
        weighted_sum = torch.sum(me * we, dim=1)
        return weighted_sum, (me - weighted_sum.unsqueeze(1)).reshape(B, KM, H, W)

    def coder_inputs(self, y: Tensor, ctx_params: Tensor):
        """What ``compress`` hands the entropy model: ``(y_to_code, scales, means, weights)``.  ``round(y_to_code)`` is
        the ``y_hat`` that ``compress`` returns (:127-149) — known before any coding happens."""
        scales_hat, means_hat, weights = self._params(ctx_params)
        if self.quantizer == "noise":
            return (y, *self._planes(scales_hat, means_hat, weights))
        weighted_sum, means_rel = self._recentre(means_hat, weights)
        d = y - weighted_sum
        # quantize_ste (compressai/ops/ops.py:66-80) is (round(d) - d) + d: the value of round(d), but +0.0 where
        # round(d) is -0.0 — kept, so that the returned y_hat has the reference's bits
        return ((torch.round(d) - d) + d, *self._planes(scales_hat, means_rel, weights))

    def compress(self, y: Tensor, ctx_params: Tensor) -> Dict[str, Any]:
        y_code, scales_hat, means_hat, weights = self.coder_inputs(y, ctx_params)
        y_strings, y_hat = self.gaussian_mixture_conditional.compress(y_code, scales_hat, means_hat, weights,
                                                                      weights_are_logits=self.fuse_softmax)
        return {"strings": [y_strings], "shape": y.shape[2:4], "y_hat": y_hat}

    def compress_many(self, prepared: List[Tuple[Tensor, Tensor, Tensor, Tensor]]) -> List[Dict[str, Any]]:
        """``compress`` of several ``coder_inputs`` results in one batched native call."""
        ys, ss, ms, ws = zip(*prepared)
        res = self.gaussian_mixture_conditional.compress_batch(list(ys), list(ss), list(ms), list(ws), weights_are_logits=self.fuse_softmax)
        return [{"strings": [(b, a, zb.to(y.device))], "shape": y.shape[2:4], "y_hat": yq}
                for ((b, a, zb), yq), y in zip(res, ys)]

    def decompress(self, strings: List[Any], shape: Tuple[int, int], ctx_params: Tensor, **kwargs: Any) -> Dict[str, Any]:
        (y_strings,) = strings
        scales_hat, means_hat, weights = self._params(ctx_params)
        if self.quantizer == "noise":
            y_hat = self.gaussian_mixture_conditional.decompress(*y_strings, *self._planes(scales_hat, means_hat, weights),
                                                                 weights_are_logits=self.fuse_softmax)
        else:
            weighted_sum, means_rel = self._recentre(means_hat, weights)
            y_hat = self.gaussian_mixture_conditional.decompress(*y_strings, *self._planes(scales_hat, means_rel, weights)) + weighted_sum
        assert tuple(y_hat.shape[2:4]) == tuple(shape)
        return {"y_hat": y_hat}

    # ---- the parameter head fused: `feat` = the input of entropy_parameters' LAST layer, `head` = a ParameterHead
    # of that layer.  The codec owning the parameter network (CheckerboardLatentCodec(fuse_head=True)) calls these.
    def _check_head(self):
        if self.quantizer != "noise" or self.param_dtype != torch.float32:
            raise RuntimeError("the fused parameter head needs quantizer='noise' and float32 parameters (as fuse_softmax)")

    def compress_many_head(self, items: List[Tuple[Tensor, Tensor]], head: ParameterHead) -> List[Dict[str, Any]]:
        """``compress`` of several ``(y [1, M, h, w], feat [1, c_in, h, w])`` of one shape in ONE native call, the parameters computed
        by the head inside the encode-side kernel (never written to HBM)"""
        self._check_head()
        ys, feats = zip(*items)
        res = self.gaussian_mixture_conditional.compress_head_batch(torch.cat(list(ys)), torch.cat(list(feats)), head)
        return [{"strings": [(b, a, zb.to(y.device))], "shape": y.shape[2:4], "y_hat": yq} for ((b, a, zb), yq), y in zip(res, ys)]

    def decompress_many_head(self, strings: List[List[Any]], shape: Tuple[int, int], feats: List[Tensor], head: ParameterHead) -> List[Dict[str, Any]]:
        """the decoder's side: the SAME arithmetic gives the parameter tensors (``head.params``), the table kernels take the logits"""
        self._check_head()
        scales, means, logits = head.params(torch.cat(list(feats)))
        items = [st[0] for st in strings]
        outs = self.gaussian_mixture_conditional.decompress_batch([it[0] for it in items], [it[1] for it in items],
                                                                  torch.stack([it[2].to("cpu", torch.int64) for it in items]), scales, means, logits,
                                                                  weights_are_logits=True)
        for y_hat in outs:
            assert tuple(y_hat.shape[2:4]) == tuple(shape)
        return [{"y_hat": y_hat} for y_hat in outs]

    def decompress_many(self, strings: List[List[Any]], shape: Tuple[int, int], ctx_params: List[Tensor]) -> List[Dict[str, Any]]:
        """``decompress`` of N independent items (the same stage of N images) in ONE batched native call: N host decoders
        side by side instead of N calls with one decoder each."""
        planes, sums = [], []
        for ctx in ctx_params:
            scales_hat, means_hat, weights = self._params(ctx)
            if self.quantizer == "noise":
                planes.append(self._planes(scales_hat, means_hat, weights))
                sums.append(None)
            else:
                weighted_sum, means_rel = self._recentre(means_hat, weights)
                planes.append(self._planes(scales_hat, means_rel, weights))
                sums.append(weighted_sum)
        items = [st[0] for st in strings]  # (bytes, abs_max, zero_bitmap) of every item
        ss, ms, ws = (list(t) for t in zip(*planes))
        outs = self.gaussian_mixture_conditional.decompress_batch([it[0] for it in items], [it[1] for it in items], [it[2] for it in items],
                                                                  ss, ms, ws, weights_are_logits=self.fuse_softmax)
        res = []
        for y_hat, add in zip(outs, sums):
            assert tuple(y_hat.shape[2:4]) == tuple(shape)
            res.append({"y_hat": y_hat if add is None else y_hat + add})
        return res

    def forward(self, y: Tensor, ctx_params: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path; use the reference's "
                                  "GaussianMixtureConditionalLatentCodec.forward (pure torch)")


class CheckerboardLatentCodec(nn.Module):
    """Two-pass checkerboard context model, ``compress`` / ``decompress`` only (checkerboard.py:275-330).

    ``latent_codec["y"]`` is a ``GaussianMixtureConditionalLatentCodec`` (its own ``entropy_parameters`` is the
    identity: the parameter network sits here, as in the reference's models); ``entropy_parameters`` must be
    pointwise (the reference's warning, checkerboard.py:80-82); batch size 1 (checkerboard.py:307)."""

    def __init__(self, latent_codec: Optional[Dict[str, nn.Module]] = None, entropy_parameters: Optional[nn.Module] = None,
                 context_prediction: Optional[nn.Module] = None, anchor_parity: str = "even", forward_method: str = "twopass",
                 fuse_head=False, **kwargs: Any):
        super().__init__()
        if anchor_parity not in ("even", "odd"):
            raise ValueError(f"anchor_parity {anchor_parity!r}")
        # fuse_head: the LAST layer of entropy_parameters - nn.Conv2d(c_in, 3*K*M, 1), models/ckbd_gmm.py:115-121 - runs inside the
        # library (flashgmm_amd.ParameterHead: matrix cores, one fixed summation order, fused with the encode-side CDF kernel; the
        # softmax over K and the sigma clamp with it): encoder and decoder must agree on this switch like on the Phi approximation -
        # the parameters differ from torch's convolution in their last bits
        # fuse_head: True / "f32" (the exact form) or "bf16x6" (ParameterHead's faster arithmetic: MI355X on both sides)
        if fuse_head not in (False, True, "f32", "bf16x6"):
            raise ValueError(f"fuse_head {fuse_head!r}")
        self.fuse_head = bool(fuse_head)
        self._head_arith = "bf16x6" if fuse_head == "bf16x6" else "f32"
        self._head: Optional[ParameterHead] = None
        self._head_key = None
        if self.fuse_head:
            last = entropy_parameters[-1] if isinstance(entropy_parameters, nn.Sequential) and len(entropy_parameters) else entropy_parameters
            if not isinstance(last, nn.Conv2d) or last.kernel_size != (1, 1) or last.stride != (1, 1) or last.groups != 1:
                raise ValueError("fuse_head needs entropy_parameters to end in a plain 1x1 nn.Conv2d (models/ckbd_gmm.py:115-121)")
        self.anchor_parity = anchor_parity
        self.non_anchor_parity = {"odd": "even", "even": "odd"}[anchor_parity]
        self.forward_method = forward_method
        self.entropy_parameters = entropy_parameters or nn.Identity()
        self.context_prediction = context_prediction or nn.Identity()
        self.latent_codec = nn.ModuleDict(latent_codec if latent_codec is not None
                                          else {"y": GaussianMixtureConditionalLatentCodec()})

    def __getitem__(self, key: str) -> nn.Module:
        return self.latent_codec[key]

    # checkerboard.py:333-377, one kernel each
    def unembed(self, y: Tensor) -> Tensor:
        return ckbd_unembed(y, self.anchor_parity)

    def embed(self, y_: Tensor) -> Tensor:
        return ckbd_embed(y_, self.anchor_parity)

    def merge(self, *args: Tensor) -> Tensor:
        return torch.cat(args, dim=1)

    def _head_parts(self):
        """-> (everything of entropy_parameters before its last layer, the ParameterHead of that layer): packed once, again when the
        layer's weights have changed (a checkpoint loaded, a device move)"""
        ep = self.entropy_parameters
        body, last = (ep[:-1], ep[-1]) if isinstance(ep, nn.Sequential) else (nn.Identity(), ep)
        key = (last.weight.data_ptr(), last.weight._version, None if last.bias is None else (last.bias.data_ptr(), last.bias._version))
        if self._head is None or self._head_key != key:
            self._head, self._head_key = ParameterHead(last, K=self.latent_codec["y"].K, arithmetic=self._head_arith), key
        return body, self._head

    def _ctx(self, y_hat_: Tensor, i: int) -> Tensor:
        """context of half i from the halves reconstructed so far (:281-284, :310-313)"""
        y_ctx_i = self.unembed(self.context_prediction(self.embed(y_hat_)))[i]
        return torch.zeros_like(y_ctx_i) if i == 0 else y_ctx_i  # _mask(., "all") for the anchors

    def prepare(self, y: Tensor, side_params: Tensor):
        """Everything of ``compress`` that is not coding: -> (coder inputs of the two halves, y_hat).  The reconstruction
        a half's parameters depend on is ``round(.)`` of data the encoder holds (checkerboard.py:282-288), so the coder
        inputs of BOTH halves — and ``y_hat`` — exist before a single symbol is coded."""
        n, c, h, w = y.shape
        if n != 1:
            raise RuntimeError("batch size 1 (as the reference's coder path, checkerboard.py:307)")
        codec = self.latent_codec["y"]
        y_hat_ = side_params.new_zeros((2, n, c, h, w // 2))
        side_params_ = self.unembed(side_params)
        y_ = self.unembed(y)
        prepared = []
        body = self._head_parts()[0] if self.fuse_head else None
        for i in range(2):
            if self.fuse_head:  # (y, the features the head's last layer reads): the parameters are the encode kernel's business
                prepared.append((y_[i], body(self.merge(self._ctx(y_hat_, i), side_params_[i]))))
            else:
                params_i = self.entropy_parameters(self.merge(self._ctx(y_hat_, i), side_params_[i]))
                prepared.append(codec.coder_inputs(y_[i], params_i))
            y_hat_[i] = torch.round(prepared[i][0])  # what compress() of this half returns as y_hat
        return prepared, self.embed(y_hat_)

    def finish(self, outs: List[Dict[str, Any]], y_hat: Tensor) -> Dict[str, Any]:
        """the two halves' coding results (``codec.compress_many``) -> the codec's result"""
        return {"strings": [o["strings"][0] for o in outs], "shape": y_hat.shape[1:], "y_hat": y_hat}

    def compress(self, y: Tensor, side_params: Tensor) -> Dict[str, Any]:
        prepared, y_hat = self.prepare(y, side_params)
        if self.fuse_head:
            return self.finish(self.latent_codec["y"].compress_many_head(prepared, self._head_parts()[1]), y_hat)
        return self.finish(self.latent_codec["y"].compress_many(prepared), y_hat)

    def decompress(self, strings: List[Any], shape: Tuple[int, ...], side_params: Tensor, **kwargs: Any) -> Dict[str, Any]:
        n = 1
        c, h, w = shape
        codec = self.latent_codec["y"]
        y_hat_ = side_params.new_zeros((2, n, c, h, w // 2))
        side_params_ = self.unembed(side_params)
        for i in range(2):  # sequential by construction: the non-anchor parameters need the decoded anchors
            if self.fuse_head:
                body, head = self._head_parts()
                feat_i = body(self.merge(self._ctx(y_hat_, i), side_params_[i]))
                y_hat_[i] = codec.decompress_many_head([[strings[i]]], (h, w // 2), [feat_i], head)[0]["y_hat"]
                continue
            params_i = self.entropy_parameters(self.merge(self._ctx(y_hat_, i), side_params_[i]))
            y_hat_[i] = codec.decompress([strings[i]], (h, w // 2), params_i)["y_hat"]
        return {"y_hat": self.embed(y_hat_)}

    def decompress_many(self, strings: List[List[Any]], shape: Tuple[int, ...], side_params: List[Tensor]) -> List[Dict[str, Any]]:
        """N images of one shape, STAGE-MAJOR: the anchors of every image in one batched native call, then the non-anchors
        of every image in one — the dependency runs inside an image (checkerboard.py:316-324), not between images, so a
        call keeps N host decoders busy where ``decompress`` image by image keeps one."""
        c, h, w = shape
        codec = self.latent_codec["y"]
        y_hat_ = [sp.new_zeros((2, 1, c, h, w // 2)) for sp in side_params]
        side_ = [self.unembed(sp) for sp in side_params]
        for i in range(2):
            if self.fuse_head:
                body, head = self._head_parts()
                feats = [body(self.merge(self._ctx(yh, i), sd[i])) for yh, sd in zip(y_hat_, side_)]
                outs = codec.decompress_many_head([[st[i]] for st in strings], (h, w // 2), feats, head)
            else:
                params = [self.entropy_parameters(self.merge(self._ctx(yh, i), sd[i])) for yh, sd in zip(y_hat_, side_)]
                outs = codec.decompress_many([[st[i]] for st in strings], (h, w // 2), params)
            for yh, o in zip(y_hat_, outs):
                yh[i] = o["y_hat"]
        return [{"y_hat": self.embed(yh)} for yh in y_hat_]

    def forward(self, y: Tensor, side_params: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path")


class ChannelGroupsLatentCodec(nn.Module):
    """Channel groups coded one after the other, each conditioned on the previous ones (channel_groups.py:111-158);
    ``latent_codec[f"y{k}"]`` is typically a ``CheckerboardLatentCodec`` (models/elic_gmm.py:198-219)."""

    def __init__(self, latent_codec: Optional[Dict[str, nn.Module]] = None,
                 channel_context: Optional[Dict[str, nn.Module]] = None, *, groups: List[int], **kwargs: Any):
        super().__init__()
        self.groups = list(groups)
        self.groups_acc = list(accumulate(self.groups, initial=0))
        self.channel_context = nn.ModuleDict(channel_context)
        self.latent_codec = nn.ModuleDict(latent_codec)

    def __getitem__(self, key: str) -> nn.Module:
        return self.latent_codec[key]

    def merge_y(self, *args: Tensor) -> Tensor:
        return torch.cat(args, dim=1)

    def merge_params(self, *args: Tensor) -> Tensor:
        return torch.cat(args, dim=1)

    def _get_ctx_params(self, k: int, side_params: Tensor, y_hat_: Tuple[Tensor, ...]) -> Tensor:
        """parameters group k is coded under: the side parameters, preceded (k > 0) by the channel context computed
        from the reconstructions of groups 0..k-1 (channel_groups.py:166-173)"""
        if k == 0:
            return side_params
        return self.merge_params(self.channel_context[f"y{k}"](self.merge_y(*y_hat_[:k])), side_params)

    def _run(self, y_hat: Tensor, side_params: Tensor, code_group):
        """Groups in order; ``code_group(k, params) -> result dict`` codes one group and its ``"y_hat"`` becomes visible
        to the later groups through views of the one full-size ``y_hat`` tensor."""
        views = y_hat.split(self.groups, dim=1)
        results = []
        for k in range(len(self.groups)):
            res = code_group(k, self._get_ctx_params(k, side_params, views))
            views[k].copy_(res["y_hat"])
            results.append(res)
        return results

    @staticmethod
    def _one_call(codecs) -> bool:
        """can all groups be coded in one batched call?  (group codecs that can prepare, one Phi approximation, one clamp
        setting and one parameter dtype: what a batch of the entropy model must share)"""
        if any(getattr(c, "fuse_head", False) for c in codecs):  # (every group has a head of its own: group by group)
            return False
        try:
            keys = {(c.latent_codec["y"].gaussian_mixture_conditional._mode(), c.latent_codec["y"].gaussian_mixture_conditional.clamp_scales,
                     c.latent_codec["y"].param_dtype, c.latent_codec["y"].fuse_softmax) for c in codecs if hasattr(c, "prepare") and hasattr(c, "finish")}
        except (AttributeError, KeyError, TypeError):
            return False
        return len(keys) == 1 and all(hasattr(c, "prepare") for c in codecs)

    def compress(self, y: Tensor, side_params: Tensor) -> Dict[str, Any]:
        parts = torch.split(y, self.groups, dim=1)
        y_hat = torch.zeros_like(y)
        codecs = [self.latent_codec[f"y{k}"] for k in range(len(self.groups))]
        if self._one_call(codecs):
            # Every group's context is round(.) of data the encoder holds (channel_groups.py:117-120 hands over y_hat of
            # the groups before; on the encoder that is known without coding them), so the groups are PREPARED in order
            # and all their bitstreams are coded in ONE batched native call: ten host coders side by side, not ten calls.
            views = y_hat.split(self.groups, dim=1)
            prepared, hats = [], []
            for k, c in enumerate(codecs):
                pk, hk = c.prepare(parts[k], self._get_ctx_params(k, side_params, views))
                views[k].copy_(hk)
                prepared.append(pk)
                hats.append(hk)
            outs = codecs[0].latent_codec["y"].compress_many([p for pk in prepared for p in pk])
            results, at = [], 0
            for c, pk, hk in zip(codecs, prepared, hats):
                results.append(c.finish(outs[at:at + len(pk)], hk))
                at += len(pk)
        else:
            results = self._run(y_hat, side_params, lambda k, params: self.latent_codec[f"y{k}"].compress(parts[k], params))
        per_group = {len(r["strings"]) for r in results}
        if len(per_group) != 1:
            raise RuntimeError("every group codec must emit the same number of strings (channel_groups.py:124)")
        return {"strings": [s for r in results for s in r["strings"]], "shape": [r["shape"] for r in results], "y_hat": y_hat}

    def compress_many(self, ys: List[Tensor], side_params: List[Tensor]) -> List[Dict[str, Any]]:
        """``compress`` of N images: every image is prepared in group order (the networks run per image, batch 1 as the
        reference's), then ALL bitstreams of all images — N x groups x 2 — are coded in ONE batched native call."""
        codecs = [self.latent_codec[f"y{k}"] for k in range(len(self.groups))]
        if not self._one_call(codecs):
            return [self.compress(y, sp) for y, sp in zip(ys, side_params)]
        flat, per_image = [], []
        for y, sp in zip(ys, side_params):
            parts = torch.split(y, self.groups, dim=1)
            y_hat = torch.zeros_like(y)
            views = y_hat.split(self.groups, dim=1)
            prepared, hats = [], []
            for k, c in enumerate(codecs):
                pk, hk = c.prepare(parts[k], self._get_ctx_params(k, sp, views))
                views[k].copy_(hk)
                prepared.append(pk)
                hats.append(hk)
                flat.extend(pk)
            per_image.append((prepared, hats, y_hat))
        outs = codecs[0].latent_codec["y"].compress_many(flat)
        results, at = [], 0
        for prepared, hats, y_hat in per_image:
            rs = []
            for c, pk, hk in zip(codecs, prepared, hats):
                rs.append(c.finish(outs[at:at + len(pk)], hk))
                at += len(pk)
            results.append({"strings": [s_ for r in rs for s_ in r["strings"]], "shape": [r["shape"] for r in rs], "y_hat": y_hat})
        return results

    def decompress_many(self, strings: List[List[Any]], shape: List[Tuple[int, ...]], side_params: List[Tensor]) -> List[Dict[str, Any]]:
        """N images of one shape in flight, STAGE-MAJOR: group by group (channel_groups.py:147-154 — sequential inside an
        image), each group's stage of every image in one batched native call (``CheckerboardLatentCodec.decompress_many``).
        One image alone is a chain of single-bitstream calls — one host decoder at work, fifteen idle; N images fill the
        workers.  Results equal ``decompress`` image by image."""
        per_group = len(strings[0]) // len(self.groups)
        channels = sum(s_[0] for s_ in shape)
        y_hats = [torch.zeros((1, channels, *shape[0][1:]), device=sp.device) for sp in side_params]
        views = [yh.split(self.groups, dim=1) for yh in y_hats]
        for k in range(len(self.groups)):
            codec = self.latent_codec[f"y{k}"]
            params = [self._get_ctx_params(k, sp, v) for sp, v in zip(side_params, views)]
            sub = [st[per_group * k: per_group * (k + 1)] for st in strings]
            if hasattr(codec, "decompress_many"):
                outs = codec.decompress_many(sub, shape[k], params)
            else:
                outs = [codec.decompress(st, shape[k], pr) for st, pr in zip(sub, params)]
            for v, o in zip(views, outs):
                v[k].copy_(o["y_hat"])
        return [{"y_hat": yh} for yh in y_hats]

    def decompress(self, strings: List[Any], shape: List[Tuple[int, ...]], side_params: Tensor, **kwargs: Any) -> Dict[str, Any]:
        per_group = len(strings) // len(self.groups)
        channels = sum(s[0] for s in shape)
        y_hat = torch.zeros((1, channels, *shape[0][1:]), device=side_params.device)  # batch 1, as the reference (:139)
        self._run(y_hat, side_params, lambda k, params: self.latent_codec[f"y{k}"].decompress(
            strings[per_group * k: per_group * (k + 1)], shape[k], params))
        return {"y_hat": y_hat}

    def forward(self, y: Tensor, side_params: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path")


class HyperLatentCodec(nn.Module):
    """Hyper branch (compressai/latent_codecs/hyper.py:48-108): ``z = h_a(y)`` is table-coded, ``params = h_s(z_hat)``.
    ``entropy_bottleneck`` is anything with the reference's ``compress(z) -> [bytes]`` / ``decompress(strings, size)``
    contract: ``flashgmm_amd.EntropyBottleneckCoder`` (built from the tables of a trained ``EntropyBottleneck``), or the
    reference's own class."""

    def __init__(self, entropy_bottleneck=None, h_a: Optional[nn.Module] = None, h_s: Optional[nn.Module] = None,
                 quantizer: str = "noise", **kwargs: Any):
        super().__init__()
        assert entropy_bottleneck is not None
        self.entropy_bottleneck = entropy_bottleneck
        self.h_a = h_a or nn.Identity()
        self.h_s = h_s or nn.Identity()
        self.quantizer = quantizer

    def compress(self, y: Tensor) -> Dict[str, Any]:  # hyper.py:94-100
        z = self.h_a(y)
        shape = z.size()[-2:]
        eb = self.entropy_bottleneck
        if isinstance(eb, EntropyBottleneckCoder):
            # the reference decodes the strings it has just written to get z_hat (hyper.py:97-98); the table coder is lossless on
            # the symbols, so that is round(z - medians) + medians: one elementwise op where z lives, bit for bit the same
            z_strings, z_hat = eb.compress(z, return_dequantized=True)
        else:  # any other entropy bottleneck with the reference's contract
            z_strings = eb.compress(z)
            z_hat = eb.decompress(z_strings, shape)
        return {"strings": [z_strings], "shape": shape, "params": self.h_s(z_hat)}

    def decompress(self, strings: List[List[bytes]], shape: Tuple[int, int], **kwargs: Any) -> Dict[str, Any]:  # hyper.py:102-108
        (z_strings,) = strings
        z_hat = self.entropy_bottleneck.decompress(z_strings, shape)
        return {"params": self.h_s(z_hat)}

    def forward(self, y: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path")


class HyperpriorLatentCodec(nn.Module):
    """``latent_codec = {"y": ..., "hyper": HyperLatentCodec}`` (compressai/latent_codecs/hyperprior.py:46-139): the complete
    nested result of the GMM models, ``{"strings": [*y_strings, z_strings], "shape": {"y": ..., "hyper": ...}, "y_hat"}``
    (models/ckbd_gmm.py:109-123: ``y`` is a ``CheckerboardLatentCodec``; models/elic_gmm.py:197-227: a
    ``ChannelGroupsLatentCodec``)."""

    def __init__(self, latent_codec: Optional[Dict[str, nn.Module]] = None, **kwargs: Any):
        super().__init__()
        if latent_codec is None or "y" not in latent_codec or "hyper" not in latent_codec:
            raise ValueError('latent_codec must hold the "y" and the "hyper" codec')
        self.latent_codec = nn.ModuleDict(latent_codec)

    def __getitem__(self, key: str) -> nn.Module:
        return self.latent_codec[key]

    def compress(self, y: Tensor) -> Dict[str, Any]:  # hyperprior.py:120-128
        hyper_out = self.latent_codec["hyper"].compress(y)
        y_out = self.latent_codec["y"].compress(y, hyper_out["params"])
        [z_strings] = hyper_out["strings"]
        return {"strings": [*y_out["strings"], z_strings], "shape": {"y": y_out["shape"], "hyper": hyper_out["shape"]},
                "y_hat": y_out["y_hat"]}

    def decompress(self, strings: List[Any], shape: Dict[str, Any], **kwargs: Any) -> Dict[str, Any]:  # hyperprior.py:130-139
        *y_strings_, z_strings = strings
        hyper_out = self.latent_codec["hyper"].decompress([z_strings], shape["hyper"])
        y_out = self.latent_codec["y"].decompress(y_strings_, shape["y"], hyper_out["params"])
        return {"y_hat": y_out["y_hat"]}

    def forward(self, y: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path")

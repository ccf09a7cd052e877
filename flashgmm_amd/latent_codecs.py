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
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .entropy_models import GaussianMixtureConditional

__all__ = ["GaussianMixtureConditionalLatentCodec"]


class GaussianMixtureConditionalLatentCodec(nn.Module):
    def __init__(self, K: int = 4, gaussian_mixture_conditional: Optional[GaussianMixtureConditional] = None,
                 entropy_parameters: Optional[nn.Module] = None, quantizer: str = "noise",
                 chunks: Tuple[str, ...] = ("scales", "means", "weights"), mode=None, **kwargs: Any):
        super().__init__()
        if quantizer not in ("noise", "weighted_mean_ste"):
            raise ValueError(f"quantizer {quantizer} not supported")
        if tuple(chunks) != ("scales", "means", "weights"):
            raise ValueError("a Gaussian-mixture codec needs chunks = ('scales', 'means', 'weights')")
        self.K = K
        self.quantizer = quantizer
        self.gaussian_mixture_conditional = gaussian_mixture_conditional or GaussianMixtureConditional(K=K, mode=mode)
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
        return scales_hat, means_hat, self._reshape_gmm_weight(weights)

    def _recentre(self, means_hat: Tensor, weights: Tensor):
        """weighted_mean_ste: sum_k pi_k mu_k and the means relative to it (:139-144)"""
        B, KM, H, W = means_hat.shape
        me = means_hat.view(B, self.K, KM // self.K, H, W)
        we = weights.view(B, self.K, KM // self.K, H, W)
        weighted_sum = torch.sum(me * we, dim=1)
        return weighted_sum, (me - weighted_sum.unsqueeze(1)).reshape(B, KM, H, W)

    def compress(self, y: Tensor, ctx_params: Tensor) -> Dict[str, Any]:
        scales_hat, means_hat, weights = self._params(ctx_params)
        if self.quantizer == "noise":
            y_strings, y_hat = self.gaussian_mixture_conditional.compress(y, scales_hat, means_hat, weights)
        else:
            weighted_sum, means_rel = self._recentre(means_hat, weights)
            y_strings, y_hat = self.gaussian_mixture_conditional.compress(torch.round(y - weighted_sum), scales_hat,
                                                                          means_rel, weights)
        return {"strings": [y_strings], "shape": y.shape[2:4], "y_hat": y_hat}

    def decompress(self, strings: List[Any], shape: Tuple[int, int], ctx_params: Tensor, **kwargs: Any) -> Dict[str, Any]:
        (y_strings,) = strings
        scales_hat, means_hat, weights = self._params(ctx_params)
        if self.quantizer == "noise":
            y_hat = self.gaussian_mixture_conditional.decompress(*y_strings, scales_hat, means_hat, weights)
        else:
            weighted_sum, means_rel = self._recentre(means_hat, weights)
            y_hat = self.gaussian_mixture_conditional.decompress(*y_strings, scales_hat, means_rel, weights) + weighted_sum
        assert tuple(y_hat.shape[2:4]) == tuple(shape)
        return {"y_hat": y_hat}

    def forward(self, y: Tensor, ctx_params: Tensor):
        raise NotImplementedError("training-time likelihoods are outside the entropy-coding path; use the reference's "
                                  "GaussianMixtureConditionalLatentCodec.forward (pure torch)")

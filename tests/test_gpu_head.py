"""GPU: the parameter head's last layer on the matrix cores, fused with the encode-side CDF kernel (SURVEY.md section 8 f2;
flashgmm_amd/csrc/fgmm_head.hip).  Replaces compressai/models/ckbd_gmm.py:115-121 (the final nn.Conv2d(640, 3*K*N, 1) of
`entropy_parameters`) + latent_codecs/gaussian_mixture_conditional.py:183-202 (chunk, softmax over K).

Bars: the parameters are BIT FOR BIT the oracle's fmaf chain (one rounding per product, k ascending, from the bias - the order the
library fixes) and within 1e-5 (relative to sum |w x|) of torch.nn.functional.conv2d in fp32; the fused kernel's bitstreams are byte
for byte those of the un-fused path fed the head kernel's own parameter planes; decode(encode(y)) == round(y)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from flashgmm_amd import GaussianMixtureConditional, ParameterHead, _lib  # noqa: E402
from tests.synth import make_head  # noqa: E402

pytestmark = pytest.mark.gpu
MODES = ["polya", "as", "logistic"]


@pytest.mark.parametrize("M,c_in,h,w", [(192, 640, 32, 24), (16, 32, 4, 8), (24, 40, 5, 7), (8, 33, 3, 3), (40, 64, 16, 20)])
def test_head_parameters_equal_the_fmaf_chain_bit_for_bit(oracle, M, c_in, h, w):
    """v_mfma_f32_32x32x2_f32 == fmaf chain, k ascending, from the bias: every one of the 3*K*M*h*w outputs, ragged sizes included
    (channels not a multiple of 16, input channels not a multiple of 32, positions not a multiple of 4 or 256)"""
    conv, x, _ = make_head(11, M, c_in, h, w, N=2)
    head = ParameterHead(conv)
    got = torch.cat(head.params(x), 1).cpu().numpy()
    Wn, bn = conv.weight.detach().cpu().numpy().reshape(12 * M, c_in), conv.bias.detach().cpu().numpy()
    for i in range(x.shape[0]):
        want = oracle.head_params(Wn, bn, x[i].cpu().numpy().reshape(c_in, h * w)).reshape(12 * M, h, w)
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (i, np.abs(got[i] - want).max())


def test_head_parameters_within_1e5_of_torch_conv2d():
    """... and they are the convolution: within 1e-5 of sum |w x| + |b| of torch's fp32 conv2d (which fixes no summation order)"""
    M, c_in = 192, 640
    conv, x, _ = make_head(12, M, c_in, 32, 24, N=3)
    head = ParameterHead(conv)
    got = torch.cat(head.params(x), 1)
    with torch.no_grad():
        want = torch.nn.functional.conv2d(x, conv.weight, conv.bias)
        scale = torch.nn.functional.conv2d(x.abs(), conv.weight.abs(), conv.bias.abs())
    rel = ((got - want).abs() / scale).max().item()
    assert rel < 1e-5, rel
    # the three chunks are what the latent codec's chunk(3, 1) gives (gaussian_mixture_conditional.py:193-195)
    s, m, lg = head.params(x)
    assert s.shape == (3, 4 * M, 32, 24) and torch.equal(torch.cat([s, m, lg], 1), got)


def test_head_without_bias():
    conv, x, _ = make_head(13, 16, 64, 8, 8, N=1)
    conv.bias = None
    got = torch.cat(ParameterHead(conv).params(x), 1)
    with torch.no_grad():
        want = torch.nn.functional.conv2d(x, conv.weight)
    assert torch.allclose(got, want, rtol=0, atol=2e-5 * want.abs().max().item())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("shape", [(192, 640, 32, 24, 4, 20), (24, 40, 5, 7, 3, 5), (32, 64, 16, 16, 2, 0)])
def test_fused_head_bytes_equal_the_unfused_path(mode, shape):
    """one kernel (matrix product + table entries) against two (head_params planes -> symtab_kernel with FGMM_PARAMS_LOGITS): the same
    bytes, abs_max, zero bitmap and y_q for every item; and the streams decode from the head's planes to round(y)"""
    M, c_in, h, w, N, dead = shape
    conv, x, y = make_head(21, M, c_in, h, w, N, dead)
    head = ParameterHead(conv)
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    fused = gmc.compress_head_batch(y, x, head)
    s, m, lg = head.params(x)
    plain = gmc.compress_batch(y, s, m, lg, weights_are_logits=True)
    assert len(fused) == N
    for i in range(N):
        (bf, af, zf), qf = fused[i]
        (bp, ap, zp), qp = plain[i]
        assert bytes(bf) == bytes(bp) and af == ap and torch.equal(zf, zp) and torch.equal(qf, qp), i
        assert 0 < int(zf.sum()) <= M - dead and len(bf) > 8  # (the killed channels, and those whose energy rounds to zero anyway)
    out = gmc.decompress_batch(fused.strings, fused.abs_maxes, fused.zero_bitmaps, s, m, lg, weights_are_logits=True, stacked_output=True)
    assert torch.equal(out, fused.y_q) and torch.equal(fused.y_q[:, 0], torch.round(y))


def test_fused_head_with_checkpoints_and_segmented_tables():
    """the fused kernel writes the SAME table layout the host encoders expect: segmented tables (a worker per bitstream, tail first)
    and checkpointed streams"""
    M, c_in, h, w, N = 192, 640, 32, 24, 6
    conv, x, y = make_head(22, M, c_in, h, w, N, dead=7)
    head = ParameterHead(conv)
    saved = _lib.get_option(0, "enc_segs")
    try:
        ref = GaussianMixtureConditional(K=4, mode="polya").compress_head_batch(y, x, head)
        _lib.set_option(0, "enc_segs", 2)  # (forced: segmented whatever the size)
        seg = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=256).compress_head_batch(y, x, head)
    finally:
        _lib.set_option(0, "enc_segs", saved)
    for i in range(N):
        assert bytes(seg.strings[i]) == bytes(ref.strings[i]) and len(seg.strings[i].ckpt) > 0


def test_head_refuses_what_it_cannot_do():
    conv, x, y = make_head(23, 16, 32, 4, 8, N=1)
    head = ParameterHead(conv)
    gmc = GaussianMixtureConditional(K=4)
    with pytest.raises(RuntimeError):
        gmc.compress_head_batch(y[:, :8], x, head)  # M mismatch
    with pytest.raises(RuntimeError):
        head.params(x[:, :16])  # c_in mismatch
    with pytest.raises(RuntimeError):
        ParameterHead(torch.nn.Conv2d(32, 12 * 16, 3).to("cuda:0"))  # not 1x1


def test_fused_head_with_non_finite_features_equals_the_unfused_path():
    """features with NaN / +-inf at a few positions give NaN / inf parameters there: the fused epilogue and symtab_kernel share one
    sym_entry (its out-of-line IEEE evaluation included), so the bytes stay equal - and input channels past c_in (a last K tile that
    is only partly real: c_in = 40) are masked, not multiplied"""
    M, c_in, h, w, N = 24, 40, 8, 12, 2
    conv, x, y = make_head(31, M, c_in, h, w, N)
    x[0, 3, 2, 5] = float("nan")
    x[0, 39, 0, 0] = float("inf")
    x[1, 0, 7, 11] = float("-inf")
    head = ParameterHead(conv)
    for mode in MODES:
        gmc = GaussianMixtureConditional(K=4, mode=mode)
        fused = gmc.compress_head_batch(y, x, head)
        s, m, lg = head.params(x)
        assert not torch.isfinite(s).all()
        plain = gmc.compress_batch(y, s, m, lg, weights_are_logits=True)
        assert [bytes(b) for b in fused.strings] == [bytes(b) for b in plain.strings] and fused.abs_maxes == plain.abs_maxes
        out = gmc.decompress_batch(fused.strings, fused.abs_maxes, fused.zero_bitmaps, s, m, lg, weights_are_logits=True, stacked_output=True)
        assert torch.equal(out, fused.y_q)


def test_head_c_abi_with_items_of_different_sizes():
    """fgmm_gmc_compress_head_batch / fgmm_head_params_batch straight through the C ABI (ctypes) with RAGGED items - 96, 300 and 0
    positions in one call (the stacked Python form has one size): every item equals its own single-item call"""
    import ctypes as C

    M, c_in = 16, 64
    conv, _, _ = make_head(41, M, c_in, 2, 2, 1)
    head = ParameterHead(conv)
    L, ctx = _lib.lib(), _lib.ctx(0)
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    rng = np.random.default_rng(5)
    sizes = [(8, 12), (15, 20), (0, 7)]
    xs = [torch.from_numpy(rng.standard_normal((1, c_in, hh, ww)).astype(np.float32)).cuda() for hh, ww in sizes]
    ys = [torch.from_numpy((rng.standard_normal((1, M, hh, ww)) * 3).astype(np.float32)).cuda() for hh, ww in sizes]
    n = len(sizes)
    items = (_lib.fgmm_item * n)()
    yq = [torch.empty_like(t) for t in ys]
    zb = [torch.empty(M, dtype=torch.int64) for _ in sizes]
    for i, (hh, ww) in enumerate(sizes):
        items[i].y, items[i].M, items[i].K, items[i].hw = ys[i].data_ptr(), M, 4, hh * ww
        items[i].yq_out, items[i].zero_bitmap = yq[i].data_ptr(), zb[i].data_ptr()
    xp = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    _lib.check(L.fgmm_gmc_compress_head_batch(ctx, None, items, xp, n, head._h, 0, 1), "fgmm_gmc_compress_head_batch")
    outs = [torch.empty((1, 12 * M, hh, ww), dtype=torch.float32, device="cuda") for hh, ww in sizes]
    op = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    hw = (C.c_int64 * n)(*[hh * ww for hh, ww in sizes])
    _lib.check(L.fgmm_head_params_batch(ctx, None, head._h, xp, op, hw, n), "fgmm_head_params_batch")
    for i, (hh, ww) in enumerate(sizes):
        got = _lib.take_bytes(items[i].bytes, items[i].bytes_len)
        if hh * ww == 0:
            assert got == bytes.fromhex("0000008000000000")  # the flushed state of an empty stream
            continue
        one = gmc.compress_head_batch(ys[i], xs[i], head)
        assert got == bytes(one.strings[0]) and items[i].abs_max == one.abs_maxes[0] and torch.equal(yq[i], one.y_q[0])
        assert torch.equal(outs[i], torch.cat(head.params(xs[i]), 1))


def _ckbd_codecs(M, seed, mode="polya", arithmetic="f32"):
    """two CheckerboardLatentCodecs over the same networks: one with fuse_head, one whose last layer is a module that calls the head's
    un-fused kernel (the same arithmetic, parameter tensors through HBM, fuse_softmax in the GMM codec)"""
    from flashgmm_amd.latent_codecs import CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec

    torch.manual_seed(seed)
    dev = "cuda:0"
    ctx_net = torch.nn.Conv2d(M, 2 * M, 5, padding=2).to(dev)
    body = [torch.nn.Conv2d(4 * M, 96, 1), torch.nn.LeakyReLU()]
    last = torch.nn.Conv2d(96, 12 * M, 1)
    with torch.no_grad():
        last.bias[: 4 * M] += 1.0  # sigma mostly positive
    ep = torch.nn.Sequential(*body, last).to(dev)

    class HeadModule(torch.nn.Module):
        def __init__(self, head):
            super().__init__()
            self.head = head

        def forward(self, x):
            return torch.cat(self.head.params(x), 1)

    fused = CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4, mode=mode)}, entropy_parameters=ep,
                                    context_prediction=ctx_net, fuse_head=True if arithmetic == "f32" else arithmetic)
    plain = CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4, mode=mode, fuse_softmax=True)},
                                    entropy_parameters=torch.nn.Sequential(*ep[:-1], HeadModule(ParameterHead(ep[-1], arithmetic=arithmetic))),
                                    context_prediction=ctx_net)
    return fused, plain


@pytest.mark.parametrize("mode,arithmetic", [("polya", "f32"), ("logistic", "f32"), ("polya", "bf16x6")])
def test_checkerboard_codec_with_the_fused_head(mode, arithmetic):
    """CheckerboardLatentCodec(fuse_head=True) - the mirror of compressai/latent_codecs/checkerboard.py:275-330 with the last layer of its
    entropy_parameters inside the library: the same strings, shape and y_hat as the codec that runs the head's un-fused kernel as its
    last layer, both halves in one encode call; decompress (stage by stage) gives y_hat back, image by image and stage-major"""
    M, h, w = 32, 16, 24
    fused, plain = _ckbd_codecs(M, 3, mode, arithmetic)
    rng = np.random.default_rng(9)
    with torch.no_grad():
        outs = []
        for img in range(2):
            y = torch.from_numpy((rng.standard_normal((1, M, h, w)) * 4).astype(np.float32)).cuda()
            side = torch.from_numpy(rng.standard_normal((1, 2 * M, h, w)).astype(np.float32)).cuda()
            a, b = fused.compress(y, side), plain.compress(y, side)
            assert len(a["strings"]) == 2 and a["shape"] == b["shape"] and torch.equal(a["y_hat"], b["y_hat"]) and torch.equal(a["y_hat"], torch.round(y))
            for (sa, ama, za), (sb, amb, zb_) in zip(a["strings"], b["strings"]):
                assert bytes(sa) == bytes(sb) and ama == amb and torch.equal(za.cpu(), zb_.cpu())
            assert torch.equal(fused.decompress(a["strings"], a["shape"], side)["y_hat"], a["y_hat"])
            assert torch.equal(plain.decompress(a["strings"], a["shape"], side)["y_hat"], a["y_hat"])
            outs.append((a, side))
        many = fused.decompress_many([o[0]["strings"] for o in outs], outs[0][0]["shape"], [o[1] for o in outs])
        for m_, (a, _) in zip(many, outs):
            assert torch.equal(m_["y_hat"], a["y_hat"])
        # new weights in the last layer: the head is packed again (not the stale one)
        fused.entropy_parameters[-1].weight.mul_(1.25)
        c = fused.compress(y, side)
        assert any(bytes(x[0]) != bytes(z[0]) for x, z in zip(c["strings"], a["strings"]))
        assert torch.equal(fused.decompress(c["strings"], c["shape"], side)["y_hat"], c["y_hat"])


def test_fuse_head_is_refused_where_it_cannot_apply():
    from flashgmm_amd.latent_codecs import CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec

    with pytest.raises(ValueError):
        CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4)}, entropy_parameters=torch.nn.Conv2d(8, 96, 3), fuse_head=True)
    codec = CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4, quantizer="weighted_mean_ste")},
                                    entropy_parameters=torch.nn.Conv2d(64, 12 * 8, 1).cuda(), context_prediction=torch.nn.Conv2d(8, 48, 1).cuda(), fuse_head=True)
    with pytest.raises(RuntimeError):
        codec.compress(torch.zeros((1, 8, 4, 8), device="cuda"), torch.zeros((1, 16, 4, 8), device="cuda"))


@pytest.mark.parametrize("M,c_in,h,w", [(192, 640, 32, 24), (24, 40, 5, 7), (16, 64, 16, 16)])
def test_bf16x6_head_is_accurate_deterministic_and_self_consistent(M, c_in, h, w):
    """ParameterHead(arithmetic="bf16x6") - FGMM_HEAD_BF16X6, fgmm_head16.hip: three bfloat16 parts per operand, six part products on the
    BF16 matrix cores.  Within 1e-5 of sum |w x| + |b| of the exact result (in fact ~1e-6: asserted at 2e-6), the same bits on every
    launch, fused bytes == un-fused bytes fed its own planes, decode(encode(y)) == round(y)."""
    conv, x, y = make_head(51, M, c_in, h, w, N=3, dead=2)
    head = ParameterHead(conv, arithmetic="bf16x6")
    got = torch.cat(head.params(x), 1)
    with torch.no_grad():
        xd, wd = x.double(), conv.weight.double()
        want = torch.nn.functional.conv2d(xd, wd, conv.bias.double())
        scale = torch.nn.functional.conv2d(xd.abs(), wd.abs(), conv.bias.double().abs())
    rel = ((got.double() - want).abs() / scale).max().item()
    assert rel < 2e-6, rel
    assert torch.equal(torch.cat(head.params(x), 1), got)  # deterministic
    exact = torch.cat(ParameterHead(conv).params(x), 1)
    assert ((exact.double() - want).abs() / scale).max().item() < 2e-6 and not torch.equal(exact, got)  # (another arithmetic: other last bits)
    for mode in MODES:
        gmc = GaussianMixtureConditional(K=4, mode=mode)
        fused = gmc.compress_head_batch(y, x, head)
        s, m, lg = head.params(x)
        plain = gmc.compress_batch(y, s, m, lg, weights_are_logits=True)
        assert [bytes(b) for b in fused.strings] == [bytes(b) for b in plain.strings] and fused.abs_maxes == plain.abs_maxes
        out = gmc.decompress_batch(fused.strings, fused.abs_maxes, fused.zero_bitmaps, s, m, lg, weights_are_logits=True, stacked_output=True)
        assert torch.equal(out, fused.y_q) and torch.equal(fused.y_q[:, 0], torch.round(y))

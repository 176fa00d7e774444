"""Size-independent properties of the HIP operators at the FULL sizes of BASELINE.json's configs (#3: UNet batch 16 at 64x64 latents,
#5: 128x128 latents), where the CPU oracle would take minutes: exact homogeneity of the contractions under powers of two, additivity,
key-permutation invariance and constant preservation of attention, the statistics GroupNorm must produce and its shift invariance, the
identities of the guidance mix.  No reference needed: each property follows from the operator's definition (cited per test)."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from lightdiffusion_amd import ops as o
    from lightdiffusion_amd._lib import lib
    lib()
    return o


def r16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().to(DEV)


def equal_where_normal(a, b, ref):
    """bit equality of a and b wherever `ref` (and so a, b) is a NORMAL fp16 number with headroom: a power-of-two scale commutes with every
    rounding of the computation except the final one into the fp16 subnormal range (|y| < 2^-14), which a few of 10^7 outputs reach"""
    m = ref.abs() >= 2.0 ** -11
    assert m.float().mean().item() > 0.995
    return torch.equal(a[m], b[m])


# every contraction of the batch-8 forward's big-tile kernels: (M, N, K) -> row-panel (K = 320), 128 x 160, 256 x 320 + split-K reduce
LINEAR_SHAPES = [(65536, 320, 320), (65536, 640, 320), (16384, 640, 640), (4096, 1280, 1280), (16384, 640, 2560), (1024, 1280, 5120)]


@pytest.mark.parametrize("M,N,K", LINEAR_SHAPES)
def test_linear_homogeneous_and_additive(ops, M, N, K):
    """Linear without bias is linear (LD.py:2361-2371: F.linear): f(2x) == 2 f(x) and f(4x) == 4 f(x) BIT FOR BIT (a power of two
    commutes with every fp16 / fp32 rounding as long as nothing overflows or goes subnormal: the kernels add in a fixed order), and
    f(x + y) ~= f(x) + f(y) to fp16 rounding."""
    x, y, w = r16((M, K), 201), r16((M, K), 202), r16((N, K), 203, 1 / math.sqrt(K))
    fx = ops.linear(x, w)
    assert equal_where_normal(ops.linear(x * 2, w), fx * 2, fx)
    assert equal_where_normal(ops.linear(x * 4, w), fx * 4, fx)
    fy = ops.linear(y, w)
    fxy = ops.linear((x.float() + y.float()).half(), w)
    assert rel_l2(fxy.float().cpu(), (fx.float() + fy.float()).cpu()) < 2e-3


@pytest.mark.parametrize("n,h,cin,cout", [(16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 8, 1280, 1280), (8, 128, 320, 320)])
def test_conv3x3_homogeneous_and_shift_equivariant(ops, n, h, cin, cout):
    """Conv2d without bias (LD.py:2373-2389: F.conv2d, padding 1): exact homogeneity under powers of two on the halo-tile / split-K paths, and
    translation equivariance — an input shifted by one pixel (zero fill) gives the output shifted by one pixel away from the borders."""
    g = torch.Generator().manual_seed(211)
    x = (torch.randn(n, h, h, cin, generator=g)).half().to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(9 * cin)).half().to(DEV)
    wp = ops.repack_conv_weight(w)
    fx = ops.conv2d(x, wp, None)
    assert equal_where_normal(ops.conv2d(x * 2, wp, None), fx * 2, fx)
    xs = torch.zeros_like(x)
    xs[:, :, 1:] = x[:, :, :-1]                                   # shift right along W
    fs = ops.conv2d(xs, wp, None)
    assert torch.equal(fs[:, :, 2:-1], fx[:, :, 1:-2])           # same taps, same order: bit-equal away from the border columns


@pytest.mark.parametrize("b,L,heads,d", [(16, 4096, 8, 40), (16, 1024, 8, 80), (8, 16384, 8, 40)])
def test_attention_permutation_and_constant(ops, b, L, heads, d):
    """softmax(QK^T/sqrt(d)) V (LD.py:3966-3978): permuting the keys together with the values changes nothing but the summation order;
    with V constant along the keys the output is that constant (the weights of a row sum to one)."""
    if L >= 16384:
        b = 2                                                     # (same kernel instantiation, a quarter of the work)
    c = heads * d
    q, k, v = r16((b, L, c), 221), r16((b, L, c), 222), r16((b, L, c), 223)
    o = ops.attention(q, k, v, heads)
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(224)).to(DEV)
    op = ops.attention(q, k[:, perm].contiguous(), v[:, perm].contiguous(), heads)
    assert rel_l2(op.float().cpu(), o.float().cpu()) < 2e-3
    vc = r16((b, 1, c), 225).expand(b, L, c).contiguous()
    oc = ops.attention(q, k, vc, heads)
    assert (oc.float() - vc.float()).abs().max().item() < 4e-3


@pytest.mark.parametrize("b,L", [(8, 4096), (2, 16384), (1, 3969)])
def test_vae_attention_permutation_and_constant(ops, b, L):
    """The VAE's single head of 512 channels (AttnBlock, LD.py:3591-3642; flash_attn512_kernel) at the sizes of the 512^2 and 1024^2 decodes and
    at a ragged 63 x 63 latent: the same size-independent properties as the UNet's attention, through the fused [q | k | v] form the
    executor hands over."""
    c = 512
    q, k, v = r16((b, L, c), 241), r16((b, L, c), 242), r16((b, L, c), 243)
    o = ops.attention_qkv(torch.cat([q, k, v], -1).contiguous(), 1)
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(244)).to(DEV)
    op = ops.attention_rowv(q, k[:, perm].contiguous(), v[:, perm].contiguous(), 1)
    assert rel_l2(op.float().cpu(), o.float().cpu()) < 2e-3
    vc = r16((b, 1, c), 245).expand(b, L, c).contiguous()
    oc = ops.attention_rowv(q, k, vc, 1)
    assert (oc.float() - vc.float()).abs().max().item() < 4e-3
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,hw,c", [(16, 4096, 320), (16, 1024, 1920), (8, 16384, 640)])
def test_groupnorm_statistics_and_shift_invariance(ops, n, hw, c):
    """GroupNorm(32, eps 1e-5, affine) (LD.py:2391-2403): with gamma = 1, beta = 0 every (image, group) of the output has mean 0 and
    variance 1 (up to eps and fp16 rounding), and adding a per-image constant to the input changes nothing."""
    x = r16((n, hw, c), 231, 2.0) + 0.5
    ones, zeros = torch.ones(c, dtype=torch.float16, device=DEV), torch.zeros(c, dtype=torch.float16, device=DEV)
    y = ops.group_norm(x, ones, zeros, 1e-5)
    yg = y.float().view(n, hw, 32, c // 32)
    assert yg.mean(dim=(1, 3)).abs().max().item() < 2e-3
    assert (yg.var(dim=(1, 3), unbiased=False) - 1).abs().max().item() < 5e-3
    shift = torch.linspace(-1.5, 1.5, n, device=DEV).half().view(n, 1, 1)
    y2 = ops.group_norm((x.float() + shift.float()).half(), ones, zeros, 1e-5)
    assert rel_l2(y2.float().cpu(), y.float().cpu()) < 2e-3


def test_cfg_combine_identities(ops):
    """cfg_function (LD.py:2594-2606): uncond + (cond - uncond) * cfg — cfg 1 returns cond, cfg 0 returns uncond, and the result is affine in cfg."""
    den2 = torch.randn(16, 4, 64, 64, generator=torch.Generator().manual_seed(241)).to(DEV)
    u, c = den2[:8], den2[8:]
    assert torch.allclose(ops.cfg_combine(den2, 1.0), c, atol=1e-6)
    assert torch.equal(ops.cfg_combine(den2, 0.0), u)
    mid = ops.cfg_combine(den2, 7.5)
    assert torch.allclose(mid, 0.5 * (ops.cfg_combine(den2, 7.0) + ops.cfg_combine(den2, 8.0)), atol=1e-5)


def test_full_size_unet_is_deterministic_and_graph_equals_eager():
    """No float atomics anywhere and every reduction in a fixed order: the batch-16 SD1.5 forward (config #3's UNet call) gives the same bits
    twice, and the hipGraph-replayed CFG step the product loop runs gives the bits of the eager one."""
    from lightdiffusion_amd import weights as W
    from lightdiffusion_amd.unet import synthetic_unet
    B = 8
    u = synthetic_unet(W.sd15_unet_config(), max_batch=2 * B, max_hw=(64, 64))
    g = torch.Generator().manual_seed(251)
    ctx = torch.randn(2 * B, 77, 768, generator=g)
    u.set_context(ctx)
    x = (torch.randn(2 * B, 4, 64, 64, generator=g) * 4.0).to(DEV)
    sig = torch.full((2 * B,), 4.0, device=DEV)
    a = u.forward(x, sig).clone()
    b = u.forward(x, sig).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    xb = x[:B].contiguous()
    d1 = u.cfg_denoise(xb, sig[:B].contiguous(), ctx.to(DEV), 7.5)      # captures, then replays
    d2 = u.cfg_denoise(xb, sig[:B].contiguous(), ctx.to(DEV), 7.5)
    assert torch.equal(d1, d2)
    d3 = u.cfg_denoise(xb, sig[:B].contiguous(), ctx.to(DEV), 7.5, use_graph=False)
    assert torch.equal(d1, d3)

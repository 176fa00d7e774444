"""GPU parity at the SIZES of BASELINE.json's configs #3 and #5, and of the reference block goldens through the operator seam.

Why this file exists: tile height, split-K, the skinny-GEMM rule and the attention workgroup shape are chosen from the GEMM M
dimension / tile counts, so UNet batch 16 at 64x64 (config #3), 128x128 latents (config #5, L = 16384) and the 512^2 / 1024^2
VAE decodes execute kernel instantiations that the N=2, 64x64 goldens never touch.  Goldens: the reference's own classes, fp32
CPU (oracle/make_golden.py `configs`, `blocks`, `bislerp`).  Tolerances as elsewhere: rel-L2 <= 5e-3 per UNet call, <= 3e-3 per
block, <= 2/255 per VAE pixel (fp16 storage, fp32 accumulate)."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import load_golden, rel_l2
from lightdiffusion_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
UNET_TOL = 5e-3


@pytest.fixture(scope="module")
def sd15_unet():
    from lightdiffusion_amd.unet import synthetic_unet
    u = synthetic_unet(W.sd15_unet_config(), max_batch=16, max_hw=(64, 64))
    yield u
    del u
    torch.cuda.empty_cache()


def _stack(g, order):
    idx = torch.tensor(order)
    return g["x"][idx].contiguous(), g["sigma"][idx].contiguous(), g["ctx"][idx].contiguous()


def test_config3_unet_batch16_64x64(sd15_unet):
    """Config #3 (batch 8 => UNet batch 16): samples are independent (LD.py:2507-2547), so each of the 16 rows must reproduce
    the reference golden of the sample it was built from — through the B=8 kernel instantiations (128-row tiles, no split-K,
    8-wave attention)."""
    g = load_golden("unet_sd15_64x64")
    order = [0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 0]
    x, s, ctx = _stack(g, order)
    sd15_unet.set_context(ctx)
    den = sd15_unet.forward(x.to(DEV), s.to(DEV)).cpu()
    eps = sd15_unet.forward(x.to(DEV), s.to(DEV), eps_only=True).cpu()
    assert torch.isfinite(den).all()
    for row, src in enumerate(order):
        assert rel_l2(den[row], g["denoised"][src]) < UNET_TOL, row
        assert rel_l2(eps[row], g["eps"][src]) < UNET_TOL, row
    assert abs(sd15_unet.last_flops / 16 - 803.3e9) / 803.3e9 < 0.03
    # rows built from the same sample agree with each other far more tightly than with the fp32 golden
    assert rel_l2(den[0], den[3]) < 1e-3 and rel_l2(den[1], den[2]) < 1e-3


def test_config3_through_the_wrapper_hook_at_batch16(sd15_unet):
    """The reference's own seam at config #3's size (UNet batch 16, 64 x 64): `MI355XUNet.__call__` replays captured hipGraphs.  Sixteen
    different samples under cond_or_uncond == [1, 0]: the speculative CFG-pair replay is found wrong ON THE DEVICE (halves differ) and the
    call itself replays the plain graph — every row must reproduce the reference golden of its sample.  Then the shape calc_cond_batch really
    sends (one latent batch twice): the pair graph, bit for bit the eager `ld_unet_forward_pair`, and its cond rows against the golden."""
    g = load_golden("unet_sd15_64x64")
    order = [0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 0]
    x, s, ctx = _stack(g, order)
    mk = lambda xx, ss, cc: {"input": xx.to(DEV), "timestep": ss.to(DEV), "c": {"c_crossattn": cc.to(DEV), "transformer_options": {"cond_or_uncond": [1, 0]}},
                             "cond_or_uncond": [1, 0]}
    sd15_unet._hook.clear()
    den = sd15_unet(None, mk(x, s, ctx)).cpu()
    run = sd15_unet._hook[(16, 64, 64)]
    assert run.halves_differed and run.plain.graph is not None
    for row, src in enumerate(order):
        assert rel_l2(den[row], g["denoised"][src]) < UNET_TOL, row
    # what the reference sends: cat([x_in, x_in]) against cat([uncond, cond]); row i of the cond half = golden sample order[i] when its
    # context row is that sample's
    sd15_unet._hook.clear()
    x8, s8 = x[:8], s[:8]
    ctx2 = torch.cat([ctx[8:], ctx[:8]])                      # uncond half: some other context; cond half: the rows' own contexts
    out = sd15_unet(None, mk(torch.cat([x8, x8]), torch.cat([s8, s8]), ctx2))
    run = sd15_unet._hook[(16, 64, 64)]
    assert run.pair.graph is not None and not run.halves_differed and run.plain.graph is None
    assert torch.equal(out, sd15_unet.forward_pair(x8.to(DEV).contiguous(), s8.to(DEV).contiguous()))
    for row in range(8):
        assert rel_l2(out[8 + row].cpu(), g["denoised"][order[row]]) < UNET_TOL, row
    sd15_unet._hook.clear()


@pytest.mark.parametrize("n", [2, 8])
def test_config5_unet_128x128(sd15_unet, n):
    """Config #5 (hires-fix, batch 4 => UNet batch 8 at 128x128 latents, self-attention over L = 16384 tokens) against the
    reference's UNet at that size; n = 2 is the golden's own batch, n = 8 the config's."""
    g = load_golden("unet_sd15_128x128")
    order = [0, 1] if n == 2 else [0, 1, 1, 0, 1, 0, 0, 1]
    x, s, ctx = _stack(g, order)
    sd15_unet._ensure(n, 128, 128, 77)  # grows the 64x64 plan of the fixture to 128x128 (what the node surface does lazily)
    sd15_unet.set_context(ctx)
    den = sd15_unet.forward(x.to(DEV), s.to(DEV)).cpu()
    eps = sd15_unet.forward(x.to(DEV), s.to(DEV), eps_only=True).cpu()
    assert torch.isfinite(den).all()
    for row, src in enumerate(order):
        assert rel_l2(den[row], g["denoised"][src]) < UNET_TOL, row
        assert rel_l2(eps[row], g["eps"][src]) < UNET_TOL, row
    assert abs(sd15_unet.last_flops / n - 4674e9) / 4674e9 < 0.03      # 4 674 GFLOP per eval at 128x128


def test_config5_forward_pair_128x128_batch4(sd15_unet):
    """The route bench.py times for config #5 (LD.py:10585-10603 through calc_cond_batch, LD.py:2515-2547): ld_unet_forward_pair at nb = 4 on 128x128
    latents — conv_in, the first ResBlock and the first transformer's self-attention (L = 16384) evaluated once on 4 samples, the rest on 8.
    The golden's two samples carry different x, so pair i runs x_{o_i} twice against contexts [ctx_{1-o_i}, ctx_{o_i}]: the cond half must be the
    reference's denoised row o_i; with the golden's own context in BOTH halves the uncond half must be it too."""
    g = load_golden("unet_sd15_128x128")
    order = [0, 1, 1, 0]
    x = torch.stack([g["x"][i] for i in order]).to(DEV).contiguous()
    s = torch.stack([g["sigma"][i] for i in order]).to(DEV).contiguous()
    sd15_unet._ensure(8, 128, 128, 77)
    sd15_unet.set_context(torch.stack([g["ctx"][1 - i] for i in order] + [g["ctx"][i] for i in order]))
    den = sd15_unet.forward_pair(x, s).cpu()
    assert den.shape == (8, 4, 128, 128) and torch.isfinite(den).all()
    for row, src in enumerate(order):
        assert rel_l2(den[4 + row], g["denoised"][src]) < UNET_TOL, row
    sd15_unet.set_context(torch.stack([g["ctx"][i] for i in order] * 2))
    den2 = sd15_unet.forward_pair(x, s).cpu()
    for row, src in enumerate(order):
        assert rel_l2(den2[row], g["denoised"][src]) < UNET_TOL and rel_l2(den2[4 + row], g["denoised"][src]) < UNET_TOL, row
    # against the plain forward on the duplicated batch: two fp16 evaluations whose tile / split choices follow the row count
    full = sd15_unet.forward(torch.cat([x, x]).contiguous(), torch.cat([s, s]).contiguous()).cpu()
    assert rel_l2(den2, full) < 2e-3


@pytest.mark.parametrize("hw,sub", [(64, 4), (128, 8)])
def test_vae_decode_512_and_1024(hw, sub):
    """SD1.5 VAE decoder at 64x64 / 128x128 latents (512^2 / 1024^2 images: the [256,512,512] / [128,1024,1024] tensors of
    configs #2 and #5): subsampled image, a full-resolution centre crop and the moments against the reference's Decoder."""
    from lightdiffusion_amd.unet import synthetic_vae
    g = load_golden(f"vae_sd15_{hw}x{hw}")
    v = synthetic_vae(W.sd15_vae_config(), max_batch=1, max_hw=(hw, hw))
    img = v.decode(g["z"])
    assert img.shape == (1, 8 * hw, 8 * hw, 3)
    assert float((img[:, ::sub, ::sub] - g["img_sub"]).abs().max()) < 2.0 / 255.0
    c0 = int(g["crop_at"][0])
    assert float((img[:, c0:c0 + 96, c0:c0 + 96] - g["crop"]).abs().max()) < 2.0 / 255.0
    assert abs(float(img.mean() - g["mean"])) < 1e-3 and abs(float(img.std() - g["std"])) < 1e-3
    if hw == 64:                         # batch 2 at 512^2 (bigger M: other tile choices) must reproduce the batch-1 image
        img2 = v.decode(torch.cat([g["z"], g["z"]]))
        assert float((img2[0] - img[0]).abs().max()) < 1.0 / 255.0 and float((img2[1] - img[0]).abs().max()) < 1.0 / 255.0
    del v
    torch.cuda.empty_cache()


def test_vae_decode_non_square_bands_against_oracle():
    """SD1.5 VAE decoder on a NON-square latent (48 x 32 -> 384 x 256 pixels), batch 2: rows of 32 / 64 pixels (too few halo tiles at this size: implicit-GEMM kernel + split-K), 128 pixels
    (halo tiles of whole rows), 256 pixels (two 128-pixel column bands per row, 4-row x 128-column tiles for N = 128 with the fused GroupNorm, the upsampling loader
    into a banded image) and the MFMA output convolution with exactly 192 tiles per image pair — against the oracle on the host."""
    from lightdiffusion_amd.unet import synthetic_vae
    from oracle import sd15_ref as O
    cfg = W.sd15_vae_config()
    v = synthetic_vae(cfg, max_batch=2, max_hw=(48, 32))
    z = torch.randn(2, 4, 48, 32, generator=torch.Generator().manual_seed(23)) * (0.2 / 0.18215)
    img = v.decode(z)
    assert img.shape == (2, 384, 256, 3)
    kinds = {r[4] for r in v.profile_decode(z)}
    # (round 5: the one-tile-wide stages — N = 256 at 256-pixel rows, N = 128 at 4-row x 128-column tiles — apply their GroupNorm inside the halo loader)
    assert {"conv6_kernel<W128,halo+groupnorm,128x512>", "conv6_kernel<W128,halo,256,up>", "conv6_kernel<W128,halo+groupnorm,256>",
            "conv6_kernel<W128,halo,32x512>"} <= kinds, kinds
    ref = O.vae_decode(W.synth_state_dict(W.vae_decoder_param_shapes(cfg)), cfg, z[:1])
    assert float((img[:1] - ref).abs().max()) < 2.0 / 255.0
    img1 = v.decode(z[1:])                                     # batch 1: other tile counts (the output conv falls back below 192 tiles)
    assert float((img1 - img[1:]).abs().max()) < 1.0 / 255.0
    del v
    torch.cuda.empty_cache()


def test_vae_latents_not_multiple_of_8():
    """5x7 and 3x3 latents: the mid-block attention's key axis (h*w = 35 / 9) is not a multiple of 8 (padded, masked softmax)."""
    from lightdiffusion_amd.unet import synthetic_vae
    from oracle import sd15_ref as O
    cfg = W.tiny_vae_config()
    sdv = W.synth_state_dict(W.vae_decoder_param_shapes(cfg))
    v = synthetic_vae(cfg, max_batch=2, max_hw=(8, 8))
    g = torch.Generator().manual_seed(19)
    for shape in ((1, 4, 5, 7), (2, 4, 3, 3), (1, 4, 7, 9)):
        z = torch.randn(shape, generator=g)
        img = v.decode(z)
        ref = O.vae_decode(sdv, cfg, z)
        assert img.shape == ref.shape and float((img - ref).abs().max()) < 2.0 / 255.0


# ------------------------------------------------------------------ reference block goldens through the operator seam
def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _tb(ops, P, x, ctx, heads):
    """BasicTransformerBlock._forward (LD.py:4117-4162) from ld_op_* calls.  x [b, L, C] fp16, ctx [b, T, cd] fp16."""
    c = x.shape[-1]
    cd = ctx.shape[-1]
    ln = lambda t, n: ops.layer_norm(t, P(n + ".weight", (c,)), P(n + ".bias", (c,)), 1e-5)
    n1 = ln(x, "norm1")
    a = ops.attention(ops.linear(n1, P("attn1.to_q.weight", (c, c))), ops.linear(n1, P("attn1.to_k.weight", (c, c))),
                      ops.linear(n1, P("attn1.to_v.weight", (c, c))), heads)
    x = ops.linear(a, P("attn1.to_out.0.weight", (c, c)), P("attn1.to_out.0.bias", (c,)), residual=x)
    n2 = ln(x, "norm2")
    a = ops.attention(ops.linear(n2, P("attn2.to_q.weight", (c, c))), ops.linear(ctx, P("attn2.to_k.weight", (c, cd))),
                      ops.linear(ctx, P("attn2.to_v.weight", (c, cd))), heads)
    x = ops.linear(a, P("attn2.to_out.0.weight", (c, c)), P("attn2.to_out.0.bias", (c,)), residual=x)
    n3 = ln(x, "norm3")
    h = ops.linear(n3, P("ff.net.0.proj.weight", (8 * c, c)), P("ff.net.0.proj.bias", (8 * c,)), act="geglu")
    return ops.linear(h, P("ff.net.2.weight", (c, 4 * c)), P("ff.net.2.bias", (c,)), residual=x)


@pytest.mark.parametrize("tag,c,heads", [("h8d8", 64, 8), ("h2d40", 80, 2)])
def test_block_transformer_golden(tag, c, heads):
    from lightdiffusion_amd import ops
    g = load_golden("block_tb_" + tag)
    P = lambda k, s: W.synth_tensor(f"blk.tb.{tag}.{k}", s).half().to(DEV)
    y = _tb(ops, P, g["x"].half().to(DEV), g["ctx"].half().to(DEV), heads)
    assert rel_l2(y.float().cpu(), g["y"]) < 3e-3


def test_block_spatial_transformer_golden():
    """SpatialTransformer.forward (LD.py:4239-4262): GroupNorm(1e-6) -> 1x1 conv -> tokens -> block -> 1x1 conv -> + x."""
    from lightdiffusion_amd import ops
    g = load_golden("block_st")
    P = lambda k, s: W.synth_tensor(f"blk.st.{k}", s).half().to(DEV)
    x = nhwc(g["x"].half()).to(DEV)                       # [2, 8, 6, 64]
    n, h, w, c = x.shape
    t = ops.group_norm(x, P("norm.weight", (c,)), P("norm.bias", (c,)), 1e-6)
    t = ops.conv2d(t, ops.repack_conv_weight(P("proj_in.weight", (c, c, 1, 1))), P("proj_in.bias", (c,)), 1)
    Pb = lambda k, s: P("transformer_blocks.0." + k, s)
    t = _tb(ops, Pb, t.view(n, h * w, c), g["ctx"].half().to(DEV), 8).view(n, h, w, c)
    y = ops.conv2d(t, ops.repack_conv_weight(P("proj_out.weight", (c, c, 1, 1))), P("proj_out.bias", (c,)), 1, residual=x)
    assert rel_l2(nchw(y.float().cpu()), g["y"]) < 3e-3


def test_block_vae_res_and_attn_goldens():
    """VAE ResnetBlock (LD.py:3531-3576, with nin_shortcut) and AttnBlock (LD.py:3605-3642) of the reference."""
    from lightdiffusion_amd import ops
    g = load_golden("block_vae_res")
    P = lambda k, s: W.synth_tensor(f"blk.vres.{k}", s).half().to(DEV)
    x = nhwc(g["x"].half()).to(DEV)
    h = ops.group_norm(x, P("norm1.weight", (128,)), P("norm1.bias", (128,)), 1e-6, True)
    h = ops.conv2d(h, ops.repack_conv_weight(P("conv1.weight", (64, 128, 3, 3))), P("conv1.bias", (64,)))
    h = ops.group_norm(h, P("norm2.weight", (64,)), P("norm2.bias", (64,)), 1e-6, True)
    sk = ops.conv2d(x, ops.repack_conv_weight(P("nin_shortcut.weight", (64, 128, 1, 1))), P("nin_shortcut.bias", (64,)), 1)
    y = ops.conv2d(h, ops.repack_conv_weight(P("conv2.weight", (64, 64, 3, 3))), P("conv2.bias", (64,)), residual=sk)
    assert rel_l2(nchw(y.float().cpu()), g["y"]) < 3e-3
    g = load_golden("block_vae_attn")
    P = lambda k, s: W.synth_tensor(f"blk.vattn.{k}", s).half().to(DEV)
    x = nhwc(g["x"].half()).to(DEV)                       # [1, 8, 8, 64]
    n, h, w, c = x.shape
    t = ops.group_norm(x, P("norm.weight", (c,)), P("norm.bias", (c,)), 1e-6)
    lin = lambda name: ops.linear(t.view(n, h * w, c), P(name + ".weight", (c, c, 1, 1)).view(c, c).contiguous(), P(name + ".bias", (c,)))
    a = ops.attention(lin("q"), lin("k"), lin("v"), 1)
    y = ops.linear(a, P("proj_out.weight", (c, c, 1, 1)).view(c, c).contiguous(), P("proj_out.bias", (c,)), residual=x.view(n, h * w, c))
    assert rel_l2(nchw(y.view(n, h, w, c).float().cpu()), g["y"]) < 3e-3


def test_bislerp_on_device():
    """a18: the hires-fix latent upscale on the HIP kernel (ld_op_bislerp) vs the reference's bislerp (LD.py:429-518),
    including a zero vector and two identical neighbours (the reference's edge branches), 2x and odd sizes."""
    from lightdiffusion_amd import nodes, ops
    g = load_golden("bislerp")
    x = g["x"].to(DEV)
    assert rel_l2(ops.bislerp(x, 12, 16).cpu(), g["y2x"]) < 1e-5
    assert rel_l2(ops.bislerp(x, 9, 11).cpu(), g["y_odd"]) < 1e-5
    assert rel_l2(nodes.bislerp(g["x"], 12, 16), g["y2x"]) < 1e-5                      # host tensor in, host tensor out
    up = nodes.LatentUpscale().upscale({"samples": torch.randn(1, 4, 8, 8)}, "bislerp", 128, 128)[0]["samples"]
    assert up.shape == (1, 4, 16, 16) and up.device.type == "cpu"
    big = torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(2))          # config #5's 64 -> 128 latent upscale
    from oracle import sd15_ref as O
    assert rel_l2(ops.bislerp(big.to(DEV), 128, 128).cpu(), O.bislerp(big, 128, 128)) < 1e-5


# ------------------------------------------------------------------ secondary seam: the `operations=` classes
class _RefShapedResBlock(nn.Module):
    """The wiring of the reference's ResBlock1 (LD.py:5189-5287: in_layers / emb_layers / out_layers / skip_connection, same
    state-dict names), built from an injected `operations` namespace exactly as the reference builds it."""

    def __init__(self, channels, emb_channels, out_channels, operations, dtype=None, device=None):
        super().__init__()
        kw = dict(dtype=dtype, device=device)
        self.in_layers = nn.Sequential(operations.GroupNorm(32, channels, **kw), nn.SiLU(),
                                       operations.conv_nd(2, channels, out_channels, 3, padding=1, **kw))
        self.emb_layers = nn.Sequential(nn.SiLU(), operations.Linear(emb_channels, out_channels, **kw))
        self.out_layers = nn.Sequential(operations.GroupNorm(32, out_channels, **kw), nn.SiLU(), nn.Dropout(p=0.0),
                                        operations.conv_nd(2, out_channels, out_channels, 3, padding=1, **kw))
        self.skip_connection = nn.Identity() if out_channels == channels else operations.conv_nd(2, channels, out_channels, 1, **kw)

    def forward(self, x, emb):
        h = self.in_layers(x)
        h = h + self.emb_layers(emb).type(h.dtype)[..., None, None]
        return self.skip_connection(x) + self.out_layers(h)


@pytest.mark.parametrize("tag,cin,cout", [("res_skip", 64, 128), ("res_id", 64, 64)])
def test_operations_namespace_builds_reference_resblock(tag, cin, cout):
    from lightdiffusion_amd import ops
    g = load_golden("block_" + tag)
    m = _RefShapedResBlock(cin, 256, cout, ops, dtype=torch.float16, device=DEV)
    sd = {k: W.synth_tensor(f"blk.{tag}.{k}", tuple(v.shape)) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)                    # torch state-dict names and shapes of the reference module
    with torch.inference_mode():
        y = m(g["x"].to(DEV), g["emb"].to(DEV))           # fp32 NCHW in, fp32 NCHW out (manual_cast semantics)
    assert y.dtype == torch.float32 and y.shape == g["y"].shape
    assert rel_l2(y.cpu(), g["y"]) < 3e-3
    with pytest.raises(ValueError):
        ops.conv_nd(3, 4, 4, 3)
    with pytest.raises(NotImplementedError):
        ops.optimized_attention(torch.zeros(1, 8, 16), torch.zeros(1, 8, 16), torch.zeros(1, 8, 16), 2, mask=torch.zeros(8, 8))


def test_optimized_attention_and_layernorm_classes():
    from lightdiffusion_amd import ops
    g = load_golden("attention")
    for heads, key in ((2, "y_h2"), (10, "y_h10")):
        y = ops.optimized_attention(g["q"].to(DEV), g["k"].to(DEV), g["v"].to(DEV), heads)
        assert y.dtype == torch.float32 and rel_l2(y.cpu(), g[key]) < 3e-3
    ln = ops.LayerNorm(320, dtype=torch.float16, device=DEV)
    ln.load_state_dict({"weight": torch.rand(320) + 0.5, "bias": torch.randn(320) * 0.1})
    x = torch.randn(3, 50, 320, generator=torch.Generator().manual_seed(4))
    ref = F.layer_norm(x.half().float(), (320,), ln.weight.float().cpu(), ln.bias.float().cpu(), 1e-5)
    assert rel_l2(ln(x.to(DEV)).cpu(), ref) < 2e-3


# ------------------------------------------------------------------ LayerNorm fold under adverse statistics
@pytest.mark.parametrize("mean_over_std,scale", [(0.0, 1.0), (50.0, 1.0), (50.0, 20.0), (-8.0, 100.0)])
def test_layernorm_fold_large_mean(mean_over_std, scale):
    """The folded path computes var = E[x^2] - mu^2 and rstd * (acc - mu * wsum): both cancel when |mu| >> std.  Rows with a
    mean of 50 std and magnitudes ~1e3 (what a real checkpoint's residual stream can look like) against fp32 LayerNorm + GEMM
    on the same fp16 t."""
    from lightdiffusion_amd import ops
    M, C, N = 512, 320, 640
    g = torch.Generator().manual_seed(int(abs(mean_over_std)) + int(scale))
    x = ((torch.randn(M, C, generator=g) + mean_over_std) * scale).half()
    eye = torch.eye(C).half()
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).half(), (0.1 * torch.randn(C, generator=g)).half()
    w, b = (torch.randn(N, C, generator=g) / math.sqrt(C)).half(), (0.1 * torch.randn(N, generator=g)).half()
    t, y = ops.linear_ln(x.to(DEV), eye.to(DEV), None, gamma.to(DEV), beta.to(DEV), w.to(DEV), b.to(DEV))
    assert torch.equal(t.cpu(), x)                        # identity producer: t is x exactly
    ref = F.linear(F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    assert rel_l2(y.float().cpu(), ref) < 3e-3


# ------------------------------------------------------------------ the product call surface runs the graph path
def _conds(g):
    return [[g["pos"], {"pooled_output": None}]], [[g["neg"], {"pooled_output": None}]]


@pytest.fixture(scope="module")
def tiny_stack():
    from lightdiffusion_amd import nodes
    return nodes.load_synthetic(DEV, max_batch=2, max_hw=(16, 16), tiny=True)


def test_ksampler_graph_equals_eager_bitwise(tiny_stack):
    """`KSampler2.sample -> common_ksampler -> CFGGuider -> sampling_function` replays a hipGraph (what bench.py times);
    the same call with the graph off must give bit-identical latents.  The un-fused wrapper-hook route runs the plain forward on cat([x, x])
    where the product path runs the CFG-pair forward (shared layers evaluated once): the same latents up to tile / split rounding."""
    from lightdiffusion_amd import nodes
    model = tiny_stack[0]
    g = load_golden("samplers")
    pos, neg = _conds(g)
    lat = nodes.EmptyLatentImage().generate(128, 96, 2)[0]
    outs = {}
    for mode, opts in (("graph", {}), ("eager", {"ld_use_graph": False}), ("hook", {"ld_eager_unbatched": True})):
        m = model.clone()
        m.model_options.update(opts)
        outs[mode] = nodes.KSampler2().sample(m, 1234, 6, 7.5, "euler_ancestral", "normal", pos, neg, lat)[0]["samples"]
        outs[mode + "_2m"] = nodes.KSampler2().sample(m, 99, 5, 7.0, "dpmpp_2m_sde", "karras", pos, neg, lat)[0]["samples"]
    unet = model.model.diffusion_model
    assert any(d._graph is not None for d in unet._denoisers.values()), "the product path did not capture a graph"
    for k in ("", "_2m"):
        assert torch.equal(outs["graph" + k], outs["eager" + k])
        assert rel_l2(outs["graph" + k], outs["hook" + k]) < 5e-3
    assert torch.isfinite(outs["graph"]).all() and not torch.equal(outs["graph"][0], outs["graph"][1])


def test_second_prompt_same_shape_is_not_stale(tiny_stack):
    """two runs on one model with different prompts of the same token count and batch.  The cached
    cross-attention K / V^T must follow the prompt (content, not tensor address)."""
    from lightdiffusion_amd import nodes
    from oracle import sd15_ref as O
    model = tiny_stack[0]
    g = load_golden("samplers")
    cfg = W.tiny_unet_config()
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    ms = O.ModelSampling()
    den = lambda xx, ss, cc: O.apply_model(sd, cfg, ms, xx, ss, cc)
    lat = nodes.EmptyLatentImage().generate(128, 96, 1)[0]
    gen = torch.Generator().manual_seed(77)
    pos2 = torch.randn(1, 77, cfg["context_dim"], generator=gen)
    for pos in (g["pos"], pos2, g["pos"]):
        conds = [[pos, {"pooled_output": None}]], [[g["neg"], {"pooled_output": None}]]
        out = nodes.KSampler2().sample(model, 1234, 4, 7.5, "euler_ancestral", "normal", conds[0], conds[1], lat)[0]["samples"]
        ref = O.ksample(den, ms, 1234, 4, 7.5, "euler_ancestral", "normal", pos, g["neg"], torch.zeros(1, 4, 12, 16))
        assert rel_l2(out, ref) < 3e-2
    # and through the raw wrapper hook (what the reference's calc_cond_batch calls), with recycled tensor addresses
    unet = model.model.diffusion_model
    x, s = torch.randn(2, 4, 12, 16, generator=gen), torch.tensor([2.0, 2.0])
    for pos in (g["pos"], pos2):
        ctx = torch.cat([g["neg"], pos]).to(DEV)          # freed and re-allocated at the same address on the next iteration
        out = unet(None, {"input": x, "timestep": s, "c": {"c_crossattn": ctx, "transformer_options": {}}, "cond_or_uncond": [1, 0]}).cpu()
        assert rel_l2(out, den(x, s, ctx.cpu())) < UNET_TOL
        del ctx


def test_long_prompt_and_hires_through_nodes(tiny_stack):
    """a > 75-token prompt (154 context tokens) and the hires pass (2x latent) through the node surface of
    a stack loaded with the default plan — both used to fail with ERR_SHAPE; the plan now grows on demand."""
    from lightdiffusion_amd import nodes
    model, clip, vae = nodes.load_synthetic(DEV, max_batch=1, max_hw=(8, 8), tiny=True)
    long_toks = [[(49406, 1.0)] + [(1000 + i, 1.0) for i in range(75)] + [(49407, 1.0)],
                 [(49406, 1.0)] + [(2000 + i, 1.1) for i in range(20)] + [(49407, 1.0)] * 56]
    neg = [[(49406, 1.0)] + [(49407, 1.0)] * 76]
    img = nodes.txt2img(model, clip.clone(), vae, long_toks, neg, width=64, height=64, batch_size=1, seed=3, steps=3, cfg=6.0,
                        sampler_name="euler_ancestral", scheduler="normal", hires=True)
    assert img.shape == (1, 128, 128, 3) and torch.isfinite(img).all()
    assert model.model.diffusion_model.ctx_shape == (2, 154)


def test_lora_merged_unet_and_clip_match_reference():
    """f3 on the device: a checkpoint with the golden's LoRA merged by `CheckpointLoaderSimple(lora=...)` — UNet step and CLIP
    hidden state against the reference's patched model (ModelPatcher.add_patches + calculate_weight, LD.py:3297-3424)."""
    from lightdiffusion_amd import nodes
    from test_host_cpu import _lora_from_golden, _synthetic_checkpoint
    g = load_golden("lora_tiny")
    sd, ucfg, vcfg, ccfg = _synthetic_checkpoint()
    sd.update({"model.diffusion_model." + k: v for k, v in W.synth_state_dict(W.unet_param_shapes(ucfg)).items()})   # fp32 base weights
    loader = nodes.CheckpointLoaderSimple(DEV, max_batch=1, max_hw=(16, 16), clip_heads=ccfg["num_attention_heads"])
    model, clip, vae = loader.load_checkpoint(dict(sd), lora=_lora_from_golden(g), lora_strength=0.8, lora_strength_clip=0.6)
    unet = model.model.diffusion_model
    unet.set_context(g["ctx"])
    den = unet.forward(g["x"].to(DEV), g["sigma"].to(DEV)).cpu()
    assert rel_l2(den, g["denoised"]) < UNET_TOL
    inter = clip.text_model(g["tokens"], intermediate_output=-2)[1].cpu()
    assert rel_l2(inter, g["clip_inter_m2"]) < 5e-3
    # without the LoRA the same stack is measurably different (the fixture is not a no-op)
    model0, clip0, _ = loader.load_checkpoint(dict(sd))
    u0 = model0.model.diffusion_model
    u0.set_context(g["ctx"])
    assert rel_l2(u0.forward(g["x"].to(DEV), g["sigma"].to(DEV)).cpu(), g["denoised"]) > 2e-2

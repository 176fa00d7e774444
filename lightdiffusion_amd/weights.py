"""Deterministic synthetic SD1.x weights (no checkpoint can be downloaded here — SURVEY.md §8c).

Every tensor is a pure function of (its checkpoint key name, its shape, a global seed), so the oracle,
the golden generator, the HIP path and bench.py all see bit-identical fp32 masters without shipping
gigabytes of fixtures.  Key names follow the single-file SD1.x checkpoint layout the reference loads
(`model.diffusion_model.*`, `first_stage_model.*`, LD.py:6446-6465), minus the prefix.

`disable_weight_init` skips `reset_parameters` in the reference (LD.py:2363), so *every* parameter
must be filled explicitly, including `out.2` which the ctor zeroes (LD.py:5675).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch

# ---------------------------------------------------------------- configs (result of detect_unet_config, LD.py:6065-6182 + sm_SD15 5964-5976)

def sd15_unet_config() -> dict:
    return dict(in_channels=4, out_channels=4, model_channels=320, channel_mult=[1, 2, 4, 4],
                num_res_blocks=[2, 2, 2, 2], transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0],
                transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], transformer_depth_middle=1,
                context_dim=768, num_heads=8)


def tiny_unet_config() -> dict:
    """Same topology as SD1.5 at 1/5 width (d_head 8/16/32); used by fast parity tests."""
    return dict(in_channels=4, out_channels=4, model_channels=64, channel_mult=[1, 2, 4, 4],
                num_res_blocks=[2, 2, 2, 2], transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0],
                transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], transformer_depth_middle=1,
                context_dim=64, num_heads=8)


def sd15_vae_config() -> dict:
    # VAE.__init__ default decoder config, LD.py:6312-6323
    return dict(z_channels=4, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, out_ch=3)


def tiny_vae_config() -> dict:
    return dict(z_channels=4, ch=64, ch_mult=[1, 2, 4, 4], num_res_blocks=2, out_ch=3)


def sd15_clip_config() -> dict:
    # _internal/clip/sd1_clip_config.json
    return dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu")


def tiny_clip_config() -> dict:
    return dict(vocab_size=49408, hidden_size=64, intermediate_size=256, num_hidden_layers=3,
                num_attention_heads=4, max_position_embeddings=77, hidden_act="quick_gelu")


# ---------------------------------------------------------------- parameter shapes

def unet_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """name -> shape for UNetModel1 with `cfg` (mirrors the ctor walk at LD.py:5379-5686)."""
    mc, cm = cfg["model_channels"], cfg["channel_mult"]
    ted, ctx = mc * 4, cfg["context_dim"]
    td_in = list(cfg["transformer_depth"])
    td_out = list(cfg["transformer_depth_output"])
    shp: Dict[str, Tuple[int, ...]] = {}

    def lin(p, o, i, bias=True):
        shp[p + ".weight"] = (o, i)
        if bias:
            shp[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        shp[p + ".weight"] = (o, i, k, k)
        shp[p + ".bias"] = (o,)

    def norm(p, c):
        shp[p + ".weight"] = (c,)
        shp[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin)
        conv(p + ".in_layers.2", cout, cin, 3)
        lin(p + ".emb_layers.1", cout, ted)
        norm(p + ".out_layers.0", cout)
        conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cout, cin, 1)

    def st(p, c):
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        b = p + ".transformer_blocks.0"
        for a, kv in (("attn1", c), ("attn2", ctx)):
            lin(f"{b}.{a}.to_q", c, c, bias=False)
            lin(f"{b}.{a}.to_k", c, kv, bias=False)
            lin(f"{b}.{a}.to_v", c, kv, bias=False)
            lin(f"{b}.{a}.to_out.0", c, c)
        lin(f"{b}.ff.net.0.proj", 8 * c, c)
        lin(f"{b}.ff.net.2", c, 4 * c)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{b}.{n}", c)
        conv(p + ".proj_out", c, c, 1)

    lin("time_embed.0", ted, mc)
    lin("time_embed.2", ted, ted)
    conv("input_blocks.0.0", mc, cfg["in_channels"], 3)
    ch, chans, idx = mc, [mc], 1
    for level, mult in enumerate(cm):
        for _ in range(cfg["num_res_blocks"][level]):
            res(f"input_blocks.{idx}.0", ch, mult * mc)
            ch = mult * mc
            if td_in.pop(0) > 0:
                st(f"input_blocks.{idx}.1", ch)
            chans.append(ch)
            idx += 1
        if level != len(cm) - 1:
            conv(f"input_blocks.{idx}.0.op", ch, ch, 3)
            chans.append(ch)
            idx += 1
    res("middle_block.0", ch, ch)
    if cfg["transformer_depth_middle"] > 0:
        st("middle_block.1", ch)
    res("middle_block.2", ch, ch)
    idx = 0
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(cfg["num_res_blocks"][level] + 1):
            ich = chans.pop()
            res(f"output_blocks.{idx}.0", ch + ich, mc * mult)
            ch = mc * mult
            j = 1
            if td_out.pop() > 0:
                st(f"output_blocks.{idx}.{j}", ch)
                j += 1
            if level and i == cfg["num_res_blocks"][level]:
                conv(f"output_blocks.{idx}.{j}.conv", ch, ch, 3)
            idx += 1
    norm("out.0", ch)
    conv("out.2", cfg["out_channels"], mc, 3)
    return shp


def vae_decoder_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """name -> shape for `post_quant_conv` + `decoder.*` (LD.py:3461-3473, 3761-3882)."""
    ch, cm, nrb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    shp: Dict[str, Tuple[int, ...]] = {}

    def conv(p, o, i, k):
        shp[p + ".weight"] = (o, i, k, k)
        shp[p + ".bias"] = (o,)

    def norm(p, c):
        shp[p + ".weight"] = (c,)
        shp[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    z = cfg["z_channels"]
    conv("post_quant_conv", z, z, 1)
    bi = ch * cm[-1]
    conv("decoder.conv_in", bi, z, 3)
    res("decoder.mid.block_1", bi, bi)
    norm("decoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"decoder.mid.attn_1.{n}", bi, bi, 1)
    res("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(len(cm))):
        bo = ch * cm[lvl]
        for b in range(nrb + 1):
            res(f"decoder.up.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi)
    conv("decoder.conv_out", cfg["out_ch"], bi, 3)
    return shp


def vae_encoder_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """name -> shape for `encoder.*` + `quant_conv` (LD.py:3649-3758, 3468)."""
    ch, cm, nrb, z = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"], cfg["z_channels"]
    shp: Dict[str, Tuple[int, ...]] = {}

    def conv(p, o, i, k):
        shp[p + ".weight"] = (o, i, k, k)
        shp[p + ".bias"] = (o,)

    def norm(p, c):
        shp[p + ".weight"] = (c,)
        shp[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    conv("encoder.conv_in", ch, cfg["out_ch"], 3)
    bi = ch
    for lvl in range(len(cm)):
        bo = ch * cm[lvl]
        for b in range(nrb):
            res(f"encoder.down.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != len(cm) - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi)
    norm("encoder.mid.attn_1.norm", bi)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"encoder.mid.attn_1.{n}", bi, bi, 1)
    res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi)
    conv("encoder.conv_out", 2 * z, bi, 3)
    conv("quant_conv", 2 * z, 2 * z, 1)
    return shp


def clip_param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """name -> shape for CLIPTextModel (LD.py:4268-4487), keys as under `transformer.`."""
    h, f = cfg["hidden_size"], cfg["intermediate_size"]
    shp: Dict[str, Tuple[int, ...]] = {
        "text_model.embeddings.token_embedding.weight": (cfg["vocab_size"], h),
        "text_model.embeddings.position_embedding.weight": (cfg["max_position_embeddings"], h),
        "text_model.final_layer_norm.weight": (h,), "text_model.final_layer_norm.bias": (h,),
    }
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            shp[f"{p}.self_attn.{n}.weight"] = (h, h)
            shp[f"{p}.self_attn.{n}.bias"] = (h,)
        for n in ("layer_norm1", "layer_norm2"):
            shp[f"{p}.{n}.weight"] = (h,)
            shp[f"{p}.{n}.bias"] = (h,)
        shp[f"{p}.mlp.fc1.weight"] = (f, h)
        shp[f"{p}.mlp.fc1.bias"] = (f,)
        shp[f"{p}.mlp.fc2.weight"] = (h, f)
        shp[f"{p}.mlp.fc2.bias"] = (h,)
    return shp


# ---------------------------------------------------------------- generator

def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """fp32 master value of one parameter.

    matrices / conv kernels ~ N(0, 1/fan_in); norm gains ~ 1 + 0.1 N(0,1); biases ~ 0.05 N(0,1);
    embeddings ~ 0.5 N(0,1).  CPU generator, so identical here and on the GPU box.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if "embedding.weight" in name:
        return x * 0.5
    if len(shape) >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return x * (1.0 / fan_in ** 0.5)
    if name.endswith(".weight"):
        return 1.0 + 0.1 * x
    return 0.05 * x


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32,
                     names: Iterable[str] | None = None) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, shapes[k], seed).to(dtype) for k in (names if names is not None else shapes)}

"""CPU: the oracle restatement (oracle/sd15_ref.py) against fixtures produced by the reference's own classes
(oracle/make_golden.py).  This is what pins the oracle."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_golden, rel_l2
from lightdiffusion_amd import weights as W
from oracle import sd15_ref as O

torch.set_grad_enabled(False)


def sdict(shapes_or_names, prefix=""):
    return {k: W.synth_tensor(prefix + k, s) for k, s in shapes_or_names.items()}


def test_schedules():
    g = load_golden("schedules")
    ms = O.ModelSampling()
    assert torch.equal(ms.sigmas, g["sigmas"]) and torch.equal(ms.log_sigmas, g["log_sigmas"])
    assert abs(float(ms.sigma_min) - 0.029167) < 1e-5 and abs(float(ms.sigma_max) - 14.614641) < 1e-4   # SURVEY §8 a3 probes
    assert torch.equal(O.calculate_sigmas(ms, "karras", 20), g["karras20"])
    assert torch.equal(O.calculate_sigmas(ms, "normal", 30), g["normal30"])
    assert torch.equal(O.calculate_sigmas(ms, "normal", 10, 0.45), g["normal10_d045"])
    assert torch.equal(ms.timestep(g["probe_sigma"]), g["probe_t"])
    assert torch.equal(ms.sigma(g["tq"]), g["sigma_of_t"])
    assert torch.equal(O.timestep_embedding(g["temb_t"], 320), g["temb"])
    for (a, b), (d, u) in zip(((14.6, 11.7), (1.0, 0.5), (0.05, 0.0)), g["anc"].tolist()):
        assert O.get_ancestral_step(a, b) == (d, u)


def _res_shapes(cin, cout, ted):
    s = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,), "in_layers.2.weight": (cout, cin, 3, 3), "in_layers.2.bias": (cout,),
         "emb_layers.1.weight": (cout, ted), "emb_layers.1.bias": (cout,), "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,),
         "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,)}
    if cin != cout:
        s.update({"skip_connection.weight": (cout, cin, 1, 1), "skip_connection.bias": (cout,)})
    return s


@pytest.mark.parametrize("tag,cin,cout", [("res_skip", 64, 128), ("res_id", 64, 64)])
def test_resblock(tag, cin, cout):
    g = load_golden("block_" + tag)
    sd = {"r." + k: v for k, v in sdict(_res_shapes(cin, cout, 256), f"blk.{tag}.").items()}
    assert rel_l2(O.resblock(g["x"], g["emb"], sd, "r"), g["y"]) < 2e-6


def test_down_up():
    g = load_golden("block_down")
    sd = {"op.weight": W.synth_tensor("blk.down.op.weight", (64, 64, 3, 3)), "op.bias": W.synth_tensor("blk.down.op.bias", (64,))}
    assert rel_l2(O._conv(g["x"], sd, "op", stride=2), g["y"]) < 2e-6
    g = load_golden("block_up")
    sd = {"conv.weight": W.synth_tensor("blk.up.conv.weight", (64, 64, 3, 3)), "conv.bias": W.synth_tensor("blk.up.conv.bias", (64,))}
    up = lambda size: O._conv(torch.nn.functional.interpolate(g["x"], size=size, mode="nearest"), sd, "conv")
    assert rel_l2(up((12, 10)), g["y"]) < 2e-6 and rel_l2(up((11, 9)), g["y_odd"]) < 2e-6


def _tb_shapes(c, cd):
    s = {}
    for a, kv in (("attn1", c), ("attn2", cd)):
        s.update({f"{a}.to_q.weight": (c, c), f"{a}.to_k.weight": (c, kv), f"{a}.to_v.weight": (c, kv),
                  f"{a}.to_out.0.weight": (c, c), f"{a}.to_out.0.bias": (c,)})
    s.update({"ff.net.0.proj.weight": (8 * c, c), "ff.net.0.proj.bias": (8 * c,), "ff.net.2.weight": (c, 4 * c), "ff.net.2.bias": (c,)})
    for n in ("norm1", "norm2", "norm3"):
        s.update({n + ".weight": (c,), n + ".bias": (c,)})
    return s


@pytest.mark.parametrize("tag,c,heads,cd", [("h8d8", 64, 8, 64), ("h2d40", 80, 2, 96)])
def test_transformer_block(tag, c, heads, cd):
    g = load_golden("block_tb_" + tag)
    sd = {"t." + k: v for k, v in sdict(_tb_shapes(c, cd), f"blk.tb.{tag}.").items()}
    assert rel_l2(O.transformer_block(g["x"], g["ctx"], sd, "t", heads), g["y"]) < 5e-6


def test_spatial_transformer_attention_geglu():
    g = load_golden("block_st")
    shp = {"norm.weight": (64,), "norm.bias": (64,), "proj_in.weight": (64, 64, 1, 1), "proj_in.bias": (64,),
           "proj_out.weight": (64, 64, 1, 1), "proj_out.bias": (64,)}
    shp.update({"transformer_blocks.0." + k: v for k, v in _tb_shapes(64, 64).items()})
    sd = {"s." + k: v for k, v in sdict(shp, "blk.st.").items()}
    assert rel_l2(O.spatial_transformer(g["x"], g["ctx"], sd, "s", 8), g["y"]) < 5e-6
    g = load_golden("attention")
    assert rel_l2(O.attention(g["q"], g["k"], g["v"], 2), g["y_h2"]) < 2e-6
    assert rel_l2(O.attention(g["q"], g["k"], g["v"], 10), g["y_h10"]) < 2e-6
    g = load_golden("block_geglu")
    w, b = W.synth_tensor("blk.geglu.proj.weight", (512, 64)), W.synth_tensor("blk.geglu.proj.bias", (512,))
    a, gate = torch.nn.functional.linear(g["x"], w, b).chunk(2, dim=-1)
    assert rel_l2(a * torch.nn.functional.gelu(gate), g["y"]) < 2e-6


def test_vae_blocks():
    g = load_golden("block_vae_res")
    shp = {"norm1.weight": (128,), "norm1.bias": (128,), "conv1.weight": (64, 128, 3, 3), "conv1.bias": (64,), "norm2.weight": (64,),
           "norm2.bias": (64,), "conv2.weight": (64, 64, 3, 3), "conv2.bias": (64,), "nin_shortcut.weight": (64, 128, 1, 1), "nin_shortcut.bias": (64,)}
    sd = {"v." + k: v for k, v in sdict(shp, "blk.vres.").items()}
    assert rel_l2(O.vae_resblock(g["x"], sd, "v"), g["y"]) < 2e-6
    g = load_golden("block_vae_attn")
    shp = {"norm.weight": (64,), "norm.bias": (64,)}
    for n in ("q", "k", "v", "proj_out"):
        shp.update({n + ".weight": (64, 64, 1, 1), n + ".bias": (64,)})
    sd = {"a." + k: v for k, v in sdict(shp, "blk.vattn.").items()}
    assert rel_l2(O.vae_attn(g["x"], sd, "a"), g["y"]) < 2e-6


@pytest.mark.parametrize("name,cfg", [("unet_tiny_16x16", "tiny"), ("unet_tiny_8x12", "tiny"), ("unet_sd15_64x64", "sd15")])
def test_unet(name, cfg):
    g = load_golden(name)
    cfg = W.tiny_unet_config() if cfg == "tiny" else W.sd15_unet_config()
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    ms = O.ModelSampling()
    assert torch.equal(ms.timestep(g["sigma"]).float(), g["t"])
    xc = g["x"] / (g["sigma"].view(-1, 1, 1, 1) ** 2 + 1.0) ** 0.5
    assert rel_l2(O.unet_forward(sd, cfg, xc, g["t"], g["ctx"]), g["eps"]) < 2e-5
    assert rel_l2(O.apply_model(sd, cfg, ms, g["x"], g["sigma"], g["ctx"]), g["denoised"]) < 2e-5


def test_samplers_and_cfg_order():
    g = load_golden("samplers")
    cfg = W.tiny_unet_config()
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    ms = O.ModelSampling()
    calls = []

    def den(x, s, c):
        calls.append((x.clone(), s.clone(), c.clone()))
        return O.apply_model(sd, cfg, ms, x, s, c)

    lat0 = torch.zeros(1, 4, 12, 16)
    y = O.ksample(den, ms, 1234, 6, 7.5, "euler_ancestral", "normal", g["pos"], g["neg"], lat0)
    assert rel_l2(y, g["euler_a_txt2img"]) < 1e-4
    # wrapper-hook contract (LD.py:2558-2567): batch order [uncond, cond], sigma repeated, cond_or_uncond == [1, 0]
    x0, s0, c0 = calls[0]
    assert g["hook_cond_or_uncond"].tolist() == [1, 0]
    assert torch.equal(x0, g["hook_input"]) and torch.equal(s0, g["hook_timestep"]) and torch.equal(c0, g["hook_ctx"])
    assert torch.equal(c0[0], g["neg"][0]) and torch.equal(c0[1], g["pos"][0])
    y = O.ksample(den, ms, 77, 4, 8.0, "euler_ancestral", "normal", g["pos"], g["neg"], g["lat2"], denoise_strength=0.45)
    assert rel_l2(y, g["euler_a_img2img"]) < 1e-4
    y = O.ksample(den, ms, 99, 6, 7.0, "dpmpp_2m_sde", "karras", g["pos"], g["neg"], lat0, sampler_opts={"eta": 0.0})
    assert rel_l2(y, g["dpmpp2m_eta0"]) < 1e-4
    gen = torch.Generator().manual_seed(5)
    ns = lambda s, sn: torch.randn(lat0.shape, generator=gen)
    y = O.ksample(den, ms, 99, 6, 7.0, "dpmpp_2m_sde", "karras", g["pos"], g["neg"], lat0, sampler_opts={"eta": 1.0, "noise_sampler": ns})
    assert rel_l2(y, g["dpmpp2m_sde_injected"]) < 1e-4


def test_toy_sampler_trajectories():
    g = load_golden("samplers")
    toy = lambda x, s: x * (1.0 / (1.0 + s.view(-1, 1, 1, 1) ** 2))
    sig = O.sigmas_karras(8, 0.03, 14.6)
    torch.manual_seed(7)
    assert rel_l2(O.sample_euler_ancestral(toy, g["toy_x0"], sig), g["toy_euler_a"]) < 1e-6
    assert rel_l2(O.sample_dpmpp_2m_sde(toy, g["toy_x0"], sig, eta=0.0), g["toy_dpmpp2m"]) < 1e-6
    assert rel_l2(O.sample_dpmpp_2m_sde(toy, g["toy_x0"], sig, eta=0.0, solver_type="heun"), g["toy_dpmpp2m_heun"]) < 1e-6


def test_dpm_adaptive():
    g = load_golden("samplers")
    toy = lambda x, s: x * (1.0 / (1.0 + s.view(-1, 1, 1, 1) ** 2))
    x, info = O.sample_dpm_adaptive(toy, g["toy_x0"], 0.03, 14.6)
    assert [info["steps"], info["nfe"], info["n_accept"], info["n_reject"]] == g["toy_dpm_adaptive_info"].tolist()
    assert rel_l2(x, g["toy_dpm_adaptive"]) < 1e-5
    cfg = W.tiny_unet_config()
    sd = W.synth_state_dict(W.unet_param_shapes(cfg))
    ms = O.ModelSampling()
    den = lambda xx, ss, cc: O.apply_model(sd, cfg, ms, xx, ss, cc)
    y = O.ksample(den, ms, 99, 6, 7.0, "dpm_adaptive", "karras", g["pos"], g["neg"], torch.zeros(1, 4, 12, 16))
    assert rel_l2(y, g["dpm_adaptive_tiny"]) < 1e-3


def test_vae_decode():
    g = load_golden("vae_tiny")
    cfg = W.tiny_vae_config()
    sd = W.synth_state_dict(W.vae_decoder_param_shapes(cfg))
    y = O.vae_decode(sd, cfg, g["z"])
    assert y.shape == g["img"].shape and float((y - g["img"]).abs().max()) < 2e-5
    g = load_golden("vae_sd15")
    cfg = W.sd15_vae_config()
    sd = W.synth_state_dict(W.vae_decoder_param_shapes(cfg))
    y = O.vae_decode(sd, cfg, g["z"])
    assert float((y[:, ::4, ::4] - g["img_sub"]).abs().max()) < 5e-5
    assert abs(float(y.mean() - g["mean"])) < 1e-5 and abs(float(y.std() - g["std"])) < 1e-5


def test_clip_and_prompt_weight_lerp():
    g = load_golden("clip_tiny")
    cfg = W.tiny_clip_config()
    sd = W.synth_state_dict(W.clip_param_shapes(cfg))
    assert rel_l2(O.clip_text_model(sd, cfg, g["tokens"], layer_idx=-2), g["inter_m2"]) < 5e-6
    assert rel_l2(O.clip_text_model(sd, cfg, g["tokens"], layer_idx=None), g["last"]) < 5e-6
    enc = lambda t: O.clip_text_model(sd, cfg, t, layer_idx=-2)
    toks = g["tokens"][0].tolist()
    pairs = [[(t, 1.3 if 2 <= i < 5 else 1.0) for i, t in enumerate(toks)]]
    empty = [49406, 49407] + [49407] * 75
    z = O.encode_token_weights(enc, pairs, empty)
    base, e = enc(g["tokens"][:1]), enc(torch.tensor([empty]))
    assert torch.allclose(z[0, 5:], base[0, 5:]) and torch.allclose(z[0, 2:5], (base[0, 2:5] - e[0, 2:5]) * 1.3 + e[0, 2:5], atol=1e-6)


def test_bislerp():
    g = load_golden("bislerp")
    assert rel_l2(O.bislerp(g["x"], 12, 16), g["y2x"]) < 2e-6
    assert rel_l2(O.bislerp(g["x"], 9, 11), g["y_odd"]) < 2e-6


@pytest.mark.parametrize("tag", ["tiny", "sd15"])
def test_vae_encode(tag):
    g = load_golden("vae_enc_" + tag)
    cfg = W.tiny_vae_config() if tag == "tiny" else W.sd15_vae_config()
    sd = W.synth_state_dict(W.vae_encoder_param_shapes(cfg))
    m = O.vae_encode_moments(sd, cfg, g["pixels"])
    assert rel_l2(m, g["moments"]) < 2e-5
    torch.manual_seed(58)
    assert rel_l2(O.vae_sample(m), g["z_seed58"]) < 2e-5


def test_config_size_goldens():
    """The oracle at the size of BASELINE config #2's image (oracle/make_golden.py `configs`): SD1.5 VAE decoder at 64x64 latents
    (512^2).  The 128x128-latent UNet and the 1024^2 decode goldens are compared with the HIP path directly on the GPU
    (tests/test_configs_gpu.py): the restatement needs minutes of CPU at those sizes."""
    g = load_golden("vae_sd15_64x64")
    cfg = W.sd15_vae_config()
    y = O.vae_decode(W.synth_state_dict(W.vae_decoder_param_shapes(cfg)), cfg, g["z"])
    c0 = int(g["crop_at"][0])
    assert float((y[:, ::4, ::4] - g["img_sub"]).abs().max()) < 5e-5 and float((y[:, c0:c0 + 96, c0:c0 + 96] - g["crop"]).abs().max()) < 5e-5
    assert abs(float(y.mean() - g["mean"])) < 1e-5

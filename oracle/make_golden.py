"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz|json by running the REFERENCE's own classes
(loaded by oracle/extract_ref.py from /root/reference, build container only) on seeded inputs with the
deterministic synthetic weights of lightdiffusion_amd/weights.py.

Run:  python oracle/make_golden.py            (≈2-3 min on 8 cores; rewrites every fixture)
The fixtures hold only inputs that cannot be regenerated from a seed, and expected outputs — never
reference source.  Weights are *not* stored: they are a pure function of (key name, shape, seed).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightdiffusion_amd import weights as W          # noqa: E402
from oracle.extract_ref import REF_PATH, load_reference         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ref = load_reference()
torch.set_grad_enabled(False)
torch.set_num_threads(os.cpu_count())


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def fill(module, prefix=""):
    """Fill every parameter of a reference module from the name-keyed generator (strict)."""
    sd = module.state_dict()
    new = {k: W.synth_tensor(prefix + k, tuple(v.shape)) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module


def ref_unet_config(cfg):
    c = dict(use_checkpoint=False, image_size=32, use_spatial_transformer=True, legacy=False, adm_in_channels=None,
             use_linear_in_transformer=False, use_temporal_resblock=False, use_temporal_attention=False)
    c.update({k: cfg[k] for k in ("in_channels", "out_channels", "model_channels", "num_res_blocks", "transformer_depth",
                                  "transformer_depth_output", "channel_mult", "transformer_depth_middle", "context_dim")})
    return c


def build_ref_model(cfg):
    """sm_SD15 → BaseModel (LD.py:5964, 5798) with fp32 weights, wrapped in a ModelPatcher (LD.py:3210)."""
    mcfg = ref.sm_SD15(ref_unet_config(cfg))
    mcfg.unet_config["num_heads"] = cfg["num_heads"]
    mcfg.set_inference_dtype(torch.float32, None)
    model = ref.BaseModel(mcfg, model_type=ref.ModelType.EPS, device=None)
    fill(model.diffusion_model)
    model.eval()
    cpu = torch.device("cpu")
    return model, ref.ModelPatcher(model, load_device=cpu, offload_device=cpu)


# ------------------------------------------------------------------ 1. schedules
def g_schedules():
    mcfg = ref.sm_SD15(ref_unet_config(W.tiny_unet_config()))
    ms = ref.model_sampling(mcfg, ref.ModelType.EPS)
    k20 = ref.calculate_sigmas(ms, "karras", 20)
    n30 = ref.calculate_sigmas(ms, "normal", 30)
    # KSampler1.set_steps with denoise 0.45, 10 steps (hires-fix, LD.py:3097-3104, 10592-10603)
    n10_045 = ref.calculate_sigmas(ms, "normal", int(10 / 0.45))[-11:]
    probe = torch.cat([k20[:-1], torch.tensor([0.03, 0.5, 1.0, 7.7, 14.6, 20.0])])
    tq = torch.tensor([0.0, 0.5, 10.25, 500.0, 998.75, 999.0])
    save("schedules", sigmas=ms.sigmas, log_sigmas=ms.log_sigmas, karras20=k20, normal30=n30, normal10_d045=n10_045,
         probe_sigma=probe, probe_t=ms.timestep(probe), tq=tq, sigma_of_t=ms.sigma(tq),
         temb_t=torch.tensor([0.0, 1.0, 37.0, 999.0]),
         temb=ref.timestep_embedding(torch.tensor([0.0, 1.0, 37.0, 999.0]), 320),
         anc=np.array([ref.get_ancestral_step(a, b) for a, b in ((14.6, 11.7), (1.0, 0.5), (0.05, 0.0))], dtype=np.float64))


# ------------------------------------------------------------------ 2. blocks
def g_blocks():
    ops = ref.disable_weight_init
    emb = rnd((2, 256), 11)
    # ResBlock1 with and without skip conv
    for tag, cin, cout in (("res_skip", 64, 128), ("res_id", 64, 64)):
        m = fill(ref.ResBlock1(cin, 256, 0.0, out_channels=cout, dims=2, operations=ops), f"blk.{tag}.")
        x = rnd((2, cin, 12, 10), 12)
        save("block_" + tag, x=x, emb=emb, y=m(x, emb))
    m = fill(ref.Downsample1(64, True, dims=2, out_channels=64, operations=ops), "blk.down.")
    x = rnd((2, 64, 12, 10), 13)
    save("block_down", x=x, y=m(x))
    m = fill(ref.Upsample1(64, True, dims=2, out_channels=64, operations=ops), "blk.up.")
    x = rnd((2, 64, 6, 5), 14)
    save("block_up", x=x, y=m(x, output_shape=(2, 64, 12, 10)), y_odd=m(x, output_shape=(2, 64, 11, 9)))
    # transformer pieces: C=64, heads=8 (d=8) and C=80, heads=2 (d=40)
    for tag, c, heads, cd in (("h8d8", 64, 8, 64), ("h2d40", 80, 2, 96)):
        ctx = rnd((2, 77, cd), 15)
        m = fill(ref.BasicTransformerBlock(c, heads, c // heads, context_dim=cd, operations=ops), f"blk.tb.{tag}.")
        x = rnd((2, 48, c), 16)
        save("block_tb_" + tag, x=x, ctx=ctx, y=m(x.clone(), context=ctx, transformer_options={"block": ("input", 1)}))
    m = fill(ref.SpatialTransformer(64, 8, 8, depth=1, context_dim=64, operations=ops), "blk.st.")
    x, ctx = rnd((2, 64, 8, 6), 17), rnd((2, 77, 64), 18)
    save("block_st", x=x, ctx=ctx, y=m(x, ctx, {"block": ("input", 1)}))
    q, k, v = rnd((2, 40, 80), 19), rnd((2, 77, 80), 20), rnd((2, 77, 80), 21)
    save("attention", q=q, k=k, v=v, y_h2=ref.attention_pytorch(q, k, v, 2), y_h10=ref.attention_pytorch(q, k, v, 10))
    # VAE blocks
    m = fill(ref.ResnetBlock(in_channels=128, out_channels=64, dropout=0.0), "blk.vres.")
    x = rnd((1, 128, 10, 12), 22)
    save("block_vae_res", x=x, y=m(x, None))
    m = fill(ref.AttnBlock(64), "blk.vattn.")
    x = rnd((1, 64, 8, 8), 23)
    save("block_vae_attn", x=x, y=m(x))
    # GEGLU alone (LD.py:4508-4515)
    m = fill(ref.GEGLU(64, 256), "blk.geglu.")
    x = rnd((2, 5, 64), 24)
    save("block_geglu", x=x, y=m(x))


# ------------------------------------------------------------------ 3. UNets
def g_unets():
    for tag, cfg, shapes in (("tiny", W.tiny_unet_config(), ((16, 16), (8, 12))), ("sd15", W.sd15_unet_config(), ((64, 64),))):
        model, _ = build_ref_model(cfg)
        ms = model.model_sampling
        for (h, w) in shapes:
            x = rnd((2, 4, h, w), 31, 3.0)
            sigma = torch.tensor([2.5, 0.7])
            ctx = rnd((2, 77, cfg["context_dim"]), 32)
            t = ms.timestep(sigma).float()
            xc = ms.calculate_input(sigma, x)
            eps = model.diffusion_model(xc, t, context=ctx, transformer_options={})
            den = model.apply_model(x, sigma, c_crossattn=ctx, transformer_options={})
            save(f"unet_{tag}_{h}x{w}", x=x, sigma=sigma, ctx=ctx, t=t, eps=eps, denoised=den)
        del model


# ------------------------------------------------------------------ 4. sampler trajectories through the reference's own call stack
def g_samplers():
    cfg = W.tiny_unet_config()
    model, patcher = build_ref_model(cfg)
    pos = [[rnd((1, 77, cfg["context_dim"]), 41), {"pooled_output": None}]]
    neg = [[rnd((1, 77, cfg["context_dim"]), 42), {"pooled_output": None}]]
    lat = ref.EmptyLatentImage().generate(128, 96, 1)[0]          # [1,4,12,16]
    out = {"pos": pos[0][0], "neg": neg[0][0]}
    # (a) node-level call, Euler-a / normal, txt2img — common_ksampler LD.py:6657; host-generator noise
    r = ref.common_ksampler(patcher, 1234, 6, 7.5, "euler_ancestral", "normal", pos, neg, lat, denoise=1.0)
    out["euler_a_txt2img"] = r[0]["samples"]
    # (b) hires-style img2img: non-zero latent, denoise 0.45
    lat2 = {"samples": rnd((1, 4, 12, 16), 43, 0.8)}
    r = ref.common_ksampler(patcher, 77, 4, 8.0, "euler_ancestral", "normal", pos, neg, lat2, denoise=0.45)
    out["lat2"] = lat2["samples"]
    out["euler_a_img2img"] = r[0]["samples"]
    # (c) DPM++ 2M = dpmpp_2m_sde with eta=0 (LD.py:1174) / karras, entered one level lower (`sample`, LD.py:3010)
    ms = model.model_sampling
    sig = ref.calculate_sigmas(ms, "karras", 6)
    noise = ref.prepare_noise(lat["samples"], 99)
    cpu = torch.device("cpu")
    # with noise_sampler=None the reference builds a torchsde BrownianTree even when eta=0 (LD.py:1187-1192);
    # torchsde is absent, so inject a sampler that must never be called on the deterministic path
    def never(s, sn):
        raise AssertionError("noise sampler called with eta=0")
    r = ref.sample(patcher, noise, pos, neg, 7.0, cpu, ref.ksampler("dpmpp_2m_sde", {"eta": 0.0, "noise_sampler": never}), sig,
                   patcher.model_options, latent_image=lat["samples"], seed=99)
    out["dpmpp2m_eta0"] = r
    # (d) DPM++ 2M SDE eta=1 with an injected, seeded noise sampler
    def mk_ns():
        g = torch.Generator().manual_seed(5)
        return lambda s, sn: torch.randn(lat["samples"].shape, generator=g)
    r = ref.sample(patcher, noise, pos, neg, 7.0, cpu, ref.ksampler("dpmpp_2m_sde", {"eta": 1.0, "noise_sampler": mk_ns()}),
                   sig, patcher.model_options, latent_image=lat["samples"], seed=99)
    out["dpmpp2m_sde_injected"] = r
    # (e) the wrapper hook contract (LD.py:2558-2567): record what the wrapper receives on one call
    seen = {}
    def hook(apply_model, params):
        if not seen:
            seen.update(input=params["input"].clone(), timestep=params["timestep"].clone(),
                        ctx=params["c"]["c_crossattn"].clone(), cond_or_uncond=np.array(params["cond_or_uncond"]))
        return apply_model(params["input"], params["timestep"], **params["c"])
    p2 = patcher.clone()
    p2.set_model_unet_function_wrapper(hook)
    r2 = ref.common_ksampler(p2, 1234, 6, 7.5, "euler_ancestral", "normal", pos, neg, lat, denoise=1.0)
    assert torch.equal(r2[0]["samples"], out["euler_a_txt2img"])
    out.update({"hook_" + k: v for k, v in seen.items()})
    # (f) toy-denoiser trajectories straight through the k-diffusion functions
    toy = lambda x, s, **kw: x * (1.0 / (1.0 + s.view(-1, 1, 1, 1) ** 2))
    x0 = rnd((2, 4, 4, 4), 44, 14.0)
    torch.manual_seed(7)
    out["toy_euler_a"] = ref.sample_euler_ancestral(toy, x0, ref.get_sigmas_karras(8, 0.03, 14.6), extra_args={})
    out["toy_dpmpp2m"] = ref.sample_dpmpp_2m_sde(toy, x0, ref.get_sigmas_karras(8, 0.03, 14.6), extra_args={}, eta=0.0,
                                                 noise_sampler=never)
    out["toy_dpmpp2m_heun"] = ref.sample_dpmpp_2m_sde(toy, x0, ref.get_sigmas_karras(8, 0.03, 14.6), extra_args={}, eta=0.0,
                                                      solver_type="heun", noise_sampler=never)
    out["toy_x0"] = x0
    xa, info = ref.sample_dpm_adaptive(toy, x0, 0.03, 14.6, extra_args={}, return_info=True)
    out["toy_dpm_adaptive"] = xa
    out["toy_dpm_adaptive_info"] = np.array([info["steps"], info["nfe"], info["n_accept"], info["n_reject"]])
    # (g) dpm_adaptive (the GUI default, LD.py:10572-10576) through the reference's ksampler on the tiny UNet
    r = ref.sample(patcher, noise, pos, neg, 7.0, cpu, ref.ksampler("dpm_adaptive"), sig, patcher.model_options,
                   latent_image=lat["samples"], seed=99)
    out["dpm_adaptive_tiny"] = r
    save("samplers", **out)


# ------------------------------------------------------------------ 5. VAE decoder
def g_vae():
    for tag, cfg, hw in (("tiny", W.tiny_vae_config(), (8, 6)), ("sd15", W.sd15_vae_config(), (32, 32))):
        dec = ref.Decoder(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=cfg["ch"],
                          ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"], attn_resolutions=[], dropout=0.0)
        eng = ref.AutoencodingEngine(None, dec, None)
        sd = eng.state_dict()
        eng.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in sd.items()
                             if not k.startswith("quant_conv")}, strict=False)
        z = rnd((1, 4) + hw, 51, 1.0 / 0.18215 * 0.2)
        img = torch.clamp((eng.decode(z) + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)     # VAE.decode LD.py:6357-6381
        if tag == "sd15":
            save("vae_sd15", z=z, img_sub=img[:, ::4, ::4], mean=img.mean(), std=img.std())
        else:
            save("vae_tiny", z=z, img=img)


# ------------------------------------------------------------------ 5b. VAE encoder (through the reference's own VAE.encode arithmetic)
def g_vae_enc():
    for tag, cfg, hw in (("tiny", W.tiny_vae_config(), (64, 48)), ("sd15", W.sd15_vae_config(), (256, 256))):
        kw = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=cfg["ch"], ch_mult=cfg["ch_mult"],
                  num_res_blocks=cfg["num_res_blocks"], attn_resolutions=[], dropout=0.0)
        eng = ref.AutoencodingEngine(ref.Encoder(**kw), ref.Decoder(**kw), ref.DiagonalGaussianRegularizer(sample=True))
        sd = eng.state_dict()
        eng.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in sd.items()}, strict=True)
        px = torch.rand((1,) + hw + (3,), generator=torch.Generator().manual_seed(57))
        x = px.movedim(-1, 1) * 2.0 - 1.0                       # VAE.process_input, LD.py:6295
        moments = eng.quant_conv(eng.encoder(x))
        torch.manual_seed(58)
        z = eng.encode(x)                                        # regularizer sample on the host generator
        save(f"vae_enc_{tag}", pixels=px, moments=moments, z_seed58=z)


# ------------------------------------------------------------------ 6. CLIP + prompt weights
def g_clip():
    cfg = W.tiny_clip_config()
    m = ref.CLIPTextModel(cfg, torch.float32, None, ref.manual_cast)
    sd = m.state_dict()
    m.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in sd.items() if k != "text_projection.weight"}, strict=False)
    g = torch.Generator().manual_seed(61)
    toks = torch.randint(1000, 40000, (2, 77), generator=g)
    toks[:, 0] = 49406
    toks[0, 9:] = 49407
    toks[1, 30:] = 49407
    x_last, x_inter, _, pooled = m(toks, None, intermediate_output=-2, final_layer_norm_intermediate=True)
    save("clip_tiny", tokens=toks, last=x_last, inter_m2=x_inter, pooled=pooled)
    strs = ["a photo of a cat", "a (red:1.4) car", "((masterpiece)), (best quality:1.2), [x] \\(lit\\)", "(a (b:0.5) c:2)"]
    tw = {s: ref.token_weights(ref.escape_important(s), 1.0) for s in strs}
    with open(os.path.join(OUT, "prompt_weights.json"), "w") as f:
        json.dump(tw, f, indent=1)
    print("wrote prompt_weights.json")


# ------------------------------------------------------------------ 6b. 77-token chunking with an injected word tokenizer
class FakeWordTokenizer:
    """Deterministic stand-in for HF CLIPTokenizer (its vocab files do not travel): word -> 1 + len(word)//4 ids."""

    @classmethod
    def from_pretrained(cls, path):
        return cls()

    @staticmethod
    def ids(word):
        import zlib
        return [100 + zlib.crc32(f"{word}#{i}".encode()) % 40000 for i in range(1 + len(word) // 4)]

    def __call__(self, word):
        return {"input_ids": [49406] + (self.ids(word) if word else []) + [49407]}

    def get_vocab(self):
        return {}


def g_tokens():
    tok = ref.SDTokenizer(tokenizer_path="unused", tokenizer_class=FakeWordTokenizer)
    long_prompt = " ".join(f"word{i}" for i in range(60)) + " " + "x" * 40 + " tail (heavy:1.5) end"
    prompts = ["a photo of a cat", "a (red:1.4) car, ((masterpiece))", long_prompt, "", "multi\nline \\(kept\\) (a (b:0.5) c:2)"]
    out = {p: tok.tokenize_with_weights(p) for p in prompts}
    with open(os.path.join(OUT, "token_chunks.json"), "w") as f:
        json.dump(out, f)
    print("wrote token_chunks.json", {k[:20]: len(v) for k, v in out.items()})


# ------------------------------------------------------------------ 6c. the same with the REAL CLIP tokenizer (vocabulary files of the reference checkout)
REAL_PROMPTS = ["a photo of a cat", "masterpiece, (best quality:1.2), 1girl, ((detailed eyes)), [blurry]", "",
                "a \\(literal\\) paren and (nested (deep:1.5) words:0.8)",
                " ".join(["extraordinarily"] * 45) + " long prompt that spills into a second chunk of seventy seven tokens, (weighted tail:1.3)"]


def g_tokens_real():
    """SDTokenizer (LD.py:4936-5031) over HuggingFace's CLIPTokenizer with the vocabulary of `_internal/sd1_tokenizer/` (which stays in the
    reference checkout: the fixture holds prompts and token ids only).  tests/test_host_cpu.py compares PromptTokenizer.from_pretrained on the
    same directory when it is present."""
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    from transformers import CLIPTokenizer
    tok = ref.SDTokenizer(tokenizer_path=os.path.join(os.path.dirname(REF_PATH), "_internal", "sd1_tokenizer"), tokenizer_class=CLIPTokenizer)
    out = {p: tok.tokenize_with_weights(p) for p in REAL_PROMPTS}
    with open(os.path.join(OUT, "token_chunks_real.json"), "w") as f:
        json.dump(out, f)
    print("wrote token_chunks_real.json", {k[:20]: len(v) for k, v in out.items()})


# ------------------------------------------------------------------ 7. bislerp
def g_bislerp():
    x = rnd((2, 4, 8, 6), 71)
    x[0, :, 2, 3] = 0.0
    x[1, :, 4, 1] = x[1, :, 4, 2]
    save("bislerp", x=x, y2x=ref.bislerp(x, 12, 16), y_odd=ref.bislerp(x, 9, 11))


# ------------------------------------------------------------------ 8. full-size goldens at the sizes of BASELINE configs #3 / #5
def g_configs():
    """SD1.5 UNet at 128x128 latents (hires-fix, config #5) and the SD1.5 VAE decoder at 64x64 / 128x128 latents
    (512^2 / 1024^2 images).  The UNet tensors are small enough to store whole; images are stored subsampled plus a
    full-resolution crop and the moments."""
    cfg = W.sd15_unet_config()
    model, _ = build_ref_model(cfg)
    ms = model.model_sampling
    x = rnd((2, 4, 128, 128), 81, 1.5)
    sigma = torch.tensor([1.2768, 0.4])                # hires-fix starts at sigma 1.2768 (normal-10 @ denoise 0.45)
    ctx = rnd((2, 77, cfg["context_dim"]), 82)
    eps = model.diffusion_model(ms.calculate_input(sigma, x), ms.timestep(sigma).float(), context=ctx, transformer_options={})
    den = model.apply_model(x, sigma, c_crossattn=ctx, transformer_options={})
    save("unet_sd15_128x128", x=x, sigma=sigma, ctx=ctx, eps=eps, denoised=den)
    del model
    vcfg = W.sd15_vae_config()
    dec = ref.Decoder(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=vcfg["ch"],
                      ch_mult=vcfg["ch_mult"], num_res_blocks=vcfg["num_res_blocks"], attn_resolutions=[], dropout=0.0)
    eng = ref.AutoencodingEngine(None, dec, None)
    sd = eng.state_dict()
    eng.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in sd.items() if not k.startswith("quant_conv")}, strict=False)
    for hw, sub in ((64, 4), (128, 8)):
        z = rnd((1, 4, hw, hw), 83 + hw, 1.0 / 0.18215 * 0.2)
        img = torch.clamp((eng.decode(z) + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)
        c0 = 8 * hw // 2 - 37                           # a crop that straddles the image centre, all pixels
        save(f"vae_sd15_{hw}x{hw}", z=z, img_sub=img[:, ::sub, ::sub], crop=img[:, c0:c0 + 96, c0:c0 + 96], crop_at=np.array([c0, c0]),
             mean=img.mean(), std=img.std())


# ------------------------------------------------------------------ 9. LoRA ingestion through the reference's own key maps + patcher
def lora_fixture():
    """A small kohya-format LoRA for the tiny UNet + tiny CLIP, keyed the ways real files are keyed (LD.py:577-629):
    ldm-flattened names, diffusers-flattened names (what kohya writes for SD1.x), a diffusers-native `unet.` key, a 3x3 conv
    pair, `lora_te_` text-encoder keys; some with `.alpha`, some without; plus one key no model has."""
    ucfg = W.tiny_unet_config()
    mc = ucfg["model_channels"]
    c1 = mc * ucfg["channel_mult"][1]
    pairs = {   # name -> (out, in, conv k, alpha or None)
        "lora_unet_input_blocks_1_1_transformer_blocks_0_attn1_to_q": (mc, mc, 1, 2.0),
        "lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_attn2_to_k": (mc, ucfg["context_dim"], 1, None),
        "lora_unet_down_blocks_1_attentions_0_transformer_blocks_0_ff_net_0_proj": (8 * c1, c1, 1, 4.0),
        "lora_unet_mid_block_attentions_0_proj_in": (None, None, 1, 1.0),        # sizes filled from the model below
        "lora_unet_up_blocks_1_attentions_2_transformer_blocks_0_attn1_to_out_0": (None, None, 1, 3.0),
        "lora_unet_down_blocks_0_resnets_0_conv1": (mc, mc, 3, 8.0),
        "lora_unet_up_blocks_0_resnets_1_time_emb_proj": (None, None, 1, None),
        "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor.to_v": (mc, mc, 1, 1.0),
        "lora_te_text_model_encoder_layers_0_self_attn_q_proj": ("clip", "clip", 1, 2.0),
        "lora_te_text_model_encoder_layers_1_mlp_fc1": ("clip_fc1", "clip", 1, None),
        "lora_unet_not_a_layer_of_this_model": (mc, mc, 1, 1.0),
    }
    return pairs


def g_lora():
    ucfg, ccfg = W.tiny_unet_config(), W.tiny_clip_config()
    model, patcher = build_ref_model(ucfg)
    key_map = ref.model_lora_keys_unet(model, {})
    class ClipWrap(torch.nn.Module):          # state-dict prefix of the reference's SD1ClipModel: clip_l.transformer.* (LD.py:560-575)
        def __init__(self, tm):
            super().__init__()
            self.clip_l = torch.nn.Module()
            self.clip_l.transformer = tm
    tm = ref.CLIPTextModel(ccfg, torch.float32, None, ref.manual_cast)
    csd = tm.state_dict()
    tm.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in csd.items() if k != "text_projection.weight"}, strict=False)
    cw = ClipWrap(tm)
    key_map = ref.model_lora_keys_clip(cw, key_map)
    msd, wsd = model.state_dict(), cw.state_dict()
    rank = 4
    lora, seed = {}, 900
    for name, (o, i, k, alpha) in lora_fixture().items():
        tgt = key_map.get(name)
        if tgt is not None:
            wt = msd[tgt] if tgt in msd else wsd[tgt]
            o, i = wt.shape[0], wt.shape[1]
        up = rnd((o, rank) if k == 1 else (o, rank, 1, 1), seed, 0.3)
        down = rnd((rank, i) if k == 1 else (rank, i, 3, 3), seed + 1, 0.3)
        seed += 2
        lora[name + ".lora_up.weight"], lora[name + ".lora_down.weight"] = up, down
        if alpha is not None:
            lora[name + ".alpha"] = torch.tensor(alpha)
    loaded = ref.load_lora(lora, key_map)
    p2 = patcher.clone()
    ku = p2.add_patches(loaded, 0.8)
    cp = ref.ModelPatcher(cw, torch.device("cpu"), torch.device("cpu"))
    kc = cp.add_patches(loaded, 0.6)
    p2.patch_model()
    cp.patch_model()
    x, sigma, ctx = rnd((2, 4, 16, 16), 931, 3.0), torch.tensor([2.5, 0.7]), rnd((2, 77, ucfg["context_dim"]), 932)
    den = model.apply_model(x, sigma, c_crossattn=ctx, transformer_options={})
    g = torch.Generator().manual_seed(933)
    toks = torch.randint(1000, 40000, (1, 77), generator=g)
    toks[:, 0] = 49406
    toks[0, 12:] = 49407
    _, inter, _, _ = tm(toks, None, intermediate_output=-2, final_layer_norm_intermediate=True)
    out = {"lora::" + k: v for k, v in lora.items()}
    save("lora_tiny", x=x, sigma=sigma, ctx=ctx, denoised=den, tokens=toks, clip_inter_m2=inter,
         patched_unet_keys=np.array(sorted(ku)), patched_clip_keys=np.array(sorted(kc)),
         w_attn1_to_q=model.state_dict()["diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"],
         w_conv1=model.state_dict()["diffusion_model.input_blocks.1.0.in_layers.2.weight"], **out)
    # the diffusers -> ldm name map itself, for the tiny and the SD1.5 layouts (LD.py:302-394)
    maps = {"tiny": ref.unet_to_diffusers(ref_unet_config(ucfg)), "sd15": ref.unet_to_diffusers(ref_unet_config(W.sd15_unet_config()))}
    with open(os.path.join(OUT, "unet_to_diffusers.json"), "w") as f:
        json.dump({k: dict(sorted(v.items())) for k, v in maps.items()}, f)
    print("wrote unet_to_diffusers.json", {k: len(v) for k, v in maps.items()})


# ------------------------------------------------------------------ 10. full-length, full-size end-to-end runs (configs #2 / #3 / #5)
E2E_POS = [(49406, 1.0), (1125, 1.0), (2368, 1.0), (539, 1.3), (320, 1.3), (2242, 1.0), (267, 1.0), (4917, 0.8), (7857, 1.0),
           (3878, 1.0), (267, 1.0), (12609, 1.1), (2870, 1.1)] + [(49407, 1.0)] * 64
E2E_NEG = [(49406, 1.0)] + [(49407, 1.0)] * 76


class _StepRecorder:
    """A `model_function_wrapper` (LD.py:2558-2567) that passes straight through to the reference's apply_model and keeps
    the latent every sampler step starts from (row 0 of the [uncond, cond] batch, every second pixel)."""

    def __init__(self):
        self.xs = []

    def __call__(self, apply_model, params):
        self.xs.append(params["input"][0:1, :, ::2, ::2].clone())
        return apply_model(params["input"], params["timestep"], **params["c"])

    def to(self, device):
        return self


def _ref_decoder():
    vcfg = W.sd15_vae_config()
    dec = ref.Decoder(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=vcfg["ch"],
                      ch_mult=vcfg["ch_mult"], num_res_blocks=vcfg["num_res_blocks"], attn_resolutions=[], dropout=0.0)
    eng = ref.AutoencodingEngine(None, dec, None)
    sd = eng.state_dict()
    eng.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in sd.items() if not k.startswith("quant_conv")}, strict=False)
    return lambda z: torch.clamp((eng.decode(z) + 1.0) / 2.0, 0.0, 1.0).movedim(1, -1)      # VAE.decode LD.py:6357-6381


def _img_fields(tag, img, sub):
    c0 = img.shape[1] // 2 - 37
    return {f"{tag}img_sub": img[:, ::sub, ::sub], f"{tag}crop": img[:, c0:c0 + 96, c0:c0 + 96], f"{tag}crop_at": np.array([c0, c0]),
            f"{tag}img_mean": img.mean(), f"{tag}img_std": img.std(),
            f"{tag}img_saturated": ((img <= 0.0) | (img >= 1.0)).float().mean()}


def g_e2e(cfg_scales=(None,)):
    """The reference's own call stack, fp32 on the CPU, SD1.5-size synthetic net, full step counts:
    token ids -> SDClipModel.encode_token_weights (clip skip -2) -> common_ksampler / sample -> Decoder.
      cfg2: B=1, 20 steps dpmpp_2m_sde eta=0 ("DPM++ 2M") / karras, cfg 7        (BASELINE config #2; #1 is this very run)
      cfg3: B=2, 30 steps euler_ancestral / normal, cfg 7, host-generator noise   (config #3)
      cfg5: cfg2's latent -> bislerp x2 -> 10 euler_ancestral steps at denoise 0.45, cfg 8 -> 1024^2 decode   (config #5)
    `anchor`: cfg2 once more with every UNet weight rounded to fp16 (what the reference's own unet_dtype1 stores,
    LD.py:6418-6423) — the drift of the REFERENCE against itself under fp16 weight storage, the yardstick for the HIP path.
    ≈ 12 min on 8 cores.  `python oracle/make_golden.py e2e` writes the cfg-7/8 files; `e2e_cfg1` the same runs at cfg 1."""
    import time
    t0 = time.time()
    suffix = "" if cfg_scales[0] is None else "_cfg1"
    cpu = torch.device("cpu")
    # ---- conditioning from token ids through the reference's token-weight encoder
    clip = ref.SDClipModel(device="cpu", dtype=torch.float32, layer="last",
                           textmodel_json_config=os.path.join(os.path.dirname(REF_PATH), "_internal", "clip", "sd1_clip_config.json"))
    tsd = clip.transformer.state_dict()
    clip.transformer.load_state_dict({k: W.synth_tensor(k, tuple(v.shape)) for k, v in tsd.items() if k != "text_projection.weight"},
                                     strict=False)
    clip.reset_clip_options()
    clip.set_clip_options({"layer": -2})                       # CLIPSetLastLayer(-2), LD.py:6604-6608 / 10016
    cpos, ppos = clip.encode_token_weights([E2E_POS])
    cneg, pneg = clip.encode_token_weights([E2E_NEG])
    del clip
    if not suffix:
        save("e2e_cond", pos_ids=np.array([t for t, _ in E2E_POS]), pos_w=np.array([w for _, w in E2E_POS], dtype=np.float32),
             neg_ids=np.array([t for t, _ in E2E_NEG]), cond_pos=cpos, cond_neg=cneg, pooled_pos=ppos)
    pos, neg = [[cpos, {"pooled_output": ppos}]], [[cneg, {"pooled_output": pneg}]]
    print(f"[e2e] clip done {time.time() - t0:.0f}s", flush=True)

    model, patcher = build_ref_model(W.sd15_unet_config())
    decode = _ref_decoder()

    def never(s, sn):
        raise AssertionError("noise sampler called with eta=0")

    def run_cfg2(p, scale):
        rec = _StepRecorder()
        p = p.clone()
        p.set_model_unet_function_wrapper(rec)
        lat = ref.EmptyLatentImage().generate(512, 512, 1)[0]["samples"]
        sig = ref.calculate_sigmas(model.model_sampling, "karras", 20)
        noise = ref.prepare_noise(lat, 2002)
        out = ref.sample(p, noise, pos, neg, scale, cpu, ref.ksampler("dpmpp_2m_sde", {"eta": 0.0, "noise_sampler": never}), sig,
                         p.model_options, latent_image=lat, seed=2002)
        return out, torch.cat(rec.xs)

    s7 = 7.0 if cfg_scales[0] is None else cfg_scales[0]
    s8 = 8.0 if cfg_scales[0] is None else cfg_scales[0]
    lat2, traj2 = run_cfg2(patcher, s7)
    img2 = decode(lat2)
    print(f"[e2e] cfg2 done {time.time() - t0:.0f}s  |lat| {float(lat2.abs().mean()):.3f}", flush=True)

    # ---- cfg3
    rec = _StepRecorder()
    p3 = patcher.clone()
    p3.set_model_unet_function_wrapper(rec)
    lat = ref.EmptyLatentImage().generate(512, 512, 2)[0]
    # the reference never repeats a conditioning to the latent batch (repeat_to_batch_size is the identity, LD.py:397-398):
    # a batch of 2 needs batch-2 conditionings
    pos_b = [[cpos.repeat(2, 1, 1), {"pooled_output": ppos}]]
    neg_b = [[cneg.repeat(2, 1, 1), {"pooled_output": pneg}]]
    lat3 = ref.common_ksampler(p3, 3003, 30, s7, "euler_ancestral", "normal", pos_b, neg_b, lat, denoise=1.0)[0]["samples"]
    traj3 = torch.cat(rec.xs)
    img3 = decode(lat3)
    f3 = {}
    f3.update(_img_fields("r0_", img3[0:1], 4))
    f3.update(_img_fields("r1_", img3[1:2], 4))
    save("e2e_cfg3" + suffix, latent=lat3, traj_sub=traj3, cfg=np.array(s7), seed=np.array(3003), **f3)
    print(f"[e2e] cfg3 done {time.time() - t0:.0f}s", flush=True)

    # ---- cfg5 (hires-fix of cfg2's result, LD.py:10585-10603)
    rec = _StepRecorder()
    p5 = patcher.clone()
    p5.set_model_unet_function_wrapper(rec)
    up = ref.LatentUpscale().upscale({"samples": lat2}, "bislerp", 1024, 1024, "disabled")[0]
    lat5 = ref.common_ksampler(p5, 5005, 10, s8, "euler_ancestral", "normal", pos, neg, up, denoise=0.45)[0]["samples"]
    traj5 = torch.cat(rec.xs)
    img5 = decode(lat5)
    save("e2e_cfg5" + suffix, upscaled=up["samples"], latent=lat5, traj_sub=traj5, cfg=np.array(s8), seed=np.array(5005),
         **_img_fields("", img5, 8))
    print(f"[e2e] cfg5 done {time.time() - t0:.0f}s", flush=True)

    # ---- anchor: the reference against itself with fp16-stored UNet weights
    for prm in model.diffusion_model.parameters():
        prm.data = prm.data.half().float()
    lat2h, traj2h = run_cfg2(patcher, s7)
    img2h = decode(lat2h)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    anchor_curve = np.array([rel(traj2h[i], traj2[i]) for i in range(traj2.shape[0])])
    save("e2e_cfg2" + suffix, latent=lat2, traj_sub=traj2, cfg=np.array(s7), seed=np.array(2002),
         anchor_traj_rel=anchor_curve, anchor_latent_rel=np.array(rel(lat2h, lat2)),
         anchor_img_maxabs=(img2h - img2).abs().max(), anchor_img_meanabs=(img2h - img2).abs().mean(),
         **_img_fields("", img2, 4))
    print(f"[e2e] anchor done {time.time() - t0:.0f}s  latent rel {rel(lat2h, lat2):.3e}  img max {float((img2h - img2).abs().max()) * 255:.2f}/255 "
          f"mean {float((img2h - img2).abs().mean()) * 255:.3f}/255", flush=True)
    print("[e2e] anchor curve", np.array2string(anchor_curve, precision=2), flush=True)


def g_e2e_cfg1():
    g_e2e((1.0,))


# ------------------------------------------------------------------ 11. several conditioning entries per list (calc_cond_batch averaging, LD.py:2492-2591)
def g_multicond():
    """Two positive entries (the second carrying the keys the reference's stripped get_area_and_mult ignores: area / strength)
    and two negative ones of different token counts (77 / 154: CONDCrossAttn.concat pads by repetition), Euler-a on the tiny UNet."""
    cfg = W.tiny_unet_config()
    model, patcher = build_ref_model(cfg)
    d = cfg["context_dim"]
    pos = [[rnd((2, 77, d), 141), {"pooled_output": None}],           # batch-sized: the reference never repeats a conditioning (LD.py:397-398)
           [rnd((2, 77, d), 142), {"pooled_output": None, "area": (4, 4, 0, 0), "strength": 0.3}]]
    neg = [[rnd((2, 77, d), 143), {"pooled_output": None}], [rnd((2, 154, d), 144), {"pooled_output": None}]]
    lat = ref.EmptyLatentImage().generate(128, 96, 2)[0]
    seen = {}
    def hook(apply_model, params):
        if not seen:
            seen.update(cond_or_uncond=np.array(params["cond_or_uncond"]), ctx=params["c"]["c_crossattn"].clone())
        return apply_model(params["input"], params["timestep"], **params["c"])
    p2 = patcher.clone()
    p2.set_model_unet_function_wrapper(hook)
    r = ref.common_ksampler(p2, 4321, 5, 6.0, "euler_ancestral", "normal", pos, neg, lat, denoise=1.0)
    save("multicond", pos0=pos[0][0], pos1=pos[1][0], neg0=neg[0][0], neg1=neg[1][0], out=r[0]["samples"],
         hook_cond_or_uncond=seen["cond_or_uncond"], hook_ctx=seen["ctx"])


if __name__ == "__main__":
    # (`e2e` takes ~8 min, `tokens_real` needs the reference checkout's tokenizer files and transformers: both on request)
    which = sys.argv[1:] or ["schedules", "blocks", "unets", "samplers", "vae", "vae_enc", "clip", "tokens", "bislerp", "configs", "lora", "multicond"]
    for n in which:
        globals()["g_" + n]()

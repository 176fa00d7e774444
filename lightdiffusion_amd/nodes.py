"""The reference's workflow-node call surface (LD.py:6573-6725, call order of `pipeline()` LD.py:10001-10086),
backed by the MI355X hot path.  A user of the reference keeps writing

    model, clip, vae = CheckpointLoaderSimple().load_checkpoint(path)
    pos = CLIPTextEncode().encode(clip, "a cat")[0];  neg = CLIPTextEncode().encode(clip, "")[0]
    lat = EmptyLatentImage().generate(512, 512, 1)[0]
    lat = KSampler2().sample(model, seed, 20, 7.0, "dpmpp_2m_sde", "karras", pos, neg, lat)[0]
    img = VAEDecode().decode(vae, lat)[0]

and gets the HIP UNet / VAE underneath.  The UNet is attached through the reference's own plugin seam:
`ModelPatcher.set_model_unet_function_wrapper` (LD.py:3277).
"""
from __future__ import annotations

import copy
from typing import Optional

import torch

from . import sampling
from . import weights as W
from .clip import CLIP, CLIPTextModel, CLIPTextModelHIP, PromptTokenizer
from .sampling import LATENT_SCALE, common_ksampler
from .unet import MI355XUNet, MI355XVAE


class SD15LatentFormat:
    """LatentFormat / SD15 (LD.py:125-147)."""
    scale_factor = LATENT_SCALE

    def process_in(self, latent):
        return latent * self.scale_factor

    def process_out(self, latent):
        return latent / self.scale_factor


class SD15Model:
    """The slice of BaseModel (LD.py:5798-5897) the sampling stack touches."""

    def __init__(self, unet: MI355XUNet):
        self.diffusion_model = unet
        self.model_sampling = sampling.ModelSampling()
        self.latent_format = SD15LatentFormat()

    def apply_model(self, x, t, c_crossattn=None, transformer_options=None, **kwargs):
        """BaseModel.apply_model (LD.py:5828-5860) — routed through the same wrapper object the hook would call."""
        return self.diffusion_model(None, {"input": x, "timestep": t, "c": {"c_crossattn": c_crossattn}, "cond_or_uncond": [1, 0]})

    def process_latent_in(self, latent):
        return self.latent_format.process_in(latent)

    def process_latent_out(self, latent):
        return self.latent_format.process_out(latent)


class ModelPatcher:
    """ModelPatcher (LD.py:3210-3437), reduced to what the hot path uses: model_options + the UNet wrapper hook."""

    def __init__(self, model: SD15Model, load_device, offload_device=None):
        self.model = model
        self.load_device = torch.device(load_device)
        self.offload_device = offload_device
        self.model_options = {"transformer_options": {}}

    def clone(self):
        n = ModelPatcher(self.model, self.load_device, self.offload_device)
        n.model_options = copy.copy(self.model_options)
        return n

    def set_model_unet_function_wrapper(self, unet_wrapper_function):
        self.model_options["model_function_wrapper"] = unet_wrapper_function

    def get_model_object(self, name):
        return getattr(self.model, name)

    def model_patches_to(self, device):
        w = self.model_options.get("model_function_wrapper")
        if w is not None and hasattr(w, "to"):
            self.model_options["model_function_wrapper"] = w.to(device)


# ------------------------------------------------------------------ nodes
class EmptyLatentImage:
    def generate(self, width, height, batch_size=1):
        return ({"samples": torch.zeros([batch_size, 4, height // 8, width // 8])},)


class CLIPTextEncode:
    def encode(self, clip: CLIP, text: str):
        cond, pooled = clip.encode_from_tokens(clip.tokenize(text), return_pooled=True)
        return ([[cond, {"pooled_output": pooled}]],)


class CLIPSetLastLayer:
    def set_last_layer(self, clip: CLIP, stop_at_clip_layer: int):
        clip = clip.clone()
        clip.clip_layer(stop_at_clip_layer)
        return (clip,)


class KSampler2:
    def sample(self, model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0):
        return common_ksampler(model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=denoise)


class VAEDecode:
    def decode(self, vae: MI355XVAE, samples):
        return (vae.decode(samples["samples"]),)


class VAEEncode:
    def encode(self, vae: MI355XVAE, pixels):
        return ({"samples": vae.encode(pixels[:, :, :, :3])},)


def bislerp(samples: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """Latent upscale of the hires-fix path (bislerp, LD.py:429-518): 2-tap separable resize along W then H whose blend of
    the two C-vectors slerps the direction and lerps the magnitude.  Runs once per image (torch ops on `samples.device`)."""
    def taps(n_src, n_dst, dev):
        pos = ((torch.arange(n_dst, dtype=torch.float32, device=dev) + 0.5) * (n_src / n_dst) - 0.5).clamp_(min=0.0)
        lo = pos.floor().clamp_(max=n_src - 1)
        fr = torch.where(lo >= n_src - 1, torch.zeros_like(pos), pos - lo)
        lo = lo.long()
        return lo, (lo + 1).clamp_(max=n_src - 1), fr

    def blend(a, b, r):
        na, nb = a.norm(dim=-1, keepdim=True), b.norm(dim=-1, keepdim=True)
        ua = torch.where(na > 0, a / na, torch.zeros_like(a))
        ub = torch.where(nb > 0, b / nb, torch.zeros_like(b))
        cw = (ua * ub).sum(dim=-1, keepdim=True)
        om = torch.acos(cw)
        so = torch.sin(om)
        res = (torch.sin((1.0 - r) * om) / so) * ua + (torch.sin(r * om) / so) * ub
        res = res * (na * (1.0 - r) + nb * r)
        res = torch.where(cw > 1 - 1e-5, a, res)
        return torch.where(cw < 1e-5 - 1, a * (1.0 - r) + b * r, res)

    x = samples.float().permute(0, 2, 3, 1)
    lo, hi, fr = taps(x.shape[2], width, x.device)
    x = blend(x[:, :, lo], x[:, :, hi], fr.view(1, 1, -1, 1))
    lo, hi, fr = taps(x.shape[1], height, x.device)
    x = blend(x[:, lo], x[:, hi], fr.view(1, -1, 1, 1))
    return x.permute(0, 3, 1, 2).to(samples.dtype)


class LatentUpscale:
    upscale_methods = ["nearest-exact", "bilinear", "area", "bicubic", "bislerp"]

    def upscale(self, samples, upscale_method, width, height, crop="disabled"):
        if width == 0 and height == 0:
            return (samples,)
        s = samples.copy()
        s["samples"] = bislerp(samples["samples"], max(64, width) // 8, max(64, height) // 8)   # the only mode (LD.py:521-523)
        return (s,)


# ------------------------------------------------------------------ loaders
def _attach(unet: MI355XUNet, device) -> ModelPatcher:
    patcher = ModelPatcher(SD15Model(unet), load_device=device)
    patcher.set_model_unet_function_wrapper(unet)
    return patcher


def load_synthetic(device="cuda:0", max_batch: int = 1, max_hw=(64, 64), tiny: bool = False, seed: int = 0, tokenizer_dir: Optional[str] = None):
    """(model, clip, vae) with deterministic random-init weights — the offline stand-in for a downloaded checkpoint."""
    ucfg, vcfg, ccfg = (W.tiny_unet_config(), W.tiny_vae_config(), W.tiny_clip_config()) if tiny else \
        (W.sd15_unet_config(), W.sd15_vae_config(), W.sd15_clip_config())
    if tiny:
        ccfg = dict(ccfg, hidden_size=ucfg["context_dim"])
    gen = lambda name, shape: W.synth_tensor(name, shape, seed)
    unet = MI355XUNet(ucfg, gen, device=device, max_batch=2 * max_batch, max_hw=max_hw)
    vae = MI355XVAE(vcfg, gen, device=device, max_batch=max_batch, max_hw=max_hw, with_encoder=True)
    tok = PromptTokenizer.from_pretrained(tokenizer_dir) if tokenizer_dir else None
    clip = CLIP(CLIPTextModelHIP(ccfg, W.synth_state_dict(W.clip_param_shapes(ccfg), seed), device=device), tok)
    return _attach(unet, device), clip, vae


class CheckpointLoaderSimple:
    """load_checkpoint(ckpt) -> (model, clip, vae) from a single-file SD1.x checkpoint (LD.py:6426-6513, 6591-6601): the
    architecture is detected from the keys (checkpoint.detect_*), weights are repacked on the device by the C library."""

    def __init__(self, device="cuda:0", max_batch: int = 1, max_hw=(64, 64), tokenizer_dir: Optional[str] = None,
                 clip_heads: int = 12, unet_heads: int = 8):
        self.device, self.max_batch, self.max_hw, self.tokenizer_dir = device, max_batch, max_hw, tokenizer_dir
        self.clip_heads, self.unet_heads = clip_heads, unet_heads

    def load_checkpoint(self, ckpt_name, output_vae=True, output_clip=True, lora: Optional[dict] = None, lora_strength: float = 1.0):
        from . import checkpoint as CK
        sd = CK.load_state_dict(ckpt_name) if isinstance(ckpt_name, str) else dict(ckpt_name)
        if lora:
            CK.merge_lora(sd, lora, lora_strength)
        unet = MI355XUNet(CK.detect_unet_config(sd, num_heads=self.unet_heads), sd, device=self.device, max_batch=2 * self.max_batch,
                          max_hw=self.max_hw)
        vcfg, has_enc = CK.detect_vae_config(sd)
        vae = MI355XVAE(vcfg, sd, device=self.device, max_batch=self.max_batch, max_hw=self.max_hw, with_encoder=has_enc)
        csd = CK.clip_state_dict(sd)
        tok = PromptTokenizer.from_pretrained(self.tokenizer_dir) if self.tokenizer_dir else None
        clip = CLIP(CLIPTextModelHIP(CK.detect_clip_config(csd, self.clip_heads), csd, device=self.device), tok)
        return _attach(unet, self.device), clip, vae


def txt2img(model, clip, vae, prompt_tokens, negative_tokens, width=512, height=512, batch_size=1, seed=0, steps=20, cfg=7.0,
            sampler_name="dpmpp_2m_sde", scheduler="karras", hires: bool = False):
    """The reference's headless `pipeline()` order (LD.py:10001-10086) with explicit arguments instead of hard-coded ones.
    `*_tokens`: a prompt string (needs a tokenizer on `clip`) or pre-tokenised [[(id, weight), ...]] chunks."""
    enc = lambda t: clip.encode_from_tokens(clip.tokenize(t) if isinstance(t, str) else t, return_pooled=True)
    (pc, pp), (nc, npool) = enc(prompt_tokens), enc(negative_tokens)
    pos, neg = [[pc, {"pooled_output": pp}]], [[nc, {"pooled_output": npool}]]
    lat = EmptyLatentImage().generate(width, height, batch_size)[0]
    lat = KSampler2().sample(model, seed, steps, cfg, sampler_name, scheduler, pos, neg, lat)[0]
    if hires:   # hires-fix (LD.py:10585-10603): bislerp x2, then 10 Euler-a steps at denoise 0.45, cfg 8
        lat = LatentUpscale().upscale(lat, "bislerp", width * 2, height * 2)[0]
        lat = KSampler2().sample(model, seed, 10, 8, "euler_ancestral", "normal", pos, neg, lat, denoise=0.45)[0]
    return VAEDecode().decode(vae, lat)[0]

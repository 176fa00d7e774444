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
from .clip import CLIP, CLIPTextModelHIP, PromptTokenizer
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


def bislerp(samples: torch.Tensor, width: int, height: int, device=None) -> torch.Tensor:
    """Latent upscale of the hires-fix path (bislerp, LD.py:429-518) on the HIP kernel `ld_op_bislerp`: a 2-tap separable
    resize along W then H whose blend of the two C-vectors slerps the direction and lerps the magnitude.  A host tensor
    (what KSampler2 returns, LD.py:3156-3203) is uploaded, resized on the device and returned to the host."""
    from . import ops
    if samples.is_cuda:
        return ops.bislerp(samples, width, height)
    # one rank per GPU: the default is THIS process's current device, never a hard-coded cuda:0
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    return ops.bislerp(samples.to(dev), width, height).to(samples.device)


class LatentUpscale:
    upscale_methods = ["nearest-exact", "bilinear", "area", "bicubic", "bislerp"]

    def __init__(self, device=None):
        self.device = device          # the model's load_device in a one-rank-per-GPU job (None: the current device)

    def upscale(self, samples, upscale_method, width, height, crop="disabled"):
        if width == 0 and height == 0:
            return (samples,)
        s = samples.copy()
        s["samples"] = bislerp(samples["samples"], max(64, width) // 8, max(64, height) // 8, device=self.device)   # the only mode (LD.py:521-523)
        return (s,)


# ------------------------------------------------------------------ loaders
def _attach(unet: MI355XUNet, device) -> ModelPatcher:
    patcher = ModelPatcher(SD15Model(unet), load_device=device)
    patcher.set_model_unet_function_wrapper(unet)
    return patcher


def load_synthetic(device="cuda:0", max_batch: int = 1, max_hw=(64, 64), tiny: bool = False, seed: int = 0, tokenizer_dir: Optional[str] = None,
                   max_tokens: int = 77):
    """(model, clip, vae) with deterministic random-init weights — the offline stand-in for a downloaded checkpoint.
    `max_batch` / `max_hw` / `max_tokens` only pre-size the workspaces: larger batches, latents (hires pass) or prompts
    (more 77-token chunks) grow them on first use (MI355XUNet._ensure, MI355XVAE._ensure)."""
    ucfg, vcfg, ccfg = (W.tiny_unet_config(), W.tiny_vae_config(), W.tiny_clip_config()) if tiny else \
        (W.sd15_unet_config(), W.sd15_vae_config(), W.sd15_clip_config())
    if tiny:
        ccfg = dict(ccfg, hidden_size=ucfg["context_dim"])
    gen = lambda name, shape: W.synth_tensor(name, shape, seed)
    unet = MI355XUNet(ucfg, gen, device=device, max_batch=2 * max_batch, max_hw=max_hw, max_tokens=max_tokens)
    vae = MI355XVAE(vcfg, gen, device=device, max_batch=max_batch, max_hw=max_hw, with_encoder=True)
    tok = PromptTokenizer.from_pretrained(tokenizer_dir) if tokenizer_dir else None
    clip = CLIP(CLIPTextModelHIP(ccfg, W.synth_state_dict(W.clip_param_shapes(ccfg), seed), device=device), tok)
    return _attach(unet, device), clip, vae


class CheckpointLoaderSimple:
    """load_checkpoint(ckpt) -> (model, clip, vae) from a single-file SD1.x checkpoint (LD.py:6426-6513, 6591-6601): the
    architecture is detected from the keys (checkpoint.detect_*), weights are repacked on the device by the C library."""

    def __init__(self, device="cuda:0", max_batch: int = 1, max_hw=(64, 64), tokenizer_dir: Optional[str] = None,
                 clip_heads: int = 12, unet_heads: int = 8, max_tokens: int = 77):
        self.device, self.max_batch, self.max_hw, self.tokenizer_dir = device, max_batch, max_hw, tokenizer_dir
        self.clip_heads, self.unet_heads, self.max_tokens = clip_heads, unet_heads, max_tokens

    def load_checkpoint(self, ckpt_name, output_vae=True, output_clip=True, lora=None, lora_strength: float = 1.0,
                        lora_strength_clip: Optional[float] = None):
        """`lora`: a LoRA state dict or a path to one; merged into UNet AND text-encoder weights before they are uploaded
        (load_lora_for_models, LD.py:6203-6219: strength_model / strength_clip)."""
        from . import checkpoint as CK
        sd = CK.load_state_dict(ckpt_name) if isinstance(ckpt_name, str) else dict(ckpt_name)
        if lora is not None:
            CK.merge_lora(sd, CK.load_state_dict(lora) if isinstance(lora, str) else lora, lora_strength, lora_strength_clip)
        unet = MI355XUNet(CK.detect_unet_config(sd, num_heads=self.unet_heads), sd, device=self.device, max_batch=2 * self.max_batch,
                          max_hw=self.max_hw, max_tokens=self.max_tokens)
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
        lat = LatentUpscale(model.load_device).upscale(lat, "bislerp", width * 2, height * 2)[0]
        lat = KSampler2().sample(model, seed, 10, 8, "euler_ancestral", "normal", pos, neg, lat, denoise=0.45)[0]
    return VAEDecode().decode(vae, lat)[0]


def txt2img_sharded(model, clip, vae, prompt_tokens, negative_tokens, width=512, height=512, global_batch=8, seed=0, steps=20, cfg=7.0,
                    sampler_name="euler_ancestral", scheduler="normal", gather: bool = True, run_sampler=None):
    """Batched generation sharded over the ranks of one node (BASELINE config #4), in the call order of the
    reference's `pipeline()` (LD.py:10001-10086): rank 0 runs CLIP and broadcasts [cond, uncond] once (RCCL over xGMI;
    the only data-path collective), every rank takes a contiguous block of the global batch, draws the FULL-batch noise
    from the single seed on its host generator and keeps its rows (initial noise and every ancestral step), runs its own
    sampler loop on its own UNet replica and decodes its images; the images are gathered on rank 0 (None elsewhere).
    Row for row the result equals the single-process run of the same global batch.
    `run_sampler(noise, latent, pos, neg, sigmas, extra_options) -> latents` replaces the device sampler loop in the
    CPU (gloo) test of this host logic; the product path never passes it."""
    import torch.distributed as tdist
    from . import dist as D
    world = D.world_size()
    rank = tdist.get_rank() if world > 1 else 0
    enc = lambda t: clip.encode_from_tokens(clip.tokenize(t) if isinstance(t, str) else t, return_pooled=False)
    pc = nc = shapes = None
    if rank == 0:
        pc, nc = enc(prompt_tokens), enc(negative_tokens)
        shapes = [tuple(pc.shape), tuple(nc.shape)]
    if world > 1:
        box = [shapes]
        tdist.broadcast_object_list(box, src=0)                 # token counts (77 x chunks): a few bytes of metadata
        on_gpu = tdist.get_backend() == "nccl"
        pc, nc = D.broadcast_conditioning([pc, nc], box[0], src=0, device=None if on_gpu else torch.device("cpu"))
        pc, nc = pc.cpu(), nc.cpu()
    rows = D.shard_rows(global_batch, rank, world)
    b = rows.stop - rows.start
    lat_shape = (global_batch, 4, height // 8, width // 8)
    if b > 0:
        noise = D.full_batch_noise(lat_shape, seed, rows)
        latent = torch.zeros((b,) + lat_shape[1:])
        sigmas = sampling.calculate_sigmas(model.get_model_object("model_sampling"), scheduler, steps)
        dev = model.load_device
        extra = {}
        if sampler_name == "euler_ancestral":     # per-step noise: full-batch draw on the host generator, this rank's rows
            extra["noise_sampler"] = sampling.host_noise_sampler(torch.empty((b,) + lat_shape[1:], device=dev), rows, global_batch)
        pos, neg = [[pc, {"pooled_output": None}]], [[nc, {"pooled_output": None}]]
        if run_sampler is not None:
            samples = run_sampler(noise, latent, pos, neg, sigmas, extra)
        else:
            samples = sampling.sample(model, noise, pos, neg, cfg, dev, sampling.ksampler(sampler_name, extra), sigmas,
                                      model.model_options, latent_image=latent, seed=seed).cpu()
        images = vae.decode(samples)
    else:
        images = torch.zeros((0, height, width, 3))
    if not gather or world == 1:
        return images
    if tdist.get_backend() == "nccl":                          # 6.3 MB per rank as uint8
        g = D.gather_images((images * 255.0).round().to(torch.uint8).to(model.load_device), dst=0)
        return None if g is None else g.cpu().float() / 255.0
    return D.gather_images(images, dst=0)

"""MI355X UNet / VAE objects that sit behind the reference's plugin seams.

* `MI355XUNet` satisfies the `model_function_wrapper` contract (LD.py:2558-2567, installed with
  `ModelPatcher.set_model_unet_function_wrapper`, LD.py:3277): `fn(apply_model, {"input", "timestep", "c",
  "cond_or_uncond"}) -> denoised fp32`, plus `.to(device)` (LD.py:3286-3291) — the same contract the reference's
  stable-fast integration implements (`StableFastPatch`, LD.py:9902-9933).
* `MI355XVAE.decode` mirrors `VAE.decode` (LD.py:6357-6381): [B,4,h,w] -> [B,8h,8w,3] fp32 in [0,1].

Both are thin: they own a C handle, feed it device pointers and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Union

import torch

from . import weights as W
from ._lib import ERR_SHAPE, F16, F32, LDError, UNetConfig, VAEConfig, check, lib

WeightSource = Union[Dict[str, torch.Tensor], Callable[[str, tuple], torch.Tensor]]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _load_params(handle, count_fn, info_fn, load_fn, src: WeightSource, device, prefixes=("",)):
    name, ndim, shape = C.c_char_p(), C.c_int(), (C.c_int64 * 4)()
    for i in range(count_fn(handle)):
        check(info_fn(handle, i, C.byref(name), C.byref(ndim), shape), "param_info")
        key = name.value.decode()
        shp = tuple(int(shape[k]) for k in range(ndim.value))
        if callable(src):
            t = src(key, shp)
        else:
            t = None
            for p in prefixes:
                if p + key in src:
                    t = src[p + key]
                    break
            if t is None:
                raise KeyError(f"checkpoint is missing '{key}'")
        if tuple(t.shape) != shp:
            raise ValueError(f"'{key}': expected shape {shp}, got {tuple(t.shape)}")
        if t.dtype not in (torch.float16, torch.float32):
            t = t.float()
        t = t.to(device).contiguous()
        check(load_fn(handle, key.encode(), t.data_ptr(), F32 if t.dtype == torch.float32 else F16, _stream()), f"load_param({key})")
    torch.cuda.synchronize(device)


def _parse_launches(fn, handle) -> list:
    buf = C.create_string_buffer(1 << 18)
    check(fn(handle, buf, len(buf)), "profile_launches")
    out = []
    for line in buf.value.decode().splitlines():
        what, a, b, c, d, fl, us, kern = line.split("\t")
        out.append((what, (int(a), int(b), int(c), int(d)), float(fl), float(us), kern))
    return out


class MI355XUNet:
    """SD1.x UNet resident on one MI355X.  `cfg` as in `weights.sd15_unet_config()`."""

    def __init__(self, cfg: dict, weights: WeightSource, device="cuda:0", max_batch: int = 2, max_hw=(64, 64), max_tokens: int = 77):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        c = UNetConfig()
        c.in_channels, c.out_channels, c.model_channels = cfg["in_channels"], cfg["out_channels"], cfg["model_channels"]
        c.num_levels = len(cfg["channel_mult"])
        for i, v in enumerate(cfg["channel_mult"]):
            c.channel_mult[i] = v
        for i, v in enumerate(cfg["num_res_blocks"]):
            c.num_res_blocks[i] = v
        for i, v in enumerate(cfg["transformer_depth"]):
            c.transformer_depth[i] = v
        for i, v in enumerate(cfg["transformer_depth_output"]):
            c.transformer_depth_output[i] = v
        c.transformer_depth_middle = cfg["transformer_depth_middle"]
        c.context_dim, c.num_heads = cfg["context_dim"], cfg["num_heads"]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ld_unet_create(C.byref(c), C.byref(self._h)), "ld_unet_create")
            _load_params(self._h, lib().ld_unet_param_count, lib().ld_unet_param_info, lib().ld_unet_load_param, weights,
                         self.device, ("", "model.diffusion_model."))
        self._reserved = (0, 0, 0, 0)
        self.reserve_epoch = 0           # bumped whenever the workspace moves: captured hipGraphs of older epochs are stale
        self.reserve(max_batch, max_hw, max_tokens)
        self._ctx_ref = None             # device copy of the context the resident K / V^T were projected from
        self._ctx_token = None           # per-run token of our own sampling stack (sampling.sampling_function)
        self.ctx_shape = None            # (n, tokens) of the resident context
        self._denoisers = {}
        self.hook_graph = True           # the wrapper hook replays captured hipGraphs (False: eager launches, for A/B tests)
        self._hook = {}                  # (N, h, w) -> pipeline.HookRunner
        self._hook_flags = None          # pinned host ints the device-side guards write (ld_op_hook_check)
        self._hook_event = None
        self._hook_epoch = 0

    def reserve(self, max_batch: int, max_hw=(64, 64), max_tokens: int = 77) -> None:
        """(Re)size the activation workspace.  Growing it frees and reallocates: the context must be set again and every
        captured graph is stale (`reserve_epoch`)."""
        with torch.cuda.device(self.device):
            check(lib().ld_unet_reserve(self._h, max_batch, max_hw[0], max_hw[1], max_tokens), "ld_unet_reserve")
        self._reserved = (max_batch, max_hw[0], max_hw[1], max_tokens)
        self.reserve_epoch += 1
        self._ctx_ref = self._ctx_token = self.ctx_shape = None
        self._denoisers = {}
        self._hook = {}

    def _ensure(self, n: int, h: int, w: int, tokens: int) -> None:
        """Lazy re-reserve (the reference accepts any batch, latent size and number of 77-token chunks): grow the plan when a
        call exceeds it instead of failing with ERR_SHAPE."""
        mn, mh, mw, mt = self._reserved
        if n > mn or h > mh or w > mw or tokens > mt:
            self.reserve(max(n, mn), (max(h, mh), max(w, mw)), max(tokens, mt))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().ld_unet_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- sizes
    @property
    def weight_bytes(self) -> int:
        return lib().ld_unet_weight_bytes(self._h)

    @property
    def workspace_bytes(self) -> int:
        return lib().ld_unet_workspace_bytes(self._h)

    @property
    def last_launches(self) -> int:
        return lib().ld_unet_last_launches(self._h)

    @property
    def last_flops(self) -> float:
        return lib().ld_unet_last_flops(self._h)

    # -- hot path
    def set_context(self, ctx: torch.Tensor) -> None:
        """ctx [N, T, context_dim], batch order exactly as the reference batches it: [uncond..., cond...]."""
        ctx = ctx.to(self.device)
        if ctx.dtype not in (torch.float16, torch.float32):
            ctx = ctx.float()
        ctx = ctx.contiguous()
        mn, mh, mw, _ = self._reserved
        self._ensure(ctx.shape[0], mh, mw, ctx.shape[1])
        with torch.cuda.device(self.device):
            check(lib().ld_unet_set_context(self._h, ctx.data_ptr(), F32 if ctx.dtype == torch.float32 else F16, ctx.shape[0],
                                            ctx.shape[1], _stream()), "ld_unet_set_context")
        self._ctx_ref = self._ctx_token = None
        self.ctx_shape = (ctx.shape[0], ctx.shape[1])

    def forward(self, x: torch.Tensor, sigma: torch.Tensor, out: Optional[torch.Tensor] = None, eps_only: bool = False) -> torch.Tensor:
        """x [N,4,h,w] fp32 (device), sigma [N] fp32 (device) -> denoised [N,4,h,w] fp32 = x - eps*sigma."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert sigma.is_cuda and sigma.dtype == torch.float32 and sigma.is_contiguous()
        if out is None:
            out = torch.empty_like(x)
        n, _, h, w = x.shape
        mn, mh, mw, _ = self._reserved
        if n > mn or h > mh or w > mw:
            raise LDError(ERR_SHAPE, f"ld_unet_forward: input {n}x{h}x{w} exceeds the reserved plan {mn}x{mh}x{mw} (reserve(), then set_context)")
        with torch.cuda.device(self.device):
            check(lib().ld_unet_forward(self._h, x.data_ptr(), sigma.data_ptr(), out.data_ptr(), n, h, w, int(eps_only), _stream()),
                  "ld_unet_forward")
        return out

    def forward_pair(self, x: torch.Tensor, sigma: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The classifier-free-guidance pair of one sampler step (calc_cond_batch's cat([x, x]) against cat([uncond, cond]), LD.py:2515-2547):
        x [B,4,h,w], sigma [B] -> denoised [2B,4,h,w] in the order [uncond.., cond..] of the resident 2B-row context.  The layers in front of
        the first cross-attention are evaluated once for both halves (`ld_unet_forward_pair`)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        assert sigma.is_cuda and sigma.dtype == torch.float32 and sigma.is_contiguous()
        b, c, h, w = x.shape
        if out is None:
            out = torch.empty(2 * b, c, h, w, dtype=torch.float32, device=x.device)
        mn, mh, mw, _ = self._reserved
        if 2 * b > mn or h > mh or w > mw:
            raise LDError(ERR_SHAPE, f"ld_unet_forward_pair: input 2x{b}x{h}x{w} exceeds the reserved plan {mn}x{mh}x{mw} (reserve(), then set_context)")
        with torch.cuda.device(self.device):
            check(lib().ld_unet_forward_pair(self._h, x.data_ptr(), sigma.data_ptr(), out.data_ptr(), b, h, w, _stream()), "ld_unet_forward_pair")
        return out

    KERNEL_CLASSES = ("conv3x3", "gemm", "attention", "groupnorm", "layernorm", "misc")

    def profile(self, x: torch.Tensor, sigma: torch.Tensor) -> dict:
        """One forward with HIP events around every launch; returns {class: (ms, flops, launches)}."""
        out = torch.empty_like(x)
        ms, fl, nl = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_int * 6)()
        n, _, h, w = x.shape
        with torch.cuda.device(self.device):
            check(lib().ld_unet_profile(self._h, x.data_ptr(), sigma.data_ptr(), out.data_ptr(), n, h, w, _stream(), ms, fl, nl),
                  "ld_unet_profile")
        return {k: (ms[i], fl[i], nl[i]) for i, k in enumerate(self.KERNEL_CLASSES)}

    def profile_pair(self, x: torch.Tensor, sigma: torch.Tensor) -> dict:
        """`profile` of the CFG-pair route (`forward_pair`): x [B,..], sigma [B]."""
        b, c, h, w = x.shape
        out = torch.empty(2 * b, c, h, w, dtype=torch.float32, device=x.device)
        ms, fl, nl = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_int * 6)()
        with torch.cuda.device(self.device):
            check(lib().ld_unet_profile_pair(self._h, x.data_ptr(), sigma.data_ptr(), out.data_ptr(), b, h, w, _stream(), ms, fl, nl),
                  "ld_unet_profile_pair")
        return {k: (ms[i], fl[i], nl[i]) for i, k in enumerate(self.KERNEL_CLASSES)}

    def profile_kernels(self) -> dict:
        """Per kernel instantiation of the last `profile()` call: {name: (ms, flops, launches)}."""
        buf = C.create_string_buffer(1 << 16)
        check(lib().ld_unet_profile_kernels(self._h, buf, len(buf)), "ld_unet_profile_kernels")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, fl = line.split("\t")
            out[name] = (float(ms), float(fl), int(n))
        return out

    def profile_launches(self) -> list:
        """Per launch of the last `profile()` call: [(what, (a, b, c, d), flops, microseconds, kernel)] in launch order."""
        return _parse_launches(lib().ld_unet_profile_launches, self._h)

    # -- the reference's plugin seam
    def _sync_context(self, ctx: torch.Tensor, n: int, h: int, w: int, token=None) -> None:
        """Make the resident cross-attention K / V^T those of `ctx`.  The reference re-concatenates the context every step
        (cond_cat, LD.py:2471-2489) into a fresh tensor whose address the caching allocator recycles, so identity proves
        nothing: the CONTENT is compared with the device copy of what was projected (one small elementwise kernel + a host
        read per call).  Our own sampling stack passes a never-reused per-run `token` instead and skips the comparison."""
        self._ensure(n, h, w, ctx.shape[1])
        if token is not None and token == self._ctx_token and self.ctx_shape == (ctx.shape[0], ctx.shape[1]):
            return
        c = ctx.to(self.device)
        same = self._ctx_ref is not None and self._ctx_ref.shape == c.shape and self._ctx_ref.dtype == c.dtype and torch.equal(c, self._ctx_ref)
        if not same:
            self.set_context(c)
            with torch.inference_mode(False):          # (a plain tensor: the wrapper hook may update it in place later, in either mode)
                self._ctx_ref = c.clone()
        self._ctx_token = token

    def __call__(self, apply_model, params: dict) -> torch.Tensor:
        """The `model_function_wrapper` contract (LD.py:2558-2567): returns the denoised batch, fp32, in the caller's batch order.

        Every call replays a captured hipGraph on static buffers (`pipeline.HookRunner`; the reference's plugin on this seam has the
        same option, `enable_cuda_graph`, LD.py:9896-9933).  When the caller batched one latent against [uncond, cond]
        (`cond_or_uncond == [1, 0]`: what `calc_cond_batch` builds, LD.py:2515-2547) the CFG-pair graph is replayed — SPECULATIVELY:
        whether the two halves of `input` / `timestep` really are the same tensors, and whether `c_crossattn` still holds the
        conditioning the resident cross-attention K / V^T were projected from, is checked on the device (`ld_op_hook_check`) into
        pinned host flags that are read only after the replay has been queued.  A failed guess (a new prompt, halves that differ)
        re-projects the context and / or replays the plain graph before the result is handed back, so the GPU never idles for the check
        and the result is always that of the plain forward on these inputs (up to the pair route's tile rounding)."""
        x = params["input"].to(self.device, torch.float32).contiguous()
        sigma = params["timestep"].to(self.device, torch.float32).contiguous()
        ctx = params["c"]["c_crossattn"]
        topt = params["c"].get("transformer_options") or {}
        token = topt.get("ld_ctx_token")
        if token is not None or not self.hook_graph:
            # our own sampling stack's un-fused route (`ld_eager_unbatched`): per-run token instead of a content check, eager launches
            self._sync_context(ctx, x.shape[0], x.shape[2], x.shape[3], token)
            return self.forward(x, sigma)
        return self._hook_replay(x, sigma, ctx, params.get("cond_or_uncond"))

    def _hook_replay(self, x: torch.Tensor, sigma: torch.Tensor, ctx: torch.Tensor, cond_or_uncond) -> torch.Tensor:
        from .pipeline import HookRunner
        n, _, h, w = x.shape
        c = ctx.to(self.device)
        if c.dtype not in (torch.float16, torch.float32):
            c = c.float()
        c = c.contiguous()
        self._ensure(n, h, w, c.shape[1])
        known = self._ctx_ref is not None and self._ctx_ref.shape == c.shape and self._ctx_ref.dtype == c.dtype
        if not known:                                  # first call, or another number of tokens / rows: nothing to speculate on
            self.set_context(c)
            with torch.inference_mode(False):          # (a plain tensor: later calls update it in place, inside or outside inference mode)
                self._ctx_ref = c.clone()
        key = (n, h, w)
        run = self._hook.get(key)
        if run is None:
            if len(self._hook) >= 4:
                self._hook.pop(next(iter(self._hook)))
            run = self._hook[key] = HookRunner(self, n, h, w)
        if self._hook_flags is None:
            with torch.inference_mode(False):
                self._hook_flags = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._hook_event = torch.cuda.Event()
        pair = run.pair is not None and cond_or_uncond is not None and list(cond_or_uncond) == [1, 0] and not run.halves_differed
        self._hook_epoch += 1
        run.x.copy_(x)
        run.sigma.copy_(sigma)
        half = n // 2
        with torch.cuda.device(self.device):
            check(lib().ld_op_hook_check(c.data_ptr() if known else None, self._ctx_ref.data_ptr() if known else None,
                                         c.numel() * c.element_size() // 4 if known else 0,
                                         run.x.data_ptr() if pair else None, run.x.numel() // 2 if pair else 0,
                                         run.sigma.data_ptr() if pair else None, half if pair else 0,
                                         self._hook_flags.data_ptr(), self._hook_epoch, _stream()), "ld_op_hook_check")
        self._hook_event.record()
        (run.pair if pair else run.plain).launch()     # speculative: queued before the flags are looked at
        self._hook_event.synchronize()                 # waits for the check kernel only; the replay behind it keeps the GPU busy
        ctx_changed = known and int(self._hook_flags[0]) == self._hook_epoch
        halves_differ = pair and int(self._hook_flags[1]) == self._hook_epoch
        if ctx_changed:
            ref = self._ctx_ref
            self.set_context(c)                        # (clears the reference: it no longer describes the resident projections)
            ref.copy_(c)
            self._ctx_ref = ref
        if halves_differ:
            run.halves_differed = True                 # this caller does not batch one latent twice: stay on the plain graph
        if ctx_changed or halves_differ:
            (run.plain if halves_differ or not pair else run.pair).launch()
        return run.out.clone()

    def cfg_denoise(self, x: torch.Tensor, timestep: torch.Tensor, ctx: torch.Tensor, cond_scale: float, token=None,
                    use_graph: bool = True) -> torch.Tensor:
        """One classifier-free-guided step (sampling_function, LD.py:2609-2626) through a cached, hipGraph-captured
        `pipeline.CFGDenoiser`: ctx is the [uncond.., cond..] batch (2B rows), x [B,4,h,w], timestep [B] (sigma) on the device.
        This is what `KSampler2.sample` reaches through `sampling.sampling_function`; the reference's counterpart is the
        CUDA-graph option of its stable-fast patch (LD.py:9902-9946)."""
        from .pipeline import CFGDenoiser
        b, _, h, w = x.shape
        x = x.to(self.device, torch.float32)
        self._sync_context(ctx, 2 * b, h, w, token)
        key = (b, h, w, float(cond_scale), bool(use_graph))
        d = self._denoisers.get(key)
        if d is None:
            if len(self._denoisers) >= 4:                      # a few shapes per session (txt2img + hires pass); drop the oldest
                self._denoisers.pop(next(iter(self._denoisers)))
            d = self._denoisers[key] = CFGDenoiser(self, b, h, w, cond_scale, use_graph=use_graph)
        return d.run(x, timestep.to(self.device, torch.float32)).clone()   # samplers keep `denoised` across steps (old_denoised)

    def to(self, device):
        """LD.py:3286-3291 calls this on load / unload.  The weights stay resident (they are not torch tensors); like the reference's
        graph-mode plugin (StableFastPatch.to, LD.py:9921-9933) an unload to the CPU drops the captured graphs and their static buffers."""
        if torch.device(device).type == "cpu":
            self._denoisers = {}
            self._hook = {}
        return self


class MI355XVAE:
    """KL-VAE decoder resident on one MI355X.  `cfg` as in `weights.sd15_vae_config()`."""

    downscale_ratio = 8
    latent_channels = 4

    def __init__(self, cfg: dict, weights: WeightSource, device="cuda:0", max_batch: int = 1, max_hw=(64, 64), with_encoder: bool = False):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.with_encoder = bool(with_encoder)
        c = VAEConfig()
        c.with_encoder = int(self.with_encoder)
        c.z_channels, c.ch, c.num_levels = cfg["z_channels"], cfg["ch"], len(cfg["ch_mult"])
        for i, v in enumerate(cfg["ch_mult"]):
            c.ch_mult[i] = v
        c.num_res_blocks, c.out_ch = cfg["num_res_blocks"], cfg["out_ch"]
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().ld_vae_create(C.byref(c), C.byref(self._h)), "ld_vae_create")
            _load_params(self._h, lib().ld_vae_param_count, lib().ld_vae_param_info, lib().ld_vae_load_param, weights,
                         self.device, ("", "first_stage_model."))
        self._reserved = (0, 0, 0)
        self._ensure(max_batch, max_hw[0], max_hw[1])

    def _ensure(self, b: int, h: int, w: int) -> None:
        """Grow the workspace when a call exceeds the plan (batch or latent size), e.g. the 1024^2 decode of a hires pass."""
        mb, mh, mw = self._reserved
        if b > mb or h > mh or w > mw:
            self._reserved = (max(b, mb), max(h, mh), max(w, mw))
            with torch.cuda.device(self.device):
                check(lib().ld_vae_reserve(self._h, *self._reserved), "ld_vae_reserve")

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().ld_vae_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return lib().ld_vae_workspace_bytes(self._h)

    @property
    def last_flops(self) -> float:
        return lib().ld_vae_last_flops(self._h)

    @property
    def last_launches(self) -> int:
        return lib().ld_vae_last_launches(self._h)

    memory_fraction = 0.9          # share of the free device memory a decode's workspace may take
    memory_budget = None           # bytes; None: ask the driver (torch.cuda.mem_get_info) — tests set it to force the split

    def batch_number(self, b: int, h: int, w: int) -> int:
        """VAE.decode's split of the batch by free memory (LD.py:6357-6362: batch_number = free_memory / memory_used): the largest slice
        of the batch whose planned workspace (`ld_vae_plan_bytes`, a host-only dry run of the executor) fits the budget; at least 1."""
        mb, mh, mw = self._reserved
        if b <= mb and h <= mh and w <= mw:
            return b                                              # fits the workspace that is already there
        budget = self.memory_budget
        if budget is None:
            free, _ = torch.cuda.mem_get_info(self.device)
            budget = self.memory_fraction * (free + (self.workspace_bytes if any(self._reserved) else 0))   # (a re-reserve frees the old one first)
        n = b
        while n > 1:
            need = lib().ld_vae_plan_bytes(self._h, n, max(h, mh), max(w, mw))
            if 0 < need <= budget:
                break
            n = (n + 1) // 2
        return n

    def decode_device(self, z: torch.Tensor) -> torch.Tensor:
        z = z.to(self.device, torch.float32).contiguous()
        b, _, h, w = z.shape
        n = self.batch_number(b, h, w)
        self._ensure(n, h, w)
        out = torch.empty(b, 8 * h, 8 * w, self.cfg["out_ch"], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for x in range(0, b, n):                              # (one slice when everything fits: the usual case with 288 GB)
                m = min(n, b - x)
                check(lib().ld_vae_decode(self._h, z[x:x + m].data_ptr(), out[x:x + m].data_ptr(), m, h, w, _stream()), "ld_vae_decode")
        return out

    def profile_decode(self, z: torch.Tensor) -> list:
        """One decode with HIP events around every launch -> [(what, dims, flops, microseconds, kernel)] in launch order."""
        z = z.to(self.device, torch.float32).contiguous()
        b, _, h, w = z.shape
        self._ensure(b, h, w)
        out = torch.empty(b, 8 * h, 8 * w, self.cfg["out_ch"], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().ld_vae_profile(self._h, z.data_ptr(), out.data_ptr(), b, h, w, _stream()), "ld_vae_profile")
        return _parse_launches(lib().ld_vae_profile_launches, self._h)

    def decode(self, samples_in: torch.Tensor) -> torch.Tensor:
        """VAE.decode (LD.py:6357-6381): returns [B, 8h, 8w, 3] fp32 in [0,1] on the CPU (`intermediate_device`)."""
        return self.decode_device(samples_in).cpu()

    def encode_moments(self, pixel_samples: torch.Tensor) -> torch.Tensor:
        """pixels [B, H, W, 3] in [0,1] -> moments [B, 2z, H/8, W/8] fp32 on the device (mean | logvar): everything of
        VAE.encode (LD.py:6383-6410) up to, but excluding, the regularizer's random sample."""
        if not self.with_encoder:
            raise RuntimeError("this MI355XVAE was created without the encoder (with_encoder=True)")
        px = pixel_samples[..., :3].to(self.device, torch.float32).movedim(-1, 1)
        px = (px * 2.0 - 1.0).contiguous()                                   # process_input, LD.py:6295
        b, _, H, W = px.shape
        h, w = H // 8, W // 8
        if h * 8 != H or w * 8 != W:
            raise ValueError("image sides must be multiples of 8")
        self._ensure(b, h, w)
        out = torch.empty(b, 2 * self.cfg["z_channels"], h, w, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().ld_vae_encode(self._h, px.data_ptr(), out.data_ptr(), b, h, w, _stream()), "ld_vae_encode")
        return out

    def encode(self, pixel_samples: torch.Tensor) -> torch.Tensor:
        """VAE.encode (LD.py:6383-6410): [B,H,W,3] in [0,1] -> latent [B,4,H/8,W/8] fp32 on the CPU.  The posterior sample
        (DiagonalGaussianDistribution.sample, LD.py:175-179) draws torch.randn on the HOST global generator, as the reference does."""
        m = self.encode_moments(pixel_samples).cpu()
        mean, logvar = torch.chunk(m, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        return mean + std * torch.randn(mean.shape)


def synthetic_unet(cfg: Optional[dict] = None, seed: int = 0, **kw) -> MI355XUNet:
    cfg = cfg or W.sd15_unet_config()
    return MI355XUNet(cfg, lambda name, shape: W.synth_tensor(name, shape, seed), **kw)


def synthetic_vae(cfg: Optional[dict] = None, seed: int = 0, **kw) -> MI355XVAE:
    cfg = cfg or W.sd15_vae_config()
    return MI355XVAE(cfg, lambda name, shape: W.synth_tensor(name, shape, seed), **kw)

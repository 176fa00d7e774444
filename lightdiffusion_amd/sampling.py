"""Host mirror of the reference's sampling call surface (LD.py:827-1351, 2435-3203, 6657-6725), driving the
MI355X UNet.  Same names, argument meaning and error behaviour; the arithmetic on latents runs on the device
through the C ABI (cfg mix, sampler axpy), the schedule math stays on the host in fp32 exactly as the reference
computes it.

RNG parity: the initial noise comes from `torch.manual_seed(seed)` on the CPU generator
(prepare_noise, LD.py:3145-3153).  Euler-ancestral's per-step noise is `torch.randn_like(x)` on x's device in the
reference — on its CPU path that is the *global CPU generator*, so this mirror draws it on the host in the same order
and uploads it; for a sharded batch every rank draws the full-batch tensor and slices its rows.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, List, Optional

import torch

from . import ops

_CTX_TOKENS = itertools.count(1)   # per-run context tokens (never reused within a process)

LATENT_SCALE = 0.18215          # SD15.scale_factor, LD.py:137-147
KSAMPLER_NAMES = ["euler_ancestral", "dpm_adaptive", "dpmpp_2m_sde"]          # LD.py:2725-2729
SCHEDULER_NAMES = ["normal", "karras", "exponential", "sgm_uniform", "simple", "ddim_uniform"]   # LD.py:3034-3041


# ------------------------------------------------------------------ model sampling (EPS + ModelSamplingDiscrete)
class ModelSampling:
    """LD.py:1258-1351.  1000-entry sigma table from scaled-linear betas (0.00085 → 0.012), built in fp64."""

    sigma_data = 1.0

    def __init__(self, linear_start: float = 0.00085, linear_end: float = 0.012, timesteps: int = 1000):
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        sig = ((1 - acp) / acp) ** 0.5
        self.sigmas = sig.float()
        self.log_sigmas = sig.log().float()
        self.num_timesteps = timesteps

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma: torch.Tensor) -> torch.Tensor:
        d = sigma.detach().float().cpu().log() - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, timestep: torch.Tensor) -> torch.Tensor:
        t = torch.clamp(timestep.float().cpu(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()

    # EPS
    def noise_scaling(self, sigma, noise, latent_image, max_denoise=False):
        scale = math.sqrt(1.0 + float(sigma) ** 2.0) if max_denoise else float(sigma)
        return noise * scale + latent_image

    def inverse_noise_scaling(self, sigma, latent):
        return latent


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0, device="cpu"):
    """LD.py:831-837."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    sig = (hi + ramp * (lo - hi)) ** rho
    return torch.cat([sig, sig.new_zeros([1])]).to(device)


def normal_scheduler(model_sampling: ModelSampling, steps: int) -> torch.Tensor:
    """LD.py:2639-2651: linspace in t, log-sigma interpolation."""
    s = model_sampling
    ts = torch.linspace(s.timestep(s.sigma_max), s.timestep(s.sigma_min), steps)
    return torch.FloatTensor([float(s.sigma(t)) for t in ts] + [0.0])


def calculate_sigmas(model_sampling: ModelSampling, scheduler_name: str, steps: int) -> torch.Tensor:
    """LD.py:3045-3054: only `karras` and `normal` exist in the reference."""
    if scheduler_name == "karras":
        return get_sigmas_karras(steps, float(model_sampling.sigma_min), float(model_sampling.sigma_max))
    if scheduler_name == "normal":
        return normal_scheduler(model_sampling, steps)
    raise ValueError(f"unknown scheduler '{scheduler_name}' (the reference implements 'karras' and 'normal')")


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """LD.py:844-850."""
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - sigma_up ** 2) ** 0.5, sigma_up


def prepare_noise(latent_image: torch.Tensor, seed: int, noise_inds=None) -> torch.Tensor:
    """LD.py:3145-3153: CPU generator seeded by `seed`, fp32."""
    gen = torch.manual_seed(seed)
    return torch.randn(latent_image.size(), dtype=latent_image.dtype, layout=latent_image.layout, generator=gen, device="cpu")


def host_noise_sampler(x: torch.Tensor, rows: Optional[slice] = None, full_batch: Optional[int] = None) -> Callable:
    """default_noise_sampler (LD.py:853-854) for a device-resident x: draw on the host's global generator, upload.
    `rows`/`full_batch`: draw the whole global batch and keep this rank's rows (batch sharding)."""
    shape = list(x.shape)
    if full_batch is not None:
        shape[0] = full_batch

    def sample(sigma, sigma_next):
        n = torch.randn(shape, dtype=x.dtype)
        if rows is not None:
            n = n[rows]
        return n.to(x.device, non_blocking=True)
    return sample


class IIDIntervalNoise:
    """Stand-in for BrownianTreeNoiseSampler (LD.py:889-903, third-party torchsde, absent): the sampler queries
    consecutive disjoint sigma intervals once each and divides by sqrt(|t1 - t0|), so the draws are iid N(0,1).
    Same distribution, different stream of numbers than torchsde — 'parity unpinned' for eta > 0 with default noise."""

    def __init__(self, x: torch.Tensor, seed: Optional[int]):
        self.gen = torch.Generator().manual_seed(0 if seed is None else int(seed) & 0x7FFFFFFFFFFFFFFF)
        self.shape, self.dtype, self.device = x.shape, x.dtype, x.device

    def __call__(self, sigma, sigma_next):
        return torch.randn(self.shape, dtype=self.dtype, generator=self.gen).to(self.device, non_blocking=True)


# ------------------------------------------------------------------ k-diffusion samplers on device latents
def _interrupted(extra_args) -> bool:
    stop = (extra_args or {}).get("should_stop")      # replaces the reference's global app.interrupt_flag (LD.py:922)
    return bool(stop and stop())


@torch.no_grad()
def sample_euler_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None):
    """LD.py:907-941.  x fp32 on the device; one fused axpy per step: x <- x + d*dt + noise*s_noise*sigma_up."""
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = host_noise_sampler(x) if noise_sampler is None else noise_sampler
    s_in = x.new_ones([x.shape[0]])
    sig = [float(s) for s in sigmas]
    x = x.clone()
    for i in range(len(sig) - 1):
        if _interrupted(extra_args):
            break
        denoised = model(x, sig[i] * s_in, **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        r = (sigma_down - sig[i]) / sig[i]                       # d*dt = (x - denoised) * r
        if sig[i + 1] > 0:
            ops.axpby_(x, 1.0 + r, denoised, -r, noise_sampler(sig[i], sig[i + 1]), s_noise * sigma_up)
        else:
            ops.axpby_(x, 1.0 + r, denoised, -r)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sig[i], "denoised": denoised})
    return x


@torch.no_grad()
def sample_dpmpp_2m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1.0, s_noise=1.0, noise_sampler=None,
                        solver_type="midpoint"):
    """LD.py:1174-1244.  eta = 0 is deterministic DPM-Solver++(2M) ("DPM++ 2M" of the configs)."""
    extra_args = {} if extra_args is None else extra_args
    if solver_type not in ("midpoint", "heun"):
        raise ValueError("solver_type must be 'midpoint' or 'heun'")
    if noise_sampler is None and eta:
        noise_sampler = IIDIntervalNoise(x, extra_args.get("seed"))
    s_in = x.new_ones([x.shape[0]])
    sig = torch.as_tensor(sigmas, dtype=torch.float32).cpu()
    x = x.clone()
    old_denoised, h_last, h = None, None, None
    for i in range(len(sig) - 1):
        if _interrupted(extra_args):
            break
        denoised = model(x, float(sig[i]) * s_in, **extra_args)
        if sig[i + 1] == 0:
            x = denoised.clone()
        else:
            t, s = -sig[i].log(), -sig[i + 1].log()
            h = s - t
            eta_h = eta * h
            a = float(sig[i + 1] / sig[i] * (-eta_h).exp())
            c1 = float((-h - eta_h).expm1().neg())
            if old_denoised is None:
                ops.axpby_(x, a, denoised, c1)
            else:
                r = h_last / h
                k = float(((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / r)) if solver_type == "heun" \
                    else float(0.5 * (-h - eta_h).expm1().neg() * (1 / r))
                ops.axpby_(x, a, denoised, c1 + k, old_denoised, -k)
            if eta:
                amp = float(sig[i + 1] * (-2 * eta_h).expm1().neg().sqrt() * s_noise)
                ops.axpby_(x, 1.0, noise_sampler(float(sig[i]), float(sig[i + 1])), amp)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": float(sig[i]), "denoised": denoised})
        old_denoised, h_last = denoised, h
    return x


class _PIDStep:
    """PID step-size controller of DPM-Solver-12/23 (LD.py:944-973): h <- h * limiter(prod inv_err_k ** b_k)."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order, accept_safety, eps=1e-8):
        self.h, self.accept_safety, self.eps = h, accept_safety, eps
        self.b = ((pcoeff + icoeff + dcoeff) / order, -(pcoeff + 2 * dcoeff) / order, dcoeff / order)
        self.hist = None

    def propose(self, error: float) -> bool:
        inv = 1.0 / (float(error) + self.eps)
        if self.hist is None:
            self.hist = [inv, inv, inv]
        self.hist[0] = inv
        factor = self.hist[0] ** self.b[0] * self.hist[1] ** self.b[1] * self.hist[2] ** self.b[2]
        factor = 1 + math.atan(factor - 1)
        accept = factor >= self.accept_safety
        if accept:
            self.hist[2], self.hist[1] = self.hist[1], self.hist[0]
        self.h = self.h * factor
        return accept


@torch.no_grad()
def sample_dpm_adaptive(model, x, sigma_min, sigma_max, extra_args=None, callback=None, disable=None, order=3, rtol=0.05, atol=0.0078,
                        h_init=0.05, pcoeff=0.0, icoeff=1.0, dcoeff=0.0, accept_safety=0.81, eta=0.0, s_noise=1.0,
                        noise_sampler=None, return_info=False):
    """DPM-Solver-12/23 with adaptive steps (LD.py:976-1170) — the reference GUI's default sampler (LD.py:10572-10576).
    Three UNet evaluations per proposed step (eps at s, at s + h/3, at s + 2h/3; the 2nd-order estimate shares the first
    two), embedded error estimate, PID step control in t = -log(sigma).  The reference's variant adds no noise (su = 0)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    extra_args = {} if extra_args is None else extra_args
    sig = lambda t: (-t).exp()                                       # t is a 0-dim fp32 tensor, like the reference's
    ones = x.new_ones([x.shape[0]])

    def eps_at(xx, t):
        s = sig(t)
        return (xx - model(xx, float(s) * ones, **extra_args)) / float(s)

    t_start, t_end = -torch.tensor(float(sigma_max)).log(), -torch.tensor(float(sigma_min)).log()
    forward = bool(t_end > t_start)
    pid = _PIDStep(abs(h_init) * (1 if forward else -1), pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
    atol_t, rtol_t = torch.tensor(atol, device=x.device), torch.tensor(rtol, device=x.device)
    s, x_prev = t_start, x
    info = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
    r1, r2 = 1.0 / 3.0, 2.0 / 3.0
    while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
        if _interrupted(extra_args):
            break
        t = torch.minimum(t_end, s + pid.h) if forward else torch.maximum(t_end, s + pid.h)
        h = t - s
        e0 = eps_at(x, s)
        s1, s2 = s + r1 * h, s + r2 * h
        u1 = x - float(sig(s1) * (r1 * h).expm1()) * e0
        e1 = eps_at(u1, s1)
        x_low = x - float(sig(t) * h.expm1()) * e0 - float(sig(t) / (2 * r1) * h.expm1()) * (e1 - e0)
        u2 = x - float(sig(s2) * (r2 * h).expm1()) * e0 - float(sig(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1)) * (e1 - e0)
        e2 = eps_at(u2, s2)
        x_high = x - float(sig(t) * h.expm1()) * e0 - float(sig(t) / r2 * (h.expm1() / h - 1)) * (e2 - e0)
        delta = torch.maximum(atol_t, rtol_t * torch.maximum(x_low.abs(), x_prev.abs()))
        error = float(torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5)
        if pid.propose(error):
            x_prev, x, s = x_low, x_high, t
            info["n_accept"] += 1
        else:
            info["n_reject"] += 1
        info["nfe"] += order
        info["steps"] += 1
        if callback is not None:
            callback({"x": x, "i": info["steps"], "sigma": float(sig(s)), "denoised": None})
    return (x, info) if return_info else x


# ------------------------------------------------------------------ guidance
def convert_cond(cond):
    """LD.py:2287-2297 (cond = [[tensor, {..}], ...])."""
    out = []
    for c in cond:
        d = c[1].copy()
        d["cross_attn"] = c[0]
        out.append(d)
    return out


def _cat_ctx(ctx_list: List[torch.Tensor]) -> torch.Tensor:
    """CONDCrossAttn.concat (LD.py:647-663): pad shorter contexts by repetition up to the lcm of the token counts."""
    tgt = 1
    for c in ctx_list:
        tgt = abs(tgt * c.shape[1]) // math.gcd(tgt, c.shape[1])
    return torch.cat([c if c.shape[1] == tgt else c.repeat(1, tgt // c.shape[1], 1) for c in ctx_list])


def sampling_function(model, x, timestep, uncond, cond, cond_scale, model_options=None, seed=None):
    """sampling_function / calc_cond_batch / cfg_function (LD.py:2492-2626): ONE batched UNet call in the order
    [uncond, cond] (the reference reverses its to-run list, LD.py:2515), then uncond + (cond - uncond) * scale.

    Several entries in a conditioning list: the reference runs every entry over the WHOLE latent with weight 1 and averages
    them per list — its `get_area_and_mult` (LD.py:2435-2458) hard-codes area = the full latent and strength = 1.0 and never
    looks at "area", "strength", "mask" or timestep-range keys, so neither does this mirror (same images as the reference;
    those keys are inert there).  That case takes one eager UNet call on (len(uncond) + len(cond)) * B samples; the single-entry
    case — every call the reference's own pipelines make — replays the captured hipGraph."""
    model_options = model_options or {}
    b = x.shape[0]
    if not cond or not uncond:
        raise ValueError("sampling_function needs at least one positive and one negative conditioning entry")

    def ctx_of(c):
        t = c["cross_attn"]
        if t.shape[0] != b:
            if t.shape[0] != 1:
                raise RuntimeError(f"conditioning batch {t.shape[0]} does not match latent batch {b}")
            t = t.expand(b, -1, -1)
        return t

    # the batched context is step-invariant: build it once per run (cache lives in the guider's per-run model_options) and
    # tag it with a token that is never reused, so the UNet wrapper re-projects K / V^T exactly once per run.
    # Batch order = the reference's: its to-run list [cond.., uncond..] is consumed from the back (LD.py:2505-2515).
    n_u, n_c = len(uncond), len(cond)
    cache = model_options.setdefault("_ld_ctx_cache", {})
    key = (b, id(uncond), id(cond))
    if key not in cache:
        cache.clear()
        cache[key] = (_cat_ctx([ctx_of(c) for c in reversed(uncond)] + [ctx_of(c) for c in reversed(cond)]).contiguous(), next(_CTX_TOKENS))
    ctx, token = cache[key]
    wrapper = model_options.get("model_function_wrapper")
    single = n_u == 1 and n_c == 1
    if single and hasattr(wrapper, "cfg_denoise") and not model_options.get("ld_eager_unbatched", False):
        # MI355X fast path: the whole guided step (cat, UNet on 2B samples, CFG mix) is one replayed hipGraph
        return wrapper.cfg_denoise(x, timestep, ctx, cond_scale, token=token, use_graph=model_options.get("ld_use_graph", True))
    chunks = n_u + n_c
    xs = torch.cat([x] * chunks)
    ss = torch.cat([timestep] * chunks)
    cou = [1] * n_u + [0] * n_c
    c = {"c_crossattn": ctx, "transformer_options": {"cond_or_uncond": cou, "sigmas": timestep, "ld_ctx_token": token}}
    if wrapper is not None:
        out = wrapper(model.apply_model, {"input": xs, "timestep": ss, "c": c, "cond_or_uncond": cou})
    else:
        out = model.apply_model(xs, ss, **c)
    if single:
        return ops.cfg_combine(out.contiguous(), cond_scale)
    parts = out.chunk(chunks)                      # out_conds[i] = sum(output * 1) / count  (LD.py:2569-2591)
    uncond_pred = torch.stack(parts[:n_u]).sum(0) / float(n_u)
    cond_pred = torch.stack(parts[n_u:]).sum(0) / float(n_c)
    return uncond_pred + (cond_pred - uncond_pred) * cond_scale


class KSAMPLER:
    """LD.py:2732-2773."""

    def __init__(self, sampler_function, extra_options=None, inpaint_options=None):
        self.sampler_function = sampler_function
        self.extra_options = extra_options or {}
        self.inpaint_options = inpaint_options or {}

    def max_denoise(self, model_wrap, sigmas):
        max_sigma = float(model_wrap.inner_model.model_sampling.sigma_max)
        sigma = float(sigmas[0])
        return math.isclose(max_sigma, sigma, rel_tol=1e-05) or sigma > max_sigma

    def sample(self, model_wrap, sigmas, extra_args, callback, noise, latent_image=None, denoise_mask=None, disable_pbar=False):
        ms = model_wrap.inner_model.model_sampling
        noise = ms.noise_scaling(sigmas[0], noise, latent_image, self.max_denoise(model_wrap, sigmas))
        model_k = lambda x, sigma, **kw: model_wrap(x, sigma, **{k: v for k, v in kw.items() if k in ("model_options", "seed")})
        samples = self.sampler_function(model_k, noise, sigmas, extra_args=extra_args, callback=callback, disable=disable_pbar,
                                        **self.extra_options)
        return ms.inverse_noise_scaling(sigmas[-1], samples)


def ksampler(sampler_name, extra_options=None, inpaint_options=None):
    """LD.py:2776-2836."""
    extra_options = extra_options or {}
    if sampler_name == "euler_ancestral":
        fn = lambda model, noise, sigmas, extra_args, callback, disable, **o: sample_euler_ancestral(
            model, noise, sigmas, extra_args=extra_args, callback=callback, disable=disable, **o)
    elif sampler_name == "dpmpp_2m_sde":
        fn = lambda model, noise, sigmas, extra_args, callback, disable, **o: sample_dpmpp_2m_sde(
            model, noise, sigmas, extra_args=extra_args, callback=callback, disable=disable, **o)
    elif sampler_name == "dpm_adaptive":
        def fn(model, noise, sigmas, extra_args, callback, disable, **o):     # dpm_adaptive_function, LD.py:2777-2797
            if len(sigmas) <= 1:
                return noise
            sigma_min = sigmas[-1] if sigmas[-1] != 0 else sigmas[-2]
            return sample_dpm_adaptive(model, noise, float(sigma_min), float(sigmas[0]), extra_args=extra_args, callback=callback,
                                       disable=disable, **o)
    else:
        raise ValueError(f"unknown sampler '{sampler_name}'")
    return KSAMPLER(fn, extra_options, inpaint_options)


class CFGGuider:
    """LD.py:2894-3007."""

    def __init__(self, model_patcher):
        self.model_patcher = model_patcher
        self.model_options = model_patcher.model_options
        self.original_conds = {}
        self.cfg = 1.0

    def set_conds(self, positive, negative):
        self.original_conds = {"positive": convert_cond(positive), "negative": convert_cond(negative)}

    def set_cfg(self, cfg):
        self.cfg = cfg

    def __call__(self, x, timestep, model_options=None, seed=None):
        return sampling_function(self.inner_model, x, timestep, self.conds.get("negative"), self.conds.get("positive"), self.cfg,
                                 model_options=model_options or {}, seed=seed)

    def sample(self, noise, latent_image, sampler, sigmas, denoise_mask=None, callback=None, disable_pbar=False, seed=None):
        self.inner_model = self.model_patcher.model
        device = self.model_patcher.load_device
        self.conds = {k: [dict(c, cross_attn=c["cross_attn"].to(device)) for c in v] for k, v in self.original_conds.items()}
        noise, latent_image = noise.to(device), latent_image.to(device)
        if torch.count_nonzero(latent_image) > 0:                 # don't shift the empty latent (LD.py:2938-2941)
            latent_image = self.inner_model.process_latent_in(latent_image)
        self.model_options = dict(self.model_options)
        self.model_options.pop("_ld_ctx_cache", None)
        extra_args = {"model_options": self.model_options, "seed": seed}
        samples = sampler.sample(self, sigmas, extra_args, callback, noise, latent_image, denoise_mask, disable_pbar)
        out = self.inner_model.process_latent_out(samples.to(torch.float32))
        del self.inner_model, self.conds
        return out


def sample(model, noise, positive, negative, cfg, device, sampler, sigmas, model_options=None, latent_image=None, denoise_mask=None,
           callback=None, disable_pbar=False, seed=None):
    """LD.py:3010-3031."""
    g = CFGGuider(model)
    g.set_conds(positive, negative)
    g.set_cfg(cfg)
    return g.sample(noise, latent_image, sampler, sigmas, denoise_mask, callback, disable_pbar, seed)


class KSampler1:
    """LD.py:3062-3142."""
    SCHEDULERS = SCHEDULER_NAMES
    SAMPLERS = KSAMPLER_NAMES

    def __init__(self, model, steps, device, sampler=None, scheduler=None, denoise=None, model_options=None):
        self.model, self.device, self.scheduler, self.sampler = model, device, scheduler, sampler
        self.set_steps(steps, denoise)
        self.denoise = denoise
        self.model_options = model_options or {}

    def calculate_sigmas(self, steps):
        return calculate_sigmas(self.model.get_model_object("model_sampling"), self.scheduler, steps)

    def set_steps(self, steps, denoise=None):
        self.steps = steps
        if denoise is None or denoise > 0.9999:
            self.sigmas = self.calculate_sigmas(steps)
        else:
            self.sigmas = self.calculate_sigmas(int(steps / denoise))[-(steps + 1):]

    def sample(self, noise, positive, negative, cfg, latent_image=None, denoise_mask=None, sigmas=None, callback=None,
               disable_pbar=False, seed=None, **_ignored):
        sigmas = self.sigmas if sigmas is None else sigmas
        return sample(self.model, noise, positive, negative, cfg, self.device, ksampler(self.sampler), sigmas, self.model_options,
                      latent_image=latent_image, denoise_mask=denoise_mask, callback=callback, disable_pbar=disable_pbar, seed=seed)


def sample1(model, noise, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0, noise_mask=None,
            sigmas=None, callback=None, disable_pbar=False, seed=None, **_ignored):
    """LD.py:3156-3203: returns the samples on the intermediate (CPU) device."""
    ks = KSampler1(model, steps=steps, device=model.load_device, sampler=sampler_name, scheduler=scheduler, denoise=denoise,
                   model_options=model.model_options)
    out = ks.sample(noise, positive, negative, cfg=cfg, latent_image=latent_image, denoise_mask=noise_mask, sigmas=sigmas,
                    callback=callback, disable_pbar=disable_pbar, seed=seed)
    return out.to("cpu")


def common_ksampler(model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent, denoise=1.0, **_ignored):
    """LD.py:6657-6701."""
    latent_image = latent["samples"]
    noise = prepare_noise(latent_image, seed, latent.get("batch_index"))
    samples = sample1(model, noise, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=denoise, seed=seed)
    out = latent.copy()
    out["samples"] = samples
    return (out,)

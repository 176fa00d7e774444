"""TEST INFRASTRUCTURE ONLY — loads selected classes/functions of the reference for golden generation.

The reference (`/root/reference/LightDiffusion.py`) cannot be imported whole: import performs
network fetches (LD.py:53-120), imports GUI / detector packages that are not installed and calls
`torch.cuda.current_device()` (LD.py:1427 via :1460).  Following SURVEY.md §8(c) this loader
parses the file, and executes ONLY the whitelisted top-level `class`/`def`/assignment nodes, in
source order, inside a private namespace seeded with the handful of globals they use.  Nothing from
the reference is copied into this repository: the reference is read where it lies, at golden
generation time, in the build container only (the GPU box has no /root/reference).

Used by `oracle/make_golden.py`.  Never imported by the product package.
"""
from __future__ import annotations

import ast
import collections
import logging
import math
import os
import threading
import types
from abc import abstractmethod
from enum import Enum
from typing import Tuple, Union

import numpy as np
import torch
import torch as th
import torch.nn as nn
import torch.nn.functional as F
from einops import rearrange

REF_PATH = os.environ.get("LD_REFERENCE", "/root/reference/LightDiffusion.py")

# top-level names executed from the reference, by kind
_DEFS = {
    # latent format / misc helpers
    "LatentFormat", "SD15", "append_dims", "repeat_to_batch_size", "bislerp", "common_upscale",
    "lcm", "CONDRegular", "CONDCrossAttn",
    # schedules / samplers (LD.py:787-1244)
    "make_beta_schedule", "checkpoint", "timestep_embedding", "zero_module", "append_zero",
    "get_sigmas_karras", "to_d", "get_ancestral_step", "default_noise_sampler",
    "sample_euler_ancestral", "sample_dpmpp_2m_sde", "PIDStepSizeController", "DPMSolver", "sample_dpm_adaptive",
    "TimestepBlock1", "TimestepEmbedSequential1", "EPS", "ModelSamplingDiscrete",
    # sampling core (LD.py:2282-3203)
    "get_models_from_cond", "convert_cond", "get_additional_models", "prepare_sampling", "cleanup_models",
    "cast_bias_weight", "CastWeightBiasOp", "disable_weight_init", "manual_cast",
    "get_area_and_mult", "cond_equal_size", "can_concat_cond", "cond_cat", "calc_cond_batch",
    "cfg_function", "sampling_function", "KSamplerX0Inpaint", "normal_scheduler",
    "resolve_areas_and_cond_masks", "create_cond_with_same_area_if_none",
    "calculate_start_end_timesteps", "pre_run_control", "apply_empty_x_to_equal_area",
    "encode_model_conds", "Sampler", "KSAMPLER", "ksampler", "process_conds", "CFGGuider", "sample",
    "calculate_sigmas", "sampler_object", "KSampler1", "prepare_noise", "sample1",
    "ModelPatcher", "module_size", "get_attr", "set_attr", "DiagonalGaussianRegularizer",
    # VAE (LD.py:3446-3882)
    "DiagonalGaussianDistribution", "DiagonalGaussianRegularizer", "AutoencodingEngine", "nonlinearity",
    "Upsample", "Downsample", "ResnetBlock", "pytorch_attention", "AttnBlock", "make_attn", "Encoder", "Decoder",
    # transformer blocks (LD.py:3898-4262, 4497-4515)
    "FeedForward", "Normalize", "attention_pytorch", "CrossAttention", "BasicTransformerBlock",
    "SpatialTransformer", "exists", "default", "GEGLU",
    # CLIP-L (LD.py:4268-4487)
    "CLIPAttention", "CLIPMLP", "CLIPLayer", "CLIPEncoder", "CLIPEmbeddings", "CLIPTextModel_", "CLIPTextModel",
    # prompt weighting parser (LD.py:4733-4793)
    "parse_parentheses", "token_weights", "escape_important", "unescape_important", "SDTokenizer",
    # token-weight encoder around the text model (LD.py:4526-4730): the e2e goldens start from token ids
    "gen_empty_tokens", "ClipTokenWeightEncoder", "SDClipModel",
    # UNet (LD.py:5083-5767)
    "forward_timestep_embed1", "Upsample1", "Downsample1", "ResBlock1", "apply_control1", "UNetModel1",
    # model wrappers (LD.py:5779-5976)
    "ModelType", "model_sampling", "BaseModel", "BASE", "sm_SD15",
    # LoRA ingestion (LD.py:232-394, 401-426, 538-629, 1986-2010, 6203-6219)
    "unet_to_diffusers", "set_attr_param", "copy_to_param", "load_lora", "model_lora_keys_clip", "model_lora_keys_unet",
    "cast_to_device", "is_intel_xpu",
    # node API (LD.py:6573-6725)
    "EmptyLatentImage", "LatentUpscale", "common_ksampler", "KSampler2",
}
_ASSIGNS = {"ops", "oai_ops", "ae_ops", "ACTIVATIONS", "_ATTN_PRECISION", "KSAMPLER_NAMES",
            "SCHEDULER_NAMES", "SAMPLER_NAMES", "PROGRESS_BAR_ENABLED",
            "UNET_MAP_ATTENTIONS", "TRANSFORMER_BLOCKS", "UNET_MAP_RESNET", "UNET_MAP_BASIC", "LORA_CLIP_MAP"}


class _NullBar:
    """tqdm stand-in: usable as `tqdm(iterable)` and as `with tqdm(disable=..) as pbar: pbar.update()` (LD.py:1143-1144)."""

    def __init__(self, it=None, **_k):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, *a, **k):
        return None


class _StubApp:
    """Stands in for the Tk `app` global the sampler loops poll (LD.py:922-937)."""
    interrupt_flag = False

    class previewer_checkbox:
        @staticmethod
        def get():
            return False

    @staticmethod
    def title(*_a, **_k):
        return None


def load_reference(path: str = REF_PATH) -> types.SimpleNamespace:
    with open(path, "r", encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)

    cpu = torch.device("cpu")
    ns = {
        "__name__": "ld_reference_extract",
        "torch": torch, "th": th, "nn": nn, "F": F, "math": math, "np": np, "logging": logging,
        "rearrange": rearrange, "collections": collections, "threading": threading,
        "abstractmethod": abstractmethod, "Enum": Enum, "Union": Union, "Tuple": Tuple,
        "trange": lambda n, disable=None: range(n), "tqdm": _NullBar,
        "app": _StubApp(),
        # stubs for the device / memory manager (LD.py:1362-2265, out of scope): CPU, never offload
        "xformers_enabled": lambda: False, "xformers_enabled_vae": lambda: False,
        "device_supports_non_blocking": lambda device: False,
        "intermediate_device": lambda: cpu, "get_torch_device": lambda: cpu,
        "unet_offload_device": lambda: cpu,
        "load_models_gpu": lambda *a, **k: None, "get_free_memory": lambda *a, **k: 1 << 50,
        "dtype_size": lambda dtype: torch.empty((), dtype=dtype).element_size(),
        "taesd_preview": lambda x: None,
        "copy": __import__("copy"), "uuid": __import__("uuid"), "json": __import__("json"),
        "isfunction": __import__("inspect").isfunction,
        "CLIPTokenizer": None,      # default argument of SDTokenizer.__init__; goldens inject their own word tokenizer
        "optimized_attention_for_device": lambda device, mask=False, small_input=False: ns["attention_pytorch"],
    }
    done = set()
    for node in tree.body:
        name = None
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in _DEFS:
            name = node.name
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in _ASSIGNS:
            name = node.targets[0].id
        if name is None:
            continue
        mod = ast.Module(body=[node], type_ignores=[])
        exec(compile(mod, path, "exec"), ns)
        done.add(name)
        if name == "attention_pytorch":
            # LD.py:3981-3988 selects the attention function at import; xformers is excluded
            ns["optimized_attention"] = ns["attention_pytorch"]
            ns["optimized_attention_masked"] = ns["attention_pytorch"]
    missing = (_DEFS | _ASSIGNS) - done
    if missing:
        raise RuntimeError(f"reference symbols not found: {sorted(missing)}")
    return types.SimpleNamespace(**ns)


if __name__ == "__main__":
    ref = load_reference()
    print("extracted", len(_DEFS | _ASSIGNS), "symbols; UNetModel1 =", ref.UNetModel1)

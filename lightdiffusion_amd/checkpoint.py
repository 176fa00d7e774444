"""Single-file SD1.x checkpoint ingestion: key layout and architecture detection.

Key layout the reference loads (LD.py:5921-5922, 5980-6009, 6446-6465):
  model.diffusion_model.*                      UNet
  first_stage_model.{encoder,decoder,quant_conv,post_quant_conv}.*   VAE
  cond_stage_model.transformer.[text_model.]*  CLIP-L
The architecture is inferred from which keys exist and their shapes, like the reference's detect_unet_config
(LD.py:6065-6182) — restated here for the SD1.x family only (no video / SDXL branches).  LoRA: the reference merges
`W += alpha * up @ down` into the weights before sampling (LD.py:3335-3354, 3407-3424); `merge_lora` does that merge on the
state dict, so the HIP side only ever sees final weights.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import torch

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."
CLIP_PREFIX = "cond_stage_model.transformer."


def _count(keys, pattern: str) -> int:
    """number of distinct integer indices i for which a key matches pattern.format(i) as a prefix"""
    rx = re.compile("^" + re.escape(pattern).replace(r"\{\}", r"(\d+)"))
    idx = {int(m.group(1)) for k in keys for m in [rx.match(k)] if m}
    return max(idx) + 1 if idx else 0


def detect_unet_config(sd: Dict[str, torch.Tensor], prefix: str = UNET_PREFIX, num_heads: int = 8) -> dict:
    keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
    ks = set(keys)
    if "input_blocks.0.0.weight" not in ks:
        raise ValueError("not an SD1.x checkpoint: no UNet keys under '%s'" % prefix)
    get = lambda k: sd[prefix + k]
    mc, cin = get("input_blocks.0.0.weight").shape[:2]
    depth_of = lambda p: _count(keys, p + ".transformer_blocks.{}.")
    channel_mult, num_res, td_in = [], [], []
    level_res, level_mult, context_dim = 0, 0, None
    for i in range(1, _count(keys, "input_blocks.{}.")):
        if f"input_blocks.{i}.0.op.weight" in ks:
            channel_mult.append(level_mult)
            num_res.append(level_res)
            level_res = 0
        else:
            level_res += 1
            level_mult = get(f"input_blocks.{i}.0.out_layers.3.weight").shape[0] // mc
            d = depth_of(f"input_blocks.{i}.1")
            td_in.append(d)
            if d and context_dim is None:
                context_dim = get(f"input_blocks.{i}.1.transformer_blocks.0.attn2.to_k.weight").shape[1]
    channel_mult.append(level_mult)
    num_res.append(level_res)
    n_out = _count(keys, "output_blocks.{}.")
    # the reference builds this list back to front and pops from its end while constructing (LD.py:6107-6150, 5621)
    td_out = [depth_of(f"output_blocks.{j}.1") for j in reversed(range(n_out))]
    if max(td_in + td_out + [0]) > 1:
        raise ValueError("transformer depth > 1 is not an SD1.x UNet")
    return dict(in_channels=int(cin), out_channels=int(get("out.2.weight").shape[0]), model_channels=int(mc), channel_mult=channel_mult,
                num_res_blocks=num_res, transformer_depth=td_in, transformer_depth_output=td_out,
                transformer_depth_middle=depth_of("middle_block.1"), context_dim=int(context_dim), num_heads=num_heads)


def detect_vae_config(sd: Dict[str, torch.Tensor], prefix: str = VAE_PREFIX) -> Tuple[dict, bool]:
    keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
    get = lambda k: sd[prefix + k]
    ch = int(get("decoder.norm_out.weight").shape[0])
    levels = _count(keys, "decoder.up.{}.")
    mult = [int(get(f"decoder.up.{l}.block.0.conv1.weight").shape[0]) // ch for l in range(levels)]
    cfg = dict(z_channels=int(get("post_quant_conv.weight").shape[0]), ch=ch, ch_mult=mult,
               num_res_blocks=_count(keys, "decoder.up.0.block.{}.") - 1, out_ch=int(get("decoder.conv_out.weight").shape[0]))
    return cfg, any(k.startswith("encoder.") for k in keys)


def clip_state_dict(sd: Dict[str, torch.Tensor], prefix: str = CLIP_PREFIX) -> Dict[str, torch.Tensor]:
    """`cond_stage_model.transformer.*` -> keys under `text_model.` (sm_SD15.process_clip_state_dict, LD.py:5980-6009)."""
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            k2 = k[len(prefix):]
            k2 = k2 if k2.startswith("text_model.") else "text_model." + k2
            if not k2.endswith("position_ids"):
                out[k2] = v
    return out


def detect_clip_config(csd: Dict[str, torch.Tensor], num_heads: int = 12) -> dict:
    h = int(csd["text_model.embeddings.token_embedding.weight"].shape[1])
    return dict(vocab_size=int(csd["text_model.embeddings.token_embedding.weight"].shape[0]), hidden_size=h,
                intermediate_size=int(csd["text_model.encoder.layers.0.mlp.fc1.weight"].shape[0]),
                num_hidden_layers=_count(csd.keys(), "text_model.encoder.layers.{}."), num_attention_heads=num_heads,
                max_position_embeddings=int(csd["text_model.embeddings.position_embedding.weight"].shape[0]), hidden_act="quick_gelu")


def merge_lora(sd: Dict[str, torch.Tensor], lora: Dict[str, torch.Tensor], strength: float = 1.0, prefix: str = UNET_PREFIX) -> int:
    """Merge kohya-style LoRA pairs (`lora_unet_<key with _>.lora_up/down.weight`, optional `.alpha`) into the UNet weights
    in place: W += strength * alpha/rank * up @ down (calculate_weight, LD.py:3407-3424).  Returns the number of merged tensors."""
    by_flat = {"lora_unet_" + k[len(prefix):-len(".weight")].replace(".", "_"): k for k in sd if k.startswith(prefix) and k.endswith(".weight")}
    n = 0
    for name, key in by_flat.items():
        up, down = lora.get(name + ".lora_up.weight"), lora.get(name + ".lora_down.weight")
        if up is None or down is None:
            continue
        alpha = float(lora[name + ".alpha"]) / down.shape[0] if name + ".alpha" in lora else 1.0
        delta = torch.mm(up.flatten(start_dim=1).float(), down.flatten(start_dim=1).float()).reshape(sd[key].shape)
        sd[key] = (sd[key].float() + strength * alpha * delta).to(sd[key].dtype)
        n += 1
    return n


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    return ck.get("state_dict", ck)

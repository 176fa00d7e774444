"""CLIP-L text conditioning: prompt-weight parsing, 77-token chunking, the 12-layer causal text
transformer (on the HIP kernels through the C ABI: `CLIPTextModelHIP`, the package's only text model) and the per-token weight
lerp.  Runs once per prompt; its output is the payload of the one RCCL broadcast.

Mirrors: token_weights / parse_parentheses (LD.py:4733-4780), SDTokenizer.tokenize_with_weights (LD.py:4936-5031),
ClipTokenWeightEncoder.encode_token_weights (LD.py:4540-4569), SDClipModel.forward (LD.py:4692-4724),
CLIPTextModel_ (LD.py:4413-4463), CLIP.tokenize / encode_from_tokens / clip_layer (LD.py:6222-6272).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

START_TOKEN, END_TOKEN = 49406, 49407
_ESC_CLOSE, _ESC_OPEN = "\0\1", "\0\2"


# ------------------------------------------------------------------ "(text:1.2)" emphasis syntax
def _split_top_level_groups(text: str) -> List[str]:
    """Cut `text` at top-level parenthesis groups: a cut before a '(' seen at depth 0 and after the ')' that returns the
    depth to 0.  Depth may go negative on stray ')' (then nothing closes until it is back at 0) — same as the reference."""
    pieces, start, depth = [], 0, 0
    for i, ch in enumerate(text):
        if ch == "(":
            if depth == 0 and i > start:
                pieces.append(text[start:i])
                start = i
            depth += 1
        elif ch == ")":
            depth -= 1
            if depth == 0:
                pieces.append(text[start:i + 1])
                start = i + 1
    if start < len(text):
        pieces.append(text[start:])
    return pieces


def parse_prompt_weights(text: str, base: float = 1.0) -> List[Tuple[str, float]]:
    """(segment, weight) list.  A "( ... )" group multiplies the weight by 1.1, or sets it to the number after the
    group's last ':' when that parses as a float; groups nest."""
    out: List[Tuple[str, float]] = []
    for piece in _split_top_level_groups(text):
        if len(piece) >= 2 and piece[0] == "(" and piece[-1] == ")":
            inner, w = piece[1:-1], base * 1.1
            colon = inner.rfind(":")
            if colon > 0:
                try:
                    w, inner = float(inner[colon + 1:]), inner[:colon]
                except ValueError:
                    pass
            out.extend(parse_prompt_weights(inner, w))
        else:
            out.append((piece, base))
    return out


def escape_important(text: str) -> str:
    return text.replace("\\)", _ESC_CLOSE).replace("\\(", _ESC_OPEN)


def unescape_important(text: str) -> str:
    return text.replace(_ESC_CLOSE, ")").replace(_ESC_OPEN, "(")


class PromptTokenizer:
    """Word-level BPE front end + the reference's chunking rules: [BOS] tokens... [EOS] padded with EOS to 77; words of
    fewer than 8 tokens are never split across chunks; longer words are split and the chunk is closed with EOS."""

    max_word_length = 8

    def __init__(self, word_tokenizer: Callable[[str], List[int]], max_length: int = 77, start_token: int = START_TOKEN,
                 end_token: int = END_TOKEN, pad_with_end: bool = True):
        self.word_tokenizer = word_tokenizer
        self.max_length, self.start_token, self.end_token = max_length, start_token, end_token
        self.pad_token = end_token if pad_with_end else 0

    @classmethod
    def from_pretrained(cls, tokenizer_dir: str) -> "PromptTokenizer":
        """Build on HuggingFace's CLIPTokenizer files (the reference loads them from `_internal/sd1_tokenizer/`)."""
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(tokenizer_dir)
        empty = tok("")["input_ids"]
        return cls(lambda w: tok(w)["input_ids"][1:-1], start_token=empty[0], end_token=empty[1])

    def tokenize_with_weights(self, text: str) -> List[List[Tuple[int, float]]]:
        words: List[List[Tuple[int, float]]] = []
        for seg, w in parse_prompt_weights(escape_important(text), 1.0):
            for word in unescape_important(seg).replace("\n", " ").split(" "):
                if word:
                    words.append([(t, w) for t in self.word_tokenizer(word)])
        chunks: List[List[Tuple[int, float]]] = [[(self.start_token, 1.0)]]
        cur = chunks[0]
        room = self.max_length - 1                       # slots before the closing EOS
        for group in words:
            big = len(group) >= self.max_word_length
            while group:
                if len(group) + len(cur) > room:
                    left = room - len(cur)
                    if big:
                        cur.extend(group[:left])
                        cur.append((self.end_token, 1.0))
                        group = group[left:]
                    else:
                        cur.append((self.end_token, 1.0))
                        cur.extend([(self.pad_token, 1.0)] * left)
                    cur = [(self.start_token, 1.0)]
                    chunks.append(cur)
                else:
                    cur.extend(group)
                    group = []
        cur.append((self.end_token, 1.0))
        cur.extend([(self.pad_token, 1.0)] * (self.max_length - len(cur)))
        return chunks


# ------------------------------------------------------------------ text transformer
class CLIPTextModelHIP:
    """CLIP-L text transformer (CLIPTextModel_, LD.py:4413-4463) on the HIP kernels — the package's ONLY text model
    (the torch CPU restatement used by host-logic tests lives in oracle/sd15_ref.py: `clip_text_model`): LayerNorm, fused [q|k|v] projection, causal
    flash attention, out-projection + residual epilogue, fc1 + quick-GELU epilogue, fc2 + residual epilogue — all through
    the C ABI (`ops`), fp16 storage / fp32 accumulation.  Only the embedding gather (77 rows) and the final row pick for the
    pooled output stay torch indexing.  The reference computes this model in fp32; the conditioning it feeds is cast to
    fp16 by `apply_model` (LD.py:5846), so the tolerance here is the per-op fp16 one."""

    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], device="cuda:0"):
        self.cfg, self.device = dict(cfg), torch.device(device)
        self.w = {k: v.to(torch.float16).to(self.device) for k, v in weights.items()}
        P = "text_model.encoder.layers."
        self.qkv = []
        for i in range(cfg["num_hidden_layers"]):
            p = f"{P}{i}.self_attn."
            w = torch.cat([self.w[p + f"{t}_proj.weight"] for t in "qkv"]).contiguous()
            b = torch.cat([self.w[p + f"{t}_proj.bias"] for t in "qkv"]).contiguous()
            self.qkv.append((w, b))

    @torch.no_grad()
    def __call__(self, tokens: torch.Tensor, intermediate_output: Optional[int] = None):
        from . import ops
        P = "text_model."
        h, heads, nl = self.cfg["hidden_size"], self.cfg["num_attention_heads"], self.cfg["num_hidden_layers"]
        tokens = tokens.to(self.device)
        x = (self.w[P + "embeddings.token_embedding.weight"][tokens].float() +
             self.w[P + "embeddings.position_embedding.weight"].float()).half().contiguous()        # [B, 77, h]
        B, L = x.shape[:2]
        stop = None if intermediate_output is None else (nl + intermediate_output if intermediate_output < 0 else intermediate_output)
        inter = None
        ln = lambda t, p: ops.layer_norm(t, self.w[p + ".weight"], self.w[p + ".bias"], 1e-5)
        for i in range(nl):
            p = f"{P}encoder.layers.{i}"
            qkv = ops.linear(ln(x, p + ".layer_norm1"), *self.qkv[i])                                 # [B, L, 3h]
            q, k, v = (qkv[..., j * h:(j + 1) * h].contiguous() for j in range(3))
            a = ops.attention(q, k, v, heads, causal=True)
            x = ops.linear(a, self.w[p + ".self_attn.out_proj.weight"], self.w[p + ".self_attn.out_proj.bias"], residual=x)
            m = ops.linear(ln(x, p + ".layer_norm2"), self.w[p + ".mlp.fc1.weight"], self.w[p + ".mlp.fc1.bias"], act="quick_gelu")
            x = ops.linear(m, self.w[p + ".mlp.fc2.weight"], self.w[p + ".mlp.fc2.bias"], residual=x)
            if i == stop:
                inter = x.clone()
        x = ln(x, P + "final_layer_norm")
        if inter is not None:
            inter = ln(inter, P + "final_layer_norm")
        pooled = x[torch.arange(B, device=self.device), tokens.to(torch.int).argmax(dim=-1)]
        return x.float(), None if inter is None else inter.float(), pooled.float()


class CLIP:
    """The object `CLIPTextEncode.encode(clip, text)` drives (LD.py:6222-6272)."""

    def __init__(self, text_model, tokenizer: Optional[PromptTokenizer] = None, layer_idx: Optional[int] = None):
        """text_model(tokens[B, 77] long, intermediate_output=layer_idx) -> (last hidden state, hidden state at layer_idx or None, pooled):
        `CLIPTextModelHIP` in the product."""
        self.text_model, self.tokenizer, self.layer_idx = text_model, tokenizer, layer_idx

    def clone(self) -> "CLIP":
        return CLIP(self.text_model, self.tokenizer, self.layer_idx)

    def clip_layer(self, layer_idx: int) -> None:           # CLIPSetLastLayer → clip skip
        self.layer_idx = layer_idx

    def tokenize(self, text: str):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer attached: build one with PromptTokenizer.from_pretrained(<sd1_tokenizer dir>)")
        return {"l": self.tokenizer.tokenize_with_weights(text)}

    def _encode_ids(self, ids: List[List[int]]):
        last, inter, pooled = self.text_model(torch.tensor(ids, dtype=torch.long), intermediate_output=self.layer_idx)
        return (last if inter is None else inter).float(), pooled.float()

    def encode_from_tokens(self, tokens, return_pooled: bool = False):
        pairs: Sequence[Sequence[Tuple[int, float]]] = tokens["l"] if isinstance(tokens, dict) else tokens
        ids = [[t for t, _ in sec] for sec in pairs]
        weighted = any(w != 1.0 for sec in pairs for _, w in sec)
        n = len(ids)
        if weighted or n == 0:
            ids = ids + [[START_TOKEN, END_TOKEN] + [END_TOKEN] * (max(len(s) for s in ids) - 2 if ids else 75)]
        out, pooled = self._encode_ids(ids)
        secs = []
        for k in range(n):
            z = out[k:k + 1]
            if weighted:
                w = torch.tensor([w for _, w in pairs[k]], dtype=z.dtype, device=z.device)[None, :, None]
                z = torch.where(w != 1.0, (z - out[-1:]) * w + out[-1:], z)
            secs.append(z)
        cond = torch.cat(secs, dim=-2).cpu()                 # intermediate_device() is the CPU in the reference
        return (cond, pooled[0:1].cpu()) if return_pooled else cond

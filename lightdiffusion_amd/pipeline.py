"""Step-level runtime: the classifier-free-guided denoiser of one sampler step as a replayable hipGraph.

One sampler step of the reference = `sampling_function` (LD.py:2609-2626): cat([x, x]) → UNet on N = 2B samples in
the order [uncond, cond] → uncond + (cond - uncond) * cfg.  Here the B-sample input, sigma and the 2B-sample output live in
static device buffers, the ~345 kernel launches of the forward plus the guidance mix are captured once into a hipGraph and
replayed per step (sigma is read from device memory, so one graph serves every step).  The forward is the library's CFG-pair
entry (`ld_unet_forward_pair`): both halves of the batch are the SAME latents, so the layers in front of the first
cross-attention — conv_in, the first ResBlock, the first transformer's self-attention — are evaluated once.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .unet import MI355XUNet


class GraphedBody:
    """A fixed launch sequence on static buffers, captured once into a hipGraph and replayed.  Re-captured when what the captured
    launches bake in besides the static buffers has changed: the workspace addresses (the UNet's reserve epoch) and the number of
    context tokens (a kernel argument).  The context VALUES live at fixed addresses and may change freely.

    Capture runs in `thread_local` error mode: the reference calls its wrapper from a worker thread (LD.py:10453), and a live
    RCCL process group's watchdog thread issues event queries that must not invalidate a capture in progress."""

    def __init__(self, unet: MI355XUNet, body, use_graph: bool = True):
        self.unet, self.body, self.use_graph = unet, body, use_graph
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._key = None

    def _state_key(self):
        return (self.unet.reserve_epoch, self.unet.ctx_shape)

    def _capture(self) -> None:
        dev = self.unet.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.body()                       # warm-up outside capture (plans the shape, touches every kernel once)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.body()
        self.graph = g
        self._key = self._state_key()

    def launch(self) -> None:
        if not self.use_graph:
            self.body()
            return
        if self.graph is None or self._key != self._state_key():
            self._capture()
        self.graph.replay()


class CFGDenoiser:
    def __init__(self, unet: MI355XUNet, batch: int, h: int, w: int, cfg_scale: float, use_graph: bool = True):
        self.unet, self.batch, self.cfg_scale = unet, batch, float(cfg_scale)
        dev = unet.device
        c = unet.cfg["in_channels"]
        with torch.inference_mode(False):      # static buffers are plain tensors whatever mode the first call came in (the reference calls
            self.x1 = torch.zeros(batch, c, h, w, dtype=torch.float32, device=dev)          # under torch.inference_mode(), LD.py:10493: buffers
            self.sigma1 = torch.ones(batch, dtype=torch.float32, device=dev)                # created there could not be updated in place by a
            self.den2 = torch.zeros(2 * batch, c, h, w, dtype=torch.float32, device=dev)    # later call outside it)
            self.den = torch.zeros(batch, c, h, w, dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self._run = GraphedBody(unet, self._body, use_graph)

    @property
    def _graph(self):
        return self._run.graph

    def set_context(self, uncond: torch.Tensor, cond: torch.Tensor) -> None:
        """uncond / cond: [1 or B, T, D].  Batched as the reference batches them: [uncond x B ; cond x B] (LD.py:2515-2547)."""
        rep = lambda t: t if t.shape[0] == self.batch else t.expand(self.batch, -1, -1)
        self.unet.set_context(torch.cat([rep(uncond), rep(cond)]).contiguous())

    def _body(self) -> None:
        self.unet.forward_pair(self.x1, self.sigma1, out=self.den2)
        from ._lib import check, lib
        check(lib().ld_op_cfg_combine(self.den2.data_ptr(), self.den.data_ptr(), self.cfg_scale, self.den.numel(),
                                      torch.cuda.current_stream().cuda_stream), "ld_op_cfg_combine")

    def _launch(self) -> torch.Tensor:
        self._run.launch()
        return self.den

    def run(self, x: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        """x [B,C,h,w] fp32, timestep [B] fp32 (sigma per sample), both on the device: no host read, so the host runs ahead
        of the GPU across steps.  Returns the static output buffer (valid until the next call)."""
        self.x1.copy_(x)
        self.sigma1.copy_(timestep)
        return self._launch()

    def __call__(self, x: torch.Tensor, sigma: float) -> torch.Tensor:
        """x [B,C,h,w] fp32 on the device, sigma a host scalar -> guided denoised x0 [B,C,h,w] (static buffer)."""
        self.x1.copy_(x)
        self.sigma1.fill_(float(sigma))
        return self._launch()


class HookRunner:
    """The UNet forward behind the reference's `model_function_wrapper` seam (LD.py:2558-2567) as a replayed hipGraph on static
    buffers — what `enable_cuda_graph` does for the reference's own plugin on that seam (StableFastPatch, LD.py:9896-9933).

    One runner per (N, h, w): `pair` replays the CFG-pair forward (`ld_unet_forward_pair`: the layers in front of the first
    cross-attention once for both halves of calc_cond_batch's cat([x_in, x_in]), LD.py:2515-2547), `plain` the forward on all N rows.
    Which one is valid for a call is decided ON THE DEVICE (`ld_op_hook_check`, flags in pinned host memory) and read after the
    speculative replay has been queued, so the GPU never waits for the host."""

    def __init__(self, unet: MI355XUNet, n: int, h: int, w: int):
        dev = unet.device
        c = unet.cfg["in_channels"]
        self.unet, self.n = unet, n
        with torch.inference_mode(False):      # (plain tensors: see CFGDenoiser)
            self.x = torch.zeros(n, c, h, w, dtype=torch.float32, device=dev)
            self.sigma = torch.ones(n, dtype=torch.float32, device=dev)
            self.out = torch.zeros(n, c, h, w, dtype=torch.float32, device=dev)
        self.plain = GraphedBody(unet, lambda: unet.forward(self.x, self.sigma, out=self.out))
        half = n // 2
        self.pair = GraphedBody(unet, lambda: unet.forward_pair(self.x[:half], self.sigma[:half], out=self.out)) if n % 2 == 0 and n else None
        self.halves_differed = False      # set once a [1, 0] call's halves were NOT the same latents: that caller stays on `plain`

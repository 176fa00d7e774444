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


class CFGDenoiser:
    def __init__(self, unet: MI355XUNet, batch: int, h: int, w: int, cfg_scale: float, use_graph: bool = True):
        self.unet, self.batch, self.cfg_scale = unet, batch, float(cfg_scale)
        dev = unet.device
        c = unet.cfg["in_channels"]
        self.x1 = torch.zeros(batch, c, h, w, dtype=torch.float32, device=dev)
        self.sigma1 = torch.ones(batch, dtype=torch.float32, device=dev)
        self.den2 = torch.zeros(2 * batch, c, h, w, dtype=torch.float32, device=dev)
        self.den = torch.zeros(batch, c, h, w, dtype=torch.float32, device=dev)
        self.use_graph = use_graph
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph_key = None

    def set_context(self, uncond: torch.Tensor, cond: torch.Tensor) -> None:
        """uncond / cond: [1 or B, T, D].  Batched as the reference batches them: [uncond x B ; cond x B] (LD.py:2515-2547)."""
        rep = lambda t: t if t.shape[0] == self.batch else t.expand(self.batch, -1, -1)
        self.unet.set_context(torch.cat([rep(uncond), rep(cond)]).contiguous())

    def _body(self) -> None:
        self.unet.forward_pair(self.x1, self.sigma1, out=self.den2)
        from ._lib import check, lib
        check(lib().ld_op_cfg_combine(self.den2.data_ptr(), self.den.data_ptr(), self.cfg_scale, self.den.numel(),
                                      torch.cuda.current_stream().cuda_stream), "ld_op_cfg_combine")

    def _capture(self) -> None:
        side = torch.cuda.Stream(device=self.unet.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._body()                      # warm-up outside capture (plans the shape, touches every kernel once)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.unet.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self._graph = g
        self._graph_key = self._state_key()

    def _state_key(self):
        # what the captured launches bake in besides the static buffers: the workspace addresses (reserve epoch) and the
        # number of context tokens (a kernel argument).  The context VALUES live at fixed addresses and may change freely.
        return (self.unet.reserve_epoch, self.unet.ctx_shape)

    def _launch(self) -> torch.Tensor:
        if not self.use_graph:
            self._body()
        else:
            if self._graph is None or self._graph_key != self._state_key():
                self._capture()
            self._graph.replay()
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

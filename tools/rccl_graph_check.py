#!/usr/bin/env python3
"""RCCL process group + hipGraph capture in ONE process, world = 1, on cuda:0.

What `bench.py --gpus N` / `nodes.txt2img_sharded` do on every rank, as far as a one-GPU box can show it: the "nccl" (= RCCL) backend is
initialised first (its watchdog thread is alive from then on and issues event queries), the conditioning goes through
`dist.broadcast_conditioning` (a real ncclBroadcast on the world-1 communicator), `KSampler2.sample` captures the step's hipGraph WHILE the
process group exists (capture_error_mode="thread_local", pipeline.GraphedBody) and replays it, the images go through `dist.gather_images`
(ncclAllGather).  The latents must be bitwise those of the same call made before the process group existed.
Started as a child process by tests/test_pipeline_gpu.py::test_rccl_group_and_graph_capture_in_one_process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from lightdiffusion_amd import dist as D
from lightdiffusion_amd import nodes

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model, clip, vae = nodes.load_synthetic(dev, max_batch=2, max_hw=(16, 16), tiny=True)
toks = [[(49406, 1.0)] + [(1000 + i, 1.0) for i in range(20)] + [(49407, 1.0)] * 56]
neg = [[(49406, 1.0)] + [(49407, 1.0)] * 76]
enc = lambda t: clip.encode_from_tokens(t, return_pooled=False)
pc, nc = enc(toks), enc(neg)
lat = nodes.EmptyLatentImage().generate(128, 96, 2)[0]
unet = model.model.diffusion_model


def run(pc, nc):
    pos, ng = [[pc, {"pooled_output": None}]], [[nc, {"pooled_output": None}]]
    return nodes.KSampler2().sample(model, 1234, 6, 7.5, "euler_ancestral", "normal", pos, ng, lat)[0]["samples"]


before = run(pc, nc)                                  # no process group yet
assert any(d._graph is not None for d in unet._denoisers.values())
unet._denoisers.clear()                               # force a fresh capture below

dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shapes = [tuple(pc.shape), tuple(nc.shape)]
pc2, nc2 = D.broadcast_conditioning([pc, nc], shapes, src=0)          # ncclBroadcast on the device
assert pc2.is_cuda and torch.equal(pc2.cpu(), pc.cpu().float()) and torch.equal(nc2.cpu(), nc.cpu().float())
after = run(pc2.cpu(), nc2.cpu())                     # captures the step graph with the RCCL watchdog thread alive, then replays it
assert any(d._graph is not None for d in unet._denoisers.values()), "no graph was captured"
assert torch.equal(before, after), "latents differ with the process group alive"
# the raw wrapper hook captures its own graphs (pipeline.HookRunner) under the same conditions
x = torch.randn(1, 4, 12, 16, generator=torch.Generator().manual_seed(3)).to(dev)
ctx = torch.cat([nc2, pc2]).to(dev)
hook = lambda: unet(None, {"input": torch.cat([x, x]), "timestep": torch.full((2,), 2.5, device=dev),
                           "c": {"c_crossattn": ctx, "transformer_options": {"cond_or_uncond": [1, 0]}}, "cond_or_uncond": [1, 0]})
h1, h2 = hook(), hook()
assert torch.equal(h1, h2) and unet._hook[(2, 12, 16)].pair.graph is not None
images = vae.decode(after)
img_dev = (images * 255).round().to(torch.uint8).to(dev)
got = D.gather_images(img_dev)                        # ncclAllGather (counts + padded images)
assert torch.equal(got.cpu(), img_dev.cpu())
dist.barrier()
torch.cuda.synchronize()
print(f"rccl world-1 + hipGraph ok: backend {dist.get_backend()}, latents bitwise equal with and without the process group, "
      f"{len(unet._denoisers)} step graph(s) + hook graph captured under a live process group, gathered {tuple(got.shape)} uint8")
dist.destroy_process_group()

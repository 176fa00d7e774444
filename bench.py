#!/usr/bin/env python3
"""bench.py — UNet denoise steps/sec at 512x512 on N MI355X (BASELINE.json metric).

A "step" is one sampler iteration = one classifier-free-guided UNet forward on 2B samples ([uncond, cond], LD.py:2515)
+ the guidance mix + the sampler update, on synthetic latents with random-init SD1.5 weights (no checkpoints offline).
Default workload = BASELINE.json configs[1]: SD1.5 512x512, DPM++ 2M (dpmpp_2m_sde eta=0) / karras-20, batch 1, fp16
storage with fp32 accumulation.  `--batch 8 --sampler euler_ancestral` is configs[2].
Multi-GPU (torchrun, one rank per GPU): each rank holds a weight replica and runs its own B-image loop (weak scaling);
the only collective is the RCCL broadcast of the CLIP conditioning before the loop.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: the box allows ~16 cores per GPU; os.cpu_count() reports the whole host."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("LD_BENCH_CPU_THREADS", "16"))))


MFMA_PEAK_F16 = 2.5e15      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU (UNet batch is 2x this under CFG)")
    ap.add_argument("--sampler", default=None, choices=[None, "dpmpp_2m", "euler_ancestral"])
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512x512 px)")
    ap.add_argument("--cfg", type=float, default=7.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))      # (rehearsals put several ranks on one GPU; the driver gives one each)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LD_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from lightdiffusion_amd import ops, sampling
    from lightdiffusion_amd import weights as W
    from lightdiffusion_amd.pipeline import CFGDenoiser
    from lightdiffusion_amd.unet import synthetic_unet

    B, L = args.batch, args.latent
    sampler = args.sampler or ("dpmpp_2m" if B == 1 else "euler_ancestral")
    cfg = W.sd15_unet_config()
    t0 = time.time()
    torch.set_num_threads(host_cores())
    log(f"rank {rank}/{world}: building SD1.5 UNet (synthetic weights) on {dev}")
    unet = synthetic_unet(cfg, max_batch=2 * B, max_hw=(L, L), device=dev)
    t_load = time.time() - t0
    log(f"weights resident in {t_load:.1f}s: {unet.weight_bytes / 2**20:.0f} MiB weights, {unet.workspace_bytes / 2**20:.0f} MiB workspace")

    # ---- conditioning: rank 0 "encodes", RCCL broadcast over xGMI to the other ranks (the only collective)
    g = torch.Generator().manual_seed(1234)
    cond2 = torch.randn(2, 77, cfg["context_dim"], generator=g).to(dev) if rank == 0 else torch.empty(2, 77, cfg["context_dim"], device=dev)
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize()
        tb = time.time()
        dist.broadcast(cond2, src=0)
        torch.cuda.synchronize()
        t_bcast = time.time() - tb
    den_fn = CFGDenoiser(unet, B, L, L, args.cfg, use_graph=not args.no_graph)
    den_fn.set_context(cond2[0:1], cond2[1:2])

    # ---- schedule + per-step update coefficients (host scalars, as the reference computes them)
    ms = sampling.ModelSampling()
    if sampler == "dpmpp_2m":
        sig = sampling.calculate_sigmas(ms, "karras", 20)
    else:
        sig = sampling.calculate_sigmas(ms, "normal", 30)
    nstep = len(sig) - 2                       # positions with sigma_next > 0 (the last position is just x = denoised)
    gen = torch.Generator().manual_seed(rank)
    x = (torch.randn(B, 4, L, L, generator=gen) * float(sig[0])).to(dev)
    noise = torch.randn(B, 4, L, L, generator=gen).to(dev)
    old = torch.zeros_like(x)
    state = {"h_last": None}

    def step(i):
        p = i % nstep
        if p == 0 and i > 0:
            ops.axpby_(x, 0.0, noise, float(sig[0]))          # wrap: re-noise to sigma_max so the data stays in range
            state["h_last"] = None
        s, sn = sig[p], sig[p + 1]
        den = den_fn(x, float(s))
        if sampler == "euler_ancestral":
            sd_, su = sampling.get_ancestral_step(float(s), float(sn))
            r = (sd_ - float(s)) / float(s)
            ops.axpby_(x, 1.0 + r, den, -r, noise, su)
        else:
            h = (-sn.log()) - (-s.log())
            a, c1 = float(sn / s), float((-h).expm1().neg())
            if state["h_last"] is None:
                ops.axpby_(x, a, den, c1)
            else:
                k = float(0.5 * (-h).expm1().neg() * (h / state["h_last"]))
                ops.axpby_(x, a, den, c1 + k, old, -k)
            old.copy_(den)
            state["h_last"] = h

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    log("warm-up done (graph captured)" if not args.no_graph else "warm-up done (eager)")
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    finite = bool(torch.isfinite(x).all())
    log(f"timed {args.steps} steps in {elapsed:.3f}s -> {world * args.steps / elapsed:.2f} steps/s")

    # ---- per-kernel-class device time: HIP events around every launch of one forward (after the timed region)
    prof = None
    for _ in range(3):
        p = unet.profile(den_fn.x2, den_fn.sigma2)
        prof = p if prof is None else {k: (prof[k][0] + v[0], v[1], v[2]) for k, v in p.items()}
    prof = {k: (v[0] / 3.0, v[1], v[2]) for k, v in prof.items()}
    dom = max(("conv3x3", "gemm", "attention"), key=lambda k: prof[k][0])
    d_ms, d_fl, d_n = prof[dom]
    step_flops = unet.last_flops

    out = {
        "metric": "UNet denoise steps/sec at 512x512",
        "value": world * args.steps / elapsed,
        "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate)", "data": "synthetic latents + random-init SD1.5 weights",
        "config": {"workload": f"SD1.5 512x512 UNet CFG step, {sampler}, batch {B}/GPU (UNet batch {2 * B})", "latent": [L, L],
                   "global_batch": B * world, "parallelism": f"dp{world} (replicas, RCCL cond broadcast)", "hip_graph": not args.no_graph},
        "unet_evals_per_s": world * args.steps * 2 * B / elapsed,
        "step_tflops": step_flops / 1e12,
        "mfma_frac_whole_step": step_flops * args.steps / elapsed / MFMA_PEAK_F16,
        "launches_per_forward": unet.last_launches,
        "cond_broadcast_ms": 1e3 * t_bcast,
        "weights_load_s": t_load, "finite": finite,
        "kernel_ms_per_forward": {k: round(v[0], 4) for k, v in prof.items()},
        "roofline": {"kernel": f"gemm3_kernel / gemm4_kernel family, launch class '{dom}'", "bound": "mfma", "achieved": d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0,
                     "peak": MFMA_PEAK_F16 / 1e12, "unit": "TFLOP/s",
                     "frac": (d_fl / (d_ms * 1e-3)) / MFMA_PEAK_F16 if d_ms > 0 else 0.0, "traffic": None,
                     "launches": d_n, "avg_launch_us": 1e3 * d_ms / max(d_n, 1), "flops_per_launch": d_fl / max(d_n, 1)},
    }

    # ---- CPU baseline: the oracle (a port of the reference's CPU path) on this host's cores; rank 0, single-GPU runs only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import sd15_ref as O
        ncpu = host_cores()
        torch.set_num_threads(ncpu)
        log(f"cpu baseline: generating fp32 weights, {ncpu} threads")
        sd = W.synth_state_dict(W.unet_param_shapes(cfg))
        log("cpu baseline: running")
        oms = O.ModelSampling()
        xc = torch.randn(1, 4, L, L) * 5.0
        ctx = cond2.cpu()
        den_cpu = lambda xx, ss, cc: O.apply_model(sd, cfg, oms, xx, ss, cc)
        with torch.no_grad():
            O.sampling_function(den_cpu, xc, torch.tensor([5.0]), ctx[1:2], ctx[0:1], args.cfg)        # warm-up
            tc = time.perf_counter()
            for j in range(args.cpu_steps):
                O.sampling_function(den_cpu, xc, torch.tensor([5.0]), ctx[1:2], ctx[0:1], args.cfg)
                log(f"cpu baseline: step {j + 1}/{args.cpu_steps}")
            tc = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": args.cpu_steps / tc, "unit": "steps/s", "cores": ncpu, "kind": "port",
                               "sample": f"{args.cpu_steps} CFG steps at batch 1 (UNet batch 2), 64x64 latent, fp32 torch CPU ops"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

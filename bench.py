#!/usr/bin/env python3
"""bench.py — UNet denoise steps/sec at 512x512 (batch 8 and batch 1) on N MI355X (BASELINE.json metric).

A "step" is one sampler iteration = one classifier-free-guided UNet forward on 2B samples ([uncond, cond], LD.py:2515)
+ the guidance mix + the sampler update, on synthetic latents with random-init SD1.5 weights (no checkpoints offline).
What is timed is the PRODUCT loop: `sampling.sample()` -> `KSAMPLER.sample` -> `sample_euler_ancestral` / `sample_dpmpp_2m_sde`
-> `CFGGuider` -> `sampling_function` -> the cached hipGraph denoiser (the call stack `KSampler2.sample` runs), in whole
sampler passes until exactly K steps have been taken.

One invocation reports
  * headline `value`: BASELINE configs[2] — SD1.5 512x512, Euler-a / normal-30, batch 8 per GPU (UNet batch 16);
  * `batch1`: BASELINE configs[1] — DPM++ 2M (dpmpp_2m_sde eta=0) / karras-20, batch 1;
  * (N = 1 only) `hires`: configs[4] — the 128x128-latent Euler-a step at batch 4 and the 1024^2 VAE decode;
    `images_per_s_e2e`: 20-step txt2img at batch 8 including the VAE decode; `cpu_baseline`: the oracle on the host cores.
each with its own `roofline` for the dominant kernel instantiation (HIP events around every launch, on the launch stream).

Multi-GPU: one rank per GPU (torchrun, or `--gpus N` alone, which starts N rank processes itself), a weight replica and an
independent B-image loop per rank (weak scaling); the only collective is the RCCL broadcast of the conditioning.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_F16 = 2.5e15      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: the box allows ~16 cores per GPU; os.cpu_count() reports the whole host."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("LD_BENCH_CPU_THREADS", "16"))))


def lib_hash():
    """sha256 (first 16 hex digits) of the HIP library this process loads: ties PMC traffic numbers to the kernels that are timed."""
    import hashlib
    from lightdiffusion_amd._lib import LIB_PATH
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def launch_min_bytes(what, dims):
    """Minimal HBM bytes of one launch (fp16 unless noted): every input and the weights read once, the output written once.
    Contractions: the RAW input pixels x Cin (not the im2col matrix) + W + C;  attention: Q + K + V^T + O;  GroupNorm: read + write."""
    a, b, c, d = dims
    if what in ("conv3", "conv1", "gemm", "geglu"):
        cin = c // 9 if what == "conv3" else c
        n_out = b // 2 if what == "geglu" else b
        return 2.0 * d * (a * cin + b * c + a * n_out)
    if what == "attention":        # (batch*heads, Lq, Lk, d)
        return 2.0 * a * (2 * b * d + 2 * c * d)
    if what in ("groupnorm", "gn_stats"):   # (images, pixels, channels, silu)
        return 2.0 * a * b * c * (2 if what == "groupnorm" else 1)
    if what == "conv_in":
        return 2.0 * a * b + 4.0 * a * 4
    if what == "conv_out":
        return 2.0 * a * c / 9 + 4.0 * a * b
    if what == "softmax":
        return 4.0 * a * b
    return 0.0


RIDGE = 2.5e15 / 8.0e12      # FLOP per HBM byte above which a launch is MFMA-bound (~315)


def rooflines(rows, ms_by_kernel, traffic_for):
    """rows: per-launch (what, dims, flops, us, kernel) of one profiled run; ms_by_kernel: averaged ms per kernel name.
    -> (roofline of the kernel that takes the most time, roofline of the dominant HBM-bound kernel or None).  A kernel is priced
    against the roof its arithmetic intensity (algorithmic FLOPs / minimal bytes) puts it under."""
    agg = {}
    for what, dims, fl, us, kern in rows:
        a = agg.setdefault(kern, {"flops": 0.0, "bytes": 0.0, "n": 0})
        a["flops"] += fl
        a["bytes"] += launch_min_bytes(what, dims)
        a["n"] += 1

    def one(kern):
        a = agg[kern]
        ms = ms_by_kernel.get(kern)
        if not ms or a["bytes"] <= 0:
            return None
        hbm = a["flops"] / a["bytes"] < RIDGE
        tr, tr_d, tr_from = traffic_for(kern)
        out = {"kernel": kern, "bound": "hbm" if hbm else "mfma", "launches": a["n"], "avg_launch_us": 1e3 * ms / a["n"],
               "flops_per_launch": a["flops"] / a["n"], "min_bytes_per_launch": a["bytes"] / a["n"],
               "intensity_flop_per_byte": a["flops"] / a["bytes"], "traffic": tr, "traffic_detail": tr_d, "traffic_from": tr_from}
        if hbm:
            ach = a["bytes"] / (ms * 1e-3)
            out.update(achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK,
                       also_tflops=a["flops"] / (ms * 1e-3) / 1e12)
        else:
            ach = a["flops"] / (ms * 1e-3)
            out.update(achieved=ach / 1e12, peak=MFMA_PEAK_F16 / 1e12, unit="TFLOP/s", frac=ach / MFMA_PEAK_F16)
        return out

    timed = [k for k in agg if ms_by_kernel.get(k)]
    if not timed:
        return None, None
    dom = max(timed, key=lambda k: ms_by_kernel[k])
    hb = [k for k in timed if agg[k]["bytes"] > 0 and agg[k]["flops"] / agg[k]["bytes"] < RIDGE and agg[k]["flops"] > 0]
    dom_h = max(hb, key=lambda k: ms_by_kernel[k]) if hb else None
    return one(dom), (one(dom_h) if dom_h else None)


def spawn_ranks(n, argv):
    """`--gpus N` without a torchrun environment: start N fresh rank processes (this process never touches the GPU)."""
    port = 29400 + os.getpid() % 400
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps K of the headline (batch 8) and of the batch-1 run")
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU of the headline run (UNet batch is 2x this under CFG)")
    ap.add_argument("--cfg", type=float, default=7.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip batch-1 / hires / e2e / cpu baseline (profiling runs)")
    ap.add_argument("--only", default=None, choices=[None, "batch8", "batch1", "hires"], help="time just one workload (profiling runs)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-prime", action="store_true", help="skip the set-up pass (graph capture + one schedule pass) before the warm-up steps")
    ap.add_argument("--reps", type=int, default=5, help="extra one-pass brackets after the timed region (median ms/step); 0 for profiling runs")
    ap.add_argument("--sync-steps", action="store_true",
                    help="synchronize after every sampler step: bounds the host's run-ahead (rocprofv3 --pmc passes: profiles/README.md round 3)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {args.gpus}, "
                         f"or pass --gpus alone and let bench.py start the ranks")

    dbg = os.environ.get("LD_BENCH_DEBUG")          # <path prefix>: fault / hang tracebacks and /proc/self/maps for post-mortems under a profiler
    if dbg:
        import faulthandler
        fh = open(dbg + ".fault", "w")
        faulthandler.enable(file=fh, all_threads=True)
        faulthandler.dump_traceback_later(int(os.environ.get("LD_BENCH_DEBUG_HANG_S", "60")), repeat=True, file=fh)

    def dump_maps(tag):
        if dbg:
            with open("/proc/self/maps") as src, open(f"{dbg}.maps", "w") as dst:
                dst.write(f"# {tag}\n" + src.read())

    import torch
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

    from lightdiffusion_amd import nodes, sampling
    from lightdiffusion_amd import weights as W
    from lightdiffusion_amd.unet import synthetic_unet, synthetic_vae

    cfg = W.sd15_unet_config()
    torch.set_num_threads(host_cores())
    t0 = time.time()
    log(f"rank {rank}/{world}: building SD1.5 UNet (synthetic weights) on {dev}")
    unet = synthetic_unet(cfg, max_batch=2 * args.batch, max_hw=(64, 64), device=dev)
    t_load = time.time() - t0
    log(f"weights resident in {t_load:.1f}s: {unet.weight_bytes / 2**20:.0f} MiB weights, {unet.workspace_bytes / 2**20:.0f} MiB workspace")
    dump_maps("weights resident")
    model = nodes._attach(unet, dev)                      # ModelPatcher + set_model_unet_function_wrapper: the reference's plugin seam
    if args.no_graph:
        model.model_options["ld_use_graph"] = False

    # ---- conditioning: rank 0 "encodes", ONE RCCL broadcast over xGMI to the other ranks (the only collective)
    g = torch.Generator().manual_seed(1234)
    cond2 = torch.randn(2, 77, cfg["context_dim"], generator=g).to(dev) if rank == 0 else torch.empty(2, 77, cfg["context_dim"], device=dev)
    t_bcast = 0.0
    if world > 1:
        torch.cuda.synchronize()
        tb = time.time()
        dist.broadcast(cond2, src=0)
        torch.cuda.synchronize()
        t_bcast = time.time() - tb
    cond_cpu = cond2.cpu()
    pos, neg = [[cond_cpu[1:2], {"pooled_output": None}]], [[cond_cpu[0:1], {"pooled_output": None}]]
    ms = sampling.ModelSampling()
    traffic_db = {}
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tp):
        traffic_db = json.load(open(tp))
    my_lib = lib_hash()

    def traffic_lookup(tag):
        """PMC bytes per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, rocprofv3 --pmc passes of THIS bench.py, tools/prof_run.sh) — only
        when the pass was taken with the library that is loaded now (`lib` hash in the tag); otherwise null + where it would come from."""
        t = traffic_db.get(tag) or {}
        ok = t.get("_lib") == my_lib

        def f(kern):
            d = t.get(kern.split("+")[0])
            if not isinstance(d, dict):
                return None, None, None
            src = f"profiles/pmc_traffic.json[{tag}] lib {t.get('_lib')}" + ("" if ok else f" (stale: loaded library is {my_lib})")
            return (d.get("hbm_bytes_per_launch"), d, src) if ok else (None, None, src)
        return f

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(tag, B, L, sampler, scheduler, sched_steps, denoise, K, Wm, cfg_scale):
        """Time exactly K steps of the product sampler loop (whole passes over the schedule, then a partial one)."""
        ks = sampling.KSampler1(model, steps=sched_steps, device=dev, sampler=sampler, scheduler=scheduler, denoise=denoise,
                                model_options=model.model_options)
        sig = ks.sigmas
        run_len = len(sig) - 1
        opts = {"eta": 0.0} if sampler == "dpmpp_2m_sde" else {}
        gen = torch.Generator().manual_seed(1000 + rank)
        latent = torch.zeros(B, 4, L, L) if denoise is None else torch.randn(B, 4, L, L, generator=gen) * 0.8
        noise = torch.randn(B, 4, L, L, generator=gen)

        cb = (lambda d: torch.cuda.synchronize()) if args.sync_steps else None

        def passes(nsteps):
            done = 0
            while done < nsteps:
                n = min(run_len, nsteps - done)
                sampling.sample(model, noise, pos, neg, cfg_scale, dev, sampling.ksampler(sampler, opts), sig[: n + 1], model.model_options,
                                latent_image=latent, callback=cb, seed=rank)
                done += n
                dump_maps(f"{tag}: {done} steps")

        # set-up, before the W warm-up steps (like loading the weights): the first call of a shape plans the workspace and captures
        # the step's hipGraph; one schedule pass brings the clocks out of idle
        if not args.no_prime:
            for _ in range(int(os.environ.get("LD_BENCH_PRIME_PASSES", "1"))):      # (experiments: how long the clocks take to settle)
                passes(run_len)
            torch.cuda.synchronize()
        passes(Wm)
        barrier()
        t0 = time.perf_counter()
        passes(K)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.barrier()
            elapsed = float(t.item())
        # repetitions (after the contract's bracket): 5 more brackets of one schedule pass each -> median ms/step
        reps = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            passes(run_len)
            torch.cuda.synchronize()
            reps.append(1e3 * (time.perf_counter() - t1) / run_len)
        # ---- per-kernel device time: HIP events around every launch of one forward, on the launch stream
        den = next(iter(d for k, d in unet._denoisers.items() if k[0] == B and k[1] == L))
        agg = {}
        cls = None
        for _ in range(3):
            c = unet.profile_pair(den.x1, den.sigma1)                 # the route the timed loop runs (ld_unet_forward_pair)
            cls = c if cls is None else {k: (cls[k][0] + v[0], v[1], v[2]) for k, v in c.items()}
            for name, (msv, fl, nl) in unet.profile_kernels().items():
                a = agg.get(name, (0.0, fl, nl))
                agg[name] = (a[0] + msv, fl, nl)
        agg = {k: (v[0] / 3.0, v[1], v[2]) for k, v in agg.items()}
        cls = {k: (v[0] / 3.0, v[1], v[2]) for k, v in cls.items()}
        rows = unet.profile_launches()                              # per launch of the last profiled forward: shapes, FLOPs, kernel
        roof, roof_hbm = rooflines(rows, {k: v[0] for k, v in agg.items()}, traffic_lookup(tag))
        step_flops = unet.last_flops
        res = {
            "workload": f"SD1.5 512x512 UNet CFG step, {sampler} / {scheduler}-{sched_steps}, batch {B}/GPU (UNet batch {2 * B}), latent {L}x{L}",
            "steps_per_s": world * K / elapsed, "ms_per_step": 1e3 * elapsed / K, "steps": K,
            "ms_per_step_median_of_5_passes": statistics.median(reps) if reps else None, "ms_per_step_passes": [round(r, 4) for r in reps],
            "unet_evals_per_s": world * K * 2 * B / elapsed,
            "step_tflops": step_flops / 1e12,
            "mfma_frac_whole_step": step_flops * K / elapsed / MFMA_PEAK_F16,
            "launches_per_forward": unet.last_launches,
            "kernel_class_ms_per_forward": {k: round(v[0], 4) for k, v in cls.items()},
            "kernels_ms_per_forward": {k: [round(v[0], 4), v[2]] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]},
            "roofline": roof, "roofline_hbm": roof_hbm,
        }
        log(f"{tag}: {K} steps in {elapsed:.3f}s -> {res['steps_per_s']:.2f} steps/s ({res['unet_evals_per_s']:.0f} UNet-evals/s); "
            f"dominant {roof['kernel']} {roof['achieved']:.0f} {roof['unit']} ({roof['bound']} roof, {100 * roof['frac']:.1f} %)")
        return res

    def run_hook_route(B, L, K, Wm, cfg_scale, product_ms):
        """The reference's OWN seam, timed: what `calc_cond_batch` + `cfg_function` do per step around `model_function_wrapper`
        (LD.py:2515-2606) — cat([x, x]), cat([sigma, sigma]), a freshly concatenated [uncond, cond] context, the wrapper call with
        cond_or_uncond == [1, 0], chunk, uncond + (cond - uncond) * cfg, an Euler update — with `MI355XUNet.__call__` on the seam (graph
        replay + device-side guards, unet.py).  The dictionary has the keys and order of the arguments recorded from the reference in
        tests/golden/samplers.npz (hook_input / hook_timestep / hook_ctx / hook_cond_or_uncond), at this workload's size."""
        gen = torch.Generator().manual_seed(4321 + rank)
        x = (torch.randn(B, 4, L, L, generator=gen) * 10.0).to(dev)
        unc = cond2[0:1].expand(B, -1, -1).contiguous()
        cnd = cond2[1:2].expand(B, -1, -1).contiguous()
        sig = sampling.KSampler1(model, steps=30, device=dev, sampler="euler_ancestral", scheduler="normal", denoise=None,
                                 model_options=model.model_options).sigmas.to(dev)
        ones = torch.ones(B, device=dev)
        state = {"x": x}

        def steps(n):
            xx = state["x"]
            for i in range(n):
                j = i % (len(sig) - 1)
                s = sig[j] * ones
                cou = [1, 0]
                c = {"c_crossattn": torch.cat([unc, cnd]), "transformer_options": {"cond_or_uncond": cou[:], "sigmas": s}}
                out = unet(None, {"input": torch.cat([xx, xx]), "timestep": torch.cat([s, s]), "c": c, "cond_or_uncond": cou})
                u, cn = out.chunk(2)
                den = u + (cn - u) * cfg_scale
                xx = xx + (xx - den) / sig[j] * (sig[j + 1] - sig[j])
            state["x"] = xx

        steps(max(Wm, 3))
        barrier()
        t0 = time.perf_counter()
        steps(K)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        run = unet._hook.get((2 * B, L, L))
        return {"what": "MI355XUNet.__call__ on model_function_wrapper (LD.py:2558-2567) driven as calc_cond_batch + cfg_function drive it: "
                        "fresh cat([x, x]) / cat([uncond, cond]) per step, torch CFG mix + Euler update",
                "batch": B, "steps": K, "steps_per_s": K / el, "ms_per_step": 1e3 * el / K,
                "route": "pair graph" if run is not None and run.pair is not None and run.pair.graph is not None and not run.halves_differed else "plain graph",
                "product_loop_ms_per_step": product_ms, "vs_product_loop": (1e3 * el / K) / product_ms if product_ms else None}

    K, Wm = args.steps, args.warmup
    out_extra = {}
    head = None
    if args.only in (None, "batch8"):
        head = run_workload("batch8", args.batch, 64, "euler_ancestral", "normal", 30, None, K, Wm, args.cfg)
    if (args.only is None and not args.no_extras) or args.only == "batch1":
        out_extra["batch1"] = run_workload("batch1", 1, 64, "dpmpp_2m_sde", "karras", 20, None, K, min(Wm, 20), args.cfg)
    if ((args.only is None and not args.no_extras) and world == 1) or args.only == "hires":
        # config #5: hires-fix second pass — [4,4,128,128], 10 Euler-a steps from sigma 1.2768 (normal, denoise 0.45, cfg 8) + VAE -> 1024^2
        out_extra["hires"] = run_workload("hires", 4, 128, "euler_ancestral", "normal", 10, 0.45, 20, 10, 8.0)
    if head is None:
        head = out_extra.get("batch1") or out_extra.get("hires")

    if args.only is None and not args.no_extras and world == 1:
        out_extra["hook_route"] = {"batch8": run_hook_route(args.batch, 64, K, Wm, args.cfg, head["ms_per_step"]),
                                   "batch1": run_hook_route(1, 64, K, min(Wm, 20), args.cfg, out_extra["batch1"]["ms_per_step"])}
        log(f"hook route: B={args.batch} {out_extra['hook_route']['batch8']['ms_per_step']:.3f} ms/step "
            f"({out_extra['hook_route']['batch8']['vs_product_loop']:.3f}x the product loop), "
            f"B=1 {out_extra['hook_route']['batch1']['ms_per_step']:.3f} ms/step ({out_extra['hook_route']['batch1']['vs_product_loop']:.3f}x)")
        vae = synthetic_vae(W.sd15_vae_config(), max_batch=args.batch, max_hw=(64, 64), device=dev)
        lat = torch.randn(args.batch, 4, 64, 64, generator=torch.Generator().manual_seed(5)) * 4.0

        def timed(fn, n=3):
            fn()
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return statistics.median(ts)

        def vae_roofs(v, z, tag):
            """dominant kernels of one decode (HIP events per launch, median of 3 profiled decodes), priced like the UNet's"""
            runs = [v.profile_decode(z) for _ in range(3)]
            ms = {}
            for i, r in enumerate(runs[0]):
                ms[r[4]] = ms.get(r[4], 0.0) + statistics.median(rr[i][3] for rr in runs) * 1e-3
            return rooflines(runs[0], ms, traffic_lookup(tag))

        t_vae = timed(lambda: vae.decode_device(lat))
        r_m, r_h = vae_roofs(vae, lat, "vae512")
        out_extra["vae_decode_512"] = {"batch": args.batch, "ms": 1e3 * t_vae, "images_per_s": args.batch / t_vae,
                                       "tflops": vae.last_flops / t_vae / 1e12, "launches": vae.last_launches,
                                       "roofline": r_m, "roofline_hbm": r_h}
        # end to end: 20-step DPM++ 2M txt2img at batch B through the node-level sampler + VAE decode (CLIP not included:
        # the conditioning is synthetic), latents stay on the device between the two
        ks20 = sampling.KSampler1(model, steps=20, device=dev, sampler="dpmpp_2m_sde", scheduler="karras", denoise=None,
                                  model_options=model.model_options)
        z0 = torch.zeros(args.batch, 4, 64, 64)
        nz = torch.randn(args.batch, 4, 64, 64, generator=torch.Generator().manual_seed(6))

        def e2e():
            s = sampling.sample(model, nz, pos, neg, args.cfg, dev, sampling.ksampler("dpmpp_2m_sde", {"eta": 0.0}), ks20.sigmas,
                                model.model_options, latent_image=z0, seed=0)
            return vae.decode_device(s)
        t_e2e = timed(e2e)
        out_extra["images_per_s_e2e"] = {"value": args.batch / t_e2e, "batch": args.batch, "seconds": t_e2e,
                                         "what": "20-step DPM++ 2M / karras txt2img 512x512 (sampler loop + VAE decode; synthetic conditioning)"}
        del vae
        vae = synthetic_vae(W.sd15_vae_config(), max_batch=4, max_hw=(128, 128), device=dev)
        lat = torch.randn(4, 4, 128, 128, generator=torch.Generator().manual_seed(7)) * 4.0
        t_v2 = timed(lambda: vae.decode_device(lat), n=2)
        r_m, r_h = vae_roofs(vae, lat, "vae1024")
        out_extra["vae_decode_1024"] = {"batch": 4, "ms": 1e3 * t_v2, "images_per_s": 4 / t_v2, "tflops": vae.last_flops / t_v2 / 1e12,
                                        "roofline": r_m, "roofline_hbm": r_h}
        del vae
        torch.cuda.empty_cache()

    out = {
        "metric": "UNet denoise steps/sec at 512x512",
        "value": head["steps_per_s"],
        "unit": "steps/s",
        "n_gpus": world, "steps": head["steps"], "warmup": Wm,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate)", "data": "synthetic latents + random-init SD1.5 weights",
        "config": {"workload": head["workload"], "global_batch": args.batch * world,
                   "parallelism": f"dp{world} (replicas, RCCL cond broadcast)", "hip_graph": not args.no_graph,
                   "setup_before_warmup": None if args.no_prime else "hipGraph capture + one untimed schedule pass per workload",
                   "library_sha256_16": my_lib,
                   "cfg_pair": "ld_unet_forward_pair: the layers in front of the first cross-attention are evaluated once for the [uncond, cond] halves "
                               "of a step (same latents); step_tflops counts the FLOPs that are executed",
                   "timed_loop": "lightdiffusion_amd.sampling.sample (product call surface)"},
        "unet_evals_per_s": head["unet_evals_per_s"],
        "step_tflops": head["step_tflops"], "mfma_frac_whole_step": head["mfma_frac_whole_step"],
        "ms_per_step_median_of_5_passes": head["ms_per_step_median_of_5_passes"],
        "launches_per_forward": head["launches_per_forward"],
        "cond_broadcast_ms": 1e3 * t_bcast, "weights_load_s": t_load,
        "kernel_class_ms_per_forward": head["kernel_class_ms_per_forward"], "kernels_ms_per_forward": head["kernels_ms_per_forward"],
        "roofline": head["roofline"], "roofline_hbm": head.get("roofline_hbm"),
    }
    out.update(out_extra)

    # ---- CPU baseline: the oracle (a port of the reference's CPU path) on this host's cores; rank 0, single-GPU runs only
    if rank == 0 and world == 1 and args.only is None and not args.no_extras:
        from oracle import sd15_ref as O
        ncpu = host_cores()
        torch.set_num_threads(ncpu)
        log(f"cpu baseline: generating fp32 weights, {ncpu} threads")
        sd = W.synth_state_dict(W.unet_param_shapes(cfg))
        log("cpu baseline: running")
        oms = O.ModelSampling()
        xc = torch.randn(1, 4, 64, 64) * 5.0
        den_cpu = lambda xx, ss, cc: O.apply_model(sd, cfg, oms, xx, ss, cc)
        with torch.no_grad():
            O.sampling_function(den_cpu, xc, torch.tensor([5.0]), cond_cpu[1:2], cond_cpu[0:1], args.cfg)        # warm-up
            tc = time.perf_counter()
            for j in range(args.cpu_steps):
                O.sampling_function(den_cpu, xc, torch.tensor([5.0]), cond_cpu[1:2], cond_cpu[0:1], args.cfg)
                log(f"cpu baseline: step {j + 1}/{args.cpu_steps}")
            tc = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": args.cpu_steps / tc, "unit": "steps/s", "cores": ncpu, "kind": "port",
                               "sample": f"{args.cpu_steps} CFG steps at batch 1 (UNet batch 2), 64x64 latent, fp32 torch CPU ops "
                                         f"(x8 the work per step at the headline's batch 8)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

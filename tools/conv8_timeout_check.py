"""conv8's in-launch slab reduction when no peer workgroup is resident (a shared GPU, a partitioned one): with LD_C8_NO_WAIT=1 the A/B build
makes every workgroup but the tile's last arriver leave at once, so the last arriver claims and sums all parts alone — the route a timed-out
wait takes.  The slabs are summed in slab order whoever sums them, so the result must be BITWISE the normal one; the counters must be left zero
(second call).  Usage: python tools/conv8_timeout_check.py   (loads lightdiffusion_amd/libld_mi355x_ab.so twice: child processes, one per mode)"""
import os, subprocess, sys, hashlib
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import math, torch
    sys.path.insert(0, '.')
    from lightdiffusion_amd import ops
    def r16(shape, seed, scale=1.0):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(shape, generator=g) * scale).half()
    h = hashlib.sha256()
    for (n, hw, cin, cout) in [(2, 16, 1280, 1280), (2, 8, 2560, 1280), (2, 32, 640, 640), (2, 64, 320, 320), (2, 8, 1280, 640)]:
        x = r16((n, hw, hw, cin), 1).cuda(); w = ops.repack_conv_weight(r16((cout, cin, 3, 3), 2, 1 / math.sqrt(9 * cin)).cuda())
        b = r16((cout,), 3, 0.1).cuda(); r = r16((n, hw, hw, cout), 4).cuda()
        for rep in range(3):                                   # (replays: the counters are left zeroed on both routes)
            y, part = ops.conv2d_gn_partials(x, w, b, None, r)
            torch.cuda.synchronize()
            h.update(y.cpu().numpy().tobytes()); h.update(part.cpu().numpy().tobytes())
    print("DIGEST", h.hexdigest())
    sys.exit(0)
env = dict(os.environ, LD_MI355X_LIB=os.path.join("lightdiffusion_amd", "libld_mi355x_ab.so"))
out = []
for mode in (None, "1"):
    e = dict(env)
    if mode: e["LD_C8_NO_WAIT"] = mode
    r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
    d = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
    if r.returncode or not d:
        print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
    out.append(d[0])
    print(("normal route:            " if not mode else "last arriver sums alone: ") + d[0])
print("identical" if out[0] == out[1] else "DIFFERENT")
sys.exit(0 if out[0] == out[1] else 1)

"""Summarise rocprofv3 output directories into the small tracked files under profiles/ (run after a gpurun profiling call).

  python tools/prof_collect.py stats <rocprof dir> profiles/<name>_kernel_stats.csv
  python tools/prof_collect.py pmc <tag> profiles/<name>_pmc.csv <rocprof dir> [<rocprof dir> ...]

`stats`: copies the largest *_kernel_stats.csv (the process that ran the GPU work).
`pmc`:   per kernel (short names as bench.py prints them) the mean of every collected counter per dispatch, and rewrites
         profiles/pmc_traffic.json[tag][kernel] = {"fetch_bytes", "write_bytes", ...} with the gfx950 FETCH_SIZE correction
         (x2: the counter tallies 128-byte requests as 64 bytes, MI355X_MICROARCH.md 'HBM'); FETCH_SIZE / WRITE_SIZE are in KiB."""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short_name(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm(\d)_kernel<(\d+), (\d+), (true|false)(?:, (\d+))?(?:, (\d+))?>", k)
    if m:   # (gemm4's trailing parameters: ring depth, waves per SIMD — 4 = the two-workgroups-per-CU instantiation gemm_launch names ",2wg")
        two = m.group(1) == "4" and m.group(6) == "4"
        return f"gemm{m.group(1)}_kernel<{m.group(2)},{m.group(3)},{'conv' if m.group(4) == 'true' else 'plain'}{',2wg' if two else ''}>"
    m = re.match(r"gemm5_kernel<(true|false)(?:, \d+)?>", k)
    if m:
        return f"gemm5_kernel<256,320,{'conv' if m.group(1) == 'true' else 'plain'}>"
    m = re.match(r"conv6_kernel<(\d+), (true|false)(?:, (\d+))?(?:, (\d+))?(?:, (true|false))?>", k)
    if m:   # the names gemm_launch gives its instantiations (bench.py prints those)
        w, gn, bn, bm, up = m.group(1), m.group(2) == "true", m.group(3) or "320", m.group(4) or "256", m.group(5) == "true"
        tail = ""
        if bm != "256":
            tail = f",{bn}x{bm}"
        elif bn != "320":
            tail = f",{bn}"
        return f"conv6_kernel<W{w},{'halo+groupnorm' if gn else 'halo'}{tail}{',up' if up else ''}>"
    m = re.match(r"gemm7_kernel<(true|false)(?:, (?:true|false))?>", k)
    if m:
        return f"gemm7_kernel<256,K320,{'geglu' if m.group(1) == 'true' else 'plain'}>"
    m = re.match(r"flash_attn2_kernel<(\d+), (true|false), (\d+), \d+(?:, (true|false))?>", k)
    if m:
        return f"flash_attn2_kernel<{m.group(1)}{',rowV' if m.group(4) == 'true' else ''},{'masked' if m.group(2) == 'true' else 'plain'},{m.group(3)}>"
    m = re.match(r"conv8_kernel<(\d+), (true|false)>", k)
    if m:
        return f"conv8_kernel<W{m.group(1)}{',up' if m.group(2) == 'true' else ''}>"
    return re.sub(r"\(.*$", "", k)


def biggest(d, pattern):
    files = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not files:
        raise SystemExit(f"no {pattern} under {d}")
    return max(files, key=os.path.getsize)


def main():
    mode = sys.argv[1]
    if mode == "stats":
        src = biggest(sys.argv[2], "*_kernel_stats.csv")
        shutil.copyfile(src, sys.argv[3])
        print("copied", src, "->", sys.argv[3])
        return
    tag, out = sys.argv[2], sys.argv[3]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[4:]:
        with open(biggest(d, "*_counter_collection.csv")) as f:
            for row in csv.DictReader(f):
                a = acc[short_name(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    with open(out, "w") as f:
        f.write("kernel,counter,mean_per_dispatch,dispatches\n")
        for k in sorted(acc):
            for c in sorted(acc[k]):
                f.write(f'"{k}",{c},{acc[k][c][0] / acc[k][c][1]:.6g},{acc[k][c][1]}\n')
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    db = json.load(open(tp)) if os.path.exists(tp) else {}
    t = db[tag] = {}   # a pass replaces the tag: kernel names change between rounds
    # provenance: the library the passes ran (bench.py only reports `traffic` when this matches the library it loads)
    import hashlib
    lib = os.environ.get("LD_MI355X_LIB") or os.path.join(ROOT, "lightdiffusion_amd", "libld_mi355x.so")
    t["_lib"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    t["_driver"] = os.environ.get("LD_PROF_DRIVER", "bench.py")
    for k in acc:
        e = {}
        if "FETCH_SIZE" in acc[k]:
            e["fetch_bytes"] = 2.0 * 1024.0 * acc[k]["FETCH_SIZE"][0] / acc[k]["FETCH_SIZE"][1]     # KiB, x2 (gfx950)
        if "WRITE_SIZE" in acc[k]:
            e["write_bytes"] = 1024.0 * acc[k]["WRITE_SIZE"][0] / acc[k]["WRITE_SIZE"][1]
        if e:
            e["hbm_bytes_per_launch"] = e.get("fetch_bytes", 0.0) + e.get("write_bytes", 0.0)
            t.setdefault(k, {}).update(e)
    json.dump(db, open(tp, "w"), indent=1, sort_keys=True)
    print("wrote", out, "and", tp)


if __name__ == "__main__":
    main()

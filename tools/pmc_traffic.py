#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    rocprofv3 -M --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o f -- python bench.py --eager ...
    rocprofv3 -M --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o w -- python bench.py --eager ...
    python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv > profiles/traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (128-byte requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-byte stores.
Output: {bench kernel id: bytes per launch}.  The conv kernel symbol is shared by forward and dgrad launches of one tile
shape, so `igemm_*_fwd` and `igemm_*_dgrad` of a tile carry the same per-launch average of that symbol."""
import csv, json, re, sys, collections

TILES = {(4, 4, 2, 2): "128x128", (4, 2, 2, 2): "128x64", (2, 2, 2, 2): "64x64", (2, 2, 4, 1): "128x32", (1, 1, 4, 1): "64x16",
         (1, 2, 2, 2): "32x64", (2, 1, 2, 2): "64x32"}


def ids(sym):
    m = re.search(r"igemm_kernelI(DF16b|f)Li(\d)ELi(\d)ELi(\d)ELi(\d)E", sym)
    if m:
        dt = "bf16" if m.group(1) == "DF16b" else "f32"
        t = TILES.get(tuple(int(m.group(i)) for i in range(2, 6)), "?")
        return [f"igemm_{dt}_{t}_fwd", f"igemm_{dt}_{t}_dgrad"]
    m = re.search(r"conv3x3_halo_kernelI(DF16b|f)Li\d+ELi\d+ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", sym) or \
        re.search(r"conv1x1_dma_kernelI(DF16b|f)Li(\d)ELi(\d)ELi(\d)ELi(\d)E", sym)
    if m:       # (the LDS-DMA kernels of conv3x3.hip report under the bench id of their tile shape, as the library's profiler does)
        dt = "bf16" if m.group(1) == "DF16b" else "f32"
        t = TILES.get(tuple(int(m.group(i)) for i in range(2, 6)), "?")
        return [f"igemm_{dt}_{t}_fwd", f"igemm_{dt}_{t}_dgrad"]
    if "wgrad_reduce" in sym:
        return ["wgrad_reduce"]
    if "wgrad" in sym:
        return ["wgrad_bf16" if "DF16b" in sym or "alltaps" in sym or "wgrad128" in sym or "wgrad_halo" in sym else "wgrad_f32"]
    if "reduce2_kernel" in sym:
        return ["bn_act_bwd_reduce" if "BwdRedF" in sym else "bn_stats_reduce"]
    if "ew2_kernel" in sym:
        return ["bn_act_bwd_apply" if "BwdApplyF" in sym else "bn_act_fwd"]
    if "lazy_ew_kernel" in sym:
        return ["bn_act_fwd"]
    return []


def load(path, counter):
    """{id: [launches, counter sum]}.  The grouped weight-gradient launches (per-tap 64- and 128-wide tiles, all-taps) are ONE unit of
    work per step for the library's profiler (`wgrad_bf16`, launches_per_step = 1): their per-symbol averages are ADDED."""
    per_sym = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        per_sym[r["Kernel_Name"]][0] += 1
        per_sym[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for sym, (n, v) in per_sym.items():
        for k in ids(sym):
            if k.startswith("wgrad_") and k != "wgrad_reduce":
                agg[k][0] = 1
                agg[k][1] += v / n
            else:
                agg[k][0] += n
                agg[k][1] += v
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fetch = 2.0 * 1024.0 * f[k][1] / max(f[k][0], 1)
        write = 1024.0 * w[k][1] / max(w[k][0], 1)
        out[k] = round(fetch + write)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

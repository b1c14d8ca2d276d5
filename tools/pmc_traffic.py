#!/usr/bin/env python3
"""HBM-side traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    rocprofv3 -M --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o f -- python bench.py --eager ...
    rocprofv3 -M --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o w -- python bench.py --eager ...
    python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv > profiles/traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (128-byte requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-byte stores.
Output: {bench kernel label without its direction: bytes per launch} -- a convolution symbol serves the forward and the data-gradient
launches of its tile shape, so "<family>/<dtype>/<tile>/fwd" and ".../dgrad" of bench.py look up the same per-launch average."""
import csv, json, re, sys, collections

def _tile(sym, pat, order):
    """label "<family>/<dtype>/<BM>x<BN>" from the mangled template arguments of a convolution kernel symbol."""
    m = re.search(pat, sym)
    if not m:
        return None
    dt = "bf16" if m.group(1) == "DF16b" else "f32"
    mi, ni, wgm, wgn = (int(m.group(i)) for i in order)
    return dt, f"{wgm * mi * 16}x{wgn * ni * 16}"


def ids(sym):
    """bench.py kernel labels (without the fwd / dgrad suffix: one symbol serves both directions) this symbol reports under."""
    I = r"Li(\d+)E"
    for fam, pat, order in (("igemm_kernel", r"igemm_kernelI(DF16b|f)" + I * 4, (2, 3, 4, 5)),
                            ("conv3x3_halo_kernel", r"conv3x3_halo_kernelI(DF16b|f)" + I * 6, (4, 5, 6, 7)),
                            ("conv1x1_dma_kernel", r"conv1x1_dma_kernelI(DF16b|f)" + I * 4, (2, 3, 4, 5)),
                            ("conv1x1_ws_kernel", r"conv1x1_ws_kernelI(DF16b|f)" + I * 4, (2, 3, 4, 5)),
                            ("conv3x3_ws_kernel", r"conv3x3_ws_kernelI(DF16b|f)" + I * 6, (4, 5, 6, 7))):
        t = _tile(sym, pat, order)
        if t:
            return [f"{fam}/{t[0]}/{t[1]}"]
    if "conv3x3_thin_ws_kernel" in sym:
        return ["conv3x3_thin_ws_kernel/bf16/128x32"]
    if "wgrad_reduce" in sym:
        return ["wgrad_reduce_grouped_kernel"]
    if "wgrad" in sym:
        return ["wgrad_grouped/bf16" if "DF16b" in sym or "alltaps" in sym or "wgrad128" in sym or "wgrad_halo" in sym else "wgrad_grouped/f32"]
    if "reduce2_kernel" in sym:
        return ["reduce2_kernel/BwdRedF" if "BwdRedF" in sym else "reduce2_kernel/StatsF"]
    if "ew2_kernel" in sym:
        return ["ew2_kernel/BwdApplyF" if "BwdApplyF" in sym else "lazy_ew_kernel+ew2_kernel/FwdF"]
    if "lazy_ew_kernel" in sym:
        return ["lazy_ew_kernel+ew2_kernel/FwdF"]
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
            if k.startswith("wgrad_grouped"):
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

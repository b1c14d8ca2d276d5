"""bench.py's stdout line must stay parseable by the driver (it reads the last ~8 KB of stdout): round 3's line grew to
46 KB (per-kernel and per-layer tables inline) and came back as `"parsed": null`.  These tests build the line from fake
profile dictionaries of the real size and hold it to the contract: one JSON object, < 6000 bytes, the required keys."""
import json

import bench

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOF_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def fake_profile(n_labels=90, n_layers=260):
    prof = {f"conv3x3_halo_kernel/bf16/128x{64 + i}/fwd": {"launches": 5 * (i + 1), "ms": 0.3 + 0.01 * i, "flops": 1e12 * (i + 1),
                                                          "bytes": 3e8 * (i + 1)} for i in range(n_labels)}
    prof["ew2_kernel/BwdApplyF"] = {"launches": 300, "ms": 9.0, "flops": 0.0, "bytes": 7e9}
    layers = {(f"conv3x3_halo_kernel/bf16/128x64/{'fwd' if i % 2 else 'dgrad'}", f"k3s1d1 {64 + i}->{64 + i} @8x80x80"):
              {"launches": 5, "ms": 0.1 + 0.001 * i, "flops": 2e10, "bytes": 1e8} for i in range(n_layers)}
    return prof, layers


def full_result(train=True):
    prof, layers = fake_profile()
    roof, table = bench.roofline_from_profile(prof, 5, None)
    by_layer = bench.roofline_by_layer(layers, 5, train)
    out = {"metric": "images/sec (640x640) train fwd+bwd", "value": 2031.123456, "unit": "images/sec", "n_gpus": 1, "steps": 30,
           "warmup": 8, "ms_per_step": 3.9384, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic (seeded uniform images, seeded boxes/masks, hash-filled weights)",
           "config": {"workload": "config 3: DeSeNet-s training step " + "x" * 150, "batch_per_gpu": 8, "img": 640, "parallelism": "dp1"},
           "roofline": roof, "roofline_by_layer": by_layer,
           "cpu_baseline": {"value": 2.2, "unit": "images/sec", "cores": 16, "kind": "port", "sample": "oracle train step " + "y" * 120},
           "kernels": table}
    child = {"metric": out["metric"], "value": 3100.0, "unit": "images/sec", "ms_per_step": 5.1, "steps": 30, "warmup": 8,
             "dtype": "f32", "workload": "config 2 ...", "roofline": roof, "layer_classes": bench.compact_layer_classes(by_layer)}
    out["also"] = {"config2_infer_fp32_b16": child, "config2_infer_bf16_b16": dict(child), "config5_m1280_bf16_b4": dict(child),
                   "broken": {"error": "Traceback " + "z" * 1000}}
    return out


def monkey_overhead(monkeypatch):
    monkeypatch.setattr(bench, "_pair_overhead_us", lambda: 2.2)


def test_line_is_small_and_complete(monkeypatch):
    monkey_overhead(monkeypatch)
    out = full_result()
    assert len(json.dumps(out)) > 20000            # the thing that must NOT go to stdout
    line = bench.compact_line(out)
    assert "\n" not in line
    assert len(line) < 6000 and len(line) <= bench.LINE_LIMIT
    j = json.loads(line)
    assert REQUIRED <= set(j)
    assert ROOF_KEYS <= set(j["roofline"]) and j["roofline"]["bound"] in ("hbm", "mfma")
    assert CPU_KEYS <= set(j["cpu_baseline"])
    assert "workload" in j["config"]
    assert "kernels" not in j and "roofline_by_layer" not in j
    assert abs(j["value"] - out["value"]) < 1e-2
    assert abs(j["roofline"]["frac"] - out["roofline"]["frac"]) < 1e-3 * out["roofline"]["frac"] + 1e-6
    # the per-class targets of north_star survive, compactly
    assert set(j["layer_classes"]) == {"c3_3x3_fwd", "c3_3x3_dgrad"}
    assert set(j["also"]) == set(out["also"]) and "error" in j["also"]["broken"] and len(j["also"]["broken"]["error"]) <= 160
    for sec in ("config2_infer_fp32_b16", "config5_m1280_bf16_b4"):
        assert {"value", "ms_per_step", "roofline"} <= set(j["also"][sec])
        assert set(j["also"][sec]["roofline"]) <= {"kernel", "bound", "frac"}


def test_line_sheds_optional_parts_rather_than_growing(monkeypatch):
    monkey_overhead(monkeypatch)
    out = full_result()
    out["config"]["workload"] = "w" * 5000
    out["also"] = {f"sec{i}": dict(out["also"]["config2_infer_fp32_b16"]) for i in range(40)}
    line = bench.compact_line(out)
    assert len(line) <= bench.LINE_LIMIT
    j = json.loads(line)
    assert REQUIRED <= set(j) and j["roofline"] is not None and j["cpu_baseline"] is not None


def test_no_profile_and_multi_gpu_shapes(monkeypatch):
    monkey_overhead(monkeypatch)
    out = full_result()
    out["roofline"] = None
    out["roofline_by_layer"] = None
    out["cpu_baseline"] = None
    out["also"] = None
    out["n_gpus"] = 8
    out["config"].update({"backend": "nccl", "rccl_ranks": 8, "allreduce_us": 412.3, "allreduce_MB": 31.0})
    j = json.loads(bench.compact_line(out))
    assert j["roofline"] is None and j["cpu_baseline"] is None and j["config"]["rccl_ranks"] == 8

"""bench.py's evidence plumbing, on the CPU: the committed PMC traffic files are the ones the bench will look
up for its three trunk engines (round 4 shipped `roofline.traffic: null` because the kernel name in the JSON and the name in
bench.py had drifted apart behind a bare `except`), and the one line rank 0 prints stays under the 8 KB the driver's record
keeps, with the contract-true engine's numbers at the top level."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_pmc_traffic_files_match_what_bench_looks_up():
    import bench
    dtype_of = {"k_tower8_c128": "fp16", "k_trunk_split_c128": "fp16x2", "k_trunk_mx_c128": "mx6"}
    seen = 0
    for kernel, fname in bench.TRAFFIC_FILE.items():
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        tj = json.load(open(path))
        seen += 1
        assert kernel.startswith(tj["kernel"]), (fname, tj["kernel"], kernel)       # what bench.py's pmc_traffic() requires
        cfg = tj["config"]
        assert (cfg["B"], cfg["res_block_nums"], cfg["dtype"], bool(cfg.get("compact", False))) == (8192, 7, dtype_of[kernel], False), (fname, cfg)
        assert tj["traffic_bytes_per_launch"] >= tj["algorithmic_bytes_per_launch"] > 30e6
        assert set(bench.TRUNK_KERNEL[d] for d in bench.TRUNK_KERNEL) == set(bench.TRAFFIC_FILE)
    assert seen >= 3
    # the rules kernels' counter traffic: measured at the size bench.py's rules_roofline uses, and equal to the ABI bytes (the
    # set form reads 91 B and writes 266 B per position: nothing is re-read)
    rj = json.load(open(os.path.join(ROOT, "profiles", "pmc_rules_traffic.json")))
    assert rj["positions"] == 1 << 20
    m = rj["kernels"]["k_movegen_mask"]["traffic_bytes_per_launch"] / float(1 << 20)
    assert 350 < m < 365, m
    # list + mask: 91 B in, 256 B of padded rows + 264 B of mask + 2 B out = 613; rows up to their count: ~40 moves = 6 pieces of 16 B
    assert 600 < rj["kernels"]["k_movegen_list<true, true>"]["traffic_bytes_per_launch"] / float(1 << 20) < 630
    assert 430 < rj["kernels"]["k_movegen_list<true, false>"]["traffic_bytes_per_launch"] / float(1 << 20) < 470


def test_compact_line_is_small_and_carries_the_contract_numbers():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04H_bench_default.json")))     # a complete record of round 4 (20 KB)
    assert len(json.dumps(full)) > 9000
    line = bench.compact_line(full)
    txt = json.dumps(line)
    assert len(txt) < 6000, len(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "strict_engine", "steady_state", "contract_steps", "roofline_tree", "roofline_rules",
              "net_error"):
        assert k in line, k
    assert line["value"] == full["value"] and line["config"]["workload"] == full["config"]["workload"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["kernel"] == "k_tower8_c128" and "traffic" in r
    se = line["strict_engine"]
    assert se["value"] == full["strict_engine"]["value"] and se["frac"] == full["strict_engine"]["roofline"]["frac"]
    assert se["dlogit_trained_like"] <= 1e-3 and se["meets_1e-3_abs_logit_and_value"] is True
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and "sample" in cb and "measured_in_this_run" not in cb   # old record: no flag
    assert set(line["roofline_rules"]) >= {"bound", "achieved", "peak", "unit", "frac"}


def test_round5_record_compacts_to_the_committed_line():
    """profiles/r05_bench_default_detail.json (the complete record of a default run at HEAD) -> compact_line == the line that run
    printed (profiles/r05_bench_default_line.json), under 8 KB, with the strict engine's kernel, rate, roofline fraction and
    measured errors at the top level."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_detail.json")))
    line = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r05_bench_default_line.json")) if l.startswith("{")][-1])
    got = bench.compact_line(full)
    got["detail_file"] = line["detail_file"]
    got = json.loads(json.dumps(got))
    assert {k: got[k] for k in line} == line      # round 6 added keys (strict_check, fast_engine): None for a round-5 record
    assert got["fast_engine"] is None and got["strict_check"] is None
    assert len(json.dumps(line)) < 8000
    se = line["strict_engine"]
    assert se["kernel"] == "k_trunk_mx_c128" and se["dtype"] == "mx6" and se["value"] > 1.7e6 and 0.25 < se["frac"] < 0.35
    assert se["dlogit_trained_like"] <= 1e-3 and se["dvalue_trained_like"] <= 1e-3 and se["meets_1e-3_abs_logit_and_value"] is True
    assert line["roofline"]["traffic"] and line["roofline"]["kernel"] == "k_tower8_c128" and line["engine"].startswith("k_tower8_c128")
    assert line["cpu_baseline"]["reference_python"]["measured_in_this_run"] is False      # the GPU box has no /root/reference


def test_reference_python_flag(monkeypatch):
    """cpu_baseline.reference_python.measured_in_this_run: True only when /root/reference was imported and timed in this run."""
    import bench
    monkeypatch.setattr(bench, "_cpu_workers", lambda specs: (_ for _ in ()).throw(RuntimeError("no workers in this test")))
    monkeypatch.setattr(bench, "reference_python_now", lambda timeout=90: None)
    out = bench.cpu_baseline(2, 1.0)
    assert out["reference_python"]["measured_in_this_run"] is False and out["reference_python"]["search_only_sims_per_s_per_core"] == 405
    monkeypatch.setattr(bench, "reference_python_now", lambda timeout=90: {"search_only_sims_per_s_per_core": 650.0})
    out = bench.cpu_baseline(2, 1.0)
    assert out["reference_python"]["measured_in_this_run"] is True and out["reference_python"]["search_only_sims_per_s_per_core"] == 650.0
    assert out["reference_python"]["static_record"]["end_to_end_7block_sims_per_s"] == 157


def test_trained_like_probe_is_informative_cpu():
    """bench.py's trained-like net-error probe.  (a) A value head whose 1x1 conv is off behind its ReLU on every
    cell (what 19 blocks produced: dvalue = 0.0 on the committed lines) is revived before the heads are calibrated: the value
    output varies across positions.  (b) Calibrating on identical positions (96 copies of the start position, --start-position
    in round 5: tanh argument scaled by 1 / 0) raises instead of producing a blown-up head.  fp32 torch engine on the CPU."""
    import numpy as np
    import torch
    from cchess_zero_amd.net import PolicyValueNet, trained_like_
    rng = np.random.default_rng(0)
    x = np.zeros((48, 9, 10, 14), np.float32)
    for i in range(48):                                   # distinct sparse one-hot planes: encoder-like inputs
        idx = rng.choice(90, 20, replace=False)
        x[i].reshape(90, 14)[idx, rng.integers(0, 14, 20)] = 1.0
    net = PolicyValueNet(2, "cpu", torch.float32, seed=4, backend="torch")
    with torch.no_grad():
        net.module.value_conv.conv.bias.fill_(-50.0)      # dead: every pre-activation far below zero
    with torch.no_grad():
        _, v0 = net.module(torch.from_numpy(x).permute(0, 3, 1, 2))
    assert float(v0.std()) == 0.0
    trained_like_(net, x)
    with torch.no_grad():
        lg, v = net.module(torch.from_numpy(x).permute(0, 3, 1, 2))
    assert float(v.std()) > 0.05 and float(v.abs().max()) < 1.0 and 5.0 < float(lg.max(dim=1).values.mean()) < 15.0
    import pytest
    net2 = PolicyValueNet(2, "cpu", torch.float32, seed=4, backend="torch")
    with pytest.raises(ValueError, match="distinct positions"):
        trained_like_(net2, np.repeat(x[:1], 96, axis=0))


def test_round6_record_compacts_to_the_committed_line():
    """profiles/r06f_bench_default_detail.json (the complete record of `python bench.py` at the round-6 code) -> compact_line == the
    line that run printed (profiles/r06f_bench_default.json), under 8 KB, and it is the STRICT engine's line: value / roofline of
    k_trunk_mx_c128 selected by the net's own measurement, the fast engine as a leg, an informative trained-like probe."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06f_bench_default_detail.json")))
    line = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r06f_bench_default.json")) if l.startswith("{")][-1])
    got = json.loads(json.dumps(bench.compact_line(full)))
    got["detail_file"] = line["detail_file"]
    assert {k: got[k] for k in line} == line and len(json.dumps(line)) < 8000
    assert line["dtype"] == "mx6" and line["roofline"]["kernel"] == "k_trunk_mx_c128" and line["engine"].startswith("k_trunk_mx_c128")
    assert 1.8e6 < line["value"] < 2.4e6 and 0.27 < line["roofline"]["frac"] < 0.36 and line["roofline"]["traffic"] > 1e9
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-9
    sc = line["strict_check"]
    assert sc["engine"] == "mx6" and sc["positions"] >= 64 and max(sc["dlogit"], sc["dvalue"]) <= sc["tol"] == 5e-4 and not sc["fell_over_from"]
    fe = line["fast_engine"]
    assert line["strict_engine"] is None and fe["kernel"] == "k_tower8_c128" and fe["value"] > 1.7 * line["value"] and fe["frac"] > 0.55
    assert fe["meets_1e-3_abs_logit_and_value"] is False                      # the reason it is not the default
    ne = line["net_error"]
    assert ne["meets_1e-3_abs_logit_and_value_as_benchmarked"] and ne["meets_1e-3_abs_logit_and_value_trained_like"] and ne["probe_informative"]
    assert ne["trained_like"]["dvalue"] > 0 and ne["strict_on_trained_like"]["engine"] in ("mx6", "fp16x2")
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["reference_python"]["measured_in_this_run"] is False

"""The N > 1 code path ON THE GPU BOX: the driver's 8-GPU launch must not be the first time this code runs
against a device.  One MI355X is available to the tests, so two ranks share cuda:0 and talk over gloo (RCCL cannot put two
ranks on one device); RCCL itself is exercised with a world of one (--force-dist): communicator set-up, all_gather_into_tensor /
all_reduce / barrier on device tensors.  What the tests pin is the plumbing that does not depend on the transport: rank
sharding of the games, the barrier + max-over-ranks timing, the fixed-capacity record exchange inside the timed region,
the data-parallel policy update (policy_value_network_gpus.py:216-250: gradient averaging; main.py:1240: data_buffer.extend of
every rank's games) leaving every replica with identical weights."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keep_logs(name, p):
    """A failing child's full output goes where the GPU session's scratch is merged back from (pytest truncates long messages)."""
    try:
        d = os.path.join(ROOT, "gpurun_out", "testlogs")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".txt"), "w") as f:
            f.write("rc %s\n==== stdout\n%s\n==== stderr\n%s\n" % (p.returncode, p.stdout, p.stderr))
    except Exception:
        pass


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, env=env,
                       timeout=timeout, stdin=subprocess.DEVNULL)
    if p.returncode != 0:
        _keep_logs("bench_" + "_".join(a.strip("-") for a in args[:6]), p)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_selfplay_with_timed_record_exchange():
    d = _bench(["--gpus", "2", "--all-on-device0", "--dist-backend", "gloo", "--selfplay", "--timed-gather", "--games", "512",
                "--playout", "40", "--steps", "64", "--warmup", "8", "--age-steps", "64", "--steady-steps", "0", "--no-cpu-baseline"])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["world_size"] == 2 and c["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert len(c["per_rank_sims_per_s"]) == 2 and all(v > 0 for v in c["per_rank_sims_per_s"])
    assert c["rank_cpus"] is None or (len(c["rank_cpus"]) == 2 and _disjoint(c["rank_cpus"])), c["rank_cpus"]   # also with --all-on-device0
    assert c["record_gather"] is True
    sp = c["selfplay"]
    assert sp["timed_gather"] is True and sp["gathers"] >= 1 and sp["gathered_records"] > 0
    assert sp["stalled_games"] == 0 and sp["dropped_records"] == 0
    assert c["trees_with_error_status"] == 0
    # whole-job value: the two ranks' simulations over the slowest rank's time
    assert d["value"] > max(c["per_rank_sims_per_s"]) and d["value"] <= sum(c["per_rank_sims_per_s"]) * 1.001
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_search_loop_default_engine():
    """The default loop (search, aged trees, steady leg, the other engine's leg) with two ranks: both legs carry two per-rank
    figures.  Since round 6 the default engine is precision "strict" (what policy_value_network() runs), selected by the net's own
    measurement; the fast fp16 engine is the extra leg."""
    d = _bench(["--gpus", "2", "--all-on-device0", "--dist-backend", "gloo", "--games", "512", "--playout", "64", "--steps", "24",
                "--warmup", "4", "--age-steps", "48", "--steady-steps", "48", "--alt-steps", "16", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["value_source"].startswith("steady_state")
    assert len(d["steady_state"]["per_rank_sims_per_s"]) == 2 and len(d["fast_engine"]["per_rank_sims_per_s"]) == 2
    assert d["contract_steps"]["steps"] == 24 and d["config"]["record_gather"] is True
    assert d["config"]["trees_with_error_status"] == 0
    assert d["dtype"] == "mx6" and d["roofline"]["kernel"] == "k_trunk_mx_c128" and d["engine"].startswith("k_trunk_mx_c128")
    assert d["strict_check"]["engine"] == "mx6" and d["strict_check"]["positions"] >= 64 and not d["strict_check"]["fell_over_from"]
    assert max(d["strict_check"]["dlogit"], d["strict_check"]["dvalue"]) <= 5e-4
    ne = d["net_error"]
    assert ne["meets_1e-3_abs_logit_and_value_as_benchmarked"] is True and ne["meets_1e-3_abs_logit_and_value_trained_like"] is True
    assert ne["probe_informative"] is True and ne["trained_like"]["dvalue"] > 0
    assert d["strict_engine"] is None and d["fast_engine"]["kernel"] == "k_tower8_c128" and d["fast_engine"]["dtype"] == "fp16"
    assert d["fast_engine"]["value"] > d["value"] > 0 and d["fast_engine"]["frac"] > d["roofline"]["frac"] > 0
    assert len(json.dumps(d)) < 8000                      # the driver's record keeps an 8 KB tail of the line
    full = json.load(open(os.path.join(ROOT, d["detail_file"])))
    assert full["net_error"]["strict_on_trained_like"]["engine"] in ("mx6", "fp16x2") and "telemetry" in json.dumps(full)
    # the fast engine by explicit choice: the strict engine is then the extra leg (round 5's layout)
    d = _bench(["--dtype", "fp16", "--games", "512", "--playout", "64", "--steps", "24", "--warmup", "4", "--age-steps", "48", "--steady-steps", "0",
                "--strict-steps", "16", "--no-cpu-baseline"])
    assert d["dtype"] == "fp16" and d["fast_engine"] is None and d["strict_engine"]["kernel"] == "k_trunk_mx_c128"
    assert d["strict_engine"]["meets_1e-3_abs_logit_and_value"] is True and d["strict_check"] is None


@pytest.mark.gpu
def test_bench_rccl_world_of_one():
    """RCCL (backend "nccl") with a world of 1: process-group set-up on the device, the timed all-gather of records, barriers."""
    d = _bench(["--force-dist", "--selfplay", "--timed-gather", "--games", "512", "--playout", "40", "--steps", "64", "--warmup", "8",
                "--age-steps", "64", "--steady-steps", "0", "--no-cpu-baseline"])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["dist_backend"] == "nccl" and c["record_gather"] is True
    assert c["selfplay"]["gathers"] >= 1 and c["selfplay"]["gathered_records"] > 0 and c["trees_with_error_status"] == 0


@pytest.mark.gpu
def test_train_two_ranks_end_with_identical_weights(tmp_path):
    """`torchrun --nproc-per-node 2 main.py --mode train`: each rank plays its own games, the records are all-gathered, the
    policy update is data-parallel (gradients averaged, control flow rank-consistent): afterwards both replicas hold the same
    bits, the same global step and the same replay buffer length."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CCHESS_DIST_BACKEND="gloo", CCHESS_ALL_ON_DEVICE0="1", CCHESS_WEIGHT_DIGEST_DIR=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "main.py"), "--mode", "train", "--games", "128", "--train_playout", "20",
           "--batch_size", "64", "--res_block_nums", "2", "--max_batches", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900, stdin=subprocess.DEVNULL)
    if p.returncode != 0:
        _keep_logs("train_two_ranks", p)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split() for r in range(2)]
    assert got[0] == got[1], got                     # weights digest, global step, buffer length
    assert int(got[0][1]) >= 1 and int(got[0][2]) > 64
    assert "samples:" in p.stdout and "kl:" in p.stdout


def _disjoint(rank_cpus):
    """[first cpu, last cpu, count] per rank -> the slices do not overlap (pin_rank_to_cpus hands out contiguous slices)"""
    spans = sorted((a, b) for a, b, n in rank_cpus)
    return all(spans[i][1] < spans[i + 1][0] for i in range(len(spans) - 1)) and all(b - a + 1 == n for a, b, n in rank_cpus)


@pytest.mark.gpu
def test_bench_eight_ranks_self_launch_selfplay_with_timed_exchange():
    """the shape of the driver's 8-GPU launch, on one device — `python bench.py --gpus 8` from a bare shell
    (bench.py starts its own 8 ranks through torch.distributed.run), self-play with the fixed-capacity record exchange inside
    the timed region (configs[3]'s exchange step; policy_value_network_gpus.py:216-250 / main.py:1240 in the reference).  Eight
    per-rank figures, disjoint CPU slices, every record delivered, nothing left in the exchange."""
    d = _bench(["--gpus", "8", "--all-on-device0", "--dist-backend", "gloo", "--selfplay", "--timed-gather", "--games", "64",
                "--playout", "16", "--steps", "128", "--warmup", "8", "--age-steps", "32", "--steady-steps", "0", "--strict-steps", "0",
                "--no-cpu-baseline"], timeout=1500)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["world_size"] == 8 and c["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert len(c["per_rank_sims_per_s"]) == 8 and all(v > 0 for v in c["per_rank_sims_per_s"])
    # whole-job value against 8 copies of the fastest / the slowest rank's own rate (own = its simulations over its time before it
    # waits at the closing barrier; the ranks complete slightly different numbers of simulations, so "min" may pass 1 by a percent)
    assert len(c["per_rank_busy_seconds"]) == 8 and 0 < c["efficiency_vs_max_rank"] <= 1.0 + 1e-6 and c["efficiency_vs_max_rank"] <= c["efficiency_vs_min_rank"] < 1.2
    assert c["per_rank_spread"] >= 0
    assert c["rank_cpus"] is None or (len(c["rank_cpus"]) == 8 and _disjoint(c["rank_cpus"])), c["rank_cpus"]
    assert c["record_gather"] is True and c["trees_with_error_status"] == 0
    sp = c["selfplay"]
    assert sp["timed_gather"] is True and sp["gathers"] >= 2 and sp["gathered_records"] > 0
    assert sp["stalled_games"] == 0 and sp["dropped_records"] == 0 and sp["pending_after_flush"] == 0
    assert d["value"] > max(c["per_rank_sims_per_s"]) and d["value"] <= sum(c["per_rank_sims_per_s"]) * 1.001
    assert len(json.dumps(d)) < 8000


@pytest.mark.gpu
def test_train_eight_ranks_end_with_identical_weights(tmp_path):
    """`main.py --mode train` with eight ranks on one device: eight self-play shards, one all-gather of records, one data-parallel
    policy update — eight identical replicas afterwards."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CCHESS_DIST_BACKEND="gloo", CCHESS_ALL_ON_DEVICE0="1", CCHESS_WEIGHT_DIGEST_DIR=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "main.py"), "--mode", "train", "--games", "32", "--train_playout", "12",
           "--batch_size", "64", "--res_block_nums", "2", "--max_batches", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=1500, stdin=subprocess.DEVNULL)
    if p.returncode != 0:
        _keep_logs("train_eight_ranks", p)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split() for r in range(8)]
    assert all(g == got[0] for g in got), got             # weights digest, global step, buffer length
    assert int(got[0][1]) >= 1 and int(got[0][2]) > 64

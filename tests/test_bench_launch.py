"""bench.py's launch logic on CPU (no GPU here): --gpus N starts N ranks by itself, refuses a
WORLD_SIZE that disagrees, refuses to run with fewer GPUs than ranks -- it never prints "n_gpus": 1
for a --gpus 8 request.  --dry-run exercises everything but the device: rendezvous (gloo), the
host-side sharding (own-rows quantiser + carrier seed exchange, sharded refresh) and the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_by_itself():
    r = run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--blocks", "30", "--launches", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                              # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None and out["scaling"] == "weak"
    assert out["config"]["blocks_per_gpu_per_step"] == 60
    assert [p["rank"] for p in out["placement"]] == [0, 1] and "[bench] rank 1: host " in r.stderr
    e = out["end_to_end"]
    assert e["blocks_per_gpu"] == 30 and e["channels"] == 16 and e["host_refresh_and_quantise_ms"] > 0.0
    assert e["streamed"]["rounds"] == 16 and e["streamed"]["blocks_per_gpu_per_round"] == 30 and e["streamed"]["seconds"] > 0.0


def test_world_size_must_agree_with_gpus():
    r = run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
    r = run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2


def test_more_ranks_than_gpus_is_an_error():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = run(["--gpus", str(have + 7), "--steps", "1", "--no-cpu-baseline"])
    assert r.returncode == 3 and "visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_8_dry_run_is_one_line_from_eight_ranks():
    """The pre-flight for the driver's 8-GPU run: 8 gloo ranks on this host's cores, one JSON line, n_gpus 8, every rank's
    share of the host threads and its own host time per round reported."""
    r = run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--blocks", "24", "--launches", "2", "--rounds", "3"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["scaling"] == "weak"
    assert out["config"]["nco_mode"].startswith("fixed") and out["config"]["host_threads_per_rank"] >= 1
    st = out["end_to_end"]["streamed"]
    assert [p["rank"] for p in st["per_rank"]] == list(range(8))
    assert all(p["host_ms_per_round"] > 0.0 and p["threads"] == out["config"]["host_threads_per_rank"] for p in st["per_rank"])
    assert st["bound"] is None and all(p["kernel_ms_per_round"] is None for p in st["per_rank"])       # no device in a dry run
    assert "bound" in out["end_to_end"] and out["end_to_end"]["bound"] is None
    # where every rank runs: one line per rank on stderr before anything else, the same in the JSON
    assert [p["rank"] for p in out["placement"]] == list(range(8)) and all(p["cpus_granted"] >= 1 and p["GPSIQ_THREADS"] for p in out["placement"])
    for k in range(8):
        assert f"[bench] rank {k}: host " in r.stderr
    assert out["end_to_end"]["streamed"]["expected_bound_at_this_host"]["threads_needed_per_gpu"] == 6
    assert out["exact_mode_unit"] == "Msamples/s" and "exact_mode_value" in out
    # GPSIQ_NCO_REFERENCE time-sharded over the 8 ranks (chain and evaluation by time: every rank its own blocks), per rank
    ref = out["reference_nco"]
    for leg in ("2M6_int8_16ch", "25M_int16_16ch"):
        pr = ref["legs"][leg]["per_rank"]
        assert [p["rank"] for p in pr] == list(range(8)) and ref["legs"][leg]["value"] is None
        assert all(p["chain_by_time_ms"] > 0.0 and p["exchange_ms"] > 0.0 and p["render_call_ms"] is None for p in pr)


def test_collective_selftest_statements_run_over_gloo():
    """bench.py's default N = 1 run executes rccl_selftest(): the collective stack of its N > 1 branch as a world of one
    rank over RCCL, after gpsiq.Context exists.  Here the same statements over gloo (no GPU): init with an explicit
    tcp://127.0.0.1 rendezvous, all_reduce(MAX) float64, all_gather uint8, barrier, a gloo side group, destroy."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    out = bench.rccl_selftest(0, backend="gloo")
    assert out["ok"] is True and "error" not in out, out
    import torch.distributed as dist
    assert not dist.is_initialized()
    # a failure is reported, not raised
    out = bench.rccl_selftest(0, backend="no_such_backend")
    assert out["ok"] is False and "init_process_group" in out["error"]

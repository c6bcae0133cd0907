"""`python bench.py --gpus N` (N > 1) without a launcher starts its own ranks (north_star / SURVEY.md section 8e: the 1/2/4/8-GPU curve).

The worker here is bench.py's stub (DRIL_BENCH_STUB=1): it loads no library and touches no GPU, so these tests run on the CPU box and prove the launcher's plumbing:
environment of every rank, a real gloo rendezvous on the port the launcher picked, a failing rank, a silent rank, a rank-count mismatch, WORLD_SIZE disagreeing with
--gpus.  The real worker behind the same launcher is what the driver's SCALE runs execute.
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def run_bench(extra_env, *argv, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(DRIL_BENCH_STUB="1", **extra_env)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.monotonic() - t0


def stub_envs(stderr):
    return [json.loads(l.split("STUBENV ", 1)[1]) for l in stderr.splitlines() if "STUBENV " in l]


def test_launcher_starts_n_ranks_with_their_environment_and_relays_rank0():
    r, _ = run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, rank 0's
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["ms_per_step"] == 2.0                      # the max over ranks went through a real gloo all-reduce: rank 1's 2 ms, not rank 0's 1 ms
    assert "launcher" in rec
    envs = sorted(stub_envs(r.stderr), key=lambda e: int(e["RANK"]))
    assert [e["RANK"] for e in envs] == ["0", "1"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1"]
    assert all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert envs[0]["MASTER_PORT"] == envs[1]["MASTER_PORT"] and int(envs[0]["MASTER_PORT"]) > 0
    # first multi-GPU contact must be self-diagnosing: RCCL's own warnings on, and every rank names the device it bound BEFORE the communicator
    assert all(e["NCCL_DEBUG"] == "WARN" for e in envs)
    banners = [l for l in r.stderr.splitlines() if "BANNER rank " in l]
    assert len(banners) == 2 and all("NCCL_DEBUG=WARN" in b and "HSA_ENABLE_IPC_MODE_LEGACY=0" in b for b in banners), r.stderr
    assert any("rank 0/2 local_rank 0 ->" in b for b in banners) and any("rank 1/2 local_rank 1 ->" in b for b in banners)


def test_the_callers_own_nccl_debug_setting_wins():
    r, _ = run_bench({"NCCL_DEBUG": "INFO"}, "--gpus", "2")
    assert r.returncode == 0, r.stderr
    assert all(e["NCCL_DEBUG"] == "INFO" for e in stub_envs(r.stderr))


def test_a_failing_rank_ends_all_ranks_and_the_exit_code_is_nonzero():
    r, dt = run_bench({"DRIL_BENCH_STUB_FAIL_RANK": "1"}, "--gpus", "2")
    assert r.returncode != 0
    assert "rank 1 exited with code 7" in r.stderr
    assert not r.stdout.strip()                           # no result line: a partial job is not a measurement
    assert dt < 120                                       # rank 0 (waiting in the rendezvous for a rank that will never come) was ended, not waited for


def test_a_silent_rank_is_ended_by_the_watchdog():
    r, dt = run_bench({"DRIL_BENCH_STUB_SILENT_RANK": "0"}, "--gpus", "2", "--silent-timeout", "4")
    assert r.returncode != 0
    assert "printed nothing for 4 s" in r.stderr
    assert not r.stdout.strip()
    assert dt < 60


def test_total_timeout():
    r, dt = run_bench({"DRIL_BENCH_STUB_SILENT_RANK": "1"}, "--gpus", "2", "--launch-timeout", "3", "--silent-timeout", "1000")
    assert r.returncode != 0 and "did not finish within 3 s" in r.stderr and dt < 60


def test_a_result_with_the_wrong_rank_count_is_refused():
    r, _ = run_bench({"DRIL_BENCH_STUB_WRONG_RANKS": "1"}, "--gpus", "2")
    assert r.returncode != 0
    assert "not an 2-rank result" in r.stderr and not r.stdout.strip()


def test_world_size_that_disagrees_with_gpus_is_refused():
    # under a launcher (WORLD_SIZE set) bench.py does not start ranks of its own; a mismatch is an error in both directions
    for ws, gpus in (("1", "2"), ("2", "1")):
        r, _ = run_bench({"WORLD_SIZE": ws, "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", gpus)
        assert r.returncode != 0 and f"--gpus {gpus} but WORLD_SIZE={ws}" in r.stderr


def test_single_gpu_stub_runs_in_process():
    r, _ = run_bench({}, "--gpus", "1")
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip())
    assert rec["n_gpus"] == 1 and "launcher" not in rec
    assert stub_envs(r.stderr)[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"      # exported at the top of main(), before any library load


def test_under_torch_distributed_run_it_behaves_as_before():
    # the driver's own launch line (WORLD_SIZE set by torchrun): no ranks of our own, rank 0 prints the line
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DRIL_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    recs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and "launcher" not in recs[0]


import pytest


@pytest.mark.gpu
def test_real_worker_on_a_one_gpu_box_a_missing_device_fails_the_whole_job():
    """the REAL worker (libdril_hip.so, no stub) behind the launcher on a box with one GPU: rank 1 asks for device 1, dril_create refuses it, the rank exits non-zero;
    the launcher ends rank 0 (which holds a handle on device 0 and waits in the rendezvous) and exits non-zero WITHOUT a result line — never a 1-GPU number under an
    `n_gpus: 2` label.  (Two ranks on one device are not possible: RCCL refuses duplicate GPUs; tests/test_gpu_dataparallel.py covers the library's N > 1 code.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DRIL_BENCH_STUB")}
    env["HIP_VISIBLE_DEVICES"] = "0"          # ONE visible device whatever the box has: on a multi-GPU node rank 1 would otherwise find device 1 and the job would succeed
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n-envs", "1024", "--n-steps", "32", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not r.stdout.strip(), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "rank 1 exited with code" in r.stderr and "all ranks ended, no result line" in r.stderr, r.stderr[-1500:]
    assert time.monotonic() - t0 < 300


def _visible_gpus() -> int:
    """device count seen by a FRESH child (this process must not initialise HIP for a count; torch.cuda.device_count() does not, on this image)"""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
    return int(r.stdout.strip() or 0) if r.returncode == 0 else 0


@pytest.mark.gpu
def test_real_worker_on_two_devices_gives_a_two_rank_result():
    """the sibling of the test above for a node with >= 2 GPUs (never the builder's one-GPU box: skipped there): the real worker behind the launcher must give ONE line
    that RCCL itself counts as two ranks, with the per-rank banners before it and the all-reduce time in it"""
    if _visible_gpus() < 2:
        pytest.skip("needs >= 2 visible GPUs (the driver's multi-GPU node); the one-GPU box runs the missing-device test instead")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DRIL_BENCH_STUB")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--n-envs", "1024", "--n-steps", "32", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["allreduce_us_per_step"] > 0
    assert sum("BANNER rank " in l for l in r.stderr.splitlines()) == 2


@pytest.mark.gpu
def test_one_rank_through_rccl_reports_the_allreduce_time():
    """DRIL_FORCE_ALLREDUCE=1 on the one-GPU box: the worker builds a 1-rank RCCL communicator, every optimiser step goes through ncclAllReduce, and the line carries
    the measured time per step (what the 8-GPU record will be read for), the banner precedes the communicator"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DRIL_BENCH_STUB")}
    env["DRIL_FORCE_ALLREDUCE"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--n-envs", "1024", "--n-steps", "64", "--no-cpu-baseline", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["rccl_ranks"] == 1 and rec["allreduce_calls"] >= 2 * 10 * 32
    assert rec["allreduce_us_per_step"] > 0 and rec["allreduce"]["launches"] >= 2 * 10 * 32 and rec["allreduce"]["gradient_bytes"] == 4 * (9155 + 8)
    err = r.stderr.splitlines()
    b = [i for i, l in enumerate(err) if "BANNER rank 0/1" in l]; c = [i for i, l in enumerate(err) if "RCCL communicator up" in l]
    assert b and c and b[0] < c[0] and "PCI " in err[b[0]] and "device 0 of" in err[b[0]]


def test_roofline_helpers_price_what_they_say():
    """bench.py's pure functions (no GPU): the §8(d) table of the bandwidth-class kernels — algorithmic bytes / HIP-event time / 8 TB/s — from a profile() dict in which only
    every 8th launch of the per-step kernels was bracketed (total_ms = timed average x all launches), and the ceiling each arithmetic is priced against"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    N, P, steps, epochs = 1 << 20, 9155, 2, 10
    prof = {"rollout_kernel": {"total_ms": 2.0, "launches": 2, "timed_ms": 2.0, "timed_launches": 2},
            "gae_kernel": {"total_ms": 0.02, "launches": 2, "timed_ms": 0.02, "timed_launches": 2},
            "adam_kernel": {"total_ms": 3.2, "launches": 640, "timed_ms": 0.4, "timed_launches": 80},
            "adv_moments_kernel": {"total_ms": 0.0, "launches": 0, "timed_ms": 0.0, "timed_launches": 0}}
    rows = {r["kernel"]: r for r in b.hbm_kernels(prof, N=N, D=4, A=2, discrete=True, P=P, epochs=epochs, steps=steps, normalize=False)}
    assert set(rows) == {"rollout_kernel", "gae_kernel", "adam_kernel"}                      # classes without launches are left out
    g = rows["gae_kernel"]
    assert g["bytes_per_unit"] == 18 and g["achieved_GBps"] == 18 * N * steps / 0.02e-3 / 1e9 and g["frac"] == g["achieved_GBps"] / 8000.0
    a = rows["adam_kernel"]
    assert a["launches"] == 640 and a["achieved_GBps"] == 28 * P * 640 / 3.2e-3 / 1e9      # per launch: all launches x the timed average
    assert rows["rollout_kernel"]["bytes_per_unit"] == 4 * 4 + 4 + 14
    r16 = b.mfma_roofline(228.0, "f16x2 split, f32 accumulate"); r32 = b.mfma_roofline(113.0, "f32")
    assert abs(r16["peak"] - 2516.6 / 3) < 0.5 and abs(r32["peak"] - 157.3) < 0.1
    assert r16["frac"] == 228.0 / r16["peak"] and r16["frac_vs_f32_peak"] == 228.0 / r16["f32_mfma_peak"] > 1 > r16["frac"]


def test_the_json_line_fits_the_drivers_tail():
    """the driver keeps the last 8 KB of stdout: a line built from full-size secondary entries must come out under that when compacted (round 4's was 17 KB and its
    tail showed none of the headline's fields)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    full = [json.loads(l) for l in (ROOT / "profiles" / "r05_bench_default_secondary_full.jsonl").read_text().splitlines() if l.strip()]
    assert len(full) == 4 and max(len(json.dumps(e)) for e in full) > 3000                  # the real, verbose entries of a run
    compact = [b.compact_entry(e) for e in full]
    head = json.loads((ROOT / "profiles" / "r05_bench_default.json").read_text()); head["secondary"] = compact
    assert len(json.dumps(head)) < 8000
    c2 = [c for c in compact if "configs[2]" in c["workload"]][0]
    assert c2["roofline"]["kernel"] == "ppo_grad_wide_split_kernel" and 0 < c2["roofline"]["frac"] < 1 and c2["value"] > 0
    c0 = [c for c in compact if "configs[0]" in c["workload"]][0]
    assert c0["cpu_baseline"]["kind"] == "port" and c0["cpu_baseline"]["value"] > 0

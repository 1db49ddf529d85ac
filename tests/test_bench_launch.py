"""bench.py's launch path on CPU: `python bench.py --gpus N` without a launcher must start N ranks itself, rendezvous
on 127.0.0.1, time between barriers, take the max over ranks and have rank 0 print exactly ONE JSON line on stdout.

The driver starts the N > 1 bench as a plain `python bench.py --gpus N` (round 2's line was lost to that: the script
demanded torchrun).  Here the same code runs with `--standin-cpu`: gloo instead of RCCL, a stand-in step instead of
the HIP operators (which have no CPU path) — everything else is what runs on the GPU box."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VMI_FORCE_DIST")}
    return env


def _one_json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {stdout[:500]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_plain_python_bench_gpus_n_launches_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "7", "--warmup", "2", "--standin-cpu"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == n and line["steps"] == 7 and line["warmup"] == 2
    assert line["self_launched"] is True
    assert line["config"]["global_batch"] == 256 * n
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert "STAND-IN" in line["metric"]          # can never be mistaken for a measurement
    # the N > 1 line says how it was timed and carries every rank's figure, not only the max; and the same K steps the
    # way rounds 1-2 measured them (blocking exchange on every step, clock behind the closing barrier)
    assert line["method_version"] == 3 and "in front of the closing barrier" in line["timing_bracket"]
    assert len(line["ms_per_step_per_rank"]) == n and max(line["ms_per_step_per_rank"]) == pytest.approx(line["ms_per_step"])
    leg = line["legacy_method_step"]
    assert leg["method_version"] == 2 and leg["ms_per_step"] > 0 and "BEHIND" in leg["timing_bracket"]


@pytest.mark.parametrize("scaling,per_rank", [("weak", 256), ("strong", 256)])
def test_world_8_stand_in_of_the_scaling_curve(scaling, per_rank):
    """The shape of the driver's 8-GPU run, on CPU over gloo: `python bench.py --gpus 8 [--scaling strong]` starts eight ranks,
    each bound to its own slice of the host's cores, 256 sequences per rank either way (weak: 8 x 256; strong: 2048 / 8), the
    token exchange once per token (every 12th layer step), one line with every rank's figure and placement."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "25", "--warmup", "2", "--standin-cpu", "--scaling", scaling],
                       capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["scaling"] == scaling and line["self_launched"] is True
    assert line["config"]["batch_per_rank"] == per_rank and line["config"]["global_batch"] == 2048
    assert line["token_exchange_every_steps"] == 12
    per = line["ms_per_step_per_rank"]
    assert len(per) == 8 and all(x > 0 for x in per) and max(per) == pytest.approx(line["ms_per_step"])
    # what the first real 8-GPU line must carry beside `value`: the ranks of the process group, every rank's figure, the
    # exchange's own duration (the real line: backend nccl = RCCL, per-rank kernel medians too — same keys, same code path)
    assert line["rccl_ranks"] == 8 and line["token_exchange_us"] > 0 and "gloo" in line["process_group_backend"]
    assert line["value"] == pytest.approx(2048 * 25 / (line["ms_per_step"] * 25 / 1e3), rel=1e-6)
    place = line["rank_placement"]
    assert [p["local_rank"] for p in place] == list(range(8))
    cores = [p["cores"] for p in place]
    if len(os.sched_getaffinity(0)) >= 8:        # every rank has its own cores
        assert len(set(cores)) == 8 and all(p["bound"] and p["n_cores"] >= 1 for p in place), place


def test_rank_placement_follows_the_gpus_numa_node(tmp_path):
    """shard.cores_for_rank / gpu_numa_node against a fake /sys: a rank runs on the cores of its GPU's NUMA node, sliced among
    the local ranks; an unknown node (-1, or no /sys) deals all allowed cores evenly."""
    sys.path.insert(0, REPO)
    from vllmini_amd import shard

    (tmp_path / "bus/pci/devices/0000:05:00.0").mkdir(parents=True)
    (tmp_path / "bus/pci/devices/0000:05:00.0/numa_node").write_text("1\n")
    (tmp_path / "devices/system/node/node1").mkdir(parents=True)
    (tmp_path / "devices/system/node/node1/cpulist").write_text("64-127,192-255\n")
    assert shard.gpu_numa_node("0000:05:00.0", str(tmp_path)) == 1
    assert shard.gpu_numa_node("0000:99:00.0", str(tmp_path)) == -1
    allowed = set(range(256))
    got = [shard.cores_for_rank(r, 8, 1, allowed, str(tmp_path)) for r in range(8)]
    assert all(len(g) == 16 for g in got) and got[0][0] == 64 and got[7][-1] == 255
    assert sorted(c for g in got for c in g) == list(range(64, 128)) + list(range(192, 256))
    even = [shard.cores_for_rank(r, 8, -1, allowed, str(tmp_path)) for r in range(8)]
    assert [len(g) for g in even] == [32] * 8 and even[3][0] == 96
    assert shard.cores_for_rank(5, 8, -1, {3}, str(tmp_path)) == [3]              # fewer cores than ranks: never empty


def test_bench_under_an_external_launcher_does_not_relaunch():
    """With RANK / WORLD_SIZE in the environment (torch.distributed.run's contract) each process is one rank."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1", "--standin-cpu"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=REPO))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    line = _one_json_line(outs[0][0])
    assert outs[1][0].strip() == ""              # only rank 0 prints
    assert line["n_gpus"] == 2 and line["self_launched"] is False


def test_single_process_standin_and_failure_propagation():
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--standin-cpu"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)["n_gpus"] == 1
    # the product path has no CPU fallback: without a HIP device (and without --standin-cpu) the bench refuses to run,
    # for N = 1 and — through the launcher's device-count check — for N > 1
    import torch

    if not torch.cuda.is_available():
        for n in ("1", "2"):
            r = subprocess.run([sys.executable, BENCH, "--gpus", n, "--steps", "3", "--warmup", "1"],
                               capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=REPO)
            assert r.returncode != 0 and r.stdout.strip() == ""
            assert "HIP device" in r.stderr


def test_launch_decision_and_cpu_thread_candidates():
    sys.path.insert(0, REPO)
    import bench

    a = bench.parse_args(["--gpus", "4"])
    assert bench.needs_self_launch(a, env={})
    assert not bench.needs_self_launch(a, env={"WORLD_SIZE": "4", "RANK": "0"})
    assert not bench.needs_self_launch(bench.parse_args([]), env={})
    assert bench.parse_args(["--kernel-samples", "10"]).kernel_samples == 50      # never fewer than 50 probed launches
    h = bench.parse_args(["--headline-only"])
    assert h.no_cpu_baseline and h.no_fused and h.no_fp8 and h.no_ragged and h.no_graph and h.no_cfg4 and h.no_e2e
    assert h.no_cfg2 and h.no_strong
    assert bench.cpu_thread_candidates(256, 128) == [8, 16, 32, 64, 128, 256]
    assert bench.cpu_thread_candidates(8, 4) == [1, 2, 4, 8]
    assert bench.cpu_thread_candidates(2, 1) == [1, 2]

"""
CPU rehearsal of the first multi-GPU run of bench.py: the driver launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N ...` on an 8-GPU node that no build round has had. Here the same
command runs with two REAL processes and a stand-in engine (tests/standin_engine.py, gloo in the
place of RCCL): the torchrun environment, the file rendezvous of qoc_amd.parallel.RcclComm between
real processes, the --gpus / WORLD_SIZE check, barrier + max-over-ranks timing and rank-0-only
printing are bench.py's own code.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(tmp_path):
    env = dict(os.environ)
    env["QOCX_RDZV_DIR"] = str(tmp_path)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _expected_sum_cost(world, seeds):
    sys.path.insert(0, ROOT)
    import bench
    return float(sum(np.sum(bench.make_controls(r * seeds, seeds) ** 2) for r in range(world)))


def test_two_rank_launch_prints_one_line_with_whole_job_throughput(tmp_path):
    seeds, steps, world = 3, 2, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1",
           "--seeds-per-gpu", str(seeds), "--standin-engine", "tests.standin_engine"]
    run = subprocess.run(cmd, env=_env(tmp_path), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == steps and line["scaling"] == "weak"
    assert "STAND-IN" in line["data"]
    # whole-job aggregate: the units of ALL ranks over the max-over-ranks time
    units = world * seeds * 1000 * steps
    assert abs(line["value"] - units / (line["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * line["value"]
    # the collective summed over both ranks' (different) seeds
    assert abs(line["check"]["sum_cost"] - _expected_sum_cost(world, seeds)) < 1e-9
    assert line["config"]["parallelism"] == "seed-sharded x2"
    # the rendezvous files of the launch were cleaned up by rank 0
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith("qocx_rdzv_")]


def test_single_process_default_and_gpus_mismatch(tmp_path):
    env = _env(tmp_path)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
                          "--seeds-per-gpu", "2", "--no-cpu-baseline", "--standin-engine", "tests.standin_engine"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    line = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and abs(line["check"]["sum_cost"] - _expected_sum_cost(1, 2)) < 1e-9
    # --gpus 2 without a launcher: refuses, names the command
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--standin-engine",
                          "tests.standin_engine"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "torch.distributed.run" in (bad.stderr + bad.stdout)
    # WORLD_SIZE that disagrees with --gpus
    env2 = dict(env, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    bad2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--standin-engine",
                           "tests.standin_engine"], env=env2, cwd=ROOT,
                          capture_output=True, text=True, timeout=300)
    assert bad2.returncode != 0 and "WORLD_SIZE=4" in (bad2.stderr + bad2.stdout)

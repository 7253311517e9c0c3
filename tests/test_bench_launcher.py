"""`python bench.py --gpus N` must itself run N ranks: the launcher path, the rendezvous and a collective,
on CPU with gloo (CXRK_BENCH_SELFTEST=1 skips the GPU work; everything else is the code the real run takes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_cmd_is_one_rank_per_gpu_on_localhost():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_cmd(4, ["--gpus", "4", "--steps", "3"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_plain_command_spawns_the_ranks_and_fails_loudly():
    env = dict(os.environ, CXRK_BENCH_SELFTEST="1", CXRK_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 2 and out["n_gpus"] == 2
    # a rank that dies makes the plain command exit non-zero (never a silent one-GPU measurement)
    env["CXRK_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0

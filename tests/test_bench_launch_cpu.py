"""bench.py started plainly with --gpus N must run N ranks (it starts its own launcher as a child process);
--dry-launch shows them without touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UNET_NGPU")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                       env=e, timeout=300)
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, recs, p.stderr


def test_plain_start_with_two_gpus_runs_two_ranks():
    rc, recs, err = run_bench("--gpus", "2", "--dry-launch")
    assert rc == 0, err
    assert sorted(r["dry_launch_rank"] for r in recs) == [0, 1]
    assert all(r["world_size"] == 2 for r in recs)
    assert all(r["master"].startswith("127.0.0.1:") for r in recs)


def test_rank_count_from_the_environment():
    rc, recs, err = run_bench("--dry-launch", env={"UNET_NGPU": "3"})
    assert rc == 0, err
    assert sorted(r["dry_launch_rank"] for r in recs) == [0, 1, 2]


def test_one_gpu_starts_no_launcher():
    rc, recs, err = run_bench("--gpus", "1", "--dry-launch")
    assert rc == 0, err
    assert recs == [{"dry_launch_rank": 0, "world_size": 1, "local_rank": 0, "master": "-:-"}]


def test_launcher_and_flag_must_agree():
    # under a launcher (WORLD_SIZE set) the script never starts a second one; a disagreement is an error
    rc, recs, err = run_bench("--gpus", "4", "--no-cpu-baseline", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and "WORLD_SIZE=2" in err

"""`bench.py --gpus N` must start N ranks by itself when no launcher did (the flag was parsed and never
read, so `python bench.py --gpus 8` measured one GPU).  Launch-only mode: every rank reports who it is and exits before
any GPU work, so this runs on the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_starts_that_many_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["launch_only"] and line["n_gpus"] == 2
    ranks = line["ranks"]
    assert sorted(r["rank"] for r in ranks) == [0, 1] and sorted(r["local_rank"] for r in ranks) == [0, 1]
    assert all(r["world"] == 2 for r in ranks)
    assert len({r["pid"] for r in ranks}) == 2                      # two processes, not one process twice
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")


def test_three_ranks_and_default_is_one():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--launch-only"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert sorted(r["rank"] for r in json.loads(out.stdout.strip().splitlines()[-1])["ranks"]) == [0, 1, 2]
    out = subprocess.run([sys.executable, BENCH, "--launch-only"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    one = json.loads(out.stdout.strip().splitlines()[-1])
    assert one["rank"] == 0 and one["world"] == 1


def test_external_launcher_must_agree_with_the_flag():
    # under torch.distributed.run the ranks exist already: --gpus has to name the same number, or the run stops
    env = _env(WORLD_SIZE="4", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    bad = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=4" in bad.stderr
    ok = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    me = json.loads(ok.stdout.strip().splitlines()[-1])
    assert me["rank"] == 1 and me["world"] == 4                      # no second generation of children under a launcher
    unset = subprocess.run([sys.executable, BENCH, "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert unset.returncode == 0 and json.loads(unset.stdout.strip().splitlines()[-1])["world"] == 4


def test_a_failing_rank_fails_the_job():
    # an unknown flag makes every child exit with an argparse error: the parent must not report success
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only", "--config-typo", "x"], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0

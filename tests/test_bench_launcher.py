"""bench.py as its own launcher: a bare `python bench.py --gpus N` must never
print a line for fewer ranks than it was asked for.  No GPU here: the ranks it starts fail, and what is
checked is that they WERE started with the rank environment, that the failure reaches the exit code and
that no line is printed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_bare_gpus_n_starts_n_ranks_and_propagates_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the launcher is exercised by the rehearsal under profiles/")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--reads", "1000", "--cpu-sample", "0", "--e2e-reads", "0"],
                       env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""                      # no line for a run that did not happen
    assert "needs a GPU" in err                                 # the children got as far as looking for their device
    assert "rank" in err and "exited with" in err


def test_external_launcher_with_another_world_size_is_refused():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                        "--reads", "1000"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300)
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""
    assert "WORLD_SIZE 1 != --gpus 4" in p.stderr.decode()


def test_launcher_withholds_a_line_of_the_wrong_size(tmp_path, monkeypatch):
    """spawn_ranks with stand-in children: rank 0 prints n_gpus = 1 for --gpus 2 -> no line, non-zero exit"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    fake = tmp_path / "fake_bench.py"
    fake.write_text("import json, os\n"
                    "if os.environ['RANK'] == '0': print(json.dumps({'n_gpus': int(os.environ.get('FAKE_N', '1'))}))\n")
    monkeypatch.setattr(bench, "__file__", str(fake))
    monkeypatch.setattr(sys, "argv", ["bench.py"])

    class A:
        gpus = 2
    import pytest
    with pytest.raises(SystemExit) as ei:
        bench.spawn_ranks(A())
    assert ei.value.code not in (0, None)
    monkeypatch.setenv("FAKE_N", "2")
    with pytest.raises(SystemExit) as ei:
        bench.spawn_ranks(A())
    assert ei.value.code == 0

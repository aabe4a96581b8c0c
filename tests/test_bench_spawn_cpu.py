"""`python bench.py --gpus N` from a plain interpreter starts its own N ranks.
CPU-only: the spawn helper is driven with stub workers, and bench.py itself must get as far as
"needs a GPU" in every rank -- i.e. fail at the device, not at the launch."""
import json
import os
import subprocess
import sys
import textwrap

from sdfest_amd.parallel import spawn_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "worker.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_spawn_ranks_sets_rendezvous_env_and_relays_rank0(tmp_path):
    out = tmp_path / "out"
    out.mkdir()
    cmd = _stub(tmp_path, f"""
        import json, os
        r = os.environ["RANK"]
        rec = {{k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                              "HSA_ENABLE_IPC_MODE_LEGACY")}}
        open(os.path.join({str(out)!r}, r + ".json"), "w").write(json.dumps(rec))
        if r == "0":
            print(json.dumps({{"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}}), flush=True)
    """)
    drv = tmp_path / "driver.py"
    drv.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom sdfest_amd.parallel import spawn_ranks\n"
                   f"raise SystemExit(spawn_ranks({cmd!r}, 3))\n")
    res = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 3}
    recs = [json.load(open(out / f"{r}.json")) for r in range(3)]
    assert [r["RANK"] for r in recs] == ["0", "1", "2"] and [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and {r["MASTER_ADDR"] for r in recs} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in recs}) == 1 and {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs} == {"0"}


def test_spawn_ranks_propagates_failure_and_stops_the_others(tmp_path):
    cmd = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)   # would hang in a collective: must be terminated, not waited for
    """)
    assert spawn_ranks(cmd, 2, timeout=50) == 7


def test_spawn_ranks_gloo_rendezvous(tmp_path):
    """the spawned ranks can really form a process group with the environment they are given"""
    cmd = _stub(tmp_path, """
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        assert t.item() == 3.0
        dist.destroy_process_group()
    """)
    assert spawn_ranks(cmd, 2, timeout=110) == 0


def test_bench_gpus2_fails_at_the_device_not_at_the_launch():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box check")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert res.stderr.count("bench.py needs a GPU") >= 1
    assert "torch.distributed.run" not in res.stderr

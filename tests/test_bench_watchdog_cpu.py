"""CPU: bench.py's first-contact guards for the multi-GPU run nobody can rehearse here -- the start-up watchdog
(a rank stuck in rendezvous / its first collective says where and exits non-zero, which ends the others) and the
device-count check."""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_watchdog_reports_the_stage_and_exits_nonzero(tmp_path):
    script = tmp_path / "hang.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        with bench.Watchdog(5, "init_process_group(nccl, world_size=8)", 1.0):
            time.sleep(30)       # a rendezvous that never completes
        print("not reached")
    """))
    t0 = time.monotonic()
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 3 and time.monotonic() - t0 < 25
    assert "stage: init_process_group(nccl, world_size=8)" in res.stderr
    assert "WATCHDOG" in res.stderr and "rank 5" in res.stderr and "not reached" not in res.stdout


def test_watchdog_is_silent_when_the_stage_finishes(tmp_path):
    script = tmp_path / "ok.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        with bench.Watchdog(0, "first all-reduce", 1.0):
            pass
        time.sleep(3.5)          # well past both timers: neither may fire after the stage is over
        print("done")
    """))
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "done" in res.stdout and "WATCHDOG" not in res.stderr


def test_a_hung_rank_ends_the_whole_spawned_job(tmp_path):
    """spawn_ranks (what `python bench.py --gpus N` uses): rank 1 never reaches the rendezvous; its watchdog exits 3
    and the launcher terminates rank 0, which would otherwise wait for ever."""
    from sdfest_amd.parallel import spawn_ranks
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        with bench.Watchdog(int(os.environ["RANK"]), "init_process_group", 1.0 if os.environ["RANK"] == "1" else 60.0):
            time.sleep(60)
    """))
    t0 = time.monotonic()
    assert spawn_ranks([sys.executable, str(script)], 2, timeout=100) == 3
    assert time.monotonic() - t0 < 40


def test_a_hung_extra_section_costs_the_extra_not_the_headline(tmp_path):
    """bench.SectionGuard: the multi-rank extras (the sharded loop, its all-reduce inside the graphs) run after the
    headline is complete; one that does not finish is abandoned -- rank 0 prints the line it has, with the fact, and
    every rank exits 0; a section that finishes in time leaves no timer behind."""
    script = tmp_path / "extra.py"
    script.write_text(textwrap.dedent(f"""
        import json, sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        rank = int(sys.argv[1])
        line = {{"metric": "m", "value": 1.5}}
        with bench.SectionGuard(rank, "quick", 1.0, line):
            line["quick"] = [1, 2]
        time.sleep(1.5)              # past the first guard's time: it must not fire any more
        with bench.SectionGuard(rank, "loop_sharded", 1.0, line):
            time.sleep(30)           # a collective that never completes
        print("not reached")
    """))
    for rank in (0, 1):
        t0 = time.monotonic()
        res = subprocess.run([sys.executable, str(script), str(rank)], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0 and time.monotonic() - t0 < 25 and "not reached" not in res.stdout
        if rank == 0:
            line = __import__("json").loads(res.stdout.strip().splitlines()[-1])
            assert line["value"] == 1.5 and line["quick"] == [1, 2] and "abandoned" in line["loop_sharded"]["error"]
            assert line["abandoned_sections"] == ["loop_sharded"]
        else:
            assert res.stdout.strip() == ""
        assert "extra section 'loop_sharded'" in res.stderr

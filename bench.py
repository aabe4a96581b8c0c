#!/usr/bin/env python3
"""Benchmark of the hot path: sphere-tracing depth render forward + backward.

Metric (BASELINE.json): depth renders/sec fwd+bwd, 640x480 @ 64^3 SDF; grad max-abs-err vs ref.
The line carries both halves: `value` (renders/s) and `grad_max_abs_err` (d/dSDF, + the `grad_*` keys beside it and
`parity`), the errors of the benchmarked build's last step against the oracle.  After the headline it appends BASELINE.json's
other single-GPU configurations (`configs`: C1, C2 with the CPU port beside them, C5 with its time to result, C3_l1).

Workload per GPU ("C3", BASELINE.json configs[2]): 256 seeded random
poses of the synthetic blobs(0) 64^3 SDF at 640x480, threshold 0.005, upstream gradient
U(-1,1).  One step = forward of the 256 views + backward of the 256 views (+ the RCCL
all-reduce of the shared d/dSDF when N > 1: `--exchange ring` batches it over M/2 steps beside the
following steps, `--exchange sync` finishes it before the next forward; both are measured and
reported, `value` is the one asked for).  N GPUs = N shards of 256 views of one
256*N-view batch (configs[3] at N=8), so scaling is weak.  Everything is resident in HBM
before the timed region.  `--batch 1` gives configs[1] (one view per launch).

    python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1: spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s; 6.29 TB/s measured copy)
HBM_COPY_PEAK = 6.29e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="views per GPU per step")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the gradient-error report against the oracle")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C2 / C5 / C3_l1 side measurements")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps before the warm-up steps until this much wall time has passed (clock ramp)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="views in the CPU sample (0 = auto)")
    ap.add_argument("--exchange", choices=("ring", "sync"), default="ring",
                    help="N > 1: which exchange of d/dSDF `value` is measured with.  ring: one all-reduce of M/2 "
                         "volumes per M/2 steps, waited for when the ring wraps (independent steps); sync: every "
                         "step's volume is summed before the next forward starts (the dependency of "
                         "render-and-compare).  The other one is measured too and reported beside it.")
    ap.add_argument("--watchdog-s", type=float, default=0.0,
                    help="N > 1: a rank that has not come out of process-group start-up / its first collective after "
                         "this many seconds reports its stage and exits non-zero (default 0 = 120 s + 15 s per rank: "
                         "communicator creation is eager (device_id) and a cold 8-rank RCCL bring-up falls inside the "
                         "first window; negative: no watchdog)")
    return ap.parse_args()


class Watchdog:
    """First-contact guard for the multi-GPU start-up: `with Watchdog(rank, "stage", seconds)` around a call that may
    hang (rendezvous, communicator set-up, the first collective).  When the time is up the rank says where it was
    and exits with status 3, so the launcher (torch.distributed.run, or spawn_ranks) ends the other ranks instead of
    the whole job sitting in a collective until the driver's limit.  Two timers: a Python thread that prints the
    stage, and -- should the hung call hold the interpreter lock -- faulthandler's C-level one two seconds later,
    which needs no lock (it dumps the stacks and exits 1).  Nothing is re-executed and no child is started."""

    def __init__(self, rank, stage, seconds):
        self.rank, self.stage, self.seconds = rank, stage, seconds

    def __enter__(self):
        import faulthandler
        import threading
        if self.seconds <= 0:
            return self
        print(f"[bench rank {self.rank}] stage: {self.stage}", file=sys.stderr, flush=True)

        def fire():
            print(f"[bench rank {self.rank}] WATCHDOG: still in '{self.stage}' after {self.seconds:.0f} s -- giving up "
                  f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                  f"WORLD_SIZE={os.environ.get('WORLD_SIZE')} LOCAL_RANK={os.environ.get('LOCAL_RANK')} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})",
                  file=sys.stderr, flush=True)
            os._exit(3)
        self.timer = threading.Timer(self.seconds, fire)
        self.timer.daemon = True
        self.timer.start()
        faulthandler.dump_traceback_later(self.seconds + 2.0, exit=True)
        return self

    def __exit__(self, *exc):
        import faulthandler
        if self.seconds > 0:
            self.timer.cancel()
            faulthandler.cancel_dump_traceback_later()
            if Watchdog.job_deadline is not None:     # re-arm the whole-job backstop this stage's timer replaced
                left = Watchdog.job_deadline - time.monotonic()
                if left > 1.0:
                    faulthandler.dump_traceback_later(left, exit=True)
        return False

    job_deadline = None


def synthetic_inputs(B_total, rank, B, W, H, device):
    """blobs(0) SDF and this rank's contiguous shard of the seeded pose list."""
    from sdfest_amd.synthetic import blobs_sdf, random_poses
    sdf = blobs_sdf(0)
    pos, quat, isc = random_poses(B_total, seed=1, width=W, height=H, f=W / 2.0)
    sl = slice(rank * B, (rank + 1) * B)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    g = torch.rand((B, H, W), device=device, generator=gen) * 2 - 1
    return sdf, (pos[sl], quat[sl], isc[sl]), t(sdf), t(pos[sl]), t(quat[sl]), t(isc[sl]), g


def _time_oracle(lib, sdf, pos, quat, isc, W, H, thr, threads, budget_cpu_s):
    """renders/s of the oracle's fwd+bwd on len(pos) views with `threads` OpenMP threads."""
    import ctypes
    S = pos.shape[0]
    depth = np.empty((S, H, W), np.float32)
    g = np.random.default_rng(0).uniform(-1, 1, (S, H, W)).astype(np.float32)
    gs = np.empty((64, 64, 64), np.float32)
    gp, gq, gi = np.empty((S, 3), np.float32), np.empty((S, 4), np.float32), np.empty(S, np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cd, ci = ctypes.c_double, ctypes.c_int
    f = W / 2.0

    def once():
        lib.sdfo_render_forward_f32(P(sdf), ci(64), P(pos), P(quat), P(isc), ci(S), ci(W), ci(H),
                                    cd(W / 2), cd(H / 2), cd(f), cd(f), cd(thr), P(depth), None,
                                    None, ci(0))
        lib.sdfo_render_backward_f32(P(g), P(depth), P(sdf), ci(64), P(pos), P(quat), P(isc), ci(S),
                                     ci(W), ci(H), cd(W / 2), cd(H / 2), cd(f), cd(f), ci(0), P(gs),
                                     P(gp), P(gq), P(gi))

    lib.sdfo_set_threads(ci(threads))
    once()  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    reps = 0
    while True:
        once()
        reps += 1
        el = time.perf_counter() - t0
        if el * threads >= budget_cpu_s or el >= 15.0 or reps >= 50:
            break
    return S * reps / el, reps


def cpu_baseline(sdf, poses, W, H, thr, sample):
    """The oracle (a CPU port of the same algorithm) timed on this host's cores, on a bounded
    sample of the SAME workload: all cores on `sample` views, and one core on 16 of them."""
    import ctypes
    import subprocess
    here = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", here, "libsdfr_oracle_native.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(here, "libsdfr_oracle_native.so"))
    pos, quat, isc = (np.ascontiguousarray(a[:sample], dtype=np.float32) for a in poses)
    sdf = np.ascontiguousarray(sdf, dtype=np.float32)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    # What this process may USE is its cgroup's CPU quota, not the cores it may be scheduled on: a GPU box of this pool
    # shows all 256 logical cores of its host and grants a 1-GPU job a share of them -- threads beyond the share are
    # throttled, and the sweep "collapsed" at 128 / 256 threads (round 5: 2208 -> 857 -> 521 renders/s) for that reason,
    # not for the port's (tools/microbench/cpu_sweep.py: the forward stops scaling where the quota ends).
    quota = cpu_quota()
    # OpenMP scaling on a shared many-core host is far from linear: try a few thread counts on
    # the same sample and report the best, with the count that achieved it.
    n1 = min(16, pos.shape[0])
    v_one, _ = _time_oracle(lib, sdf, pos[:n1].copy(), quat[:n1].copy(), isc[:n1].copy(), W, H, thr,
                            1, 3.0)
    sweep = {1: v_one}
    reps = {}
    cap = ncpu if quota is None else min(ncpu, max(8, int(4 * quota)))
    for threads in sorted({t for t in (4, 8, 16, 32, 64, 128, ncpu) if 1 < t <= cap}):
        sweep[threads], reps[threads] = _time_oracle(lib, sdf, pos, quat, isc, W, H, thr, threads, 3.0)
    best = max(sweep, key=lambda k: sweep[k])
    return {"value": round(sweep[best], 2), "unit": "renders/s", "cores": best, "kind": "port",
            "sample": f"{pos.shape[0]} views of the same workload (first poses of the seeded list), "
                      f"fwd+bwd, x{reps.get(best, 1)} repeats; oracle built as libsdfr_oracle_native.so "
                      f"(gcc -O3 -march=native -fopenmp; backward with per-thread d/dSDF slabs); host shows {ncpu} "
                      f"logical cores, cgroup CPU quota "
                      + ("none" if quota is None else f"{quota:.1f} cores") +
                      f"; thread sweep renders/s: " + ", ".join(f"{k}:{v:.0f}" for k, v in sorted(sweep.items())),
            "cpu_quota_cores": quota,
            "value_1thread": round(v_one, 2)}


def cpu_quota():
    """cores' worth of CPU time the process's cgroup grants (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), None: unlimited"""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(period)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / period
    except Exception:
        return None


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), valid only for
    the build they were taken on: the file carries the sha of the kernel sources, and a line produced by other
    sources reports traffic null."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(p))
    except Exception:
        return None, "profiles/pmc_traffic.json missing"
    from tools.bench_extra import kernel_sources_sha
    here = kernel_sources_sha()
    if t.get("kernel_sources_sha16") != here:
        return None, (f"profiles/pmc_traffic.json was taken on kernel sources {t.get('kernel_sources_sha16')}, "
                      f"this build is {here}: not reported")
    return t, f"{t.get('source')}, kernel sources {here}"


def gpus_without_the_runtime():
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd topology: nodes with SIMDs), for the launcher
    parent, which must not touch the HIP runtime before it starts its ranks (spawn_ranks' precondition) --
    torch.cuda.device_count() stays clear of it only while amdsmi discovery works and calls hipGetDeviceCount (hipInit)
    otherwise.  0 when the topology cannot be read: the check is then left to the ranks themselves."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in os.listdir(base):
            try:
                props = open(os.path.join(base, node, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                if line.startswith("simd_count ") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        return 0
    return n


def sharded_loop_section(N, rank, barrier):
    """N > 1: the render-and-compare LOOP sharded over the ranks (sdfest_amd.pipeline, process_group="world"): what the
    one all-reduce INSIDE an iteration costs over xGMI -- the headline's steps are independent and do not show it.
    8 N and 64 N views of the C5 scene, 50 iterations, both exchanges, beside the same per-rank view count as a single
    process.  Runs after the headline is complete, under a SectionGuard; an exception becomes {"error": ...}."""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    from tools._loop_scene import c5_scene
    rows = []
    for per_rank in (8, 64):
        V = per_rank * N
        s = c5_scene(views=V, max_iterations=50)

        def time_loop(loop):
            loop(*s["init"])             # warm-up iteration, captures, 50 iterations
            torch.cuda.synchronize()
            ts, res = [], None
            for _ in range(3):
                barrier()
                t0 = time.perf_counter()
                res = loop(*s["init"])
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 50 * 1e3)
            return round(float(np.median(ts)), 4), res
        row = {"views": V, "views_per_rank": per_rank}
        for exchange in ("sdf", "latent"):
            loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], process_group="world",
                                         exchange=exchange)
            row[f"ms_per_iteration_{exchange}"], res = time_loop(loop)
            row[f"final_position_error_mm_{exchange}"] = round((res[0] - s["p_true"]).norm().item() * 1e3, 3)
            del loop
        single = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"][:per_rank].contiguous())
        row["single_rank_ms_same_views_per_rank"], _ = time_loop(single)
        del single, s
        torch.cuda.empty_cache()
        rows.append(row)
    return rows


class SectionGuard:
    """An optional multi-rank section that comes AFTER the headline is complete: if it has not finished after `seconds`,
    rank 0 prints the line it already has (with the fact under `key`) and every rank leaves -- a hang in an extra can
    cost the extra, never the measurement."""

    def __init__(self, rank, key, seconds, line):
        import threading

        def give_up():
            # (the headline is on stdout already -- main() prints it before the first extra --; this is the line again
            # with the fact.  Exit status 0 on purpose: the MEASUREMENT is complete and a launcher that fails the job on
            # any rank's status would throw it away; the hang is in the line (`abandoned_sections`) and on stderr.)
            sys.stderr.write(f"bench.py rank {rank}: extra section '{key}' not finished after {seconds:.0f} s: abandoned\n")
            sys.stderr.flush()
            if rank == 0:
                line[key] = {"error": f"not finished after {seconds:.0f} s: abandoned"}
                line.setdefault("abandoned_sections", []).append(key)
                print(json.dumps(line), flush=True)
            os._exit(0)
        self.timer = threading.Timer(seconds, give_up)
        self.timer.daemon = True

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def collective_in_graph_section(N, rank, barrier, seconds, line):
    """The sharded loop with its all-reduce captured INSIDE the hipGraphs (FusedRenderAndCompare(graph_collective=True):
    measured at world size 1, 0.140 -> 0.125 ms per iteration) -- never run over more than one GPU before this line, so
    it comes LAST and under a guard: if it has not finished after `seconds`, rank 0 prints the line it already has
    (with the fact) and every rank leaves.  Returns the rows, or {"error": ...}."""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    from tools._loop_scene import c5_scene
    guard = SectionGuard(rank, "loop_sharded_collective_in_graph", seconds, line)
    guard.__enter__()
    try:
        rows = []
        for per_rank in (8, 64):
            s = c5_scene(views=per_rank * N, max_iterations=50)
            row = {"views": per_rank * N, "views_per_rank": per_rank}
            for exchange in ("sdf", "latent"):
                loop = FusedRenderAndCompare(s["decoder"], s["camera"], s["config"], s["targets"], process_group="world",
                                             exchange=exchange, graph_collective=True)
                loop(*s["init"])
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    barrier()
                    t0 = time.perf_counter()
                    res = loop(*s["init"])
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) / 50 * 1e3)
                row[f"ms_per_iteration_{exchange}"] = (round(float(np.median(ts)), 4)
                                                       if loop.graph_whole_one is not None else None)
                row[f"captured_{exchange}"] = (True if loop.graph_whole_one is not None
                                               else f"refused: {loop.graph_collective_error}")
                row[f"final_position_error_mm_{exchange}"] = round((res[0] - s["p_true"]).norm().item() * 1e3, 3)
                del loop
            del s
            torch.cuda.empty_cache()
            rows.append(row)
        return rows
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        guard.__exit__()


def main():
    args = parse()
    N = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on a box with fewer GPUs than ranks (never a measurement): SDFR_BENCH_SHARE_GPU=1 puts every rank on
    # GPU 0 and SDFR_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    share_gpu = os.environ.get("SDFR_BENCH_SHARE_GPU") == "1"
    if "RANK" not in os.environ and N > 1:
        n_dev = gpus_without_the_runtime()
        # plain `python bench.py --gpus N`: this process has not touched the GPU; it starts the N
        # ranks as child processes (what torch.distributed.run would do), relays their output
        # (rank 0 prints the result line) and exits with their status.
        if 0 < n_dev < N and not share_gpu:
            raise SystemExit(f"bench.py --gpus {N}: this node shows {n_dev} GPU(s) (torch.cuda.device_count()); "
                             f"one rank per GPU needs {N}.  (A rehearsal on fewer GPUs: SDFR_BENCH_SHARE_GPU=1 "
                             f"SDFR_BENCH_BACKEND=gloo -- marked REHEARSAL in the line, never a measurement.)")
        from sdfest_amd.parallel import spawn_ranks
        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], N))
    if world != N:
        N = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); "
                         "the product has no CPU path")
    n_dev = torch.cuda.device_count()          # (a rank: it is about to initialise its GPU anyway)
    if share_gpu:
        local_rank = 0
    elif local_rank >= n_dev:
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK={local_rank} but this node shows {n_dev} GPU(s) "
                         f"(torch.cuda.device_count()); start one rank per GPU (--nproc-per-node <= {n_dev})")
    backend = os.environ.get("SDFR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # under torch.distributed.run (RANK set) the collective path is taken even at world size 1,
    # so a 1-GPU torchrun exercises exactly the code the multi-GPU runs execute
    use_dist = N > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if args.watchdog_s == 0.0:
        args.watchdog_s = 120.0 + 15.0 * N
    elif args.watchdog_s < 0.0:
        args.watchdog_s = 0.0
    if use_dist:
        import faulthandler
        import torch.distributed as dist
        # backstop for the whole multi-rank run: a rank still here after 9 minutes dumps every thread's stack and
        # exits 1 (C-level timer, needs no interpreter lock), so a hang past start-up also says where it is
        # (540 s: below the 600 s the driver gives a bench run, so that it can fire under the driver)
        job_s = float(os.environ.get("SDFR_BENCH_JOB_WATCHDOG_S", "540"))
        Watchdog.job_deadline = time.monotonic() + job_s
        faulthandler.dump_traceback_later(job_s, exit=True)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (this pool's host driver supports only dmabuf IPC: without this RCCL's set-up fails with "hipIpcGetMemHandle:
        # invalid argument" -- the environment's own statement; a value the node exports wins)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout is for the one result
        # line, so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            with Watchdog(rank, f"init_process_group({backend}, world_size={N})", args.watchdog_s):
                kw = dict(rank=rank, world_size=N, timeout=datetime.timedelta(seconds=max(60.0, 4 * args.watchdog_s)))
                if backend == "nccl":
                    dist.init_process_group("nccl", device_id=device, **kw)
                else:
                    dist.init_process_group(backend, **kw)
            with Watchdog(rank, "first all-reduce (communicator set-up over xGMI)", args.watchdog_s):
                warm = torch.ones(1, device=device)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
                if int(warm.item()) != N:
                    raise SystemExit(f"bench.py rank {rank}: the first all-reduce over {N} ranks summed to "
                                     f"{warm.item()}, expected {N}")
            with Watchdog(rank, "first 1 MiB all-reduce", args.watchdog_s):
                warm = torch.ones(64 ** 3, device=device)
                dist.all_reduce(warm)
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from sdfest_amd import BatchRenderPlan, Camera
    from tools.bench_extra import Telemetry, extra_configs, parity_report

    tele = Telemetry(local_rank)
    tele.start()
    W, H, B, thr = args.width, args.height, args.batch, 0.005
    sdf_np, poses_np, sdf, pos, quat, isc, g = synthetic_inputs(B * N, rank, B, W, H, device)
    cam = Camera(W, H, W / 2.0, W / 2.0, W / 2.0, H / 2.0, pixel_center=0.5)
    # The one exchange of a step -- the all-reduce of d/dSDF (RCCL) -- runs beside the following steps.  The plan
    # cycles through M gradient volumes that are ONE contiguous buffer (a step's forward zero-fills the volume its
    # backward will add into); every M/2 steps the M/2 volumes just written are summed over the ranks by ONE
    # all-reduce (1 MiB per volume is latency-bound over xGMI: four volumes cost one latency, not four), and the
    # compute stream waits for an exchange only when its half of the ring comes up for re-use, M/2 steps later.
    # Every cross-stream dependency is a queue packet of ~10 us on this platform: this way the
    # two streams meet once per M/2 steps in each direction instead of once per step.  Same bytes exchanged per
    # step, nothing skipped; the exchanges still outstanding at the end are waited for inside the timed region.
    M = max(2, int(os.environ.get("SDFR_BENCH_GRAD_VOLUMES", "8" if use_dist else "2")))
    M += M % 2
    half = M // 2
    # The SDFR_BWD_HALF_GRID hint of include/sdfr.h is chosen by the plan itself (close_views="auto", its default):
    # the forward's prologue counts the close views on the device into a pinned host word and each backward reads
    # whatever count has arrived -- this file never looks at the poses.
    # (SDFR_BENCH_CLOSE_VIEWS=1 / 0 forces the hint on / off: the counter passes of tools/profile_gpu.sh run 5 cold
    # steps under a serialising profiler, too few for the count to arrive -- they force what the steady state chooses)
    cv = {"1": True, "0": False}.get(os.environ.get("SDFR_BENCH_CLOSE_VIEWS", ""), "auto")
    plan = BatchRenderPlan(64, B, cam, device=device, grad_volumes=M, close_views=cv)
    state = {"k": 0, "pending": [None, None], "mode": args.exchange if use_dist else "none"}
    # (begin, end) around every synchronous exchange of one timed region, made before it (no allocation inside)
    coll_pool = ([(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(args.steps)] if use_dist else [])
    state["n_coll_events"] = 0

    def finish_exchanges():
        for i, w in enumerate(state["pending"]):
            if w is not None:
                w.wait()
                state["pending"][i] = None

    def step_sync(ev=None, time_collective=False):
        """forward, backward, and the all-reduce of THIS step's volume finished before the next forward may start:
        the dependency structure of render-and-compare (the next SDF needs this gradient)."""
        if ev:
            ev[0].record()
        plan.forward(sdf, pos, quat, isc, thr, prepare_backward=True)
        if ev:
            ev[1].record()
        g_sdf = plan.backward(g, sdf, pos, quat, isc)[0]
        if ev:
            ev[2].record()
        # Issued on the compute stream: c10d makes its own collective stream wait for the backward and the compute
        # stream wait for the sum.  (A side stream of this file's own around the call was measured first: two more
        # stream crossings per step, and stalls of up to 19 ms in single steps -- 0.30 ... 0.44 ms per step against
        # 0.267 this way, one rank under torch.distributed.run.)  The two events bracket the collective on the compute
        # stream: the time from the end of the backward until the next forward may start.
        ev_pair = None
        if time_collective and state["n_coll_events"] < len(coll_pool):
            ev_pair = coll_pool[state["n_coll_events"]]
            state["n_coll_events"] += 1
            ev_pair[0].record()
        dist.all_reduce(g_sdf, op=dist.ReduceOp.SUM)
        if ev_pair:
            ev_pair[1].record()
        state["k"] += 1

    def step(ev=None, time_collective=False):
        if state["mode"] == "sync":
            return step_sync(ev, time_collective)
        k = state["k"]
        if ev:
            ev[0].record()
        if use_dist and k % half == 0:
            h = (k // half) % 2           # the half of the ring this step starts to re-use
            if state["pending"][h] is not None:
                state["pending"][h].wait()
                state["pending"][h] = None
        # forward + backward of the same views = one step (sdfr_render_step_forward / _backward)
        plan.forward(sdf, pos, quat, isc, thr, prepare_backward=True)
        if ev:
            ev[1].record()
        plan.backward(g, sdf, pos, quat, isc)
        if ev:
            ev[2].record()
        state["k"] = k + 1
        if use_dist and (k + 1) % half == 0:
            h = (k // half) % 2
            state["pending"][h] = dist.all_reduce(plan.grad_ring[h * half:(h + 1) * half], op=dist.ReduceOp.SUM,
                                                  async_op=True)

    def flush_tail():
        """exchange the volumes of the steps since the last full half (K need not be a multiple of M/2)"""
        k = state["k"]
        if state["mode"] == "sync":
            return
        if use_dist and k % half:
            h = (k // half) % 2
            if state["pending"][h] is not None:
                state["pending"][h].wait()
            state["pending"][h] = dist.all_reduce(plan.grad_ring[h * half:h * half + k % half],
                                                  op=dist.ReduceOp.SUM, async_op=True)
            state["k"] = (k // half + 1) * half

    def barrier():
        flush_tail()
        finish_exchanges()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        plan.ring_reset()       # step k writes volume k % M again
        state["k"] = 0

    # untimed pre-warm: the driver's run is 25 steps = 7 ms from a cold, idle GPU, shorter than the clock governor's
    # ramp (round 2: 855 k renders/s on the driver's box against 903 k in a longer session).  The steady state is
    # what the metric means, so the GPU runs the same steps for `--prewarm-ms` before the W warm-up steps; the
    # telemetry in the line shows the clocks of the timed region.
    def ranks_agree(go_on):
        """A loop that ends by the CLOCK must end after the same number of rounds on every rank -- each round issues
        collectives, and ranks that disagree by one round would wait for each other in different collectives for
        ever.  Rank 0's clock decides (one tiny broadcast per round, untimed regions only)."""
        if not use_dist:
            return go_on
        flag = torch.tensor([1 if go_on else 0], device=device, dtype=torch.int32)
        dist.broadcast(flag, src=0)
        return bool(flag.item())

    t_pre = time.perf_counter()
    n_pre = 0
    while ranks_agree((time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms):
        for _ in range(16):
            step()
        n_pre += 16
        torch.cuda.synchronize()
    barrier()
    prewarm_ms = (time.perf_counter() - t_pre) * 1e3
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the two calls of every `stride`-th timed step (each record is a packet of its own between
    # two kernels; SDFR_BENCH_EVENT_STRIDE=1 brackets every step)
    stride = max(1, int(os.environ.get("SDFR_BENCH_EVENT_STRIDE", "4")))
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] if k % stride == 0 else None
              for k in range(args.steps)]
    fallbacks0 = plan.prologue_fallbacks()
    half0 = plan.half_grid_steps

    def timed(mode, evs, time_collective=False, mark=False):
        """EXACTLY args.steps steps between two barrier + synchronize pairs; the max over ranks"""
        state["mode"] = mode
        if mode != state.get("warmed"):
            # the other exchange gets its own untimed run-in (its first collectives on a fresh stream grow c10d's and
            # HIP's event pools: measured cold, 200 sync steps took twice the time they take after 100 ms of the same)
            t_w = time.perf_counter()
            while ranks_agree((time.perf_counter() - t_w) * 1e3 < min(100.0, max(args.prewarm_ms, 1.0))):
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
        barrier()
        if mark:
            tele.mark("t0")      # (also reads the clocks here, just outside the timed region)
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(evs[k] if evs else None, time_collective)
        t_issue = time.perf_counter() - t0
        barrier()
        el = time.perf_counter() - t0
        if mark:
            tele.mark("t1")
        if use_dist:
            tt = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, t_issue

    main_mode = state["mode"]
    state["warmed"] = main_mode
    # (t_enqueued: host time to issue the K steps -- GPU-bound if below `elapsed`)
    elapsed, t_enqueued = timed(main_mode, events, time_collective=True, mark=True)
    events = [e for e in events if e is not None]
    # the other exchange, K steps of its own after the headline's region (N > 1 only)
    elapsed_other = None
    if use_dist:
        other_mode = "sync" if main_mode == "ring" else "ring"
        elapsed_other, _ = timed(other_mode, None, time_collective=True)
        state["mode"] = main_mode
    time.sleep(0.005)
    tele.stop()
    def plain_barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    bwd_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
    step_ms = np.array([e[0].elapsed_time(e[2]) for e in events])
    hits = int((plan.depth > 0).sum().item())
    fallbacks = plan.prologue_fallbacks() - fallbacks0
    collective = None
    if use_dist:
        # what the process group really is: the driver can check that RCCL saw N ranks and every rank rendered
        per_rank = [None] * N
        dist.all_gather_object(per_rank, {"rank": rank, "hit_pixels": hits, "device": torch.cuda.get_device_name(device),
                                          "local_device_index": local_rank, "prologue_fallbacks": fallbacks})
        ar_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in coll_pool[:state["n_coll_events"]]]
        collective = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                      "exchange_of_value": main_mode,
                      "exchange_ring": f"one all-reduce (sum, fp32) of {half} x 1 MiB d/dSDF volumes per {half} steps, "
                                       f"waited for {half} steps later (ring of {M} volumes): steps are independent",
                      "exchange_sync": "one all-reduce (sum, fp32) of the step's 1 MiB d/dSDF volume, finished before "
                                       "the next step's forward starts (the dependency of render-and-compare)",
                      # span of the per-step all-reduce as the compute stream sees it: from the end of the backward
                      # until the next forward may start (events on that stream; the sync-exchange region)
                      "allreduce_us": ({"median": round(float(np.median(ar_us)), 1), "min": round(min(ar_us), 1),
                                        "max": round(max(ar_us), 1), "n": len(ar_us)} if ar_us else None),
                      "per_rank": per_rank}
        # the sync exchange read against its own parts: a step that waits for its sum costs the
        # rank's compute (forward + backward calls, events) + the all-reduce span; the ring exchange hides the
        # collective, so its per-GPU rate should stay the single-GPU rate (weak scaling ~N).  The xGMI figure is
        # SURVEY section 5's model of a 1 MiB ring all-reduce: 2 (N - 1) hops of (1 MiB / N) / 153 GB/s + ~2 us each.
        hop_us = (64 ** 3 * 4 / max(N, 1)) / 153e9 * 1e6 + 2.0
        collective["sync_exchange_model"] = {
            "compute_ms_per_step_events": round(fwd_ms + bwd_ms, 4),
            "allreduce_us_measured_median": (round(float(np.median(ar_us)), 1) if ar_us else None),
            "sum_ms_per_step": (round(fwd_ms + bwd_ms + float(np.median(ar_us)) * 1e-3, 4) if ar_us else None),
            "allreduce_us_predicted_xgmi_ring": round(2 * (N - 1) * hop_us, 1) if N > 1 else None,
            "note": "ms_per_step_sync_exchange (measured) against sum_ms_per_step; value_ring_exchange / n_gpus against "
                    "the single-GPU value"}

    if rank == 0:
        views = B * N * args.steps
        value = views / elapsed
        R3 = 64 ** 3
        bytes_per_view = 12 * W * H + 12 * R3 / B + 32          # fwd+bwd
        fwd_bytes = B * (4 * W * H + 16) + 4 * R3               # forward launch: depth out, sdf in
        bwd_bytes = B * (8 * W * H + 16) + 8 * R3               # backward launch: 2 images in, sdf in, g_sdf out
        dominant = "render_forward_kernel" if fwd_ms >= bwd_ms else "render_backward_kernel"
        k_bytes, k_ms = (fwd_bytes, fwd_ms) if fwd_ms >= bwd_ms else (bwd_bytes, bwd_ms)
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        # the committed PMC passes were taken on the default workload only
        traffic, traffic_source = (load_traffic() if (B == 256 and W == 640 and H == 480)
                                   else (None, "counter passes exist for the default workload only"))
        line = {
            "metric": "depth renders/sec fwd+bwd, 640×480 @ 64³ SDF; grad max-abs-err vs ref",
            "value": round(value, 1),
            "unit": "renders/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"C3: {B} seeded random poses per GPU of the blobs(0) 64^3 SDF, "
                                   f"{W}x{H}, threshold 0.005, forward+backward"
                                   + ((", RCCL all-reduce of dSDF "
                                       + (f"batched over {half} steps and overlapped with the following ones"
                                          if main_mode == "ring" else "after every step, before the next forward"))
                                      if N > 1 and backend == "nccl" else "")
                                   + (f", REHEARSAL: {backend} backend" if backend != "nccl" else "")
                                   + (", REHEARSAL: all ranks on GPU 0" if share_gpu else ""),
                       "views_per_gpu": B, "width": W, "height": H, "sdf_resolution": 64,
                       "parallelism": f"views sharded over {N} GPU(s)",
                       "backward_half_grid": {"chosen_by": ("device-side count of close views (pinned word), "
                                                            "previous steps' value; no host poses") if cv == "auto"
                                              else f"forced {cv} by SDFR_BENCH_CLOSE_VIEWS (profiling)",
                                              "timed_steps_with_hint": plan.half_grid_steps - half0,
                                              "close_views_last_counted": plan.close_views_seen()[1]},
                       "hit_pixels_rank0": hits},
            "roofline": {"bound": "hbm", "kernel": dominant,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved * 1e9 / HBM_PEAK, 5),
                         "traffic": (traffic or {}).get(dominant),
                         "traffic_source": traffic_source,
                         # the resource that BINDS this kernel (it is not an HBM kernel): the CU's vector-memory
                         # pipeline, from the same committed counter passes (tools/summarize_pmc.py has the formulas:
                         # bytes = 16 B x 64 lane slots x vector read instructions; unit busy = *_BUSY / (cycles x CUs);
                         # valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles))
                         "binding": ((traffic or {}).get("binding") or {}).get(dominant),
                         # the kernel alone (the call above includes the forward's prologue launch): its average
                         # duration in the committed rocprofv3 kernel trace of the same command
                         "kernel_trace": (lambda us: {"avg_launch_us": us, "frac": round(k_bytes / (us * 1e-6) / HBM_PEAK, 5)}
                                          if us else None)(((traffic or {}).get("kernel_avg_us_from_trace") or {}).get(dominant)),
                         "algorithmic_bytes_per_launch": int(k_bytes),
                         "avg_launch_ms": round(k_ms, 4)},
            "roofline_step": {"bytes_per_view": bytes_per_view,
                              "achieved": round(bytes_per_view * value / N / 1e9, 2),
                              "unit": "GB/s per GPU",
                              "frac": round(bytes_per_view * value / N / HBM_PEAK, 5),
                              # also against the measured copy ceiling of the part (6.29 TB/s)
                              "frac_of_measured_copy_peak": round(bytes_per_view * value / N / HBM_COPY_PEAK, 5)},
            "kernel_ms": {"forward_call": round(fwd_ms, 4), "backward_call": round(bwd_ms, 4)},
            "step_ms_events": {"min": round(float(step_ms.min()), 4), "median": round(float(np.median(step_ms)), 4),
                               "max": round(float(step_ms.max()), 4), "n": int(step_ms.size)},
            "host_issue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
            "prewarm": {"untimed_steps": n_pre, "ms": round(prewarm_ms, 1),
                        "why": "clock ramp: the timed region follows >= --prewarm-ms of the same steps"},
            "prologue_fallbacks_in_timed_region": fallbacks,
            "gpu_telemetry": tele.summary("t0", "t1"),
        }
        if collective:
            line["collective"] = collective
            v_other = views / elapsed_other
            line["value_ring_exchange"] = round(value if main_mode == "ring" else v_other, 1)
            line["value_sync_exchange"] = round(value if main_mode == "sync" else v_other, 1)
            line["ms_per_step_sync_exchange"] = round((elapsed if main_mode == "sync" else elapsed_other)
                                                      / args.steps * 1e3, 4)
        if N == 1 and not args.no_parity:
            # the other half of the metric: errors of the benchmarked build's last step against the reference
            # semantics (sdf_renderer_cuda.cu:334-467, simple_renderer.py:317-458) through the pinned oracle
            par = parity_report(plan, g, sdf, (pos, quat, isc), sdf_np, poses_np, W, H, thr)
            line["parity"] = par
            # the metric's second half: d/dSDF of the benchmarked step (max-abs, and relative to its maximum), and
            # the plain relative error of the well-conditioned pose-gradient components (`parity` has the rest)
            # (named for what they are -- ADVICE r3: the filtered relative error is not "the maximum")
            line["grad_max_abs_err"] = par["grad_sdf"]["max_abs_err"]
            line["grad_sdf_max_err_over_max"] = par["grad_sdf"]["max_err_over_max"]
            line["grad_pose_max_rel_err_well_conditioned_ones_upstream"] = \
                par["ones_upstream"]["grad_pose_max_rel_err_well_conditioned"]
            line["grad_pose_max_abs_err_benchmark_upstream"] = par["grad_pose_benchmark_upstream"]["max_abs_err"]
            line["grad_pose_max_err_over_sum_of_term_magnitudes"] = \
                par["grad_pose_benchmark_upstream"]["max_err_over_sum_of_term_magnitudes"]
        if N == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or min(B, 256)
            line["cpu_baseline"] = cpu_baseline(sdf_np, poses_np, W, H, thr, sample)
            line["speedup_vs_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        if N == 1 and not args.no_configs:
            del plan
            torch.cuda.empty_cache()
            line["configs"] = extra_configs(sdf_np, device, HBM_PEAK)
    line = line if rank == 0 else {}
    # the extras of a multi-rank run, AFTER the headline is complete and each under its own guard (every rank enters;
    # only rank 0's line matters): the sharded loop between graphs, then (opt-in) with its all-reduce inside the graphs
    extras = use_dist and (N > 1 or os.environ.get("SDFR_BENCH_LOOP_SHARDED") == "1")
    if extras and rank == 0:
        # The finished measurement goes out NOW: a native abort inside an extra -- a fault in a collective, the
        # process group's watchdog, the job watchdog -- kills the process without a word, and the guards above only
        # cover hangs and Python exceptions.  The line is printed again, augmented, at the end: readers take the last.
        print(json.dumps(line), flush=True)
    if extras:
        with SectionGuard(rank, "loop_sharded", float(os.environ.get("SDFR_BENCH_LOOP_SHARDED_S", "180")), line):
            try:
                line["loop_sharded"] = sharded_loop_section(N, rank, plain_barrier)
            except Exception as e:      # (every rank takes the same path: the section's collectives are all inside)
                line["loop_sharded"] = {"error": f"{type(e).__name__}: {e}"}
    if extras and backend == "nccl" and os.environ.get("SDFR_BENCH_GRAPH_COLLECTIVE") == "1":
        # opt-in until it has run once over several GPUs (it has only ever captured RCCL's all-reduce with ONE rank)
        # (every rank enters; only rank 0's line matters)
        line["loop_sharded_collective_in_graph"] = collective_in_graph_section(
            N, rank, plain_barrier, float(os.environ.get("SDFR_BENCH_GRAPH_COLLECTIVE_S", "120")), line)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        # (the line is out: a rank that never arrives here must not turn the run into a failure)
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()

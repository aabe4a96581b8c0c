"""Where a synchronous per-step all-reduce spends its time on the host (one rank under torch.distributed.run):
per-call host time of dist.all_reduce on the compute stream / on a side stream, with and without async_op."""
import os, sys, time
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def main():
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g = torch.ones(64 ** 3, device=dev)
    work = torch.empty(64 * 1024 * 1024, device=dev)      # ~0.25 ms of GPU work per "step"
    dist.all_reduce(g); torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    def busy(): work.add_(1.0)
    def t(fn, n=200):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter(); hosts = []
        for _ in range(n):
            h0 = time.perf_counter(); fn(); hosts.append(time.perf_counter() - h0)
        issue = time.perf_counter() - t0
        torch.cuda.synchronize(); total = time.perf_counter() - t0
        hosts.sort()
        return f"total {total / n * 1e6:7.1f} us/step, host issue {issue / n * 1e6:7.1f}, host median per call {hosts[n // 2] * 1e6:6.1f}"
    def only_work(): busy()
    def same_stream(): busy(); dist.all_reduce(g)
    def same_stream_async():
        busy(); h = dist.all_reduce(g, async_op=True); h.wait()
    def side_stream():
        busy(); cur = torch.cuda.current_stream(dev); side.wait_stream(cur)
        with torch.cuda.stream(side): dist.all_reduce(g)
        cur.wait_stream(side)
    for name, fn in (("work only", only_work), ("all_reduce on the compute stream", same_stream),
                     ("async_op + wait()", same_stream_async), ("all_reduce on a side stream", side_stream)):
        print(f"{name:36s} {t(fn)}", flush=True)
    dist.destroy_process_group()
main()

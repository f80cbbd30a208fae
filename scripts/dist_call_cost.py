import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.zeros(160, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
for _ in range(10): dist.all_reduce(t)
torch.cuda.synchronize()
for mode in ("sync current stream", "sync side stream ctx", "async + wait later"):
    t0 = time.perf_counter()
    ws = []
    for _ in range(200):
        if mode == "sync current stream": dist.all_reduce(t)
        elif mode == "sync side stream ctx":
            with torch.cuda.stream(s): dist.all_reduce(t)
        else: ws.append(dist.all_reduce(t, async_op=True))
    host = (time.perf_counter() - t0) / 200 * 1e6
    for w in ws: w.wait()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 200 * 1e6
    print("%-24s host %.1f us per call, incl. device drain %.1f us" % (mode, host, tot))
# device-side cost: kernel A -> allreduce -> kernel B chain timed with events
a = torch.zeros(1 << 10, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for with_ar in (0, 1):
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        a.add_(1.0)
        if with_ar: dist.all_reduce(t)
    e1.record(); e1.synchronize()
    print("chain of 200 x (tiny kernel%s): %.1f us per link" % (" + all_reduce" if with_ar else "", e0.elapsed_time(e1) * 1e3 / 200))
dist.destroy_process_group()

import torch, time
x = torch.zeros(31*1024*1024, dtype=torch.uint8, device="cuda")
for pinned in (True, False):
    h = torch.empty(31*1024*1024, dtype=torch.uint8, pin_memory=pinned)
    for n in (31*1024*1024, 6*1024*1024):
        h[:n].copy_(x[:n]); torch.cuda.synchronize()
        t=time.perf_counter()
        for _ in range(5):
            h[:n].copy_(x[:n], non_blocking=True); torch.cuda.synchronize()
        dt=(time.perf_counter()-t)/5
        print(f"pinned={pinned} {n/2**20:.0f} MiB: {dt*1e3:.2f} ms = {n/dt/1e9:.2f} GB/s")

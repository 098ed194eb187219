import torch, time
for mb in (33.5, 67, 268):
    n = int(mb * 1e6 / 4)
    x = torch.empty(n, device="cuda"); y = torch.randn(n, device="cuda")
    for name, f in (("fill", lambda: x.zero_()), ("copy", lambda: x.copy_(y))):
        for _ in range(5): f()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        print(f"{name} {mb} MB: {us:.1f} us  write {mb/us*1e-3*1e3:.2f} GB/ms = {mb*1e6/us/1e6:.2f} TB/s")

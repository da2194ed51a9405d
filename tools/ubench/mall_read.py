import torch
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    x = torch.ones(mb * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3): s = x.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{mb:5d} MB  sum: {t:7.3f} ms  {mb / 1024 / t * 1e3 / 1e3:6.2f} TB/s")

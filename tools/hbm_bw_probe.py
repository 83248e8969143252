import torch
x = torch.empty(1<<29, dtype=torch.bfloat16, device='cuda'); y = torch.empty_like(x)
for f, name, nbytes in ((lambda: y.copy_(x), "copy", 2 * x.numel() * 2), (lambda: y.zero_(), "fill", x.numel() * 2), (lambda: torch.add(x, x, out=y), "read1+write1 add", 2 * x.numel() * 2)):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    print(name, nbytes * 10 / (s.elapsed_time(e) * 1e-3) / 1e12, "TB/s")

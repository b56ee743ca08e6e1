import torch
x = torch.empty(2**30, dtype=torch.bfloat16, device="cuda")
y = torch.empty_like(x)
for name, fn, gb in (("zero_ (write 2.1 GB)", lambda: x.zero_(), 2.147), ("copy_ (read+write 4.3 GB)", lambda: y.copy_(x), 4.295)):
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[2]
    print(f"{name}: {t:.3f} ms = {gb / t:.2f} TB/s")

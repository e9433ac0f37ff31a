import torch, time
dev = "cuda:0"
def bw(nbytes, reps=20, op="copy"):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    fn = (lambda: b.copy_(a)) if op == "copy" else (lambda: a.sum()) if op == "read" else (lambda: b.fill_(1.0))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    moved = nbytes * (2 if op == "copy" else 1)
    return moved / t / 1e12, t * 1e6
for mb in (16, 64, 100, 200, 512, 2048):
    for op in ("copy", "read", "fill"):
        tb, us = bw(mb << 20, op=op)
        print(f"{op:5s} {mb:5d} MiB: {tb:6.2f} TB/s  ({us:8.1f} us)")

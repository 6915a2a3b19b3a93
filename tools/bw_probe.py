"""Pure-store / pure-load / copy bandwidth of the box, as torch sees it (context for the roofline fractions in DESIGN.md)."""
import torch
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")   # 4 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
gb = x.numel() * 4 / 1e9
print("fill  (store only)  %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("sum   (load only)   %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("copy  (load+store)  %.2f TB/s moved" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
small = torch.empty(26 << 20, dtype=torch.float32, device="cuda")    # 104 MB, rewritten in place
print("fill 104 MB in place %.2f TB/s" % (small.numel() * 4 / 1e9 / t(lambda: small.fill_(1.0), 50) / 1e3))

"""What does this MI355X sustain for plain streaming writes, reads and copies?  (torch elementwise kernels on 8 GiB; the record path
of log2m >= 17 writes 40 GB and reads 42 GB per 64 x 5 Mbp step: is 2.4-2.7 TB/s of writes the chip's ceiling or the kernel's?)"""
import time, torch
n = 8 << 30
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
xi = x.view(torch.int32); yi = y.view(torch.int32)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
w = t(lambda: xi.fill_(7))
print(f"fill   8 GiB: {w*1e3:.2f} ms = {n/w/1e12:.2f} TB/s written")
r = t(lambda: xi.sum())
print(f"sum    8 GiB: {r*1e3:.2f} ms = {n/r/1e12:.2f} TB/s read")
c = t(lambda: yi.copy_(xi))
print(f"copy   8 GiB: {c*1e3:.2f} ms = {n/c/1e12:.2f} TB/s read + {n/c/1e12:.2f} TB/s written")
a = t(lambda: yi.add_(1))
print(f"add_   8 GiB: {a*1e3:.2f} ms = {n/a/1e12:.2f} TB/s read + as much written (in place)")
for mb in (64, 128, 192, 256, 512):   # does a buffer that fits the 256 MiB memory-side cache stream faster?
    m = mb << 20
    s = xi[: m // 4]
    w = t(lambda: s.fill_(3), 50); r = t(lambda: s.sum(), 50)
    print(f"{mb:4d} MiB buffer: fill {m/w/1e12:.2f} TB/s, sum {m/r/1e12:.2f} TB/s")

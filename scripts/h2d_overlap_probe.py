import torch, time
n = 101 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        d.copy_(h, non_blocking=True)
    s.synchronize(); dt = time.perf_counter() - t0
    print(f"H2D idle: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")
# busy GPU: big matmul loop on default stream
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    torch.cuda.synchronize()
    for _ in range(20): b = a @ a
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        d.copy_(h, non_blocking=True)
    s.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"H2D while GPU busy: {dt*1e3:.2f} ms = {n/dt/1e9:.1f} GB/s")

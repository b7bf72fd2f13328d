import torch, time
p = torch.cuda.get_device_properties(0)
print(p)
print("CUs", p.multi_processor_count, "clock", getattr(p, "clock_rate", None), "L2", p.L2_cache_size)
for dt in (torch.float16, torch.bfloat16):
    a = torch.randn(8192, 8192, device="cuda", dtype=dt); b = torch.randn(8192, 8192, device="cuda", dtype=dt)
    for _ in range(3): c = a @ b
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): c = a @ b
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(dt, "8192^3 GEMM", 2 * 8192**3 / t / 1e12, "TFLOP/s")
x = torch.empty(1 << 30, device="cuda", dtype=torch.uint8); y = torch.empty_like(x)
for _ in range(2): y.copy_(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
print("copy 1GiB:", 2 * (1 << 30) / t / 1e12, "TB/s")

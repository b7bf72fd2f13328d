"""Reference point for the conv roofline: the vendor GEMM (torch.matmul -> hipBLASLt) on this device, f16 and bf16,
8192^3, uniform random [-1,1) operands, >= 2 s of back-to-back launches.  Prints one JSON line per dtype."""
import json, time, torch
n = 8192
for dt in (torch.float16, torch.bfloat16):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).to(dt)
    b = (torch.rand(n, n, device="cuda") * 2 - 1).to(dt)
    c = torch.empty(n, n, device="cuda", dtype=dt)
    for _ in range(5):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < 2.0:
        for _ in range(20):
            torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        iters += 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        torch.matmul(a, b, out=c)
    ev1.record()
    torch.cuda.synchronize()
    s = ev0.elapsed_time(ev1) / 1e3 / 50
    print(json.dumps({"op": f"torch.matmul {n}^3", "dtype": str(dt), "ms": round(s * 1e3, 4),
                      "tflops": round(2 * n ** 3 / s / 1e12, 1), "frac_of_2500": round(2 * n ** 3 / s / 2.5e15, 3)}))

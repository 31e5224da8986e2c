"""what the vendor library reaches on this box for FP64 products of the trailing update's shapes (calibration of 'achievable')"""
import torch, time
dev = "cuda"
def bench(m, n, k, reps=20):
    a = torch.randn(m, k, device=dev, dtype=torch.float64); b = torch.randn(k, n, device=dev, dtype=torch.float64); c = torch.randn(m, n, device=dev, dtype=torch.float64)
    for _ in range(3): torch.addmm(c, a, b, alpha=-1.0)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): torch.addmm(c, a, b, alpha=-1.0, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"C({m}x{n}) -= A({m}x{k}) B({k}x{n}): {ms:.3f} ms, {2.0*m*n*k/ms/1e9:.1f} TFLOP/s FP64", flush=True)
for (m, n, k) in [(8192, 8192, 8192), (6000, 6000, 128), (6000, 6000, 256), (6000, 6000, 512), (12000, 12000, 256), (24000, 24000, 256), (24000, 24000, 2048)]:
    bench(m, n, k)

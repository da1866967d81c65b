"""Reference point: the stock (rocBLAS / hipBLASLt) fp32 batched GEMM at the FFM's product shapes."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, P = 8, 16384
for M, K in ((256, 128), (128, 256), (256, 384)):
    w = torch.randn(M, K, device="cuda")
    x = torch.randn(B, K, P, device="cuda")
    out = torch.empty(B, M, P, device="cuda")
    t = timeit(lambda: torch.matmul(w, x, out=out))
    print(f"W({M}x{K}) @ X({B}x{K}x{P}): {t:.1f} us  {2.0 * B * M * K * P / t / 1e6:.1f} TF/s")
# dW shape: dz (B, 256, P) x fsp (B, 128, P)^T summed over images
dz = torch.randn(B, 256, P, device="cuda")
fs = torch.randn(B, 128, P, device="cuda")
t = timeit(lambda: torch.einsum("bop,bcp->oc", dz, fs))
print(f"dW einsum: {t:.1f} us  {2.0 * B * 256 * 128 * P / t / 1e6:.1f} TF/s")

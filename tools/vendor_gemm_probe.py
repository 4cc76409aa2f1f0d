"""What the vendor library reaches on the production GEMM shapes in plain fp16 / bf16 (one pass): a ceiling check for the
k-loop skeleton, not part of the product.  usage: python tools/vendor_gemm_probe.py"""
import torch, time
M = 512 * 1214
for dt in (torch.float16, torch.bfloat16):
    for name, N, K in (("qkv", 2304, 768), ("fc1", 3072, 768), ("o", 768, 768), ("fc2", 768, 3072)):
        x = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
        for _ in range(3):
            y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = torch.nn.functional.linear(x, w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{dt} {name} M={M} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
        del x, w, y

"""Practical MFMA ceiling on this box: hipBLASLt / rocBLAS bf16 GEMMs of the implicit-GEMM shapes of the step's conv layers."""
import torch, json
for (M, K, N) in [(131072, 4608, 512), (524288, 2304, 256), (2097152, 1152, 128), (8192, 8192, 8192), (32768, 4608, 512)]:
    a = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(K, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): torch.mm(a, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): torch.mm(a, b)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(json.dumps(dict(M=M, K=K, N=N, ms=round(ms, 4), TFLOPs=round(2.0 * M * K * N / ms / 1e9, 1))))
    del a, b

"""How fast is the vendor fp32 GEMM (rocBLAS / hipBLASLt through torch.mm) on the GEMM shapes of the pix2pix layers?
Context for the roofline fraction in DESIGN.md: not part of the product, nothing here is used by it."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(73728, 128, 1024), (18432, 256, 2048), (4608, 512, 4096), (1152, 512, 8192), (73728, 256, 1024),
          (8192, 8192, 8192), (16384, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device='cuda')
    b = torch.randn(K, N, device='cuda')
    for _ in range(5):
        torch.mm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        torch.mm(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('sgemm M=%6d N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s' % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))

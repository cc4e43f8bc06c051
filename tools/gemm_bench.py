"""Time tir_gemm_tn (weight-gradient GEMMs of the training step) at the shapes the decoders use."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensoir_amd import ops

dev = torch.device("cuda", 0)
res = []
for impl in ("mfma", "bf16x3"):
 for n in (30000, 231000, 400000):
    for (M, lda, N, ldb, ones) in ((128, 128, 150, 160, True), (128, 128, 128, 128, True), (4, 4, 128, 128, True),
                                   (27, 28, 144, 144, False)):
        A = torch.randn(n, lda, device=dev)
        B = torch.randn(n, ldb, device=dev)
        C = torch.zeros(M if M > 4 else 4, 160 + 4, device=dev)
        for _ in range(3):
            ops.gemm_tn(A, M, B, N, C, ones, impl=impl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_tn(A, M, B, N, C, ones, impl=impl)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        C.zero_(); ops.gemm_tn(A, M, B, N, C, ones, impl=impl)
        ref = A[:, :M].double().T @ B[:, :N].double()
        err = float((C[:M, :N].double() - ref).abs().max() / ref.abs().max())
        fl = 2.0 * n * M * (N + ones)
        res.append({"impl": impl, "n": n, "M": M, "N": N, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2), "rel_err": err})
        print(res[-1], flush=True)
json.dump(res, open(os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "gemm_bench.json"), "w"), indent=1)

"""One shape of the weight-gradient GEMM, a few launches -- for rocprofv3 --pmc runs (tools/pmc_one.sh)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensoir_amd import ops
n, M, N = 231000, 128, 150
A = torch.randn(n, 128, device="cuda"); B = torch.randn(n, 160, device="cuda"); C = torch.zeros(128, 164, device="cuda")
for impl in ("bf16x3", "mfma"):
    for _ in range(5):
        ops.gemm_tn(A, M, B, N, C, True, impl=impl)
torch.cuda.synchronize()

"""GPU probe: the four GEMM shapes of one CLIP text block at the ragged row count of the bench (M ~ 2464)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import gemm_case
for M in (2464, 4928):
    for (N, K, act) in [(1536, 512, 0), (512, 512, 0), (2048, 512, 2), (512, 2048, 0)]:
        gemm_case(M, N, K, act)
gemm_case(8192, 8192, 8192, 0)

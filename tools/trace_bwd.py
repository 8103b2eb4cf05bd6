"""Phase breakdown of the dgrad kernel (mlp_bwd_h16_kernel) from in-kernel clock stamps of wave 0 of one workgroup.
Needs a library whose mlp_bf16.o was compiled with -DPLNERF_TRACE=<block> and linked with the other objects of csrc/
(PLNERF_HIP_LIB=<that library> PLNERF_ALLOW_TOOLS_BUILD=1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plnerf_amd as P
from plnerf_amd import _lib
dev = torch.device("cuda:0")
R = 4096
torch.manual_seed(0)
net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision="bf16").to(dev)
pts = (torch.rand(R, 192, 3, device=dev) * 2 - 1) * 3
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
lib = _lib.lib()
lib.plnerf_debug_trace_f16.argtypes = [ctypes.c_void_p]
lib.plnerf_debug_trace_f16.restype = ctypes.c_int
for _ in range(3):
    raw = net.query(pts, vd)       # bf16 forward: the f16 trace buffer is then written by the dgrad kernel only
    raw.sum().backward()
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
assert lib.plnerf_debug_trace_f16(buf.ctypes.data) == 0
t = buf.astype(np.int64)
print(f"dgrad: total {t[42]-t[0]} clk;  views+feature+L7 part {t[1]-t[0]}")
if t[43] and not os.environ.get("PLNERF_TRACE_PIPE"):
    print(f"  head: mask requests + g_raw load + barrier {t[43]-t[0]}  dz_view + barrier {t[44]-t[43]}  mask commit + K(view, 8 steps) {t[45]-t[44]}  "
          f"barrier + epilogue + barrier {t[46]-t[45]}  K(feature) {t[47]-t[46]}  barrier + epilogue(alpha) + barrier {t[1]-t[47]}")
names = ("K(B) + epilogue(A)", "barrier", "K(A, next) + epilogue(B)", "barrier") if os.environ.get("PLNERF_TRACE_PIPE") else \
        ("K loop", "barrier", "store_dz", "barrier")
for k in range(7):
    b = 2 + 5 * k
    print(f"  layer {7-k}: loop top {t[b]-(t[1] if k == 0 else t[b-1]):6d}  " +
          "  ".join(f"{n} {t[b+1+i]-t[b+i]:6d}" for i, n in enumerate(names)))

"""Phase breakdown of the fused MLP forward (16-bit MFMA modes) from in-kernel clock stamps.
Needs a library built with -DPLNERF_TRACE=<block>; run via tools/trace_fwd.sh on the GPU box.
Ping-pong kernel: per layer, phase A (K loop of half A + epilogue of half B), barrier, phase B, barrier."""
import ctypes, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plnerf_amd as P
from plnerf_amd import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
train = len(sys.argv) > 2 and sys.argv[2] == "train"
R = 32768 if not train else 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=prec).to(dev)
pts = (torch.rand(R, 192, 3, device=dev) * 2 - 1) * 3
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
lib = _lib.lib()
read_trace = lib.plnerf_debug_trace_f16 if prec in ("f16x3", "f16") else lib.plnerf_debug_trace   # element type
read_trace.argtypes = [ctypes.c_void_p]
read_trace.restype = ctypes.c_int
for _ in range(3):
    if train:
        raw = net.query(pts, vd)
    else:
        with torch.no_grad():
            raw = net.query(pts, vd)
torch.cuda.synchronize()
buf = np.zeros(64, dtype=np.uint64)
assert read_trace(buf.ctypes.data) == 0
t = buf.astype(np.int64)
t0 = t[0]
wall = (t[49] - t[48]) / 100.0  # us at the 100 MHz constant clock
print(f"{prec} {'train' if train else 'inference'}: total {t[42]-t0} clk in {wall:.2f} us -> {(t[42]-t0)/max(wall,1e-9)/1e3:.3f} GHz shader clock")
print(f"  prologue (encode)      {t[1]-t0:7d}   barrier {t[2]-t[1]:6d}")
prev = t[2]
for l in range(9):
    b = 4 + 4 * l
    print(f"  layer {l}: phase A {t[b]-prev:7d}  barrier {t[b+1]-t[b]:6d}  phase B {t[b+2]-t[b+1]:6d}  barrier {t[b+3]-t[b+2]:6d}")
    prev = t[b + 3]
print(f"  tail: E(B,feat)+view   {t[41]-t[40]:7d}")
print(f"  heads + store          {t[42]-t[41]:7d}")

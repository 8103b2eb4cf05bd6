"""Shader-clock / wall-clock stamps of one workgroup of the register-resident forward kernel (a library built with
-DRR_TRACE=<block>, see tools/ab_libs.sh): cycles per tile, time per tile, effective shader clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from plnerf_amd import _lib
dev = torch.device("cuda:0")
for prec in sys.argv[1:] or ["f16x3", "f16"]:
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=prec).to(dev)
    R, S = 65536, 192
    pts = (torch.rand(R, S, 3, device=dev) * 2 - 1) * 3
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    with torch.no_grad():
        for _ in range(3):
            net.query(pts, vd)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    fn = _lib.lib().plnerf_debug_rr_trace
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    fn(buf)
    cyc, wall = buf[2] - buf[0], (buf[3] - buf[1]) * 10.0     # wall clock: 100 MHz
    print(f"{prec}: {cyc} shader cycles, {wall / 1000:.1f} us per tile -> {cyc / wall:.3f} GHz effective; "
          f"rows per tile {128 if prec.endswith('x3') else 256}; MFMA floor {3480 * 32 if prec.endswith('x3') else 2320 * 32} cycles = {100.0 * (3480 * 32 if prec.endswith('x3') else 2320 * 32) / cyc:.1f} % of the walk")

"""Shader-clock / wall-clock stamps of one workgroup of the register-resident forward kernel (a library built with
-DRR_TRACE=<block>, see tools/ab.sh -m mlp): cycles per tile, time per tile, effective shader clock."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnerf_amd as P
from plnerf_amd import _lib
dev = torch.device("cuda:0")
for prec in sys.argv[1:] or ["f16x3", "f16"]:
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True, precision=prec).to(dev)
    train = bool(os.environ.get("TRAIN"))      # TRAIN=1: the training forward (saves the backward's state), 4096 x 192 rows
    R, S = (4096 if train else 65536), 192
    pts = (torch.rand(R, S, 3, device=dev) * 2 - 1) * 3
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    with torch.set_grad_enabled(train):
        for _ in range(3):
            net.query(pts, vd)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    fn = _lib.lib().plnerf_debug_rr_trace
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    fn(buf)
    cyc, wall = buf[2] - buf[0], (buf[3] - buf[1]) * 10.0     # wall clock: 100 MHz
    print(f"{prec}: {cyc} shader cycles, {wall / 1000:.1f} us per tile -> {cyc / wall:.3f} GHz effective; "
          f"rows per tile {128 if prec.endswith('x3') else 256}; MFMA floor {3480 * 32 if prec.endswith('x3') else 2320 * 32} cycles = {100.0 * (3480 * 32 if prec.endswith('x3') else 2320 * 32) / cyc:.1f} % of the walk")
    # per unit: shader cycles from the start of unit u to the start of unit u + 1, against the unit's MFMA issue time
    split = prec.endswith("x3")
    units = []          # (layer, products) in schedule order, as csrc/mlp_rr_body.inc's unit_desc lays them out
    ks = [4, 16, 16, 16, 16, 20, 16, 16, 16, 18]
    slabs = [8, 8, 8, 8, 8, 8, 8, 8, 8, 4]
    for l in range(10):
        if ks[l] == 4:
            units += [(l, 16)] * 2
        elif ks[l] == 16:
            units += [(l, 16)] * slabs[l] if split else [(l, 32)] * (slabs[l] // 2)
        else:
            units += ([(l, ks[l] // 2), (l, ks[l] - ks[l] // 2)] * slabs[l]) if split else [(l, ks[l])] * slabs[l]
    per_prod = 3 * 32 if split else 2 * 32      # MFMA issue cycles per k-step product (split: 3 MFMAs; plain: 2 row tiles)
    starts = [buf[8 + u] for u in range(len(units))] + [buf[2]]
    by_layer = {}
    for u, (l, n) in enumerate(units):
        c, f = starts[u + 1] - starts[u], n * per_prod
        a = by_layer.setdefault(l, [0, 0, 0]); a[0] += c; a[1] += f; a[2] += 1
    print("   prologue: %d cycles = DMA requests + head block copy %d | rows' inputs + encoding %d | wait for unit 0 + barrier %d | "
          "first fragments %d" % (starts[0] - buf[0], buf[4] - buf[0], buf[5] - buf[4], buf[6] - buf[5], starts[0] - buf[6]))
    for l, (c, f, n) in sorted(by_layer.items()):
        print(f"   layer {l}: {n:2d} units, {c:6d} cycles, MFMA floor {f:6d} = {100.0 * f / c:5.1f} %   (+{(c - f) / n:6.0f} cycles per unit)")

"""NeRF shapes outside the compiled trunk, layer by layer on the HIP path.

The reference's `NeRF` (run_nerf_helpers.py:76-128) builds any netdepth / netwidth / skip list / encoding width
(flags run_plnerf.py:784-825).  The fused kernels serve the trunk every shipped configuration uses and what maps onto it
exactly (nerf.NeRF.param_list); a network that does not -- netwidth > 256, netdepth > 8, several live skips, multires > 10,
multires_views > 4 -- runs here: every nn.Linear as ONE exact-fp32 MFMA product (plnerf_gemm_f32: bias and ReLU in its
epilogue), its backward as two more (the ReLU's derivative gated into their first operand, the bias gradient as an extra
column of the weight gradient), the concatenations as torch.cat (autograd splits their gradients), the positional encoding
as plnerf_embed_rows.  Unfused -- every layer's activations make a round trip through HBM -- and fp32 whatever `precision`
says: a correct native route for shapes no shipped configuration uses, several times slower per row than the fused
kernels.  No CPU fallback here either."""
import torch

from . import _lib as L


def _gemm(a, a_rs, a_cs, b, b_rs, b_cs, M, N, K, bias=None, gate=None, relu=False, ones_col=False, out=None):
    c = torch.empty(M, N, device=a.device, dtype=torch.float32) if out is None else out
    if M == 0 or N == 0:
        return c
    # a product with few 64 x 64 output tiles and a long k (a weight gradient: k = the rows): deal k out so that ~1000
    # workgroups share it (64 of them walking 262,144 rows each measured 430 ms per backward at 8 x 512)
    tiles = -(-M // 64) * -(-N // 64)
    splits = max(1, min(1024 // tiles, K // 512)) if tiles < 256 else 1
    partials = torch.empty(splits * M * N, device=a.device, dtype=torch.float32) if splits > 1 else None
    L.check(L.lib().plnerf_gemm_f32(L.dptr(a, "a"), a_rs, a_cs, L.dptr(b, "b"), b_rs, b_cs, L.dptr(bias, "bias"), L.dptr(gate, "gate"),
                                    M, N, K, int(relu), 0, int(ones_col), L.dptr(c, "c"), c.stride(0), splits,
                                    L.dptr(partials, "partials"), L.stream()), "plnerf_gemm_f32")
    return c


class LinearFn(torch.autograd.Function):
    """y = relu?(x W^T + b) as one plnerf_gemm_f32; backward: dx = (g * [y > 0]) W, [dW | db] = (g * [y > 0])^T [x | 1]."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = x.detach().to(torch.float32).contiguous()
        w = weight.detach().to(torch.float32).contiguous()
        M, K = x.shape
        N = w.shape[0]
        if w.shape[1] != K:      # (as F.linear: e.g. a skip after the LAST trunk layer, whose concatenation no head layer takes)
            raise RuntimeError(f"plnerf_amd: a layer with {w.shape[1]} input features got rows of {K}")
        # B(k, j) = W[j, k]: row stride 1, column stride K
        y = _gemm(x, K, 1, w, 1, K, M, N, K, bias=bias.detach().to(torch.float32).contiguous(), relu=relu)
        ctx.relu = bool(relu)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        M, K = x.shape
        N = w.shape[0]
        gx = gwb = None
        if ctx.needs_input_grad[0]:
            # dx[M, K] = gate(g)[M, N] . W[N, K]
            gx = _gemm(g, N, 1, w, K, 1, M, K, N, gate=y)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # [dW | db][N, K + 1] = gate(g)^T[N, M] . [x | 1][M, K + 1]: A(i = n, k = m) = g[m, n]
            gwb = _gemm(g, 1, N, x, K, 1, N, K + 1, M, gate=y, ones_col=True)
        gw = gwb[:, :K].contiguous() if gwb is not None and ctx.needs_input_grad[1] else None
        gb = gwb[:, K].contiguous() if gwb is not None and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


def linear(x, layer, relu):
    if not x.is_cuda:
        raise RuntimeError(f"plnerf_amd: the generic NeRF route got a tensor on {x.device}; the HIP path needs a GPU tensor "
                           "(there is no CPU fallback)")
    return LinearFn.apply(x, layer.weight, layer.bias, relu)


def _softplus_density(cols, at):
    """F.softplus(beta=10) on column `at` (the depth-supervised variant's density channel, model/run_nerf_helpers.py:200)."""
    return torch.cat([cols[:, :at], torch.nn.functional.softplus(cols[:, at:at + 1], beta=10), cols[:, at + 1:]], 1)


def forward(net, rows):
    """The module's layers applied to embedded rows [N, input_ch + view_ch], whatever its shape: what NeRF.forward computes
    (run_nerf_helpers.py:105-128; with the camera columns and the softplus density of the depth-supervised variant,
    depth_supervised_exps/model/run_nerf_helpers.py:164-205), every nn.Linear through LinearFn."""
    xyz = rows[:, :net.input_ch]
    act = xyz
    for depth, fc in enumerate(net.pts_linears):
        act = linear(act, fc, relu=True)
        if depth in net.skips:      # the encoding re-enters BEHIND this layer, its channels first (:111-112)
            act = torch.cat((xyz, act), 1)
    if not net.use_viewdirs:
        out = linear(act, net.output_linear, relu=False)
        return _softplus_density(out, 3) if net.density_activation == "softplus" else out
    sigma = linear(act, net.alpha_linear, relu=False)
    side = rows[:, net.input_ch:net.input_ch + net.view_ch]      # direction encoding (+ camera code)
    head = torch.cat((linear(act, net.feature_linear, relu=False), side), 1)
    for fc in net.views_linears:
        head = linear(head, fc, relu=True)
    out = torch.cat((linear(head, net.rgb_linear, relu=False), sigma), 1)
    return _softplus_density(out, 3) if net.density_activation == "softplus" else out

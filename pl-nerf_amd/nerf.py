"""Embedder / get_embedder / NeRF with the reference's interface
(run_nerf_helpers.py:24-72, 76-128), backed by the fused HIP MLP.

`NeRF` keeps the reference's module tree, so `state_dict()` keys/shapes and
`.parameters()` order are identical and reference checkpoints load unchanged
(run_plnerf.py:454-471, 1324-1332).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import functional as Fn
from .functional import MlpFn

# The architecture the HIP kernels are specialised for: the reference's trunk (run_plnerf.py:784-825),
# used by every config under configs/.  The two input widths are run-time: 63 / 27 at the default
# flags (the in-kernel positional encoding exists for exactly these), anything up to 64 / 32 with a
# caller-side encoding -- e.g. 57 / 3 for the depth-supervised variant.
SUPPORTED = dict(D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True)
MAX_INPUT_CH, MAX_VIEW_CH = 64, 32


class Embedder:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]
    (run_nerf_helpers.py:24-54).  Calling the object evaluates the encoding with torch
    ops (generic path, any device); the fused MLP kernel recognises Embedder instances and
    computes the encoding in its own prologue instead, so on the hot path this __call__
    never runs."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.input_dims = kwargs["input_dims"]
        self.include_input = kwargs["include_input"]
        self.num_freqs = kwargs["num_freqs"]
        self.max_freq_log2 = kwargs["max_freq_log2"]
        self.log_sampling = kwargs["log_sampling"]
        self.periodic_fns = kwargs["periodic_fns"]
        # depth-supervised variant: fn(x * pi * freq) (depth_supervised_exps/model/run_nerf_helpers.py:123)
        self.input_scale = kwargs.get("input_scale", None)
        if self.log_sampling:
            self.freq_bands = [float(2.0 ** f) for f in
                               torch.linspace(0.0, self.max_freq_log2, steps=self.num_freqs).tolist()]
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** self.max_freq_log2,
                                             steps=self.num_freqs).tolist()
        self.out_dim = self.input_dims * ((1 if self.include_input else 0) +
                                          self.num_freqs * len(self.periodic_fns))

    def is_standard(self, n_freqs):
        return (self.input_scale is None and
                self.input_dims == 3 and self.include_input and self.log_sampling and
                self.num_freqs == n_freqs and self.max_freq_log2 == n_freqs - 1 and
                list(self.periodic_fns) == [torch.sin, torch.cos])

    def embed(self, inputs):
        blocks = [inputs] if self.include_input else []
        for f in self.freq_bands:
            for fn in self.periodic_fns:
                # same association as the reference: (x * pi) * freq
                blocks.append(fn(inputs * f) if self.input_scale is None else fn(inputs * self.input_scale * f))
        return torch.cat(blocks, -1)

    __call__ = embed


def get_embedder(multires, i=0):
    """(embed_fn, out_dim) -- run_nerf_helpers.py:57-72.  i == -1 selects the identity."""
    if i == -1:
        return nn.Identity(), 3
    emb = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                   log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return emb, emb.out_dim


class NeRF(nn.Module):
    """The reference MLP (run_nerf_helpers.py:76-128).  forward(x) takes the already
    embedded input [N, input_ch + input_ch_views] like the reference and runs the fused
    HIP kernel; `precision` selects the contraction arithmetic ("fp32" = exact fp32 MFMA)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 precision="fp32", input_ch_cam=0, density_activation=None, dense_layer_init=False):
        """input_ch_cam, density_activation="softplus" (beta 10 on the density channel) and
        dense_layer_init (xavier-uniform weights with the layer's gain, zero biases) are the
        depth-supervised variant's NeRF (depth_supervised_exps/model/run_nerf_helpers.py:90-98,
        143-205); the defaults are run_nerf_helpers.py:76-128."""
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views, self.input_ch_cam = input_ch, input_ch_views, input_ch_cam
        self.view_ch = input_ch_views + input_ch_cam
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.precision = precision
        if density_activation not in (None, "softplus"):
            raise ValueError("density_activation must be None or 'softplus'")
        self.density_activation = density_activation
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] +
            [nn.Linear(W + input_ch, W) if i in skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.view_ch + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        if dense_layer_init:
            relu_layers = list(self.pts_linears) + list(self.views_linears)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    gain = nn.init.calculate_gain("relu" if any(m is r for r in relu_layers) else "linear")
                    nn.init.xavier_uniform_(m.weight, gain=gain)
                    nn.init.zeros_(m.bias)
        self._packed = None

    # -- HIP plumbing ---------------------------------------------------------------
    def _skip_layout(self):
        """How the reference module's skip list lands on the compiled trunk (five layers, the concatenation, three
        layers): ("at", k) -- ONE live skip, after layer k <= 4, with one to three layers behind it to consume it
        (netdepth k + 2 .. k + 4; the shipped configurations are ("at", 4) with netdepth 8): layers 0..k sit in the
        compiled slots 0..k, identities fill up to slot 4, layer k + 1 is the compiled skip layer; "none" -- no entry of
        `skips` takes effect (run_nerf_helpers.py:88-89, 109-112: entry k concatenates after layer k and widens layer
        k + 1, so k >= D does nothing); None -- anything else (two live skips; more than five layers before or three
        behind the concatenation; a skip after the LAST layer, which the reference's own head layers cannot consume
        either)."""
        live = sorted(set(k for k in self.skips if 0 <= k < self.D))
        if len(live) == 1 and live[0] <= SUPPORTED["skips"][0] and 1 <= self.D - live[0] - 1 <= SUPPORTED["D"] - 5:
            return ("at", live[0])
        if not live and 1 <= self.D <= SUPPORTED["D"]:
            return "none"
        return None

    def _slots(self):
        """slot_layer[i]: the module's trunk layer that sits in the compiled network's layer i, or None (an identity)."""
        layout = self._skip_layout()
        if layout == "none":
            return [i if i < self.D else None for i in range(SUPPORTED["D"])]
        k = layout[1]
        behind = list(range(k + 1, self.D))
        return list(range(k + 1)) + [None] * (SUPPORTED["skips"][0] - k) + behind + [None] * (SUPPORTED["D"] - 5 - len(behind))

    def is_supported(self):
        # (narrower and shallower trunks, and trunks without a live skip, are padded into the compiled one: param_list)
        trunk = (self._skip_layout() is not None and 8 <= self.W <= SUPPORTED["W"] and self.W % 2 == 0 and
                 1 <= self.input_ch <= MAX_INPUT_CH)
        if not self.use_viewdirs:      # (see param_list; the reference ignores the view columns of x, if any)
            return trunk
        return trunk and 1 <= self.view_ch <= MAX_VIEW_CH

    def is_native(self):
        """The kernels' 24 parameter tensors ARE this module's nn.Parameters (the reference's own configuration: 8 x 256,
        skip after layer 4, view directions) -- no padding / identity layers between them that autograd would have to
        carry gradients through (param_list).  What functional.mlp_backward_multi needs to assign `.grad` directly."""
        return (self.use_viewdirs and self.D == SUPPORTED["D"] and self.W == SUPPORTED["W"] and
                self._skip_layout() == ("at", SUPPORTED["skips"][0]) and self.is_supported())

    def has_fused_encoding(self):
        """True when the kernel's own positional encoding can be this network's: 3 + 6 L position channels (L <= 10) and
        3 + 6 M direction channels (M <= 4) -- a prefix of the compiled 63 | 27 (the unused bands meet zero-padded
        weights) -- and no camera code.  The encoder's input scale (1, or pi for the depth variant) is an argument of
        the call (query(..., input_scale=...))."""
        ok_x = self.input_ch >= 3 and (self.input_ch - 3) % 6 == 0 and self.input_ch <= SUPPORTED["input_ch"]
        if not self.use_viewdirs:
            return ok_x
        ok_d = self.input_ch_views >= 3 and (self.input_ch_views - 3) % 6 == 0 and \
            self.input_ch_views <= SUPPORTED["input_ch_views"]
        return ok_x and ok_d and self.input_ch_cam == 0

    def _require_supported(self):
        if not self.is_supported():
            raise NotImplementedError(
                "plnerf_amd's HIP MLP is compiled for the reference's trunk "
                f"(D={SUPPORTED['D']}, W={SUPPORTED['W']}, skips={SUPPORTED['skips']}) and runs what can be expressed exactly "
                f"in it: one live skip after layer k <= {SUPPORTED['skips'][0]} with netdepth k + 2 .. k + 4, netdepth "
                f"1..{SUPPORTED['D']} without a live skip, even netwidth 8..{SUPPORTED['W']}, input_ch <= "
                f"{MAX_INPUT_CH}, input_ch_views + input_ch_cam <= {MAX_VIEW_CH}, with or without view directions; got "
                f"D={self.D}, W={self.W}, "
                f"input_ch={self.input_ch}, input_ch_views={self.input_ch_views}, input_ch_cam={self.input_ch_cam}, "
                f"skips={self.skips}, use_viewdirs={self.use_viewdirs}.  There is no generic/CPU fallback.")

    def _outputs(self, raw):
        """The kernels' four channels as the reference module returns them: with output_linear the reference has
        output_ch columns (5 when N_importance > 0, run_plnerf.py:424); raw2outputs reads the first four, the rest is
        returned as zeros.  (The density activation, if any, happened in the kernel: density_beta.)"""
        if not self.use_viewdirs and self.output_linear.out_features > 4:
            raw = torch.cat([raw, raw.new_zeros(*raw.shape[:-1], self.output_linear.out_features - 4)], -1)
        return raw

    @property
    def density_beta(self):
        """plnerf_mlp_fwd / _bwd's `density_beta`: F.softplus(alpha, beta=10) on the density channel
        (depth_supervised_exps/model/run_nerf_helpers.py:200) inside the kernels, 0 = none."""
        return 10.0 if self.density_activation == "softplus" else 0.0

    def param_list(self):
        """The 24 parameter tensors of the network the kernels are compiled for (D = 8, W = 256, skip after layer 4,
        view-dependent head), in state_dict order: the C ABI's `params[24]`.  The reference's own configuration IS
        that network and passes its parameters through.  Other shapes of the reference class are expressed EXACTLY in
        it with differentiable torch ops, so autograd carries the kernels' 24 gradients back to the real parameters:

          * netwidth < 256: weights and biases zero-padded (the extra units compute relu(0) = 0 and feed nothing);
          * netdepth 6 or 7: the missing trunk layers as identities (their inputs are post-ReLU, so relu(I h) = h);
          * one live skip after another layer k <= 4 with one to three layers behind it (skips=[2] with netdepth 4..6,
            say): layers 0..k in the compiled slots 0..k, identities up to slot 4, layer k + 1 as the compiled skip layer
            (_skip_layout, _slots);
          * no live skip (skips=[], or every entry >= netdepth, e.g. netdepth 4 with the default skips=[4]), netdepth
            1..8: the compiled skip layer's encoding columns are zero -- [0 | W_5], or [0 | I] when layer 5 itself is
            one of the identities;
          * use_viewdirs=False (run_nerf_helpers.py:102-103, 125-126: `output_linear` on the trunk): feature rows
            0..2 = W_out[rgb], rows 3..5 = -W_out[rgb]; the view layer copies those six features (identity weights, zero
            direction columns); rgb = relu(F) - relu(-F) + b = F + b; sigma = the alpha row = W_out[3]."""
        layout = self._skip_layout()
        if self.use_viewdirs and self.D == SUPPORTED["D"] and self.W == SUPPORTED["W"] and layout == ("at", SUPPORTED["skips"][0]):
            # (the Parameter objects never change -- .to(), load_state_dict and optim.FlatAdam all work on their .data --
            # so the walk over the module tree, 24 generators deep, is done once: it was 100 us of every training step)
            plist = self.__dict__.get("_plist")
            if plist is None:
                plist = self.__dict__["_plist"] = tuple(self.parameters())
            return list(plist)
        import torch.nn.functional as F
        KW, W, D, cin = SUPPORTED["W"], self.W, self.D, self.input_ch
        ref = self.pts_linears[0].weight
        z = lambda *shape: ref.new_zeros(*shape)
        pad_rows = lambda t, n: F.pad(t, (0, 0, 0, n - t.shape[0])) if t.dim() == 2 else F.pad(t, (0, n - t.shape[0]))
        pad_cols = lambda t, n: F.pad(t, (0, n - t.shape[1]))
        out = []
        skip_slot = SUPPORTED["skips"][0] + 1
        for i, j in enumerate(self._slots()):
            if j is not None:
                w, b = self.pts_linears[j].weight, self.pts_linears[j].bias
                if i == 0:
                    w = pad_rows(w, KW)
                elif i == skip_slot and layout != "none":      # the layer behind the live skip: [encoding | hidden] columns
                    w = pad_rows(torch.cat([w[:, :cin], pad_cols(w[:, cin:], KW)], 1), KW)
                elif i == skip_slot:      # no live skip: the compiled layer's encoding columns are zero
                    w = pad_rows(torch.cat([z(w.shape[0], cin), pad_cols(w, KW)], 1), KW)
                else:
                    w = pad_rows(pad_cols(w, KW), KW)
                out += [w, pad_rows(b, KW)]
            else:
                eye = torch.eye(KW, device=ref.device, dtype=ref.dtype)
                out += [torch.cat([z(KW, cin), eye], 1) if i == skip_slot else eye, z(KW)]
        HV, vch = KW // 2, self.hip_view_ch
        if self.use_viewdirs:
            vw = self.views_linears[0].weight                                  # [W/2, W + view_ch]: [feature | direction]
            out += [pad_rows(torch.cat([pad_cols(vw[:, :W], KW), vw[:, W:]], 1), HV), pad_rows(self.views_linears[0].bias, HV),
                    pad_rows(pad_cols(self.feature_linear.weight, KW), KW), pad_rows(self.feature_linear.bias, KW),
                    pad_cols(self.alpha_linear.weight, KW), self.alpha_linear.bias,
                    pad_cols(self.rgb_linear.weight, HV), self.rgb_linear.bias]
            return out
        Wo, bo = self.output_linear.weight, self.output_linear.bias
        if Wo.shape[0] < 4:
            raise NotImplementedError("plnerf_amd: output_linear needs at least 4 outputs (rgb, sigma)")
        views_w, rgb_w = z(HV, KW + vch), z(3, HV)
        for c in range(6):
            views_w[c, c] = 1.0
        for c in range(3):
            rgb_w[c, c], rgb_w[c, 3 + c] = 1.0, -1.0
        Wp = pad_cols(Wo, KW)
        out += [views_w, z(HV), torch.cat([Wp[0:3], -Wp[0:3], z(KW - 6, KW)], 0), z(KW), Wp[3:4], bo[3:4], rgb_w, bo[0:3]]
        return out

    @property
    def hip_view_ch(self):
        """Direction channels as the kernels see them: without view directions, the in-kernel encoding's 27 (all of
        them multiplied by zero weights)."""
        return self.view_ch if self.use_viewdirs else SUPPORTED["input_ch_views"]

    def packed_weights(self):
        """Weights re-laid-out in MFMA fragment order (plnerf_mlp_pack_weights).  Re-packed on EVERY
        call: one ~10 us pass over 2.4 MB, negligible next to any forward -- and the only safe policy,
        because in-place parameter updates are not reliably observable (fused/foreach optimizers do
        not bump `Tensor._version`; a version-keyed cache silently served stale weights once)."""
        self._require_supported()
        params = self.param_list()
        prec = L.PRECISION[self.precision]
        nbytes = L.lib().plnerf_mlp_packed_bytes(prec)
        if nbytes == 0:
            raise NotImplementedError(f"precision mode {self.precision!r} is not built")
        dev = params[0].device
        flat = [p.detach() for p in params]
        self._ensure_packed(dev, nbytes)
        L.check(L.lib().plnerf_mlp_pack_weights(L.ptr_table(flat), prec, int(self.input_ch), int(self.hip_view_ch),
                                                L.dptr(self._packed), L.stream()), "plnerf_mlp_pack_weights")
        return self._packed

    def _ensure_packed(self, dev, nbytes):
        if self._packed is None or self._packed.device != dev or self._packed.numel() * 4 != nbytes:
            self._packed = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
            self.status_word().zero_()         # the library only ever ORs into it

    def status_word(self):
        """The packed buffer's range status word as a 1-element int32 tensor (a view: the kernels OR into it).  Half-
        element modes only set it (include/plnerf_hip.h: PLNERF_RANGE_*); optim.FlatAdam takes it as its guard."""
        self._require_supported()
        prec = L.PRECISION[self.precision]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("plnerf_amd: the range status word lives on the GPU (there is no CPU fallback)")
        self._ensure_packed(dev, L.lib().plnerf_mlp_packed_bytes(prec))
        off = L.lib().plnerf_mlp_status_offset(prec) // 4
        return self._packed.view(torch.int32)[off:off + 1]

    def range_status(self, reset=False):
        """Bits set since the word was last cleared (synchronises): _lib.RANGE_ACTIVATION -- a forward met an activation
        beyond the IEEE-half range (the `f16x3` / `f16` modes clamp there: their result is wrong; `bf16x3` carries
        fp32's exponent range in the forward); _lib.RANGE_WEIGHT -- a weight beyond it (or not finite) was packed;
        _lib.RANGE_SAVED -- a training forward of a bf16-element mode clamped an activation on its way into the IEEE-half
        saved planes (forward right, gradients wrong)."""
        w = self.status_word()
        bits = int(w.item())
        if reset and bits:
            w.zero_()
        return bits

    def check_range(self):
        """Raise if a kernel of this network left the half range since the last check (and clear the word)."""
        bits = self.range_status(reset=True)
        if bits:
            what = [n for b, n in ((L.RANGE_ACTIVATION, "an activation"), (L.RANGE_WEIGHT, "a weight"),
                                   (L.RANGE_SAVED, "an activation saved for the backward")) if bits & b]
            if bits & (L.RANGE_ACTIVATION | L.RANGE_WEIGHT):
                advice = ("results since the last check were clamped and the optimizer steps were withheld.  Use "
                          "precision='bf16x3' (fp32 exponent range in the forward, same speed) for inference, 'fp32' for "
                          "training this network")
            else:
                advice = ("the forward results were right (bf16 elements carry fp32's exponent range) but the IEEE-half "
                          "planes the backward reads were clamped: the gradients of those steps were wrong and the "
                          "optimizer steps were withheld.  Train this network in precision='fp32'")
            raise FloatingPointError(
                f"plnerf_amd: {' and '.join(what)} exceeded the IEEE-half range (65,504) in precision={self.precision!r}: "
                + advice + ".")

    # -- reference interface --------------------------------------------------------
    def forward(self, x, cam=None):
        """x [..., input_ch + input_ch_views + input_ch_cam] embedded rows (the reference module's signature).  `cam`
        (extension): the camera code the caller has repeated into the last input_ch_cam columns of every row, when it
        is to receive a gradient (functional.MlpFn); x itself carries none."""
        lead = x.shape[:-1]
        flat = x.reshape(-1, x.shape[-1])
        if flat.shape[-1] != self.input_ch + self.view_ch:
            raise ValueError(f"NeRF.forward expects {self.input_ch + self.view_ch} embedded channels, got {flat.shape[-1]}")
        if not self.is_supported():      # a shape outside the compiled trunk: layer by layer (generic.py), exact fp32
            from . import generic
            if cam is not None and cam.requires_grad:
                raise NotImplementedError("plnerf_amd: a trainable camera code needs the fused route (a supported trunk)")
            return generic.forward(self, flat).reshape(*lead, -1)
        if not self.use_viewdirs:      # the kernels' direction channels: zeros (their weights are zero as well)
            flat = torch.cat([flat[:, :self.input_ch], flat.new_zeros(flat.shape[0], self.hip_view_ch)], -1)
        out = MlpFn.apply(None, None, flat, cam, 1, self, torch.is_grad_enabled(), *self.param_list())
        if Fn.MLP_TAPE is not None:
            Fn.MLP_TAPE.outs.append(out)
        return self._outputs(out.reshape(*lead, 4))

    def query(self, pts, viewdirs, input_scale=1.0):
        """Fused entry: pts [R,S,3], viewdirs [R,3] -> raw [R,S,4]; the encoding gamma(x) = [x, sin / cos(x s 2^k)] with
        s = input_scale happens in the kernel prologue (what run_network does on the hot path)."""
        if not self.is_supported():
            raise NotImplementedError("NeRF.query is the fused entry (in-kernel encoding) of the compiled trunk; a shape outside "
                                      "it embeds on the caller side and uses forward() -- run_network does")
        if not self.has_fused_encoding():
            raise NotImplementedError("the in-kernel encoding covers 3 + 6 L | 3 + 6 M channels (L <= 10, M <= 4) without "
                                      "a camera code; embed on the caller side and use forward()")
        R, S = pts.shape[0], pts.shape[1]
        if viewdirs is None:
            if self.use_viewdirs:
                raise ValueError("this network takes view directions")
            viewdirs = pts.new_zeros(R, 3)
        self._query_scale = float(input_scale)
        try:
            out = MlpFn.apply(pts.reshape(-1, 3), viewdirs, None, None, S, self, torch.is_grad_enabled(), *self.param_list())
        finally:
            self._query_scale = 1.0
        if Fn.MLP_TAPE is not None:
            Fn.MLP_TAPE.outs.append(out)
        return self._outputs(out.reshape(R, S, 4))

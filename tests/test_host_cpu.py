"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares; host-side logic; the data-parallel path on world_size-2 gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "pl-nerf_amd", "libplnerf_hip.so")):
        ge.build()
    import plnerf_amd
    return plnerf_amd


def test_library_exports_every_declared_symbol(built):
    import ctypes
    from plnerf_amd import _lib
    header = open(os.path.join(ROOT, "include", "plnerf_hip.h")).read()
    declared = set(re.findall(r"\b(plnerf_[a-z0-9_]+)\s*\(", header))
    declared.discard("plnerf_stream_t")
    assert declared, "no declarations parsed"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in plnerf_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    declared = int(re.search(r"#define\s+PLNERF_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "plnerf_hip.h")).read()).group(1))
    assert built.library_version() == declared >= 230
    from plnerf_amd import _lib as _L
    assert _L.lib().plnerf_build_flags() == 0      # no ablation / trace switches in the product library
    assert _lib.lib().plnerf_error_string(-3).decode().startswith("size outside")


def test_header_is_plain_c_and_links(built, tmp_path):
    """include/plnerf_hip.h compiled as C99 by gcc, every entry point referenced with its declared prototype, linked
    against the built library, and the GPU-free calls executed (tests/abi_check.c): the boundary a cgo / JNI / ctypes
    binding would bind."""
    import re
    from plnerf_amd import _lib
    header = open(os.path.join(ROOT, "include", "plnerf_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(plnerf_\w+)\s*\(", header, flags=re.M))
    listed = set(re.findall(r"\)\s*=\s*(plnerf_\w+);", open(os.path.join(ROOT, "tests", "abi_check.c")).read()))
    assert declared == listed == set(_lib.SIGNATURES), (declared ^ listed, declared ^ set(_lib.SIGNATURES))
    exe = str(tmp_path / "abi_check")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe, "-L", libdir, "-lplnerf_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert f"{len(declared)} entry points" in out.stdout


def _header_prototypes():
    """{name: (return type, [parameter types])} parsed from include/plnerf_hip.h (comments stripped)."""
    import re
    header = open(os.path.join(ROOT, "include", "plnerf_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"^(int|size_t|const char\*)\s+(plnerf_\w+)\s*\(([^;]*?)\)\s*;", code, flags=re.M | re.S):
        params = []
        for a in (x.strip() for x in " ".join(args.split()).split(",")):
            if a == "void":
                continue
            params.append(re.match(r"^(.*?)\b\w+$", a).group(1).strip())
        protos[name] = (ret, params)
    return protos


def test_ctypes_signatures_match_the_header(built):
    """_lib.SIGNATURES restates the header's argument lists by hand; a swapped c_int / c_float in a 30-argument call would
    be silent undefined behaviour.  Parse every prototype of include/plnerf_hip.h into its ABI class per argument
    (pointer, 32-bit int, 32-bit unsigned, 64-bit int, 64-bit unsigned / size_t, float) and compare with the ctypes
    declaration's, argument by argument, return type included."""
    import ctypes
    from plnerf_amd import _lib

    def c_class(t):
        t = t.replace("const ", "").strip()
        if t.endswith("*") or t == "plnerf_stream_t":
            return "ptr"
        return {"int": "i32", "float": "f32", "uint64_t": "u64", "uint32_t": "u32", "int64_t": "i64", "size_t": "u64",
                "unsigned": "u32", "double": "f64"}[t]      # (size_t and uint64_t are one ctypes object on LP64)

    def ct_class(t):
        if t is ctypes.c_char_p or t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_float: "f32", ctypes.c_uint64: "u64", ctypes.c_uint32: "u32",
                ctypes.c_int64: "i64", ctypes.c_size_t: "u64", ctypes.c_double: "f64"}[t]
    protos = _header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = _lib.SIGNATURES[name]
        want_ret = "ptr" if ret.endswith("*") else c_class(ret)
        assert ct_class(res) == want_ret, (name, "return", ret, res)
        assert len(args) == len(params), (name, len(args), len(params))
        for k, (c, t) in enumerate(zip(params, args)):
            assert ct_class(t) == c_class(c), (name, k, c, t)
    # host-side pointer arguments are declared as typed ctypes pointers (not void*): the binding passes arrays there
    assert _lib.SIGNATURES["plnerf_select_rays"][1][6] is not ctypes.c_void_p
    assert _lib.SIGNATURES["plnerf_embed_rows"][1][9] is not ctypes.c_void_p


def test_pe_sincos_reduction_on_host(tmp_path):
    """pl-nerf_amd/csrc/pe_sincos.h (the encoding's shared argument reduction) is plain C++: compile it for the
    host and compare with double-precision sin / cos over scene-scale, large and near-k*pi/2 arguments (the
    checker exits non-zero above 1.2e-7 absolute)."""
    exe = str(tmp_path / "pe_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "pl-nerf_amd", "csrc"),
                    os.path.join(ROOT, "tools", "probes", "pe_sincos_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_buffer_size_queries(built):
    from plnerf_amd import _lib
    L = _lib.lib()
    assert L.plnerf_mlp_packed_bytes(0) > 595844 * 4          # padded fwd + bwd layouts
    assert L.plnerf_mlp_saved_bytes(1000, 0) == 1000 * 2596 * 4
    assert L.plnerf_mlp_saved_bytes(1000, 3) == 1024 * (2272 * 2 + 272)      # 16-bit modes: half planes + relu masks, rows padded to whole 256-row workgroup tiles
    assert L.plnerf_mlp_bwd_workspace_bytes(1000, 3) < L.plnerf_mlp_bwd_workspace_bytes(1000, 0)
    assert L.plnerf_mlp_bwd_workspace_bytes(1000, 0) > 1000 * 2432 * 4
    # half dz planes: rows padded to the dgrad kernel's 192-row tiles (tiled planes, mlp_layout.h; 64 rows until round 3)
    # (+ 16 B per row: the upstream gradient after the density activation's derivative, plnerf_mlp_bwd's g_eff)
    assert L.plnerf_mlp_bwd_workspace_bytes(961, 3) - 961 * 16 == L.plnerf_mlp_bwd_workspace_bytes(1152, 3) - 1152 * 16      # 6 x 192
    assert L.plnerf_mlp_bwd_workspace_bytes(1153, 3) - L.plnerf_mlp_bwd_workspace_bytes(1152, 3) == 192 * 2176 * 2 + 16
    assert L.plnerf_mlp_saved_bytes(1025, 3) - L.plnerf_mlp_saved_bytes(1024, 3) == 256 * (2272 * 2 + 272)
    # the layout tag a caller hands back to plnerf_mlp_bwd: the split modes write tiled planes (with or without a
    # caller-embedded input); exact fp32 and the plain 16-bit modes row-major
    # the saved layout is a pure function of (precision, embedded input, forward-kernel argument): no environment
    AUTO, RR, PP = 0, 1, 2
    for k in (AUTO, RR, PP):
        assert L.plnerf_mlp_saved_layout(0, 0, k) == 0 and L.plnerf_mlp_saved_layout(0, 1, k) == 0       # fp32: rows
    assert L.plnerf_mlp_saved_layout(3, 0, AUTO) == 1 and L.plnerf_mlp_saved_layout(3, 1, AUTO) == 1     # f16x3: tiled
    assert L.plnerf_mlp_saved_layout(1, 0, AUTO) == 1 and L.plnerf_mlp_saved_layout(1, 1, AUTO) == 1     # bf16x3
    assert L.plnerf_mlp_saved_layout(4, 0, AUTO) == 0 and L.plnerf_mlp_saved_layout(4, 1, AUTO) == 0     # f16: ping-pong
    assert L.plnerf_mlp_saved_layout(2, 0, AUTO) == 0 and L.plnerf_mlp_saved_layout(2, 1, AUTO) == 0
    assert L.plnerf_mlp_saved_layout(3, 0, PP) == 0 and L.plnerf_mlp_saved_layout(1, 0, PP) == 0
    assert L.plnerf_mlp_saved_layout(4, 0, RR) == 1 and L.plnerf_mlp_saved_layout(4, 1, RR) == 0         # no plain embedded rr
    assert L.plnerf_mlp_saved_layout(3, 0, 7) < 0 and L.plnerf_mlp_saved_layout(9, 0, AUTO) < 0
    assert L.plnerf_mlp_packed_bytes(7) == 0


def test_network_without_view_directions_maps_onto_the_kernels_head(built):
    """use_viewdirs=False: NeRF.param_list() expresses output_linear in the 24 tensors of the view-dependent head the
    kernels implement (shapes of a use_viewdirs network with 27 direction channels), and that head -- evaluated here in
    plain torch -- reproduces output_linear exactly; the state_dict keeps the reference's keys."""
    import torch
    import plnerf_amd as P
    torch.manual_seed(3)
    net = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    ref = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    assert net.is_supported() and net.has_fused_encoding() and net.hip_view_ch == 27
    assert "output_linear.weight" in net.state_dict() and "alpha_linear.weight" not in net.state_dict()
    mine, theirs = net.param_list(), list(ref.parameters())
    assert [tuple(t.shape) for t in mine] == [tuple(t.shape) for t in theirs]
    F = torch.nn.functional
    h = torch.randn(50, 256)
    views_w, views_b, feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b = mine[16:]
    feat = F.linear(h, feat_w, feat_b)
    hv = F.relu(F.linear(torch.cat([feat, torch.randn(50, 27)], -1), views_w, views_b))
    out = torch.cat([F.linear(hv, rgb_w, rgb_b), F.linear(h, alpha_w, alpha_b)], -1)
    want = F.linear(h, net.output_linear.weight, net.output_linear.bias)[:, :4]
    assert float((out - want).abs().max()) <= 1e-6
    out.sum().backward()
    assert net.output_linear.weight.grad is not None and float(net.output_linear.weight.grad[:4].abs().max()) > 0


@pytest.mark.parametrize("shape", [(8, 128, True, [4]), (6, 256, True, [4]), (7, 64, False, [4]), (6, 32, True, [4]),
                                   (8, 256, True, []), (4, 256, True, [4]), (3, 64, False, []), (1, 128, True, [4]),
                                   (5, 96, True, [7]), (6, 256, True, [2]), (5, 128, True, [1]), (7, 64, True, [3]),
                                   (4, 96, False, [0]), (6, 160, True, [2, 9]), (3, 256, True, [1])])
def test_other_network_shapes_map_exactly_onto_the_compiled_network(built, shape):
    """NeRF.param_list() for netwidth < 256 / netdepth 6, 7 / no view directions / no live skip with netdepth 1..8 / one
    live skip after a layer k <= 4 with one to three layers behind it: the
    24 tensors of the compiled 8 x 256 network, evaluated here in plain fp64 torch, reproduce the real module's
    function exactly (the module's own forward, run_nerf_helpers.py:105-128, restated)."""
    import torch
    import plnerf_amd as P
    D, Wd, use_viewdirs, skips = shape
    F = torch.nn.functional
    torch.manual_seed(5)
    net = P.NeRF(D=D, W=Wd, input_ch=63, input_ch_views=27 if use_viewdirs else 0, output_ch=5, skips=skips,
                 use_viewdirs=use_viewdirs).double()
    assert net.is_supported()
    x, v = torch.randn(40, 63, dtype=torch.float64), torch.randn(40, 27, dtype=torch.float64)
    sd = net.state_dict()
    h = x
    for i in range(D):
        h = F.relu(F.linear(h, sd[f"pts_linears.{i}.weight"], sd[f"pts_linears.{i}.bias"]))
        if i in skips:
            h = torch.cat([x, h], -1)
    assert h.shape[-1] == Wd      # (the shapes here leave the head a plain W-wide input, as the reference's head needs)
    if use_viewdirs:
        sigma = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
        feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
        hv = F.relu(F.linear(torch.cat([feat, v], -1), sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
        want = torch.cat([F.linear(hv, sd["rgb_linear.weight"], sd["rgb_linear.bias"]), sigma], -1)
    else:
        want = F.linear(h, sd["output_linear.weight"], sd["output_linear.bias"])[:, :4]
    p = net.param_list()
    ref = P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    assert [tuple(t.shape) for t in p] == [tuple(t.shape) for t in ref.parameters()]
    h = x
    for i in range(8):
        h = F.relu(F.linear(h, p[2 * i], p[2 * i + 1]))
        if i == 4:
            h = torch.cat([x, h], -1)
    views_w, views_b, feat_w, feat_b, alpha_w, alpha_b, rgb_w, rgb_b = p[16:]
    hv = F.relu(F.linear(torch.cat([F.linear(h, feat_w, feat_b), v], -1), views_w, views_b))
    got = torch.cat([F.linear(hv, rgb_w, rgb_b), F.linear(h, alpha_w, alpha_b)], -1)
    assert float((got - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max()))


def test_unsupported_network_shapes_are_refused(built):
    """What the compiled trunk cannot express is constructible (same state_dict as the reference) and the FUSED entries
    refuse it (NeRF.forward then takes the layer-by-layer route, generic.py): more than three layers behind a skip or more
    than five before it, two live skips, a skip after the last layer (the reference's own head cannot consume that one
    either), more than 8 layers, more than 256 units."""
    import plnerf_amd as P
    for kw in (dict(D=8, skips=[2]), dict(D=5, skips=[4]), dict(D=9, skips=[4]), dict(D=8, W=512, skips=[4]),
               dict(D=8, skips=[4, 6]), dict(D=8, skips=[5]), dict(D=6, skips=[1, 3])):
        net = P.NeRF(input_ch=63, input_ch_views=27, output_ch=5, use_viewdirs=True, **kw)
        assert not net.is_supported(), kw
        with pytest.raises(NotImplementedError):
            net.param_list() if False else net._require_supported()


def _nvs_args(tmp, **over):
    from argparse import Namespace
    a = dict(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64, netdepth=8,
             netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, coarse_lrate=5e-4,
             ft_path=None, ckpt_dir=str(tmp), expname="exp", no_reload=True, perturb=1.0, white_bkgd=True,
             raw_noise_std=0.0, mode="linear", color_mode="midpoint", dataset="blender", no_ndc=False, lindisp=False,
             lrate_decay=250, constant_init=0, chunk=32768, precision="f16x3", N_rand=256)
    a.update(over)
    return Namespace(**a)


# create_nerf flag combinations: (overrides, route).  INTEGRATION.md's table is this list.
CREATE_NERF_SHAPES = [
    (dict(), "fused"),                                                  # the reference's configs: 8 x 256, skip after 4, 63 | 27
    (dict(netwidth=128, netwidth_fine=128), "fused"),                   # narrower (zero-padded)
    (dict(netdepth=6, netdepth_fine=6), "fused"),                       # shallower behind the skip (identity layers)
    (dict(netdepth=4, netdepth_fine=4), "fused"),                       # the default skips=[4] is not live below 6 layers
    (dict(use_viewdirs=False), "fused"),                                # output_linear on the trunk
    (dict(multires=6, multires_views=2), "fused"),                      # fewer frequency bands (a prefix of 63 | 27)
    (dict(N_importance=0), "fused"),                                    # one network, two Adams
    (dict(netdepth=9), "generic"), (dict(netdepth_fine=10), "generic"), # deeper than the compiled trunk
    (dict(netwidth=512), "generic"), (dict(netwidth_fine=384), "generic"),   # wider
    (dict(netwidth=255), "generic"),                                    # odd width (the view layer halves it)
    (dict(multires=11), "generic"),                                     # 69 position channels > 64
    (dict(multires_views=5), "generic"),                                # 33 direction channels > 32
    (dict(netdepth=5, netdepth_fine=5), "refused"),                     # a skip after the LAST layer: the reference's head cannot consume it either
]


@pytest.mark.parametrize("over,route", CREATE_NERF_SHAPES, ids=lambda v: "-".join(f"{k}{x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_create_nerf_routes_every_shape_at_the_boundary(built, tmp_path, over, route):
    """The flags of run_plnerf.py:784-799: what the compiled trunk expresses runs on the fused kernels; what it cannot
    (deeper, wider, more encoding channels) is served layer by layer on exact-fp32 MFMA products (generic.py) and create_nerf
    says so ONCE, at the boundary (run_plnerf.py:417-447), before a reference-style run has loaded its data; what the
    reference's own forward cannot run raises there.  NeRF.__init__ stays permissive (same state_dict as the reference)."""
    import warnings
    (tmp_path / "exp").mkdir()
    args = _nvs_args(tmp_path, **over)
    if route == "refused":
        with pytest.raises(NotImplementedError, match="after the last trunk layer"):
            built.create_nerf(args, device=torch.device("cpu"))
        return
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        kw = built.create_nerf(args, device=torch.device("cpu"))[0]
    nets = [n for n in (kw["network_fn"], kw["network_fine"]) if n is not None]
    told = [w for w in caught if issubclass(w.category, RuntimeWarning) and "layer by layer" in str(w.message)]
    if route == "fused":
        assert all(n.is_supported() for n in nets) and not told
    else:
        assert not all(n.is_supported() for n in nets)
        assert len(told) == sum(not n.is_supported() for n in nets)
        # (no CPU fallback on this route either)
        net = next(n for n in nets if not n.is_supported())
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net(torch.zeros(3, net.input_ch + net.view_ch))


def test_no_cpu_fallback(built):
    net = built.NeRF(input_ch=63, input_ch_views=27, use_viewdirs=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(3, 90))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        built.raw2outputs(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 1), torch.ones(2, 1),
                          torch.ones(2, 3), "linear", "midpoint")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pl-nerf_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, re.M), f"{fn} imports the oracle"
            assert "plnerf_oracle" not in src and "importlib" not in src, f"{fn} reaches for the oracle"


def test_module_interface_matches_reference(built):
    """state_dict keys / shapes / parameter order of the drop-in NeRF (checkpoint compatibility,
    run_plnerf.py:454-471) and the get_embedder contract."""
    from oracle import plnerf_oracle as orc
    net = built.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == \
        [(k, tuple(s)) for k, s in orc.param_shapes()]
    assert [tuple(p.shape) for p in net.parameters()] == [tuple(s) for _, s in orc.param_shapes()]
    net.load_state_dict(orc.closed_form_state_dict(0))
    emb, ch = built.get_embedder(10, 0)
    assert ch == 63 and emb.is_standard(10)
    x = torch.randn(5, 3)
    assert torch.equal(emb(x), orc.positional_encoding(x, 10))
    ident, ch = built.get_embedder(10, -1)
    assert ch == 3 and torch.equal(ident(x), x)


def test_rays_match_golden(built, golden):
    g = golden("g7_rays")
    H, W, f = int(g["H"]), int(g["W"]), float(g["focal"])
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    o, d = built.get_rays(H, W, K, torch.from_numpy(g["c2w"]))
    assert torch.equal(o, torch.from_numpy(g["rays_o"])) and torch.equal(d, torch.from_numpy(g["rays_d"]))
    o2, d2 = built.ndc_rays(H, W, f, 1.0, o, d)
    assert torch.allclose(o2, torch.from_numpy(g["ndc_o"]), atol=1e-6, rtol=1e-6)
    assert torch.allclose(d2, torch.from_numpy(g["ndc_d"]), atol=1e-6, rtol=1e-6)


def test_contiguous_runs_of_gradient_slices(built):
    """optim.contiguous_runs / flat_view_of (host logic of FlatAdam and of the DP bucket): consecutive slices of one
    buffer merge into one flat view; separate allocations, gaps, non-contiguous and non-fp32 tensors do not."""
    from plnerf_amd.optim import contiguous_runs, flat_view_of
    flat = torch.arange(100, dtype=torch.float32)
    a, b, c = flat[0:12].view(3, 4), flat[12:20], flat[20:50].view(5, 6)
    runs = contiguous_runs([a, b, c])
    assert len(runs) == 1 and runs[0][:2] == (0, 3) and torch.equal(runs[0][2], flat[:50])
    runs[0][2].mul_(2.0)                                   # it is a view, not a copy
    assert float(b[0]) == 24.0
    assert flat_view_of([a, b, c]).data_ptr() == flat.data_ptr()
    # a gap (slice 50:60 skipped) and an unrelated allocation split the list
    d, e = flat[60:70], torch.zeros(7)
    runs = contiguous_runs([a, b, c, d, e])
    assert [(r[0], r[1]) for r in runs] == [(0, 3), (3, 4), (4, 5)] and flat_view_of([a, b, c, d]) is None
    # order matters; non-contiguous / non-fp32 members stand alone without a flat view
    assert flat_view_of([b, a]) is None
    t = flat[70:90].view(4, 5).t()
    runs = contiguous_runs([t, torch.zeros(3, dtype=torch.float64)])
    assert [r[2] is None for r in runs] == [True, True]
    assert flat_view_of([]) is None and flat_view_of([a, None]) is None
    # two neighbouring allocations that merely touch must not merge: emulate with views of distinct storages
    x, y = torch.zeros(8), torch.zeros(8)
    assert len(contiguous_runs([x, y])) == 2


def test_shard_rays():
    from plnerf_amd import dp
    spans = [dp.shard_rays(32768, r, 8) for r in range(8)]
    assert spans[0] == (0, 4096) and spans[-1] == (28672, 32768)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
    with pytest.raises(ValueError):
        dp.shard_rays(1000, 0, 3)


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import plnerf_amd
from plnerf_amd import dp
rank, world, _ = dp.init_from_env(backend="gloo")
torch.manual_seed(1234 + rank)               # deliberately different init per rank
nets = [torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)]
dp.broadcast_parameters(nets, src=0)
w0 = [p.detach().clone() for n in nets for p in n.parameters()]
gathered = [None] * world
dist.all_gather_object(gathered, [w.tolist() for w in w0])
assert gathered[0] == gathered[1], "replicas differ after broadcast"
# global batch of 8 rows, split 4/4; the loss is a mean over rows like img2mse
g = torch.Generator().manual_seed(0)
X, Y = torch.randn(8, 7, generator=g), torch.randn(8, 3, generator=g)
lo, hi = dp.shard_rays(8, rank, world)
bucket = dp.GradientBucket(nets)
loss = ((nets[1](torch.relu(nets[0](X[lo:hi]))) - Y[lo:hi]) ** 2).mean()
loss.backward()
bucket.allreduce_mean()
# reference: the full batch on one process
ref = [torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)]
for r, n in zip(ref, nets):
    r.load_state_dict(n.state_dict())
((ref[1](torch.relu(ref[0](X))) - Y) ** 2).mean().backward()
for n, r in zip(nets, ref):
    for p, q in zip(n.parameters(), r.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6), (rank, (p.grad - q.grad).abs().max())
# optimizer state and the loop's step count travel with the weights: rank 0 "restored a checkpoint" (has moments and a
# step count), rank 1 starts empty -- after the broadcast both hold rank 0's
opt = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=1e-3)
if rank == 0:
    for k in range(3):
        opt.step()
dp.broadcast_optimizer_state([opt])
start = dp.broadcast_scalar(1234 if rank == 0 else 0)
assert start == 1234 and isinstance(start, int)
state = [[float(opt.state[p]['step']), opt.state[p]['exp_avg'].tolist(), opt.state[p]['exp_avg_sq'].tolist()]
         for n in nets for p in n.parameters()]
dist.all_gather_object(gathered, state)
assert gathered[0] == gathered[1] and gathered[0][0][0] == 3.0, "optimizer state differs after the broadcast"
print(f"rank {rank} ok")
'''


def test_data_parallel_gradient_allreduce_gloo(tmp_path):
    """world_size 2 on CPU/gloo: after the bucketed all-reduce every rank holds the gradient of
    the full (unsharded) batch, and replicas start bit-identical."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    port = 29600 + (os.getpid() % 300)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out


_DP_NERF_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import plnerf_amd as P
from plnerf_amd import dp
from plnerf_amd.optim import flat_view_of
rank, world, _ = dp.init_from_env(backend="gloo")


class FlatGradFn(torch.autograd.Function):
    """Stands in for functional.MlpFn on the CPU: out[r] = sum_k cos(k x[r]) * sum(p_k); its backward returns the 24
    parameter gradients as consecutive slices of ONE buffer, exactly the layout MlpFn.backward produces."""
    @staticmethod
    def forward(ctx, x, *params):
        ctx.save_for_backward(x)
        ctx.shapes = [p.shape for p in params]
        return sum(torch.cos((k + 1) * x) * p.sum() for k, p in enumerate(params))

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        sizes = [int(torch.Size(s).numel()) for s in ctx.shapes]
        flat = torch.empty(sum(sizes), dtype=torch.float32)
        grads = [t.view(s) for t, s in zip(flat.split(sizes), ctx.shapes)]
        for k, gr in enumerate(grads):
            gr.fill_(float((g * torch.cos((k + 1) * x)).sum()))
        return (None,) + tuple(grads)


torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
nets = [P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True) for _ in range(2)]
dp.broadcast_parameters(nets)
bucket = dp.GradientBucket(nets)
opts = [torch.optim.Adam(n.parameters(), lr=5e-4) for n in nets]
gen = torch.Generator().manual_seed(0)
X, Y = torch.rand(16, generator=gen), torch.rand(16, generator=gen)
lo, hi = dp.shard_rays(16, rank, world)
# single-process reference on the full batch
ref = [P.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True) for _ in range(2)]
for r, n in zip(ref, nets):
    r.load_state_dict(n.state_dict())
ref_opts = [torch.optim.Adam(n.parameters(), lr=5e-4) for n in ref]
for step in range(3):
    for o in opts + ref_opts:
        o.zero_grad()
    out = FlatGradFn.apply(X[lo:hi], *nets[0].parameters()) + 0.5 * FlatGradFn.apply(X[lo:hi], *nets[1].parameters())
    ((out - Y[lo:hi]) ** 2).mean().backward()
    for n in nets:                                    # one flat buffer per network, as on the GPU path
        assert flat_view_of([p.grad for p in n.parameters()]) is not None
    assert bucket.pending() == 2, bucket.pending()    # both collectives were enqueued from the hooks, in place
    ptrs = [next(n.parameters()).grad.data_ptr() for n in nets]
    assert bucket.allreduce_mean() == 2
    assert ptrs == [next(n.parameters()).grad.data_ptr() for n in nets]       # no gather / scatter copies
    out = FlatGradFn.apply(X, *ref[0].parameters()) + 0.5 * FlatGradFn.apply(X, *ref[1].parameters())
    ((out - Y) ** 2).mean().backward()
    for n, r in zip(nets, ref):
        for p, q in zip(n.parameters(), r.parameters()):
            # (the two trajectories round differently: fp32 sums over 65,536-element tensors)
            assert float((p.grad - q.grad).abs().max()) <= 1e-3 * float(q.grad.abs().max()) + 1e-6, (rank, step)
    for o in opts + ref_opts:
        o.step()
    digest = [float(p.detach().double().sum()) for n in nets for p in n.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    assert gathered[0] == gathered[1], f"replicas diverged at step {step}"
# The exchange as train.TrainStep uses it since round 4: the buffer carries the network's range status as a tail element
# (functional.MlpFn.backward's layout: 24 gradients + GRAD_TAIL floats, registered as net._grad_flat), `finish` leaves the
# SUM in .grad and hands back the 1 / world factor for the optimizer's step kernel, `tails` the summed status words --
# one collective per network, no scaling pass, no status collective.
class FlatGradTailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, net, flag, *params):
        ctx.save_for_backward(x)
        ctx.shapes, ctx.net, ctx.flag = [p.shape for p in params], net, flag
        return sum(torch.cos((k + 1) * x) * p.sum() for k, p in enumerate(params))

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        sizes = [int(torch.Size(s).numel()) for s in ctx.shapes]
        full = torch.empty(sum(sizes) + dp.GradientBucket.TAIL, dtype=torch.float32)
        full[sum(sizes):] = 0.0
        full[sum(sizes)] = ctx.flag                       # (what the weight-gradient reduction kernel writes)
        ctx.net.__dict__["_grad_flat"] = full
        grads = [t.view(s) for t, s in zip(full[:sum(sizes)].split(sizes), ctx.shapes)]
        for k, gr in enumerate(grads):
            gr.fill_(float((g * torch.cos((k + 1) * x)).sum()))
        return (None, None, None) + tuple(grads)


for n in nets:
    n.zero_grad(set_to_none=True)
out = FlatGradTailFn.apply(X[lo:hi], nets[0], 1.0 if rank == 1 else 0.0, *nets[0].parameters()) + \
    0.5 * FlatGradTailFn.apply(X[lo:hi], nets[1], 0.0, *nets[1].parameters())
((out - Y[lo:hi]) ** 2).sum().backward()                  # (sum, not mean: the SUM over ranks is then the full-batch gradient)
assert bucket.pending() == 2
scale = bucket.finish([nets[1]], defer_scale=True)        # the fine network first, as the two-stream step does ...
assert scale == 0.5 and bucket.pending() == 1
scale = bucket.finish([nets[0]], defer_scale=True)        # ... then the other one
assert scale == 0.5 and bucket.pending() == 0 and bucket.collectives == 1
tails = bucket.tails()
assert [float(t) for t in tails] == [1.0, 0.0], [float(t) for t in tails]      # rank 1's flag reached both ranks, net 0 only
for r in ref:
    r.zero_grad(set_to_none=True)
out = FlatGradFn.apply(X, *ref[0].parameters()) + 0.5 * FlatGradFn.apply(X, *ref[1].parameters())
((out - Y) ** 2).sum().backward()
for n, r in zip(nets, ref):
    for p, q in zip(n.parameters(), r.parameters()):      # .grad holds the SUM over the ranks = the full batch's gradient
        assert float((p.grad - q.grad).abs().max()) <= 1e-3 * float(q.grad.abs().max()) + 1e-6
    assert "_grad_flat" not in n.__dict__       # the bucket let go of the buffer once exchanged (its tail view keeps it alive)
for t, n in zip(bucket.tails(), nets):
    assert t.data_ptr() == next(n.parameters()).grad.data_ptr() + 4 * sum(p.numel() for p in n.parameters())       # in place
print(f"rank {rank} ok")
'''


def test_data_parallel_per_network_inplace_allreduce_gloo(tmp_path):
    """world_size 2 on CPU/gloo with the real NeRF modules and gradients laid out exactly like MlpFn.backward's flat
    slices: the hooks enqueue ONE in-place all-reduce per network during backward (no gather / scatter copies), every
    rank ends up with the full-batch gradient, and replicas stay bit-identical over 3 Adam steps."""
    script = tmp_path / "dp_nerf_worker.py"
    script.write_text(_DP_NERF_WORKER)
    port = 29900 + (os.getpid() % 90)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out


def _args(ckpt_dir, **over):
    from argparse import Namespace
    a = dict(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, N_samples=64, netdepth=8,
             netwidth=256, netdepth_fine=8, netwidth_fine=256, netchunk=65536, lrate=5e-4, coarse_lrate=5e-4,
             ft_path=None, ckpt_dir=ckpt_dir, expname="exp", no_reload=False, perturb=1.0, white_bkgd=True,
             raw_noise_std=0.0, mode="linear", color_mode="midpoint", dataset="blender", no_ndc=False, lindisp=False,
             lrate_decay=500, constant_init=1000, chunk=32768)
    a.update(over)
    return Namespace(**a)


def test_checkpoint_wire_format_roundtrip(built, tmp_path, capsys):
    """Checkpoints use the reference's dict keys / file naming (run_plnerf.py:1324-1332) and create_nerf
    resumes from the newest '*tar*' in ckpt_dir/expname (454-471)."""
    (tmp_path / "exp").mkdir()
    args = _args(str(tmp_path))
    cpu = torch.device("cpu")
    kw, kw_test, start, grad_vars, opt, opt_c = built.create_nerf(args, device=cpu)
    assert start == 0 and set(kw) >= {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples",
                                      "network_fn", "white_bkgd", "raw_noise_std", "mode", "color_mode"}
    assert kw_test["perturb"] is True and kw_test["raw_noise_std"] == 0.
    assert grad_vars[0] is next(kw["network_fine"].parameters())          # `optimizer` drives the fine network
    path = built.checkpoint_path(str(tmp_path), "exp", 1234)
    built.save_checkpoint(path, 1234, kw["network_fn"], kw["network_fine"], opt)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"global_step", "network_fn_state_dict", "network_fine_state_dict", "optimizer_state_dict"}
    assert list(ck["network_fn_state_dict"]) == [k for k, _ in __import__("oracle.plnerf_oracle", fromlist=["x"]).param_shapes()]
    kw2, _, start2, _, _, _ = built.create_nerf(args, device=cpu)
    assert start2 == 1234
    for a, b in zip(kw["network_fine"].parameters(), kw2["network_fine"].parameters()):
        assert torch.equal(a, b)
    for a, b in zip(kw["network_fn"].parameters(), kw2["network_fn"].parameters()):
        assert torch.equal(a, b)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_reference_reloads_a_checkpoint_written_here(built, tmp_path, capsys):
    """Wire format, the other direction (run_plnerf.py:454-471): the reference's own create_nerf resumes from a file
    written by plnerf_amd.save_checkpoint -- step counter, both networks and the fine optimizer's Adam state."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from _ref_import import import_reference
    R_, _ = import_reference()
    (tmp_path / "exp").mkdir()
    args = _args(str(tmp_path), no_reload=True)
    cpu = torch.device("cpu")
    kw, _, _, grad_vars, opt, opt_c = built.create_nerf(args, device=cpu)
    for i, p in enumerate(grad_vars):      # a non-trivial optimizer state: one step on synthetic gradients
        p.grad = torch.full_like(p, 1e-3 * (i + 1))
    opt.step()
    built.save_checkpoint(built.checkpoint_path(str(tmp_path), "exp", 77), 77, kw["network_fn"], kw["network_fine"], opt)
    args_ref = _args(str(tmp_path), no_reload=False)
    kw_r, _, start_r, grad_vars_r, opt_r, _ = R_.create_nerf(args_ref)
    assert start_r == 77
    for a, b in zip(kw["network_fn"].parameters(), kw_r["network_fn"].parameters()):
        assert torch.equal(a.detach(), b.detach().cpu())
    for a, b in zip(kw["network_fine"].parameters(), kw_r["network_fine"].parameters()):
        assert torch.equal(a.detach(), b.detach().cpu())
    for p, q in zip(grad_vars, grad_vars_r):
        sa, sb = opt.state[p], opt_r.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 1.0
        assert torch.equal(sa["exp_avg"], sb["exp_avg"].cpu()) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"].cpu())


def test_select_rays_matches_get_rays(built):
    H, W, f = 20, 30, 25.0
    K = [[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]
    c2w = built.rays.pose_spherical(40.0, -30.0, 4.0)[:3, :4]
    o, d = built.get_rays(H, W, K, c2w)
    g = torch.Generator().manual_seed(3)
    batch, rows, cols = built.select_rays(H, W, K, c2w, 50, generator=g)
    assert len(set((rows * W + cols).tolist())) == 50                     # distinct pixels
    assert torch.allclose(batch[1], d[rows, cols], atol=1e-6) and torch.equal(batch[0], o[rows, cols])
    batch, rows, cols = built.select_rays(H, W, K, c2w, 30, generator=g, precrop=(5, 6))
    assert rows.min() >= H // 2 - 5 and rows.max() < H // 2 + 5 and cols.min() >= W // 2 - 6 and cols.max() < W // 2 + 6


def test_lr_schedule_matches_reference_quirk(built, tmp_path):
    (tmp_path / "exp").mkdir()
    args = _args(str(tmp_path), no_reload=True, lrate=5e-4, coarse_lrate=1e-4)
    kw, _, start, _, opt, opt_c = built.create_nerf(args, device=torch.device("cpu"))
    ts = built.TrainStep(args, kw, opt, opt_c, start=start, distributed=False)
    ts.global_step = 2500
    assert abs(ts.learning_rate() - 5e-4 * 0.1 ** (2500 / 500000)) < 1e-12
    assert opt_c.param_groups[0]["lr"] == 1e-4        # until the first step; afterwards the fine rate (line 1315)


def test_depth_variant_host_logic(built):
    """CPU-checkable pieces of the depth-supervised mirror (plnerf_amd.depth): the pi-scaled encoder equals the
    oracle's bit for bit, the networks come out 57 | 3 wide with the DenseLayer initialisation (xavier-uniform,
    zero biases; depth_supervised_exps/model/run_nerf_helpers.py:89-98), and the space-carving loss equals the
    oracle's restatement including its gradient."""
    from argparse import Namespace
    from oracle import plnerf_oracle as orc
    from plnerf_amd import depth as Dp
    emb, ch = Dp.get_embedder(9, 0)
    embd, chv = Dp.get_embedder(0, 0)
    assert (ch, chv) == (57, 3)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_depth_variant.npz"))
    T = torch.from_numpy
    ro, rd = Dp.get_rays(int(g["rays_H"]), int(g["rays_W"]), T(g["rays_intrinsic"]), T(g["rays_c2w"]))
    assert torch.equal(ro, T(g["rays_o"])) and torch.equal(rd, T(g["rays_d"]))
    ro, rd = Dp.get_rays(int(g["rays_H"]), int(g["rays_W"]), T(g["rays_intrinsic"]), T(g["rays_c2w"]), T(g["rays_coords"]))
    assert torch.equal(ro, T(g["rays_o_coords"])) and torch.equal(rd, T(g["rays_d_coords"]))
    x = torch.randn(11, 3, generator=torch.Generator().manual_seed(0)) * 2
    assert torch.equal(emb(x), orc.positional_encoding_pi(x, 9)) and torch.equal(embd(x), x)
    args = Namespace(multires=9, i_embed=0, use_viewdirs=True, multires_views=0, input_ch_cam=0, N_importance=8,
                     N_samples=8, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, lrate=5e-4, perturb=1.0,
                     white_bkgd=True, raw_noise_std=0.0, mode="linear", color_mode="midpoint")
    kw, kw_test, start, grad_vars, opt = Dp.create_nerf(args, device=torch.device("cpu"))
    net = kw["network_fn"]
    assert [tuple(p.shape) for p in net.parameters()] == [s for _, s in orc.param_shapes_depth()]
    # (57 | 3 channels are a prefix of the compiled 63 | 27: since round 3 the kernel's own encoding serves them, with the
    # encoder's input scale pi as an argument of the call; a camera code still takes the embedded route)
    assert net.is_supported() and net.has_fused_encoding() and net.density_activation == "softplus"
    import plnerf_amd as P
    assert not P.NeRF(input_ch=57, input_ch_views=3, input_ch_cam=4, use_viewdirs=True).has_fused_encoding()
    assert not P.NeRF(input_ch=58, input_ch_views=3, use_viewdirs=True).has_fused_encoding()
    assert all(float(m.bias.detach().abs().max()) == 0.0 for m in net.modules() if isinstance(m, torch.nn.Linear))
    w = net.pts_linears[1].weight
    bound = (2.0 ** 0.5) * (6.0 / (256 + 256)) ** 0.5                    # xavier-uniform with the relu gain
    assert float(w.detach().abs().max()) <= bound + 1e-6 and float(w.detach().abs().max()) > 0.9 * bound
    assert len(grad_vars) == 48 and len(opt.param_groups[0]["params"]) == 48 and kw_test["perturb"] is False
    # the space-carving loss is a kernel (plnerf_depth_loss) behind the reference's function name since round 4: like
    # every operator of the package it refuses CPU tensors loudly instead of falling back (its values and gradients are
    # compared with the oracle's restatement on the GPU: tests/test_gpu_depth_kernels.py)
    gen = torch.Generator().manual_seed(2)
    hyp = (torch.rand(7, 16, generator=gen) * 4 + 2).requires_grad_(True)
    target_h = torch.rand(3, 7, 1, generator=gen) * 4 + 2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Dp.compute_space_carving_loss(hyp, target_h)
    # the cache helper stays a few torch reductions (off the training step): the reference's choice of hypothesis
    pred = torch.rand(4, 5, 6, generator=gen) * 4 + 2
    th = torch.rand(3, 4, 5, 1, generator=gen) * 4 + 2
    idx = Dp.get_space_carving_idx(pred, th)
    assert idx.shape == (4, 5, 6) and torch.equal(idx, (pred[None] - th).abs().argmin(0))
    joint = Dp.get_space_carving_idx(pred, th, is_joint=True)
    best = int((pred[None] - th).abs().reshape(3, -1).mean(1).argmin())
    assert joint.shape == (4, 5, 1) and bool((joint == best).all())


def _run_bench(*flags, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), env=e, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_spawns_its_own_ranks(built, n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the driver's command form) must launch the N
    ranks itself: here on CPU / gloo with the stand-in step, the real rendezvous, dp.GradientBucket and report path.
    Rank 0 prints ONE JSON line as the last line of stdout; the world size is what torch.distributed reports.  N = 8 is
    BASELINE configs[2]'s rank count (eight launch loops on one host)."""
    import json
    r = _run_bench("--gpus", str(n), "--stub-cpu", "--steps", "4", "--warmup", "1", timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert r.stdout.strip().splitlines()[-1] == lines[0]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 4 and out["warmup"] == 1
    assert out["config"]["rccl_world_size"] == n and out["config"]["self_launched"] is True
    assert out["config"]["backend"] == "gloo"
    rk = out["ranks"]
    assert rk["ms_per_step"]["min"] <= rk["ms_per_step"]["max"] and rk["allreduce_exposed_ms_max_over_ranks"] > 0
    assert "NOT a measurement" in out["metric"]
    # launch hygiene: every self-launched rank runs with a bounded OpenMP / intra-op thread count (the logical
    # CPUs shared out over the ranks, at most 8) instead of one thread per CPU per rank, and reports how long its host needed
    # to ENQUEUE a step -- what tells a slow launch loop from an exposed collective on the first multi-GPU run
    import bench
    assert rk["omp_num_threads"] == bench.rank_threads(n) and 1 <= rk["omp_num_threads"] <= 8
    assert 0 < rk["host_ms_per_step"]["min"] <= rk["host_ms_per_step"]["max"] <= rk["ms_per_step"]["max"] * 1.001


def test_bench_rank_pinning_is_silent_without_topology(built, monkeypatch):
    """bench.pin_rank: binds a rank to the cores of its GPU's NUMA node when the KFD topology says which (read from sysfs), and
    does nothing -- silently -- when it cannot be read (this container: no /sys/class/kfd).  The affinity mask only ever
    shrinks to a subset of the current one."""
    import bench
    before = os.sched_getaffinity(0)
    try:
        info = bench.pin_rank(0, 8)
        after = os.sched_getaffinity(0)
        assert after <= before and len(after) >= 1
        assert info["omp_num_threads"] >= 1
        if bench.gpu_numa_node(0) is None:
            assert after == before and info["numa_node"] is None
        assert bench.pin_rank(0, 1)["numa_node"] is None          # a single rank is never pinned
    finally:
        os.sched_setaffinity(0, before)


def test_bench_launcher_refuses_without_devices_and_propagates_failures(built):
    """No GPU here: `--gpus 2` must say so and exit 2 without launching anything; a rank that dies must take the
    launcher down with a non-zero exit code (the other rank, blocked in the rendezvous, is stopped by PID)."""
    import torch
    if not torch.cuda.is_available():
        r = _run_bench("--gpus", "2")
        assert r.returncode == 2 and "needs 2 visible devices" in r.stderr, (r.returncode, r.stderr[-500:])
    r = _run_bench("--gpus", "2", "--stub-cpu", "--stub-fail-rank", "1", "--launch-timeout", "120")
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr and "simulated rank failure" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_under_torchrun_environment(built):
    """The torchrun contract still works: with RANK / WORLD_SIZE in the environment bench.py is one rank and does
    not spawn.  (world size 1 here: a lone stub rank over gloo through --force-dist.)"""
    import json
    port = 29900 + (os.getpid() % 90)
    r = _run_bench("--gpus", "1", "--stub-cpu", "--force-dist", "--steps", "2", "--warmup", "1",
                   env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["self_launched"] is False

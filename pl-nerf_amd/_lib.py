"""ctypes binding of libplnerf_hip.so (C ABI declared in include/plnerf_hip.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on a HIP
device, the call raises.  Build the library with `python __graft_entry__.py` (or
`make -C pl-nerf_amd/csrc`); hipcc cross-compiles gfx950 without a GPU present.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLNERF_HIP_LIB overrides the library path (kernel experiments: ablation builds under /tmp)
LIB_PATH = os.environ.get("PLNERF_HIP_LIB") or os.path.join(_HERE, "libplnerf_hip.so")

MODE = {"constant": 0, "linear": 1}
COLOR = {"midpoint": 0, "left": 1}
PRECISION = {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3, "f16": 4}
GUARDED_PRECISIONS = ("f16x3", "f16", "bf16x3", "bf16")      # modes whose kernels can set a bit of the range status word
N_PARAM_TENSORS = 24
ABI_VERSION = 600                      # PLNERF_VERSION of include/plnerf_hip.h this binding was written against
QUAD_RAYS_PER_GROUP = 4                # PLNERF_QUAD_RAYS_PER_GROUP: rays per workgroup of plnerf_quad_bwd (its absmax_out)
DEPTH_LOSS_WORKSPACE_BYTES = 4096      # PLNERF_DEPTH_LOSS_WORKSPACE_BYTES
IMAGE_LOSS_WORKSPACE_BYTES = 4096      # PLNERF_IMAGE_LOSS_WORKSPACE_BYTES
# plnerf_mlp_fwd's `fwd_kernel` argument (PLNERF_FWD_KERNEL_*).  The library has no setting of its own; this BINDING
# takes its default from the environment (PLNERF_FWD_KERNEL=rr | pp: the test suite's and tools/' passes over both
# forward kernels) and hands it to every call.
FWD_KERNELS = {"auto": 0, "rr": 1, "pp": 2}
FWD_KERNEL = FWD_KERNELS.get(os.environ.get("PLNERF_FWD_KERNEL") or "auto", 0)
RANGE_ACTIVATION, RANGE_WEIGHT, RANGE_SAVED = 1, 2, 4      # bits of the packed buffer's status word (plnerf_hip.h)

c_f = ctypes.c_void_p      # device pointer
c_i = ctypes.c_int
c_s = ctypes.c_void_p      # hipStream_t

# name -> (restype, argtypes); mirrors include/plnerf_hip.h one to one
SIGNATURES = {
    "plnerf_version": (c_i, []),
    "plnerf_build_flags": (c_i, []),
    "plnerf_error_string": (ctypes.c_char_p, [c_i]),
    "plnerf_quad_fwd": (c_i, [c_f] * 6 + [c_i] * 6 + [c_f] * 7 + [c_s]),
    "plnerf_quad_bwd": (c_i, [c_f] * 6 + [c_i] * 6 + [c_f] * 7 + [c_f, c_s]),
    "plnerf_quad_bwd_rays": (c_i, [c_f] * 6 + [c_i] * 6 + [c_f] * 7 + [c_f] * 4 + [c_s]),
    "plnerf_sample_const": (c_i, [c_f] * 3 + [c_i] * 4 + [c_f] * 2 + [c_s]),
    "plnerf_sample_const_bwd": (c_i, [c_f] * 3 + [c_i] + [c_f] * 2 + [c_i] * 3 + [c_f] + [c_s]),
    "plnerf_sample_pl": (c_i, [c_f] * 7 + [c_i] * 4 + [ctypes.c_float] * 2 + [c_f] * 5 + [c_s]),
    "plnerf_sample_pl_bwd": (c_i, [c_f] * 6 + [c_i] + [c_f] * 2 + [c_i] * 3 + [ctypes.c_float] * 2 + [c_f] * 2 + [c_s]),
    "plnerf_sample_pl_bwd_rays": (c_i, [c_f] * 6 + [c_i] + [c_f] * 2 + [c_i] * 3 + [ctypes.c_float] * 2 + [c_f] * 3 + [c_s]),
    "plnerf_stratified_z": (c_i, [c_f] * 4 + [c_i] * 3 + [c_f] + [c_s]),
    "plnerf_ray_points": (c_i, [c_f] * 3 + [c_i] * 2 + [c_f] + [c_s]),
    "plnerf_merge_sort": (c_i, [c_f] * 4 + [c_i] * 3 + [c_f] + [c_s]),
    "plnerf_coarse_epilogue": (c_i, [c_f] * 8 + [c_i, ctypes.c_uint64, ctypes.c_uint32] + [c_i] * 7 +
                               [ctypes.c_float] * 2 + [c_f] * 10 + [c_s]),
    "plnerf_fine_epilogue": (c_i, [c_f] * 7 + [c_i, ctypes.c_uint64, ctypes.c_uint32] + [c_i] * 7 + [ctypes.c_float] * 2 +
                             [c_f] * 11 + [c_s]),
    "plnerf_uniform": (c_i, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i, c_i, c_i, c_f, c_s]),
    "plnerf_normal": (c_i, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_i, c_i, c_i, c_f, c_s]),
    "plnerf_select_rays": (c_i, [c_i, c_i] + [ctypes.c_float] * 4 + [ctypes.POINTER(ctypes.c_float), c_f] + [c_i] * 4 +
                           [ctypes.c_uint64, ctypes.c_uint32, c_i, c_i, ctypes.c_float, ctypes.c_float] + [c_f] * 7 +
                           [c_s]),
    "plnerf_ndc_rays": (c_i, [c_i, c_i, ctypes.c_double, ctypes.c_double, c_f, c_f, c_i, c_f, c_f, c_s]),
    "plnerf_coarse_samples": (c_i, [c_f] * 6 + [ctypes.c_uint64, ctypes.c_uint32] + [c_i] * 5 + [c_f] * 2 + [c_s]),
    "plnerf_image_loss": (c_i, [c_f] * 3 + [c_i] + [c_f] * 5 + [c_s]),
    "plnerf_depth_loss": (c_i, [c_f] * 6 + [c_i] * 5 + [c_f] + [ctypes.c_float] * 2 + [c_f] * 5 + [c_s]),
    "plnerf_depth_joint_sums": (c_i, [c_f] * 3 + [c_i] * 4 + [ctypes.c_float, c_f, c_s]),
    "plnerf_embed_rows": (c_i, [c_f] * 3 + [c_i] * 5 + [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.c_float,
                                c_f, c_s]),
    "plnerf_gemm_f32": (c_i, [c_f, ctypes.c_int64, ctypes.c_int64, c_f, ctypes.c_int64, ctypes.c_int64, c_f, c_f] + [c_i] * 6 +
                        [c_f, ctypes.c_int64, c_i, c_f, c_s]),
    "plnerf_mlp_packed_bytes": (ctypes.c_size_t, [c_i]),
    "plnerf_mlp_pack_weights": (c_i, [ctypes.POINTER(ctypes.c_void_p), c_i, c_i, c_i, c_f, c_s]),
    "plnerf_mlp_input_grad": (c_i, [ctypes.POINTER(ctypes.c_void_p), c_i, c_i, c_i, c_i, c_f, c_f, c_s]),
    "plnerf_mlp_saved_bytes": (ctypes.c_size_t, [c_i, c_i]),
    "plnerf_mlp_bwd_workspace_bytes": (ctypes.c_size_t, [c_i, c_i]),
    "plnerf_mlp_fwd": (c_i, [c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, ctypes.c_float, c_f, c_f, c_i, c_s]),
    "plnerf_mlp_bwd": (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_f, ctypes.c_float, c_f,
                             ctypes.POINTER(ctypes.c_void_p), c_f, c_s]),
    "plnerf_mlp_bwd_multi": (c_i, [c_i, ctypes.POINTER(ctypes.c_void_p), c_i, ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), c_i, c_i,
                                   ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.c_float, ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_s]),
    "plnerf_mlp_saved_layout": (c_i, [c_i, c_i, c_i]),
    "plnerf_adam_step": (c_i, [c_f, c_f, c_f, c_f, ctypes.c_int64] + [ctypes.c_float] * 4 + [c_i, ctypes.c_float, ctypes.c_float,
                                c_f, c_f, c_f, c_s]),
    "plnerf_mlp_status_offset": (ctypes.c_size_t, [c_i]),
}

_lib = None


def lib():
    """The loaded library; raises ImportError (with the build recipe) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or "
                "`make -C pl-nerf_amd/csrc` (hipcc --offload-arch=gfx950). "
                "plnerf_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        tools_build = os.environ.get("PLNERF_ALLOW_TOOLS_BUILD") == "1"
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                if tools_build:      # (tools/ab.sh against a library of an earlier commit: entry points it lacks stay unbound)
                    continue
                raise AttributeError(f"{LIB_PATH} does not export {name}: header / library mismatch")
            fn.restype = res
            fn.argtypes = args
        # The argument lists above are positional: a library of another ABI (a stale build, a variant linked from old
        # objects) would be called with shifted pointers -- compare before the first call.  tools/ A/B legs
        # against an older library whose signatures are known to match set PLNERF_ALLOW_TOOLS_BUILD=1.
        version = handle.plnerf_version()
        if version != ABI_VERSION and not tools_build:
            raise ImportError(
                f"{LIB_PATH} reports plnerf_version() = {version}, this binding is written against {ABI_VERSION} "
                "(include/plnerf_hip.h): rebuild with `make -C pl-nerf_amd/csrc clean all`.")
        flags = handle.plnerf_build_flags()
        if flags and os.environ.get("PLNERF_ALLOW_TOOLS_BUILD") != "1":
            raise ImportError(
                f"{LIB_PATH} was built with tools-only switches (plnerf_build_flags() = {flags}: bit 2 = trace hooks; "
                "bits 0 / 1 = the timing ablations of builds before round 5).  It is not the "
                "product library: rebuild with `make -C pl-nerf_amd/csrc` (no -D switches), or set "
                "PLNERF_ALLOW_TOOLS_BUILD=1 for a measurement script under tools/.")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().plnerf_error_string(rc).decode()} (code {rc})")


def dptr(t, name="tensor", dtype=torch.float32):
    """Device pointer of a contiguous tensor on a HIP device (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            f"plnerf_amd: `{name}` is on {t.device}; the HIP path needs a GPU tensor "
            "(there is no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"plnerf_amd: `{name}` must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"plnerf_amd: `{name}` must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as the ABI's `plnerf_stream_t` (every entry point enqueues on the
    stream it is handed).  torch.cuda.current_stream() builds a Stream object through three Python layers (11 us; a
    training step makes ~11 calls); the raw getter is the same value without them."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_table(tensors, name="params"):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = dptr(t, f"{name}[{i}]").value
    return arr

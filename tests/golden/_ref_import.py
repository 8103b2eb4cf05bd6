"""Import the read-only PL-NeRF reference (/root/reference) as a CPU oracle.

Only used by tests/golden/make_golden.py, in the build container.  Nothing on
the GPU box may import this: /root/reference does not exist there.

The reference's run_plnerf.py imports eight packages that are absent in this
image (torchvision, imageio, skimage, lpips, configargparse, cv2, natsort,
tensorboard; run_plnerf.py:10-35) but uses none of them on the hot path, so
inert stand-in modules are registered in sys.modules before the import.  No
reference source is copied; bytecode writing is disabled.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        return _Inert()


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


_STUBS = [
    "torchvision", "torchvision.transforms", "torchvision.utils", "imageio",
    "skimage", "skimage.metrics", "lpips", "configargparse", "cv2", "natsort",
    "tensorboard", "torch.utils.tensorboard", "mcubes", "trimesh", "PIL",
    "PIL.Image", "matplotlib", "matplotlib.pyplot",
]


def import_reference():
    """Returns (run_plnerf module, run_nerf_helpers module)."""
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _InertModule(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import run_nerf_helpers  # noqa: E402
    import run_plnerf  # noqa: E402
    return run_plnerf, run_nerf_helpers

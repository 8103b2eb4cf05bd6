"""plnerf_amd -- MI355X-native implementation of PL-NeRF's volume-rendering hot path.

Drop-in for the render operator surface of the reference's run_plnerf.py /
run_nerf_helpers.py (create_nerf, render, render_rays, raw2outputs, sample_pdf,
sample_pdf_reformulation, NeRF, get_embedder, run_network, ...).  All numerical work runs
in the hand-written gfx950 kernels of libplnerf_hip.so; there is no CPU fallback.
"""
from . import _lib
from .nerf import NeRF, Embedder, get_embedder
from .rays import get_rays, get_rays_np, ndc_rays
from .optim import FlatAdam
from .train import TrainStep, checkpoint_path, img2mse, save_checkpoint, select_rays, select_view_rays
from .raybatch import RayColumns
from .functional import DrawSource, set_draw_source
from . import depth   # depth-supervised variant of the path (depth_supervised_exps/)
from .render import (batchify, batchify_rays, compute_weights, compute_weights_piecewise_linear, create_nerf,
                     raw2outputs, render, render_path, render_rays, run_network, sample_pdf,
                     sample_pdf_reformulation)



def library_path():
    return _lib.LIB_PATH


def library_version():
    return _lib.lib().plnerf_version()


__all__ = [
    "NeRF", "Embedder", "get_embedder", "get_rays", "get_rays_np", "ndc_rays", "batchify", "batchify_rays",
    "compute_weights", "compute_weights_piecewise_linear", "create_nerf", "raw2outputs", "render", "render_path",
    "render_rays", "run_network", "sample_pdf", "sample_pdf_reformulation", "img2mse", "library_path",
    "library_version", "depth", "FlatAdam", "TrainStep", "save_checkpoint", "checkpoint_path", "select_rays", "select_view_rays", "RayColumns", "DrawSource",
    "set_draw_source",
]

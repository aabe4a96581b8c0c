"""sdfest_amd -- MI355X-native implementation of sdfest's render-and-compare hot path.

Public surface mirrors the reference's ``sdfest.differentiable_renderer``
(``Camera``, ``render_depth_gpu``) so that ``SDFPipeline.render`` can bind to it
unchanged; see INTEGRATION.md.
"""
from .differentiable_renderer import (BatchRenderPlan, Camera, SDFRendererFunctionGPU,
                                      render_depth_batch, render_depth_l1_batch,
                                      render_depth_gpu)

from .losses import nn_loss, pc_loss, pc_loss_batch, point_constraint_loss
from .vae import SDFDecoder
from .pipeline import FusedRenderAndCompare, RenderAndCompare
from .init_network import NoDepthError, SDFPoseNet, nn_init
from .simple_setup import SDFPipeline
from .so3grid import SO3Grid

__all__ = ["SDFPipeline", "NoDepthError", "RenderAndCompare", "FusedRenderAndCompare", "SDFPoseNet", "nn_init", "SO3Grid", "SDFDecoder", "pc_loss", "pc_loss_batch", "nn_loss", "point_constraint_loss", "BatchRenderPlan", "Camera", "SDFRendererFunctionGPU", "render_depth_gpu", "render_depth_batch", "render_depth_l1_batch"]

"""slim-switch-moe-vit on MI355X: the Switch-MoE ViT hot path (router -> dispatch -> grouped expert
GEMM -> combine) as hand-written gfx950 HIP kernels behind the reference's nn.Module surface."""
from . import _lib, ops  # noqa: F401
from .fmoe import (FMoETransformerMLP, FMoELinear, NaiveGate, SwitchGate, fastmoe_v11_state_dict,  # noqa: F401
                   ddp_ignore_expert_parameters)
from .vit import (Block, VisionTransformer, create_model, register_model, list_models,  # noqa: F401
                  deit_tiny_patch16_224, deit_base_patch16_224)
from .resmoe import *  # noqa: F401,F403
from .engine import evaluate, accuracy, train_one_epoch, GraphedForward, GraphedTrainStep  # noqa: F401
from .optim import AdamW, NativeScaler, invalidate_weight_images  # noqa: F401

__version__ = "0.1.0"

"""MI355X-native (gfx950) implementation of the neural-process forward/backward hot path of
MarinerQ/npf_GWwaveform, behind the reference's own module API.  See DESIGN.md.

The compute runs in hand-written HIP kernels (``csrc/``) reached through a C ABI
(``include/npf_hip.h``); there is no CPU or eager-PyTorch fallback for the path.
"""
from .architectures import (MLP, DotAttender, MergeFlatInputs, MultiheadAttender, SelfAttention, TransformerAttender, get_attender,
                            merge_flat_input)
from .chain import set_compute_dtype
from .datasplit import CntxtTrgtGetter, GetRandomIndcs, GetRangeIndcs, get_all_indcs
from .evaluate import eval_loglike
from .losses import CNPFLoss, ELBOLossLNPF, LightTailPareto, NLLLossLNPF, SUMOLossLNPF
from .neuralproc import (CNP, LNP, AttnCNP, AttnLNP, HeadDistribution, LatentNeuralProcessFamily, MultivariateNormalDiag,
                         NeuralProcessFamily)

# north-star aliases: NPFModel / encode / aggregate / decode
NPFModel = NeuralProcessFamily


def encode(model: NeuralProcessFamily, X, Y):
    """Per-point encoding: x_encoder then xy_encoder (the per-point part of
    ``encode_globally``); row-major [B, P, r]."""
    return model.xy_encoder(model.x_encoder(X), Y)


def aggregate(model: NeuralProcessFamily, X_cntxt_enc, R, X_trgt_enc, z_samples=None):
    """Context aggregation: mean (CNP/LNP) or scaled-dot cross attention (Attn*), plus the
    latent merge -- i.e. ``trgt_dependent_representation``."""
    return model.trgt_dependent_representation(X_cntxt_enc, z_samples, R, X_trgt_enc)


def decode(model: NeuralProcessFamily, X_trgt_enc, R_trgt):
    """``NeuralProcessFamily.decode``."""
    return model.decode(X_trgt_enc, R_trgt)


__all__ = [
    "MLP", "MergeFlatInputs", "merge_flat_input", "DotAttender", "MultiheadAttender", "TransformerAttender", "SelfAttention", "get_attender",
    "NeuralProcessFamily", "LatentNeuralProcessFamily", "CNP", "LNP", "AttnCNP", "AttnLNP", "NPFModel",
    "CNPFLoss", "ELBOLossLNPF", "NLLLossLNPF", "SUMOLossLNPF", "LightTailPareto", "MultivariateNormalDiag", "encode", "aggregate", "decode",
    "CntxtTrgtGetter", "GetRandomIndcs", "GetRangeIndcs", "get_all_indcs", "set_compute_dtype", "eval_loglike", "HeadDistribution",
]

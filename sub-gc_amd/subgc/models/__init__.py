"""Model factory with the reference's entry point: `models.setup(opt)` (models/__init__.py:43-59)."""
from __future__ import annotations

import os

import torch

from .AttModel import AttModel, TopDownModel  # noqa: F401
from .CaptionModel import CaptionModel  # noqa: F401
from .loss_wrapper import LossWrapper  # noqa: F401


def setup(opt):
    """Build the captioner named by `opt.caption_model` (only 'topdown' exists, as in the reference)
    and, when `opt.start_from` is set, resume its weights from `<start_from>/model.pth`."""
    if opt.caption_model != "topdown":
        raise Exception("Caption model not supported: {}".format(opt.caption_model))
    model = TopDownModel(opt)
    start = vars(opt).get("start_from", None)
    if start is not None:
        assert os.path.isdir(start), " %s must be a a path" % start
        assert os.path.isfile(os.path.join(start, "infos_" + opt.id + ".pkl")), "infos.pkl file does not exist in path %s" % start
        model.load_state_dict(torch.load(os.path.join(start, "model.pth"), map_location="cpu"))
    return model


def total_loss(out):
    """`lang_loss + gpn_loss` of a LossWrapper result (train.py:154-156), summed by a C-ABI launch (gpn_loss may be None: Full-GC)."""
    from .. import functions as F_
    if out.get("gpn_loss") is None:
        return out["lang_loss"]
    return F_.SumScalarsFn.apply(out["lang_loss"], out["gpn_loss"])

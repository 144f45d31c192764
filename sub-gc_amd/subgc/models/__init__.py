"""Model factory with the reference's entry points: `models.setup(opt)` (models/__init__.py:43-59) and the Flickr30k fine-tune
restore `optimistic_restore` (models/__init__.py:14-41)."""
from __future__ import annotations

import os

import numpy as np
import torch

from .AttModel import AttModel, TopDownModel  # noqa: F401
from .CaptionModel import CaptionModel  # noqa: F401
from .loss_wrapper import LossWrapper  # noqa: F401


def optimistic_restore(network, state_dict, word_map_path="data/word_mapping.npy"):
    """Size-tolerant restore of a COCO-trained checkpoint into a Flickr30k model (reference models/__init__.py:14-41, same order of
    effects, same messages, same return value):

    1. every checkpoint tensor whose name AND size match is copied in; an unknown name or a size mismatch is reported and makes the
       result False (the tensor is skipped, the network keeps its own values);
    2. rows of `embed.0.weight` are remapped through `word_map_path` (int array over the NETWORK's vocabulary rows: entry i is the
       checkpoint row to copy into row i, -1 = keep) -- after step 1, so a remapped row wins over a same-size straight copy;
    3. network tensors the checkpoint lacks are reported and make the result False.

    The copies go through `network.state_dict()`, whose tensors share storage with the flat parameter buffer, so the bf16 weight
    snapshot and the decode caches see the change through the parameters' version counters like after `load_state_dict`."""
    mismatch = False
    own_state = network.state_dict()
    with torch.no_grad():
        for name, param in state_dict.items():
            if name not in own_state:
                print("Unexpected key {} in state_dict with size {}".format(name, param.size()))
                mismatch = True
            elif param.size() == own_state[name].size():
                own_state[name].copy_(param)
            else:
                print("Network has {} with size {}, ckpt has {}".format(name, own_state[name].size(), param.size()))
                mismatch = True
        word_map = np.load(word_map_path)
        for name in ["embed.0.weight"]:
            rows = np.nonzero(word_map != -1)[0]
            if rows.size:
                dst, src = own_state[name], state_dict[name]
                take = torch.as_tensor(word_map[rows].astype(np.int64), device=src.device)
                dst.index_copy_(0, torch.as_tensor(rows.astype(np.int64), device=dst.device), src.index_select(0, take).to(dst.device, dst.dtype))
    print("\ncopy COCO-pre-trained embedding done!\n")
    missing = set(own_state.keys()) - set(state_dict.keys())
    if len(missing) > 0:
        print("We couldn't find {}".format(",".join(missing)))
        mismatch = True
    return not mismatch


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

"""`LossWrapper`: model + language-model criterion, the module `train.py` wraps in DataParallel.

Same call signature and return dict as reference `models/loss_wrapper.py:14-27`; the criterion
(`misc/utils.py:115-124`) runs as the masked-NLL HIP kernels.
"""
from __future__ import annotations

import torch

from .. import functions as F_


class LanguageModelCriterion(torch.nn.Module):
    def forward(self, input, target, mask):
        target = target[:, : input.size(1)]
        mask = mask[:, : input.size(1)]
        return F_.MaskedNLLFn.apply(input, target, mask)


class LossWrapper(torch.nn.Module):
    def __init__(self, model, opt):
        super().__init__()
        self.opt = opt
        self.model = model
        self.crit = LanguageModelCriterion()

    def forward(self, fc_feats, att_feats, labels, masks, att_masks, gts, gt_indices, trip_pred, obj_dist, obj_box, rel_ind,
                pred_fmap, pred_dist, gpn_obj_ind, gpn_pred_ind, gpn_nrel_ind, gpn_pool_mtx):
        T = labels.size(1) - 1
        fused = (labels[:, 1:], masks[:, 1:T + 1]) if getattr(self.model, "supports_fused_crit", False) else None
        kw = {"fused_crit": fused, "need_outputs": False} if fused is not None else {}
        lang_output, gpn_loss, subgraph_score = self.model(fc_feats, att_feats, labels, att_masks, trip_pred, obj_dist, obj_box,
                                                           rel_ind, pred_fmap, pred_dist, gpn_obj_ind, gpn_pred_ind, gpn_nrel_ind,
                                                           gpn_pool_mtx, **kw)
        if fused is not None:
            lang_loss = self.model.fused_lang_loss          # criterion evaluated inside the decoder Function
        else:
            lang_loss = self.crit(lang_output, labels[:, 1:], masks[:, 1:]) if lang_output is not None else None
        return {"gpn_loss": gpn_loss, "lang_loss": lang_loss}

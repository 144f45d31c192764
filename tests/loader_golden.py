"""Helpers shared by the CPU (oracle) and GPU (HIP) replays of tests/golden/loader_*.npz: the fabricated dataset entries the
reference's `DataLoader.__getitem__` was driven on (make_golden.py loader_cases) and what it returned."""
import numpy as np

NAMES = ("fc_feats", "att_feats", "obj_dist", "rel_ind", "pred_dist", "labels", "masks", "gpn_obj_ind", "gpn_pred_ind", "gpn_nrel_ind",
         "att_masks", "gpn_pool_mtx")


class LoaderCase:
    def __init__(self, golden):
        g = golden("loader")
        self.meta, self.raw, self.out = g.meta, g.group("inputs"), g.group("out")
        m = self.meta
        self.S, self.hb, self.obj_num, self.rel_num, self.Lq = m["seq_per_img"], m["gpn_batch"], m["obj_num"], m["rel_num"], m["seq_length"]

    def image(self, b):
        r = self.raw
        off = r[f"img{b}_nrel_off"]
        nrel = [r[f"img{b}_nrel"][off[j]:off[j + 1]] for j in range(len(off) - 1)]
        return dict(iou=r[f"img{b}_node_iou_mtx"], node_masks=r[f"img{b}_node_masks"], pred_masks=r[f"img{b}_pred_masks"], nrel=nrel,
                    object_fmap=r[f"img{b}_object_fmap"], object_dist=r[f"img{b}_object_dist"], pred_dist=r[f"img{b}_pred_dist"],
                    rel_ind=r[f"img{b}_rel_ind"])

    def chosen(self, im, mask_idx):
        """mask_idx [S, hb, 2] -> (node_masks [S, 2, hb, W], pred_masks [S, 2, hb, Wp], nrel[i][side][k]) of the chosen sub-graphs."""
        ids = np.transpose(mask_idx, (0, 2, 1))                   # [S, side, k]
        nrel = [[[im["nrel"][ids[i, s, k]] for k in range(self.hb)] for s in range(2)] for i in range(self.S)]
        return im["node_masks"][ids], im["pred_masks"][ids], nrel

    def gt_ids(self):
        return np.broadcast_to(np.arange(self.S)[:, None, None], (self.S, self.hb, 2)).copy()

    def expect(self, tag, b):
        return {k: self.out[f"{tag}{b}_{k}"] for k in NAMES}

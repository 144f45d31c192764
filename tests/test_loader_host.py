"""The host half of batch assembly (subgc.assemble.choose_subgraphs / pick_captions: integer work on a few numbers per image)
against what the reference's own `DataLoader.__getitem__` drew (tests/golden/loader_*.npz) -- no GPU needed."""
import random as pyrandom

import numpy as np

from loader_golden import LoaderCase
from subgc import assemble


def test_host_sampling_draws_what_the_reference_loader_drew(golden):
    c = LoaderCase(golden)
    m = c.meta
    np.random.seed(m["np_seed"][0])
    pyrandom.seed(m["py_seed"][0])
    for b in range(m["n_images"]):
        ids = assemble.choose_subgraphs(c.image(b)["iou"], m["thres"], c.hb)
        np.testing.assert_array_equal(ids, c.out[f"smp{b}_mask_idx"], err_msg=f"image {b}")
        caps = assemble.pick_captions(c.raw["label"], c.raw["label_start_ix"], c.raw["label_end_ix"], b, c.S, c.Lq)
        np.testing.assert_array_equal(caps, c.out[f"smp{b}_labels"][:, 1:-1], err_msg=f"captions of image {b}")
    # an explicit RandomState / Random replays the same stream (no global state needed)
    rs, pr = np.random.RandomState(m["np_seed"][0]), pyrandom.Random(m["py_seed"][0])
    for b in range(m["n_images"]):
        np.testing.assert_array_equal(assemble.choose_subgraphs(c.image(b)["iou"], m["thres"], c.hb, rng=rs), c.out[f"smp{b}_mask_idx"])
        caps = assemble.pick_captions(c.raw["label"], c.raw["label_start_ix"], c.raw["label_end_ix"], b, c.S, c.Lq, rng=pr)
        np.testing.assert_array_equal(caps, c.out[f"smp{b}_labels"][:, 1:-1])

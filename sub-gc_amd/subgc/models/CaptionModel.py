"""`CaptionModel`: the mode-dispatching base class of the reference API.

Mirrors reference `models/CaptionModel.py:21-26`: `model(*args, mode='sample', **kw)` calls
`self._sample(*args, **kw)`, the default mode is `'forward'` (-> `self._forward`).  The diverse
beam search of `CaptionModel.py:28-175` is a "next" row of SURVEY.md section 8(f) and is not
built yet: asking for it fails loudly instead of silently decoding greedily.
"""
from __future__ import annotations

import torch.nn as nn


class CaptionModel(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, *args, **kwargs):
        mode = kwargs.pop("mode", "forward")
        return getattr(self, "_" + mode)(*args, **kwargs)

    def beam_search(self, init_state, init_logprobs, *args, **kwargs):
        raise NotImplementedError(
            "beam search (reference CaptionModel.py:28-175) is not part of the round-1 HIP path; "
            "decode with beam_size=1 (greedy / top-k sampling)")

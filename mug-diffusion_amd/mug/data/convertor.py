"""mug.data.convertor -- the note-grid thresholding contract of the reference
(mug/data/convertor.py:205-264): decoded logits (16, T) laid out
[is_start x4 | offset_start x4 | is_holding x4 | offset_end x4] -> hit-object strings.
A cell is on iff logit > 0 (from_logits=True).  (.osu file parsing/writing, gridify and
the mini-jack filter are host post-processing outside the hot path: SURVEY.md 8(f) rank 1.)"""
from typing import List

import numpy as np


class BeatmapMeta:
    def __init__(self, cs=4, **kw):
        self.cs = cs
        self.__dict__.update(kw)


class BaseOsuConvertor:
    def __init__(self, frame_ms, max_frame, mirror=False, from_logits=False, mirror_at_interval_prob=0, random=False,
                 rate=1.0, offset_ms=0):
        self.frame_ms = frame_ms
        self.max_frame = max_frame
        self.from_logits = from_logits
        self.rate = rate
        self.offset_ms = offset_ms


class OsuManiaConvertor(BaseOsuConvertor):
    def is_binary_positive(self, input):
        return input > 0 if self.from_logits else input > 0.5

    def note_grid(self, note_array, key_count=4):
        """Boolean (is_start, is_holding) grids -- the bit-exact parity target of the sampler."""
        a = np.asarray(note_array)
        return self.is_binary_positive(a[..., 0:key_count, :]), self.is_binary_positive(a[..., 2 * key_count:3 * key_count, :])

    def array_to_objects(self, note_array: np.ndarray, meta) -> List[str]:
        """convertor.py:232-264."""
        a = np.asarray(note_array).transpose()
        key_count = int(meta.cs)
        column_width = int(512 / key_count)
        out = []
        n = len(a)
        for column in range(key_count):
            for s in np.where(self.is_binary_positive(a[:, column]))[0]:
                start = int(round((s + np.clip(a[s, column + key_count], 0, 1)) * self.frame_ms))
                end = -1
                if s != n - 1:
                    i = s + 1
                    while (i < n and self.is_binary_positive(a[i, column + key_count * 2])
                           and not self.is_binary_positive(a[i, column])):
                        i += 1
                    e = i - 1
                    if e != s:
                        end = int(round((e + np.clip(a[e, column + key_count * 3], 0, 1)) * self.frame_ms))
                x = int(round((column + 0.5) * column_width))
                line = f"{x},192,{start},1,0,0:0:0:0:" if end == -1 else f"{x},192,{start},128,0,{end}:0:0:0:0:"
                out.append((line, start))
        out.sort(key=lambda r: r[1])
        return [r[0] for r in out]

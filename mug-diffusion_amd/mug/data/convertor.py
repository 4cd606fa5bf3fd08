"""mug.data.convertor -- what the sampling callers need from the reference's mug/data/convertor.py:

* the note-grid thresholding contract (:205-264): decoded logits (16, T) laid out
  [is_start x4 | offset_start x4 | is_holding x4 | offset_end x4] -> hit-object strings; a cell is on iff logit > 0
  (from_logits=True);
* `.osu` template parsing and chart writing (`parse_osu_file` :37-92, `save_osu_file` :94-120, `BeatmapMeta` :11-28), which
  `webui.py:397-414` / `scripts/mapping.py:486-505` call around the sampler -- SURVEY.md 8(f) rank 1, the first row after
  the hot path.  Output files are byte-identical to the reference's (tests/test_osu_io.py).

The training-side halves (`objects_to_array`, `timing_to_array`, mirroring / random augmentation) are not provided here;
`gridify` and the mini-jack filter stay the reference's own `mug/data/utils.py` (package fall-through, see mug/__init__.py)."""
import os
import string
import traceback
from typing import List, Optional, Tuple

import numpy as np

_KEEP = set("-_.()[]/\\' " + string.ascii_letters + string.digits)


def slugify(text):
    return "".join(ch for ch in text if ch in _KEEP)


def read_item(line):
    return line.split(":")[-1].strip()


class BeatmapMeta:
    """Fields of the reference dataclass (convertor.py:11-28)."""

    def __init__(self, path="", audio="", game_mode=0, convertor=None, cs=0, version="", set_id=-1, file_meta=None,
                 timing_points=None, **extra):
        self.path, self.audio, self.game_mode, self.convertor = path, audio, game_mode, convertor
        self.cs, self.version, self.set_id = cs, version, set_id
        self.file_meta = [] if file_meta is None else file_meta
        self.timing_points = [] if timing_points is None else timing_points
        self.__dict__.update(extra)

    def for_batch(self):
        return {k: getattr(self, k) for k in ("path", "audio", "game_mode", "cs", "version", "set_id")}


def _resolve_audio(osu_path, item):
    """AudioFilename next to the .osu, trying the literal name, its slug, its lower case and the slug of that (:56-72)."""
    folder = os.path.dirname(osu_path)
    candidates = (item, slugify(item), item.lower(), slugify(item.lower()))
    for name in candidates[:-1]:
        path = os.path.join(folder, name)
        if os.path.isfile(path):
            return path
    return os.path.join(folder, candidates[-1])


def parse_osu_file(osu_path, convertor_params: Optional[dict]) -> Tuple[List[str], BeatmapMeta]:
    """Splits an .osu file into its hit-object lines and everything else (`meta.file_meta`, replayed verbatim by
    save_osu_file), picking up the few fields the callers read.  A section header takes effect from the NEXT line."""
    with open(osu_path, "r", encoding="utf-8") as f:
        lines = f.read().split("\n")
    meta = BeatmapMeta(path=osu_path)
    hit_objects: List[str] = []
    section = ""
    for raw in lines:
        line = raw.strip()
        is_record = "," in line
        if section == "[HitObjects]" and is_record:
            hit_objects.append(line)
        elif section == "[TimingPoints]" and is_record:
            meta.file_meta.append(line)
            meta.timing_points.append(line)
        else:
            if line != "[HitObjects]":
                meta.file_meta.append(line)
            if section == "[General]":
                if line.startswith("AudioFilename"):
                    meta.audio = _resolve_audio(osu_path, read_item(line))
                elif line.startswith("Mode"):
                    meta.game_mode = int(read_item(line))
                    if convertor_params is not None:
                        meta.convertor = MOD_CONVERTOR[meta.game_mode](**convertor_params)
            elif section == "[Metadata]":
                if line.startswith("Version"):
                    meta.version = read_item(line)
                elif line.startswith("BeatmapSetID"):
                    meta.set_id = int(read_item(line))
            elif section == "[Difficulty]":
                if line.startswith("CircleSize"):
                    meta.cs = float(read_item(line))
        if line.startswith("["):
            section = line
    return hit_objects, meta


def save_osu_file(meta: BeatmapMeta, note_array: np.ndarray, path=None, override=None, gridify=None):
    """Writes the chart: the template's non-hit-object lines (with `override` replacing `Key:` lines), one red timing
    line from `gridify` (bpm / offset fit + snapping of the hit objects), then the hit objects (:94-120)."""
    hit_objects = meta.convertor.array_to_objects(note_array, meta)
    try:
        bpm, offset, hit_objects = gridify(hit_objects)
    except Exception:      # includes gridify=None, like the reference: fall back to 120 bpm / offset 0, objects unsnapped
        traceback.print_exc()
        bpm, offset = 120, 0
    with open(path, "w", encoding="utf8") as f:
        for line in meta.file_meta:
            if override is not None:
                for key, value in override.items():
                    if line.startswith(key + ":"):
                        line = f"{key}: {value}"
                        break
            f.write(line + "\n")
        if gridify is not None:
            f.write(f"[TimingPoints]\n{offset},{60000 / bpm},4,2,1,20,1,0\n\n")
        f.write("[HitObjects]\n")
        for obj in hit_objects:
            f.write(obj + "\n")


class BaseOsuConvertor:
    def __init__(self, frame_ms, max_frame, mirror=False, from_logits=False, offset_ms=0, random=False, rate=1.0,
                 mirror_at_interval_prob=0.0):
        self.frame_ms = frame_ms
        self.max_frame = max_frame
        self.mirror = mirror
        self.from_logits = from_logits
        self.offset_ms = offset_ms
        self.random = random
        self.rate = rate
        self.mirror_at_interval_prob = mirror_at_interval_prob


class OsuManiaConvertor(BaseOsuConvertor):
    def is_binary_positive(self, input):
        return input > 0 if self.from_logits else input > 0.5

    def note_grid(self, note_array, key_count=4):
        """Boolean (is_start, is_holding) grids -- the bit-exact parity target of the sampler."""
        a = np.asarray(note_array)
        return self.is_binary_positive(a[..., 0:key_count, :]), self.is_binary_positive(a[..., 2 * key_count:3 * key_count, :])

    def array_to_objects(self, note_array: np.ndarray, meta) -> List[str]:
        """Note grid -> hit-object lines (convertor.py:232-264), one vectorised pass per column: a note starts where
        channel `column` is positive; it is a long note if the frames right after it are 'holding' (channel column+2K)
        without a new start, and then ends at the last such frame; sub-frame offsets (channels +K / +3K, clipped to
        [0, 1]) refine both times.  Sorted by start time (stable, columns in order), like the reference."""
        a = np.asarray(note_array).transpose()
        key_count = int(meta.cs)
        column_width = int(512 / key_count)
        n = len(a)
        frames = np.arange(n)
        lines, starts_ms = [], []
        for column in range(key_count):
            is_start = self.is_binary_positive(a[:, column])
            s = np.nonzero(is_start)[0]
            if len(s) == 0:
                continue
            held = self.is_binary_positive(a[:, column + key_count * 2]) & ~is_start
            # first frame >= i that does not continue a hold (n if none): the hold that follows start s ends just before it
            stop = np.minimum.accumulate(np.where(held, n, frames)[::-1])[::-1]
            e = np.where(s < n - 1, stop[np.minimum(s + 1, n - 1)] - 1, s)
            start = np.round((s + np.clip(a[s, column + key_count], 0, 1)) * self.frame_ms).astype(np.int64)
            end = np.round((e + np.clip(a[e, column + key_count * 3], 0, 1)) * self.frame_ms).astype(np.int64)
            x = int(round((column + 0.5) * column_width))
            for k in range(len(s)):
                if e[k] == s[k]:
                    lines.append(f"{x},192,{start[k]},1,0,0:0:0:0:")
                else:
                    lines.append(f"{x},192,{start[k]},128,0,{end[k]}:0:0:0:0:")
            starts_ms.append(start)
        if not lines:
            return []
        order = np.argsort(np.concatenate(starts_ms), kind="stable")
        return [lines[k] for k in order]


MOD_CONVERTOR = {3: OsuManiaConvertor}

"""`.osu` template parsing / chart writing of the drop-in (mug/data/convertor.py) against the reference's own functions on
the same inputs: parsed fields equal, written files byte-identical.  Needs the reference checkout -> authoring container only;
the reference-free half (round trip of a written chart) runs everywhere."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from mug.data import convertor as mine

REF = "/root/reference/mug/data/convertor.py"

TEMPLATE = """osu file format v14

[General]
AudioFilename: audio.mp3
AudioLeadIn: 0
Mode: 3

[Metadata]
Title:Some / Song [x]
Version:Insane: 4K
BeatmapSetID:1234

[Difficulty]
HPDrainRate:8
CircleSize:4
OverallDifficulty:8

[TimingPoints]
24,333.333,4,2,1,40,1,0
5000,-50,4,2,1,40,0,0

[HitObjects]
64,192,1000,1,0,0:0:0:0:
192,192,1500,128,0,2000:0:0:0:0:
"""


def _reference():
    spec = importlib.util.spec_from_file_location("ref_convertor_for_test", REF)
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m          # dataclasses resolve string annotations through sys.modules
    spec.loader.exec_module(m)
    return m


def _logits(seed, T=96):
    g = np.random.default_rng(seed)
    a = g.normal(-1.2, 1.5, (16, T)).astype(np.float32)
    a[4:8] = g.random((4, T))
    a[12:16] = g.random((4, T)) * 1.4 - 0.2
    a[8:12] = np.where(g.random((4, T)) > 0.6, 1.0, -1.0) + a[0:4].clip(-0.5, 0.5)
    return a


def _gridify(objs):
    return 181.5, 37, [o.replace(",192,", ",192,") for o in objs][::-1][::-1]


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (authoring container only)")
@pytest.mark.parametrize("use_gridify", [True, False])
def test_parse_and_save_match_the_reference(tmp_path, use_gridify):
    ref = _reference()
    osu = tmp_path / "Template File.osu"
    osu.write_text(TEMPLATE, encoding="utf-8")
    (tmp_path / "audio.mp3").write_bytes(b"x")
    params = dict(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=True)
    ho_r, meta_r = ref.parse_osu_file(str(osu), dict(params))
    ho_m, meta_m = mine.parse_osu_file(str(osu), dict(params))
    assert ho_m == ho_r
    for f in ("path", "audio", "game_mode", "cs", "version", "set_id", "file_meta", "timing_points"):
        assert getattr(meta_m, f) == getattr(meta_r, f), f
    assert meta_m.for_batch() == meta_r.for_batch()
    for seed in (0, 1):
        a = _logits(seed)
        out_r, out_m = tmp_path / ("r%d.osu" % seed), tmp_path / ("m%d.osu" % seed)
        override = {"Version": "AI v%d" % seed, "Title": "t"}
        ref.save_osu_file(meta_r, a, path=str(out_r), override=override, gridify=_gridify if use_gridify else None)
        mine.save_osu_file(meta_m, a, path=str(out_m), override=override, gridify=_gridify if use_gridify else None)
        assert out_m.read_bytes() == out_r.read_bytes()
        assert len(out_m.read_text().split("[HitObjects]")[1].strip().splitlines()) > 5


def test_audio_fallbacks_and_roundtrip(tmp_path):
    osu = tmp_path / "t.osu"
    osu.write_text(TEMPLATE.replace("audio.mp3", "Au#dio.MP3"), encoding="utf-8")
    (tmp_path / "audio.mp3").write_bytes(b"x")            # only reachable through lower() + slugify()
    _, meta = mine.parse_osu_file(str(osu), dict(frame_ms=46.44, max_frame=4096, from_logits=True))
    assert meta.audio == str(tmp_path / "audio.mp3")
    assert meta.game_mode == 3 and meta.cs == 4.0 and meta.version == "4K" and meta.set_id == 1234
    out = tmp_path / "o.osu"
    mine.save_osu_file(meta, _logits(3), path=str(out), override={"Version": "v"}, gridify=_gridify)
    ho, meta2 = mine.parse_osu_file(str(out), None)
    assert meta2.version == "v" and meta2.convertor is None
    assert ho == _gridify(meta.convertor.array_to_objects(_logits(3), meta))[2]
    assert meta2.timing_points[-1].startswith("37,")


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (authoring container only)")
@pytest.mark.parametrize("from_logits", [True, False])
def test_array_to_objects_matches_the_reference_on_long_and_edgy_grids(from_logits):
    """The vectorised note extraction against the reference's per-note loops (convertor.py:232-264): full-length grids,
    holds that run into the last frame, a start in the last frame, back-to-back starts inside a hold, empty columns."""
    ref = _reference()
    params = dict(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=from_logits)
    r, m = ref.OsuManiaConvertor(**params), mine.OsuManiaConvertor(**params)

    class Meta:
        cs = 4

    grids = [_logits(seed, T) for seed, T in ((5, 4096), (6, 1000), (7, 1), (8, 2))]
    edge = _logits(9, 64)
    hi, lo = (3.0, -3.0) if from_logits else (0.9, 0.1)
    edge[0:4] = lo
    edge[8:12] = lo
    edge[0, [3, 10, 11, 40, 63]] = hi          # column 0: starts, one in the last frame
    edge[8, 4:10] = hi                          # ... a hold cut by the next start
    edge[8, 41:64] = hi                         # ... a hold running into the last frame
    edge[9, :] = hi                             # column 1: holding everywhere but never started
    edge[2, 62] = hi
    edge[10, 63] = hi                           # column 2: a 1-frame hold ending in the last frame
    grids.append(edge)
    grids.append(np.full((16, 32), lo, dtype=np.float32))       # nothing at all
    if not from_logits:
        grids = [1 / (1 + np.exp(-g)) if g is not edge and g.min() < 0 else g for g in grids]
    total = 0
    for g in grids:
        want = r.array_to_objects(g, Meta())
        assert m.array_to_objects(g, Meta()) == want
        total += len(want)
    assert total > 1500

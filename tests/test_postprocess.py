"""Chart post-processing: `gridify` (BPM / offset fit + snapping) and mini-jack removal of the
drop-in (mug/data/utils.py -> mugd_timing_sweep + mugd_remove_mini_jacks) against

  * tests/golden/postprocess_golden.json.gz: inputs and outputs of the REAL reference functions (mug/data/utils.py of the
    reference, run by oracle/gen_golden.py in the authoring container) -- strings equal, bpm / offset bit-equal;
  * oracle/postprocess.py (the CPU restatement, itself pinned to the same goldens) on further seeded charts.

The `lib` fixture runs everything twice: on the CPU emulation build here, on the real HIP build under `-m gpu`."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest
import torch

from mug.data import utils as product
from oracle import postprocess as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess_golden.json.gz")


def charts():
    with gzip.open(GOLDEN, "rt", encoding="utf8") as f:
        return json.load(f)["charts"]


CHARTS = charts()
IDS = [str(c["spec"]).replace(" ", "") for c in CHARTS]
EMU_MAX_NOTES = 3000            # (every golden chart fits; the guard stays for longer ones)


def hexes(bpm, offset):
    return float(bpm).hex(), float(offset).hex(), type(offset).__name__


# ------------------------------------------------------------------------------------------------ the oracle's pin
@pytest.mark.parametrize("chart", CHARTS, ids=IDS)
def test_oracle_restatement_matches_the_reference(chart):
    snapped, bpm, offset = oracle.gridify(chart["objects"])
    assert snapped == chart["snapped"]
    assert hexes(bpm, offset) == (chart["bpm"], chart["offset"], chart["offset_dtype"])
    for name, want in chart["jacks"].items():
        src = chart["snapped"] if name.startswith("after_snap") else chart["objects"]
        assert oracle.remove_mini_jacks(src, int(name.rsplit("_", 1)[1])) == want, name


# ------------------------------------------------------------------------------------------------ product vs reference
def skip_if_slow(lib, n_notes):
    if lib.device.type == "cpu" and n_notes > EMU_MAX_NOTES:
        pytest.skip("chart too long for the CPU emulation of the sweep kernel; runs under -m gpu")


@pytest.mark.parametrize("chart", CHARTS, ids=IDS)
def test_gridify_matches_the_reference(lib, chart):
    skip_if_slow(lib, len(chart["objects"]))
    snapped, bpm, offset = product.gridify(chart["objects"], verbose=False, lib=lib)
    assert hexes(bpm, offset) == (chart["bpm"], chart["offset"], chart["offset_dtype"])
    assert snapped == chart["snapped"]


@pytest.mark.parametrize("chart", CHARTS, ids=IDS)
def test_mini_jack_removal_matches_the_reference(lib, chart):
    for name, want in chart["jacks"].items():
        src = chart["snapped"] if name.startswith("after_snap") else chart["objects"]
        got = product.remove_intractable_mania_mini_jacks(src, verbose=False, jack_interval=int(name.rsplit("_", 1)[1]), lib=lib)
        assert got == want, name


def test_goldens_exercise_moves_and_removals():
    """The fixtures are not vacuous: the reference both moved and removed notes in them."""
    removed = moved = 0
    for ch in CHARTS:
        want = ch["jacks"]["raw_90"]
        removed += len(ch["objects"]) - len(want)
        pool = {}
        for o in ch["objects"]:
            pool.setdefault(o.split(",", 1)[1], []).append(o.split(",", 1)[0])
        moved += sum(1 for w in want if w.split(",", 1)[0] not in pool[w.split(",", 1)[1]])
    assert removed > 50 and moved > 50, (removed, moved)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_product_matches_oracle_on_other_charts(lib, seed):
    g = np.random.default_rng(seed)
    objs = oracle.synthetic_chart(seed, beats=int(g.integers(30, 90)), bpm=float(g.uniform(140, 310)),
                                  offset=float(g.uniform(0, 3000)), jitter=float(g.uniform(0, 9)), ln_p=float(g.uniform(0, 0.5)))
    want, w_bpm, w_off = oracle.gridify(objs)
    got, bpm, off = product.gridify(objs, verbose=False, lib=lib)
    assert hexes(bpm, off) == hexes(w_bpm, w_off) and got == want
    for interval in (60, 90, 200):
        assert product.remove_intractable_mania_mini_jacks(got, False, interval, lib=lib) == oracle.remove_mini_jacks(want, interval)


@pytest.mark.gpu
def test_full_length_chart_on_the_gpu(gpu_lib):
    """A 3-minute-sized chart (2500 notes) end to end, and the point of the exercise: time per chart."""
    import time
    chart = max(CHARTS, key=lambda c: len(c["objects"]))
    product.gridify(chart["objects"], verbose=False, lib=gpu_lib)          # warm-up (sklearn import, first launch)
    t0 = time.perf_counter()
    snapped, bpm, offset = product.gridify(chart["objects"], verbose=False, lib=gpu_lib)
    t1 = time.perf_counter()
    out = product.remove_intractable_mania_mini_jacks(snapped, verbose=False, lib=gpu_lib)
    t2 = time.perf_counter()
    assert snapped == chart["snapped"] and out == chart["jacks"]["after_snap_90"]
    assert hexes(bpm, offset) == (chart["bpm"], chart["offset"], chart["offset_dtype"])
    print("\n%d notes: gridify %.1f ms, mini-jacks %.2f ms" % (len(snapped), (t1 - t0) * 1e3, (t2 - t1) * 1e3))


# ------------------------------------------------------------------------------------------------ pieces
def test_sweep_kernel_counts_are_numpys(lib):
    g = np.random.default_rng(5)
    times = np.sort(g.uniform(0, 180000, 777)).round().astype(np.float32)
    n_cand = 203
    bpm = g.uniform(150, 300, n_cand)
    div = g.choice([1, 2, 3, 4, 6, 16], n_cand)
    gap = 60 * 1000 / (bpm * div)
    offset = g.uniform(-500, 3000, n_cand)
    f32 = g.random(n_cand) < 0.4
    offset[f32] = offset[f32].astype(np.float32)
    offset[:3] = times[0]                    # a note exactly on the origin
    got = lib.timing_sweep(torch.from_numpy(times).to(lib.device), gap, offset, f32, product.epsilon)
    for c in range(n_cand):
        off = np.float32(offset[c]) if f32[c] else np.float64(offset[c])
        valid, _ = product._score_on_host(times, np.float64(bpm[c]), off, div[c])
        assert got[c] == valid.sum(), c
        assert np.array_equal(valid, oracle.candidate_counts(times, np.float64(bpm[c]), off, div[c])[0])


def test_quarter_shifts_are_numpy_aranges():
    g = np.random.default_rng(6)
    gaps = 60000 / np.arange(150, 300, 0.1)
    lengths = set()
    for k in range(40):
        start = g.uniform(-4000, 200000) if k % 4 else np.float32(g.uniform(0, 5000))
        values, length = product._quarter_shifts(start, gaps)
        for r in range(0, len(gaps), 7):
            want = np.arange(start, start - gaps[r], -gaps[r] / 4)
            assert length[r] == len(want)
            assert np.array_equal(values[r, :length[r]], want)
            lengths.add(len(want))
    assert lengths == {4, 5}              # both roundings of the stop value occur: the masking in the sweep matters


def test_snap_matches_the_scalar_rule():
    g = np.random.default_rng(7)
    times = g.integers(0, 200000, 500)
    for bpm, offset in ((np.float64(187.31), np.float64(412.7)), (np.float64(150.0), np.float32(1000.0)),
                        (np.float64(299.9), np.float64(-3.25))):
        want = [oracle.snap_time(int(t), bpm, offset) for t in times]
        assert [str(int(v)) for v in product._snap(times, bpm, offset)] == want


def test_mini_jack_abi_rejects_bad_arguments(lib):
    f = lib.dll.mugd_remove_mini_jacks
    one = (C.c_double * 1)(1.0)
    col = (C.c_int32 * 1)(0)
    out = (C.c_int32 * 1)(0)
    keep = (C.c_uint8 * 1)(0)
    p = lambda a: C.cast(a, C.c_void_p)
    assert f(1, p(one), p(col), p(one), 90.0, 128, p(out), p(keep)) == 0 and keep[0] == 1
    assert f(1, None, p(col), p(one), 90.0, 128, p(out), p(keep)) < 0
    assert f(1, p(one), p(col), p(one), 90.0, 0, p(out), p(keep)) < 0
    assert f(-1, p(one), p(col), p(one), 90.0, 128, p(out), p(keep)) < 0
    assert f(0, None, None, None, 90.0, 128, None, None) == 0
    with pytest.raises(Exception):
        lib.timing_sweep(torch.zeros(0, dtype=torch.float32, device=lib.device), np.ones(1), np.ones(1), np.zeros(1, bool))


def test_gridify_of_an_empty_chart_raises_like_the_reference(lib):
    with pytest.raises(IndexError):
        product.gridify([], verbose=False, lib=lib)


# ------------------------------------------------------------------------------------------------ whole chart files
def golden_files():
    with gzip.open(GOLDEN, "rt", encoding="utf8") as f:
        return json.load(f)["files"]


@pytest.mark.parametrize("gold", golden_files(), ids=lambda g: "frames%d" % g["frames"])
def test_chart_file_is_byte_identical_to_the_reference(lib, tmp_path, gold):
    """Note grid -> .osu through the drop-in only (parse_osu_file, array_to_objects, gridify, mini-jacks, save_osu_file,
    wired like webui.py:401-407,431-445) == the file the real reference wrote for the same grid."""
    from mug.data import convertor
    osu = tmp_path / "template.osu"
    osu.write_text(oracle.TEMPLATE_OSU, encoding="utf8")
    (tmp_path / "audio.mp3").write_bytes(b"")
    _, meta = convertor.parse_osu_file(str(osu), dict(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=True))

    def ui_gridify(objs):
        snapped, bpm, offset = product.gridify(objs, verbose=False, lib=lib)
        return bpm, offset, product.remove_intractable_mania_mini_jacks(snapped, verbose=False, jack_interval=90, lib=lib)

    out = tmp_path / "out.osu"
    convertor.save_osu_file(meta, oracle.synthetic_note_grid(gold["seed"], gold["frames"]), path=str(out),
                            override={"Version": "AI v%d" % gold["seed"], "Creator": "golden"}, gridify=ui_gridify)
    assert out.read_text(encoding="utf8") == gold["text"]

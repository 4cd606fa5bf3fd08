"""Chart post-processing of the drop-in: BPM / offset fit + snapping (`gridify`) and mini-jack removal, with the
reference's interface (`mug/data/utils.py:106-139` and `:140-255`; callers `webui.py:401-407`, `scripts/mapping.py:496-498`)
and byte-identical results (tests/test_postprocess.py against outputs of the real reference).

What runs where:
  * the ~7500-candidate sweep of `timing()` (utils.py:46-97) is one HIP launch per "best so far" epoch
    (`mugd_timing_sweep`, csrc/k_timing.hip): all remaining (tempo, offset) candidates are scored speculatively under the
    current best; rows of the sweep that contain no improving candidate cannot change the state and are skipped, the
    first row that does is replayed on the host exactly as the reference orders it;
  * the refit of an improving candidate is scikit-learn's weighted `LinearRegression`, as in the reference (its LAPACK
    result is only reproducible by calling it; ~10 calls per chart);
  * snapping of the note times is vectorised NumPy with the reference's scalar arithmetic;
  * the mini-jack pass is sequential and data-dependent: host C++ behind `mugd_remove_mini_jacks`.
There is no CPU fallback for the sweep: without libmugd.so / a GPU `gridify` raises like the rest of the package.
"""
import numpy as np
import torch

epsilon = 10
_DIVISORS = (1, 2, 4, 3, 6, 8, 16, 32)


def parse_hit_objects(line, column_width):
    """-> (start_ms, column, end_ms or None); (None, None, None) for a deleted line (utils.py:7-13)."""
    if line is None:
        return None, None, None
    fields = line.split(",")
    end = float(fields[5].split(":")[0]) if int(fields[3]) == 128 else None
    return float(fields[2]), int(int(float(fields[0])) / column_width), end


def _score_on_host(time_list, bpm, offset, div=1):
    gap = 60 * 1000 / (bpm * div)
    meter = (time_list - offset) / gap
    meter_round = np.round(meter)
    valid = (np.abs(meter - meter_round) < epsilon / gap).astype(np.int32)
    return valid, meter_round


def _refit(time_list, meter_round, valid, bpm, offset):
    """Weighted least squares of note time on grid index (utils.py:29-40); tempo folded into [150, 300)."""
    if np.sum(valid) < 2:
        return bpm, offset
    from sklearn.linear_model import LinearRegression
    fit = LinearRegression(fit_intercept=True)
    fit.fit(meter_round.reshape((-1, 1)), time_list, sample_weight=valid)
    slope = fit.coef_
    if np.isinf(slope) or np.isnan(slope) or slope[0] == 0:
        return bpm, offset
    tempo = 60000 / slope[0] / 4
    while tempo < 150:
        tempo = tempo * 2
    while tempo >= 300:
        tempo = tempo / 2
    return tempo, fit.intercept_


def test_timing(time_list, test_bpm, test_offset, div, refine):
    """One candidate on the host, reference signature (utils.py:16-43) -> (valid_ratio, valid, bpm, offset)."""
    valid, meter_round = _score_on_host(time_list, test_bpm, test_offset, div)
    bpm, offset = _refit(time_list, meter_round, valid, test_bpm, test_offset) if refine else (test_bpm, test_offset)
    return np.sum(valid) / test_bpm, valid, bpm, offset


test_timing.__test__ = False        # not a pytest test, whatever module imports it


def _quarter_shifts(start, gaps):
    """np.arange(start, start - gap, -gap / 4) for every gap at once, element for element what NumPy builds:
    length ceil((stop - start) / step), values start + j * ((start + step) - start).  -> (values (n, 5), lengths)."""
    start = np.float64(start)
    step = -gaps / 4
    length = np.ceil(((start - gaps) - start) / step).astype(np.int64)
    delta = (start + step) - start
    values = start + np.arange(max(2, int(length.max())), dtype=np.float64)[None, :] * delta[:, None]
    values[:, 0] = start
    values[:, 1] = start + step
    return values, length


class _Sweep:
    """Scores of every not-yet-visited candidate under the current best offset, through one native launch."""

    def __init__(self, lib, time_list):
        self.lib = lib
        self.first = time_list[0]
        self.tempi = np.arange(150, 300, 0.1)
        self.gaps = 60000 / self.tempi
        self.times = torch.from_numpy(np.ascontiguousarray(time_list)).to(lib.device)

    def first_improving_row(self, row, best_offset, best_score):
        """Smallest row index >= `row` holding a candidate whose valid / bpm beats `best_score`, or None."""
        tempi, gaps = self.tempi[row:], self.gaps[row:]
        shifts, length = _quarter_shifts(best_offset, gaps)
        width = shifts.shape[1]
        gap = np.repeat(gaps, 1 + width)
        offset = np.concatenate([np.full((len(gaps), 1), np.float64(self.first)), shifts], axis=1).reshape(-1)
        is_f32 = np.zeros((len(gaps), 1 + width), dtype=bool)
        is_f32[:, 0] = True
        counts = self.lib.timing_sweep(self.times, gap, offset, is_f32.reshape(-1), epsilon).reshape(len(gaps), 1 + width)
        live = np.ones_like(is_f32)
        live[:, 1:] = np.arange(width)[None, :] < length[:, None]
        better = ((counts / tempi[:, None]) > best_score) & live
        rows = np.nonzero(better.any(axis=1))[0]
        return None if len(rows) == 0 else row + int(rows[0])


def timing(time_list, verbose=True, lib=None):
    """Best (bpm, offset) for the note times (float32 array, ms): utils.py:46-97."""
    if lib is None:
        from mug._native import get_lib
        lib = get_lib()
    time_list = np.ascontiguousarray(time_list, dtype=np.float32)
    sweep = _Sweep(lib, time_list)
    first = time_list[0]
    best_bpm, best_offset, best_score = None, None, -1
    refits = 0
    row = 0
    while row is not None and row < len(sweep.tempi):
        # replay one row in the reference's order: the tempo at the first note's offset, then the quarter-beat shifts
        tempo = sweep.tempi[row]
        valid, meter_round = _score_on_host(time_list, tempo, first)
        if np.sum(valid) / tempo > best_score:
            best_score = np.sum(valid) / tempo
            best_bpm, best_offset = _refit(time_list, meter_round, valid, tempo, first)
            tempo = best_bpm
            refits += 1
        gap = 60000 / tempo
        for shifted in np.arange(best_offset, best_offset - gap, -gap / 4):
            valid, meter_round = _score_on_host(time_list, tempo, shifted)
            score = np.sum(valid) / tempo
            if score > best_score:
                best_score = score
                tempo, best_offset = _refit(time_list, meter_round, valid, tempo, shifted)
                best_bpm = tempo
                refits += 1
        row += 1
        if row < len(sweep.tempi):
            row = sweep.first_improving_row(row, best_offset, best_score)
    if verbose:
        print(f"Final bpm: {best_bpm}, offset: {best_offset} ({refits} refits, score {best_score})")
    return best_bpm, best_offset


def _snap(times, bpm, offset):
    """`format_time` (utils.py:123-131) for an int64 array: the first divisor whose grid has a line within `epsilon`
    ms wins; unsnappable times stay.  Python's `int - np.float32` is a float32 subtraction (degenerate charts whose
    offset never left the float32 first-note time), anything else is float64."""
    times = np.asarray(times, dtype=np.int64)
    out = times.copy()
    todo = np.ones(len(times), dtype=bool)
    as_float = times.astype(np.float32) if isinstance(offset, np.float32) else times.astype(np.float64)
    for div in _DIVISORS:
        gap = 60 * 1000 / (bpm * div)
        meter = (as_float - offset) / gap
        line = np.round(meter)
        hit = todo & (np.abs(meter - line) < epsilon / gap)
        out[hit] = (line[hit] * gap + offset).astype(np.int64)
        todo &= ~hit
    return out


def gridify(hit_objects, verbose=True, lib=None):
    """-> (hit objects with start / long-note end times snapped to the fitted grid, bpm, offset): utils.py:106-139."""
    fields = [line.split(",") for line in hit_objects]
    starts = np.asarray([float(f[2]) for f in fields], dtype=np.float32)
    bpm, offset = timing(starts, verbose, lib)
    snapped = _snap([int(f[2]) for f in fields], bpm, offset)
    long_notes = [i for i, f in enumerate(fields) if int(f[3]) == 128]
    tails = {i: fields[i][5].split(":") for i in long_notes}
    ends = _snap([int(tails[i][0]) for i in long_notes], bpm, offset)
    for i, f in enumerate(fields):
        f[2] = str(int(snapped[i]))
    for i, end in zip(long_notes, ends):
        tails[i][0] = str(int(end))
        fields[i][5] = ":".join(tails[i])
    return [",".join(f) for f in fields], bpm, offset


def remove_intractable_mania_mini_jacks(hit_objects, verbose=True, jack_interval=90, lib=None):
    """Moves or removes notes that repeat a column within `jack_interval` ms (utils.py:140-255).  The decisions are made
    by the native host pass on parsed arrays; this wrapper parses, and rewrites field 0 of moved notes."""
    if lib is None:
        from mug._native import get_lib
        lib = get_lib()
    column_width = int(512 / 4)                       # key_count is fixed to 4 in the reference too (utils.py:141)
    parsed = [parse_hit_objects(line, column_width) for line in hit_objects]
    start = [p[0] for p in parsed]
    column = [p[1] for p in parsed]
    end = [np.nan if p[2] is None else p[2] for p in parsed]
    new_x, keep = lib.remove_mini_jacks(start, column, end, jack_interval, column_width)
    moved = new_x != np.iinfo(np.int32).min
    out = []
    for i, line in enumerate(hit_objects):
        if not keep[i]:
            continue
        if moved[i]:
            f = line.split(",")
            f[0] = str(int(new_x[i]))
            line = ",".join(f)
        out.append(line)
    if verbose:
        print(f"mini-jacks: {int(moved.sum())} moved, {int((~keep).sum())} removed of {len(hit_objects)} notes")
    return out

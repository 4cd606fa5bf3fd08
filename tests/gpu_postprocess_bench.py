"""Dev tool (GPU box): time per chart of the drop-in's post-processing (mug/data/utils.py: HIP candidate sweep + host C++
mini-jack pass) next to the oracle's restatement of the reference algorithm (7500 NumPy-level calls) on the same chart.
    python tests/gpu_postprocess_bench.py --out gpurun_out/postprocess.json"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mug-diffusion_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

from mug._native import get_lib  # noqa: E402
from mug.data import convertor, utils as product  # noqa: E402
from oracle import postprocess as oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    lib = get_lib()
    rows = []
    for name, kw in (("3 min, 2500 notes", dict(seed=5, beats=700, bpm=174.0, offset=733.0, jitter=3.0)),
                     ("10 min, 8000 notes", dict(seed=6, beats=2300, bpm=174.0, offset=733.0, jitter=3.0)),
                     ("30 s, 500 notes", dict(seed=1, beats=160))):
        objs = oracle.synthetic_chart(**kw)
        product.gridify(objs, verbose=False, lib=lib)
        tg, tj = [], []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            snapped, bpm, off = product.gridify(objs, verbose=False, lib=lib)
            t1 = time.perf_counter()
            out = product.remove_intractable_mania_mini_jacks(snapped, verbose=False, lib=lib)
            t2 = time.perf_counter()
            tg.append(t1 - t0)
            tj.append(t2 - t1)
        t0 = time.perf_counter()
        o_snapped, o_bpm, o_off = oracle.gridify(objs)
        t1 = time.perf_counter()
        o_out = oracle.remove_mini_jacks(o_snapped, 90)
        t2 = time.perf_counter()
        same = (snapped == o_snapped and out == o_out and float(bpm).hex() == float(o_bpm).hex()
                and float(off).hex() == float(o_off).hex())
        rows.append(dict(chart=name, notes=len(objs), identical=same,
                         gridify_ms=statistics.median(tg) * 1e3, mini_jacks_ms=statistics.median(tj) * 1e3,
                         cpu_restatement_gridify_ms=(t1 - t0) * 1e3, cpu_restatement_mini_jacks_ms=(t2 - t1) * 1e3))
        print(rows[-1])
    # note extraction from a full-length grid, for the per-chart total
    g = np.random.default_rng(0)
    grid = g.normal(-2.0, 1.5, (16, 4096)).astype(np.float32)
    grid[4:8], grid[12:16] = g.random((4, 4096)), g.random((4, 4096))
    conv = convertor.OsuManiaConvertor(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=True)

    class Meta:
        cs = 4

    t0 = time.perf_counter()
    n = len(conv.array_to_objects(grid, Meta()))
    rows.append(dict(chart="array_to_objects 4096 frames", notes=n, ms=(time.perf_counter() - t0) * 1e3))
    print(rows[-1])
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()

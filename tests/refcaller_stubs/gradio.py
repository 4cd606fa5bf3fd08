"""TEST-ONLY stand-in for `gradio` (not installed here): webui.py's `startMapping` only touches gr.Progress (its .tqdm wrapper, also handed
to DDIMSampler.sample as tqdm_class), gr.Error and gr.update; the Blocks UI is built under `if __name__ == "__main__"` only.
Used by tests/test_mirror.py::test_reference_webui_start_mapping_runs_unchanged_on_the_drop_in."""


class Error(Exception):
    pass


def update(**kw):
    return dict(kw)


class Progress:
    """gr.Progress: called as progress.tqdm(iterable, desc=...) -- and, through DDIMSampler.sample(tqdm_class=progress.tqdm), as
    tqdm_class(iterator, desc=..., total=...).  Returns a plain iterable like Gradio's tracker does; counts what went through it."""

    def __init__(self, track_tqdm=False):
        self.calls = []

    def tqdm(self, iterable, desc=None, total=None, unit="steps", **kw):
        self.calls.append((desc, total))

        def gen():
            for x in iterable:
                yield x
        return gen()

class _Pic:
    height = 4000


class PlayField:
    def __init__(self, m=None, duration_per_px=5, padding=40):
        self.m = m

    def __add__(self, part):
        return self

    def export(self):
        return _Pic()

    def export_fold(self, max_height=None):
        return ("preview", self.m, max_height)

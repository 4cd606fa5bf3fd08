class _Part:
    pass


class PFDrawBpm(_Part): pass
class PFDrawBeatLines(_Part): pass
class PFDrawColumnLines(_Part): pass
class PFDrawNotes(_Part): pass
class PFDrawOffsets(_Part): pass

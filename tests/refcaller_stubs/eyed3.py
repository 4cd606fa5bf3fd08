"""TEST-ONLY stand-in for the `eyed3` tag reader (not installed here): scripts/mapping.py:417-419 only reads
`eyed3.load(path).tag.title / .artist`.  Used by tests/test_mirror.py::test_reference_cli_runs_unchanged_on_the_drop_in."""


class _Tag:
    title = "Test Song"
    artist = "Test Artist"


class _File:
    tag = _Tag()


def load(path):
    return _File()

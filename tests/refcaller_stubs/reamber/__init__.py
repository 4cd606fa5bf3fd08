"""TEST-ONLY stand-in for `reamber` (not installed here): webui.py:443-455 reads the chart it has just written back with OsuMap.read_file and
renders a preview picture with PlayField -- display only, nothing flows back into the chart."""

"""TEST-ONLY stand-in for the `minacalc` extension webui.py imports at module level (only the inversion tab uses it)."""

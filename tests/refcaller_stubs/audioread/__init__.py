"""TEST-ONLY stand-in for `audioread` (webui.py:29 imports audioread.ffdec to print a warning when ffmpeg is missing)."""

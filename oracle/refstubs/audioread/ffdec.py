class FFmpegAudioFile:
    pass

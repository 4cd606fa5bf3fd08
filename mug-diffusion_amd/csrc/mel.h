// log-mel front-end (the arithmetic of mug/util.py:138-143 after decode/resample).
#pragma once
#include "net.h"

void log_mel(Ctx* ctx, const float* pcm, long long n, int sr, int n_fft, int hop, int n_mels, float* out);

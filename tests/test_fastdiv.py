"""fastdiv() (csrc/common.h) decodes the conv_gemm workgroup index with a host-computed reciprocal.  The plain __umulhi
estimate is only exact while n * d < 2^32; long wave-encoder / VAE launches at large batch pass that range.
The function under test is compiled from the UNMODIFIED header (emulated HIP include path) into a small host program that
compares it with integer division around and far above the old bound."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include "common.h"
#include "kernels.h"
#include <cstdio>
#include <cstdint>
int main() {
    long long bad = 0, n_checked = 0;
    const unsigned ds[] = {1u, 2u, 3u, 7u, 32u, 255u, 3264u, 52224u, 65535u, 65536u, 65537u, 104448u, 1000003u, 16777259u, 0x7fffffffu};
    for (unsigned d : ds) {
        const unsigned m = conv_fastdiv_mul(d);
        uint64_t seed = 0x9e3779b97f4a7c15ull ^ d;
        for (int i = 0; i < 200000; ++i) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            unsigned n;
            if (i < 1000) n = 0x7fffffffu - (unsigned)i;                      // top of the int range
            else if (i < 3000) { const unsigned k = 1 + (unsigned)(seed >> 40) % (0x7fffffffu / d); n = k * d - (i & 1); }   // multiples of d and their predecessors
            else n = (unsigned)(seed >> 33);                                   // uniform 31-bit
            ++n_checked;
            if ((unsigned)fastdiv((int)n, m, (int)d) != n / d) { if (bad < 5) printf("n=%u d=%u got %d want %u\n", n, d, fastdiv((int)n, m, (int)d), n / d); ++bad; }
        }
    }
    printf("checked %lld, bad %lld\n", n_checked, bad);
    return bad ? 1 : 0;
}
"""


def test_fastdiv_is_exact_for_every_grid(tmp_path):
    src = tmp_path / "fd.cpp"
    src.write_text(SRC)
    exe = tmp_path / "fd"
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    subprocess.run([cxx, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "emu", "include"), "-I", os.path.join(ROOT, "mug-diffusion_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout

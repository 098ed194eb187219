"""csrc/lrnde_math.hpp on the host: the select form of tanh (the SDE step kernel's, no control flow) returns the bits of the
branching form (the one the oracle restates, oracle/lrnde_oracle.c) for every float it is given: every 61st float of the whole
range — all exponents, infinities, NaNs — and every float within 20 000 ulps of the piece boundaries."""
import os, subprocess, sys, textwrap
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include "lrnde_math.hpp"
    #include <cstdio>
    #include <cstring>
    #include <cstdint>
    int main() {
      unsigned long long bad = 0, n = 0;
      auto check = [&](float x) {
        const float a = lrnde::tanhf_c(x), b = lrnde::tanhf_sel(x);
        uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        const bool nan_both = (a != a) && (b != b);
        if (ua != ub && !nan_both) { if (bad < 5) printf("x=%a c=%a sel=%a\n", x, a, b); ++bad; }
        ++n;
      };
      // every 61st float of the whole range, then every float around the piece boundaries
      for (uint64_t u = 0; u < (1ull << 32); u += 61) { uint32_t v = (uint32_t)u; float x; memcpy(&x, &v, 4); check(x); }
      const float edges[] = {0.625f, 9.0f, 43.5f, 87.0f, 0.0f, 1.0f};
      for (float e : edges) {
        uint32_t v; memcpy(&v, &e, 4);
        for (int d = -20000; d <= 20000; ++d) { uint32_t w = v + (uint32_t)d; float x; memcpy(&x, &w, 4); check(x); check(-x); }
      }
      printf("checked %llu mismatches %llu\n", n, bad);
      return bad ? 1 : 0;
    }
''')


def test_select_form_tanh_has_the_branching_forms_bits(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "localregneuralde.jl_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout, r.stdout

"""The host-side chip generators (csrc/prn_codes.cpp: LFSR/Weil/memory-code tables, the replica sampler) rebuilt with
AddressSanitizer + UBSan and driven over every code, every PRN and the replica shapes the signals use.  A clean exit means
no out-of-bounds table access, no signed overflow and no misaligned read anywhere in the generators."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gnss-dsp-tools_amd", "csrc")

DRIVER = r'''
#include <cstdio>
#include <cstdint>
#include <vector>
#include "gacq.h"
int main() {
  long total = 0;
  const int ncodes = gacq_code_count();
  for (int c = 0; c < ncodes; c++) {
    const char* name = gacq_code_name(c);
    const int L = gacq_code_length(name);
    if (L <= 0) { std::printf("bad length for %s\n", name); return 2; }
    std::vector<int> prns(4096);
    const int np = gacq_code_prns(name, prns.data(), (int)prns.size());
    if (np <= 0) { std::printf("no prns for %s\n", name); return 3; }
    std::vector<uint8_t> chips(L);
    for (int k = 0; k < np; k++) {
      if (gacq_code_chips(name, prns[k], chips.data(), L) < 0) { std::printf("chips failed %s %d\n", name, prns[k]); return 4; }
      for (int i = 0; i < L; i++) total += chips[i];
    }
    // replica sampler at a non-integer samples-per-chip ratio, with and without BOC, first and last PRN
    const int n = 2 * L + 37;
    std::vector<float> rep(n);
    for (int boc = 0; boc < 2; boc++)
      for (int k : {0, np - 1})
        if (gacq_code_replica(name, prns[k], n, boc, rep.data()) < 0) { std::printf("replica failed %s\n", name); return 5; }
    // out-of-table requests must be refused, not read past the tables
    // (single-code families -- the GLONASS codes -- take no PRN argument in the reference and ignore it here too)
    if (np > 1 && gacq_code_chips(name, 100000, chips.data(), L) >= 0) { std::printf("accepted bad prn for %s\n", name); return 6; }
    if (gacq_code_chips(name, prns[0], chips.data(), L - 1) >= 0) { std::printf("accepted short buffer for %s\n", name); return 7; }
  }
  if (gacq_code_length("no.such.code") >= 0) return 8;
  std::printf("ok %d codes, chip sum %ld\n", ncodes, total);
  return 0;
}
'''


@pytest.mark.timeout(300)
def test_chip_generators_clean_under_asan_ubsan(tmp_path):
    drv = tmp_path / "driver.cpp"
    drv.write_text(DRIVER)
    exe = tmp_path / "prn_san"
    memcodes = os.path.join(ROOT, "gnss-dsp-tools_amd", "data", "memcodes.bin")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-DGACQ_MEMCODES_PATH=\"%s\"" % memcodes, "-I", os.path.join(ROOT, "include"),
           os.path.join(CSRC, "prn_codes.cpp"), str(drv), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=240,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert run.stdout.startswith("ok 32 codes")

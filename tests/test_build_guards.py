"""Guards on the compiled device code of lib/libgacq.so (no GPU needed): the code objects are pulled out of the library's .hip_fatbin
section and their kernel metadata is read with the ROCm LLVM tools.

* Kernels that carry the LDS-access separators (GACQ_UNPAIR, csrc/gacq_cplx.h: an empty asm statement between LDS accesses) must not
  use accumulator registers: this toolchain produced wrong rows in a kernel that spilled VGPRs to AGPRs around such statements
  (DESIGN 5.2).  The complex128 re-evaluation kernels (gacq_tiesafe.hip) are the only AGPR users and are compiled without separators.
* The hot kernels must not spill to scratch (a spill there is a performance cliff that no parity test would notice)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gnss-dsp-tools_amd", "lib", "libgacq.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_metadata(tmp_path):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or ROCm LLVM tools not present")
    fat = tmp_path / "fat.bin"
    subprocess.run([tools[0], "--dump-section", ".hip_fatbin=%s" % fat, LIB], check=True)
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    kernels = {}
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part = tmp_path / ("bundle%d.bin" % n)
        part.write_bytes(blob[a:b])
        co = tmp_path / ("code%d.elf" % n)
        subprocess.run([tools[1], "--unbundle", "--type=o", "--input=%s" % part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=%s" % co],
                       check=True, capture_output=True)
        if not co.exists() or co.stat().st_size == 0:
            continue
        notes = subprocess.run([tools[2], "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:                     # one block per kernel; .symbol is "<mangled name>.kd"
            fields = {"agpr_count": block.split()[0]}
            fields.update(re.findall(r"^\s+\.(symbol|vgpr_count|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)\s*$", block, flags=re.M))
            kernels[fields["symbol"]] = {k: int(fields[k]) for k in ("agpr_count", "vgpr_count", "private_segment_fixed_size", "vgpr_spill_count")}
    return kernels


def test_kernels_with_lds_separators_use_no_accumulator_registers_and_hot_kernels_do_not_spill(tmp_path):
    kernels = kernel_metadata(tmp_path)
    assert len(kernels) > 40, sorted(kernels)
    with_agprs = sorted(k for k, m in kernels.items() if m["agpr_count"])
    assert with_agprs and all("tie_recheck" in k for k in with_agprs), with_agprs
    for fragment in ("lds_fused4k_kernelILi4ELb1E", "lds16k_correlate_kernelILb0", "r32_correlate_kernelILb0", "r32_fused_kernelILb0", "fused4k_c128_kernel", "lds_inner_correlate_kernel",
                     "lds_correlate_kernelILi2E", "pfa_inner_corr_kernelILi1980ELi3E", "pfa_inner_corr_kernelILi990ELi2E",
                     "pfa_outer_inverse_kernelILi1980ELi0E", "pfa_outer_inverse_kernelILi990ELi2E", "pfa_outer_inverse_mfma_kernelILi1980ELi0E"):
        hit = [k for k in kernels if fragment in k]
        assert hit, fragment
        for k in hit:
            if fragment.startswith("r32_correlate_kernel"):
                # 256 registers: the record pointer (two dwords) is stored before the item loop and reloaded in the per-unit tail; nothing
                # is spilled inside the row loop (checked in the ISA: both reloads sit at loop depth 2, once per B rows)
                assert kernels[k]["private_segment_fixed_size"] <= 16 and kernels[k]["vgpr_spill_count"] <= 2, (k, kernels[k])
                continue
            assert kernels[k]["private_segment_fixed_size"] == 0 and kernels[k]["vgpr_spill_count"] == 0, (k, kernels[k])

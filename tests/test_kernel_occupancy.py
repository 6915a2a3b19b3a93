"""Waves per SIMD of the hot kernels, read from the BUILT library's code-object notes (no GPU, no recompilation): a floor per kernel.

Round 4 added the terminal-row side output to every rollout kernel; left alone, the register allocator then took 157 registers for
snake's SAME_STEP rollout instance (round 3: 116), 258 for parking's (255: two waves per SIMD became one and the whole-episode rollout
went from 12.7 to 21.4 us per step) and 129 for climate's (127) — and nothing noticed for most of the round.  The
floors below are the occupancies the measurements in DESIGN.md were taken at."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "custom_gymnasium_environments_amd", "libcge_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# mangled-name fragment -> minimum waves per SIMD (gfx950: 512 registers per SIMD lane, allocated in blocks of 8; arch + accumulation registers)
FLOORS = {
    "5snake14rollout_kernelILi10ELi256ELi1ELi1ELb0ELb0E": 4,      # the headline kernel: round 3's 116 registers
    "5snake14rollout_kernelILi10ELi256ELi1ELi1ELb0ELb1E": 3,      # with the terminal-row side output
    "5snake11step_kernelILi10ELi256ELi1ELi1E": 4,
    "7traffic11step_kernelILi9ELi3ELb1E": 4,
    "7traffic11step_kernelILi9ELi3ELb0E": 5,
    "7climate11step_kernelILb1E": 4,
    "7climate11step_kernelILb0E": 4,
    "7parking11step_kernelILb1E": 2,
    "7parking11step_kernelILb0E": 2,
    "4hosp11step_kernelILb1E": 2,
    "4hosp11step_kernelILb0E": 2,
    "3mfg11step_kernelILb1E": 2,
    "3mfg11step_kernelILb0E": 2,
    "6crypto15resident_kernelILb0E": 2,
    "6crypto15resident_kernelILb1E": 2,
    "5fleet11step_kernelILb1E": 4,
    "5fleet11step_kernelILb0E": 4,
}


def _kernels(tmp):
    lib = os.path.join(tmp, "lib.so")
    shutil.copy(LIB, lib)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], cwd=tmp, check=True, capture_output=True)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            m = re.search(r"\.name:\s+(\S+)", blk)
            a = re.match(r"\s*(\d+)", blk)
            v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            if m and a and v:
                out[m.group(1)] = (int(v.group(1)), int(a.group(1)))
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))), reason="needs the built library and the ROCm LLVM tools")
def test_hot_kernels_keep_their_waves_per_simd(tmp_path):
    ks = _kernels(str(tmp_path))
    assert len(ks) > 50, "no kernel notes found in the library"
    seen = set()
    for name, (vgpr, agpr) in ks.items():
        for frag, floor in FLOORS.items():
            if frag in name:
                seen.add(frag)
                total = -(-vgpr // 8) * 8 + -(-agpr // 8) * 8
                waves = min(8, 512 // max(total, 8))
                assert waves >= floor, f"{name}: {vgpr} + {agpr} registers = {waves} waves per SIMD, the floor is {floor}"
    assert seen == set(FLOORS), f"kernels not found in the library: {sorted(set(FLOORS) - seen)}"

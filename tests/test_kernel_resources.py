"""Register / scratch budget of the compiled gfx950 kernels, read from the code objects inside the built library (no GPU needed).
A refactoring of the exact-size kernel's prologue once cost config 2 fifteen per cent without failing any parity test: a struct
passed by reference landed in scratch memory and the register count crossed an occupancy step.  This pins what the measured
numbers rely on: no kernel uses scratch, and the exact-size kernels keep the occupancy their LDS slice allows."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dbcsr_amd", "libdbcsr_acc_amd.so")          # the shipping build
LAB = os.path.join(ROOT, "dbcsr_amd", "libdbcsr_acc_amd_lab.so")      # + the experimental dataflows (dbcsr_amd/csrc/Makefile)
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels_of_library(tmp_path, lib=LIB):
    objcopy, bundler, readelf = (os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    if not (os.path.exists(lib) and all(os.path.exists(t) for t in (objcopy, bundler, readelf))):
        pytest.skip("library or LLVM binutils not available")
    fat = tmp_path / "fat.bin"
    subprocess.check_call([objcopy, "--dump-section", ".hip_fatbin=%s" % fat, lib, str(tmp_path / "unused.so")])
    blob = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = {}
    for i, s in enumerate(starts):   # one bundle per translation unit
        part = tmp_path / ("bundle%d.bin" % i)
        part.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / ("gfx950_%d.co" % i)
        r = subprocess.run([bundler, "--type=o", "--unbundle", "--input=%s" % part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--output=%s" % co], capture_output=True, text=True)
        if r.returncode != 0 or not co.exists() or co.stat().st_size == 0:
            continue
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count|sgpr_count|group_segment_fixed_size):\s*(\S+)", line)
            if not m:
                if line.strip().startswith("- .") and cur.get("name"):
                    cur = {}
                continue
            k, v = m.groups()
            cur[k] = v if k == "name" else int(v)
            if "name" in cur and "private_segment_fixed_size" in cur and "vgpr_count" in cur:
                out[cur["name"]] = dict(cur)
    return out


def demangle(names):
    filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def test_no_scratch_and_exact_kernels_keep_their_occupancy(tmp_path):
    ks = kernels_of_library(tmp_path)
    assert len(ks) > 100, "expected the whole kernel set of the library, got %d" % len(ks)
    pretty = demangle(sorted(ks))
    spilled = {pretty[n]: k["private_segment_fixed_size"] for n, k in ks.items() if k["private_segment_fixed_size"] > 0}
    assert not spilled, "kernels using scratch memory: %s" % spilled
    hot64 = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_hot<" in pretty[n]}
    hot32 = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f32_hot<" in pretty[n]}
    assert len(hot64) >= 20 and len(hot32) >= 20
    # 512 registers per SIMD lane.  Blocks up to 24 x 24: 4 waves per SIMD (16 per CU, what a 9.5 KB LDS slice allows for 23 x 23)
    # need <= 128 (vgpr_count of the code object = arch + accumulation registers, allocation granule included); blocks of 25 ... 32
    # hold 16 accumulators and their 13-16 KB LDS slice allows 10-12 waves per CU anyway: 152-180 today (2-3 waves per SIMD)
    size_of = lambda n: int(re.search(r"hot<(\d+),", n).group(1))
    too_big = {n: v for n, v in hot64.items() if v > (128 if size_of(n) <= 24 else 184)}
    assert not too_big, too_big
    assert all(v <= 96 for v in hot32.values()), hot32
    # round 5.  The fp32 direct kernels: five waves per SIMD (<= 96 registers).  The workgroup-per-C-block kernels of the blocks of 33 ... 80:
    # 16 shapes, the largest (5 x 5 tiles per wave: 50 accumulator registers) within 3 waves per SIMD (<= 168), the smallest within 8.
    direct32 = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f32_direct<" in pretty[n]}
    assert len(direct32) == 3 and all(v <= 96 for v in direct32.values()), direct32
    big = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_big<" in pretty[n]}
    assert len(big) == 16 and all(v <= 168 for v in big.values()), big
    # round 6.  The one-wave slab kernels (mm_numeric_f64_mid: 6 ... 10 units of 4 x 4 per dimension, the larger at least 8; mm_mid.hip: mid_f64_serves): three waves per SIMD
    mid = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_mid<" in pretty[n]}
    assert len(mid) == 43 and all(v <= 168 for v in mid.values()), mid


def test_shipping_build_holds_no_experiment(tmp_path):
    """the library bench.py and the hosts load carries the production kernels only: no tile / band / LDS-DMA ring / persistent kernels and no
    ablation or keep-alive variant of the exact-size kernel (what ships must be auditable)"""
    ks = kernels_of_library(tmp_path)
    pretty = demangle(sorted(ks))
    lab_only = [v for v in pretty.values() if re.search(r"mm_numeric_f64_(tile|band|dma|hot_persistent|group)<", v) or "mm_numeric_f32_group<" in v or
                re.search(r"mm_numeric_f64_hot<\d+, \d+, \d+, [1-9]>", v) or re.search(r"\b(tile|band)_(lists|descs|remainder|select)", v)]
    assert not lab_only, lab_only
    assert sum("mm_numeric_f64_hot<" in v for v in pretty.values()) == 24
    # round 6: the EXPORT list too -- no diagnostics or launchers of the experimental dataflows
    nm = shutil.which("nm") or os.path.join(LLVM, "llvm-nm")
    exported = [l.split()[-1] for l in subprocess.run([nm, "-D", "--defined-only", LIB], capture_output=True, text=True).stdout.splitlines() if l.strip()]
    assert len(exported) > 60
    leftovers = [s for s in exported if re.search(r"tile|band|group|persistent|dma_", s, re.I)]
    assert not leftovers, leftovers


def test_lab_build_kernels_fit_their_occupancy(tmp_path):
    ks = kernels_of_library(tmp_path, LAB)
    pretty = demangle(sorted(ks))
    spilled = {pretty[n]: k["private_segment_fixed_size"] for n, k in ks.items() if k["private_segment_fixed_size"] > 0}
    assert not spilled, "kernels using scratch memory: %s" % spilled
    # the tile kernels (mm_tile.hip), no scratch (what -mllvm -structurizecfg-skip-uniform-regions is there for, see the Makefile):
    # shape 0 (8 waves per workgroup, two per SIMD) 81 accumulators per lane and everything else in 256 registers; shape 1 (4 waves per
    # workgroup, one per SIMD) 108 accumulators in the 512
    tile = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_tile<" in pretty[n]}
    wg_waves = lambda n: int(re.search(r"tile<\d+, \d+, \d+, \d+, \d+, \d+, (\d+),", n).group(1))
    assert tile and all(v <= (256 if wg_waves(n) == 8 else 512) for n, v in tile.items()), tile
    # the band kernels (mm_band.hip): 8 waves per workgroup (two per SIMD) in 256 registers, 16 waves (four per SIMD) in 128
    band = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_band<" in pretty[n]}
    band_waves = lambda n: int(re.search(r"band<\d+, \d+, \d+, \d+, \d+, (\d+),", n).group(1))
    assert band and all(v <= (256 if band_waves(n) == 8 else 128) for n, v in band.items()), band
    # the group kernels (mm_group.hip fp32, round 5; mm_group64.hip fp64, round 6): R accumulator sets in architectural registers, two waves per
    # SIMD (<= 256 registers) for every R
    group = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f32_group<" in pretty[n]}
    assert len(group) == 9 and all(v <= 256 for v in group.values()), group
    group64 = {pretty[n]: k["vgpr_count"] for n, k in ks.items() if "mm_numeric_f64_group<" in pretty[n]}
    assert len(group64) == 16 and all(v <= 256 for v in group64.values()), group64


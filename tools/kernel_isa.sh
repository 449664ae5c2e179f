#!/bin/bash
# tools/kernel_isa.sh <object or library> <out prefix>: the gfx950 code objects inside it -> <prefix>_N.co / .s, and a resource table
# (registers, scratch, LDS) of every kernel.  No GPU needed.
set -e
LLVM=/opt/rocm/lib/llvm/bin
in=$1; pre=${2:-/tmp/isa}
$LLVM/llvm-objcopy --dump-section .hip_fatbin=${pre}.fat "$in" ${pre}.unused
python3 - "$pre" <<'PY'
import re, subprocess, sys
pre = sys.argv[1]
blob = open(pre + ".fat", "rb").read()
starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
for i, s in enumerate(starts):
    part = "%s_b%d.bin" % (pre, i)
    open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
    co = "%s_%d.co" % (pre, i)
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + part,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
    if r.returncode:
        continue
    with open("%s_%d.s" % (pre, i), "w") as f:
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=f)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|vgpr_spill_count):\s*(\S+)", line)
        if m:
            cur[m.group(1)] = m.group(2)
        if line.strip().startswith("- .agpr_count") and cur.get("name"):
            pass
        if ".wavefront_size" in line and cur.get("name"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            print("%-90s vgpr %s agpr %s sgpr %s scratch %s spill %s" % (name[:90], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"),
                  cur.get("private_segment_fixed_size"), cur.get("vgpr_spill_count")))
            cur = {}
PY

#!/usr/bin/env python3
"""Builds the UNCHANGED reference Fortran host (DBCSR library + its test drivers) against this repo's
acc back end: SURVEY.md row f4.

Build-container tool.  /root/reference/src and /root/reference/tests are expanded with tools/fypp_lite.py
WHERE THEY LIE into a scratch directory (default /tmp/dbcsr_host -- never into the repo), compiled with
amdflang (serial MPI: __parallel undefined, the reference's own stubs in src/mpi/dbcsr_mpiwrap.F), and linked
  * variant "cpu":  no accelerator (reference CPU path; BLAS/LAPACK from /opt/conda/lib/libmkl_rt.so)
  * variant "acc":  -D__DBCSR_ACC against dbcsr_amd/libdbcsr_acc_amd.so
into oracle/_ref/host_<variant>/{dbcsr_perf,dbcsr_unittest1,dbcsr_unittest3,dbcsr_ref_dump} (git-ignored, travel to the GPU box).

    python tools/build_dbcsr_host.py [cpu|acc|both] [--scratch DIR] [--jobs N] [--patch FILE]...
"""
import argparse
import concurrent.futures as cf
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fypp_lite  # noqa: E402

SKIP_DIRS = ("tensors", "tas")
SKIP_FILES = ("dbcsr_api_c.F",)
FC = "/opt/rocm/bin/amdflang"
TEST_PROGRAMS = {
    "dbcsr_perf": ["dbcsr_performance_driver.F", "dbcsr_performance_multiply.F"],
    "dbcsr_unittest1": ["dbcsr_unittest1.F", "dbcsr_test_add.F", "dbcsr_test_multiply.F"],
    "dbcsr_unittest3": ["dbcsr_unittest3.F", "dbcsr_test_multiply.F"],
}


def expand_tree(scratch, patches):
    out = os.path.join(scratch, "expanded")
    if os.path.isdir(out):
        shutil.rmtree(out)
    files = []
    for sub in ("src", "tests"):
        base = os.path.join(REF, sub)
        for d, _, fs in os.walk(base):
            rel = os.path.relpath(d, base)
            if sub == "src" and rel.split(os.sep)[0] in SKIP_DIRS:
                continue
            for f in fs:
                if not (f.endswith(".F") or f.endswith(".f90")) or f in SKIP_FILES:
                    continue
                if sub == "tests" and not any(f in v for v in TEST_PROGRAMS.values()):
                    continue
                src = os.path.join(d, f)
                dst = os.path.join(out, sub, rel, f[:-2] + ".F90" if f.endswith(".F") else f)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                text = fypp_lite.Expander([os.path.join(REF, "src")]).expand_file(src)
                with open(dst, "w") as fh:
                    fh.write(text)
                if f.endswith(".F"):  # the .f90 files are #include fragments, not compilation units
                    files.append(dst)
    if patches:
        # the call-site change of INTEGRATION.md section 2: this repository's glue module next to the reference's mm sources,
        # then the unified diff(s) relative to the expanded tree
        glue = os.path.join(ROOT, "dbcsr_amd", "fortran", "dbcsr_amd_resident.F")
        dst = os.path.join(out, "src", "mm", "dbcsr_amd_resident.F90")
        shutil.copy(glue, dst)
        files.append(dst)
    for p in patches:
        subprocess.check_call(["patch", "-p1", "-d", out, "-i", os.path.abspath(p)])
    return out, files


MOD_RE = re.compile(r"^\s*module\s+(\w+)\s*$", re.I | re.M)
USE_RE = re.compile(r"^\s*use\s*(?:,\s*intrinsic\s*)?(?:::)?\s*(\w+)", re.I | re.M)
INTRINSIC = {"iso_c_binding", "iso_fortran_env", "omp_lib", "omp_lib_kinds", "mpi", "mpi_f08", "ieee_arithmetic", "ieee_exceptions"}


INC_RE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def read_with_includes(f, incdirs, depth=0):
    text = open(f, errors="replace").read()
    if depth < 4:
        for inc in INC_RE.findall(text):
            for d in [os.path.dirname(f)] + incdirs:
                p = os.path.join(d, inc)
                if os.path.exists(p):
                    text += "\n" + read_with_includes(p, incdirs, depth + 1)
                    break
    return text


def toposort(files, incdirs):
    prov, uses = {}, {}
    for f in files:
        text = read_with_includes(f, incdirs)
        for m in MOD_RE.findall(text):
            if m.lower() != "procedure":
                prov[m.lower()] = f
        uses[f] = {u.lower() for u in USE_RE.findall(text)} - INTRINSIC
    deps = {f: {prov[u] for u in uses[f] if u in prov and prov[u] != f} for f in files}
    levels, done = [], set()
    while len(done) < len(files):
        lvl = [f for f in files if f not in done and deps[f] <= done]
        if not lvl:
            raise SystemExit("dependency cycle among: %s" % [f for f in files if f not in done])
        levels.append(lvl)
        done.update(lvl)
    return levels


def build_variant(variant, scratch, exp, files, jobs, reuse=False):
    bdir = os.path.join(scratch, "build_" + variant)
    if not reuse:
        shutil.rmtree(bdir, ignore_errors=True)
    os.makedirs(bdir, exist_ok=True)
    flags = ["-cpp", "-O2", "-fopenmp", "-D__MKL", "-D__NO_STATM_ACCESS", "-J", bdir, "-I", bdir,
             "-I", os.path.join(exp, "src"), "-I", os.path.join(exp, "src", "base")]
    base = variant[:-4] if variant.endswith("_mpi") else variant
    if base in ("acc", "resident"):
        flags += ["-D__DBCSR_ACC"]
    if variant.endswith("_mpi"):
        # a REAL multi-rank build: -D__parallel switches the reference's MPI layer (src/mpi/dbcsr_mpiwrap.F) from its serial stubs to
        # MPI calls.  It says USE mpi; the image's mpi.mod was written by gfortran and amdflang cannot read it, but the image's
        # MPICH also ships the compiler-independent Fortran 77 interface (mpif.h: constants, COMMON blocks, external routines
        # with the f77 calling convention of libmpifort): a module `mpi` that just INCLUDEs it is compiled first.
        flags += ["-D__parallel", "-I", "/opt/conda/include"]
        mod_src = os.path.join(bdir, "mpi_from_mpif_h.F90")
        with open(mod_src, "w") as fh:
            fh.write("MODULE mpi\n   IMPLICIT NONE\n   INCLUDE 'mpif.h'\nEND MODULE mpi\n")
        subprocess.check_call([FC] + flags + ["-c", mod_src, "-o", os.path.join(bdir, "mpi_from_mpif_h.o")])
    lib_files = [f for f in files if os.sep + "src" + os.sep in f]
    test_files = [f for f in files if os.sep + "tests" + os.sep in f]

    def compile_one(f):
        o = os.path.join(bdir, os.path.basename(f)[:-4] + ".o")
        short = os.path.basename(f)
        if reuse and os.path.exists(o) and not dirty:
            return o
        rebuilt.append(o)
        r = subprocess.run([FC] + flags + ['-D__SHORT_FILE__="%s"' % short, "-c", f, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write("FAILED %s\n%s\n" % (f, r.stderr[-3000:]))
            raise SystemExit(1)
        return o

    objs = {}
    dirty, rebuilt = [], []   # --reuse: once one object had to be rebuilt, every later level is (its .mod files may have changed)
    for lvl in toposort(lib_files + test_files, [os.path.join(exp, "src"), os.path.join(exp, "src", "base")]):
        with cf.ThreadPoolExecutor(jobs) as ex:
            for f, o in zip(lvl, ex.map(compile_one, lvl)):
                objs[f] = o
        if rebuilt:
            dirty.append(True)
    outdir = os.path.join(ROOT, "oracle", "_ref", "host_" + variant)
    os.makedirs(outdir, exist_ok=True)
    lib_objs = [objs[f] for f in lib_files]
    link = ["-fopenmp", "-L/opt/conda/lib", "-lmkl_rt", "-Wl,-rpath,/opt/conda/lib"]
    if variant.endswith("_mpi"):
        link = [os.path.join(bdir, "mpi_from_mpif_h.o")] + link + ["-lmpifort", "-lmpi"]
    if base in ("acc", "resident"):
        link += ["-L" + os.path.join(ROOT, "dbcsr_amd"), "-ldbcsr_acc_amd", "-Wl,-rpath,$ORIGIN/../../../dbcsr_amd"]
    for prog, srcs in TEST_PROGRAMS.items():
        pobjs = [objs[f] for f in test_files if os.path.basename(f)[:-4] + ".F" in srcs]
        subprocess.check_call([FC] + pobjs + lib_objs + link + ["-o", os.path.join(outdir, prog)])
    # this repository's own fixture generator on top of the reference library (tests/fortran/dbcsr_ref_dump.F90)
    dump_src = os.path.join(ROOT, "tests", "fortran", "dbcsr_ref_dump.F90")
    dump_obj = os.path.join(bdir, "dbcsr_ref_dump.o")
    subprocess.check_call([FC] + flags + ["-c", dump_src, "-o", dump_obj])
    subprocess.check_call([FC, dump_obj] + lib_objs + link + ["-o", os.path.join(outdir, "dbcsr_ref_dump")])
    if base == "resident":
        # a host that keeps its matrices on the device across multiplies (dbcsr_amd_dev_* of the glue module)
        loop_src = os.path.join(ROOT, "tests", "fortran", "dbcsr_resident_loop.F90")
        loop_obj = os.path.join(bdir, "dbcsr_resident_loop.o")
        subprocess.check_call([FC] + flags + ["-c", loop_src, "-o", loop_obj])
        subprocess.check_call([FC, loop_obj] + lib_objs + link + ["-o", os.path.join(outdir, "dbcsr_resident_loop")])
    print("built", outdir, sorted(os.listdir(outdir)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variant", nargs="?", default="both", choices=["cpu", "acc", "both", "resident", "cpu_mpi", "acc_mpi", "resident_mpi"],
                    help="resident = acc + the call-site patch (dbcsr_multiply -> device-resident engine), output oracle/_ref/host_resident; "
                         "cpu_mpi / acc_mpi = the same library as a real multi-rank MPI build (MPICH of the image, mpiexec)")
    ap.add_argument("--scratch", default="/tmp/dbcsr_host")
    ap.add_argument("--jobs", type=int, default=16)
    ap.add_argument("--patch", action="append", default=[])
    ap.add_argument("--reuse", action="store_true", help="keep the objects of a previous run (only relink / rebuild the dump program)")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit("the reference is not mounted here: nothing to build (the GPU box uses the prebuilt oracle/_ref)")
    os.makedirs(a.scratch, exist_ok=True)
    if a.variant.startswith("resident") and not a.patch:
        a.patch = [os.path.join(ROOT, "dbcsr_amd", "fortran", "dbcsr_mm_call_site.patch")]
    if a.reuse and os.path.isdir(os.path.join(a.scratch, "expanded")):
        exp = os.path.join(a.scratch, "expanded")
        files = [os.path.join(d, f) for d, _, fs in os.walk(exp) for f in fs if f.endswith(".F90")]
        glue_dst = os.path.join(exp, "src", "mm", "dbcsr_amd_resident.F90")
        if os.path.exists(glue_dst):   # this repository's own module may have changed since: refresh it and what uses it
            shutil.copy(os.path.join(ROOT, "dbcsr_amd", "fortran", "dbcsr_amd_resident.F"), glue_dst)
            o = os.path.join(a.scratch, "build_" + a.variant, "dbcsr_amd_resident.o")
            if os.path.exists(o):
                os.remove(o)
    else:
        exp, files = expand_tree(a.scratch, a.patch)
    for v in (["cpu", "acc"] if a.variant == "both" else [a.variant]):
        build_variant(v, a.scratch, exp, files, a.jobs, a.reuse)


if __name__ == "__main__":
    main()

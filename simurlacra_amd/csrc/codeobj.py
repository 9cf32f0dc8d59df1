"""Register / scratch table of the built kernels, read from the gfx950 code objects inside build/*.o.

The library is ~1 100 kernel instantiations; whether one of them spills is not something to eyeball.  This module pulls
the AMDGPU metadata note (`llvm-readelf --notes`) out of every translation unit's device code object and returns one row per
kernel: vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size (scratch bytes
per lane), group_segment_fixed_size (static LDS), max_flat_workgroup_size.  tests/test_codeobj.py asserts on it (no
scratch in the kernels the BASELINE configs launch), `python -m simurlacra_amd.csrc.codeobj` prints it.
"""
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ_DIR = os.path.join(HERE, "build")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size", "uses_dynamic_stack")


def _tool(name):
    for cand in (os.path.join(LLVM_BIN, name), shutil.which(name)):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError(f"{name} not found (ROCm LLVM tools)")


def device_code_object(obj_path, out_path):
    """The gfx950 code object bundled inside a host object built by hipcc."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([_tool("llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj_path], check=True, capture_output=True)
        subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}",
                        f"--output={out_path}"], check=True, capture_output=True)
    return out_path


def _demangle(names):
    filt = shutil.which("c++filt") or _tool("llvm-cxxfilt")
    res = subprocess.run([filt], input="\n".join(names) + "\n", capture_output=True, text=True, check=True)
    return res.stdout.splitlines()


def kernels_of(obj_path):
    """[{name, demangled, <FIELDS>}] for every kernel of one translation unit."""
    with tempfile.TemporaryDirectory() as tmp:
        co = device_code_object(obj_path, os.path.join(tmp, "dev.co"))
        notes = subprocess.run([_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    rows, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S.*)$", line)
        if line.lstrip().startswith("- .agpr_count:") or (line.startswith("  - .") and cur is None):
            cur = {}
            rows.append(cur)
        if m and cur is not None:
            key, valtxt = m.group(1), m.group(2).strip()
            if key == "name" and "name" not in cur and not line.lstrip().startswith("- .name"):
                cur["name"] = valtxt
            elif key in FIELDS:
                cur[key] = (valtxt == "true") if valtxt in ("true", "false") else int(valtxt)
    rows = [r for r in rows if "name" in r and "vgpr_count" in r]
    for r, dm in zip(rows, _demangle([r["name"] for r in rows])):
        r["demangled"] = re.sub(r"\(vs::Task.*$", "", dm).replace("void ", "").replace("vs::", "")
    return rows


def disassemble(obj_path, mangled_name):
    """the ISA of one kernel of a translation unit (llvm-objdump of its gfx950 code object)"""
    with tempfile.TemporaryDirectory() as tmp:
        co = device_code_object(obj_path, os.path.join(tmp, "dev.co"))
        return subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={mangled_name}", co],
                              capture_output=True, text=True, check=True).stdout


def table(obj_dir=OBJ_DIR):
    rows = []
    for fn in sorted(os.listdir(obj_dir)):
        if fn.endswith(".o"):
            for r in kernels_of(os.path.join(obj_dir, fn)):
                r["unit"] = fn
                rows.append(r)
    return rows


def spills(rows):
    return [r for r in rows if r["vgpr_spill_count"] or r["sgpr_spill_count"] or r["private_segment_fixed_size"]
            or r.get("uses_dynamic_stack")]


if __name__ == "__main__":
    import sys

    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    rows = table()
    print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratchB':>8s} {'ldsB':>6s}")
    for r in rows:
        if pat and not pat.search(r["demangled"]):
            continue
        print(f"{r['demangled'][:110]:110s} {r['vgpr_count']:5d} {r['agpr_count']:5d} {r['sgpr_count']:5d} "
              f"{r['vgpr_spill_count']:6d} {r['sgpr_spill_count']:6d} {r['private_segment_fixed_size']:8d} "
              f"{r['group_segment_fixed_size']:6d}")
    bad = spills(rows)
    print(f"{len(rows)} kernels, {len(bad)} with scratch")

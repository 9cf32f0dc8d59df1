"""No kernel of libvecsim spills: read from the gfx950 code objects of the in-tree build (CPU-only, no GPU needed).

Round 2 shipped auto-reset kernels that spilled VGPRs to scratch inside the step loop (the cartpole's BASELINE config 3
kernel 35 spills / 144 B per lane, the ball balancer's 177 / 712 B, the headline kernel 8 / 36 B).  The cause was
loop-invariant address arithmetic of the rare reset / redraw blocks hoisted out of the step loop (`cold_lane` in
csrc/vecsim_kernels.h); this file keeps it from coming back:

  * the kernel `vs_step_random` launches for every BASELINE configuration: vgpr_spill_count == 0, private_segment_fixed_size == 0
    (or a dead frame object of a few bytes that no instruction of the kernel refers to, see below), and a VGPR budget that
    keeps the occupancy the launcher counts on;
  * every kernel of the library: not one scratch instruction in its ISA (a few kernels keep a dead 36-byte frame object
    from SGPR spill slots that were lowered to VGPR lanes -- the disassembly is the check that nothing touches it), and no
    VGPR spill except the in-kernel policy's deepest shapes, which spill into AGPRs, not memory.
"""
import os
import re
import subprocess
import tempfile

import pytest

from simurlacra_amd.csrc import build as vbuild
from simurlacra_amd.csrc import codeobj

pytestmark = pytest.mark.skipif(not os.path.isdir(codeobj.OBJ_DIR) or not os.listdir(codeobj.OBJ_DIR),
                                reason="no in-tree build objects (run __graft_entry__.build())")


@pytest.fixture(scope="module", autouse=True)
def fresh_build():
    """the code objects read here are those of the sources as they stand: a stale in-tree build is rebuilt first (what
    __graft_entry__.build() does; ~3 minutes on 8 cores, nothing when the build is current)"""
    if vbuild.is_stale():
        vbuild.build()


@pytest.fixture(scope="module")
def rows(fresh_build):
    return codeobj.table()


# (BASELINE config, demangled kernel name, VGPR ceiling): the automatic choice of Launch<E>::variant for that batch on 256 CUs,
# per-env constants (what bench.py and the samplers run), auto-reset, record mode 1; the last template argument (DRK) says
# whether the redraw of a live randomizer is compiled in: config 3's kernel only
SCRATCH_INSN = re.compile(r"\bscratch_(load|store)|\bbuffer_(load|store)\w* .*\boffen\b")
BASELINE_KERNELS = [
    ("headline: 65 536 QQube", "k_rollout_ws<QQT<0>, false, true, 1, 4, 256, false, 3, 0>", 168),   # three waves per SIMD
    ("config 2: 4 096 QQube", "k_rollout_ws<QQT<0>, false, true, 1, 4, 64, false, 3, 0>", 168),
    ("config 3: 65 536 cartpole + live DR", "k_rollout_ws<QcpT<0>, false, true, 1, 4, 64, false, 3, 1>", 168),
    ("config 4: 32 768 ball balancer (and 65 536)", "k_rollout_ws<Qbb, false, true, 1, 4, 64, false, 3, 0>", 168),
    ("ball balancer, two-role 64-env shape (one wave per SIMD by design)", "k_rollout_ws<Qbb, false, true, 1, 4, 64, false, 2, 0>", 512),
    ("ball-on-beam at 65 536", "k_rollout_ws<BobT<0>, false, true, 1, 4, 64, false, 3, 0>", 168),
    ("config 5: mixed batch", "k_rollout_mixed<true, 1, false>(Segs const*, int, unsigned long)", 128),    # four waves per SIMD
    ("config 1 / policy in the loop: oscillator step", "k_step<Omo, false, false, false, 0, false, false>", 128),
    ("large-N step", "k_step<QQT<0>, false, true, false, 0, false, true>", 128),
]


@pytest.mark.parametrize("what,name,vgpr_cap", BASELINE_KERNELS, ids=[b[0] for b in BASELINE_KERNELS])
def test_baseline_config_kernels_have_no_scratch(rows, what, name, vgpr_cap):
    hit = [r for r in rows if r["demangled"] == name]
    assert len(hit) == 1, f"{name}: {len(hit)} kernels of that name in the build"
    r = hit[0]
    assert r["vgpr_spill_count"] == 0, r
    assert not r.get("uses_dynamic_stack"), r
    assert r["vgpr_count"] + r["agpr_count"] <= vgpr_cap, r
    if r["private_segment_fixed_size"] != 0:
        # LLVM sometimes leaves a dead frame object behind when it lowers SGPR spills to VGPR lanes (8 .. 68 bytes, which
        # instantiations get one changes with unrelated edits): accepted only if NOT ONE instruction of this kernel touches scratch
        assert r["sgpr_spill_count"] > 0 and r["private_segment_fixed_size"] <= 68, r
        isa = codeobj.disassemble(os.path.join(codeobj.OBJ_DIR, r["unit"]), r["name"])
        assert isa.count("\n") > 100 and not SCRATCH_INSN.search(isa), (name, r["private_segment_fixed_size"])


def test_every_fused_kernel_family_is_spill_free(rows):
    """Every instantiation of the step / rollout kernels (all families, record modes, auto-reset on and off, both constant
    forms): no VGPR spill.  The in-kernel policy's four-layer shapes (512 registers, one wave per SIMD) are the exception:
    they overflow into AGPRs (private_segment_fixed_size stays 0)."""
    bad = []
    for r in rows:
        if r["vgpr_spill_count"] == 0:
            continue
        if r["demangled"].startswith("k_rollout_fnn<") and r["private_segment_fixed_size"] == 0:
            continue  # spilled into AGPRs
        bad.append((r["demangled"], r["vgpr_spill_count"], r["private_segment_fixed_size"]))
    assert not bad, bad


def test_wave_specialised_kernels_keep_their_occupancy(rows):
    """k_rollout_ws: the three-role shape needs three waves per SIMD (<= 168 VGPRs), the two-role shapes two (<= 256);
    with the rare blocks' addresses out of the loop they sit far below (QQube 107, cartpole 130, ball balancer 138)."""
    for r in rows:
        if not r["demangled"].startswith("k_rollout_ws<"):
            continue
        three = bool(re.search(r", 3, [012]>$", r["demangled"]))  # (.., NR, DRK>)
        assert r["vgpr_count"] <= (168 if three else 256), r
        assert r["vgpr_spill_count"] == 0, r


def test_no_kernel_executes_a_scratch_instruction():
    """The ISA of every translation unit: no scratch_load / scratch_store / buffer access off the scratch descriptor."""
    objdump = codeobj._tool("llvm-objdump")
    pat = SCRATCH_INSN
    for fn in sorted(os.listdir(codeobj.OBJ_DIR)):
        if not fn.endswith(".o"):
            continue
        with tempfile.TemporaryDirectory() as tmp:
            co = codeobj.device_code_object(os.path.join(codeobj.OBJ_DIR, fn), os.path.join(tmp, "dev.co"))
            isa = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
        hits = [ln for ln in isa.splitlines() if pat.search(ln)]
        assert not hits, (fn, hits[:5])

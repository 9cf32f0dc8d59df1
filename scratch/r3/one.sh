#!/bin/bash
# one.sh '<E>' '<template args>' [extra flags]  -> prints the register / scratch row of that k_rollout_ws instantiation
E="$1"; A="$2"; shift 2
O=${TMPDIR:-/tmp}/one_$$.o
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -fno-slp-vectorize -Wno-unused-function \
  "-DK_E=$E" "-DK_ARGS=$A" "$@" -c -o $O /root/repo/scratch/r3/one_kernel.hip || exit 1
python3 - "$O" <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from simurlacra_amd.csrc import codeobj
for r in codeobj.kernels_of(sys.argv[1]):
    print(f"{r['demangled'][:90]:90s} vgpr {r['vgpr_count']:3d} agpr {r['agpr_count']:3d} sgpr {r['sgpr_count']:3d} vspill {r['vgpr_spill_count']:3d} sspill {r['sgpr_spill_count']:3d} scratch {r['private_segment_fixed_size']:4d} lds {r['group_segment_fixed_size']}")
PY
[ -n "$KEEP" ] && cp $O $KEEP; rm -f $O

#!/bin/bash
# Where does the production encoder's product phase lie within a 64-byte line?  (csrc/lstm_coop.hip: the anchor in front of the step loop
# wants 44.)  Runs on the build container: hipcc cross-compiles, llvm-objdump disassembles.      bash tools/r06/loop_anchor_check.sh [extra hipcc flags]
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
cd $R/gnnpn-sc_amd/csrc
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I. -I../../include "$@" --cuda-device-only --no-gpu-bundle-output -c -o $T/lstm_coop.o lstm_coop.hip 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/lstm_coop.o > $T/dis.txt
python3 - $T/dis.txt <<'PY'
import re, sys
on, first = False, None
for l in open(sys.argv[1]):
    if re.match(r'^[0-9a-f]+ <_Z23lstm_encode_coop_kernelILi2ELb0ELb0E', l):
        on = True
        continue
    if on and re.match(r'^[0-9a-f]+ <', l):
        break
    m = re.search(r'//\s*([0-9A-Fa-f]+):', l) if on else None
    if m and 'v_mfma_f32_16x16x32' in l:
        first = int(m[1], 16)
        break
print(f"lstm_encode_coop_kernel<2,false,false>: first v_mfma_f32_16x16x32_f16 at {first:#x}: {first % 64} bytes into its 64-byte line (wanted: 44; 12 is as good solo)")
PY
rm -rf $T

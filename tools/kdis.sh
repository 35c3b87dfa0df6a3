# usage: bash tools/kdis.sh <lib.so> <kernel name filter> [out.s]  -- disassembly of one kernel of a built library + its LDS / MFMA / memory
# instruction mix (no GPU needed): which ds_* forms the compiler really emitted (a float2 access without known 8-byte alignment
# becomes ds_read2_b32 / ds_write2_b32 offset1:1, which the LDS serves as two 4-byte passes)
lib=${1:-molgym_amd/libmolgym_hip.so}; filt=$2; out=${3:-/tmp/kdis.s}
tmp=$(mktemp -d)
python3 - "$lib" "$tmp" <<'PY'
import sys, re
data = open(sys.argv[1], 'rb').read()
i = [m.start() for m in re.finditer(b'\x7fELF\x02\x01\x01@', data)][0]
open(sys.argv[2] + '/co.elf', 'wb').write(data[i:])
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $tmp/co.elf 2>/dev/null | awk -v f="$filt" '
  /^[0-9a-f]+ <.*>:/ {on = ($0 ~ f)} on {print}' > $out
grep -E "^[0-9a-f]+ <" $out
grep -oE "\b(ds_[a-z0-9_]+|v_mfma[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+|s_waitcnt|s_barrier|scratch_[a-z0-9_]+)\b" $out | sort | uniq -c | sort -rn
wc -l $out
rm -rf $tmp

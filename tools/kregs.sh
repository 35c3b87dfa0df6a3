# usage: bash tools/kregs.sh <lib.so> [name filter]  -- VGPRs / SGPRs / LDS / scratch of the kernels in a built library
# (reads the code object's metadata notes; no GPU needed)
lib=${1:-molgym_amd/libmolgym_hip.so}
filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$lib 2>/dev/null | head -0
python3 - "$lib" "$tmp" <<'PY'
import sys, subprocess, re
lib, tmp = sys.argv[1], sys.argv[2]
data = open(lib, 'rb').read()
# the fat binary holds an ELF per offload arch: find the gfx950 code object by its ELF magic after the bundle header
idx = [m.start() for m in re.finditer(b'\x7fELF\x02\x01\x01@', data)]
for k, i in enumerate(idx):
    open(f'{tmp}/co{k}.elf', 'wb').write(data[i:])
print(len(idx))
PY
for f in $tmp/co*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | awk '
    /\.name:/ {name=$2}
    /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
    /\.agpr_count:/ {a=$2}
    /\.wavefront_size:/ {printf "%-90s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s\n", name, v, a, s, l, p}' | grep -E "$filt"
done
rm -rf $tmp

cd $GRAFT_REPO_ROOT
for e in "$@"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMG_TS -DMG_EXP=$e molgym_amd/csrc/molgym_hip.hip -o /tmp/libmg_ts$e.so 2>&1 | grep -E "error" | head
echo "EXP $e"
timeout -k 5 200 python tools/ts_heads.py /tmp/libmg_ts$e.so cfg2 2>&1 | grep -A5 "natoms 7"
done

cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMG_TS $MG_XFLAGS molgym_amd/csrc/molgym_hip.hip -o /tmp/libmg_ts.so 2>&1 | grep -E "error" | head
timeout -k 5 200 python tools/ts_int_heads.py /tmp/libmg_ts.so 2>&1 | tail -24

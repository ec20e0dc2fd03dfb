python tools/attn_debug.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
python tools/attn_probe.py 20 2>&1 | grep -v amdgpu.ids

cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06_tw; mkdir -p $O; rm -f $O/margins.txt
VLR_MARGINS=$O/margins.txt timeout 2400 python -m pytest tests/test_hip_true_width.py tests/test_hip_depth.py -q -s -k "true_width or outlier" 2>&1 | grep -v amdgpu.ids > $O/pytest.txt
grep -E "oracle rounding|planted|passed|failed|Error|assert" $O/pytest.txt | head -40; cat $O/margins.txt

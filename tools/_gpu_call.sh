cd /root/repo
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06_lora3; mkdir -p $O
bash tools/gpu_lora.sh $O step "VLR_LORA_ROWS=0" "VLR_LORA_ROWS=1"

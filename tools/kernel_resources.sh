# Registers, scratch and occupancy of every kernel of a source file, as the compiler reports them (runs without a GPU):
#   bash tools/kernel_resources.sh ls-spa_amd/csrc/k_factor.hip [extra hipcc flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - -

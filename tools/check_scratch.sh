#!/bin/bash
# Lists kernels that use private (scratch) memory: a dynamically indexed private array or kernel-argument array, or a
# register spill.  Expected output: only the two step kernels' 24-byte spill of the integrator record in their prologue.
cd "$(dirname "$0")/../localregneuralde.jl_amd/csrc" || exit 1
for f in lrnde_kernels.hip lrnde_conv.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -fPIC -I../../include -Wno-unused-function -Wno-unused-value -Rpass-analysis=kernel-resource-usage -c -o /dev/null $f 2>&1 |
    grep -E "Function Name|ScratchSize" | paste - - | grep -v "ScratchSize \[bytes/lane\]: 0 " |
    sed -E 's/.*Function Name: (\S+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*/\2 bytes\/lane  \1/' | sort -u
done

# Compile-time variants of ONE kernel source for A/B runs on the GPU box:
#   bash tools/variants.sh cfconv_fused "-DMDG_BWD_THETA_BLOCKS=768" v768      -> mdgrad_amd/lib/variants/libmdgrad_hip_v768.so
# then   MDG_LIB=mdgrad_amd/lib/variants/libmdgrad_hip_v768.so python tools/kbench_cfconv.py --rows16
# (the other objects are the ones of the regular build: run `python -m mdgrad_amd.build` first)
set -e
SRC=$1; DEFS=$2; TAG=$3
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/mdgrad_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DEFS -c $R/mdgrad_amd/csrc/$SRC.hip -o $R/mdgrad_amd/lib/variants/${SRC}_$TAG.o 2>/dev/null
OBJS=$(ls $R/mdgrad_amd/lib/obj/*.o | grep -v "/$SRC.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mdgrad_amd/lib/variants/libmdgrad_hip_$TAG.so $OBJS $R/mdgrad_amd/lib/variants/${SRC}_$TAG.o
echo built $R/mdgrad_amd/lib/variants/libmdgrad_hip_$TAG.so

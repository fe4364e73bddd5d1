# A/B of the MFE kernel at several widths: product library vs every tools/abl_*.so (see tools/gpu_mfe_cmp.py)
# usage: bash tools/mfe_cmp.sh [n:W ...]
mkdir -p gpurun_out/cmp
out=gpurun_out/cmp/cmp_$(date +%H%M%S).txt
specs="$*"
[ -z "$specs" ] && specs="262144:120 131072:100 131072:128 65536:200 65536:160"
for spec in $specs; do
  python tools/gpu_mfe_cmp.py ${spec%%:*} ${spec##*:} 2>&1 | grep -v amdgpu.ids | tee -a $out
done

#!/bin/bash
# Builds an experimental engine variant next to the product library (never committed, not gated):
#   tools/build_variant.sh <name> [extra hipcc flags...]  ->  slam-pose_estimation_amd/lib/ab/<name>.so
# Used with tools/ab.sh for same-box A/B timing.  Same flags as csrc/Makefile (MachineLICM off for the kernel TUs only).
#   tools/build_variant.sh <name> --patch tools/variants/X.patch [flags...]  builds from a scratch copy of csrc/ with the patch
#   applied (`patch -p1` from the repo root's point of view): measured-and-rejected variants live as patches, not as
#   switches in the product headers.  e.g. the rejected four-wavefront variant of round 3:
#   tools/build_variant.sh compact64 --patch tools/variants/r03_rejected_switches.patch -DUKFB_COMPACT64=1 -DUKFB_LATE_XM=1
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/slam-pose_estimation_amd/csrc
if [ "$1" = "--patch" ]; then
  pf=$(cd "$(dirname "$2")" && pwd)/$(basename "$2"); shift 2
  scratch=$(mktemp -d)
  trap 'rm -rf "$scratch"' EXIT
  mkdir -p $scratch/slam-pose_estimation_amd $scratch/include
  cp -r $src $scratch/slam-pose_estimation_amd/csrc
  cp -r $root/include/. $scratch/include/
  (cd $scratch && patch -s -p1 < "$pf")
  src=$scratch/slam-pose_estimation_amd/csrc
fi
out=$root/slam-pose_estimation_amd/lib/ab
obj=$out/obj_$name
mkdir -p $obj
pids=""
common="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -DUKFB_GENERIC_F64=0"
for tu in ukf_batch ukf_group; do
  /opt/rocm/bin/hipcc $common "$@" -c $src/$tu.hip -o $obj/$tu.o 2> $obj/$tu.remarks &
  pids="$pids $!"
done
for tu in ukf_launch_pose_f64 ukf_launch_pose_f32 ukf_launch_orient_f64 ukf_launch_orient_f32 ukf_launch_pose_f32w ukf_launch_orient_f32w; do
  /opt/rocm/bin/hipcc $common -mllvm -disable-machine-licm "$@" \
      -Rpass-analysis=kernel-resource-usage -c $src/$tu.hip -o $obj/$tu.o 2> $obj/$tu.remarks &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
python3 $root/tools/check_resources.py $obj/ukf_launch_*.remarks | grep kernel16 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $obj/*.o -ldl -pthread
rm -rf $obj
echo "built $out/$name.so"

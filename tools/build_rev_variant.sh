#!/bin/bash
# Builds the engine of another git revision next to the product library, for same-box A/B timing with tools/ab.sh:
#   [REV_FLAGS="..."] tools/build_rev_variant.sh <name> <git-rev>  ->  slam-pose_estimation_amd/lib/ab/<name>.so
#   (REV_FLAGS: extra hipcc flags of that revision's Makefile; default = the current ones, empty for revisions before a2036cf)
set -e
name=$1; rev=$2
root=$(cd "$(dirname "$0")/.." && pwd)
wt=$(mktemp -d /tmp/ukfb_wt_XXXX)
git -C $root worktree add -f --detach $wt $rev > /dev/null 2>&1
src=$wt/slam-pose_estimation_amd/csrc
out=$root/slam-pose_estimation_amd/lib/ab; obj=$out/obj_$name; mkdir -p $obj
pids=""
tus="ukf_batch ukf_launch_pose_f64 ukf_launch_pose_f32 ukf_launch_orient_f64 ukf_launch_orient_f32"
[ -f $src/ukf_group.hip ] && tus="$tus ukf_group"     # (device groups: round 3 on)
[ -f $src/ukf_launch_pose_f32w.hip ] && tus="$tus ukf_launch_pose_f32w ukf_launch_orient_f32w"   # (wide arithmetic: round 4 on)
for tu in $tus; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-slp-vectorize -DUKFB_GENERIC_F64=0 ${REV_FLAGS--mllvm -disable-machine-licm} -c $src/$tu.hip -o $obj/$tu.o 2> $obj/$tu.log &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so $obj/*.o -ldl -pthread
rm -rf $obj; git -C $root worktree remove --force $wt
echo "built $out/$name.so from $rev"

# round 6, twelfth call: (a) the widened features in the fast flavour with this round's trees and root pass (Example-class scene; a crash or a NaN frame would show here),
# (b) the LDS split of the ray workgroups between stack entries and staged nodes with the new trees: 48 KB / 64 KB (default) / 80 KB of stack
out=gpurun_out/r06l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/gpu_feature_cost.sh > $out/feature_cost.txt 2>&1; cat $out/feature_cost.txt
for rep in 1 2; do
  WORKLOADS="hall scan" bash tools/gpu_ab_variants.sh $out/ab.txt default stack48 stack80
done

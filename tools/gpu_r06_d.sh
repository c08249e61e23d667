# round 6, fourth call: small switches re-measured on the new trees (two-triangle leaves, optimal collapse), and two ablations of k_shade (timing only)
#   vote11 / vote13: the phase vote's triangle : node ratio 1:1 / 1:3 (default 1:2)   refill48 / refill32: refill threshold (default 40)
#   allslots: the closest-hit rays fetch both leaf slots unconditionally   prefrand: a candidate's random pair requested one candidate ahead (k_shade)
#   ablroot1: ONE resampling lane in the light tree's root pass instead of eight (wrong images)   abllanes4: four candidates instead of eight (wrong images)
out=gpurun_out/r06d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  WORKLOADS="hall example" bash tools/gpu_ab_variants.sh $out/ab.txt default vote11 vote13 refill48 refill32 allslots prefrand
done
WORKLOADS="hall" bash tools/gpu_ab_variants.sh $out/ab.txt default ablroot1 abllanes4

# round 6, ninth call: what k_shade's parts cost, by leaving them out (wrong images, timing only; hall, 64 ids per pass, ms per 3 steps)
#   abl_root1: one resampling lane in the root pass   abl_nocand: the root pass, but no candidate   abl_nobsdf: candidates without their BSDF evaluation
#   abl_nolight: no light sampling at all (root pass + candidates)   abl_nobsdfdir: no BSDF-sampled light direction   abl_all: neither of those two nor a real bounce sample
out=gpurun_out/r06i; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WORKLOADS="hall" bash tools/gpu_ab_variants.sh $out/ab.txt default abl_root1 abl_nocand abl_nobsdf abl_nolight abl_nobsdfdir abl_all default
WORKLOADS="example" bash tools/gpu_ab_variants.sh $out/ab.txt default abl_root1 abl_nocand abl_nolight

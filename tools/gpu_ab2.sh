# compile-time x runtime A/B: bash tools/gpu_ab2.sh "<cxxflags>" "<env1>" "<env2>" ...
flags="$1"; shift
LUM_CXXFLAGS="$flags" python -m luminary_amd.build --force > /dev/null 2>&1
grep -E "k_trace|k_shadow_rays" -A12 luminary_amd/lib/obj/kernel_resource_usage.txt | grep -E "VGPRs:|Scratch|Occupancy" | head -3 | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | tr '\n' ' '; echo "[$flags]"
bash tools/gpu_ab_env.sh "$@"

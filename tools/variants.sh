#!/bin/bash
# Builds experiment variants of the library on the GPU box and measures each:
#   tools/variants.sh "<tag>|<EXTRA flags>|<env>|<measure command>" ...
# e.g. tools/variants.sh "rank8|-DCPOL_RANK_WAVES=8||python tools/stage_times.py --config c2"
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for spec in "$@"; do
  IFS='|' read -r tag extra envs cmd <<< "$spec"
  make -s -C cosmo_pol_amd/csrc clean >/dev/null
  if ! make -s -C cosmo_pol_amd/csrc EXTRA="$extra" > gpurun_out/build_$tag.log 2>&1; then echo "$tag BUILD FAILED"; tail -5 gpurun_out/build_$tag.log; continue; fi
  echo "== $tag [$extra] [$envs]"
  env $envs $cmd --tag "$tag" 2>>gpurun_out/variants.err | grep -E "^\{|^all|^rec|^ice|^melt|^void |^k_|^== " | cut -c1-700
done
make -s -C cosmo_pol_amd/csrc clean >/dev/null; make -s -C cosmo_pol_amd/csrc >/dev/null

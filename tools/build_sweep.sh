#!/bin/bash
# Builds the library with every "<tag>|<EXTRA flags>" given and runs the same stage_times.py measurements with each:
#   MEASURE="--config c3 --elev 3 --steps 40;--config c4 --volume --rays 45 --steps 12" tools/build_sweep.sh "base|" "wpe4|-DCPOL_LOOKUP_WPE=4" ...
# (restores the default build at the end)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
IFS=';' read -r -a cmds <<< "${MEASURE:---config c2 --steps 60}"
for spec in "$@"; do
  tag=${spec%%|*}; extra=${spec#*|}
  make -s -C cosmo_pol_amd/csrc clean >/dev/null
  if ! make -s -C cosmo_pol_amd/csrc EXTRA="$extra" > gpurun_out/build_$tag.log 2>&1; then echo "== $tag BUILD FAILED"; tail -5 gpurun_out/build_$tag.log; continue; fi
  echo "== $tag [$extra]"
  for cmd in "${cmds[@]}"; do
    python tools/stage_times.py $cmd --tag "$tag" 2>>gpurun_out/build_sweep.err | grep -E "^\{" | cut -c1-600
  done
done
make -s -C cosmo_pol_amd/csrc clean >/dev/null; make -s -C cosmo_pol_amd/csrc >/dev/null

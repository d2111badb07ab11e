#!/bin/bash
# Degree / panel width of the 1-D integral tables: builds the library for every "<degree>:<panels per octave>" given and
# measures the accuracy gate (itab_check per slot, table build time) and the sweeps that gather 1-D blocks:
#   tools/itab_degree_sweep.sh 10:8 6:32 7:16 ... > gpurun_out/r5_itab1_degree.txt
# (restores the default build at the end)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for spec in "$@"; do
  deg=${spec%%:*}; ppo=${spec##*:}
  make -s -C cosmo_pol_amd/csrc clean >/dev/null
  if ! make -s -C cosmo_pol_amd/csrc EXTRA="-DCPOL_ITAB1_DEGREE=$deg -DCPOL_ITAB_PPO=$ppo" > gpurun_out/build_itab_${deg}_${ppo}.log 2>&1; then
    echo "== degree $deg ppo $ppo BUILD FAILED"; tail -5 gpurun_out/build_itab_${deg}_${ppo}.log; continue
  fi
  echo "== degree $deg, $ppo panels per octave (rows per block $((deg + 1)), bytes gathered per item $(( (deg + 1) * 96 )))"
  for cmd in "--config c2 --steps 60" "--config c3 --elev 3 --steps 40" "--config c4 --volume --rays 45 --steps 12" "--config c4 --volume --steps 5"; do
    python tools/stage_times.py $cmd --tag "d${deg}p${ppo}" 2>>gpurun_out/itab_degree_sweep.err | grep -E "^\{" | cut -c1-600
  done
done
make -s -C cosmo_pol_amd/csrc clean >/dev/null; make -s -C cosmo_pol_amd/csrc >/dev/null

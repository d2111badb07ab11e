cd /tmp && export TMPDIR=/tmp && export CPOL_GATE1_RAY=1
rm -rf /tmp/pmcT
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr --kernel-trace --output-format csv -d /tmp/pmcT -- python3 $GRAFT_REPO_ROOT/tools/stage_times.py --config c2 --steps 30 > $GRAFT_REPO_ROOT/gpurun_out/r6_pmc_l1.log 2>&1
f=$(find /tmp/pmcT -name "*counter_collection.csv" | head -n 1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    if k in ('k_gate1_ray','k_interp_sweep','k_scan_rays'):
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    print(k, {n: round(sum(v)/len(v),1) for n,v in c.items()})
PY

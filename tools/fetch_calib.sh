#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run on the GPU box):
#   tools/fetch_calib.sh   -> gpurun_out/fetch_calib.txt
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fcF /tmp/fcW
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fcF -- $root/tools/microbench/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/fcW -- $root/tools/microbench/fetch_calib > /dev/null 2>&1
python3 - <<PY > $out/fetch_calib.txt
import csv, glob, collections
def pmc(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(row['Kernel_Name'], row['Counter_Name'])].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, W = pmc('/tmp/fcF'), pmc('/tmp/fcW')
GiB = 1 << 30
print('kernel                 FETCH_SIZE [KiB]   / true bytes read      WRITE_SIZE [KiB]   / true bytes written')
for k in ('k_read4', 'k_read8', 'k_read16', 'k_gather16', 'k_gather4', 'k_sread96', 'k_write4', 'k_write8', 'k_write16'):
    f = [v for (n, c), v in F.items() if n.startswith(k + '(') and c == 'FETCH_SIZE']
    w = [v for (n, c), v in W.items() if n.startswith(k + '(') and c == 'WRITE_SIZE']
    true_r = (GiB // 128 * 96 if k == 'k_sread96' else GiB) if 'read' in k else (GiB // 128 * (16 if k == 'k_gather16' else 4) if 'gather' in k else 0)
    true_w = GiB if 'write' in k else 0
    fr = f[0] * 1024 if f else float('nan'); wr = w[0] * 1024 if w else float('nan')
    print('%-20s %14.0f   %8.3f of the bytes   %14.0f   %8.3f' % (k, f[0] if f else -1, fr / true_r if true_r else float('nan'), w[0] if w else -1, wr / true_w if true_w else float('nan')))
print('(gathers: true bytes = the 16 / 4 bytes the lanes asked for; one 128-B line each = %d bytes of lines)' % GiB)
PY
cat $out/fetch_calib.txt

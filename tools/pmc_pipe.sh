#!/bin/bash
# PMC counters + HBM traffic of one forward-conv shape (three separate rocprofv3 passes):  bash tools/pmc_pipe.sh N Cin Cout H W [env...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape="$1 $2 $3 $4 $5"; shift 5
for e in "$@"; do export "$e"; done
rm -rf /tmp/ps /tmp/pf /tmp/pw
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/ps -o p -- python tools/time_conv.py $shape > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python tools/time_conv.py $shape > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- python tools/time_conv.py $shape > /dev/null 2>&1
python - $shape "$@" <<'PY'
import csv, sys
N, Cin, Cout, H, W = [int(v) for v in sys.argv[1:6]]
def rows(path):
    return [r for r in csv.DictReader(open(path)) if 'conv2d_fwd' in r['Kernel_Name']]
def med(path, counter):
    v = [(float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows(path) if r['Counter_Name'] == counter]
    v.sort(); return v[len(v) // 2]
rs = rows('/tmp/ps/p_counter_collection.csv')
last = {}
for r in rs: last[r['Counter_Name']] = (float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'][:70], r['VGPR_Count'], r['LDS_Block_Size'], r['Grid_Size'])
v = {k: x[0] for k, x in last.items()}; d = list(last.values())[0]
cyc = v['SQ_BUSY_CYCLES'] / 32
f, d1 = med('/tmp/pf/p_counter_collection.csv', 'FETCH_SIZE'); w, d2 = med('/tmp/pw/p_counter_collection.csv', 'WRITE_SIZE')
alg = N * H * W * (Cin + Cout) * 2
print('==', sys.argv[1:])
print(d[2], 'dur_us %.0f' % (d[1] / 1e3), 'vgpr', d[3], 'lds', d[4], 'grid', d[5])
print('  clock GHz %.2f' % (cyc / d[1]), 'mfma_busy %.3f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc), 'wait_any %.3f' % (v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']),
      'wait_inst %.3f' % (v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']), 'wait_lds %.3f' % (v['SQ_WAIT_INST_LDS'] / v['SQ_WAVE_CYCLES']),
      'lds_busy %.3f' % (v['SQ_LDS_IDX_ACTIVE'] / 256 / cyc), 'lds_conflict_frac %.3f' % (v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)))
print('  fetch_MB %.0f (x2 corr)' % (2 * f / 1024), 'write_MB %.0f' % (w / 1024), 'algorithmic_MB %.0f' % (alg / 1e6), 'traffic/alg %.2f' % ((2 * f + w) * 1024 / alg),
      'HBM TB/s (profiled run) %.2f' % ((2 * f + w) * 1024 / d1 / 1e3))
PY

#!/bin/bash
# PMC counters of the filtered_lrelu kernels on one StyleGAN3 layer (separate pass from timing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/pmc; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc -o p -- python tools/flr_one.py "$@" 1 > /dev/null 2>&1
  rm -rf /tmp/pmc2; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python tools/flr_one.py "$@" 1 > /dev/null 2>&1
  echo "== $@"
  python - <<'PY'
import csv
for f in ['/tmp/pmc/p_counter_collection.csv', '/tmp/pmc2/p_counter_collection.csv']:
    rows=[r for r in csv.DictReader(open(f)) if 'flr' in r['Kernel_Name'] or 'filtered' in r['Kernel_Name']]
    bykern={}
    for r in rows: bykern.setdefault(r['Kernel_Name'][:60], {})[r['Counter_Name']] = (float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['VGPR_Count'], r['LDS_Block_Size'], r['Grid_Size'])
    for k, v in bykern.items():
        d = list(v.values())[0]
        print(k, 'dur_us %.1f' % (d[1]/1e3), 'vgpr', d[2], 'lds', d[3], 'grid', d[4])
        print('   ', {n: x[0] for n, x in v.items()})
PY
}
run "$@"

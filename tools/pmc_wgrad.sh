#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/pmc; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc -o p -- python tools/conv_one.py "$@" 2 > /dev/null 2>&1
  echo "== $@ DL=$AGF_WGRAD_DL"
  python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/tmp/pmc/p_counter_collection.csv')) if 'conv2d' in r['Kernel_Name']]
last={}
for r in rows: last[r['Counter_Name']]=(float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'][:60], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Grid_Size'])
v={k:x[0] for k,x in last.items()}
d=list(last.values())[0]
print(d[2], 'dur_us', d[1]/1e3, 'vgpr', d[3], 'agpr', d[4], 'lds', d[5], 'grid', d[6])
cyc=v['SQ_BUSY_CYCLES']/32
print('clock GHz %.2f'%(cyc/d[1]), 'mfma_util %.3f'%(v['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc), 'wait_any %.3f'%(v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']), 'wait_inst %.3f'%(v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']), 'wait_lds %.3f'%(v['SQ_WAIT_INST_LDS']/v['SQ_WAVE_CYCLES']), 'lds_busy %.3f'%(v['SQ_LDS_IDX_ACTIVE']/256/cyc), 'lds_conflict_frac %.3f'%(v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1)))
PY
}
export AGF_WGRAD_DL=0; run 128 128 128 128 128 wgrad
export AGF_WGRAD_DL=1; run 128 128 128 128 128 wgrad

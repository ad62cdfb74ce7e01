#!/bin/bash
# Per-phase PMC picture of the register-blocked filtered_lrelu kernels: VALU / LDS / SALU instruction counts, VALU and LDS busy
# cycles, bank conflicts, for builds that leave phases out (see tools/flr_phases.sh).  Counters in separate passes, no other tracing.
#   tools/flr_phases_pmc.sh <out.txt> <layer> <fwd|bwd> [masks...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; layer=$2; mode=$3; shift 3
masks=${@:-0 11 14 13 7}
mkdir -p $(dirname $out); : > $out
for sk in $masks; do
  touch animeface_amd/csrc/agf_filtered_lrelu.hip
  AGF_EXTRA_CXXFLAGS="-DAGF_PROFILE_PHASES=$sk" bash animeface_amd/csrc/build.sh > /dev/null 2>&1
  echo "##### phases left out: mask $sk (1 load, 2 up-FIR, 4 act, 8 down-FIR)" >> $out
  rm -rf /tmp/pmc; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc -o p -- python tools/flr_one.py $layer $mode 1 16 > /dev/null 2>&1
  rm -rf /tmp/pmc2; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python tools/flr_one.py $layer $mode 1 16 > /dev/null 2>&1
  python - >> $out <<'PY'
import csv, collections
for f in ['/tmp/pmc/p_counter_collection.csv', '/tmp/pmc2/p_counter_collection.csv']:
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'flr' not in r['Kernel_Name']: continue
        k = r['Kernel_Name'][:52]
        d = by.setdefault(k, {'dur': [], 'c': {}})
        d['c'].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        d['dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, d in by.items():
        print(k, 'dur_us %.1f' % (sum(d['dur']) / len(d['dur']) / 1e3), {n: round(sum(v) / len(v) / 1e6, 2) for n, v in d['c'].items()})
PY
done
touch animeface_amd/csrc/agf_filtered_lrelu.hip
bash animeface_amd/csrc/build.sh > /dev/null 2>&1
cat $out

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "variants_vs_aten or sign_bits or mask_vs_composite or run_to_run or post_scale" 2>&1 | tail -1)
(cd $R && AGF_SK_TILE=1 AGF_SK_T=1024 AGF_SK_C=2 timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "variants_vs_aten or sign_bits or mask_vs_composite or run_to_run" 2>&1 | tail -1)
run() { label=$1; shift
  rm -rf /tmp/prof_$label
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -o p -- python $R/tools/time_conv.py $SHAPE > /dev/null 2>&1
  f=$(find /tmp/prof_$label -name "*kernel_stats.csv" | head -1)
  python - "$f" "$SHAPE $label" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv2d_fwd_kernel' in r['Name']:
        print(sys.argv[2], 'calls', r['Calls'], 'avg_us %.2f' % (float(r['AverageNs']) / 1000), r['Name'][:70])
P
}
for SHAPE in "64 512 512 4 4" "128 512 512 4 4" "64 512 512 8 8" "128 512 512 8 8" "16 512 512 8 8"; do
  run base AGF_SK=0 AGF_PF2=0
  run pf2 AGF_SK=0 AGF_PF2=1
  for t in 256 512 1024; do
    run pf2_t1_T${t}_sc1 AGF_SK_TILE=1 AGF_SK_T=$t AGF_SK_C=2 AGF_SK_SC1=1
    run pf2_t1_T${t}_rel AGF_SK_TILE=1 AGF_SK_T=$t AGF_SK_C=2 AGF_SK_SC1=0
  done
  run t0_T512_sc1 AGF_SK_TILE=0 AGF_SK_T=512 AGF_SK_C=4 AGF_SK_SC1=1
  run t2_T512_sc1 AGF_SK_TILE=2 AGF_SK_T=512 AGF_SK_C=4 AGF_SK_SC1=1
done

# A/B of the channel-sliced small-map launches (probe build: AGF_SK* read from the environment)
for tile in 0 1 2; do AGF_SK_TILE=$tile timeout 600 python -m pytest tests/test_hip_conv.py -x -q -m gpu -k "variants_vs_aten or sign_bits or mask_vs_composite or run_to_run" 2>&1 | tail -1; done
for sh in "64 512 512 4 4" "128 512 512 4 4" "64 512 512 8 8" "128 512 512 8 8" "128 520 512 4 4"; do
  echo "== $sh"; echo -n "off: "; AGF_SK=0 python tools/time_conv.py $sh 2>/dev/null
  for tile in 0 1 2; do for t in 256 512 1024; do for c in 2 4; do echo -n "tile=$tile T=$t C=$c: "; AGF_SK_TILE=$tile AGF_SK_T=$t AGF_SK_C=$c python tools/time_conv.py $sh 2>/dev/null; done; done; done
done

for s in "128 512 512 16 16" "128 128 128 64 64" "128 64 64 128 128"; do
  for d in 0 3 4 5; do AGF_WGRAD_RING_DBG=$d python tools/time_wgrad.py $s 2>&1 | grep shape | sed "s/^/dbg$d /"; done
done

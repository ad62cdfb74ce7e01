cd $GRAFT_REPO_ROOT
for shape in "128 128 128 128 128" "128 256 256 64 64" "128 64 128 128 128"; do
  for m in "" bias demod mask "" bias; do
    python tools/time_conv.py $shape $m 2>&1 | tail -1
  done
done

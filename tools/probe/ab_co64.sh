for shape in "128 128 128 128 128" "128 256 256 64 64" "128 512 512 32 32" "64 256 256 64 64" "128 128 256 64 64"; do
  for mode in "" "maskbits" "bias"; do
    for lib in shipped co64; do
      echo -n "lib=$lib "; if [ $lib = shipped ]; then python tools/time_conv.py $shape $mode 2>&1 | tail -1; else AGF_PROBE_LIB=$lib python tools/time_conv.py $shape $mode 2>&1 | tail -1; fi
    done
  done
done

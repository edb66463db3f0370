# pass times with and without the row padding for other shapes (one process each)
for shape in "256 1024 1024" "1024 576 4096" "1024 1024 2048" "128 512 512"; do
  for pads in "default" "0,0"; do
    if [ "$pads" = "0,0" ]; then export MI_FFT_XPAD=0 MI_FFT_ZPAD=0; else unset MI_FFT_XPAD MI_FFT_ZPAD; fi
    echo "shape $shape pads $pads"; timeout -k 10 200 python profiles/shape_time.py $shape 2>&1 | tail -6
  done
done

for n in "4 4" "5 4" "5 5" "6 5" "6 6"; do
  for k in auto wide; do
    echo "n=$n kernel=$k"; FF_WIDE_ELOC_FROM=99 FF_ELOC_KERNEL=$k timeout 200 python tools/probes/wide_c5.py $n 2 16384 3 2>&1 | tail -1 | sed 's/.*eloc=/eloc=/'
  done
done

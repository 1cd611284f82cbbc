for lib in "" nowload noload nok; do
  for shape in "63 64 64 64" "31 32 128 128"; do
    set -- $shape
    if [ -n "$lib" ]; then export MANTLE_LIB=pbml_mantle_convection_amd/build/lib_$lib.so; else unset MANTLE_LIB; fi
    echo "== ${lib:-base} $shape"
    MC_CONV_RR=0 python tools/bench_conv.py --dtype mixed --hw $1 $2 --cin $3 --cout $4 --which fwd,dgrad --iters 20 2>&1 | grep -E "^(fwd|dgrad)" | cut -c1-70
  done
done

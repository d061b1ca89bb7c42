for lib in sqlrs_amd/csrc/libsqlrs_hip.so tools/_bin/lib_f_w8_c16.so tools/_bin/lib_f_w4_c16.so tools/_bin/lib_f_w4_c32.so; do
  for occ in 1 2 3 4; do
    for s in 0.5 0.01; do
      echo -n "occ $occ "; LIB=$lib SQLRS_FILTER_OCC=$occ SEL=$s REPS=5 python tools/c2_filter.py 2>&1 | grep "^C2"
    done
  done
done

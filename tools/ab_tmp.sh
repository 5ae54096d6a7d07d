set -e
cd $GRAFT_REPO_ROOT
for g in 1 2; do make -s -C pixell.jl_amd/csrc OUT=/tmp/libpxl_g$g.so EXTRA=-DPXL_UW_G=$g; done
for rep in 1 2 3; do
  for g in 4 1 2; do
    lib=/tmp/libpxl_g$g.so; [ $g = 4 ] && lib=$GRAFT_REPO_ROOT/pixell.jl_amd/libpixell_hip.so
    echo -n "G=$g: "; PXL_LIB_PATH=$lib python tools/prof_unwind.py 2>/dev/null | awk '{print $(NF-1)}' | tr '\n' ' '; echo
  done
done

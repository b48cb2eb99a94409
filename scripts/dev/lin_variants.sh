# usage: lin_variants.sh [variant ...]   (libraries build_variants/libbear_<variant>.so; PAIRED=1 for the paired lists)
vars="${@:-LIN_MIX LIN_NOSYNC LIN_SKIP_C LIN_SKIP_A LIN_SKIP_B LIN_SKIP_TRIPLE}"
timeout -k 10 120 python scripts/dev/lin_variants.py 2>/dev/null | tail -1
for n in $vars; do BEAR_AMD_LIB=$PWD/build_variants/libbear_$n.so timeout -k 10 120 python scripts/dev/lin_variants.py 2>/dev/null | tail -1; done
timeout -k 10 120 python scripts/dev/lin_variants.py 2>/dev/null | tail -1

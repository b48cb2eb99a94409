# usage: lin_variants.sh [variant ...]   (libraries under build_variants/, see scripts/dev/lin_variants.py)
vars="${@:-linskipbc linskipc lindbg1 lindbg2 lindbg4}"
for n in $vars; do BEAR_AMD_LIB=$PWD/build_variants/libbear_$n.so timeout -k 10 120 python scripts/dev/lin_variants.py 2>/dev/null | tail -1; done; timeout -k 10 120 python scripts/dev/lin_variants.py 2>/dev/null | tail -1

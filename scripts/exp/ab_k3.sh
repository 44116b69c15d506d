# usage: bash scripts/exp/ab_k3.sh "libA.so libB.so" spec...   -- k3.py with each build (""= in-tree), alternating, 2 rounds
LIBS=$1; shift
for i in 1 2; do
  for lib in in-tree $LIBS; do
    echo "== $lib"
    if [ "$lib" = in-tree ]; then timeout -k 10 300 python scripts/exp/k3.py "$@" || exit 1
    else CFD_LIB=$PWD/$lib timeout -k 10 300 python scripts/exp/k3.py "$@" || exit 1; fi
  done
done

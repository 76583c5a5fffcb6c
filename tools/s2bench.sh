for spec in "128 56" "256 28" "512 14"; do set -- $spec; C=$1; W=$2
  python tools/kbench.py s2 $C $C $W --stride 2 --pro 2 --epi 1 --iters 30 | sed "s/^/fwd  /"
  python tools/kbench.py s2 $C $C $W --stride 2 --pro 0 --epi 2 --mode 2 --iters 30 | sed "s/^/dgrad /"
  python tools/kbench.py igemm $C $C $W --stride 2 --pro 2 --epi 1 --iters 30 | sed "s/^/igemm fwd /"
done

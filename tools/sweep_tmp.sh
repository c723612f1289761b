for fp in 100 150 200 300; do
  bash tools/bench_one.sh fp$fp c3 --option first_pass_permille=$fp || exit 1
  bash tools/bench_one.sh fp$fp c4 --option first_pass_permille=$fp || exit 1
  bash tools/bench_one.sh fp$fp c5 --base-rows 1250000 --queries 12500 --steps 8 --option first_pass_permille=$fp || exit 1
done

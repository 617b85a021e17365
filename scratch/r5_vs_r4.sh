# the same shapes through the round-4 tree (_r4, its own python layer and library) and through this one, on one box
for s in "1000000 3" "300000 26" "600000 12" "300000 40" "400000 30" "2000000 5" "200000 16" "50000 10" "1000000 10"; do
  set -- $s
  echo "SHAPE $1 x $2 r4" >> gpurun_out/r5_vs_r4.txt
  (cd _r4 && timeout 300 python3 scratch/seg_bench.py $1 $2 1 2>/dev/null | grep SEG >> ../gpurun_out/r5_vs_r4.txt)
  echo "SHAPE $1 x $2 r5" >> gpurun_out/r5_vs_r4.txt
  timeout 300 python3 scratch/seg_bench.py $1 $2 1 2>/dev/null | grep SEG >> gpurun_out/r5_vs_r4.txt
done

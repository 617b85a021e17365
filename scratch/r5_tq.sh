for lib in libdcdensity.so variants/nntq4.so variants/nntq4x.so; do
  for wt in 0 20000; do
    echo "LIB=$lib WT=$wt" >> gpurun_out/r5_tq.txt
    export DC_LIB_PATH=$PWD/clustering_amd/lib/$lib
    if [ $wt -eq 0 ]; then unset DC_WAVE_TARGET; else export DC_WAVE_TARGET=$wt; fi
    python3 scratch/seg_bench.py 1000000 10 8 2>/dev/null | grep SEG >> gpurun_out/r5_tq.txt
  done
  unset DC_WAVE_TARGET
  python3 scratch/seg_bench.py 1000000 10 1 2>/dev/null | grep SEG >> gpurun_out/r5_tq.txt
  python3 scratch/nn_counts.py 2>/dev/null | grep CNT >> gpurun_out/r5_tq.txt
done
